import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_counter_collection.csv'))[-1]
shapes = [(2, 256, 256, 256, 256), (2, 128, 128, 256, 256), (2, 64, 64, 256, 256), (2, 32, 32, 256, 256), (256, 14, 14, 256, 256), (512, 7, 7, 256, 256), (2, 256, 256, 64, 64), (2, 128, 128, 128, 128), (2, 32, 32, 512, 512)]
i = 0
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == 'FETCH_SIZE' and 'k_wino_input' in r['Kernel_Name']:
        N, H, W, Ci, Co = shapes[i]; i += 1
        inp = N * H * W * Ci * 4 / 1e6
        print('%-28s %-30s fetch %8.1f MB  input %8.1f MB  ratio %.3f' % (shapes[i-1], r['Kernel_Name'][:30], float(r['Counter_Value']) * 2048 / 1e6, inp, float(r['Counter_Value']) * 2048 / 1e6 / inp))
