"""Per-step HBM traffic of the kernel families from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py.
FETCH_SIZE is doubled (MI355X_MICROARCH.md: on gfx950 it reports half the bytes of wide coalesced reads); both counters are
in KiB.  usage: pmc_step_traffic.py <fetch_dir> <write_dir> <steps_profiled> > profiles/r02_step_pmc_traffic.json"""
import csv, glob, json, re, sys, collections


def load(d, counter):
    f = sorted(glob.glob(d + '/*/*_counter_collection.csv'))[-1]
    acc = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '')
        acc[n] += float(r['Counter_Value'])
        cnt[n] += 1
    return acc, cnt


fetch, cnt = load(sys.argv[1], 'FETCH_SIZE')
write, _ = load(sys.argv[2], 'WRITE_SIZE')
steps = float(sys.argv[3])
fam = {'conv_gemm': lambda n: 'k_conv_igemm' in n or 'k_pgemm' in n, 'winograd_transforms': lambda n: 'k_wino' in n,
       'slab_tail_column_sums': lambda n: any(k in n for k in ('k_sum_slabs', 'k_tail_sum', 'k_colsum')),
       'batchnorm': lambda n: 'k_bn_' in n, 'roi_align': lambda n: 'k_roi_align' in n}
out = {}
for name, pred in fam.items():
    ks = [k for k in fetch if pred(k)]
    fb = sum(fetch[k] for k in ks) * 1024 * 2 / steps
    wb = sum(write.get(k, 0.0) for k in ks) * 1024 / steps
    out[name] = {'launches_per_step': round(sum(cnt[k] for k in ks) / steps, 1), 'fetch_bytes_per_step': fb, 'write_bytes_per_step': wb,
                 'hbm_bytes_per_step': fb + wb}
conv = [out[k]['hbm_bytes_per_step'] for k in ('conv_gemm', 'winograd_transforms', 'slab_tail_column_sums')]
out['conv_bracket'] = {'hbm_bytes_per_step': sum(conv), 'note': 'GEMM launches + Winograd transforms + slab / tail / column sums'}
out['_method'] = 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 tools/step_pmc_run.py %d` (that many configs[2] steps and nothing else); FETCH_SIZE x 2 x 1024, WRITE_SIZE x 1024, summed over every launch of the family, divided by the steps' % int(steps)
print(json.dumps(out, indent=1))
