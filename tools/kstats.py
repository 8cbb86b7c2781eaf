"""Average duration of selected kernels from a rocprofv3 --stats run: kstats.py <dir> <substring> [<substring> ...]"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Name']).split('(')[0].replace('void ', '')
    if any(k in n for k in sys.argv[2:]):
        print('%-50s calls %5s  avg %9.1f us  total %9.1f us' % (n[:50], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3))
