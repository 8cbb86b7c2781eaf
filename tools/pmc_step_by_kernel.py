"""Per-step HBM traffic BY KERNEL NAME from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/step_pmc_run.py:
pmc_step_by_kernel.py <fetch_dir> <write_dir> <steps> - MB per step, launches, fetch : write, largest first."""
import csv, glob, re, sys, collections


def load(d, counter):
    f = sorted(glob.glob(d + '/*/*_counter_collection.csv'))[-1]
    acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '')
        acc[n] += float(r['Counter_Value']); cnt[n] += 1
    return acc, cnt


fetch, cnt = load(sys.argv[1], 'FETCH_SIZE')
write, _ = load(sys.argv[2], 'WRITE_SIZE')
steps = float(sys.argv[3])
rows = sorted(((fetch[k] * 2048 + write.get(k, 0.0) * 1024) / steps, k) for k in fetch)
tot = sum(r[0] for r in rows)
print('whole step: %.1f GB' % (tot / 1e9))
for b, k in reversed(rows[-45:]):
    print('%-56s n=%5.1f  fetch %9.1f MB  write %9.1f MB  total %9.1f MB' % (k[:56], cnt[k] / steps, fetch[k] * 2048 / steps / 1e6, write.get(k, 0.0) * 1024 / steps / 1e6, b / 1e6))
