"""Per-instance listing of selected kernels of ONE step from a rocprofv3 kernel trace of bench.py (steps are delimited by k_sgd):
trace_kernels.py <dir> <step index> <substring> [...]: duration, grid size, stream (queue) of every matching launch, in time order,
plus how much of its duration other kernels were running beside it."""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0), r.get('Queue_Id', '?')) for r in rows)
sg = [e for e in ev if 'k_sgd' in e[2]]
k = int(sys.argv[2])
t0, t1 = sg[k][1], sg[k + 1][1]
ks = [e for e in ev if e[0] >= t0 and e[1] <= t1]
pats = sys.argv[3:]
for s, e, n, g, q in ks:
    nm = re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0].replace('void ', '')
    if not any(p in nm for p in pats):
        continue
    ov = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2, n2, _, _ in ks if (s2, e2, n2) != (s, e, n) and s2 < e and e2 > s)
    print('%9.1f us  t=%8.1f  %-44s grid %9d  queue %s  others alongside %5.0f%%' % ((e - s) / 1e3, (s - t0) / 1e3, nm[:44], g, q, 100.0 * ov / max(e - s, 1)))
