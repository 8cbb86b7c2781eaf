"""Kernel-by-kernel listing of a window of one step from a rocprofv3 kernel trace of bench.py (steps delimited by k_sgd):
trace_window.py <dir> <step> <from_ms> <to_ms> [queue]  - start offset, duration, queue, name."""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in csv.DictReader(open(f)))
sg = [e for e in ev if 'k_sgd' in e[2]]
k, a, z = int(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
q = sys.argv[5] if len(sys.argv) > 5 else None
t0 = sg[k][1]
for e in ev:
    o = (e[0] - t0) / 1e6
    if a <= o < z and (q is None or e[3] == q):
        n = re.sub(r'\(anonymous namespace\)::|void ', '', e[2]).split('(')[0]
        print('+%8.3f ms  %7.1f us  q%s  %s' % (o, (e[1] - e[0]) / 1e3, e[3], n[:70]))
