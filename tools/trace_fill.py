"""Where is the chip under-filled?  From a rocprofv3 kernel trace of bench.py (steps delimited by k_sgd): every kernel's fill =
min(1, workgroups / 256 CUs); the timeline of the SUM of the fills of the kernels in flight; the windows in which it stays below
a threshold, longest first, with the kernels that ran in them.  usage: trace_fill.py <dir> [step index] [threshold]"""
import csv, glob, re, sys, collections
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
def wgs(r):
    g = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) * max(1, int(r.get('Grid_Size_Y', 1) or 1)) * max(1, int(r.get('Grid_Size_Z', 1) or 1))
    w = int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1) * max(1, int(r.get('Workgroup_Size_Y', 1) or 1)) * max(1, int(r.get('Workgroup_Size_Z', 1) or 1))
    return max(1, g // max(1, w))
def short(n):
    m = re.search(r'(k_[a-z0-9_]+|[A-Za-z_]+Buffer[A-Za-z]*|elementwise_kernel)', n)
    return m.group(1) if m else n[:40]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), min(1.0, wgs(r) / 256.0), r['Queue_Id']) for r in rows)
sg = [e for e in ev if 'k_sgd' in e[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.35
t0, t1 = sg[k][1], sg[k + 1][1]
ks = [e for e in ev if e[0] >= t0 and e[1] <= t1]
pts = sorted([(e[0], e[3]) for e in ks] + [(e[1], -e[3]) for e in ks])
segs, cur, last = [], 0.0, t0
for t, d in pts:
    if t > last: segs.append((last, t, cur))
    cur += d; last = t
if t1 > last: segs.append((last, t1, cur))
low = sum(b - a for a, b, c in segs if c < thr - 1e-9)
idle = sum(b - a for a, b, c in segs if c < 1e-9)
print('step %.2f ms; fill < %.2f for %.2f ms of it (no kernel at all: %.2f ms); time-average fill %.2f' % ((t1 - t0) / 1e6, thr, low / 1e6, idle / 1e6,
      sum((b - a) * min(c, 1.0) for a, b, c in segs) / (t1 - t0)))
# merge consecutive low segments into windows
wins, cs = [], None
for a, b, c in segs:
    if c < thr - 1e-9:
        if cs is None: cs = [a, b]
        else: cs[1] = b
    elif cs is not None and b - a > 2000:       # a filled stretch of > 2 us ends the window
        wins.append(tuple(cs)); cs = None
if cs is not None: wins.append(tuple(cs))
print('%d windows; the longest:' % len(wins))
for a, b in sorted(wins, key=lambda w: w[0] - w[1])[:14]:
    inw = collections.Counter()
    for e in ks:
        ov = min(e[1], b) - max(e[0], a)
        if ov > 0: inw[e[2] + ('@q' + e[4])] += ov
    print('  +%6.2f ms  %7.1f us  %s' % ((a - t0) / 1e6, (b - a) / 1e3, ', '.join('%s %.0f' % (n, v / 1e3) for n, v in inw.most_common(6))))
