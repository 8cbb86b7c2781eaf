"""Where is the whole chip idle inside a step?  From a rocprofv3 kernel trace of bench.py: intervals of one step (delimited by k_sgd) in which NO
kernel of any queue runs, longest first, with the kernels that end before and start after each.
trace_gaps.py <dir> <step index> [min gap us]"""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
nm = lambda n: re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0].replace('void ', '')[:40]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), nm(r['Kernel_Name']), r.get('Queue_Id', '?')) for r in rows)
sg = [e for e in ev if 'k_sgd' in e[2]]
k = int(sys.argv[2]); mingap = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
t0, t1 = sg[k][1], sg[k + 1][1]
ks = [e for e in ev if e[0] >= t0 and e[1] <= t1]
gaps, cur_end, last = [], ks[0][1], ks[0]
for e in ks[1:]:
    if e[0] > cur_end:
        gaps.append((e[0] - cur_end, cur_end - t0, last, e))
    if e[1] > cur_end:
        cur_end, last = e[1], e
tot = sum(g[0] for g in gaps)
print('step %.2f ms, %d kernels, chip idle %.3f ms in %d gaps (%.3f ms in gaps >= %.0f us)' % ((t1 - t0) / 1e6, len(ks), tot / 1e6, len(gaps),
      sum(g[0] for g in gaps if g[0] >= mingap * 1e3) / 1e6, mingap))
for g in sorted(gaps, key=lambda g: -g[0])[:40]:
    if g[0] < mingap * 1e3: break
    print('%7.1f us idle at t=%8.1f us: after %-40s (q%s) before %-40s (q%s)' % (g[0] / 1e3, g[1] / 1e3, g[2][2], g[2][3], g[3][2], g[3][3]))
