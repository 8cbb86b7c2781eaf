"""Per-step kernel breakdown from a rocprofv3 kernel trace of bench.py (steps are delimited by k_sgd)."""
import csv, glob, collections, re, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
sg = [e for e in ev if 'k_sgd' in e[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t0, t1 = sg[k][1], sg[k + 1][1]
ks = [e for e in ev if e[0] >= t0 and e[1] <= t1]
agg = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in ks:
    n = re.sub(r'\(anonymous namespace\)::', '', n).split('(')[0].replace('void ', '')
    agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
busy = 0; cs = ce = None
for s, e, n in ks:
    if ce is None: cs, ce = s, e
    elif s > ce: busy += ce - cs; cs, ce = s, e
    else: ce = max(ce, e)
busy += ce - cs
tot = sum(v[1] for v in agg.values())
print('wall %.2f ms, busy %.2f ms, sum of durations %.2f ms, kernels %d' % ((t1 - t0) / 1e6, busy / 1e6, tot / 1e3, len(ks)))
print('conv kernels sum %.2f ms' % (sum(v[1] for kk, v in agg.items() if 'k_conv_igemm' in kk or 'k_pgemm' in kk) / 1e3))
for kk, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if 'k_conv_igemm' in kk or 'k_pgemm' in kk or v[1] < 30: continue
    print('%-60s n=%4d %8.1f us' % (kk[:60], v[0], v[1]))
