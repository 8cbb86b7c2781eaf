"""Per-queue (HIP stream) busy time of one step from a rocprofv3 kernel trace of bench.py (steps delimited by k_sgd):
which stream is the critical path, and how much of the other streams' work hides under it."""
import csv, glob, collections, re, sys
f = sorted(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv'))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Queue_Id']) for r in rows)
sg = [e for e in ev if 'k_sgd' in e[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t0, t1 = sg[k][1], sg[k + 1][1]
ks = [e for e in ev if e[0] >= t0 and e[1] <= t1]
print('step wall %.2f ms, %d kernels' % ((t1 - t0) / 1e6, len(ks)))
byq = collections.defaultdict(list)
for e in ks: byq[e[3]].append(e)
for q, es in sorted(byq.items(), key=lambda kv: -sum(e[1] - e[0] for e in kv[1])):
    busy = sum(e[1] - e[0] for e in es)
    print('queue %s: %4d kernels, busy %.2f ms, first start +%.2f ms, last end +%.2f ms' % (q, len(es), busy / 1e6, (es[0][0] - t0) / 1e6, (max(e[1] for e in es) - t0) / 1e6))
# timeline in 1-ms bins: busy fraction per queue
nb = int((t1 - t0) / 1e6) + 1
qs = sorted(byq)
print('ms   ' + ' '.join('%8s' % ('q' + q) for q in qs) + '   any')
for b in range(nb):
    a, z = t0 + b * 1e6, t0 + (b + 1) * 1e6
    row = []
    for q in qs:
        row.append(sum(max(0, min(e[1], z) - max(e[0], a)) for e in byq[q]) / 1e6)
    # union
    segs = sorted((max(e[0], a), min(e[1], z)) for e in ks if e[1] > a and e[0] < z)
    u = 0; cs = ce = None
    for s, e in segs:
        if ce is None: cs, ce = s, e
        elif s > ce: u += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    if ce is not None: u += ce - cs
    print('%3d  ' % b + ' '.join('%8.2f' % v for v in row) + '  %5.2f' % (u / 1e6))


# forward / backward phases of the step: the backward pass begins with the first loss-gradient or *bwd* kernel
def cat(n):
    if 'k_conv_igemm' in n or 'k_pgemm' in n: return 'gemm'
    if 'k_wino' in n: return 'winograd transforms'
    if 'k_bn_' in n: return 'batchnorm'
    if 'slab' in n or 'colsum' in n or 'k_tail_sum' in n: return 'slab / column sums'
    return 'other'
tb = min((e[0] for e in ks if 'bwd' in e[2] or 'k_sgd' in e[2]), default=t1)
for nm, lo, hi in (('forward', t0, tb), ('backward + update', tb, t1)):
    es = [e for e in ks if lo <= e[0] < hi]
    segs = sorted((e[0], e[1]) for e in es)
    u = 0; cs = ce = None
    for s, e in segs:
        if ce is None: cs, ce = s, e
        elif s > ce: u += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    if ce is not None: u += ce - cs
    print('%s: wall %.2f ms, chip busy %.2f ms, %d kernels, sum of durations %.2f ms' % (nm, (hi - lo) / 1e6, u / 1e6, len(es), sum(e[1] - e[0] for e in es) / 1e6))
    agg = collections.defaultdict(lambda: [0, 0])
    for e in es:
        a = agg[cat(e[2])]; a[0] += 1; a[1] += e[1] - e[0]
    for c, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print('    %-22s n=%4d  %7.2f ms' % (c, n, d / 1e6))
    oth = collections.defaultdict(lambda: [0, 0])
    for e in es:
        if cat(e[2]) == 'other':
            mm = re.search(r'(k_[a-z0-9_]+|[A-Za-z_]+Buffer[A-Za-z]*|elementwise_kernel)', e[2]); a = oth[mm.group(1) if mm else e[2][:50]]; a[0] += 1; a[1] += e[1] - e[0]
    for c, (n, d) in sorted(oth.items(), key=lambda kv: -kv[1][1])[:12]:
        print('        %-50s n=%3d %7.1f us' % (c, n, d / 1e3))
