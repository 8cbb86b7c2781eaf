"""the full-width parity statistics of the benchmarked configuration (two 1024x1024 images, full width, all 256
mask rows) over FIVE batches (seeds 100..104), for the float32-MFMA path and the float32-accurate bf16x6 emulation - is the
"at most 3 % of the tensors above 3 x floor" bar a property of the arithmetic or a noise statistic?  One float64 + two float32
oracle evaluations per seed (~90 s on the GPU box), every mode against the same oracle.
usage: python tests/tools/seed_table.py [modes ...]  (default: shipped bf16x6 bf16x6_bwd_only)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
from tests import test_full_width_gpu as t
modes = sys.argv[1:] or ['shipped', 'bf16x6', 'bf16x6_bwd_only']
seeds = [int(s) for s in os.environ.get('SEEDS', '100,101,102,103,104').split(',')]
print('%-18s %5s | %9s %9s | %8s %8s %8s | %6s %6s | %9s | %s' % ('mode', 'seed', 'act max', 'loss max', 'med r', 'max r', '>3x (n)', '>3x %', '>6x', 'iso', 'verdict (bars: act 1e-3, loss 1e-4, every tensor < 6x, <= 3 % above 3x, median <= 1.3)'), flush=True)
for seed in seeds:
    for mode in modes:
        acts, losses, rows, iso = t._run(1024, mode, N=2, seed=seed, G=8)
        ratios = sorted(e / max(fl, 1e-12) for n, e, fl in rows if e >= 1e-3)
        above3 = [r for r in rows if not r[1] < max(1e-3, 3 * r[2])]
        above6 = [r for r in rows if not r[1] < max(1e-3, 6 * r[2])]
        ok = max(acts.values()) <= 1e-3 and max(losses.values()) <= 1e-4 and not above6 and len(above3) <= 0.03 * len(rows) and ratios[len(ratios) // 2] <= 1.3
        print('%-18s %5d | %9.2e %9.2e | %8.2f %8.2f %8d | %6.1f %6d | %9.2e | %s  worst: %s' % (
            mode, seed, max(acts.values()), max(losses.values()), ratios[len(ratios) // 2], max(r[1] / max(r[2], 1e-12) for r in rows if r[1] >= 1e-3),
            len(above3), 100.0 * len(above3) / len(rows), len(above6), iso, 'PASS' if ok else 'FAIL',
            ', '.join('%s %.1fx' % (r[0], r[1] / max(r[2], 1e-12)) for r in sorted(above3, key=lambda r: -r[1] / max(r[2], 1e-12))[:4])), flush=True)
    t._cache.pop(('oracle', 1024, False, 2, seed, 8), None)
