"""(test infrastructure: uses the oracle through tests/test_full_width_gpu.py - hence under tests/, not tools/)
Which pass of F(4x4,3x3) costs gradient accuracy?  The full-width parity run of tests/test_full_width_gpu.py (one
512x512 image, float64 + float32 oracles) under per-pass tile choices (mrcnn_conv2d_set_winograd_pass_tiles)."""
import os, sys
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R_, 'chainer-maskrcnn_amd')); sys.path.insert(0, R_)
import numpy as np
from chainer_maskrcnn import _hip
import tests.test_full_width_gpu as T
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
KEYS = ('extractor/conv_p2/W', 'extractor/conv_p3/W', 'extractor/conv_p4/W', 'extractor/lat_p2/W', 'extractor/lat_p4/W', 'extractor/toplayer/W',
        'rpn/conv/W', 'rpn/loc_score/W', 'head/conv1/W', 'head/mask1/W', 'head/mask4/W', 'extractor/resnet/res2/a/conv2/W', 'extractor/resnet/res4/b1/conv2/W')
print('%-18s' % 'fwd,bwdD,bwdF', ' '.join('%-10s' % k.split('/')[-2][:10] for k in KEYS), ' act_max  med(dev/floor)  #>1e-3&>3floor')
# optional per-layer rules: MRCNN_PROBE_RULES="resnet;conv_p;rpn/,head/" -> one run per rule with the forward tile 0 (F(4x4)
# where the layer is large enough) for layers whose name contains one of the rule's comma-separated substrings, F(2x2)
# for every other layer; backward passes as shipped
from chainer_maskrcnn.nn import core as _core
RULES = [r for r in os.environ.get('MRCNN_PROBE_RULES', '').split(';') if r]
COMBOS = [tuple(int(v) for v in c.split(',')) for c in sys.argv[2:]] or [(2, 2, 2), (4, 2, 2), (0, 2, 2), (2, 4, 2), (2, 2, 4), (4, 4, 2), (4, 4, 4), (0, 0, 0)]
RUNS = [(None, c) for c in COMBOS] if not RULES else [(r, (2, 0, 0)) for r in RULES]
for rule, combo in RUNS:
    if rule:
        _core.FWD_TILE_RULE = (lambda subs: (lambda name: 0 if any(s_ in name for s_ in subs) else 2))(rule.split(','))
    T.MODES['probe'] = ((256, 2048, 0), combo)
    acts, losses, rows, iso = T._run(S, 'probe')
    d = {n: (e, f) for n, e, f in rows}
    bad = sum(1 for n, e, f in rows if not e < max(1e-3, 3 * f))
    bad5 = [(n, round(e / f, 1)) for n, e, f in rows if not e < max(1e-3, 5 * f)]
    med = np.median([e / max(f, 1e-12) for n, e, f in rows])
    print('%-18s' % (rule or str(combo)), ' '.join('%-10.2e' % d[k][0] for k in KEYS), ' %.1e  %.2f  %d' % (max(acts.values()), med, bad), ' >5x floor:', bad5)
print('%-18s' % 'fp32 floor', ' '.join('%-10.2e' % d[k][1] for k in KEYS))
