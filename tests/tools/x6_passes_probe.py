"""Which pass of the bf16x6 emulation moves the FPN gradients on the benchmarked batch?  (3,0,0) and (0,3,3) against the float64 oracle."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd')); sys.path.insert(0, R)
from tests import test_full_width_gpu as t
SEED = int(os.environ.get('PROBE_SEED', '100'))
for mode in sys.argv[1:]:
    try:
        t._check(1024, mode, N=2, seed=SEED, G=8)
        print(mode, 'passes every bar')
    except AssertionError as e:
        print(mode, 'FAILS', str(e)[:300])
