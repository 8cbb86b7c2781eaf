"""MI355X-native FPN Mask R-CNN training path.

Keeps the module paths and call signatures of katotetsuro/chainer-maskrcnn
(``chainer_maskrcnn.functions...``, ``chainer_maskrcnn.model...``,
``chainer_maskrcnn.utils...``) with ``torch.Tensor`` on a HIP device in place of
``chainer.Variable``/CuPy arrays.  All arithmetic on the path runs in hand-written
gfx950 HIP kernels from ``csrc/`` behind the C ABI of ``include/mrcnn_hip.h``;
there is no CPU or eager-PyTorch fallback - a missing library raises.
"""


def train_step_available():
    """bench.py switches to the full training-step workload (BASELINE.json configs[2]) when this is True."""
    return True
