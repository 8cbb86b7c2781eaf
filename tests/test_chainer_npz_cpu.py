"""CPU tests of the Chainer-NPZ weight mapping (train.py:99-101,134-137): key names and array
layouts of a Chainer snapshot, exact round trip, and the two non-trivial layout changes (fc1 flattening order, the
deconvolution as 1x1 convolution + pixel shuffle) checked functionally with plain torch ops."""
import numpy as np
import torch
import torch.nn.functional as F

from chainer_maskrcnn.model.maskrcnn import MaskRCNN
from chainer_maskrcnn.utils.chainer_npz import ChainerNpzMap, save_npz, load_npz, load_resnet50_npz


def _model(seed):
    return MaskRCNN(n_fg_class=80, device='cpu', seed=seed, _test_shrink=dict(stages=(2, 1, 1, 1), width_div=2))


def test_chainer_key_names_and_shapes():
    m = _model(1)
    d = ChainerNpzMap(m).to_chainer()
    assert d['extractor/resnet/conv1/W'].shape == (32, 3, 7, 7)            # (Cout, Cin, KH, KW), un-padded
    assert d['extractor/resnet/res2/a/conv2/W'].shape == (32, 32, 3, 3)
    assert d['extractor/resnet/res2/b1/bn3/gamma'].shape == (128,)
    assert 'extractor/resnet/res2/a/bn4/avg_var' in d and 'extractor/resnet/bn1/N' in d
    assert d['extractor/lat_p3/W'].shape == (128, 256, 1, 1)
    assert d['rpn/loc/W'].shape == (12, 128, 1, 1) and d['rpn/score/W'].shape == (6, 128, 1, 1)
    assert d['head/cls_loc/W'].shape == (4, 512) and d['head/score/W'].shape == (81, 512)
    assert d['head/fc1/W'].shape == (512, 128 * 7 * 7) and d['head/fc2/W'].shape == (512, 512)
    assert d['head/deconv1/W'].shape == (128, 128, 2, 2) and d['head/conv2/W'].shape == (80, 128, 1, 1)
    assert d['head/mask4/b'].shape == (128,)


def test_round_trip_is_exact(tmp_path):
    a, b = _model(1), _model(2)
    assert not torch.equal(a.ps.params, b.ps.params)
    path = str(tmp_path / 'model_5000.npz')
    save_npz(path, a)
    loaded = load_npz(path, b, strict=True)
    assert len(loaded) > 100
    assert torch.equal(a.ps.params, b.ps.params)          # incl. the zero padding
    for k in a.ps.buffers:
        assert torch.equal(a.ps.buffers[k], b.ps.buffers[k])


def test_fc1_and_deconv_layout_equivalence():
    m = _model(3)
    d = ChainerNpzMap(m).to_chainer()
    h = m.head
    g = torch.Generator().manual_seed(0)
    c, s = h.channels, h.roi_size_box
    x = torch.randn((4, c, s, s), generator=g)                                            # NCHW, Chainer side
    ref = F.linear(x.reshape(4, -1), torch.from_numpy(d['head/fc1/W']), torch.from_numpy(d['head/fc1/b']))
    ours = F.linear(x.permute(0, 2, 3, 1).reshape(4, -1), m.ps.p(h.fc1.name + '/W')[:, 0, 0, :], m.ps.p(h.fc1.name + '/b'))
    assert torch.allclose(ref, ours[:, :h.fc1.cout], atol=1e-5)
    y = torch.randn((2, c, 5, 5), generator=g)
    ref = F.conv_transpose2d(y, torch.from_numpy(d['head/deconv1/W']), torch.from_numpy(d['head/deconv1/b']), stride=2)
    t = F.conv2d(y, m.ps.p(h.deconv1.name + '/W')[:, 0, 0, :c][:, :, None, None])         # (2, 4C, 5, 5), channel = (a*2+b)*C+o
    ours = t.reshape(2, 2, 2, c, 5, 5).permute(0, 3, 4, 1, 5, 2).reshape(2, c, 10, 10) + m.ps.p(h.deconv_b)[None, :, None, None]
    assert torch.allclose(ref, ours, atol=1e-5)


def test_resnet50_snapshot_initialises_the_bottom_up_pathway(tmp_path):
    """A ``ResNet50Layers`` snapshot (keys without the ``extractor/resnet/`` prefix, with an ``fc6`` the FPN deletes,
    feature_pyramid_network.py:22-23) fills exactly the bottom-up parameters and leaves everything else alone."""
    src, dst = _model(4), _model(5)
    d = ChainerNpzMap(src).to_chainer()
    pre = 'extractor/resnet/'
    snap = {k[len(pre):]: v for k, v in d.items() if k.startswith(pre)}
    snap['fc6/W'], snap['fc6/b'] = np.zeros((1000, 16), np.float32), np.zeros((1000,), np.float32)
    path = str(tmp_path / 'ResNet-50-model.npz')
    np.savez(path, **snap)
    before = ChainerNpzMap(dst).to_chainer()
    loaded = load_resnet50_npz(path, dst)
    assert set(loaded) == {pre + k for k in snap if not k.startswith('fc6/') and not k.endswith('/N')}        # (N: Chainer's step counter)
    after = ChainerNpzMap(dst).to_chainer()
    for k in after:
        want = d[k] if k.startswith(pre) else before[k]
        assert np.array_equal(after[k], want), k
