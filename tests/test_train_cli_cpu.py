"""train.py / train_keypoints.py accept every flag of the reference's scripts, spelled as there, with the reference's defaults
(/root/reference/train.py:62-74, /root/reference/train_keypoints.py:73-89), and `--dataset depth` has its dataset
(chainer_maskrcnn/dataset/depth_dataset.py:7-61, utils/depth_transformer.py:4-10)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (flag, short, a value to pass, the reference's default)
REF_TRAIN = [('--gpu', '-g', '1', 0), ('--lr', '-l', '0.01', 1e-3), ('--out', '-o', 'x', 'result'), ('--iteration', '-i', '7', 200000),
             ('--weight', '-w', 'w.npz', ''), ('--label_file', '-f', 'l.txt', 'data/label_coco.txt'), ('--backbone', None, 'fpn', 'fpn'),
             ('--head-arch', '-a', 'fpn', 'fpn'), ('--multi-gpu', '-m', '1', 0), ('--batch-size', '-b', '2', 1)]
REF_KEYPOINTS = [('--gpu', '-g', '1', 0), ('--lr', '-l', '0.01', 1e-3), ('--out', '-o', 'x', 'result'), ('--iteration', '-i', '7', 200000),
                 ('--weight', '-w', 'w.npz', ''), ('--label_file', '-f', 'l.txt', 'data/label_coco.txt'), ('--backbone', None, 'fpn', 'fpn'),
                 ('--head_arch', '-a', 'fpn_keypoint', 'fpn_keypoint'), ('--multi_gpu', '-m', '1', 0), ('--batch_size', '-b', '2', 1),
                 ('--dataset', None, 'depth', 'coco'), ('--n_mask_convs', None, '4', None), ('--min_size', None, '512', 600),
                 ('--max_size', None, '800', 1000)]


@pytest.mark.parametrize('keypoints,flags', [(False, REF_TRAIN), (True, REF_KEYPOINTS)], ids=['train.py', 'train_keypoints.py'])
def test_parser_accepts_the_reference_flag_set(keypoints, flags):
    import train
    p = train.build_parser(keypoints=keypoints)
    d = vars(p.parse_args([]))
    for flag, short, value, default in flags:
        dest = flag.lstrip('-').replace('-', '_')
        assert d[dest] == default, (flag, d[dest], default)
    argv = []
    for flag, short, value, default in flags:
        argv += [flag, value]
    a = vars(p.parse_args(argv))
    for flag, short, value, default in flags:
        dest = flag.lstrip('-').replace('-', '_')
        want = type(default)(value) if default is not None else int(value)
        assert a[dest] == want, (flag, a[dest], want)
    argv = []
    for flag, short, value, default in flags:       # the short spellings
        if short:
            argv += [short, value]
    a = vars(p.parse_args(argv))
    for flag, short, value, default in flags:
        if short:
            assert a[flag.lstrip('-').replace('-', '_')] == type(default)(value)
    with pytest.raises(SystemExit):
        p.parse_args(['--dataset', 'voc'] if keypoints else ['--head_arch', 'fpn'])     # (train.py spells it --head-arch)


def test_depth_dataset_example(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'chainer-maskrcnn_amd'))
    from chainer_maskrcnn.dataset.depth_dataset import DepthDataset, DepthTransformer
    rs = np.random.RandomState(3)
    depth = (rs.rand(48, 64) * 3000 + 1000).astype(np.uint16)
    kp = np.concatenate([rs.rand(20, 2) * [70, 70] - 3, rs.rand(20, 1)], axis=1)       # some outside the frame, confidences around 0.2
    np.savez(tmp_path / 'a.npz', depth=depth, keypoints=kp)
    np.savez(tmp_path / 'b.npz', depth=depth, keypoints=kp[:, :2])
    (tmp_path / 'list.txt').write_text('a.npz\nb.npz\n')
    ds = DepthDataset(str(tmp_path / 'list.txt'), root=str(tmp_path))
    assert len(ds) == 2 and ds.n_keypoints == 20
    img, bbox, k = ds[0]
    assert img.shape == (3, 48, 64) and img.dtype == np.float32 and bbox.shape == (1, 4) and k.shape == (1, 20, 3)
    np.testing.assert_array_equal(img[0], (depth.astype(np.float32) - 1000) / 3000 * 255)
    c = np.clip(kp[:, :2], 0, [47, 63])
    np.testing.assert_array_equal(k[0, :, :2], c[:, [1, 0]])
    np.testing.assert_array_equal(k[0, :, 2], (kp[:, 2] > 0.2) * 2)
    np.testing.assert_array_equal(bbox[0], np.concatenate([np.clip(c.min(0) - [10, 10], 0, [47, 63]), np.clip(c.max(0) + [0, 10], 0, [47, 63])]))
    assert (ds[1][2][0, :, 2] == 2).all()
    with pytest.raises(IndexError):
        ds[2]
    x, b2, k2 = DepthTransformer(np.random.RandomState(0))((img, bbox, k))
    off = (np.random.RandomState(0).rand(1).astype(np.float32) - 0.5) * 30
    np.testing.assert_array_equal(x, img + off)
    assert b2 is bbox and k2 is k
