"""GPU test of the host data pipeline end to end: COCO files -> loaders -> Transform -> BatchLoader (pinned memory,
copy stream) -> one training step on the device.  The dataset is written on the fly (PNG + COCO JSON)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from chainer_maskrcnn.dataset.coco_dataset import COCOMaskLoader  # noqa: E402
from chainer_maskrcnn.dataset.loader import BatchLoader  # noqa: E402
from chainer_maskrcnn.dataset.transforms import Transform  # noqa: E402
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss  # noqa: E402
from chainer_maskrcnn.model.maskrcnn import MaskRCNN  # noqa: E402
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay  # noqa: E402

DEV = 'cuda:0'


def _write_dataset(root, n_img=4):
    from PIL import Image
    os.makedirs(os.path.join(root, 'annotations'))
    os.makedirs(os.path.join(root, 'train2017'))
    rs = np.random.RandomState(0)
    images, anns = [], []
    aid = 1
    for i in range(n_img):
        h, w = int(rs.randint(90, 130)), int(rs.randint(100, 160))
        Image.fromarray(rs.randint(0, 256, (h, w, 3)).astype(np.uint8)).save(os.path.join(root, 'train2017', '%d.png' % i))
        images.append({'id': i, 'file_name': '%d.png' % i, 'height': h, 'width': w})
        for _ in range(3):
            bw, bh = int(rs.randint(20, 60)), int(rs.randint(20, 60))
            x, y = int(rs.randint(0, w - bw)), int(rs.randint(0, h - bh))
            poly = [x, y, x + bw, y + bh // 3, x + bw, y + bh, x + bw // 2, y + bh, x, y + bh // 2]
            anns.append({'id': aid, 'image_id': i, 'category_id': int(rs.choice([1, 3])), 'bbox': [x, y, bw, bh], 'iscrowd': 0,
                         'segmentation': [poly]})
            aid += 1
    cats = [{'id': 1, 'name': 'person'}, {'id': 3, 'name': 'car'}]
    json.dump({'images': images, 'annotations': anns, 'categories': cats},
              open(os.path.join(root, 'annotations', 'instances_train2017.json'), 'w'))


def test_coco_files_to_training_step(tmp_path):
    root = str(tmp_path)
    _write_dataset(root)
    m = MaskRCNN(n_fg_class=2, device=DEV, seed=3, min_size=128, max_size=192, _test_shrink=dict(stages=(1, 1, 1, 1), width_div=4))
    chain = FPNMaskRCNNTrainChain(m, mask_loss_fun=calc_mask_loss)
    opt = MomentumSGD(lr=1e-3).setup(chain)
    opt.add_hook(WeightDecay(5e-4))
    ds = COCOMaskLoader(anno_dir=root + '/annotations', img_dir=root, split='train', data_type='2017', category_filter=['person', 'car'])
    ld = BatchLoader(ds, Transform(m), batch_size=2, shuffle=True, seed=1, num_workers=2, max_gt=3, device=DEV)
    try:
        for _ in range(3):
            b = next(ld)
            assert b['imgs'].is_cuda and b['imgs'].shape[2] % 64 == 0 and b['masks'].dtype == torch.uint8
            assert isinstance(b['scales'], np.ndarray)
            opt.update(chain, b['imgs'], b['bboxes'], b['labels'], b['masks'], float(b['scales'][0]))
            obs = {k: float(v) for k, v in chain.observation.items()}
            assert all(np.isfinite(v) for v in obs.values()), obs
            assert obs['mask_loss'] > 0 and obs['rpn_cls_loss'] > 0
    finally:
        ld.close()


def test_device_transform_equals_host_transform(tmp_path):
    """RawTransform + the two resize kernels produce bit-identical batches to the host Transform + collate."""
    from chainer_maskrcnn.dataset.transforms import RawTransform

    class Sizes(object):
        min_size, max_size = 200, 256
    root = str(tmp_path)
    _write_dataset(root, n_img=5)
    ds = COCOMaskLoader(anno_dir=root + '/annotations', img_dir=root, split='train', data_type='2017')
    host = BatchLoader(ds, Transform(Sizes()), batch_size=2, shuffle=True, seed=2, num_workers=2, max_gt=4, device=DEV)
    devl = BatchLoader(ds, RawTransform(Sizes()), batch_size=2, shuffle=True, seed=2, num_workers=2, max_gt=4, device=DEV)
    try:
        for _ in range(4):
            a, b = next(host), next(devl)
            assert torch.equal(a['imgs'], b['imgs'])
            assert torch.equal(a['masks'], b['masks'])
            assert torch.equal(a['bboxes'], b['bboxes']) and torch.equal(a['labels'], b['labels'])
            np.testing.assert_array_equal(a['scales'], b['scales'])
    finally:
        host.close()
        devl.close()
