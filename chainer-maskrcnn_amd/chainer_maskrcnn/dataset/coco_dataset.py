"""COCO instance-mask and person-keypoint loaders.

Mirror of chainer_maskrcnn/dataset/coco_dataset.py:11-161: same constructor arguments, same example tuples
  COCOMaskLoader.get_example(i)      -> img (3,H,W) float32 RGB 0..255, bbox (G,4) float32 (y1,x1,y2,x2) from the
                                        integer-truncated COCO (x,y,w,h), label (G,) int32 = position of the category
                                        in the filtered category list (COCO ids are not contiguous, :83-85),
                                        masks = list of G (H,W) uint8 arrays (left as a list for Transform, :99)
  COCOKeypointsLoader.get_example(i) -> img, bbox (G,4) with w,h clamped to >= 1, keypoints (G,17,3) (x,y,v)
on ``coco_api.COCO`` (no pycocotools) and PIL (the reference reads through chainercv.utils.read_image = PIL).
"""
from os.path import join

import numpy as np

from chainer_maskrcnn.dataset.coco_api import COCO


def read_image(path, color=True):
    """chainercv.utils.read_image: (C,H,W) float32, RGB order, values 0..255."""
    from PIL import Image
    with Image.open(path) as f:
        img = np.asarray(f.convert('RGB' if color else 'P'), dtype=np.float32)
    return img[None] if img.ndim == 2 else np.ascontiguousarray(img.transpose(2, 0, 1))


class _Dataset(object):
    def __len__(self):
        return self.length

    def __getitem__(self, i):
        return self.get_example(i)


class COCOMaskLoader(_Dataset):
    def __init__(self, anno_dir='data/annotations', img_dir='data', split='train', data_type='2014', category_filter=None):
        if split not in ['train', 'val', 'validation']:
            raise ValueError('please pick split from \'train\', \'val\',\'validation\'')
        if split == 'validation':
            split = 'val'
        ann_file = '{}/instances_{}{}.json'.format(anno_dir, split, data_type)
        self.coco = COCO(ann_file)
        self.img_dir = '{}/{}{}'.format(img_dir, split, data_type)
        target_cats = [] if category_filter is None else category_filter
        self.cat_ids = self.coco.getCatIds(catNms=target_cats)
        img_ids = set()                       # images that contain ANY of the categories (:32-35)
        for cat_id in self.cat_ids:
            img_ids |= set(self.coco.getImgIds(catIds=[cat_id]))
        self.img_infos = [(i['file_name'], i['id']) for i in self.coco.loadImgs(sorted(img_ids))]
        self.length = len(self.img_infos)

    def _contain_large_enough_annotation(self, img_id, min_w=10, min_h=10):
        for ann in self.coco.loadAnns(self.coco.getAnnIds(imgIds=img_id)):
            x, y, w, h = [int(j) for j in ann['bbox']]
            if w <= min_w or h <= min_h:
                continue
            if ann['category_id'] in self.cat_ids:
                return True
        return False

    def _contain_large_annotation_only(self, img_id, min_w=10, min_h=10):
        for ann in self.coco.loadAnns(self.coco.getAnnIds(imgIds=img_id)):
            x, y, w, h = [int(j) for j in ann['bbox']]
            if ann['category_id'] in self.cat_ids and (w <= min_w or h <= min_h):
                return False
        return True

    def get_example(self, i):
        if i >= self.length:
            raise IndexError('index is out of bounds.')
        file_name, img_id = self.img_infos[i]
        img = read_image(join(self.img_dir, file_name), color=True)
        assert img.shape[0] == 3
        gt_boxes, gt_masks, gt_labels = [], [], []
        for ann in self.coco.loadAnns(self.coco.getAnnIds(imgIds=img_id)):
            x, y, w, h = [int(j) for j in ann['bbox']]
            if ann['category_id'] in self.cat_ids:
                gt_boxes.append(np.array([y, x, y + h, x + w], dtype=np.float32))
                gt_masks.append(self.coco.annToMask(ann))
                gt_labels.append(self.cat_ids.index(ann['category_id']))
        return img, np.array(gt_boxes), np.array(gt_labels, dtype=np.int32), gt_masks


class COCOKeypointsLoader(_Dataset):
    n_keypoints = 17

    def __init__(self, anno_dir='data/annotations', img_dir='data', split='train', data_type='2014'):
        if split not in ['train', 'val', 'validation']:
            raise ValueError('please pick split from \'train\', \'val\',\'validation\'')
        if split == 'validation':
            split = 'val'
        ann_file = '{}/person_keypoints_{}{}.json'.format(anno_dir, split, data_type)
        self.coco = COCO(ann_file)
        self.img_dir = '{}/{}{}'.format(img_dir, split, data_type)
        img_ids = self.coco.getImgIds(catIds=[1])        # person only (:118)
        all_img_infos = [(i['file_name'], i['id']) for i in self.coco.loadImgs(sorted(img_ids))]
        self.img_infos = [info for info in all_img_infos                  # images without annotations are dropped (:121-127)
                          if len(self.coco.loadAnns(self.coco.getAnnIds(imgIds=info[1]))) > 0]
        self.length = len(self.img_infos)

    def get_example(self, i):
        if i >= self.length:
            raise IndexError()
        file_name, img_id = self.img_infos[i]
        img = read_image(join(self.img_dir, file_name), color=True)
        keypoints, gt_boxes = [], []
        for ann in self.coco.loadAnns(self.coco.getAnnIds(imgIds=img_id)):
            keypoints.append(np.array(ann['keypoints']).reshape((-1, 3)))
            x, y, w, h = [int(j) for j in ann['bbox']]
            h = max(1.0, h)
            w = max(1.0, w)
            gt_boxes.append(np.array([y, x, y + h, x + w], dtype=np.float32))
        keypoints = np.array(keypoints).reshape((-1, 17, 3))
        gt_boxes = np.array(gt_boxes).reshape((-1, 4))
        return img, gt_boxes, keypoints
