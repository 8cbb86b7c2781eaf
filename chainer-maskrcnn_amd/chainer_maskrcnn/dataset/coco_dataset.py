"""COCO instance-mask and person-keypoint datasets for the host side of the training pipeline.

Role of chainer_maskrcnn/dataset/coco_dataset.py:11-161 in the reference (same class names, constructor arguments and
example tuples, so train.py's wiring is unchanged):

  COCOMaskLoader[i]      -> img (3,H,W) float32 RGB 0..255, bbox (G,4) float32 (y1,x1,y2,x2) built from the
                            integer-truncated COCO (x,y,w,h), label (G,) int32 = rank of the category among the selected
                            categories (COCO ids are not contiguous), masks = list of G (H,W) uint8 arrays
  COCOKeypointsLoader[i] -> img, bbox (G,4) with w,h >= 1, keypoints (G,17,3) (x,y,v)

Built differently from the reference: the annotation file is indexed ONCE at construction into a per-image table
(boxes, labels and references to the segmentations), straight from ``coco_api.COCO``'s image->annotation index, so an
example costs a JPEG decode plus the mask rasterisation and no per-call annotation queries - the loader threads
(dataset/loader.py) call ``get_example`` concurrently.  No pycocotools, no chainercv: PIL decodes the image.
"""
import os

import numpy as np

from chainer_maskrcnn.dataset.coco_api import COCO

_SPLITS = {'train': 'train', 'val': 'val', 'validation': 'val'}


def read_image(path, color=True):
    """(C,H,W) float32, RGB order, values 0..255 (what chainercv.utils.read_image returns)."""
    from PIL import Image
    with Image.open(path) as f:
        img = np.asarray(f.convert('RGB' if color else 'P'), dtype=np.float32)
    return img[None] if img.ndim == 2 else np.ascontiguousarray(img.transpose(2, 0, 1))


def _yxyx(ann, min_side=0):
    x, y, w, h = (int(v) for v in ann['bbox'])
    if min_side:
        w, h = max(min_side, w), max(min_side, h)
    return (y, x, y + h, x + w)


class _CocoSplit(object):
    """One annotation file + image directory; ``records`` = [(file name, image id, annotations kept)]."""

    def __init__(self, kind, anno_dir, img_dir, split, data_type):
        if split not in _SPLITS:
            raise ValueError('please pick split from \'train\', \'val\',\'validation\'')
        tag = _SPLITS[split] + data_type
        self.coco = COCO(os.path.join(anno_dir, '%s_%s.json' % (kind, tag)))
        self.img_dir = os.path.join(img_dir, tag)
        self.records = []

    def _index(self, keep):
        """keep(ann) -> bool.  Images (ascending id) with at least one kept annotation; annotation order = file order."""
        for img_id in sorted(self.coco.imgs):
            anns = [a for a in self.coco.imgToAnns.get(img_id, ()) if keep(a)]
            if anns:
                self.records.append((self.coco.imgs[img_id]['file_name'], img_id, anns))

    @property
    def img_infos(self):
        return [(name, img_id) for name, img_id, _ in self.records]

    @property
    def length(self):
        return len(self.records)

    def __len__(self):
        return len(self.records)

    def __getitem__(self, i):
        return self.get_example(i)

    def _record(self, i):
        if not 0 <= i < len(self.records):
            raise IndexError('index is out of bounds.')
        name, _, anns = self.records[i]
        return read_image(os.path.join(self.img_dir, name), color=True), anns


class COCOMaskLoader(_CocoSplit):
    def __init__(self, anno_dir='data/annotations', img_dir='data', split='train', data_type='2014', category_filter=None):
        super().__init__('instances', anno_dir, img_dir, split, data_type)
        self.cat_ids = self.coco.getCatIds(catNms=list(category_filter or []))
        self._label_of = {c: k for k, c in enumerate(self.cat_ids)}
        self._index(lambda a: a['category_id'] in self._label_of)
        # box / label arrays are fixed per image: computed here, not per example
        self._boxes = [np.array([_yxyx(a) for a in anns], np.float32).reshape(-1, 4) for _, _, anns in self.records]
        self._labels = [np.array([self._label_of[a['category_id']] for a in anns], np.int32) for _, _, anns in self.records]

    def get_example(self, i):
        img, anns = self._record(i)
        return img, self._boxes[i].copy(), self._labels[i].copy(), [self.coco.annToMask(a) for a in anns]


class COCOKeypointsLoader(_CocoSplit):
    n_keypoints = 17

    def __init__(self, anno_dir='data/annotations', img_dir='data', split='train', data_type='2014'):
        super().__init__('person_keypoints', anno_dir, img_dir, split, data_type)
        person = set(self.coco.getImgIds(catIds=[1]))
        self._index(lambda a: a['image_id'] in person)       # every annotation of an image that shows a person
        self._boxes = [np.array([_yxyx(a, min_side=1) for a in anns], np.float32).reshape(-1, 4) for _, _, anns in self.records]
        self._kps = [np.array([a['keypoints'] for a in anns]).reshape(-1, self.n_keypoints, 3) for _, _, anns in self.records]

    def get_example(self, i):
        img, _ = self._record(i)
        return img, self._boxes[i].copy(), self._kps[i].copy()
