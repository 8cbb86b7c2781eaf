"""The subset of ``pycocotools.coco.COCO`` that chainer_maskrcnn/dataset/coco_dataset.py uses (:26-38, :47-91,
:118-131, :145-158), without pycocotools (a C extension that is not installed on the target machine).

Parity unpinned: pycocotools 2.0 is an un-vendored dependency of the reference; the index methods restate its public
Python (``coco.py``: createIndex / getCatIds / getImgIds / getAnnIds / loadImgs / loadAnns / annToRLE / annToMask)
and the mask routines restate its C (``common/maskApi.c``: rleFrPoly, rleFrString, rleMerge-as-union, rleDecode).
Masks are column-major run-length codes; the polygon rasteriser works on a x5 up-sampled integer grid and keeps
the pixels whose centres lie inside the polygon with maskApi's tie rules - restated operation for operation, so
that real COCO annotations decode to the masks the reference trains on.
"""
import json
from collections import defaultdict

import numpy as np


class COCO(object):
    def __init__(self, annotation_file=None):
        self.dataset, self.anns, self.cats, self.imgs = {}, {}, {}, {}
        self.imgToAnns, self.catToImgs = defaultdict(list), defaultdict(list)
        if annotation_file is not None:
            with open(annotation_file, 'r') as f:
                self.dataset = json.load(f)
            self.createIndex()

    def createIndex(self):
        for ann in self.dataset.get('annotations', []):
            self.imgToAnns[ann['image_id']].append(ann)
            self.anns[ann['id']] = ann
        for img in self.dataset.get('images', []):
            self.imgs[img['id']] = img
        for cat in self.dataset.get('categories', []):
            self.cats[cat['id']] = cat
        if 'annotations' in self.dataset and 'categories' in self.dataset:
            for ann in self.dataset['annotations']:
                self.catToImgs[ann['category_id']].append(ann['image_id'])

    # ---- queries (argument subset used by the reference) ----------------------------------------
    def getCatIds(self, catNms=[]):
        cats = self.dataset.get('categories', [])
        if len(catNms) > 0:
            cats = [c for c in cats if c['name'] in catNms]
        return [c['id'] for c in cats]

    def getImgIds(self, imgIds=[], catIds=[]):
        if len(imgIds) == 0 and len(catIds) == 0:
            return list(self.imgs.keys())
        ids = set(imgIds)
        for i, cat_id in enumerate(catIds):
            if i == 0 and len(ids) == 0:
                ids = set(self.catToImgs[cat_id])
            else:
                ids &= set(self.catToImgs[cat_id])
        return list(ids)

    def getAnnIds(self, imgIds=[]):
        imgIds = imgIds if isinstance(imgIds, (list, tuple, set)) else [imgIds]
        if len(imgIds) == 0:
            return [a['id'] for a in self.dataset.get('annotations', [])]
        return [a['id'] for i in imgIds for a in self.imgToAnns.get(i, [])]

    def loadAnns(self, ids=[]):
        return [self.anns[i] for i in ids] if isinstance(ids, (list, tuple)) else [self.anns[ids]]

    def loadImgs(self, ids=[]):
        return [self.imgs[i] for i in ids] if isinstance(ids, (list, tuple, set)) else [self.imgs[ids]]

    # ---- masks ----------------------------------------------------------------------------------
    def annToMask(self, ann):
        """Binary mask (h, w) uint8 of an annotation: polygons (union), uncompressed or compressed RLE."""
        t = self.imgs[ann['image_id']]
        h, w = t['height'], t['width']
        segm = ann['segmentation']
        if isinstance(segm, list):
            m = np.zeros((h, w), np.uint8)
            for poly in segm:
                m |= rle_decode(rle_from_polygon(np.asarray(poly, np.float64), h, w), h, w)
            return m
        counts = segm['counts']
        if isinstance(counts, (str, bytes)):
            counts = rle_from_string(counts)
        return rle_decode(np.asarray(counts, np.int64), h, w)


def rle_decode(counts, h, w):
    """maskApi.c rleDecode: runs alternate 0 / 1 starting with 0, over the column-major flattening of (h, w)."""
    counts = np.asarray(counts, np.int64)
    vals = (np.arange(counts.shape[0]) & 1).astype(np.uint8)
    flat = np.repeat(vals, counts)
    if flat.shape[0] != h * w:
        raise ValueError('RLE covers %d pixels, the mask has %d' % (flat.shape[0], h * w))
    return np.ascontiguousarray(flat.reshape(w, h).T)


def rle_from_string(s):
    """maskApi.c rleFrString: 6 bits per character (offset 48), bit 5 = continuation, sign-extended 5-bit groups,
    every count after the second is a delta to the count two places before."""
    if isinstance(s, str):
        s = s.encode('ascii')
    cnts = []
    p, n = 0, len(s)
    while p < n:
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    return np.asarray(cnts, np.int64)


def rle_from_polygon(xy, h, w):
    """maskApi.c rleFrPoly: xy = (x0, y0, x1, y1, ...) in pixels.  Returns the run-length counts."""
    k = xy.shape[0] // 2
    scale = 5.0
    x = np.empty(k + 1, np.int64)
    y = np.empty(k + 1, np.int64)
    x[:k] = (scale * xy[0:2 * k:2] + 0.5).astype(np.int64)        # (int) truncation; COCO coordinates are >= 0
    y[:k] = (scale * xy[1:2 * k:2] + 0.5).astype(np.int64)
    x[k], y[k] = x[0], y[0]
    us, vs = [], []
    for j in range(k):                                            # all points along every edge, end points included
        xs, xe, ys, ye = int(x[j]), int(x[j + 1]), int(y[j]), int(y[j + 1])
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if dx >= dy:
            s = 0.0 if dx == 0 else float(ye - ys) / dx
            d = np.arange(dx + 1)
            t = dx - d if flip else d
            us.append(t + xs)
            vs.append((ys + s * t + 0.5).astype(np.int64))
        else:
            s = float(xe - xs) / dy
            d = np.arange(dy + 1)
            t = dy - d if flip else d
            vs.append(t + ys)
            us.append((xs + s * t + 0.5).astype(np.int64))
    u = np.concatenate(us) if us else np.zeros(0, np.int64)
    v = np.concatenate(vs) if vs else np.zeros(0, np.int64)
    # points where the boundary crosses a column of the up-sampled grid, mapped back to pixel columns / rows
    if u.shape[0] > 1:
        cross = u[1:] != u[:-1]
        u1, u0, v1, v0 = u[1:][cross], u[:-1][cross], v[1:][cross], v[:-1][cross]
        xd = np.where(u1 < u0, u1, u1 - 1).astype(np.float64)
        xd = (xd + 0.5) / scale - 0.5
        ok = (np.floor(xd) == xd) & (xd >= 0) & (xd <= w - 1)
        yd = np.where(v1 < v0, v1, v0).astype(np.float64)
        yd = (yd + 0.5) / scale - 0.5
        yd = np.ceil(np.clip(yd, 0, h))
        xi, yi = xd[ok].astype(np.int64), yd[ok].astype(np.int64)
    else:
        xi = yi = np.zeros(0, np.int64)
    a = np.sort(np.concatenate([xi * h + yi, [h * w]]))
    a = np.diff(np.concatenate([[0], a]))
    # merge zero-length runs: a zero run glues its neighbours together
    b = []
    j, n = 0, a.shape[0]
    if n:
        b.append(int(a[0]))
        j = 1
    while j < n:
        if a[j] > 0:
            b.append(int(a[j]))
            j += 1
        else:
            j += 1
            if j < n:
                b[-1] += int(a[j])
                j += 1
    return np.asarray(b, np.int64)
