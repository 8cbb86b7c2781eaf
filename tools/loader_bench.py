"""Host data pipeline throughput: COCO-sized JPEGs (640x480, 8 polygon instances of ~40 vertices) through
COCOMaskLoader + Transform (resize to 800x1066, nearest-resized masks) + BatchLoader, examples/s by worker count."""
import json, os, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'chainer-maskrcnn_amd'))
import numpy as np
from PIL import Image
from chainer_maskrcnn.dataset.coco_dataset import COCOMaskLoader
from chainer_maskrcnn.dataset.loader import BatchLoader
from chainer_maskrcnn.dataset.transforms import Transform, RawTransform


class Sizes(object):
    min_size, max_size = 800, 1333


root = tempfile.mkdtemp()
os.makedirs(root + '/annotations'); os.makedirs(root + '/train2017')
rs = np.random.RandomState(0)
images, anns, aid = [], [], 1
for i in range(32):
    h, w = 480, 640
    base = rs.randint(0, 256, (h // 8, w // 8, 3)).astype(np.uint8)
    Image.fromarray(base).resize((w, h), Image.BILINEAR).save(root + '/train2017/%d.jpg' % i, quality=90)
    images.append({'id': i, 'file_name': '%d.jpg' % i, 'height': h, 'width': w})
    for _ in range(8):
        cx, cy, r = rs.uniform(100, 540), rs.uniform(100, 380), rs.uniform(20, 90)
        ang = np.sort(rs.uniform(0, 2 * np.pi, 40))
        rad = r * rs.uniform(0.7, 1.0, 40)
        poly = np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1).reshape(-1)
        x0, y0, x1, y1 = poly[0::2].min(), poly[1::2].min(), poly[0::2].max(), poly[1::2].max()
        anns.append({'id': aid, 'image_id': i, 'category_id': 1, 'bbox': [float(x0), float(y0), float(x1 - x0), float(y1 - y0)],
                     'iscrowd': 0, 'segmentation': [poly.round(2).tolist()]}); aid += 1
json.dump({'images': images, 'annotations': anns, 'categories': [{'id': 1, 'name': 'person'}]},
          open(root + '/annotations/instances_train2017.json', 'w'))
ds = COCOMaskLoader(anno_dir=root + '/annotations', img_dir=root, split='train', data_type='2017')
tf = Transform(Sizes())
t0 = time.perf_counter(); ex = ds[0]; t1 = time.perf_counter(); out = tf(ex); t2 = time.perf_counter()
print('one example: decode + annToMask %.1f ms, transform %.1f ms, output image %s masks %s' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, out[0].shape, out[3].shape))
for workers in (1, 2, 4, 8, 16):
    ld = BatchLoader(ds, tf, batch_size=2, num_workers=workers, max_gt=8)
    next(ld)
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 4.0:
        next(ld); n += 2
    dt = time.perf_counter() - t0
    ld.close()
    print('workers %2d: %.1f images/s (%d CPUs visible)' % (workers, n / dt, os.cpu_count()))

import torch
if torch.cuda.is_available():
    dev = torch.device('cuda:0')
    for name, t in (('host Transform + H2D of the prepared batch', tf), ('RawTransform + device resize', RawTransform(Sizes()))):
        for workers in (4, 8, 16):
            ld = BatchLoader(ds, t, batch_size=2, num_workers=workers, max_gt=8, device=dev)
            next(ld); torch.cuda.synchronize()
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < 4.0:
                b = next(ld); n += 2
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ld.close()
            print('%-46s workers %2d: %.1f images/s' % (name, workers, n / dt))
