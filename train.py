#!/usr/bin/env python3
"""Training entry point with the flags, constants and log keys of the reference's train.py (:62-74 flags, :95-98
model, :107-109 optimizer, :117-132 updater wiring, :134-161 snapshot / LR shift / log), on the MI355X-native path.

The reference's train.py imports chainer / chainercv / chainerui / cv2 / pycocotools (train.py:1-8), none of which
exist on the target machine, so this is the repo's own counterpart: same CLI, JSON-lines log in --out.  Data:
synthetic COCO-shaped batches (--synthetic 1, the default - the box has no dataset), or real COCO through
chainer_maskrcnn/dataset (--synthetic 0 --anno-dir data/annotations --img-dir data --data-type 2017: the
reference's COCOMaskLoader / COCOKeypointsLoader + Transform, a prefetching loader, no pycocotools / cv2).
Multi GPU: launch with `python -m torch.distributed.run --nproc-per-node N train.py --multi-gpu 1 ...`
(one process per GPU, RCCL all-reduce of the flat gradient buffer; the reference forks 8 workers itself).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'chainer-maskrcnn_amd'))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def build_parser(keypoints=False):
    parser = argparse.ArgumentParser(description='Mask R-CNN')
    parser.add_argument('--gpu', '-g', type=int, default=0)
    parser.add_argument('--lr', '-l', type=float, default=1e-3)
    parser.add_argument('--out', '-o', default='result', help='Output directory')
    parser.add_argument('--iteration', '-i', type=int, default=200000)
    parser.add_argument('--weight', '-w', type=str, default='')
    parser.add_argument('--resnet50-npz', type=str, default='', help="a chainer.links.ResNet50Layers snapshot for the bottom-up pathway (what ResNet50Layers('auto') loads in the reference, feature_pyramid_network.py:22)")
    if keypoints:   # train_keypoints.py spells its flags with underscores (train_keypoints.py:73-89)
        parser.add_argument('--label_file', '-f', type=str, default='data/label_coco.txt')      # (:79-80; unused there too: n_fg_class is 1)
        parser.add_argument('--backbone', type=str, default='fpn')
        parser.add_argument('--head_arch', '-a', type=str, default='fpn_keypoint')
        parser.add_argument('--multi_gpu', '-m', type=int, default=0)
        parser.add_argument('--batch_size', '-b', type=int, default=1)
        parser.add_argument('--dataset', default='coco', choices=['coco', 'depth'])               # :86
        parser.add_argument('--n_mask_convs', type=int, default=None)                             # :87 (None: the model's 8, maskrcnn.py:110-111)
        parser.add_argument('--min_size', type=int, default=600)                                  # :88
        parser.add_argument('--max_size', type=int, default=1000)                                 # :89
        parser.add_argument('--depth-list', default='data/rgbd/train.txt', help='--dataset depth: the list file (train_keypoints.py:105)')
        parser.add_argument('--depth-root', default='data/rgbd/', help='--dataset depth: directory the listed .npz paths are relative to')
    else:
        parser.add_argument('--label_file', '-f', type=str, default='data/label_coco.txt')
        parser.add_argument('--backbone', type=str, default='fpn')
        parser.add_argument('--head-arch', '-a', type=str, default='fpn')
        parser.add_argument('--multi-gpu', '-m', type=int, default=0)
        parser.add_argument('--batch-size', '-b', type=int, default=1)
    parser.add_argument('--synthetic', type=int, default=1, help='1: synthetic COCO-shaped batches; 0: COCO from --anno-dir / --img-dir')
    parser.add_argument('--anno-dir', default='data/annotations')
    parser.add_argument('--img-dir', default='data')
    parser.add_argument('--data-type', default='2017')
    parser.add_argument('--num-workers', type=int, default=8, help='decode / transform threads of the COCO loader')
    parser.add_argument('--max-gt', type=int, default=0, help='instances kept per image (0: all; static shapes when > 0)')
    parser.add_argument('--image-size', type=int, nargs=2, default=[800, 800])
    parser.add_argument('--gemm-arithmetic', default='bf16x6_behind_backbone', choices=['f32', 'bf16x6_behind_backbone', 'bf16x6_backward', 'bf16x6'],
                        help='arithmetic of the convolution GEMMs (model/fpn_maskrcnn_train_chain.py GEMM_ARITHMETIC): float32 tensors and float32 '
                             'accumulation in all three; bf16x6 = float32-accurate three-plane emulation on the bf16 MFMA')
    parser.add_argument('--log-interval', type=int, default=100)
    parser.add_argument('--snapshot-interval', type=int, default=5000)
    parser.add_argument('--lr-shift-interval', type=int, default=0, help='iterations between lr x0.1 (reference: 2 epochs)')
    parser.add_argument('--grad-average', type=int, default=0, help='multi GPU: 1 = average the gradients over the ranks (extension); '
                                                                   '0 = sum them with the un-scaled lr, as the reference does')
    parser.add_argument('--resume', default='', help='trainer_<iteration>.pt written next to the NPZ snapshots: parameters, momentum, '
                                                     'BN statistics, sampler seeds, iteration and lr - continues bit-identically')
    parser.add_argument('--profile', type=int, nargs=2, default=None, metavar=('FIRST', 'LAST'),
                        help='bracket iterations FIRST..LAST with roctx ranges (step / forward+backward / update) for '
                             '`rocprofv3 --marker-trace --kernel-trace -- python3 train.py ...`')
    return parser


def run(args, keypoints=False):
    from chainer_maskrcnn.model.maskrcnn import MaskRCNN
    from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss, calc_keypoint_loss
    from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay
    from chainer_maskrcnn.utils.synthetic import make_batch
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', args.gpu))
    ndev = max(1, torch.cuda.device_count())
    lws = int(os.environ.get('LOCAL_WORLD_SIZE', 1))
    if lws > ndev and ndev != 1:        # (ndev == 1: a launcher that shows every rank only its own GPU)
        raise SystemExit('train.py: %d local ranks but %d visible GPUs - one process per GPU is required' % (lws, ndev))
    dev = torch.device('cuda', local % ndev if 'LOCAL_RANK' in os.environ else local)
    if world > 1:       # one process per GPU: host threads (enqueue loop, loader workers, RCCL proxy) on the GPU's NUMA node
        from chainer_maskrcnn.utils.affinity import pin_rank
        cpus = pin_rank(local, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
        if cpus:
            print('rank %d: pinned to %d cores (%d..%d)' % (rank, len(cpus), cpus[0], cpus[-1]))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        from chainer_maskrcnn.optimizers import init_process_group
        init_process_group('nccl')
    if keypoints:
        n_fg, K = 1, 20 if args.dataset == 'depth' else 17          # DepthDataset.n_keypoints / COCOKeypointsLoader.n_keypoints
        faster_rcnn = MaskRCNN(n_fg_class=n_fg, n_keypoints=K, backbone=args.backbone, head_arch=args.head_arch, n_mask_convs=args.n_mask_convs,
                               min_size=args.min_size, max_size=args.max_size, device=dev)
        loss_fun = calc_keypoint_loss
        if K != 17:         # train_keypoints.py:122: lambda x, y, z, w: calc_mask_loss(x, y, z, w, num_keypoints=n_keypoints)
            loss_fun = lambda x, y, z, w: calc_keypoint_loss(x, y, z, w, num_keypoints=K)
            loss_fun.fused_kind = calc_keypoint_loss.fused_kind
        model = FPNMaskRCNNTrainChain(faster_rcnn, mask_loss_fun=loss_fun, binary_mask=False, gemm_arithmetic=args.gemm_arithmetic)
    else:
        n_fg, K = 80, None
        if os.path.exists(args.label_file):
            with open(args.label_file) as f:
                n_fg = len(f.read().strip().split('\n'))
        faster_rcnn = MaskRCNN(n_fg_class=n_fg, backbone=args.backbone, head_arch=args.head_arch, device=dev)
        model = FPNMaskRCNNTrainChain(faster_rcnn, mask_loss_fun=calc_mask_loss, gemm_arithmetic=args.gemm_arithmetic)
    labels = None
    if not keypoints and os.path.exists(args.label_file):
        with open(args.label_file) as f:
            labels = f.read().strip().split('\n')
    faster_rcnn.use_preset('evaluate')
    if args.resnet50_npz:           # ImageNet initialisation of the bottom-up pathway (before --weight, which may override it)
        from chainer_maskrcnn.utils import chainer_npz
        n = len(chainer_npz.load_resnet50_npz(args.resnet50_npz, faster_rcnn))
        if rank == 0:
            print('ResNet-50 snapshot: %d arrays loaded from %s' % (n, args.resnet50_npz))
    if args.weight and os.path.exists(args.weight):
        load_npz(args.weight, faster_rcnn)
    optimizer = MomentumSGD(lr=args.lr, momentum=0.9)
    optimizer.setup(model)
    optimizer.add_hook(WeightDecay(rate=0.0005))
    if world > 1:
        optimizer.enable_data_parallel(average=bool(args.grad_average))
    bs = args.batch_size
    H, W = args.image_size
    os.makedirs(args.out, exist_ok=True)
    log = open(os.path.join(args.out, 'log'), 'a') if rank == 0 else None
    keys = ('loss', 'rpn_loc_loss', 'rpn_cls_loss', 'roi_loc_loss', 'roi_cls_loss', 'mask_loss')
    acc = {k: 0.0 for k in keys}
    loader = None
    resume = torch.load(args.resume, map_location='cpu', weights_only=False) if args.resume else None
    if not args.synthetic:          # train.py:111-126: COCOMaskLoader(category_filter=labels, data_type='2017') + Transform
        from chainer_maskrcnn.dataset.coco_dataset import COCOMaskLoader, COCOKeypointsLoader
        from chainer_maskrcnn.dataset.transforms import RawTransform
        from chainer_maskrcnn.dataset.loader import BatchLoader
        if keypoints and args.dataset == 'depth':       # train_keypoints.py:103-109, 135: DepthDataset -> DepthTransformer -> Transform, on the host
            from chainer_maskrcnn.dataset.depth_dataset import DepthDataset, DepthTransformer
            from chainer_maskrcnn.dataset.transforms import KeypointTransform
            data = DepthDataset(path=args.depth_list, root=args.depth_root)
            jitter, kt = DepthTransformer(np.random.RandomState(4321 + rank)), KeypointTransform(faster_rcnn)
            tf = lambda ex: kt(jitter(ex))
        elif keypoints:
            data = COCOKeypointsLoader(anno_dir=args.anno_dir, img_dir=args.img_dir, data_type=args.data_type)
            tf = RawTransform(faster_rcnn, keypoints=True)      # host decodes, the GPU resizes (dataset/loader.py)
        else:
            data = COCOMaskLoader(anno_dir=args.anno_dir, img_dir=args.img_dir, data_type=args.data_type, category_filter=labels)
            tf = RawTransform(faster_rcnn)
        loader = BatchLoader(data, tf, batch_size=bs, shuffle=True, seed=1234, rank=rank, world=world,
                             num_workers=args.num_workers, max_gt=args.max_gt or None, keypoints=keypoints, device=dev,
                             start_ticket=_rank_ticket(resume['loader_ticket'], rank) if resume else 0)
    pool = []
    if loader is None:
        for j in range(8):
            b = make_batch((j + 1) * world + rank, bs, H, W, G=8, n_fg_class=n_fg, n_keypoints=K)
            pool.append([torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'keypoints' if keypoints else 'masks')])
    first_it = 1
    if resume is not None:
        optimizer.load_state_dict(resume['optimizer'])
        first_it = resume['iteration'] + 1
    rtx = _Roctx() if args.profile else None
    t0 = time.time()
    for it in range(first_it, args.iteration + 1):
        if loader is not None:
            b = next(loader)
            batch = [b[k] for k in ('imgs', 'bboxes', 'labels', 'keypoints' if keypoints else 'masks')]
            # every image keeps its own resize factor and its own size inside the padded batch (the reference runs batch 1
            # per process, so its img_size / scale are always those of THE image: fpn_maskrcnn_train_chain.py:60-70)
            scale, sizes = b['scales'], b['sizes']
        else:           # a small pool of device-resident synthetic batches, cycled (generating one per step is host-bound)
            batch = pool[it % len(pool)]
            scale, sizes = 1.0, None
        if rtx is not None and args.profile[0] <= it <= args.profile[1]:
            with rtx.range('step %d' % it):
                with rtx.range('forward+backward'):
                    if optimizer.sync is not None:
                        optimizer.sync.begin()
                    loss = model(*batch, scale, img_sizes=sizes)
                    model.unit_upstream = True
                    try:
                        loss.backward()
                    finally:
                        model.unit_upstream = False
                with rtx.range('all-reduce wait + sgd'):
                    optimizer.update()
        else:
            optimizer.update(model, *batch, scale, img_sizes=sizes)
        if it % args.log_interval == 0 or it == args.iteration:       # one device->host sync per log interval
            obs = {k: float(v) for k, v in model.observation.items()}
            if any(not np.isfinite(v) for v in obs.values()):
                raise FloatingPointError('non-finite loss at iteration %d: %r' % (it, obs))
            entry = {'iteration': it, 'lr': optimizer.lr, 'elapsed_time': time.time() - t0,
                     'images/sec': (it - first_it + 1) * bs * world / (time.time() - t0)}
            entry.update({'main/' + k: v for k, v in obs.items()})
            if rank == 0:
                log.write(json.dumps(entry) + '\n')
                log.flush()
                print(entry)
        if args.lr_shift_interval and it % args.lr_shift_interval == 0:
            optimizer.lr *= 0.1                                       # ExponentialShift('lr', 0.1), train.py:139-140
        if it % args.snapshot_interval == 0:
            # every rank's data position goes into the trainer state: tickets count popped examples (skipped empty ones
            # included), so ranks can stand at different positions of their shards - a resumed rank continues from ITS own
            tickets = _all_rank_tickets(loader.ticket if loader is not None else 0, world, dev)
            if rank == 0:
                save_npz(os.path.join(args.out, 'model_%d.npz' % it), faster_rcnn)      # snapshot_object, train.py:134-137
                torch.save({'iteration': it, 'optimizer': optimizer.state_dict(), 'loader_ticket': tickets},
                           os.path.join(args.out, 'trainer_%d.pt' % it))
    if world > 1:
        torch.distributed.destroy_process_group()


def _all_rank_tickets(ticket, world, dev):
    """Every rank's loader ticket, in rank order (a collective when world > 1: called by all ranks)."""
    if world == 1:
        return [int(ticket)]
    t = torch.tensor([int(ticket)], dtype=torch.int64, device=dev if torch.distributed.get_backend() == 'nccl' else 'cpu')
    out = [torch.zeros_like(t) for _ in range(world)]
    torch.distributed.all_gather(out, t)
    return [int(o.item()) for o in out]


def _rank_ticket(saved, rank):
    """A trainer state holds one ticket per rank (older files: a single number = rank 0's)."""
    if isinstance(saved, (list, tuple)):
        return int(saved[rank]) if rank < len(saved) else int(saved[0])
    return int(saved)


class _Roctx(object):
    """roctx ranges through libroctx64 (ROCm's marker API; rocprofv3 --marker-trace shows them over the kernel trace)."""

    def __init__(self):
        import ctypes
        self.lib = None
        for name in ('librocprofiler-sdk-roctx.so', 'libroctx64.so'):
            try:
                self.lib = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if self.lib is None:
            raise RuntimeError('--profile: neither librocprofiler-sdk-roctx.so nor libroctx64.so can be loaded')

    def range(self, name):
        rtx = self

        class _R(object):
            def __enter__(self_):
                rtx.lib.roctxRangePushA(name.encode())

            def __exit__(self_, *a):
                rtx.lib.roctxRangePop()
        return _R()


def save_npz(path, faster_rcnn):
    """Chainer-NPZ key names and array layouts (chainer_maskrcnn/utils/chainer_npz.py), like snapshot_object."""
    from chainer_maskrcnn.utils import chainer_npz
    chainer_npz.save_npz(path, faster_rcnn)


def load_npz(path, faster_rcnn):
    """strict=False like train.py:99-101: keys present in the file are loaded, the rest keep their initial values."""
    from chainer_maskrcnn.utils import chainer_npz
    return chainer_npz.load_npz(path, faster_rcnn, strict=False)


def main():
    args = build_parser().parse_args()
    print('lr:{}'.format(args.lr))
    print('output:{}'.format(args.out))
    print('iteration::{}'.format(args.iteration))
    print('backbone architecture:{}'.format(args.backbone))
    print('head architecture:{}'.format(args.head_arch))
    run(args)


if __name__ == '__main__':
    main()
