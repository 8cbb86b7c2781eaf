"""Child process of tests/test_dp_gpu.py (started by tests/dp/launcher.py; never imported by pytest).

    worker.py dp  <rank> <world> <port> <outdir>    one data-parallel rank: batch of ONE image, gloo all-reduce of the
                                                    flat gradient buffer (RCCL refuses two ranks on one GPU), 3 steps
    worker.py emu <outdir>                          single process: the same 3 steps computed as g(img0) + g(img1)
    worker.py rccl1 <port> <outdir>                 ONE rank on RCCL (backend 'nccl'): the code path an N-GPU job executes -
                                                    optimizers.init_process_group (high-priority collective streams),
                                                    the rank-0 broadcast, device-buffer bucket all-reduces on the side
                                                    stream, timing_report(), finish() - against the same steps without
                                                    any synchroniser (a SUM over one rank is the identity)
Both save the gradient buffer after step 1 and the parameters after step 3."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'chainer-maskrcnn_amd'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from chainer_maskrcnn.model.maskrcnn import MaskRCNN  # noqa: E402
from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss  # noqa: E402
from chainer_maskrcnn.optimizers import MomentumSGD, WeightDecay  # noqa: E402
from chainer_maskrcnn.utils.synthetic import make_batch  # noqa: E402

DEV = 'cuda:0'
SHRINK = dict(stages=(1, 1, 1, 1), width_div=2)
STEPS = 3


def image(r):
    b = make_batch(40 + r, 1, 128, 160, G=3)
    b['bboxes'][:, :, 2:] = np.minimum(b['bboxes'][:, :, 2:], [128, 160])
    return [torch.from_numpy(b[k]).to(DEV) for k in ('imgs', 'bboxes', 'labels', 'masks')]


def make_chain(model, r):
    chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss)
    chain.proposal_target_creator.set_seed(100 + r)
    chain.anchor_target_creator.set_seed(200 + r)
    return chain


def make_opt(link):
    opt = MomentumSGD(lr=1e-2, momentum=0.9).setup(link)
    opt.add_hook(WeightDecay(0.0005))
    return opt


def run_dp(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.distributed.init_process_group('gloo', rank=rank, world_size=world)
    try:
        # every rank starts from its OWN seed: enable_data_parallel() must broadcast rank 0's replica
        model = MaskRCNN(n_fg_class=80, device=DEV, seed=7 + 13 * rank, _test_shrink=SHRINK)
        chain = make_chain(model, rank)
        opt = make_opt(chain)
        res = {'p_init': model.ps.params.cpu().clone()}       # this rank's own initialisation
        opt.enable_data_parallel(bucket_bytes=1 << 20)        # several buckets on the small test network
        assert len(opt.sync.buckets) > 3
        batch = image(rank)
        res['p0'] = model.ps.params.cpu().clone()             # after the rank-0 broadcast
        for s in range(STEPS):
            opt.update(chain, *batch, 1.0)
            if s == 0:
                res['grads'] = model.ps.grads.cpu().clone()
                res['loss0'] = float(chain.observation['loss'])
        res['params'] = model.ps.params.cpu().clone()
        torch.save(res, os.path.join(out, 'dp_rank%d.pt' % rank))
    finally:
        torch.distributed.destroy_process_group()


def run_emu(out):
    model = MaskRCNN(n_fg_class=80, device=DEV, seed=7, _test_shrink=SHRINK)
    chains = [make_chain(model, r) for r in range(2)]
    opt = make_opt(model)
    batches = [image(r) for r in range(2)]
    res = {'p0': model.ps.params.cpu().clone(), 'losses': []}
    for s in range(STEPS):
        gs = []
        for c, b in zip(chains, batches):
            c.backward_follows = c.unit_upstream = True        # what MomentumSGD.update tells the chain (the ranks go through update):
            try:                                               # the RPN's backward runs early, the loss seed is 1
                c(*b, 1.0).backward()
            finally:
                c.backward_follows = c.unit_upstream = False
            torch.cuda.synchronize()
            gs.append(model.ps.grads.clone())
            if s == 0:
                res['losses'].append(float(c.observation['loss']))
        model.ps.grads.copy_(gs[0] + gs[1])            # SUM over ranks, un-scaled learning rate (train.py:117-121)
        if s == 0:
            res['grads'] = model.ps.grads.cpu().clone()
        opt.update()
    res['params'] = model.ps.params.cpu().clone()
    torch.save(res, os.path.join(out, 'emu.pt'))


def run_rccl1(port, out):
    from chainer_maskrcnn import optimizers
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)

    def steps(sync):
        model = MaskRCNN(n_fg_class=80, device=DEV, seed=7, _test_shrink=SHRINK)
        chain = make_chain(model, 0)
        opt = make_opt(chain)
        rep = None
        if sync:
            opt.enable_data_parallel(bucket_bytes=1 << 20, sync_single_rank=True)
            assert opt.sync.active and opt.sync.world == 1 and len(opt.sync.buckets) > 3
            opt.sync.timing = True
        batch = image(0)
        for s in range(STEPS):
            opt.update(chain, *batch, 1.0)
            if s == 0:
                g = model.ps.grads.cpu().clone()
        torch.cuda.synchronize()
        if sync:
            rep = opt.sync.timing_report()
            assert opt.sync.next == len(opt.sync.buckets)            # every bucket went through the collective
        return g, model.ps.params.cpu().clone(), float(chain.observation['loss']), rep

    g0, p0, l0, _ = steps(False)
    optimizers.init_process_group('nccl', rank=0, world_size=1)
    try:
        assert torch.distributed.get_backend() == 'nccl'
        g1, p1, l1, rep = steps(True)
    finally:
        torch.distributed.destroy_process_group()
    torch.save(dict(g0=g0, p0=p0, l0=l0, g1=g1, p1=p1, l1=l1, report=rep), os.path.join(out, 'rccl1.pt'))


if __name__ == '__main__':
    if sys.argv[1] == 'rccl1':
        run_rccl1(int(sys.argv[2]), sys.argv[3])
    elif sys.argv[1] == 'dp':
        run_dp(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    else:
        run_emu(sys.argv[2])
