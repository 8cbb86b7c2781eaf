#!/usr/bin/env python3
"""bench.py - headline measurement (contract in the task statement; BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload roialign|step]

Workloads
  roialign  BASELINE.json configs[1]: roi_align_2d fwd+bwd, 512 RoIs on a 256x200x272 map,
            7x7, fp32.  One "step" = one forward + one backward over the 512-RoI batch, inputs
            resident in HBM.  value = algorithmic ROIAlign-backward GB/s (the second half of
            BASELINE.json's metric), whole job (sum over ranks; ranks run independent batches).
  step      BASELINE.json configs[2]: full ResNet50-FPN Mask R-CNN training step, bs=2/GPU,
            1024x1024 (images/sec) - selected automatically once the training path is built.  Runs with the
            shipped GEMM arithmetic of train.py (config.gemm_arithmetic says which MFMA instruction every
            pass uses); the all-float32-MFMA step is always measured in the same process and printed as
            config.images_per_sec_f32_mfma.
  keypoint  BASELINE.json configs[4] per-GPU shape: the Keypoint R-CNN step of train_keypoints.py
            (17 keypoints, 56x56 heat maps), same batch / image size; a secondary line, never the default.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel, measured with HIP
events inside the timed region on the launch stream; `cpu_baseline` is the NumPy oracle timed
on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'chainer-maskrcnn_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = 'images/sec (1024^2 COCO, bs=2/GPU) at 1/2/4/8 MI355X; ROIAlign bwd HBM GB/s'


def roialign_inputs(seed_shift=0):
    from chainer_maskrcnn.utils.synthetic import config2_inputs
    x, yx, gy = config2_inputs()
    if seed_shift:          # distinct batches per rank
        rs = np.random.RandomState(100 + seed_shift)
        yx = yx[rs.permutation(yx.shape[0])]
    return x, yx, gy


def _pmc_traffic(kernel):
    """(HBM bytes per launch, source) from the newest committed rocprofv3 --pmc summary (profiles/), or (None, None).  The
    counters need rocprofv3, so this is a constant from the repository, not something this run measured - the line says so."""
    from chainer_maskrcnn.bench_step import latest_profile
    f, d = latest_profile('roialign_pmc_traffic.json')
    try:
        key = [k for k in d if k.startswith(kernel)][0]
        return d[key]['hbm_bytes'], {'file': f, 'collected_at_commit': d.get('_commit'),
                                     'kind': 'constant read from the committed rocprofv3 --pmc summary, not collected by this run'}
    except Exception:
        return None, None


def bench_roialign(args, rank, world):
    from chainer_maskrcnn.functions.roi_align_2d_yx import _roi_align_2d_yx
    from chainer_maskrcnn.functions.roi_align.roi_align_2d import roi_align_2d
    from chainer_maskrcnn import _hip
    dev = torch.device('cuda', _local_device())
    torch.cuda.set_device(dev)
    x, yx, gy = roialign_inputs(rank)
    N, C, H, W = x.shape
    R, _, PH, PW = gy.shape
    xt = torch.from_numpy(x).to(dev).contiguous(memory_format=torch.channels_last)
    rois_xy = torch.from_numpy(yx[:, [0, 2, 1, 4, 3]].copy()).to(dev)
    gyt = torch.from_numpy(gy).to(dev).contiguous(memory_format=torch.channels_last)
    y = torch.empty((R, C, PH, PW), device=dev).contiguous(memory_format=torch.channels_last)
    gx = torch.empty_like(xt)
    lib = _hip.lib()
    algo_bytes = 4 * (N * C * H * W + R * C * PH * PW) + 20 * R

    # caller-owned scratch of the forward: the map-order permutation of the RoIs (ranking kernel + forward kernel per call)
    nbf = lib.mrcnn_roi_align_fwd_workspace_bytes(R)
    wsf = torch.empty((max(nbf, 1),), dtype=torch.uint8, device=dev)

    def fwd(sr=2):
        _hip.check(lib.mrcnn_roi_align_fwd_ws_f32(_hip.ptr(xt), 1, N, C, H, W, _hip.ptr(rois_xy), R, PH, PW,
                                                  0.25, sr, _hip.ptr(y), _hip.ptr(wsf), nbf, _hip.stream_ptr()))

    # caller-owned scratch of the backward (per-RoI sample tables of the table-driven kernel), sized by the library's query
    nb = max(lib.mrcnn_roi_align_bwd_workspace_bytes(N, C, H, W, R, PH, PW, 2), lib.mrcnn_roi_align_bwd_workspace_bytes(N, C, H, W, R, PH, PW, 0))
    ws = torch.empty((max(nb, 1),), dtype=torch.uint8, device=dev)

    def bwd_fused(sr=2):
        _hip.check(lib.mrcnn_roi_align_bwd_ws_f32(_hip.ptr(gyt), 1, N, C, H, W, _hip.ptr(rois_xy), R, PH, PW,
                                                  0.25, sr, _hip.ptr(gx), _hip.ptr(ws), max(nb, 1), _hip.stream_ptr()))

    # (ABI v9) the backward in two launches: the entry lists of the RoIs (geometry only: no gy) are built ahead - in a training step beside the
    # forward pass, here as its own timed launch - and the backward streams gy along them (same bits as the fused kernel,
    # tests/test_roi_align_gpu.py).  `verified`: the plan's status was read back once, outside the timed region (header valid, no tile
    # flagged): the backward is then the lean kernel alone; unverified, a second launch follows for whatever the plan could not hold.
    import ctypes
    Hs, Ws, sc = (ctypes.c_int * 1)(H), (ctypes.c_int * 1)(W), (ctypes.c_float * 1)(0.25)
    gxp = (ctypes.c_void_p * 1)(gx.data_ptr())
    pb = lib.mrcnn_roi_align_fpn_bwd_plan_bytes(Hs, Ws, 1, N, R, PH, PW, 0)
    plan = torch.zeros((max(pb, 256),), dtype=torch.uint8, device=dev)

    def plan_build():
        _hip.check(lib.mrcnn_roi_align_fpn_bwd_plan_f32(Hs, Ws, sc, 1, N, C, _hip.ptr(rois_xy), None, R, PH, PW, 2, 0, _hip.ptr(plan), pb, _hip.stream_ptr()))

    def bwd_planned(verified):
        _hip.check(lib.mrcnn_roi_align_fpn_bwd_planned_f32(_hip.ptr(gyt), gxp, Hs, Ws, sc, 1, N, C, _hip.ptr(rois_xy), None, R, PH, PW, 2, 0, None, 0,
                                                           _hip.ptr(plan), pb, int(verified), _hip.stream_ptr()))
    plan_build()
    st3 = (ctypes.c_int * 3)()
    _hip.check(lib.mrcnn_roi_align_bwd_plan_status(_hip.ptr(plan), plan.numel(), st3, _hip.stream_ptr()))
    plan_ok = bool(st3[0]) and st3[1] == 0

    # What is timed and reported as `value` / `roofline` is what roi_align_2d(...).backward() and the training step launch: the FUSED wave
    # kernel (VERDICT r5 / ADVICE r5: the two-launch form below is faster for the backward alone and slower for the pair, so it does not
    # ship; its numbers stay in roi_align_bwd_forms_us).
    bwd = bwd_fused

    for _ in range(args.warmup):
        fwd(); bwd()
    K = args.steps
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
    sync_all(world)
    t0 = time.perf_counter()
    for k in range(K):          # a step of this workload: forward, backward - the operator's own launches
        ev[k][0].record(); fwd(); ev[k][1].record(); bwd(); ev[k][2].record()
    sync_all(world)
    dt = time.perf_counter() - t0
    # a second run with the ADAPTIVE sampling grid (sampling_ratio 0 = ceil(roi / pooled) samples
    # per bin; the reference's configuration is 2) - reported, not part of `value`
    ad = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for _ in range(3):
        fwd(0); bwd(0)
    ad[0].record()
    for _ in range(20):
        fwd(0)
    ad[1].record()
    for _ in range(20):
        bwd(0)
    ad[2].record()
    torch.cuda.synchronize()
    adaptive = {'fwd_avg_launch_us': round(ad[0].elapsed_time(ad[1]) / 20 * 1e3, 2), 'bwd_avg_launch_us': round(ad[1].elapsed_time(ad[2]) / 20 * 1e3, 2)}
    fwd_ms = np.array([ev[k][0].elapsed_time(ev[k][1]) for k in range(K)])
    bwd_ms = np.array([ev[k][1].elapsed_time(ev[k][2]) for k in range(K)])
    dt = max_over_ranks(dt, world, dev)
    # Kernel duration for the roofline: an event pair around ONE launch also contains the dispatch latency of the launch and
    # of the closing event's packet (~3.5 us here against a ~28 us kernel; rocprofv3 --kernel-trace of this same command,
    # profiles/, gives the kernel itself).  So the same launches are also timed back to back, GROUP launches per event
    # pair: that average is what `roofline.achieved` uses, the per-launch bracket stays next to it.
    GROUP, NG = 10, max(2, K // 10)

    def back_to_back(fn):
        es = [torch.cuda.Event(enable_timing=True) for _ in range(NG + 1)]
        es[0].record()
        for g in range(NG):
            for _ in range(GROUP):
                fn()
            es[g + 1].record()
        torch.cuda.synchronize()
        return np.array([es[g].elapsed_time(es[g + 1]) / GROUP for g in range(NG)])
    bwd_b2b, fwd_b2b = back_to_back(bwd), back_to_back(fwd)
    plan_b2b = back_to_back(plan_build)
    lean_b2b = back_to_back(lambda: bwd_planned(True)) if plan_ok else bwd_b2b
    unver_b2b = back_to_back(lambda: bwd_planned(False)) if plan_ok else bwd_b2b
    pair_fused = back_to_back(lambda: (fwd(), bwd_fused()))
    pair_planned = back_to_back(lambda: (fwd(), plan_build(), bwd_planned(False))) if plan_ok else pair_fused
    bwd_avg_s = float(bwd_b2b.mean()) * 1e-3
    fwd_avg_s = float(fwd_b2b.mean()) * 1e-3
    bwd_gbps = algo_bytes / bwd_avg_s / 1e9
    out = {
        'metric': METRIC, 'value': round(bwd_gbps * world, 2), 'unit': 'GB/s (ROIAlign bwd, algorithmic bytes)',
        'n_gpus': world, 'steps': K, 'warmup': args.warmup, 'ms_per_step': round(dt / K * 1e3, 5),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'configs[1] roi_align_2d fwd+bwd microbench: 512 RoIs, x=(1,256,200,272) NHWC, '
                               '7x7, sampling 2x2, spatial_scale 0.25', 'rois_per_step': R * world,
                   'parallelism': 'independent batch per rank, no collective'},
        'roofline': {'bound': 'hbm', 'kernel': 'k_roi_align_bwd_waves (the fused kernel: what roi_align_2d(...).backward() and the training step launch)', 'achieved': round(bwd_gbps, 2),
                     'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': round(bwd_gbps / HBM_PEAK_GBPS, 4),
                     'traffic': _pmc_traffic('k_roi_align_bwd_waves')[0], 'traffic_source': _pmc_traffic('k_roi_align_bwd_waves')[1],
                     'algorithmic_bytes_per_launch': algo_bytes,
                     # informational, not `frac`: the counter traffic over the same duration - what the launch actually asks of the HBM
                     # (the library's pure streaming kernels reach 5.4 - 5.7 TB/s of the 8 on this part: DESIGN 3.2 / 5.8)
                     'counter_traffic_GBps': (round(_pmc_traffic('k_roi_align_bwd_waves')[0] / bwd_avg_s / 1e9, 1)
                                              if _pmc_traffic('k_roi_align_bwd_waves')[0] else None),
                     'avg_launch_us': round(bwd_avg_s * 1e6, 3), 'median_launch_us': round(float(np.median(bwd_b2b)) * 1e3, 3),
                     'event_pair_per_launch_us': round(float(bwd_ms.mean()) * 1e3, 3),
                     'note': 'avg_launch_us: %d groups of %d back-to-back launches per HIP event pair; event_pair_per_launch_us: one '
                             'event pair per launch inside the timed fwd+bwd loop (includes ~3.5 us of dispatch latency)' % (NG, GROUP)},
        'roi_align_fwd': {'avg_launch_us': round(fwd_avg_s * 1e6, 3), 'event_pair_per_launch_us': round(float(fwd_ms.mean()) * 1e3, 3),
                          'achieved_GBps': round(algo_bytes / fwd_avg_s / 1e9, 2),
                          'frac': round(algo_bytes / fwd_avg_s / 1e9 / HBM_PEAK_GBPS, 4), 'kernel': 'k_roi_map_order + k_roi_align_fwd_rows (RoIs walked in map order)',
                          'traffic': _pmc_traffic('k_roi_align_fwd')[0]},
        'roi_align_adaptive_sampling': adaptive,
        'roi_align_bwd_forms_us': {
            'fused_wave_kernel_shipped': round(bwd_avg_s * 1e6, 3), 'fwd': round(fwd_avg_s * 1e6, 3),
            'pair_fwd_plus_fused_shipped': round(float(pair_fused.mean()) * 1e3, 3),
            'opt_in_two_launch_form': {
                'plan_build': round(float(plan_b2b.mean()) * 1e3, 3),
                'lean_kernel_plus_fallback_launch': round(float(unver_b2b.mean()) * 1e3, 3),
                'lean_kernel_alone_plan_verified_by_a_synchronising_query': round(float(lean_b2b.mean()) * 1e3, 3),
                'lean_alone_frac_of_hbm_peak': round(algo_bytes / (float(lean_b2b.mean()) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                'pair_fwd_plus_plan_plus_lean_plus_fallback': round(float(pair_planned.mean()) * 1e3, 3),
                'plan_status': {'header_valid': bool(st3[0]), 'tiles_flagged': int(st3[1]), 'pool_nodes_used': int(st3[2]), 'plan_bytes': int(pb)}},
            'note': 'value / roofline = the fused kernel, the operator default.  The two-launch form (entry lists built from the RoIs by '
                    'k_roi_align_bwd_waves<MODE 1>, then k_roi_align_bwd_lean along them: mrcnn_roi_align_fpn_bwd_plan_f32 + _planned_f32, same '
                    'bits) makes the backward ALONE faster and the forward + backward PAIR slower, so it is an opt-in entry point, not the default'},
    }
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline_roialign(x, yx, gy, algo_bytes)
    return out


def cpu_baseline_roialign(x, yx, gy, algo_bytes):
    """NumPy oracle ("port" of the reference algorithm) on the host: the same 512-RoI batch, once."""
    from oracle import roi_align as ora
    xy = yx[:, [0, 2, 1, 4, 3]]
    t0 = time.perf_counter()
    ora.roi_align_fwd(x, xy, 7, 7, 0.25, 2)
    t1 = time.perf_counter()
    ora.roi_align_bwd(gy, xy, x.shape, 0.25, 2)
    t2 = time.perf_counter()
    return {'value': round(algo_bytes / (t2 - t1) / 1e9, 4), 'unit': 'GB/s (ROIAlign bwd, algorithmic bytes)',
            'cores': 1, 'kind': 'port',
            'sample': 'full configs[1] batch (512 RoIs) once: fwd %.2f s, bwd %.2f s, single-thread NumPy oracle'
                      % (t1 - t0, t2 - t1),
            'fwd_GBps': round(algo_bytes / (t1 - t0) / 1e9, 4), 'host_cpus': os.cpu_count()}


def cpu_baseline_step(model, dev):
    """The CPU restatement (oracle/model.py, kind 'port') of ONE training step on ONE 1024x1024 image (the
    reference's own batch size), float32, timed on this box's host cores: forward + backward of the whole network with
    the sampled targets of a device step on the same image.  Checker code is timed here, never shipped."""
    from oracle import model as om
    from chainer_maskrcnn.model.fpn_maskrcnn_train_chain import FPNMaskRCNNTrainChain, calc_mask_loss
    from chainer_maskrcnn.utils.synthetic import make_batch
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    b = make_batch(7, 1, 1024, 1024, G=8)
    chain = FPNMaskRCNNTrainChain(model, mask_loss_fun=calc_mask_loss, mask_rows='all')
    imgs, bb, lab, masks = (torch.from_numpy(b[k]).to(dev) for k in ('imgs', 'bboxes', 'labels', 'masks'))
    chain(imgs, bb, lab, masks, 1.0)
    t = {k: v.cpu().numpy() for k, v in chain.targets.items() if torch.is_tensor(v)}
    t['gt_rpn_loc'], t['gt_rpn_label'] = (x.cpu().numpy() for x in chain.rpn_targets)
    t['mask_rois_xy5'], t['mask_levels'], t['mask_label'] = (x.cpu().numpy() for x in chain.mask_inputs)
    om.set_dtype(torch.float32)
    try:
        ps = model.ps
        params = {n: ps.p(n).detach().cpu().requires_grad_(True) for n in ps.names()}
        oracle = om.OracleStep(params, tuple(len(s) for s in model.extractor.stages), model.head.n_class, model.head.LOC0)
        img4 = torch.cat([torch.from_numpy(b['imgs']).permute(0, 2, 3, 1), torch.zeros((1, 1024, 1024, 1))], -1)
        t0 = time.perf_counter()
        out = oracle.losses(img4, t)
        t1 = time.perf_counter()
        sum(out[k] for k in ('rpn_loc_loss', 'rpn_cls_loss', 'roi_loc_loss', 'roi_cls_loss', 'mask_loss')).backward()
        t2 = time.perf_counter()
    finally:
        om.set_dtype(torch.float64)
    return {'value': round(1.0 / (t2 - t0), 4), 'unit': 'images/sec', 'cores': threads, 'kind': 'port',
            'sample': 'one full training step (fwd %.1f s + bwd %.1f s) on ONE 1024x1024 image, 256 sampled RoIs, fp32: '
                      'torch-CPU convolutions on %d threads + single-thread NumPy ROIAlign oracle; targets taken from a '
                      'device step (samplers not timed)' % (t1 - t0, t2 - t1, threads),
            'host_cpus': os.cpu_count()}


def _local_device():
    """LOCAL_RANK, or 0 for every rank when MRCNN_BENCH_SINGLE_DEVICE=1 (functional check on a 1-GPU box).  A launcher that shows every
    rank only its own GPU (HIP_VISIBLE_DEVICES=k per rank) leaves one visible device: LOCAL_RANK is taken modulo the visible count, and
    two ranks that really share a device are refused with a clear message by optimizers.rccl_evidence, not by an invalid ordinal here."""
    if os.environ.get('MRCNN_BENCH_SINGLE_DEVICE') == '1':
        return 0
    return int(os.environ.get('LOCAL_RANK', 0)) % max(1, torch.cuda.device_count())


def sync_all(world):
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
        torch.cuda.synchronize()


def max_over_ranks(v, world, dev):
    if world == 1:
        return v
    t = torch.tensor([v], dtype=torch.float64, device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--workload', default='auto', choices=['auto', 'roialign', 'step', 'keypoint'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--mask-rows', default='all', choices=['positives', 'all'])
    ap.add_argument('--gemm-arithmetic', default=None, choices=['f32', 'bf16x6_behind_backbone', 'bf16x6_backward', 'bf16x6'],
                    help='arithmetic of the convolution GEMMs of the step workloads (default: the shipped training default, train.py)')
    ap.add_argument('--graph', type=int, default=0, help='capture the step into a HIP graph (single GPU; measured slower than eager multi-stream launches on ROCm 7.2, so off by default)')
    args = ap.parse_args()
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    # one process per GPU: this rank's host threads stay on cores of its GPU's NUMA node (ranks sharing a node split it)
    from chainer_maskrcnn.utils.affinity import pin_rank
    cpus = pin_rank(int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('LOCAL_WORLD_SIZE', world))) if world > 1 else None
    args.cpu_affinity = None if cpus is None else '%d cores: %d..%d' % (len(cpus), cpus[0], cpus[-1])
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(_local_device())
        # The line should show what RCCL saw (config.rccl.transport) without anybody having to ask for it: unless the launcher chose its
        # own settings, the init / topology / transport messages go to a per-process FILE (never to stdout, where the JSON line goes).
        # Rank 0 only (the one whose log is read), init-time subsystems only (nothing is logged during the timed steps), and the file is
        # removed once it has been parsed (optimizers.rccl_evidence) - ADVICE r4.
        if 'NCCL_DEBUG' not in os.environ and 'NCCL_DEBUG_FILE' not in os.environ and rank == 0:
            os.environ['NCCL_DEBUG'] = 'INFO'
            os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,GRAPH')
            os.environ['NCCL_DEBUG_FILE'] = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'mrcnn_rccl.%h.%p.log')
            os.environ['MRCNN_RCCL_LOG_IS_OURS'] = '1'          # rccl_evidence deletes a log this script asked for, never the launcher's
        # RCCL over xGMI; MRCNN_BENCH_BACKEND=gloo only for the 1-GPU functional check of the multi-process path
        from chainer_maskrcnn.optimizers import init_process_group
        init_process_group(os.environ.get('MRCNN_BENCH_BACKEND', 'nccl'))
    if args.gpus != world and rank == 0 and world > 1:
        print('warning: --gpus %d but WORLD_SIZE %d' % (args.gpus, world), file=sys.stderr)
    workload = args.workload
    if workload == 'auto':
        workload = 'roialign'
        try:
            from chainer_maskrcnn import train_step_available
            if train_step_available():
                workload = 'step'
        except ImportError:
            pass
    args.keypoints = workload == 'keypoint'
    if workload in ('step', 'keypoint'):
        from chainer_maskrcnn.bench_step import bench_step
        args.steps = args.steps or 20
        args.warmup = 3 if args.warmup is None else args.warmup
        args.mask_rows = args.mask_rows
        out, model, dev = bench_step(args, rank, world)
        if world == 1 and rank == 0 and not args.no_cpu_baseline and not args.keypoints:
            out['cpu_baseline'] = cpu_baseline_step(model, dev)
        if rank == 0:       # second half of BASELINE.json's metric: ROIAlign backward HBM GB/s on configs[1]
            ra = argparse.Namespace(steps=100, warmup=10, no_cpu_baseline=True)
            r = bench_roialign(ra, 0, 1)
            out['roi_align_microbench'] = {'workload': r['config']['workload'], 'bwd': r['roofline'], 'fwd': r['roi_align_fwd'],
                                           'forms_us': r['roi_align_bwd_forms_us']}
    else:
        args.steps = args.steps or 200
        args.warmup = 20 if args.warmup is None else args.warmup
        out = bench_roialign(args, rank, world)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        # every rank leaves together: rank 0 was still measuring the ROIAlign microbench while the others were done, and a rank that
        # tears its communicator down early must not race rank 0's store
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
