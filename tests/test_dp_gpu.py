"""Data parallelism on the real training chain: two ranks with ONE image each (gloo, both on the one
GPU of the test box - RCCL refuses two ranks on a device) against a single process that computes the same step as
g(img0) + g(img1):

  * SUM semantics with the un-scaled learning rate (train.py:117-121): the all-reduced gradient
    buffer of both ranks equals the sum of the two single-image gradients, bit for bit (deterministic kernels);
  * replicas that start from DIFFERENT seeds are made equal by the rank-0 broadcast in enable_data_parallel();
  * after 3 steps both ranks hold identical parameters, equal to the single-process emulation.

(A 2-rank batch-1 step is NOT a 1-rank batch-2 step - BatchNorm statistics and the loss normalisers are per rank, in the
reference too - so the emulation evaluates the two images separately.)  The child processes are started by
tests/conftest.py before this process touches the GPU (tests/dp/launcher.py) and only read back here."""
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_rank_data_parallel_equals_sum_of_single_image_steps():
    out = os.environ.get('MRCNN_DP_TEST_DIR')
    if not out:
        pytest.skip('the data-parallel workers are started by tests/conftest.py when pytest runs with -m gpu')
    t0 = time.time()
    while not os.path.exists(os.path.join(out, 'done')):
        assert time.time() - t0 < 900, 'data-parallel workers did not finish'
        time.sleep(1.0)
    log = open(os.path.join(out, 'log.txt')).read()
    assert open(os.path.join(out, 'done')).read() == '0', log[-4000:]
    r0, r1, emu = (torch.load(os.path.join(out, f)) for f in ('dp_rank0.pt', 'dp_rank1.pt', 'emu.pt'))
    assert not torch.equal(r0['p_init'], r1['p_init'])          # the ranks really started from different replicas ...
    assert torch.equal(r0['p0'], r1['p0']) and torch.equal(r0['p0'], r0['p_init'])      # ... the broadcast made them rank 0's
    assert torch.equal(r0['p0'], emu['p0'])                     # ... which is the emulation's
    assert abs(r0['loss0'] - emu['losses'][0]) == 0 and abs(r1['loss0'] - emu['losses'][1]) == 0
    assert torch.equal(r0['grads'], r1['grads'])
    assert torch.equal(r0['grads'], emu['grads'])               # SUM of the two ranks' gradients, no scaling
    assert float(emu['grads'].abs().max()) > 0
    assert torch.equal(r0['params'], r1['params'])
    assert torch.equal(r0['params'], emu['params'])


def test_rccl_path_single_rank():
    """The RCCL branch of the data-parallel code, before an 8-GPU node meets it: a child process creates
    a ONE-rank process group on backend 'nccl' through optimizers.init_process_group (high-priority collective streams) and
    drives GradientSynchronizer on the DEVICE gradient buffer - rank-0 broadcast, > 3 buckets each all-reduced by RCCL on
    the side stream as backward passes their offsets, timing_report(), finish() - for 3 steps; a SUM over one rank is the
    identity, so gradients, parameters and loss must equal the un-synchronised run bit for bit."""
    out = os.environ.get('MRCNN_DP_TEST_DIR')
    if not out:
        pytest.skip('started by tests/conftest.py when pytest runs with -m gpu')
    t0 = time.time()
    while not os.path.exists(os.path.join(out, 'rccl1_done')):
        assert time.time() - t0 < 1200, 'the RCCL worker did not finish'
        time.sleep(1.0)
    log = open(os.path.join(out, 'log.txt')).read()
    assert open(os.path.join(out, 'rccl1_done')).read() == '0', log[-4000:]
    r = torch.load(os.path.join(out, 'rccl1.pt'))
    assert torch.equal(r['g0'], r['g1']) and float(r['g1'].abs().max()) > 0
    assert torch.equal(r['p0'], r['p1']) and r['l0'] == r['l1']
    rep = r['report']
    assert rep is not None and len(rep['bucket_ms']) > 3 and all(v >= 0 for v in rep['bucket_ms'])
    assert rep['allreduce_ms_total'] > 0 and 0.0 <= rep['overlap_fraction'] <= 1.0


def test_bench_two_ranks_as_the_driver_launches_it():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2
    --steps K --warmup W` (the driver's command for N > 1): rank 0 prints ONE JSON line for the whole job - n_gpus 2, weak
    scaling, the data-parallel all-reduce report - with both ranks on the one GPU of the test box (gloo in place of RCCL)."""
    import json
    out = os.environ.get('MRCNN_DP_TEST_DIR')
    if not out:
        pytest.skip('started by tests/conftest.py when pytest runs with -m gpu')
    t0 = time.time()
    while not os.path.exists(os.path.join(out, 'bench_done')):
        assert time.time() - t0 < 1500, 'bench.py --gpus 2 did not finish'
        time.sleep(1.0)
    log = open(os.path.join(out, 'bench_log.txt')).read()
    assert open(os.path.join(out, 'bench_done')).read() == '0', log[-4000:]
    lines = [l for l in log.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, log[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['value'] > 0
    assert d['config']['global_batch'] == 4 and 'allreduce_rank0' in d['config']
    pr = d['config']['ms_per_step_per_rank']            # every rank's own clock beside the job number (a straggler is visible)
    assert len(pr['all']) == 2 and pr['min'] <= pr['max'] and abs(pr['max'] - d['ms_per_step']) <= 1e-2 * d['ms_per_step']
    assert d['roofline']['traffic_source']['file'].startswith('profiles/')
    assert abs(d['value'] - 4 * 1e3 / d['ms_per_step']) <= 1e-2 * d['value']          # whole-job images/s = global batch / step time
    assert d['roofline']['frac'] <= 1.0 and 'roi_align_microbench' in d
    # the ruling's fallback number is in the N > 1 line too: the all-float32-MFMA step of the same processes, whole-job rate
    assert d['config']['gemm_arithmetic']['name'] == 'bf16x6_behind_backbone' and d['config']['images_per_sec_f32_mfma'] > 0
    # the line says what the collective library saw: ranks, their devices, backend, library version
    rc = d['config']['rccl']
    assert rc['world_size'] == 2 and rc['backend'] == 'gloo' and len(rc['ranks']) == 2
    assert sorted(r['rank'] for r in rc['ranks']) == [0, 1] and all(r['device_index'] == 0 and r['pci_bus_id'] for r in rc['ranks'])
    assert rc['distinct_devices'] == 1          # both ranks on the one GPU: accepted only because MRCNN_BENCH_SINGLE_DEVICE=1
