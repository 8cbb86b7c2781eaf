"""Starts the child processes of tests/test_dp_gpu.py and waits for them.  Imports no torch / HIP itself and is started
by tests/conftest.py BEFORE the pytest process has touched the GPU (children are started from GPU-free processes only)."""
import os
import socket
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def main(out):
    w = os.path.join(HERE, 'worker.py')
    log = open(os.path.join(out, 'log.txt'), 'w')
    # tests/conftest.py writes `go` when the first test of the oracle-heavy files (or of test_dp_gpu.py) is set up: until then the GPU
    # belongs to the kernel tests alone
    while not os.path.exists(os.path.join(out, 'go')):
        if not os.path.isdir(out):
            return
        time.sleep(0.5)
    rc = subprocess.call([sys.executable, w, 'emu', out], stdout=log, stderr=subprocess.STDOUT)
    for attempt in range(3):       # the rendezvous port is picked and released here, re-bound by rank 0: retried on a collision
        port = _free_port()
        procs = [subprocess.Popen([sys.executable, w, 'dp', str(r), '2', str(port), out], stdout=log, stderr=subprocess.STDOUT)
                 for r in range(2)]
        rcd = 0
        for p in procs:
            try:
                rcd |= p.wait(timeout=600)
            except subprocess.TimeoutExpired:
                p.kill()
                rcd |= 1
        if rcd == 0:
            break
    rc |= rcd
    open(os.path.join(out, 'done'), 'w').write(str(rc))
    # ---- one rank on RCCL: the collective code path of an N-GPU job (tests/test_dp_gpu.py::test_rccl_path_single_rank)
    try:
        rr = subprocess.call([sys.executable, w, 'rccl1', str(_free_port()), out], stdout=log, stderr=subprocess.STDOUT, timeout=600)
    except subprocess.TimeoutExpired:
        rr = 1
    log.flush()
    open(os.path.join(out, 'rccl1_done'), 'w').write(str(rr))
    # ---- bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per "GPU"): both ranks on the one
    #      device of the test box, gloo in place of RCCL (MRCNN_BENCH_SINGLE_DEVICE / MRCNN_BENCH_BACKEND)
    root = os.path.dirname(os.path.dirname(HERE))
    env = dict(os.environ, MRCNN_BENCH_SINGLE_DEVICE='1', MRCNN_BENCH_BACKEND='gloo')
    blog = open(os.path.join(out, 'bench_log.txt'), 'w')
    brc = 1
    for attempt in range(3):
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1']
        try:
            brc = subprocess.call(cmd, stdout=blog, stderr=subprocess.STDOUT, env=env, cwd=root, timeout=600)
        except subprocess.TimeoutExpired:
            brc = 1
        if brc == 0:
            break
    open(os.path.join(out, 'bench_done'), 'w').write(str(brc))


if __name__ == '__main__':
    main(sys.argv[1])
