import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'chainer-maskrcnn_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    _start_dp_workers(config)


def _start_dp_workers(config):
    """tests/test_dp_gpu.py compares child processes (2 data-parallel ranks + a single-process emulation).  They are
    started HERE - before any test module is imported, i.e. before this process has initialised the GPU - by a launcher
    that itself never touches the GPU; the test only waits for their result files."""
    expr = config.getoption('markexpr', '') or ''
    if 'gpu' not in expr or 'not gpu' in expr or os.environ.get('MRCNN_DP_TEST_DIR'):
        return
    try:
        import torch
        if torch.cuda.device_count() < 1:       # does not initialise the device
            return
    except Exception:
        return
    import subprocess
    import tempfile
    out = tempfile.mkdtemp(prefix='mrcnn_dp_')
    os.environ['MRCNN_DP_TEST_DIR'] = out
    subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dp', 'launcher.py'), out])


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
