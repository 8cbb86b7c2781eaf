import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'chainer-maskrcnn_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


_DP = {'proc': None, 'dir': None}


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'gpu_long: sweeps over modes / seeds / geometries on the GPU box; NOT part of `-m gpu` '
                            '(they carry the gpu marker too and are deselected below) - run them with `-m gpu_long`')
    _start_dp_workers(config)


# `-m gpu` runs the files in this order: the cheap index- / bit-exact kernel-vs-oracle and kernel-vs-golden files first, the files whose
# time is a float64 CPU oracle of whole images last, the data-parallel children's results at the very end - a time limit can then only
# ever cut the sweeps, never an index-exact row.  Files not listed keep their alphabetical place after the listed ones.
GPU_FILE_ORDER = ['test_roi_align_gpu.py', 'test_rpn_gpu.py', 'test_targets_gpu.py', 'test_step_reference_gpu.py', 'test_loss_gpu.py',
                  'test_predict_reference_gpu.py', 'test_legacy_reference_gpu.py', 'test_predict_gpu.py', 'test_nn_gpu.py',
                  'test_dataset_gpu.py', 'test_chainer_npz_gpu.py', 'test_conv_gpu.py', 'test_split_gemm_gpu.py', 'test_step_gpu.py',
                  'test_legacy_gpu.py', 'test_train_gpu.py', 'test_config0_gpu.py', 'test_full_width_gpu.py', 'test_dp_gpu.py']
# the data-parallel children (tests/dp/launcher.py) start their GPU work when the first test of one of these files is set up: they
# then share the GPU with tests whose time is spent in the CPU oracle, not with the kernel tests before them
DP_GO_FILES = ('test_config0_gpu.py', 'test_full_width_gpu.py', 'test_dp_gpu.py')


def _wants_long(config):
    return 'gpu_long' in (config.getoption('markexpr', '') or '')


def pytest_collection_modifyitems(config, items):
    if not _wants_long(config):
        long_ones = [it for it in items if it.get_closest_marker('gpu_long') is not None]
        if long_ones:
            config.hook.pytest_deselected(items=long_ones)
            items[:] = [it for it in items if it.get_closest_marker('gpu_long') is None]
    rank = {f: i for i, f in enumerate(GPU_FILE_ORDER)}
    items.sort(key=lambda it: rank.get(os.path.basename(str(it.fspath)), rank['test_config0_gpu.py'] - 0.5 if str(it.fspath).endswith('_gpu.py') else -1))


def pytest_runtest_setup(item):
    out = _DP['dir']
    if out is not None and os.path.basename(str(item.fspath)) in DP_GO_FILES:
        go = os.path.join(out, 'go')
        if not os.path.exists(go):
            open(go, 'w').close()


def _selects_dp_tests(config):
    """True when this session can run tests/test_dp_gpu.py: -m gpu, and the file is among (or below) the given paths."""
    expr = config.getoption('markexpr', '') or ''
    if 'gpu' not in expr or 'not gpu' in expr or 'gpu_long' in expr:
        return False
    target = os.path.join(ROOT, 'tests', 'test_dp_gpu.py')
    args = [a for a in (config.args or []) if not a.startswith('-')] or [os.path.join(ROOT, 'tests')]
    for a in args:
        path = os.path.abspath(os.path.join(str(config.invocation_params.dir), a.split('::')[0]))
        if path == target or (os.path.isdir(path) and target.startswith(path.rstrip(os.sep) + os.sep)):
            return True
    return False


def _start_dp_workers(config):
    """tests/test_dp_gpu.py compares child processes (2 data-parallel ranks, a single-process emulation, a one-rank RCCL
    run, bench.py under torch.distributed.run).  They are started HERE - before any test module is imported, i.e. before
    this process has initialised the GPU (a process that has must not fork + exec on the GPU pool) - by a launcher that
    itself never touches the GPU; the tests only wait for the result files.  Only when that file is part of the session
    and the box has a GPU device node (checked without loading any GPU runtime); the launcher and its directory are
    cleaned up in pytest_unconfigure."""
    if os.environ.get('MRCNN_DP_TEST_DIR') or not _selects_dp_tests(config) or not os.path.exists('/dev/kfd'):
        return
    import subprocess
    import tempfile
    out = tempfile.mkdtemp(prefix='mrcnn_dp_')
    os.environ['MRCNN_DP_TEST_DIR'] = out
    _DP['dir'] = out
    _DP['proc'] = subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dp', 'launcher.py'), out], start_new_session=True)


def pytest_unconfigure(config):
    proc, out = _DP['proc'], _DP['dir']
    if proc is not None:
        if proc.poll() is None:             # an aborted session: end the launcher and its children (its own process group)
            import signal
            try:
                os.killpg(proc.pid, signal.SIGTERM)
            except OSError:
                pass
            try:
                proc.wait(timeout=20)
            except Exception:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
        _DP['proc'] = None
    if out is not None:
        import shutil
        shutil.rmtree(out, ignore_errors=True)
        os.environ.pop('MRCNN_DP_TEST_DIR', None)
        _DP['dir'] = None


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _library_defaults_after_each_test():
    """Process-wide library settings a test (or train.run / a train chain with gemm_arithmetic set) may have left behind are put back to
    the library defaults after every test: float32 GEMMs, no measurement knobs.  Only when the library is already loaded (CPU tests never
    load it)."""
    yield
    try:
        from chainer_maskrcnn import _hip
    except Exception:
        return
    if getattr(_hip, '_lib', None) is not None:
        _hip._lib.mrcnn_conv2d_set_split_operands(0, 0, 0)
        from chainer_maskrcnn.nn import core
        core.FWD_EMULATION_IN_BACKBONE = core.FWD_EMULATION_BEHIND_BACKBONE = True
        _hip._lib.mrcnn_debug_conv_parts(0)
        _hip._lib.mrcnn_conv2d_set_debug_skip(0)
