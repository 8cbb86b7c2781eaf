"""The two GPU test tiers (tests/conftest.py): `-m gpu` must not collect a `gpu_long` test, `-m gpu_long` collects exactly those, `-m "not gpu"`
neither; and `-m gpu` runs the kernel-vs-oracle files before the float64-oracle files (a time limit can only ever cut the latter)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _collect(expr):
    out = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests'), '--collect-only', '-q', '-m', expr, '-p', 'no:cacheprovider'],
                         capture_output=True, text=True, cwd=ROOT, timeout=300).stdout
    return [l.strip() for l in out.splitlines() if '::' in l]


def test_gpu_long_is_its_own_tier():
    gpu, long_, cpu = _collect('gpu'), _collect('gpu_long'), _collect('not gpu')
    assert len(gpu) > 350 and len(long_) > 30 and len(cpu) > 60
    assert not set(gpu) & set(long_) and not set(cpu) & set(long_) and not set(cpu) & set(gpu)
    assert any('other_batches' in n for n in long_) and not any('other_batches' in n for n in gpu)
    # what the driver's run must contain: the benchmarked batch in the shipped arithmetic, the keypoint model at its shape, configs[0], configs[1]
    for must in ('test_full_width_1024_batch2_shipped_arithmetic', 'test_full_width_keypoint_1024_batch2_shipped', 'test_call_800x800_train_mode_matches_oracle',
                 'test_config2_all_512_rois_against_the_oracle', 'test_planned_backward_equals_the_fused_backward_config2', 'test_composite_bottleneck_calls_give_the_same_bits'):
        assert any(must in n for n in gpu), must
    files = []
    for n in gpu:
        f = n.split('::')[0]
        if not files or files[-1] != f:
            files.append(f)
    assert len(files) == len(set(files)), 'a file is visited twice'
    pos = {os.path.basename(f): i for i, f in enumerate(files)}
    assert pos['test_roi_align_gpu.py'] < pos['test_rpn_gpu.py'] < pos['test_targets_gpu.py'] < pos['test_step_gpu.py'] < pos['test_full_width_gpu.py'] < pos['test_dp_gpu.py']
