"""train.py on the device (checkpoint / resume; train.py:134-161 snapshot + log): a run interrupted at
a snapshot and resumed from trainer_<it>.pt ends with bitwise the same NPZ snapshot as the uninterrupted run (parameters,
momentum, BatchNorm statistics, sampler seeds and the data position are all part of the trainer state), and the log
carries the reference's keys."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import train  # noqa: E402


def _args(out, iteration, resume=''):
    return train.build_parser().parse_args(['--out', out, '--iteration', str(iteration), '--batch-size', '1', '--image-size', '256', '320',
                                            '--log-interval', '2', '--snapshot-interval', '2', '--lr-shift-interval', '3',
                                            '--label_file', '/nonexistent'] + (['--resume', resume] if resume else []))


def test_resume_is_bit_identical(tmp_path):
    a, b = str(tmp_path / 'a'), str(tmp_path / 'b')
    train.run(_args(a, 4))
    assert os.path.exists(os.path.join(a, 'trainer_2.pt')) and os.path.exists(os.path.join(a, 'model_4.npz'))
    train.run(_args(b, 4, resume=os.path.join(a, 'trainer_2.pt')))
    za, zb = np.load(os.path.join(a, 'model_4.npz')), np.load(os.path.join(b, 'model_4.npz'))
    assert sorted(za.files) == sorted(zb.files) and len(za.files) > 100
    for k in za.files:
        np.testing.assert_array_equal(za[k], zb[k], err_msg=k)
    assert not np.array_equal(za['head/fc2/W'], np.load(os.path.join(a, 'model_2.npz'))['head/fc2/W'])
    log = [json.loads(l) for l in open(os.path.join(a, 'log'))]
    assert [e['iteration'] for e in log] == [2, 4]
    for key in ('main/loss', 'main/rpn_loc_loss', 'main/rpn_cls_loss', 'main/roi_loc_loss', 'main/roi_cls_loss', 'main/mask_loss', 'lr'):
        assert key in log[0]
    assert log[1]['lr'] == pytest.approx(1e-4)          # ExponentialShift('lr', 0.1) after iteration 3
    lb = [json.loads(l) for l in open(os.path.join(b, 'log'))]
    assert lb[-1]['iteration'] == 4 and lb[-1]['main/loss'] == log[1]['main/loss'] and lb[-1]['lr'] == log[1]['lr']
