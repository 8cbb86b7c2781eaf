"""The instrument of DESIGN 5.9 (tests/tools/noise_floor_study.py) on made-up errors: the leave-one-out floor, statistics that one tensor
does not decide, and the admission rule."""
import random

from tests.tools.noise_floor_stats import admissible, stats


def _realisation(rs, names, scale=1.0):
    return {n: scale * 2e-3 * (0.5 + rs.random()) for n in names}


def test_a_realisation_of_the_same_noise_sits_at_the_floor_of_the_others():
    rs = random.Random(1)
    names = ['t%d' % i for i in range(200)]
    reals = [_realisation(rs, names) for _ in range(7)]
    for i, r in enumerate(reals):
        s = stats(r, [o for j, o in enumerate(reals) if j != i])
        assert 0.5 < s['median'] <= 1.0 and s['p90'] <= 1.5 and s['n3'] == 0 and s['n6'] == 0


def test_a_noisier_arithmetic_shows_in_the_quantiles_and_one_outlier_tensor_does_not():
    rs = random.Random(2)
    names = ['t%d' % i for i in range(200)]
    reals = [_realisation(rs, names) for _ in range(6)]
    noisy = _realisation(rs, names, scale=2.0)
    s = stats(noisy, reals)
    assert s['median'] > 1.2 and s['p90'] > 1.5
    one = _realisation(rs, names)
    one['t7'] = 1.0                      # one flipped near-tie: a single tensor far out
    s1 = stats(one, reals)
    assert s1['worst'] > 100 and s1['n3'] == 1 and s1['median'] <= 1.0 and s1['p90'] <= 1.5
    # tensors below 1e-3 count for the shares (as "not above") but not for the ratios
    tiny = {n: 1e-5 for n in names}
    st = stats(tiny, reals)
    assert st['n3'] == 0 and st['median'] == 0.0


def test_admission_rule_compares_worst_and_mean_at_the_printed_resolution():
    null = {'median': (0.95, 0.92), 'p90': (1.03, 1.02), 'share3': (0.0, 0.0)}
    assert admissible({'median': (0.9312, 0.90), 'p90': (1.034, 1.0), 'share3': (0.0004, 0.0)}, null)      # 1.034 prints as 1.03, 0.04 % as 0.0 %
    assert not admissible({'median': (0.94, 0.90), 'p90': (1.78, 1.24), 'share3': (0.04, 0.008)}, null)
    assert not admissible({'median': (0.94, 0.93), 'p90': (1.00, 1.0), 'share3': (0.0, 0.0)}, null)         # the mean over the batches counts too


def test_the_committed_study_gives_its_verdict_again():
    """profiles/r05_noise_floor_study.json holds the per-tensor errors of every realisation on the seven batches: the verdict of DESIGN 5.9
    recomputed from them - the shipped arithmetic within the float32 realisations' own range, the emulation in the backbone's forward pass not."""
    import json, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles', 'r05_noise_floor_study.json')
    d = json.load(open(path))
    seeds = [str(s) for s in d['seeds']]
    assert len(seeds) >= 7
    gpu32 = ['gpu:shipped', 'gpu:direct', 'gpu:direct/plan', 'gpu:uniform_f2', 'gpu:winograd_f2']
    f32 = gpu32 + ['cpu:a', 'cpu:b']
    strict = ['gpu:direct', 'gpu:direct/plan', 'cpu:a', 'cpu:b']
    cands = [l for l in d['errs'][seeds[0]] if l not in f32]
    verdict = {}
    for floor_set in (f32, strict):
        per = {l: {s: stats(d['errs'][s][l], [d['errs'][s][o] for o in floor_set if o != l]) for s in seeds} for l in gpu32 + cands}
        agg = lambda l: {k: (max(per[l][s][k] for s in seeds), sum(per[l][s][k] for s in seeds) / len(seeds)) for k in ('median', 'p90', 'share3')}
        null = {k: (max(agg(l)[k][0] for l in gpu32), max(agg(l)[k][1] for l in gpu32)) for k in ('median', 'p90', 'share3')}
        for l in cands:
            verdict.setdefault(l, []).append(admissible(agg(l), null))
        # the float32 GPU realisations never put a tensor above 3 x the floor of the others, on any batch
        assert all(per[l][s]['n3'] == 0 for l in gpu32 for s in seeds)
    ok = {l: all(v) for l, v in verdict.items()}
    assert ok['bf16x6 behind the backbone (shipped)'] and ok['bf16x6 backward passes only']
    assert not ok['bf16x6 everywhere'] and not ok['bf16x6 backbone forward only']
