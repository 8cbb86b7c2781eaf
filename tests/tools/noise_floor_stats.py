"""The statistics of tests/tools/noise_floor_study.py (pure Python: tested on the CPU by tests/test_noise_floor_stats_cpu.py)."""


def stats(e, others):
    """e: tensor name -> error of the judged realisation; others: list of tensor name -> error of the floor set (the judged one excluded).
    floor(t) = the largest error any of `others` makes on t.  Returns the median and the 90th percentile of e / floor over the tensors with
    e >= 1e-3, the count / share of ALL tensors with e >= max(1e-3, 3 x floor), the count above 6 x floor, and the largest ratio."""
    names = list(e)
    floor = {n: max(o[n] for o in others) for n in names}
    ratios = sorted(e[n] / max(floor[n], 1e-12) for n in names if e[n] >= 1e-3)
    if not ratios:
        ratios = [0.0]
    above3 = sum(1 for n in names if not e[n] < max(1e-3, 3 * floor[n]))
    above6 = sum(1 for n in names if not e[n] < max(1e-3, 6 * floor[n]))
    return dict(median=ratios[len(ratios) // 2], p90=ratios[int(0.9 * (len(ratios) - 1))], share3=above3 / len(names), n3=above3, n6=above6,
                worst=ratios[-1])


def admissible(row, null):
    """row / null: statistic -> (worst over the batches, mean over the batches) of a candidate / of the float32 GPU realisations' largest.
    Compared at the printed resolution (two decimals; 0.1 % for the share)."""
    rnd = lambda k, v: round(v, 3 if k == 'share3' else 2)
    return all(rnd(k, row[k][0]) <= rnd(k, null[k][0]) and rnd(k, row[k][1]) <= rnd(k, null[k][1]) for k in row)
