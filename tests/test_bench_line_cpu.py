"""Host logic of the bench line that does not need a GPU (bench_step.py): the hardware-counter fields are CONSTANTS read from the summaries
committed under profiles/ (collected by tools/round_end.sh with rocprofv3 --pmc in separate passes) - the line must say which file and
which commit, pick the newest round, and survive a summary that lacks a key (a KeyError after all measurements are done would
lose the whole line)."""
import json
import os

from chainer_maskrcnn import bench_step as bs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_counters_come_from_the_newest_committed_summary():
    f, d = bs.latest_profile('step_pmc_traffic.json')
    rounds = sorted(n for n in os.listdir(os.path.join(ROOT, 'profiles')) if n.endswith('_step_pmc_traffic.json'))
    assert f == 'profiles/' + rounds[-1] and d is not None and 'conv_bracket' in d
    r = bs._pmc_step_counters()
    assert r['traffic'] == d['conv_bracket']['hbm_bytes_per_step'] > 0
    src = r['traffic_source']
    assert src['file'] == f and src['collected_at_commit'] and 'not collected by this run' in src['kind'] and 'FETCH_SIZE' in src['method']
    fam = r['hbm_bytes_per_step_by_family']
    assert {'conv_gemm', 'winograd_transforms', 'slab_tail_column_sums', 'batchnorm', 'roi_align'} <= set(fam)
    assert abs(r['hbm_bytes_per_step_whole_step'] - sum(fam.values())) < 1.0
    # MFMA-busy fractions of the GEMM kernels (plane GEMMs and every k_conv_igemm instantiation), each a fraction
    busy = r['pmc_mfma_busy_fraction_by_kernel']
    assert any(k.startswith('k_pgemm') for k in busy) and any(k.startswith('k_conv_igemm') for k in busy)
    assert all(v is None or 0.0 < v <= 1.0 for v in busy.values())


def test_a_summary_without_the_expected_keys_gives_none_fields_not_an_exception(monkeypatch):
    def broken(name):
        return 'profiles/r99_' + name, ({'conv_bracket': {}} if name.startswith('step') else {'kernels': {'k_pgemm_pp': {}}})
    monkeypatch.setattr(bs, 'latest_profile', broken)
    r = bs._pmc_step_counters()
    assert r['traffic'] is None and 'error' in r['traffic_source']
    assert r['pmc_mfma_busy_fraction_by_kernel'] == {'k_pgemm_pp': None}
    monkeypatch.setattr(bs, 'latest_profile', lambda name: (None, None))          # no summary at all
    r = bs._pmc_step_counters()
    assert r['traffic'] is None and r['traffic_source'] is None


def test_committed_bench_line_of_the_round_has_the_contract_fields():
    """profiles/rNN_bench_step_n1.json is the line bench.py printed on the GPU box: the driver's contract keys, the ruling's conditions
    (dtype f32, config.gemm_arithmetic, config.images_per_sec_f32_mfma) and the two extra objects."""
    f, d = bs.latest_profile('bench_step_n1.json')
    assert d is not None, 'no committed bench line'
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
              'roofline', 'cpu_baseline'):
        assert k in d, (f, k)
    assert d['dtype'] == 'f32' and d['n_gpus'] == 1 and d['scaling'] == 'weak' and d['vs_baseline'] is None and d['data'] == 'synthetic'
    assert abs(d['value'] - 2 * 1e3 / d['ms_per_step']) <= 1e-2 * d['value']
    ga = d['config']['gemm_arithmetic']
    assert ga['name'] in ('f32', 'bf16x6_behind_backbone', 'bf16x6_backward', 'bf16x6') and 'v_mfma' in ga['scheme']
    assert 0 < d['config']['images_per_sec_f32_mfma'] <= 1.02 * d['value'] or ga['name'] == 'f32'
    r = d['roofline']
    assert r['bound'] == 'mfma' and r['unit'] == 'TFLOP/s' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3 and 0 < r['frac'] <= 1
    assert r['peak'] == (2500.0 if 'bf16' in r['pipe'] else 157.3)          # priced on the pipe the headline GEMMs run on
    assert 'effective_fp32_TFLOPs' in r and r['traffic'] > 0
    cb = d['cpu_baseline']
    assert cb['kind'] in ('port', 'reference') and cb['cores'] >= 1 and cb['value'] > 0 and cb['sample']
    json.dumps(d)
