"""MFMA-busy fraction per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass over tools/step_pmc_run.py:
busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs); duration-weighted over the launches of a kernel.
usage: pmc_mfma_summary.py <dir> <steps> > profiles/r02_conv_pmc_mfma.json"""
import csv, glob, json, re, sys, collections
f = sorted(glob.glob(sys.argv[1] + '/*/*_counter_collection.csv'))[-1]
steps = float(sys.argv[2])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '')
    acc[n][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        cnt[n] += 1
out = {'source': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (one pass, no trace domains) over `python3 tools/step_pmc_run.py %d` '
                 '(configs[2] training steps, nothing else), MI355X, round 2; mfma_busy_fraction = sum SQ_VALU_MFMA_BUSY_CYCLES / '
                 '(sum GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) over all launches of the kernel (kernels are serialised in counter mode)' % int(steps),
       'kernels': {}}
tot_b = tot_a = 0.0
for n, c in sorted(acc.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0)):
    a, b = c.get('GRBM_GUI_ACTIVE', 0.0), c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    if 'k_conv_igemm' not in n and 'k_pgemm' not in n:
        continue
    out['kernels'][n] = {'launches_per_step': round(cnt[n] / steps, 1), 'mfma_busy_fraction': round(b / (a / 8 * 1024), 4) if a else None,
                         'gpu_cycles_per_step': round(a / 8 / steps)}
    tot_b += b; tot_a += a
out['all_k_conv_igemm'] = {'mfma_busy_fraction': round(tot_b / (tot_a / 8 * 1024), 4), 'gpu_cycles_per_step': round(tot_a / 8 / steps)}
print(json.dumps(out, indent=1))
