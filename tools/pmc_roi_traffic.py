"""ROIAlign HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/roi_pmc_run.py:
pmc_roi_traffic.py <fetch dir> <write dir> -> JSON (KB counters x 1024; FETCH_SIZE doubled per MI355X_MICROARCH.md)."""
import csv, glob, json, re, sys, collections


def per_kernel(d, counter):
    f = sorted(glob.glob(d + '/*/*_counter_collection.csv'))[-1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '').split('<')[0]
        acc[n].append(float(r['Counter_Value']))
    return {n: sum(v) / len(v) for n, v in acc.items()}


fe, wr = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace domains) on `python3 tools/roi_pmc_run.py 5` (forward + the shipped fused backward) '
                 '(configs[1]: 512 RoIs, x = (1,256,200,272) NHWC, 7x7, sampling 2), MI355X; mean per launch; KB counters x 1024; FETCH_SIZE '
                 'doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for wide coalesced reads)'}
for n in sorted(fe):
    if 'roi' not in n:
        continue
    out[n] = {'FETCH_SIZE_KB': round(fe[n], 1), 'WRITE_SIZE_KB': round(wr.get(n, 0.0), 1),
              'hbm_bytes': int(fe[n] * 2 * 1024 + wr.get(n, 0.0) * 1024)}
print(json.dumps(out, indent=1))
