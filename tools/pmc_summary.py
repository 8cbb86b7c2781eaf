"""Summarise a rocprofv3 --pmc counter_collection csv per kernel (mean of each counter over launches)."""
import csv, glob, json, re, sys, collections
f = sorted(glob.glob(sys.argv[1] + '/*/*_counter_collection.csv'))[-1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']).split('(')[0].replace('void ', '')
    acc[n][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for n, cs in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in n:
        continue
    out[n] = {'launches': len(next(iter(cs.values())))}
    for c, v in cs.items():
        out[n][c] = sum(v) / len(v)
print(json.dumps(out, indent=1))
