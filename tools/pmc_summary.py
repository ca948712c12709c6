#!/usr/bin/env python3
"""Per-kernel means of the counters collected by tools/pmc_passes.sh -> JSON on stdout."""
import csv, glob, json, os, sys
from collections import defaultdict

out = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
for path in sorted(glob.glob(os.path.join(sys.argv[1], '*', '**', '*counter_collection.csv'), recursive=True)):
    seen = set()
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row['Kernel_Name']
            if 'ure::' not in k:
                continue
            k = k.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
            c = out[k][row['Counter_Name']]
            c[0] += float(row['Counter_Value']); c[1] += 1
            if (path, row['Dispatch_Id']) not in seen:
                seen.add((path, row['Dispatch_Id']))
                d = dur[k]
                d[0] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3; d[1] += 1
res = {}
for k, cs in out.items():
    res[k] = {c: {'mean': round(v[0] / v[1], 2), 'dispatches': v[1]} for c, v in cs.items()}
    res[k]['_mean_us_under_profiler'] = round(dur[k][0] / max(dur[k][1], 1), 2)
print(json.dumps(res, indent=1))
