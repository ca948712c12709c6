#!/usr/bin/env python3
"""Top rows of a rocprofv3 *kernel_stats.csv: name (short), calls, average / min / max us, share."""
import csv, sys
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows) or 1.0
    print(f, 'total ms', round(tot / 1e6, 3))
    for r in rows[:12]:
        name = r["Name"].replace("void ", "").replace("ure::", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
        print(f"  {name:44s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:9.2f}  min {float(r['MinNs'])/1e3:9.2f}  max {float(r['MaxNs'])/1e3:9.2f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
