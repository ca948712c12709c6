"""Prints a rocprofv3 kernel_stats.csv: python tools/kstats.py FILE [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{r['Name'][:64]:64s} calls {r['Calls']:>6s} total_us {float(r['TotalDurationNs'])/1e3:>10.1f} avg_us {float(r['AverageNs'])/1e3:>9.2f} pct {float(r['Percentage']):>6.2f} min {float(r['MinNs'])/1e3:>8.1f} max {float(r['MaxNs'])/1e3:>8.1f}")
