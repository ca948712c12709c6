#!/bin/bash
# step-kernel time and algorithmic bandwidth over shard counts and table widths (ml-1m-shaped data)
for cfg in "1 32" "2 32" "5 16" "5 32" "5 64" "8 64" "16 16" "5 128" "5 256"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-unlearn --shards $1 --d $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('shards $1 d $2: %.2f G inter/s, %.2f us/launch, %.0f GB/s algorithmic (%.0f%% of 8 TB/s), %d interactions/launch' % (d['value']/1e9, r['avg_launch_us'], r['achieved'], 100*r['frac'], d['interactions_timed']//r['launches_timed']))" || exit 1
done
