#!/bin/bash
# configs[3]'s shape (32 shards, 27 steps per epoch) in touch_mode 2 against touch_mode 3 forced (URE_TOUCH_INDEX=2), d = 16 and d = 128
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/${1:-r4p}; mkdir -p $OUT
for d in ${DS:-16 128}; do for v in 1 2; do
  URE_TOUCH_INDEX=$v timeout -k 10 400 python3 $ROOT/bench.py --workload ml25m --shards 32 --d $d --no-cpu-baseline --no-unlearn --steps 3 --warmup 1 --roofline-steps 2 > $OUT/cfg3_d${d}_index$v.json 2> $OUT/err.txt || exit 1
  python3 - <<PY
import json
j=json.loads(open('$OUT/cfg3_d${d}_index$v.json').read().strip().splitlines()[-1]); r=j['roofline']
print('d', $d, 'URE_TOUCH_INDEX', $v, 'mode', r['touch_mode'], 'value', round(j['value']/1e9,3), 'G/s', 'ms_per_step', j['ms_per_step'], 'avg_launch_us', r['avg_launch_us'], 'frac', r['frac'])
PY
done; done
