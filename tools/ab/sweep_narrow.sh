#!/bin/bash
# per-launch time of the step kernel for every tools/ab/lib_*.so variant at d = 16 (configs[4] shape: 16 shards) and d = 8
# usage (GPU box): bash tools/ab/sweep_narrow.sh > gpurun_out/sweep_narrow.txt
for d in 16 8; do
  for f in tools/ab/lib_*.so; do
    r=$(URE_LIB=$PWD/$f timeout -k 10 200 python bench.py --shards 16 --d $d --steps 60 --warmup 10 --no-cpu-baseline --no-unlearn 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'], d['roofline']['per_launch_event_us'])") || exit 1
    echo "d=$d $(basename $f) $r"
  done
done
