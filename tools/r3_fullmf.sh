#!/bin/bash
# Full MF at the 25 M shape (config.py:182-188 runFull: ONE shard, 22.5 M train rows, 750 optimizer steps per epoch), d = 128:
# the dense default kernel (URE_TOUCH=0: what round 2 ran, touch mode refused more than 64 steps per epoch) against touch mode
# in 64-step windows.  bash tools/r3_fullmf.sh OUTDIR
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r3h}
mkdir -p "$OUT"
ARGS="--workload ml25m --shards 1 --d 128 --no-cpu-baseline --no-unlearn --steps 1 --warmup 1 --roofline-steps 1"
URE_TOUCH=0 timeout -k 10 500 python3 "$ROOT/bench.py" $ARGS > "$OUT/fullmf25m_d128_dense.json" 2> "$OUT/fullmf_dense.err"; echo "dense rc=$?"
timeout -k 10 500 python3 "$ROOT/bench.py" $ARGS > "$OUT/fullmf25m_d128_touch.json" 2> "$OUT/fullmf_touch.err"; echo "touch rc=$?"
