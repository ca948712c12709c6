#!/bin/bash
# Full MF at the 25 M shape (config.py:182-188 runFull: ONE shard, 22.5 M train rows, 750 optimizer steps per epoch), d = 128:
# touch_mode 3 (the epoch's slots sorted by step, csrc/mf_index.h) against touch mode in 64-step windows (URE_TOUCH_INDEX=0: round 3).
# bash tools/r4_fullmf.sh OUTDIR [prof]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r4a}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--workload ml25m --shards 1 --d 128 --no-cpu-baseline --no-unlearn --steps 1 --warmup 1 --roofline-steps 1"
timeout -k 10 500 python3 "$ROOT/bench.py" $ARGS > "$OUT/fullmf25m_d128_index.json" 2> "$OUT/fullmf_index.err"; echo "index rc=$?"
if [ "$2" = "prof" ]; then
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_index" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/fullmf25m_d128_index_prof.json" 2> "$OUT/fullmf_index_prof.err"; echo "prof rc=$?"
  find "$OUT/prof_index" -name '*kernel_stats.csv' -exec cp {} "$OUT/fullmf_index_kernel_stats.csv" \;
  rm -rf "$OUT/prof_index"
fi
if [ "$2" != "noref" ] && [ "$3" != "noref" ]; then
  URE_TOUCH_INDEX=0 timeout -k 10 500 python3 "$ROOT/bench.py" $ARGS > "$OUT/fullmf25m_d128_windows.json" 2> "$OUT/fullmf_windows.err"; echo "windows rc=$?"
fi
