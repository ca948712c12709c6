#!/bin/bash
# rocprof stats of the configs[3]-shape leg with launch C split into mark / advance (tools/ab/lib_touch_split.so)
# build it first:  python -m ultrare_amd.build --out tools/ab/lib_touch_split.so -DURE_TOUCH_SPLIT   (touch mode 1: URE_TOUCH_AHEAD=0)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r3f}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export URE_TOUCH_AHEAD=0
export URE_LIB=$ROOT/tools/ab/lib_touch_split.so
rm -rf "$OUT/trace_split"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_split" -- python3 "$ROOT/bench.py" --workload ml25m --no-cpu-baseline --no-unlearn --steps 5 --warmup 1 > "$OUT/split.stdout" 2> "$OUT/split.stderr"
f=$(find "$OUT/trace_split" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "$OUT/split_kernel_stats.csv"
rm -rf "$OUT/trace_split"
