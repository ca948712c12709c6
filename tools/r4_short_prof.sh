#!/bin/bash
# kernel-trace stats of configs[3]'s shape at d = 16 in touch_mode 3 (short epochs): bash tools/r4_short_prof.sh OUTDIR   (URE_INDEX_STAGED=0/1)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r4sp}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$ROOT/bench.py" --workload ml25m --shards 32 --d ${D:-16} --no-cpu-baseline --no-unlearn --steps 3 --warmup 1 --roofline-steps 2 > "$OUT/bench.json" 2> "$OUT/err.txt"; echo "prof rc=$?"
f=$(find "$OUT/prof" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/kernel_stats.csv"
rm -rf "$OUT/prof"
python3 "$ROOT/tools/kstats.py" "$OUT/kernel_stats.csv" 18
