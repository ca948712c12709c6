#!/bin/bash
# kernel-trace stats of full MF at the 25 M shape in touch_mode 3: bash tools/r4_index_prof.sh OUTDIR
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r4g}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 "$ROOT/tools/exp_index.py" > "$OUT/index_plain.jsonl" 2> "$OUT/index_plain.err" || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python3 "$ROOT/tools/exp_index.py" > "$OUT/index_prof.jsonl" 2> "$OUT/index_prof.err"; echo "prof rc=$?"
f=$(find "$OUT/prof" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/index_kernel_stats.csv"
rm -rf "$OUT/prof"
cat "$OUT/index_plain.jsonl"
