#!/bin/bash
# bash tools/r4_index_ab.sh OUTDIR lib1.so lib2.so ...   (product library first)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r4b}; shift
mkdir -p "$OUT"
timeout -k 10 300 python3 "$ROOT/tools/exp_index.py" >> "$OUT/index_ab.jsonl" 2> "$OUT/index_ab.err" || exit 1
for lib in "$@"; do
  URE_LIB="$ROOT/$lib" URE_ALLOW_STALE_LIB=1 timeout -k 10 300 python3 "$ROOT/tools/exp_index.py" --label "$lib" >> "$OUT/index_ab.jsonl" 2>> "$OUT/index_ab.err" || exit 1
done
cat "$OUT/index_ab.jsonl"
