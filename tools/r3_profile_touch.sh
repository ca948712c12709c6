#!/bin/bash
# kernel-trace stats of the configs[3]-shape leg (touch mode) and of the honest e2e: bash tools/r3_profile_touch.sh OUTDIR
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r3d}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
stats() {
  local name=$1; shift
  rm -rf "$OUT/trace_$name"
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -- python3 "$@" > "$OUT/$name.stdout" 2> "$OUT/$name.stderr"
  echo "$name rc=$?"
  f=$(find "$OUT/trace_$name" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv"
  rm -rf "$OUT/trace_$name"
}
stats bench_ml25m "$ROOT/bench.py" --workload ml25m --no-cpu-baseline --no-unlearn --steps 5 --warmup 1
stats e2e_sisa "$ROOT/tools/e2e_sisa.py"
