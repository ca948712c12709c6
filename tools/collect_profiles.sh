#!/bin/bash
# Everything profiles/rNN/ holds, in one pass on the GPU box:  bash tools/collect_profiles.sh r02
# Writes gpurun_out/profiles_<tag>/ ; copy what should be judged into profiles/<tag>/.
# rocprofv3 runs the program itself after `--` (python3 <script>), kernel-trace / stats only; the PMC passes are separate
# runs (tools/pmc_traffic.py), never combined with another tracing domain.
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
stats() {   # name, script, args...
  local name=$1; shift
  rm -rf "$OUT/trace_$name"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -- python3 "$@" > "$OUT/$name.stdout" 2> "$OUT/$name.stderr"
  echo "$name rc=$?"
  f=$(find "$OUT/trace_$name" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv"
  rm -rf "$OUT/trace_$name"
}
python3 "$ROOT/bench.py" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"; echo "bench rc=$?"
python3 "$ROOT/bench.py" --workload ml25m --no-cpu-baseline > "$OUT/bench_ml25m.json" 2> "$OUT/bench_ml25m.err"; echo "bench ml25m rc=$?"
python3 "$ROOT/bench.py" --gpus 2 --backend gloo --force-device 0 > "$OUT/bench_2ranks_gloo_one_gpu.json" 2> "$OUT/bench_2ranks.err"; echo "bench 2 ranks rc=$?"
python3 "$ROOT/bench.py" --gpus 1 --force-dist --no-cpu-baseline > "$OUT/bench_1rank_rccl.json" 2> "$OUT/bench_1rank_rccl.err"; echo "bench rccl rc=$?"
stats bench "$ROOT/bench.py" --no-cpu-baseline
stats bench_ml25m "$ROOT/bench.py" --workload ml25m --no-cpu-baseline --no-unlearn --steps 5 --warmup 1
stats e2e_sisa "$ROOT/tools/e2e_sisa.py"
stats ot "$ROOT/tools/profile_ot.py" --rounds 4
python3 "$ROOT/tools/pmc_traffic.py" "$OUT/pmc" --tag "$TAG" > "$OUT/pmc_ml1m.log" 2>&1; echo "pmc ml1m rc=$?"
python3 "$ROOT/tools/pmc_traffic.py" "$OUT/pmc" --tag "$TAG" -- --workload ml25m > "$OUT/pmc_ml25m.log" 2>&1; echo "pmc ml25m rc=$?"
cp "$OUT"/pmc/*_pmc_hbm_traffic_*.json "$OUT"/ 2>/dev/null
rm -rf "$OUT/pmc"
python3 "$ROOT/tools/e2e_sisa.py" > "$OUT/e2e_sisa.json" 2>/dev/null
python3 "$ROOT/tools/profile_ot.py" --rounds 6 > "$OUT/ot_rounds.json" 2>/dev/null
python3 "$ROOT/tools/exp_touch.py" > "$OUT/touch_vs_default_ml25m.json" 2>/dev/null
ls -la "$OUT"
