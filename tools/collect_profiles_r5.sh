#!/bin/bash
# Everything profiles/r05/ holds, on the GPU box:  bash tools/collect_profiles_r5.sh [part ...]   (parts: bench stats pmc e2e fullmf k16 cfg4 big multirank shuffle; default all)
# Writes gpurun_out/profiles_r05/ ; copy what should be judged into profiles/r05/.
# rocprofv3 runs the program itself after `--` (python3 <script>), kernel-trace / stats only; the PMC passes are separate
# runs (tools/pmc_traffic.py), never combined with another tracing domain.
TAG=r05
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
PARTS=${*:-bench stats pmc e2e fullmf k16 cfg4 big multirank shuffle}
cd /tmp && export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
stats() {   # name, [VAR=VALUE ...] script, args...
  local name=$1; shift
  rm -rf "$OUT/trace_$name"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -- python3 "$@" > "$OUT/$name.stdout" 2> "$OUT/$name.stderr"
  echo "$name rc=$?"
  f=$(find "$OUT/trace_$name" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv"
  rm -rf "$OUT/trace_$name" "$OUT/$name.stdout" "$OUT/$name.stderr"
}
if has bench; then
  python3 "$ROOT/tools/probe_host.py" > "$OUT/probe_host.json" 2>/dev/null; echo "probe rc=$?"
  python3 "$ROOT/bench.py" > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"; echo "bench rc=$?"
  python3 "$ROOT/bench.py" --workload ml25m --no-cpu-baseline > "$OUT/bench_ml25m.json" 2> "$OUT/bench_ml25m.err"; echo "bench ml25m rc=$?"
  python3 "$ROOT/bench.py" --gpus 2 --backend gloo --force-device 0 --no-hbm-leg > "$OUT/bench_2ranks_gloo_one_gpu.json" 2> "$OUT/bench_2ranks.err"; echo "bench 2 ranks rc=$?"
  python3 "$ROOT/bench.py" --gpus 1 --force-dist --no-cpu-baseline > "$OUT/bench_1rank_rccl.json" 2> "$OUT/bench_1rank_rccl.err"; echo "bench rccl rc=$?"
fi
if has stats; then
  # the headline region alone: the process launches mf_step_kernel in the warm-up (21), the inclusive region (350) and the event-pair pass (7) only, so
  # the kernel's average here is what roofline.avg_launch_us must agree with; its stdout (the bench line of that run) is kept beside it
  rm -rf "$OUT/trace_headline"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_headline" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-hbm-leg --no-cold --no-unlearn --no-ot --no-resident --roofline-steps 1 > "$OUT/bench_headline.json" 2> /dev/null; echo "headline rc=$?"
  f=$(find "$OUT/trace_headline" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$OUT/bench_headline_kernel_stats.csv"; rm -rf "$OUT/trace_headline"
  stats bench "$ROOT/bench.py" --no-cpu-baseline --no-hbm-leg --no-cold
  stats bench_ml25m "$ROOT/bench.py" --workload ml25m --no-cpu-baseline --no-unlearn --no-hbm-leg --steps 5 --warmup 1
  stats e2e_sisa "$ROOT/tools/e2e_sisa.py"
fi
if has cfg4; then
  stats cfg4_d16 "$ROOT/bench.py" --shards 16 --d 16 --steps 60 --warmup 10 --no-cpu-baseline --no-unlearn --no-hbm-leg
fi
if has pmc; then
  python3 "$ROOT/tools/pmc_traffic.py" "$OUT/pmc" --tag "$TAG" > "$OUT/pmc_ml1m.log" 2>&1; echo "pmc ml1m rc=$?"
  python3 "$ROOT/tools/pmc_traffic.py" "$OUT/pmc" --tag "$TAG" -- --workload ml25m > "$OUT/pmc_ml25m.log" 2>&1; echo "pmc ml25m rc=$?"
  python3 "$ROOT/tools/pmc_traffic.py" "$OUT/pmc" --tag "$TAG" -- --workload ml25m --shards 1 --d 128 --steps 1 --warmup 1 --roofline-steps 1 > "$OUT/pmc_fullmf.log" 2>&1; echo "pmc fullmf touch rc=$?"
  python3 "$ROOT/tools/pmc_traffic.py" "$OUT/pmc" --tag "$TAG" -- --workload ml25m --d 16 > "$OUT/pmc_ml25m_k16.log" 2>&1; echo "pmc ml25m k16 rc=$?"
  cp "$OUT"/pmc/*_pmc_hbm_traffic_*.json "$OUT"/ 2>/dev/null
  rm -rf "$OUT/pmc" "$OUT/pmc_dense"
fi
if has e2e; then
  URE_HOST_TRACE=1 python3 "$ROOT/tools/profile_e2e.py" > "$OUT/host_profile_sisa_learn.txt" 2>&1
  python3 "$ROOT/tools/e2e_sisa.py" > "$OUT/e2e_sisa.json" 2>/dev/null
  python3 "$ROOT/tools/e2e_sisa.py" --shards 16 --k 16 > "$OUT/e2e_sisa_config4.json" 2>/dev/null
  python3 "$ROOT/tools/e2e_cold.py" > "$OUT/e2e_cold.json" 2>/dev/null
  python3 "$ROOT/tools/profile_ot.py" --rounds 6 > "$OUT/ot_rounds.json" 2>/dev/null
fi
if has pmcfull; then
  python3 "$ROOT/tools/pmc_traffic.py" "$OUT/pmc" --tag "$TAG" -- --workload ml25m --shards 1 --d 128 --steps 1 --warmup 1 --roofline-steps 1 > "$OUT/pmc_fullmf.log" 2>&1; echo "pmc fullmf touch rc=$?"
  cp "$OUT"/pmc/*_pmc_hbm_traffic_*.json "$OUT"/ 2>/dev/null
  rm -rf "$OUT/pmc"
fi
if has fullmf; then
  # full MF at the 25 M shape (750 steps per epoch): touch_mode 3 (the epoch's slots sorted by step) against 64-step windows (round 3)
  ARGS="--workload ml25m --shards 1 --d 128 --no-cpu-baseline --no-unlearn --no-hbm-leg --steps 1 --warmup 1 --roofline-steps 1"
  timeout -k 10 500 python3 "$ROOT/bench.py" $ARGS > "$OUT/fullmf25m_d128_index.json" 2> /dev/null; echo "fullmf index rc=$?"
  URE_TOUCH_INDEX=0 timeout -k 10 500 python3 "$ROOT/bench.py" $ARGS > "$OUT/fullmf25m_d128_touch_windows.json" 2> /dev/null; echo "fullmf windows rc=$?"
  stats fullmf_index "$ROOT/tools/exp_index.py" --epochs 3
fi
if has k16; then
  # BASELINE.json configs[3]'s 32 shards at k = 16 (25-27 steps per epoch: touch_mode 3 with the short-epoch scatter), per kernel
  stats cfg3_d16_index "$ROOT/bench.py" --workload ml25m --shards 32 --d 16 --no-cpu-baseline --no-unlearn --no-hbm-leg --steps 3 --warmup 1
fi
if has multirank; then
  timeout -k 10 500 python3 "$ROOT/tools/multirank_timeline.py" > "$OUT/multirank_timeline_2ranks_ml25m_k128.json" 2> "$OUT/multirank.err"; echo "multirank rc=$?"
  timeout -k 10 300 python3 "$ROOT/tools/timeline_request.py" > "$OUT/timeline_request_ml25m_s32_k128.txt" 2>&1; echo "timeline rc=$?"
fi
if has big; then
  # BASELINE.json configs[3] through the operator surface: per-epoch logs from compact snapshots (round 2: NaN beyond 8 GiB of full ones)
  timeout -k 10 900 python3 "$ROOT/tools/e2e_sisa.py" --workload ml25m --shards 32 --k 128 --epochs 5 --reps 4 > "$OUT/e2e_sisa_ml25m_s32_k128_e5.json" 2> "$OUT/e2e_sisa_ml25m.err"; echo "e2e ml25m rc=$?"
  # the same at the reference's default width k = 16 (config.py:19), where the arithmetic stays finite
  timeout -k 10 900 python3 "$ROOT/tools/e2e_sisa.py" --workload ml25m --shards 32 --k 16 --epochs 5 --reps 4 > "$OUT/e2e_sisa_ml25m_s32_k16_e5.json" 2>> "$OUT/e2e_sisa_ml25m.err"; echo "e2e ml25m k16 rc=$?"
fi
if has shuffle; then
  # the two device shuffles on the shapes a request makes (csrc/perm_chain.hip against csrc/perm_tags.hip), and the chain's kernels per shape
  timeout -k 10 300 python3 "$ROOT/tools/exp_shuffle.py" > "$OUT/exp_shuffle.json" 2> /dev/null; echo "exp_shuffle rc=$?"
  stats exp_shuffle_request "$ROOT/tools/exp_shuffle.py" --only request_5x50x180k --which chain --reps 10
  stats exp_shuffle_first_epoch "$ROOT/tools/exp_shuffle.py" --only first_epoch_5x180k --which chain --reps 10
  stats exp_shuffle_22m "$ROOT/tools/exp_shuffle.py" --only one_epoch_22.5M --which chain --reps 10
  URE_SHUFFLE=reservations timeout -k 10 300 python3 "$ROOT/bench.py" --no-cpu-baseline --no-hbm-leg --no-cold --no-ot > "$OUT/bench_shuffle_reservations.json" 2> /dev/null; echo "bench reservations rc=$?"
fi
ls -la "$OUT"
