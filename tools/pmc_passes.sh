#!/bin/bash
# rocprofv3 PMC passes over the bench workload, one pass per counter group (never combined with
# tracing domains other than --kernel-trace).  Usage on the GPU box:  bash tools/pmc_passes.sh TAG
# Writes gpurun_out/pmc_TAG/<n>/ and gpurun_out/pmc_TAG/summary.json (tools/pmc_summary.py).
# Optional 2nd argument: a file with one counter group per line (default: the list below).  A
# group the hardware cannot collect together makes rocprofv3 abort: every pass runs under timeout.
set -e
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
GROUPS_FILE=${2:+$(readlink -f "$2")}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
n=0
while read -r group; do
  [ -z "$group" ] && continue
  n=$((n+1))
  echo "pass $n: $group"
  timeout -k 10 150 rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/$n" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-unlearn --steps 5 --warmup 1 > "$OUT/$n.log" 2>&1
done < <(if [ -n "$GROUPS_FILE" ]; then cat "$GROUPS_FILE"; else cat <<'GROUPS'
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum
TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum
SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU
SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU
GRBM_GUI_ACTIVE GRBM_COUNT
GROUPS
fi)
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
