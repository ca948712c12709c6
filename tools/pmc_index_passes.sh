#!/bin/bash
# rocprofv3 PMC passes over one epoch of full MF at the 25 M shape in touch_mode 3 (tools/exp_index.py), one pass per counter group, never combined with
# a tracing domain other than --kernel-trace: what the epoch-start kernels (csrc/mf_index.h) wait for.  bash tools/pmc_index_passes.sh [TAG]
# Writes gpurun_out/pmc_index_TAG/summary.json (tools/pmc_summary.py).  A group the hardware cannot collect together makes rocprofv3 abort: every pass
# runs under timeout and the next one starts only if it came back.
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_index_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_[A-Z_]*WRITE[A-Z_]*\|TCP_[A-Z_]*WRITE[A-Z_]*" | sort -u > "$OUT/avail_tcc.txt"
n=0
while read -r group; do
  [ -z "$group" ] && continue
  n=$((n+1))
  echo "pass $n: $group"
  timeout -k 10 200 rocprofv3 --pmc $group --kernel-trace --output-format csv -d "$OUT/$n" -- python3 "$ROOT/tools/exp_index.py" --epochs 1 > "$OUT/$n.log" 2>&1 || { echo "pass $n failed"; tail -3 "$OUT/$n.log"; }
done <<'GROUPS'
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum TCC_READ_sum
SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU
SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
GROUPS
python3 "$ROOT/tools/pmc_summary.py" "$OUT" > "$OUT/summary.json"
rm -rf "$OUT"/[0-9]*/
echo done
