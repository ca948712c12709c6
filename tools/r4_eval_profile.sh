#!/bin/bash
# kernel-trace stats of the e2e learn + unlearn (ultrare_amd.measure.sisa_request through tools/e2e_sisa.py): the share of evaluation
# in device time.  bash tools/r4_eval_profile.sh OUTDIR
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/${1:-r4j}; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/e2e_sisa.py > $OUT/e2e.json 2> $OUT/e2e.err; echo "prof rc=$?"
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/e2e_sisa_kernel_stats.csv; rm -rf $OUT/trace
python3 $ROOT/tools/kstats.py $OUT/e2e_sisa_kernel_stats.csv 16
