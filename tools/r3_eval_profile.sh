cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/evalx; mkdir -p $OUT
cd $ROOT && timeout -k 10 500 python -m pytest tests -x -q -m gpu -k "eval or metrics or series or golden or ndcg or pins or surface" > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/e2e_sisa.py > $OUT/e2e.json 2> $OUT/e2e.err; echo "prof rc=$?"
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/e2e_sisa_kernel_stats.csv; rm -rf $OUT/trace
head -12 $OUT/e2e_sisa_kernel_stats.csv
