#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/${1:-r4m}; mkdir -p $OUT
for i in 1 2; do for v in 1 0; do URE_INDEX_OVERLAP=$v timeout -k 10 300 python3 $ROOT/tools/exp_index.py --epochs 3 >> $OUT/overlap_ab.jsonl 2>> $OUT/overlap_ab.err || exit 1; done; done
cat $OUT/overlap_ab.jsonl
