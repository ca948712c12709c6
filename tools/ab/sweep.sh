#!/bin/bash
# bench every tools/ab/lib_*.so variant (copied over the product library one at a time)
cp ultrare_amd/libultrare_hip.so /tmp/product.so
for f in tools/ab/lib_*.so; do
  cp $f ultrare_amd/libultrare_hip.so
  r=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-unlearn $SWEEP_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['avg_launch_us'])") || { cp /tmp/product.so ultrare_amd/libultrare_hip.so; exit 1; }
  echo "$(basename $f) $r"
done
cp /tmp/product.so ultrare_amd/libultrare_hip.so
