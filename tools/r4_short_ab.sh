#!/bin/bash
# configs[3]'s shape at d = D (default 16) in touch_mode 3 with experiment builds of the library: bash tools/r4_short_ab.sh OUTDIR lib1.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/${1:-r4sab}; shift
mkdir -p "$OUT"
run() {
  timeout -k 10 400 python3 "$ROOT/bench.py" --workload ml25m --shards 32 --d ${D:-16} --no-cpu-baseline --no-unlearn --steps 3 --warmup 1 --roofline-steps 2 > "$OUT/b.json" 2> "$OUT/err.txt" || exit 1
  python3 - "$1" <<PY
import json, sys
j = json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); r = j['roofline']
print(sys.argv[1], 'mode', r['touch_mode'], 'value %.3f G/s' % (j['value'] / 1e9), 'ms_per_step', j['ms_per_step'], 'avg_launch_us', r['avg_launch_us'], 'prep share', j.get('prep_share_of_device_time'))
PY
}
run product
for lib in "$@"; do URE_LIB="$ROOT/$lib" URE_ALLOW_STALE_LIB=1 run "$lib"; done
