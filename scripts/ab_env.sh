#!/bin/bash
# same-box A/B of environment switches on the current build: C3 bench (no CPU baseline) per setting, twice, interleaved
# usage: bash scripts/ab_env.sh TAG "VAR=a" "VAR=b" ...
set -o pipefail
TAG=${1:-abenv}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2; do
  for v in "$@"; do
    name=$(echo "$v" | tr ' =' '__')
    env $v timeout -k 10 300 python bench.py --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_${name}_$rep.json 2> $OUT/bench_${name}_$rep.err || { echo "$v failed"; tail -3 $OUT/bench_${name}_$rep.err; continue; }
    python - <<PY
import json
o=json.load(open("$OUT/bench_${name}_$rep.json"))
print("$v rep $rep: C3 %.1f audio-s/s  ms/step %.2f  step-frame %.3f ms  | C2 frame %.3f ms rtf %.4f first %.1f ms" % (o["value"], o["ms_per_step"], o["batch_decode_ms_per_step_frame"], o.get("decode_ms_per_frame",0), o.get("rtf",0), o.get("first_chunk_ms_p50",0)))
PY
  done
done
