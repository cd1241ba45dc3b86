#!/bin/bash
# same-box A/B of several builds of libq3tts.so (scripts/bin/ab/libq3tts_<name>.so): C3 bench (no CPU baseline) with each, twice, interleaved
# usage: bash scripts/ab_libs.sh TAG name1 name2 ...
set -o pipefail
TAG=${1:-ab}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
cp qwen3-tts-rust_amd/libq3tts.so $OUT/lib_backup.so
for rep in 1 2; do
  for v in "$@"; do
    cp scripts/bin/ab/libq3tts_$v.so qwen3-tts-rust_amd/libq3tts.so
    env ${AB_ENV:-Q3_NOP=1} timeout -k 10 300 python bench.py --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_${v}_$rep.json 2> $OUT/bench_${v}_$rep.err || { echo "$v failed"; tail -3 $OUT/bench_${v}_$rep.err; }
    python - <<PY
import json
o=json.load(open("$OUT/bench_${v}_$rep.json"))
print("$v rep $rep: C3 %.1f audio-s/s  step-frame %.3f ms  | C2 frame %.3f ms rtf %.4f" % (o["value"], o["batch_decode_ms_per_step_frame"], o.get("decode_ms_per_frame",0), o.get("rtf",0)))
PY
  done
done
cp $OUT/lib_backup.so qwen3-tts-rust_amd/libq3tts.so; rm -f $OUT/lib_backup.so
