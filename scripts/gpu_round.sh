#!/bin/bash
# One GPU-box visit: parity tests, default bench (C3 + C2 leg), optional A/B leg(s), kernel stats.  Outputs under gpurun_out/<tag>/.
set -o pipefail
TAG=${1:-run}
shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 ${TEST_TIMEOUT:-600} python -m pytest tests -m gpu -x -q ${TEST_ARGS:-} > $OUT/pytest.log 2>&1
  echo "pytest rc=$?" | tee -a $OUT/pytest.log
  tail -4 $OUT/pytest.log
  grep -q "pytest rc=0" $OUT/pytest.log || exit 1
fi
timeout -k 10 300 python bench.py ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
python - <<PY
import json
o=json.load(open("$OUT/bench.json"))
print("C3 value %.1f  ms/step %.2f  step-frame %.2f ms  gu %.1f us frac %.3f | C2 rtf %.4f frame %.3f ms first %.1f ms" % (o["value"], o["ms_per_step"], o["batch_decode_ms_per_step_frame"], o["roofline"]["avg_launch_us"], o["roofline"]["frac"], o.get("rtf",0), o.get("decode_ms_per_frame",0), o.get("first_chunk_ms_p50",0)))
PY
for ab in "$@"; do
  name=$(echo "$ab" | tr ' =' '__')
  env $ab timeout -k 10 300 python bench.py --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_$name.json 2> $OUT/bench_$name.err || { echo "A/B $ab failed"; tail -3 $OUT/bench_$name.err; continue; }
  python - <<PY
import json
o=json.load(open("$OUT/bench_$name.json"))
print("A/B $ab: value %.1f  step-frame %.2f ms  gu %.1f us | C2 frame %.3f ms" % (o["value"], o["batch_decode_ms_per_step_frame"], o["roofline"]["avg_launch_us"], o.get("decode_ms_per_frame",0)))
PY
done
if [ "${PROFILE:-1}" = "1" ]; then
  cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o c3 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-c2-leg ${BENCH_ARGS:-} > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1
  cd $GRAFT_REPO_ROOT
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -12 $OUT/kernel_stats.csv | cut -c1-150
  find $OUT/prof -name "*kernel_trace.csv" -size +40M -delete
fi
