#!/bin/bash
# rocprofv3 kernel stats of the single-utterance leg (C2) for each quantisation given: bash scripts/gpu_prof_c2.sh q5_k_m q8_0
set -o pipefail
export TMPDIR=/tmp
for q in "$@"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_c2_$q
  rm -rf $OUT; mkdir -p $OUT
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $OUT -o c --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --quant $q --no-cpu-baseline --steps 16 --warmup 2 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
  cd $GRAFT_REPO_ROOT
  find $OUT -name "*kernel_trace.csv" -delete
  echo "=== $q"
  python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/c_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%6.2f%% %7d calls %8.2f us avg  %s" % (100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
done
