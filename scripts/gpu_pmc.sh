#!/bin/bash
# PMC passes for HBM traffic (FETCH_SIZE / WRITE_SIZE, separate runs) + MFMA utilisation of the C3 bench; outputs under gpurun_out/<tag>/pmc
set -o pipefail
TAG=${1:-pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-c2-leg ${BENCH_ARGS:-} > $OUT/$c.log 2>&1 || { echo "$c pass failed"; tail -5 $OUT/$c.log; exit 1; }
done
cd $GRAFT_REPO_ROOT
F=$(find $OUT/FETCH_SIZE -name "*counter_collection.csv" | head -1); W=$(find $OUT/WRITE_SIZE -name "*counter_collection.csv" | head -1)
python3 scripts/pmc_traffic.py $F $W $OUT/hbm_traffic.json $OUT/hbm_traffic.md ${ROUND_TAG:-r02}
find $OUT -name "*.csv" -size +30M -delete
