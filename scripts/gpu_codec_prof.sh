#!/bin/bash
# per-kernel durations of the codec alone (scripts/dev_gpu_codec_group.py) under rocprofv3, for A/B of codec GEMM forms
set -o pipefail
TAG=${1:-codecprof}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o c --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/dev_gpu_codec_group.py > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_trace.py $OUT/prof $OUT/trace_summary.csv
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
tail -4 $OUT/run.log
head -30 $OUT/trace_summary.csv | cut -c1-150
