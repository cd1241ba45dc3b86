#!/bin/bash
# runs the prebuilt scripts/bin/ubench_chain (scripts/build_ubench.sh, cross-compiled before the gpurun call); output under gpurun_out/<tag>/ubench.txt
set -o pipefail
TAG=${1:-ubench}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
[ -x scripts/bin/ubench_chain ] || bash scripts/build_ubench.sh || exit 1
timeout -k 10 300 scripts/bin/ubench_chain "$@" > $OUT/ubench.txt 2>&1
echo "ubench rc=$?"
cat $OUT/ubench.txt
