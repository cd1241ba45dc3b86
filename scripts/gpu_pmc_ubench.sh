#!/bin/bash
# SQ counters of the batched int8 GEMM in isolation (scripts/bin/ubench_gemm quick): where the waves' cycles go
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_ubench}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $OUT/sq -o p --output-format csv -- $GRAFT_REPO_ROOT/scripts/bin/ubench_gemm quick > $OUT/sq.log 2>&1 || { tail -5 $OUT/sq.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_MFMA -d $OUT/sq2 -o p --output-format csv -- $GRAFT_REPO_ROOT/scripts/bin/ubench_gemm quick > $OUT/sq2.log 2>&1 || { tail -5 $OUT/sq2.log; }
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, collections, glob
for d in ("sq", "sq2"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True)
    if not fs: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("void q3::", "").split("(")[0]
        agg[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in agg.items():
        if "gemm" not in key[0]: continue
        print(key, {c: round(sum(v) / len(v)) for c, v in cs.items()})
PY
