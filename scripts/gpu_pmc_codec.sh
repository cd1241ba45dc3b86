#!/bin/bash
# SQ counters of the codec's GEMM kernels (codec alone): is the split-f16 conv GEMM issue-bound (VALU) or parked (memory)?
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_codec}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $OUT/sq -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/dev_gpu_codec_group.py > $OUT/sq.log 2>&1 || { tail -5 $OUT/sq.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, collections, glob
fs = glob.glob("$OUT/sq/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].replace("void q3::", "").split("(")[0]
    agg[(k[:40], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for key, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    n = len(next(iter(cs.values())))
    rows.append((m.get("SQ_WAVE_CYCLES", 0) * n, key, m, n))
for _, key, m, n in sorted(rows, reverse=True)[:14]:
    wc = m.get("SQ_WAVE_CYCLES", 1) or 1
    print(key, n, "wave_cyc %.2e parked %.0f%% stall %.0f%% active %.0f%% valu_active %.0f%% | insts_valu %.2e mfma_busy_cyc %.2e busy_cyc %.2e" % (
        wc, 100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        100 * m.get("SQ_ACTIVE_INST_VALU", 0) / wc, m.get("SQ_INSTS_VALU", 0), m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), m.get("SQ_BUSY_CYCLES", 0)))
PY
find $OUT -name "*.csv" -size +20M -delete
