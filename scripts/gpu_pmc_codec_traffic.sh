#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and SQ counters of the codec's kernels, codec alone (scripts/dev_gpu_codec_group.py).
# usage (on the GPU box): bash scripts/gpu_pmc_codec_traffic.sh TAG
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc_codec_traffic}
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for g in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU"; do
  rocprofv3 --kernel-trace --pmc $g -d $OUT/g$i -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/dev_gpu_codec_group.py > $OUT/g$i.log 2>&1 || { echo "pass $g failed"; tail -5 $OUT/g$i.log; exit 1; }
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void q3::", "").replace("q3::", "").split("(")[0]
        k = re.sub(r'_ZN2q3\d+(k_\w+?)I((?:L[ib]\d+E)+)EvNS_8GemmArgsEPKDF16_S3_', lambda m: m.group(1) + "<" + ",".join(re.findall(r'L[ib](\d+)E', m.group(2))) + ">", k)
        agg[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/codec_counters.md", "w") as o:
    o.write("| kernel | grid threads | launches | HBM read MB (FETCH_SIZE x 1 KiB x 2, gfx950) | HBM written MB | parked | issue stall | active | VALU active | MFMA busy cycles |\n|---|---|---|---|---|---|---|---|---|---|\n")
    rows = []
    for (k, g), v in agg.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        n = len(next(iter(v.values())))
        rows.append((m.get("SQ_WAVE_CYCLES", 0) * n, k, g, n, m))
    for _, k, g, n, m in sorted(rows, reverse=True)[:24]:
        wc = m.get("SQ_WAVE_CYCLES", 0) or 1

        rd = m.get("FETCH_SIZE", 0) * 1024 * 2 / 1e6   # gfx950: FETCH_SIZE reports half of the bytes of wide streaming reads (MI355X_MICROARCH.md)
        wr = m.get("WRITE_SIZE", 0) * 1024 / 1e6
        o.write("| %s | %s | %d | %.1f | %.1f | %.0f %% | %.0f %% | %.0f %% | %.0f %% | %.2e |\n" % (k, g, n, rd, wr, 100 * m.get("SQ_WAIT_ANY", 0) / wc, 100 * m.get("SQ_WAIT_INST_ANY", 0) / wc,
                100 * m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * m.get("SQ_ACTIVE_INST_VALU", 0) / wc, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)))
print(open("$OUT/codec_counters.md").read())
PY
find $OUT -name "*.csv" -size +20M -delete
