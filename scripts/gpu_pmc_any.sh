#!/bin/bash
# one rocprofv3 --pmc pass per counter group over a bench.py run; prints per-kernel averages of every counter
# usage: bash scripts/gpu_pmc_any.sh TAG "CTR_A CTR_B" ["CTR_C" ...] -- <bench.py args>
set -o pipefail
TAG=$1; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for g in "${GROUPS_[@]}"; do
  rocprofv3 --kernel-trace --pmc $g -d $OUT/g$i -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $OUT/g$i.log 2>&1 || { echo "pass $g failed"; tail -5 $OUT/g$i.log; exit 1; }
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void q3::", "").replace("q3::", "").split("(")[0]
        agg[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in agg.values() for c in v})
print("kernel,grid,launches," + ",".join(names))
for (k, g), v in sorted(agg.items(), key=lambda kv: -len(next(iter(kv[1].values()))))[:40]:
    n = len(next(iter(v.values())))
    print("%s,%s,%d," % (k[:60], g, n) + ",".join("%.1f" % (sum(v[c]) / len(v[c])) if c in v else "" for c in names))
PY
find $OUT -name "*.csv" -size +20M -delete
