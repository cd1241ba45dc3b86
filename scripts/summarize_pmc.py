"""Per-(kernel, grid) mean of one rocprofv3 --pmc counter (FETCH_SIZE / WRITE_SIZE, unit KiB-ish per MI355X_MICROARCH.md)."""
import collections
import csv
import sys

agg = collections.defaultdict(list)
name = None
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void q3::", "").replace("q3::", "").split("(")[0]
    agg[(k, r["Grid_Size"], r["Workgroup_Size"])].append(float(r["Counter_Value"]))
    name = r["Counter_Name"]
print("kernel,grid,workgroup,launches,mean_%s,min,max" % name)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%s,%s,%s,%d,%.1f,%.1f,%.1f" % (k[0].replace(",", ";"), k[1], k[2], len(v), sum(v) / len(v), min(v), max(v)))
