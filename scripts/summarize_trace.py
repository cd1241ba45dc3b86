"""Compacts a rocprofv3 --kernel-trace CSV (tens of MB) into a per-(kernel, grid) table that fits under profiles/."""
import collections
import csv
import glob
import statistics
import sys

src, dst = sys.argv[1], sys.argv[2]
f = (glob.glob(src + "/*/*_kernel_trace.csv") + glob.glob(src + "/*_kernel_trace.csv"))[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("void q3::", "").replace("q3::", "").split("(")[0]
    key = (name, "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]), r["Workgroup_Size_X"], r["VGPR_Count"], r["LDS_Block_Size"])
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
with open(dst, "w") as o:
    o.write("kernel,grid_threads,workgroup,vgpr,lds_bytes,launches,avg_us,median_us,min_us,total_ms,percent\n")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        o.write("%s,%s,%s,%s,%s,%d,%.2f,%.2f,%.2f,%.3f,%.2f\n" % (k[0].replace(",", ";"), k[1], k[2], k[3], k[4], len(v), sum(v) / len(v), statistics.median(v), min(v),
                                                      sum(v) / 1e3, 100 * sum(v) / tot))
