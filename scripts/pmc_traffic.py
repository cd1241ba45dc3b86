"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: separate passes, kernel-trace only, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes; FETCH_SIZE x 2 on gfx950 for wide coalesced reads, unit KiB).
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <out.md> <round tag>
The json is what bench.py's roofline.traffic reads (key = the kernel key bench.py uses, value = bytes per launch)."""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("void q3::", "").replace("q3::", "").split("(")[0]
        agg[(k, r["Grid_Size"], r["Workgroup_Size"])].append(float(r["Counter_Value"]))
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
tag = sys.argv[5] if len(sys.argv) > 5 else "r02"
# algorithmic bytes per launch of the weight-streaming launches of Q3TTS-1.7B-synth Q8_0 (rows x K x 1.0625 B), keyed by (kernel prefix, grid threads)
ALG = {("k_gemm_q8_mfma<true", "196608"): 12288 * 2048 * 1.0625, ("k_gemm_q8_tile1<true", "196608"): 12288 * 2048 * 1.0625,
       ("k_gemm_q8_tile1<true", "49152"): 6144 * 1024 * 1.0625,
       ("k_gateup_swiglu<1, 8", "98304"): 12288 * 2048 * 1.0625, ("k_gateup_swiglu<1, 4", "24576"): 6144 * 1024 * 1.0625}
rows, table = [], {}
for key, v in sorted(fetch.items(), key=lambda kv: -sum(kv[1]))[:24]:
    rd = sum(v) / len(v) * 1024 * 2
    wv = write.get(key)
    wr = (sum(wv) / len(wv) * 1024) if wv else 0.0
    alg = next((a for (pfx, grid), a in ALG.items() if key[0].startswith(pfx) and key[1] == grid), None)
    rows.append((key, len(v), rd, wr, alg))
    if alg:
        name = key[0].split("<")[0] + ("<true>" if "<true" in key[0] else "<" + key[0].split("<", 1)[1] if "<" in key[0] else "")
        table.setdefault(name if "gemm" in name else key[0], {"read_bytes": rd, "write_bytes": wr, "launches": len(v), "grid_threads": key[1], "algorithmic_bytes": alg})
table["_meta"] = {"collected": "%s, this commit's kernels" % tag}
json.dump(table, open(sys.argv[3], "w"), indent=1)
lines = ["# HBM traffic per launch from rocprofv3 PMC passes (%s)" % tag, "",
         "    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-c2-leg",
         "    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-c2-leg", "",
         "FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE counts half of a wide coalesced read stream (MI355X_MICROARCH.md), so",
         "read bytes = FETCH_SIZE x 1024 x 2.  `bench.py` takes `roofline.traffic` (read + write bytes per launch) from `hbm_traffic_latest.json`.", "",
         "| kernel | grid (threads) | launches | read MB | write MB | algorithmic MB | read / algorithmic |", "|---|---|---|---|---|---|---|"]
for key, n, rd, wr, alg in rows:
    lines.append("| %s | %s | %d | %.2f | %.2f | %s | %s |" % (key[0], key[1], n, rd / 1e6, wr / 1e6, "%.2f" % (alg / 1e6) if alg else "-", "%.2f" % (rd / alg) if alg else "-"))
open(sys.argv[4], "w").write("\n".join(lines) + "\n")
print(json.dumps(table))
