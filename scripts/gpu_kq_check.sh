# K-quant path on the GPU box: parity subset, then the C2 line for Q5_K_M and Q8_0
cd /root/repo
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "q5_k_m or formats or gemv or engine or tf_eval or replay" 2>&1 | tail -4
for q in q5_k_m q8_0; do
  timeout -k 10 300 python bench.py --config c2 --quant $q --no-cpu-baseline > gpurun_out/kq_c2_$q.json 2> gpurun_out/kq_c2.err || tail -5 gpurun_out/kq_c2.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/kq_c2_$q.json").read().strip().splitlines()[-1])
print("$q", {k: d[k] for k in ("value", "rtf", "decode_ms_per_frame", "first_chunk_ms_p50") if k in d})
r = d["roofline"]; print({k: r[k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "bytes_per_launch")})
PY
done
