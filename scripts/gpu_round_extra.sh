#!/bin/bash
# after scripts/gpu_round.sh: the other published bench lines (256 slots, ragged C3, C5 clone, C2 for both quantisations) and two launcher rehearsals
# (torchrun with one rank; the RCCL code path forced at world size 1).  Outputs under gpurun_out/<tag>/.
set -o pipefail
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
line() { python - "$1" <<'PY'
import json, sys
o = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%s: value %.1f %s  ms/step %.2f  rtf %s  frame %s ms  first %s ms  n_gpus %d" % (sys.argv[1].split("/")[-1], o["value"], o["unit"], o["ms_per_step"], o.get("rtf"), o.get("decode_ms_per_frame"), o.get("first_chunk_ms_p50"), o["n_gpus"]))
PY
}
timeout -k 10 300 python bench.py --no-cpu-baseline --no-c2-leg --batch 256 --requests 256 > $OUT/bench_b256.json 2> $OUT/e.err && line $OUT/bench_b256.json || tail -3 $OUT/e.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-c2-leg --requests 192 --ragged > $OUT/bench_c3_ragged192.json 2> $OUT/e.err && line $OUT/bench_c3_ragged192.json || tail -3 $OUT/e.err
timeout -k 10 300 python bench.py --config c5 --no-cpu-baseline > $OUT/bench_c5.json 2> $OUT/e.err && line $OUT/bench_c5.json || tail -3 $OUT/e.err
timeout -k 10 300 python bench.py --config c2 --no-cpu-baseline > $OUT/bench_c2.json 2> $OUT/e.err && line $OUT/bench_c2.json || tail -3 $OUT/e.err
timeout -k 10 300 python bench.py --quant q5_k_m --no-cpu-baseline --no-c2-leg > $OUT/bench_c3_q5_k_m.json 2> $OUT/e.err && line $OUT/bench_c3_q5_k_m.json || tail -3 $OUT/e.err
timeout -k 10 300 python bench.py --config c2 --quant q5_k_m --no-cpu-baseline > $OUT/bench_c2_q5_k_m.json 2> $OUT/e.err && line $OUT/bench_c2_q5_k_m.json || tail -3 $OUT/e.err
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-c2-leg > $OUT/bench_torchrun1.json 2> $OUT/e.err && line $OUT/bench_torchrun1.json || tail -3 $OUT/e.err
Q3_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --no-c2-leg > $OUT/bench_rccl1.json 2> $OUT/e.err && line $OUT/bench_rccl1.json || tail -3 $OUT/e.err
