# norm-as-GEMM-tail (norm_tail.h): batched-path parity subset with a short timeout first, then the C3 / 256-slot bench with it on and off
set -e
cd /root/repo
timeout -k 10 120 python -m pytest tests -m gpu -x -q -k "wide_batch or batched or batch" 2>&1 | tail -4
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "fullsize or engine or tf_eval or scheduler or group" 2>&1 | tail -4
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench', d['value'], d['ms_per_step'], d.get('batch_decode_ms_per_step_frame'))"; }
for V in Q3_NORM_TAIL=1 Q3_NORM_TAIL=0 Q3_NORM_TAIL=1 Q3_NORM_TAIL=0; do
  echo "=== $V"
  env $V timeout -k 10 200 python bench.py --no-cpu-baseline --no-c2-leg 2>/dev/null | one
  env $V timeout -k 10 200 python bench.py --no-cpu-baseline --no-c2-leg --batch 256 --requests 256 2>/dev/null | one
done
