# C5 (bf16, clone prompts) A/B of one environment switch: parity subset, then the bench twice each way
cd /root/repo
V="$1"
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "bf16 or float or f16" 2>&1 | tail -3
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   c5', round(d['value'],1), round(d['ms_per_step'],2), 'prefill', round(d.get('prefill_ms',0),1))"; }
for X in "$V" Q3_NOP=1 "$V" Q3_NOP=1; do echo "=== $X"; env $X timeout -k 10 200 python bench.py --config c5 --no-cpu-baseline --no-c2-leg 2>/dev/null | one; done
