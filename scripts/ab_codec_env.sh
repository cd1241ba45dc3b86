# A/B of one codec switch on the GPU box: parity subset with the switch on, then codec-only and C3 bench with it on and off.
# usage: bash scripts/ab_codec_env.sh VAR=VALUE
set -e
cd /root/repo
V="$1"
env $V timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "codec or fullsize or engine_matches or decoder_state" 2>&1 | tail -5
echo "--- $V"; env $V timeout -k 10 200 python scripts/dev_gpu_codec_group.py
echo "--- default"; timeout -k 10 200 python scripts/dev_gpu_codec_group.py
echo "--- bench $V"; env $V timeout -k 10 300 python bench.py --no-cpu-baseline --no-c2-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "--- bench default"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-c2-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "--- bench 256 $V"; env $V timeout -k 10 300 python bench.py --no-cpu-baseline --no-c2-leg --batch 256 --requests 256 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "--- bench 256 default"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-c2-leg --batch 256 --requests 256 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
