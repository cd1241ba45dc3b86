# codec-only pass time (G = 16, 32) and the C3 / 256-slot bench value for each environment setting given as an argument ("-" = defaults)
cd /root/repo
one() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   bench', d['value'], d['ms_per_step'])"; }
for V in "$@"; do
  [ "$V" = "-" ] && V="Q3_NOP=1"
  echo "=== $V"
  env $V timeout -k 10 200 python scripts/dev_gpu_codec_group.py | grep -E "G=(16|32)" || exit 1
  env $V timeout -k 10 300 python bench.py --no-cpu-baseline --no-c2-leg 2>/dev/null | one || exit 1
  env $V timeout -k 10 300 python bench.py --no-cpu-baseline --no-c2-leg --batch 256 --requests 256 2>/dev/null | one || exit 1
done
