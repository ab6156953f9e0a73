# same-box comparison of several settings of one environment switch on the working tree, two interleaved rounds:
#   scripts/ab_env_n.sh VAR V1 V2 [V3 ...]      e.g.  scripts/ab_env_n.sh SGNN_TUNE conv_dw_blocks=256 conv_dw_blocks=128
# prints: setting, blocks/s, ms/step, library launches per step, convolution ms per step (HIP events, eager roofline leg)
VAR=$1; shift
for round in 1 2; do
  for v in "$@"; do
    env $VAR=$v timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-traffic --no-other-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$VAR=$v', d['value'], d['ms_per_step'], d.get('launches_per_step'), r.get('conv_ms_per_step'))"
  done
done
