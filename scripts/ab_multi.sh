#!/bin/bash
# same-box comparison of several environment settings on the working tree, two interleaved rounds; each argument is a
# space-separated list of VAR=value pairs ("-" = the defaults):
#   scripts/ab_multi.sh - "SGNN_TUNE=sgnn_prog_set_bn_fold=1,sgnn_prog_set_bn_fold_rows=40000" "SGNN_DENSE_RULEBOOK_MIN_ROWS=4096"
# prints: setting, blocks/s, ms/step, library launches per step, convolution ms per step (HIP events, eager roofline leg)
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then E=""; else E="$v"; fi
    env $E timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-traffic --no-other-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('[$v]', d['value'], d['ms_per_step'], d.get('launches_per_step'), r.get('conv_ms_per_step'))"
  done
done
