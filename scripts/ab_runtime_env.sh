# same-box sweep of HIP runtime switches that touch graph execution: each line = one bench run (blocks/s, ms/step, host ms per
# graph launch).  usage: scripts/ab_runtime_env.sh "VAR=V" "VAR=V VAR2=V" ...   ("" = default environment)
for round in 1 2; do
  for v in "$@"; do
    env $v timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-traffic --no-other-mode 2>/dev/null | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', d['value'], d['ms_per_step'], d.get('host_graph_launch_ms'))
except Exception as e:
    print('[$v] failed', e)"
  done
done
