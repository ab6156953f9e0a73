# same-box A/B of an environment switch on the working tree: scripts/ab_env2.sh VAR A B
VAR=$1; A=$2; B=$3
for v in $A $B $A $B; do
  env $VAR=$v timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-traffic --no-other-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$VAR=$v', d['value'], d['ms_per_step'], d.get('launches_per_step'), r.get('conv_ms_per_step'))"
done
