# same-box A/B of an environment switch: scripts/ab_env.sh VAR VALUE_A VALUE_B  (two alternating runs each)
VAR=$1; A=$2; B=$3
for v in $A $B $A $B; do
  env $VAR=$v timeout -k 10 150 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$VAR=$v', d['value'], d['ms_per_step'], r.get('conv_ms_per_step'))"
done
