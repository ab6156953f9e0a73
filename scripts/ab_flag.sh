# same-box A/B of a bench flag: scripts/ab_flag.sh "--flag-a" "--flag-b"   (two alternating runs each; "" = no flag)
for v in "$1" "$2" "$1" "$2"; do
  timeout -k 10 150 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-traffic --no-other-mode $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('flags [$v]', d['value'], d['ms_per_step'], r.get('conv_ms_per_step'))"
done
