# same-box A/B/C: ab_head/ (a `git archive <rev>` copy built in place), ab_mid/ (another one), the working tree; two rounds
for v in head mid work head mid work; do
  case $v in head) B=ab_head/bench.py;; mid) B=ab_mid/bench.py;; *) B=bench.py;; esac
  [ -f $B ] || continue
  timeout -k 10 200 python $B --steps 60 --warmup 20 --no-cpu-baseline --no-traffic --no-other-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], d.get('launches_per_step'), r.get('conv_ms_per_step'), [(l['mean_rows'], l['avg_us']) for l in r['by_level_size'][:3]])"
done
