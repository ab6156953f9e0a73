# usage: mkdir ab_head && git archive HEAD | tar -x -C ab_head && make -C ab_head/sgnn_amd/csrc, then run this on the GPU box
# same-box A/B: the committed tree (ab_head/, a `git archive HEAD` copy built in place) against the working tree
for v in head work head work; do
  if [ $v = head ]; then B=ab_head/bench.py; else B=bench.py; fi
  timeout -k 10 150 python $B --steps 40 --warmup 10 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], r.get('conv_ms_per_step'), [(l['mean_rows'], l['avg_us']) for l in r['by_level_size'][:3]])"
done
