#!/bin/bash
TAG=$1
mkdir -p gpurun_out
for t in 1 0 1 0; do
  SGNN_SIDE_TARGETS=$t timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-traffic > gpurun_out/${TAG}_t${t}.json 2> gpurun_out/${TAG}_t${t}.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_t${t}.json').read().strip().splitlines()[-1])
    o=d['other_legs']
    print('side targets ${t}: graph %.3f ms | graph tf %.3f | batch1 %.3f' % (d['ms_per_step'], o['graph_teacher_forced']['ms_per_step'], o['batch1']['ms_per_step']))
except Exception as e:
    print('t ${t} failed', e); print(open('gpurun_out/${TAG}_t${t}.err').read()[-1500:])
PY
done
