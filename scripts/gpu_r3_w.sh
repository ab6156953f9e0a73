#!/bin/bash
TAG=$1
mkdir -p gpurun_out
for q in 4 8 2 16; do
  SGNN_BENCH_HW_QUEUES=$q timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic > gpurun_out/${TAG}_q${q}.json 2> gpurun_out/${TAG}_q${q}.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_q${q}.json').read().strip().splitlines()[-1])
    o=d['other_legs']
    print('GPU_MAX_HW_QUEUES ${q}: graph %.3f ms | classic free %.3f | classic tf+prefetch %.3f | graph tf %.3f | batch1 %.3f' % (d['ms_per_step'], o['classic_eager_free_running']['ms_per_step'], o['classic_eager_teacher_forced_prefetch']['ms_per_step'], o['graph_teacher_forced']['ms_per_step'], o['batch1']['ms_per_step']))
except Exception as e:
    print('q ${q} failed', e); print(open('gpurun_out/${TAG}_q${q}.err').read()[-1500:])
PY
done
