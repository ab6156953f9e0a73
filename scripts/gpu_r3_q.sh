#!/bin/bash
# A/B: priority of the weight-gradient lane (same box)
TAG=$1
mkdir -p gpurun_out
python - <<'PY'
import torch
print('stream priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')
for p in (-2,-1,0,1,2):
    s=torch.cuda.Stream(priority=p); print(p, '->', s.priority)
PY
for pr in 0 1 -1 0 1; do
  SGNN_SIDE_PRIORITY=$pr timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic > gpurun_out/${TAG}_prio${pr}.json 2> gpurun_out/${TAG}_prio${pr}.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_prio${pr}.json').read().strip().splitlines()[-1])
    o=d['other_legs']
    print('side priority ${pr}: graph %.3f ms | classic free %.3f | classic tf+prefetch %.3f | graph tf %.3f | batch1 %.3f' % (d['ms_per_step'], o['classic_eager_free_running']['ms_per_step'], o['classic_eager_teacher_forced_prefetch']['ms_per_step'], o['graph_teacher_forced']['ms_per_step'], o['batch1']['ms_per_step']))
except Exception as e:
    print('prio ${pr} failed', e); print(open('gpurun_out/${TAG}_prio${pr}.err').read()[-500:])
PY
done
