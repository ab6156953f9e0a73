#!/bin/bash
TAG=$1
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_capacity.py tests/test_gpu_inference_layout.py tests/test_gpu_distributed.py -x -q > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc $?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/${TAG}_tests.log | head -20
for sp in 1 0 1; do
  SGNN_SIDE_PYRAMID=$sp timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic > gpurun_out/${TAG}_pyr${sp}.json 2> gpurun_out/${TAG}_pyr${sp}.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_pyr${sp}.json').read().strip().splitlines()[-1])
    o=d['other_legs']
    print('side pyramid ${sp}: graph %.3f ms | graph tf %.3f | batch1 %.3f | launches %s' % (d['ms_per_step'], o['graph_teacher_forced']['ms_per_step'], o['batch1']['ms_per_step'], d['launches_per_step']))
except Exception as e:
    print('pyr ${sp} failed', e); print(open('gpurun_out/${TAG}_pyr${sp}.err').read()[-1500:])
PY
done
