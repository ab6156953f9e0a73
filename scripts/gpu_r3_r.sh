#!/bin/bash
TAG=$1
mkdir -p gpurun_out
for pr in 0 -1 0 -1; do
  SGNN_CAPTURE_PRIORITY=$pr timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic --no-other-mode > gpurun_out/${TAG}_cprio${pr}.json 2> gpurun_out/${TAG}_cprio${pr}.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_cprio${pr}.json').read().strip().splitlines()[-1])
    print('capture priority ${pr}: graph %.3f ms' % d['ms_per_step'])
except Exception as e:
    print('cprio ${pr} failed', e); print(open('gpurun_out/${TAG}_cprio${pr}.err').read()[-800:])
PY
done
