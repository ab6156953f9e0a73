#!/bin/bash
TAG=$1
mkdir -p gpurun_out
for cfg in "--steps 20 --warmup 5" "--steps 20 --warmup 10" "--steps 100 --warmup 50" "--steps 30 --warmup 10"; do
  timeout -k 10 300 python bench.py $cfg --no-cpu-baseline --no-traffic --no-other-mode > gpurun_out/${TAG}_b.json 2> gpurun_out/${TAG}_b.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_b.json').read().strip().splitlines()[-1])
    g=d['config']['graph']
    print('$cfg: %.3f ms | sites %s | %s' % (d['ms_per_step'], d['config']['generated_sites_per_level'], {k:g[k] for k in g if k not in ('capacity','live_rows')}))
except Exception as e:
    print('failed', e); print(open('gpurun_out/${TAG}_b.err').read()[-1500:])
PY
done
timeout -k 10 300 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_capacity.py -x -q 2>&1 | tail -3
