#!/bin/bash
# same-box A/B of one environment switch on the quick bench: scripts/gpu_ab_env.sh TAG VAR "v1 v2 v1 v2" [extra bench flags]
TAG=$1; VAR=$2; VALS=$3; shift 3
mkdir -p gpurun_out
for v in $VALS; do
  env $VAR=$v timeout -k 10 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic "$@" > gpurun_out/${TAG}_$v.json 2> gpurun_out/${TAG}_$v.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_$v.json').read().strip().splitlines()[-1])
    o=d.get('other_legs') or {}
    print('$VAR=$v: %.3f ms | sites %s | legs %s' % (d['ms_per_step'], d['config']['generated_sites_per_level'], {k:x.get('ms_per_step') for k,x in o.items()}))
except Exception as e:
    print('$VAR=$v failed', e); print(open('gpurun_out/${TAG}_$v.err').read()[-1500:])
PY
done
