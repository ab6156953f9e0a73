#!/bin/bash
# A/B of kernel-variant library builds (scripts/build_variant.sh): stand-alone conv timings + the in-step bench
# usage: scripts/gpu_variants.sh TAG name1 name2 ...   ("default" = the regular build)
TAG=$1; shift
mkdir -p gpurun_out
for v in "$@"; do
  if [ "$v" = "default" ]; then unset SGNN_LIB; else export SGNN_LIB=$(pwd)/sgnn_amd/lib/variants/libsgnn_hip_$v.so; fi
  echo "=== $v"
  RULEBOOK_ONLY= timeout 200 python scripts/bench_conv.py --iters 50 --cases ${CASES:-16x16,8x8,12x12,26x16,48x16} 2>&1 | grep "conv_"
  timeout -k 10 300 python bench.py --steps 40 --warmup 10 --settle ${SETTLE:-250} --no-cpu-baseline --no-traffic --no-other-mode > gpurun_out/${TAG}_$v.json 2> gpurun_out/${TAG}_$v.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_$v.json').read().strip().splitlines()[-1])
    print('in-step: %.3f ms | sites %s' % (d['ms_per_step'], d['config']['generated_sites_per_level']))
except Exception as e:
    print('failed', e); print(open('gpurun_out/${TAG}_$v.err').read()[-800:])
PY
done
