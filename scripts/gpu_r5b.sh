#!/bin/bash
# Round 5 (second session): parity of the geometry / glue launch fusions, then a same-box A/B of all of them on vs off.
# Usage: scripts/gpu_r5b.sh TAG ["extra pytest files"]
TAG=$1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_glue_fusions.py tests/test_gpu_capacity.py tests/test_gpu_chain.py tests/test_gpu_ops.py \
  tests/test_gpu_configs.py tests/test_gpu_dense_heads.py $2 -x -q > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/${TAG}_pytest.log
OFF="scan_inline=0,chain_merged=0"
for v in off on off on; do
  if [ $v = off ]; then export SGNN_TUNE=$OFF SGNN_FUSED_GLUE=0 SGNN_DENSE_PARITY=0; else unset SGNN_TUNE; export SGNN_FUSED_GLUE=1 SGNN_DENSE_PARITY=1; fi
  timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-traffic --no-other-mode 2>gpurun_out/${TAG}_bench_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('fusions $v', d['value'], d['ms_per_step'], d.get('launches_per_step'), d['config']['graph'].get('library_launches_per_step'), r.get('conv_ms_per_step'))"
done | tee gpurun_out/${TAG}_ab.txt
