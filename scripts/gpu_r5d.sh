#!/bin/bash
# Round 5 (second session): the full GPU suite on the working tree, then a same-box A/B of one integer switch.
#   scripts/gpu_r5d.sh TAG prog_lin_add
TAG=$1; SW=$2
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/${TAG}_pytest_gpu.log
[ -n "$SW" ] && bash scripts/ab_env2.sh SGNN_TUNE $SW=0 $SW=1 | tee gpurun_out/${TAG}_ab.txt
