#!/bin/bash
# complete GPU suite (no -x: list every failure), then the quick bench
TAG=$1
mkdir -p gpurun_out
timeout -k 10 1200 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc $?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/${TAG}_tests.log | head -40
bash scripts/gpu_r3.sh $TAG bench 40
