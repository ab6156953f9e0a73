#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3 4; do
  timeout 600 python -m pytest tests/test_gpu_distributed.py -x -q > gpurun_out/dp_$i.log 2>&1
  echo "run $i rc $?"; tail -2 gpurun_out/dp_$i.log
  grep -n "^E  " gpurun_out/dp_$i.log | head -30
done
