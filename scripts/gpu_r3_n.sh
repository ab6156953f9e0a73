#!/bin/bash
# inference layout + memory accounting + reference-cycle diagnosis
TAG=$1
mkdir -p gpurun_out
timeout -k 10 120 python scripts/diag_cycles.py > gpurun_out/${TAG}_cycles.txt 2>&1; echo "cycles rc $?"; cat gpurun_out/${TAG}_cycles.txt | tail -20
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py tests/test_gpu_program.py tests/test_gpu_edges.py -q -x > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc $?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/${TAG}_tests.log | head -20
for c in c1 c3 c4; do
  timeout -k 10 300 python -u scripts/memory_report.py $c 2>&1 | grep -v amdgpu.ids >> gpurun_out/${TAG}_memory.txt
  echo "memory $c rc $?"
done
cat gpurun_out/${TAG}_memory.txt
