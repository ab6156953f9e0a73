#!/bin/bash
TAG=$1
mkdir -p gpurun_out
bash scripts/gpu_r3_trace.sh $TAG
sed -n 12,75p gpurun_out/${TAG}_trace_summary.txt
timeout -k 10 300 python -u scripts/memory_report.py c1 2>&1 | grep -v "amdgpu.ids\|UserWarning\|run_backward" > gpurun_out/${TAG}_memory_c1.txt; cat gpurun_out/${TAG}_memory_c1.txt
