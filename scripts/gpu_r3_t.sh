#!/bin/bash
TAG=$1
mkdir -p gpurun_out
bash scripts/gpu_r3_trace.sh $TAG
sed -n 12,40p gpurun_out/${TAG}_trace_summary.txt
tail -22 gpurun_out/${TAG}_trace_summary.txt
