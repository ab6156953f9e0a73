#!/bin/bash
# Round 5 (second session): targeted parity tests + a same-box A/B of one environment switch.
#   scripts/gpu_r5e.sh TAG "test files" VAR A B
TAG=$1; FILES=$2; VAR=$3; A=$4; B=$5
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest $FILES -x -q > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/${TAG}_pytest.log
[ -n "$VAR" ] && bash scripts/ab_env2.sh $VAR $A $B | tee gpurun_out/${TAG}_ab.txt
