#!/bin/bash
# Round 5 (second session): parity of the dense k4/s2 layers by parity groups, then a same-box A/B of SGNN_DENSE_PARITY.
TAG=$1
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_dense_heads.py tests/test_gpu_model.py tests/test_gpu_graphstep_parity.py \
  tests/test_gpu_capacity.py $2 -x -q > gpurun_out/${TAG}_pytest.log 2>&1
echo "pytest rc $?"; tail -5 gpurun_out/${TAG}_pytest.log
bash scripts/ab_env2.sh SGNN_DENSE_PARITY 0 1 | tee gpurun_out/${TAG}_ab.txt
