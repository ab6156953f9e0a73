#!/bin/bash
# one-off: failing prefetch test verbosely, then the suites touched by the dense-bottleneck change, then bench
TAG=$1
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest "tests/test_gpu_program.py::test_prefetched_geometry_is_the_same_training_run" -x -q > gpurun_out/${TAG}_prefetch.log 2>&1
echo "prefetch rc $?"; grep -n "^E " gpurun_out/${TAG}_prefetch.log | head -20
timeout -k 10 900 python -m pytest tests/test_gpu_dense_heads.py tests/test_gpu_model.py tests/test_gpu_capacity.py tests/test_gpu_configs.py tests/test_gpu_program.py tests/test_gpu_distributed.py -q > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc $?"; tail -12 gpurun_out/${TAG}_tests.log
bash scripts/gpu_r3.sh $TAG bench 40
