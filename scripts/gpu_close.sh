#!/bin/bash
# closing run of a round: evidence (scripts/gpu_final.sh TAG) + the full GPU test suite
TAG=$1
bash scripts/gpu_final.sh $TAG > gpurun_out/${TAG}_final.log 2>&1
tail -40 gpurun_out/${TAG}_final.log
timeout -k 10 1500 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/${TAG}_pytest_gpu.log
