#!/bin/bash
# run a test selection N times (race hunting): scripts/gpu_flaky.sh N tests...
N=$1; shift
for i in $(seq 1 $N); do
  timeout 600 python -m pytest "$@" -x -q 2>&1 | tail -1
done
