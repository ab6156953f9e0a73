#!/bin/bash
# Hunt the two-ranks-on-one-device crash (tests/test_gpu_distributed.py::test_bench_gpus_2_starts_two_ranks): run the test's
# command up to N times, keep stdout / stderr of every failing run.   scripts/flake_hunt.sh N [ENV=VALUE ...]
N=${1:-20}; shift
mkdir -p gpurun_out/flake
fails=0
for i in $(seq 1 $N); do
  env SGNN_BENCH_SHARE_GPU=1 "$@" timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 2 --batch 2 --dim 32 --no-cpu-baseline \
      > gpurun_out/flake/run_$i.out 2> gpurun_out/flake/run_$i.err
  rc=$?
  if [ $rc -ne 0 ]; then
    fails=$((fails+1)); echo "run $i: rc=$rc"; tail -c 3000 gpurun_out/flake/run_$i.err > gpurun_out/flake/fail_$i.txt
  else
    rm -f gpurun_out/flake/run_$i.out gpurun_out/flake/run_$i.err
  fi
done
echo "flake hunt: $fails failures in $N runs ($*)"
