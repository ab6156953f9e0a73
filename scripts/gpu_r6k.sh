timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused.py tests/test_gpu_program.py tests/test_gpu_graphstep_parity.py tests/test_gpu_capacity.py -x -q > gpurun_out/r06k_tests.log 2>&1; tail -3 gpurun_out/r06k_tests.log
for round in 1 2; do for v in default bnold; do
  if [ "$v" = "default" ]; then unset SGNN_LIB; else export SGNN_LIB=$(pwd)/sgnn_amd/lib/variants/libsgnn_hip_$v.so; fi
  timeout -k 10 200 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-traffic --no-other-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], d.get('launches_per_step'))"
done; done
