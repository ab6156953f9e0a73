mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_program.py tests/test_gpu_configs.py -x -q > gpurun_out/r02i_tests.log 2>&1; tail -3 gpurun_out/r02i_tests.log
for v in -1 45000 -1 45000; do
SGNN_TILE_MIN_ROWS=$v timeout -k 10 150 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile_min_rows $v', d['value'], d['ms_per_step'], d['roofline'])"
done
