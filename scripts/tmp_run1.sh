python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused.py tests/test_gpu_program.py tests/test_gpu_capacity.py -x -q 2>&1 | tail -2
for t in 1 0; do echo "== dw_c1 $t"; SGNN_TUNE=sgnn_conv_set_dw_c1=$t python scripts/bench_conv.py --cases 1x8 --iters 50 2>&1 | grep "conv_dw"; done
bash scripts/ab_env2.sh SGNN_TUNE sgnn_conv_set_dw_c1=1 sgnn_conv_set_dw_c1=0
