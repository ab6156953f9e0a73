python -m pytest tests/test_gpu_ops.py tests/test_gpu_fused.py tests/test_gpu_program.py -x -q 2>&1 | tail -2
python scripts/bench_conv_ab.py --batch 1 --iters 200 2>&1 | grep "16,16\|8,8\|12,12" | grep "looped"
bash scripts/ab3.sh 2>&1 | grep -v "^mid"
