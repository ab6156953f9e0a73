# the data-parallel code path (two graphs + RCCL all-reduce of the flat gradient buffer between them) with ONE rank, at
# bench size, beside the plain single-GPU step; also dumps the two graphs' stream assignment (DEBUG_HIP_GRAPH_DOT_PRINT)
mkdir -p gpurun_out/dot_dp
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-traffic --no-other-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single', d['value'], d['ms_per_step'], d.get('host_graph_launch_ms'))"
cd gpurun_out/dot_dp
SGNN_BENCH_FORCE_DIST=1 DEBUG_HIP_GRAPH_DOT_PRINT=1 HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 36123 ../../bench.py --gpus 1 --steps 40 --warmup 10 --no-cpu-baseline --no-traffic --no-other-mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dp-one-rank', d['value'], d['ms_per_step'], d.get('host_graph_launch_ms'), d['config']['collective'][:40])"
ls
