"""Cost of a grid-wide barrier (atomic counter, agent-scope release / acquire, 1 KiB handed between workgroups per phase) for
32 .. 1024 resident workgroups, next to the cost of a dependent kernel node in a replayed graph (scripts/bench_graph_node.py:
1.7 us).  VERDICT r5 item 2 (one cooperative kernel per small-level residual block) stands or falls with this number.
  python scripts/bench_grid_barrier.py"""
import ctypes, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, 'scripts', 'kernels', 'libgrid_barrier.so'))
lib.grid_barrier_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
st = torch.cuda.current_stream().cuda_stream
err = torch.zeros(1, dtype=torch.int32, device='cuda')
for blocks in (32, 64, 128, 256, 512, 1024):
    buf = torch.zeros(2 * blocks * 256, device='cuda')
    res = []
    for phases in (200, 2200):
        counter = torch.zeros(1, dtype=torch.int32, device='cuda')
        lib.grid_barrier_run(buf.data_ptr(), counter.data_ptr(), blocks, 10, err.data_ptr(), st)      # warm-up
        torch.cuda.synchronize()
        counter.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lib.grid_barrier_run(buf.data_ptr(), counter.data_ptr(), blocks, phases, err.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        res.append((phases, e0.elapsed_time(e1) * 1e3))
    per = (res[1][1] - res[0][1]) / (res[1][0] - res[0][0])
    print('%5d workgroups of 256 threads: %.2f us per phase (row exchange + grid barrier)%s'
          % (blocks, per, '   BARRIER NEVER MET' if int(err[0]) else ''))
