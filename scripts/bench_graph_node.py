"""What does one dependent, near-empty kernel cost inside a replayed HIP graph — without a profiler attached?

HISTORY.md §4d argues from a same-box A/B (39 launches fewer, step unchanged) that the 4-5 us such kernels show in a
rocprofv3 trace are mostly the profiler's.  This measures the thing directly: a captured chain of N dependent launches of the
library's smallest kernel (sgnn_add over 256 floats, in place, so every launch depends on the one before), replayed R times,
wall clock / (R * N); the same chain issued eagerly on the stream for comparison.  Usage (GPU box):
    python scripts/bench_graph_node.py [N] [R]         -> one line per variant
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import _lib                      # noqa: E402
from sgnn_amd._lib import ptr                  # noqa: E402


def chain(y, b, n_launch, count):
    for _ in range(n_launch):
        _lib.call('sgnn_add', ptr(y), ptr(b), count, ptr(y))


def main():
    n_launch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    dev = torch.device('cuda', 0)
    _lib.require_gpu()
    for count in (256, 1 << 20):               # a near-empty kernel; one with 12 MB of traffic for scale
        y = torch.zeros(count, device=dev)
        b = torch.full((count,), 1e-6, device=dev)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            chain(y, b, 8, count)                  # warm-up (module load)
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(max(1, reps // 10)):
                chain(y, b, n_launch, count)
            s.synchronize()
            eager = (time.perf_counter() - t0) / (max(1, reps // 10) * n_launch)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                chain(y, b, n_launch, count)
            for _ in range(3):
                g.replay()
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                g.replay()
            s.synchronize()
            graph = (time.perf_counter() - t0) / (reps * n_launch)
        print('sgnn_add over %8d floats, chain of %d dependent launches: eager %.2f us per launch, graph replay %.2f us per node'
              % (count, n_launch, eager * 1e6, graph * 1e6))


if __name__ == '__main__':
    main()
