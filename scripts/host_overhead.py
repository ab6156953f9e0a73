"""Diagnostic (needs GPU): how much of a training step is host-side (Python/launch) time?
Measures wall per step, time blocked in device->host count reads, and the step's enqueue time when the
GPU is made artificially idle (sync before each step) vs. pure kernel time from HIP events."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgnn_amd import synth
from sgnn_amd.model import GenModel
from sgnn_amd.train import train_step, to_device, make_optimizer
from sgnn_amd.scn import metadata as MD

wait = [0.0]
orig = MD._Runtime.read_count
def timed(self):
    t = time.perf_counter(); r = orig(self); wait[0] += time.perf_counter() - t; return r
MD._Runtime.read_count = timed

torch.manual_seed(1234)
m = GenModel(8, (64,) * 3, 1, 16, 16, 4, True, True, 1, 1).cuda()
opt = make_optimizer(m.parameters(), lr=1e-3)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
batch = to_device(synth.make_batch(B, (64,) * 3, cfg=2), 'cuda')
lw = np.ones(5, dtype=np.float32)
for _ in range(3): train_step(m, opt, batch, lw)
torch.cuda.synchronize()
for trial in range(2):
    wait[0] = 0.0
    t0 = time.perf_counter()
    n = 5
    for _ in range(n): train_step(m, opt, batch, lw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('B=%d  wall/step %.2f ms | host returns after %.2f ms/step | blocked in count reads %.2f ms/step | host busy (not blocked) %.2f ms/step'
          % (B, 1e3 * (t2 - t0) / n, 1e3 * (t1 - t0) / n, 1e3 * wait[0] / n, 1e3 * (t1 - t0 - wait[0]) / n))
