"""How fast do the free-running masks move while the bench trains fresh weights?  Prints, every 10 steps, the live row
counts of the last two generative levels and GraphStep's statistics (re-plans / captures)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench as B          # noqa: E402  (sets GPU_MAX_HW_QUEUES)
from sgnn_amd import synth
from sgnn_amd.model import GenModel
from sgnn_amd.train import GraphStep, to_device
from sgnn_amd.scn import program as P_
P_.PERSISTENT_ARENAS = True
torch.manual_seed(1234)
batches = [to_device(synth.make_batch(32, (64,) * 3, cfg=2 + j), 'cuda') for j in range(2)]
m = GenModel(8, (64,) * 3, 1, 16, 16, 4, True, True, 1, 1).cuda()
gs = GraphStep(m, lr=1e-3, headroom=float(sys.argv[1]) if len(sys.argv) > 1 else 1.3)
lw = np.ones(5, dtype=np.float32)
for i in range(400):
    gs(batches[i % 2], lw)
    if i % 10 == 9:
        torch.cuda.synchronize()
        live = gs.capacity.read()
        print(i + 1, [g[0] for g in live['gen']], dict(gs.stats, replay_host_ms=0))
