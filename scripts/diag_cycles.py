"""Does a classic training step leave reference cycles that hold device memory until the cyclic collector runs?
Runs 6 steps with the collector disabled, collecting by hand after each: prints unreachable-object counts by type and
the device bytes the collection released."""
import collections
import gc
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
from sgnn_amd import model as M, synth
from sgnn_amd.train import train_step, to_device, make_optimizer
from util import param_fill

dims, cfg = (32, 32, 32), 17
batches = [to_device(synth.make_batch(2, dims, cfg=cfg + j, occupancy=0.08), 'cuda') for j in range(2)]
lw = np.ones(5, dtype=np.float32)
m = param_fill(M.GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1), cfg).train().cuda()
opt = make_optimizer(m.parameters(), lr=1e-3)
gc.collect()
gc.disable()
for i in range(6):
    loss, _, _ = train_step(m, opt, batches[i % 2], lw, teacher_forced=True)
    del loss
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    gc.set_debug(gc.DEBUG_SAVEALL)
    n = gc.collect()
    kinds = collections.Counter(type(o).__name__ for o in gc.garbage)
    tens = [o for o in gc.garbage if torch.is_tensor(o)]
    shapes = collections.Counter((tuple(t.shape), str(t.dtype)) for t in tens if t.is_cuda)
    del gc.garbage[:]
    del tens
    gc.set_debug(0)
    gc.collect()
    after = torch.cuda.memory_allocated()
    print('step %d: %d unreachable objects, device bytes released %d; kinds %s' % (i, n, before - after, kinds.most_common(8)))
    if shapes:
        print('   cuda tensors in cycles:', shapes.most_common(8))
