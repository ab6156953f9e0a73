"""Bisect helper for a crash inside hipStreamEndCapture (GraphStep._capture): python scripts/repro_capture.py KEEP LR READ
KEEP: GraphStep(keep_outputs=...), LR: learning rate, READ: 1 = read outputs / gradients between the calls like the parity test."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from util import param_fill
from sgnn_amd import synth
from sgnn_amd.model import GenModel
from sgnn_amd.train import GraphStep
keep, lr, read = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
dims, cfg = (32, 32, 32), 11
m = param_fill(GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1), seed=cfg).train().cuda()
d = synth.make_batch(2, dims, cfg=cfg, occupancy=0.08)
batch = {'input': [d['input'][0].cuda(), d['input'][1].cuda()], 'sdf': d['sdf'].cuda(), 'known': d['known'].cuda(),
         'hierarchy': [h.cuda() for h in d['hierarchy']]}
lw = np.ones(5, dtype=np.float32)
gs = GraphStep(m, lr=lr, settle=False, keep_outputs=bool(keep))
for it in range(4):
    loss = float(gs(batch, lw))
    torch.cuda.synchronize()
    if read:
        if keep:
            from sgnn_amd.scn.capacity import trim
            osdf, oocc = gs.outputs
            n = [int(trim(o[0]).shape[0]) for o in oocc]
        g = gs.opt.named_gradients(m)
    print('call', it, 'loss', loss, gs.stats['captures'], gs.stats['replays'], flush=True)
print('OK', sys.argv[1:])
