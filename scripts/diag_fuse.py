import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from util import param_fill
from sgnn_amd import synth, _lib
from sgnn_amd.model import GenModel
from sgnn_amd import loss as L
from sgnn_amd.scn import program as P
lib = _lib.load()
def run(enabled, fused):
    P.ENABLED = enabled
    _lib.tune('prog_fusion', int(fused))
    dims, cfg = (32, 32, 32), 17
    data = synth.make_batch(2, dims, cfg=cfg, occupancy=0.08)
    m = param_fill(GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1), cfg).train(True).cuda()
    lw = np.ones(5, dtype=np.float32)
    osdf, oocc = m([data['input'][0].cuda(), data['input'][1].cuda()], lw)
    return [o[1].detach().double().cpu() for o in oocc] + [osdf[1].detach().double().cpu()]
a = run(True, True); b = run(True, False); c = run(False, False); a2 = run(True, True)
for h in range(5):
    print(h, 'fused-vs-unfusedprog %.3g  unfusedprog-vs-layer %.3g  fused-vs-fused %.3g  scale %.3g' % (
        (a[h]-b[h]).abs().max(), (b[h]-c[h]).abs().max(), (a[h]-a2[h]).abs().max(), c[h].abs().max()))
