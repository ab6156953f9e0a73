import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from util import param_fill
from sgnn_amd import synth, _lib, loss as L
from sgnn_amd.model import GenModel
lib = _lib.load()
def run(fused, small=1):
    _lib.tune('prog_fusion', int(fused)); _lib.tune('conv_small', small)
    dims, cfg = (32, 32, 32), 17
    data = synth.make_batch(2, dims, cfg=cfg, occupancy=0.08)
    m = param_fill(GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1), cfg).train(True).cuda()
    lw = np.ones(5, dtype=np.float32)
    t = L.compute_targets(data['sdf'].clone().cuda(), [h.clone().cuda() for h in data['hierarchy']], 4, 3, True, data['known'].cuda())
    osdf, oocc = m([data['input'][0].cuda(), data['input'][1].cuda()], lw)
    loss, _ = L.compute_loss(osdf, oocc, t[0], t[1], t[2], lw, 3, True, 5.0, data['input'][0].cuda(), True, data['known'].cuda())
    loss.backward()
    return dict((n, p.grad.double().cpu()) for n, p in m.named_parameters()), loss.item()
a, la = run(True); b, lb = run(False); c, lc = run(True, 0); d, ld = run(False, 0)
print('loss', la, lb, lc, ld)
rows = []
for n in a:
    s = max(1.0, b[n].abs().max().item())
    rows.append(((a[n]-b[n]).abs().max().item()/s, (c[n]-d[n]).abs().max().item()/s, (b[n]-d[n]).abs().max().item()/s, n))
rows.sort(reverse=True)
print('rel diff: fused-vs-unfused(small) | fused-vs-unfused(old kernel) | small-vs-old (unfused)')
for r in rows[:12]: print('%.2e %.2e %.2e %s' % r)
print('median', np.median([r[0] for r in rows]), np.median([r[1] for r in rows]), np.median([r[2] for r in rows]))
