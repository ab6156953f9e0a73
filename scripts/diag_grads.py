"""Diagnostic: every parameter gradient of the HIP model vs the float64 oracle (needs GPU)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import model_oracle as mo
from util import param_fill
from sgnn_amd import synth, loss as L
from sgnn_amd.model import GenModel
dims, cfg, B = (32, 32, 32), 21, 3
if len(sys.argv) > 1: cfg = int(sys.argv[1])
data = synth.make_batch(B, dims, cfg=cfg, occupancy=0.07)
locs, feats = data['input']
lw = np.ones(5, dtype=np.float32)
om = param_fill(mo.GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1), cfg).train().double()
t = mo.compute_targets(data['sdf'].clone().double(), [h.clone().double() for h in data['hierarchy']], 4, 3, True, data['known'])
osdf, oocc = om([locs, feats.double()], lw)
ol, _ = mo.compute_loss(osdf, oocc, t[0], t[1], t[2], lw, 3, True, 5.0, locs, True, data['known'])
ol.backward()
hm = param_fill(GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1), cfg).train().cuda()
th = L.compute_targets(data['sdf'].clone().cuda(), [h.clone().cuda() for h in data['hierarchy']], 4, 3, True, data['known'].cuda())
hsdf, hocc = hm([locs.cuda(), feats.cuda()], lw)
hl, _ = L.compute_loss(hsdf, hocc, th[0], th[1], th[2], lw, 3, True, 5.0, locs.cuda(), True, data['known'].cuda())
hl.backward()
print('loss', ol.item(), hl.item())
hp = dict(hm.named_parameters())
rows = []
for n, p in om.named_parameters():
    g = hp[n].grad.cpu().double()
    rows.append(((p.grad - g).abs().max().item() / max(1e-12, p.grad.abs().max().item()), n, p.grad.abs().max().item()))
for r in sorted(rows, reverse=True)[:25]:
    print('%.3e  %-55s |g|max %.3e' % r)
print('median rel err %.3e' % np.median([r[0] for r in rows]))
