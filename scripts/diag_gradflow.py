"""Diagnostic: gradient w.r.t. every leaf module's output, HIP vs float64 oracle (needs GPU)."""
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
data = synth.make_batch(B, dims, cfg=cfg, occupancy=0.07)
locs, feats = data['input']
lw = np.ones(5, dtype=np.float32)

def feats_of(o):
    if torch.is_tensor(o): return o
    if hasattr(o, 'features'): return o.features
    return None

def instrument(model, rec, order):
    for n, mod in model.named_modules():
        if len(list(mod.children())) == 0:
            def hk(m, i, o, n=n):
                f = feats_of(o)
                if f is not None and f.requires_grad:
                    order.append(n)
                    f.register_hook(lambda g, n=n: rec.setdefault(n, []).append(g.detach().double().cpu()))
            mod.register_forward_hook(hk)

om = param_fill(mo.GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1), cfg).train().double()
ro, oo = {}, []
instrument(om, ro, oo)
t = mo.compute_targets(data['sdf'].clone().double(), [h.clone().double() for h in data['hierarchy']], 4, 3, True, data['known'])
osdf, oocc = om([locs, feats.double()], lw)
ol, _ = mo.compute_loss(osdf, oocc, t[0], t[1], t[2], lw, 3, True, 5.0, locs, True, data['known'])
ol.backward()
hm = param_fill(GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1), cfg).train().cuda()
rh, oh = {}, []
instrument(hm, rh, oh)
th = L.compute_targets(data['sdf'].clone().cuda(), [h.clone().cuda() for h in data['hierarchy']], 4, 3, True, data['known'].cuda())
hsdf, hocc = hm([locs.cuda(), feats.cuda()], lw)
hl, _ = L.compute_loss(hsdf, hocc, th[0], th[1], th[2], lw, 3, True, 5.0, locs.cuda(), True, data['known'].cuda())
hl.backward()
for n in reversed(oo):
    if n in rh and n in ro:
        a, b = sum(ro[n]), sum(rh[n])
        if a.shape != b.shape: print(n, 'shape', a.shape, b.shape); continue
        print('%-50s rel %.3e  |g|max %.3e  calls %d/%d' % (n, (a - b).abs().max() / max(1e-30, a.abs().max()), a.abs().max(), len(ro[n]), len(rh[n])))
