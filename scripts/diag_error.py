"""Diagnostic (needs GPU + oracle): per-module output error of the HIP fp32 model and of the oracle fp32
model against the oracle evaluated in float64, on identical inputs/weights."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import model_oracle as mo
from util import param_fill
from sgnn_amd import synth
from sgnn_amd.model import GenModel

dims, cfg, B = (32, 32, 32), 21, 3
data = synth.make_batch(B, dims, cfg=cfg, occupancy=0.07)
locs, feats = data['input']
lw = np.ones(5, dtype=np.float32)

def feats_of(o):
    if torch.is_tensor(o):
        return o
    if hasattr(o, 'features'):
        return o.features
    return None

def run(model, x):
    rec = {}
    hooks = []
    for n, mod in model.named_modules():
        if len(list(mod.children())) == 0:
            def hk(m, i, o, n=n):
                f = feats_of(o)
                if f is not None:
                    rec.setdefault(n, []).append(f.detach().double().cpu())
            hooks.append(mod.register_forward_hook(hk))
    with torch.no_grad():
        model(x, lw)
    for h in hooks: h.remove()
    return rec

m64 = mo.GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1); param_fill(m64, seed=cfg); m64.train(); m64 = m64.double()
m32 = mo.GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1); param_fill(m32, seed=cfg); m32.train()
mh = GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1); param_fill(mh, seed=cfg); mh.train(); mh = mh.cuda()
r64 = run(m64, [locs, feats.double()])
r32 = run(m32, [locs, feats])
rh = run(mh, [locs.cuda(), feats.cuda()])
print('%-50s %10s %10s %10s %10s' % ('module', 'hip_max', 'hip_rms', 'cpu32_max', 'cpu32_rms'))
for n in r64:
    for i, t in enumerate(r64[n]):
        if n not in rh or i >= len(rh[n]) or rh[n][i].shape != t.shape:
            print(n, 'shape mismatch'); continue
        eh = (rh[n][i] - t).abs(); ec = (r32[n][i] - t).abs()
        print('%-50s %10.2e %10.2e %10.2e %10.2e   |x|max %.2e' % (n, eh.max(), eh.pow(2).mean().sqrt(), ec.max(), ec.pow(2).mean().sqrt(), t.abs().max()))
