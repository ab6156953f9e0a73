"""Diagnostic (needs GPU): cProfile of the host side of training steps at a tiny batch (host-bound)."""
import os, sys, cProfile, pstats
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgnn_amd import synth
from sgnn_amd.model import GenModel
from sgnn_amd.train import train_step, to_device, make_optimizer
torch.manual_seed(1234)
m = GenModel(8, (64,) * 3, 1, 16, 16, 4, True, True, 1, 1).cuda()
opt = make_optimizer(m.parameters(), lr=1e-3)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 2
batch = to_device(synth.make_batch(NB, (64,) * 3, cfg=2), 'cuda')
lw = np.ones(5, dtype=np.float32)
TF = True
for _ in range(5): train_step(m, opt, batch, lw, teacher_forced=TF)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): train_step(m, opt, batch, lw, teacher_forced=TF)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(60)
st.sort_stats('tottime').print_stats(30)
