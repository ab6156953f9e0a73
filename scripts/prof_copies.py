"""Where do the step's device copies / fills come from?  torch.profiler with stacks over 3 steps; prints the python
call sites of aten::copy_/clone/fill_/zero_ sorted by count."""
import collections
import os
import sys

import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgnn_amd import synth  # noqa: E402
from sgnn_amd.model import GenModel  # noqa: E402
from sgnn_amd.train import make_optimizer, to_device, train_step  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
model = GenModel(8, (64,) * 3, 1, 16, 16, 4, True, True, 1, 1).to(dev)
opt = make_optimizer(model.parameters())
lw = np.ones(5, dtype=np.float32)
batches = [to_device(synth.make_batch(32, (64,) * 3, cfg=2, first_block=32 * j), dev) for j in range(2)]
for i in range(6):
    train_step(model, opt, batches[i % 2], lw)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    for i in range(3):
        train_step(model, opt, batches[i % 2], lw)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ('aten::copy_', 'aten::clone', 'aten::fill_', 'aten::zero_', 'aten::contiguous', 'aten::cat',
                   'aten::to', 'aten::_to_copy', 'aten::index', 'aten::add_', 'aten::mul'):
        sites[(ev.name, str(ev.input_shapes)[:90])] += 1
for (name, shp), n in sites.most_common(60):
    print('%5.1f/step  %-16s %s' % (n / 3.0, name, shp))
