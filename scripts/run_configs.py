"""BASELINE.json configs 4 and 5 as functional + timing checks (needs GPU).
  C5: 8 blocks of 128^3 at 20 % iid occupancy, fwd+bwd+Adam.   C4: one (128,512,512) scene at ~5 %, forward only
  (test_scene.py path: eval mode, update_sizes, coords may stay on the host)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgnn_amd import synth
from sgnn_amd.model import GenModel
from sgnn_amd.train import train_step, to_device, make_optimizer

from sgnn_amd.scn import program as P_
P_.PERSISTENT_ARENAS = True          # grow-only arenas: no allocator stalls when the generated level sizes change
which = sys.argv[1] if len(sys.argv) > 1 else 'c5'
lw = np.ones(5, dtype=np.float32)
torch.manual_seed(1234)
if which == 'c5':
    B, D = 8, 128
    t0 = time.time()
    batch = to_device(synth.make_batch(B, (D,) * 3, cfg=5, occupancy=0.2, dist='iid'), 'cuda')
    print('C5 data: %d sites (%.1f s to generate)' % (batch['input'][0].shape[0], time.time() - t0))
    m = GenModel(8, (D,) * 3, 1, 16, 16, 4, True, True, 1, 1).cuda()
    opt = make_optimizer(m.parameters())
    for i in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, _, outs = train_step(m, opt, batch, lw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        lv = [int(o[0].shape[0]) if len(o[0]) else 0 for o in outs[1]]
        print('C5 step %d: %.1f ms  (%.1f blocks/s)  loss %.4f  sites/level %s  peak mem %.1f GB'
              % (i, 1e3 * dt, B / dt, loss.item(), lv, torch.cuda.max_memory_allocated() / 2**30))
else:
    dims = (128, 512, 512)
    t0 = time.time()
    scene = synth.make_scene(dims, cfg=4, occupancy=0.05)              # 128 surface tiles of 64^3
    print('C4 data: %d sites (%.1f s)' % (scene[0].shape[0], time.time() - t0))
    m = GenModel(8, (128, 128, 128), 1, 16, 16, 4, True, True, 1, 1).cuda()
    m.update_sizes(np.array(dims), np.array(dims) // 8)
    locs, feats = scene[0], scene[1].cuda()
    # random-init weights with the default running statistics (mean 0, var 1) predict empty levels in eval mode; one
    # training-mode pass with "replace" momentum sets the running statistics to this scene's batch statistics, which
    # makes the eval pass generate what a training step would (a trained checkpoint is not available offline)
    saved = []
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm3d):
            saved.append((mod, mod.momentum)); mod.momentum = 1.0
        elif hasattr(mod, 'running_mean') and hasattr(mod, 'momentum'):
            saved.append((mod, mod.momentum)); mod.momentum = 0.0
    with torch.no_grad():
        m.train()
        m([locs, feats], lw)
        for mod, mom in saved:
            mod.momentum = mom
        m.eval()
        for i in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            osdf, oocc = m([locs, feats], lw)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            lv = [int(o[0].shape[0]) if len(o[0]) else 0 for o in oocc]
            print('C4 pass %d: %.1f ms  sites/level %s  out %d  peak mem %.1f GB'
                  % (i, 1e3 * dt, lv, len(osdf[0]), torch.cuda.max_memory_allocated() / 2**30))
