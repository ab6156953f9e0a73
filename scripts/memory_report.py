"""Memory accounting for the BASELINE.json configurations: peak device bytes (allocated by live
tensors / reserved by the caching allocator), bytes per input site and per site processed, and what the persistent
arenas hold.   python scripts/memory_report.py c1|c3|c4
  c1: configs[1], 32 blocks of 64^3, training step (classic eager, then GraphStep capacity mode + HIP graph)
  c3: configs[3], one (128,512,512) scene, forward only in eval mode (test_scene.py path)
  c4: configs[4], 8 blocks of 128^3 at 20 % iid occupancy, training step; then the SAME model steps on 1 block for
      40 steps: persistent arenas must shrink (scn.program.ARENA_SHRINK_AFTER)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgnn_amd import synth
from sgnn_amd.model import GenModel
from sgnn_amd.train import train_step, to_device, make_optimizer, GraphStep
from sgnn_amd.scn import program as P_

GB = 2.0 ** 30
which = sys.argv[1] if len(sys.argv) > 1 else 'c1'
lw = np.ones(5, dtype=np.float32)
torch.manual_seed(1234)


def arenas(model):
    fwd = sum(4 * t.numel() for p in P_.programs_of(model) for t in (p.__dict__.get('_arenas') or {}).values()
              if torch.is_tensor(t))
    d = P_.arena_bytes()
    return 'program arenas: forward %.2f GB, gradient %.2f GB, inference %.2f GB' % (fwd / GB, d['gradient'] / GB,
                                                                                   d['inference'] / GB)


def line(tag, n_in, n_proc, model, ms):
    peak, res = torch.cuda.max_memory_allocated(), torch.cuda.max_memory_reserved()
    print('%s: %.1f ms | peak allocated %.2f GB, reserved %.2f GB | %d input sites -> %.0f B per input site | %d sites '
          'over all levels -> %.0f B per site processed | %s' % (tag, ms, peak / GB, res / GB, n_in, peak / max(n_in, 1),
                                                                n_proc, peak / max(n_proc, 1), arenas(model)))


def processed(n_in, outs):
    return n_in + sum(int(o[0].shape[0]) if len(o[0]) else 0 for o in outs[1])


def run_steps(step, n, sync=True):
    """n steps; returns the last step's outputs and the median time of the last (up to) 5 steps in ms."""
    outs, ts = None, []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = step()
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    return outs, float(np.median(ts[-5:]))


if which == 'c1':
    P_.PERSISTENT_ARENAS = True
    batch = to_device(synth.make_batch(32, (64,) * 3, cfg=2), 'cuda')
    n_in = int(batch['input'][0].shape[0])
    m = GenModel(8, (64,) * 3, 1, 16, 16, 4, True, True, 1, 1).cuda()
    opt = make_optimizer(m.parameters(), lr=1e-3)
    torch.cuda.reset_peak_memory_stats()
    outs, ms = run_steps(lambda: train_step(m, opt, batch, lw), 6)
    line('configs[1] bs32 64^3 training, classic eager', n_in, processed(n_in, outs[2]), m, ms)
    del opt, outs          # (a loss kept from the eager steps keeps their AccumulateGrad nodes alive: capture would abort)
    gs = GraphStep(m, lr=1e-3, settle=False)
    torch.cuda.reset_peak_memory_stats()
    for _ in range(8):
        gs(batch, lw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        gs(batch, lw)
    torch.cuda.synchronize()
    ms = 1e2 * (time.perf_counter() - t0)
    cap = gs.capacity
    n_cap = cap.input_rows + sum(cap.enc) + sum(g[0] * 8 + sum(g[1]) for g in cap.gen) if hasattr(cap, 'gen') else 0
    line('configs[1] bs32 64^3 training, capacity mode + HIP graph (stats %s)' % gs.stats, n_in, n_cap, m, ms)
elif which == 'c4':
    P_.PERSISTENT_ARENAS = True
    B, D = 8, 128
    batch = to_device(synth.make_batch(B, (D,) * 3, cfg=5, occupancy=0.2, dist='iid'), 'cuda')
    n_in = int(batch['input'][0].shape[0])
    m = GenModel(8, (D,) * 3, 1, 16, 16, 4, True, True, 1, 1).cuda()
    opt = make_optimizer(m.parameters())
    torch.cuda.reset_peak_memory_stats()
    outs, ms = run_steps(lambda: train_step(m, opt, batch, lw), 12)
    line('configs[4] bs8 128^3 @20 pct training, classic eager', n_in, processed(n_in, outs[2]), m, ms)
    small = to_device(synth.make_batch(1, (D,) * 3, cfg=5, occupancy=0.2, dist='iid'), 'cuda')
    del outs
    outs, ms = run_steps(lambda: train_step(m, opt, small, lw), P_.ARENA_SHRINK_AFTER + 8)
    torch.cuda.synchronize()
    print('after %d steps on ONE block: %s; allocated now %.2f GB' % (P_.ARENA_SHRINK_AFTER + 8, arenas(m),
                                                                     torch.cuda.memory_allocated() / GB))
else:
    dims = (128, 512, 512)
    scene = synth.make_scene(dims, cfg=4, occupancy=0.05)
    m = GenModel(8, (128, 128, 128), 1, 16, 16, 4, True, True, 1, 1).cuda()
    m.update_sizes(np.array(dims), np.array(dims) // 8)
    locs, feats = scene[0], scene[1].cuda()
    n_in = int(locs.shape[0])
    saved = []          # running statistics of this scene (see scripts/run_configs.py)
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
        torch.cuda.synchronize()
        print('after the statistics pass (training-mode kernels, no_grad): peak allocated %.2f GB'
              % (torch.cuda.max_memory_allocated() / GB))
        torch.cuda.reset_peak_memory_stats()
        for i in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            osdf, oocc = m([locs, feats], lw)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0)
        n_proc = n_in + sum(int(o[0].shape[0]) if len(o[0]) else 0 for o in oocc)
        line('configs[3] (128,512,512) scene forward, eval', n_in, n_proc, m, ms)
        print('   sites per level %s, final %d' % ([int(o[0].shape[0]) if len(o[0]) else 0 for o in oocc], len(osdf[0])))
