"""Un-profiled timing of the lane forks and joins of the replayed step ("which chain kernel waits for
the lane").  Events cannot be recorded inside a replayed graph and a kernel trace changes the overlap it is asked about, so
the library writes the device's 100 MHz clock from one-thread kernels captured at the points of interest (sgnn_stamp):

  fork        training stream, right before a hierarchy's pyramid lane is forked (scn/metadata.py PendingChain)
  lane< lane> first / last thing on the pyramid lane (stride-2 chain + the coarse 3x3x3 rulebooks)
  prog< prog> around sgnn_prog_forward of the network that consumes the hierarchy
  down-wait<  training stream inside the program, in front of the first Convolution(2,2): the lane join
  down-wait>  the first thing after the join
  bwd< bwd>   around sgnn_prog_backward (the weight-gradient lane forks inside)
  join< lane-end join>   the single deferred join of the weight-gradient lane in front of Adam

    python scripts/lane_stamps.py [--steps 30]     ->  one table per step phase, medians over the replays, microseconds
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import _lib, synth                      # noqa: E402
from sgnn_amd.model import GenModel                   # noqa: E402
from sgnn_amd.train import GraphStep, to_device       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=30)
ap.add_argument('--settle', type=int, default=250)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--no-stamps', action='store_true', help='the same loop without stamps: what the stamps cost')
ap.add_argument('--group', type=int, default=1, help='replays issued back to back before each synchronise + read-out (the '
                'stamps read are the last replay\'s); 1 = a synchronise after every replay')
args = ap.parse_args()
dev = torch.device('cuda', 0)
lib = _lib.load()
MAXS = 256
buf = torch.zeros(MAXS, dtype=torch.int64, device=dev)
if not args.no_stamps:
    lib.sgnn_stamp_enable(buf.data_ptr(), MAXS)
    _lib.STAMPS = True
lw = np.ones(5, dtype=np.float32)
batches = [to_device(synth.make_batch(args.batch, (64,) * 3, cfg=2, first_block=j * args.batch, occupancy=0.05), dev)
           for j in range(2)]
torch.manual_seed(1234)
model = GenModel(8, (64,) * 3, 1, 16, 16, 4, True, True, 1, 1).to(dev)
gs = GraphStep(model, lr=1e-3, headroom=1.3)
for i in range(args.settle):
    gs(batches[i % 2], lw)
gs.headroom = 1.15
gs.replan()
for i in range(12):
    gs(batches[i % 2], lw)
assert gs.graphs is not None, 'the step was not captured'
torch.cuda.synchronize()
import time
rows = []
t0 = time.perf_counter()
for i in range(args.steps * args.group):
    gs(batches[i % 2], lw)
    if (i + 1) % args.group == 0:
        torch.cuda.synchronize()
        rows.append(buf.cpu().numpy().copy())
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / args.group
if args.no_stamps:
    print('no stamps: %.3f ms per step (%d replays between synchronises)' % (1e3 * el / args.steps, args.group))
    sys.exit(0)
n = lib.sgnn_stamp_count()
labels = [lib.sgnn_stamp_label(i).decode() for i in range(n)]
t = np.stack(rows)[:, :n].astype(np.float64) / 100.0         # 100 MHz -> microseconds
t = t - t.min(axis=1, keepdims=True)
med = np.median(t, axis=0)
print('# %d stamps per replayed step, medians over %d read-outs (%d replays back to back before each), microseconds from '
      'the first stamp' % (n, args.steps, args.group))
print('# slot  label        at_us')
for i, (l, m) in enumerate(zip(labels, med)):
    print('%5d  %-12s %9.1f' % (i, l, m))

# ---- forward: one record per hierarchy (fork ... down-wait>) ----------------------------------------------------------
print('\n# forward, per hierarchy: the lane (stride-2 chain + coarse rulebooks) against the training stream')
print('# %3s %10s %10s %10s %12s %12s %10s' % ('h', 'fork_at', 'lane_start', 'lane_us', 'train_work', 'join_wait', 'lane_slack'))
print('#     (lane_start = lane< - fork; lane_us = lane> - lane<; train_work = down-wait< - fork: what the training stream ran '
      'between the fork and the join;\n#      join_wait = down-wait> - down-wait<; lane_slack = down-wait< - lane>: > 0 the lane '
      'was done before the training stream asked)')
idx = {k: [i for i, l in enumerate(labels) if l == k] for k in set(labels)}
per = lambda a, b: np.median(t[:, b] - t[:, a])
forks = idx.get('fork', [])
tot_wait = 0.0
for h, f in enumerate(forks):
    nxt = forks[h + 1] if h + 1 < len(forks) else n
    seg = {l: i for i, l in enumerate(labels[f:nxt], start=f) if l in ('lane<', 'lane>', 'down-wait<', 'down-wait>')}
    if len(seg) < 4:
        continue
    w = per(seg['down-wait<'], seg['down-wait>'])
    tot_wait += w
    print('  %3d %10.1f %10.1f %10.1f %12.1f %12.1f %10.1f' % (
        h, med[f], per(f, seg['lane<']), per(seg['lane<'], seg['lane>']), per(f, seg['down-wait<']), w,
        per(seg['lane>'], seg['down-wait<'])))
print('# sum of join waits %.1f us per step' % tot_wait)

# ---- backward ------------------------------------------------------------------------------------------------------
if 'join<' in idx and 'join>' in idx:
    j0, j1 = idx['join<'][-1], idx['join>'][-1]
    print('\n# backward: deferred join of the weight-gradient lane in front of Adam')
    print('#   training stream reaches the join at %.1f us, the join costs the training stream %.1f us' % (med[j0], per(j0, j1)))
    if 'lane-end' in idx:
        print('#   the lane\'s last kernel is done at %.1f us' % med[idx['lane-end'][-1]])
    b0 = idx.get('bwd<', [])
    b1 = idx.get('bwd>', [])
    for k, (a, b) in enumerate(zip(b0, b1)):
        print('#   sgnn_prog_backward %d: %.1f us on the training stream' % (k, per(a, b)))
print('# whole step, first to last stamp: %.1f us; %.3f ms per step on the host clock (%d replays between synchronises)' % (
    np.median(t.max(axis=1)), 1e3 * el / args.steps, args.group))
