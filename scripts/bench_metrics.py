"""Evaluation-metric cost at the configs[1] geometry: IoU of the finest level + target-surface
L1 on the device vs the reference-style numpy path (oracle/metrics_oracle.py).  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from sgnn_amd import loss as L, metrics, synth  # noqa: E402

B, D, TRUNC = 32, 64, 3.0
data = synth.make_batch(B, D, cfg=5, occupancy=0.05)
known = data['known'].cuda()
tgt_sdf, tgt_occs, _ = L.compute_targets(data['sdf'].cuda(), [h.cuda() for h in data['hierarchy']], 4, TRUNC, True, known)
rng = np.random.default_rng(3)
cand = np.argwhere(np.abs(data['sdf'][:, 0].numpy()) < 4.5)
locs = np.concatenate([cand[:, 1:], cand[:, :1]], 1).astype(np.int64)
logits = rng.normal(0.5, 2.0, (len(locs), 2)).astype(np.float32)
vals = rng.normal(0, 2, len(locs)).astype(np.float32)
dl, dg, dv = torch.from_numpy(locs).cuda(), torch.from_numpy(logits).cuda(), torch.from_numpy(vals).cuda()


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


t_iou = timed(lambda: metrics.iou_counts(dl, tgt_occs[3], True, logits=dg))
ws_out = torch.empty(3, dtype=torch.float64, device='cuda')
t_l1 = timed(lambda: metrics.compute_l1_tgtsurf_sparse_dense(dl, dv, tgt_sdf, TRUNC, True, known), n=20)
vol = B * D ** 3
iou_bytes = vol * 4 + len(locs) * (32 + 4 + 4)               # dense f32 target + rows (locs, logit, gathered target)
l1_bytes = vol * 5 + len(locs) * (32 + 4 + 5)                # dense f32 target + u8 known + rows

import metrics_oracle as mo  # noqa: E402  (cpu leg)
occ_u8 = tgt_occs[3].cpu().numpy().astype(np.int8).astype(np.uint8)
keep = 1.0 / (1.0 + np.exp(-logits[:, 0])) > 0.5
t0 = time.perf_counter()
pred = [locs[(locs[:, 3] == b) & keep][:, :3] for b in range(B)]
mo.compute_iou_sparse_dense(pred, occ_u8, True)
c_iou = time.perf_counter() - t0
t0 = time.perf_counter()
mo.compute_l1_tgtsurf_sparse_dense(locs, vals, tgt_sdf.cpu().numpy(), TRUNC, True, known.cpu().numpy())
c_l1 = time.perf_counter() - t0
print(json.dumps({'workload': 'configs[1] geometry: 32 x 64^3, %d predicted rows' % len(locs),
                  'iou_ms': round(t_iou, 4), 'iou_GBps': round(iou_bytes / t_iou / 1e6, 1),
                  'l1_tgtsurf_ms_incl_readback': round(t_l1, 4), 'l1_GBps': round(l1_bytes / t_l1 / 1e6, 1),
                  'cpu_baseline': {'iou_ms': round(c_iou * 1e3, 1), 'l1_tgtsurf_ms': round(c_l1 * 1e3, 1), 'cores': 1,
                                   'kind': 'port', 'sample': 'same inputs through oracle/metrics_oracle.py (numpy '
                                                             'intersect1d/union1d per sample, loss.py:84-120)'}}))
