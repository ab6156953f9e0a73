"""Generates tests/golden/targets_expected.npz by running the REAL reference loss code (read-only /root/reference,
authoring container only): torch/loss.py compute_targets (:15-32), compute_weights_missing_geo (:35-48) and compute_loss
(:160-199) on seeded synthetic batches and seeded random sparse predictions.  Pins SURVEY.md §8 row f1 (the fused
target / loss kernels) to the reference itself instead of to this build's own tensor-op restatement.

Substitutions as in make_golden.py: `sparseconvnet` := the oracle (loss.py imports it, unused), plyfile /
marching_cubes_cpp := empty stubs, Tensor.cuda := identity (loss.py:41 hard-codes .cuda()).

Usage:  python tests/golden/make_golden_targets.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))

import scn_oracle  # noqa: E402

sys.modules['sparseconvnet'] = scn_oracle
sys.modules['plyfile'] = types.ModuleType('plyfile')
sys.modules['marching_cubes_cpp'] = types.ModuleType('marching_cubes_cpp')
torch.Tensor.cuda = lambda self, *a, **k: self
sys.path.insert(0, '/root/reference/torch')
import loss as ref_loss  # noqa: E402

from sgnn_amd import synth  # noqa: E402


def predictions(dims, batch, seed, n_per_level=(400, 900, 2500, 6000)):
    """Seeded random sparse 'predictions' per hierarchy level: unique sites, (occ logit, sdf) values."""
    g = torch.Generator().manual_seed(seed)
    outs = []
    for h, f in enumerate((8, 4, 2, 1)):
        d = [v // f for v in dims]
        total = batch * d[0] * d[1] * d[2]
        n = min(n_per_level[h], total)
        flat = torch.randperm(total, generator=g)[:n].sort().values
        b = flat // (d[0] * d[1] * d[2])
        r = flat % (d[0] * d[1] * d[2])
        locs = torch.stack([r // (d[1] * d[2]), (r // d[2]) % d[1], r % d[2], b], 1).long()
        outs.append([locs, torch.randn(n, 2, generator=g) * 2.0])
    sdf_vals = torch.randn(outs[3][0].shape[0], 1, generator=g) * 1.5
    return outs, [outs[3][0], sdf_vals]


def run(out, tag, dims, batch, cfg, masking, wgeo):
    data = synth.make_batch(batch, dims, cfg=cfg, occupancy=0.07)
    sdf, known, hier = data['sdf'].clone(), data['known'], [h.clone() for h in data['hierarchy']]
    locs = data['input'][0]
    tsdf, occs, hiers = ref_loss.compute_targets(sdf, hier, 4, 3, masking, known)
    out[tag + '_cfg'] = np.array([dims[0], dims[1], dims[2], batch, cfg, int(masking)])
    out[tag + '_wgeo'] = np.float64(wgeo)
    out[tag + '_tsdf'] = tsdf.numpy()
    for h in range(4):
        out['%s_occ%d' % (tag, h)] = occs[h].numpy().astype(np.int8)            # values in {-1, 0, 1}
        out['%s_hier%d' % (tag, h)] = hiers[h].numpy()
    if wgeo > 1:
        w = ref_loss.compute_weights_missing_geo(wgeo, locs, occs, 3)
        for h in range(4):
            out['%s_w%d' % (tag, h)] = w[h].numpy().astype(np.float32)
    occ_pred, sdf_pred = predictions(dims, batch, 100 + cfg)
    for v in [o[1] for o in occ_pred] + [sdf_pred[1]]:
        v.requires_grad_(True)
    lw = np.array([1.0, 0.5, 1.0, 1.0, 2.0], dtype=np.float32)
    loss, losses = ref_loss.compute_loss(sdf_pred, occ_pred, tsdf, occs, hiers, lw, 3, True, wgeo, locs, masking, known)
    loss.backward()
    out[tag + '_lw'] = lw
    out[tag + '_loss'] = np.float64(loss.item())
    out[tag + '_losses'] = np.array(losses, dtype=np.float64)
    for h in range(4):
        out['%s_pred%d_locs' % (tag, h)] = occ_pred[h][0].numpy()
        out['%s_pred%d_vals' % (tag, h)] = occ_pred[h][1].detach().numpy()
        out['%s_pred%d_grad' % (tag, h)] = occ_pred[h][1].grad.numpy()
    out[tag + '_sdf_vals'] = sdf_pred[1].detach().numpy()
    out[tag + '_sdf_grad'] = sdf_pred[1].grad.numpy()
    print(tag, dims, batch, 'loss %.6f' % loss.item(), losses)


if __name__ == '__main__':
    res = {}
    run(res, 'rect_mask_w5', (64, 32, 48), 3, 41, True, 5.0)
    run(res, 'rect_nomask_w1', (64, 32, 48), 2, 42, False, 1.0)
    run(res, 'cube64_mask_w5', (64, 64, 64), 2, 43, True, 5.0)
    path = os.path.join(HERE, 'targets_expected.npz')
    np.savez_compressed(path, **res)
    print(path, os.path.getsize(path) // 1024, 'KiB')
