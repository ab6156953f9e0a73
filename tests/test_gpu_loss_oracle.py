"""SURVEY.md §8 row f1 anchored on the reference and the oracle: the fused target kernels
(sgnn_loss_targets) and the fused hierarchical loss (sgnn_loss_levels_fwd / _bwd) against
  * tests/golden/targets_expected.npz — outputs of the REFERENCE's torch/loss.py (compute_targets :15-32,
    compute_weights_missing_geo :35-48, compute_loss :160-199) on seeded batches and seeded sparse predictions, generated
    by tests/golden/make_golden_targets.py in the authoring container;
  * oracle/model_oracle.py's restatement of the same functions, run live on the same inputs.
Targets, occupancy pyramids and weights are bit-exact (clamps, comparisons, maxima); the loss value is held to 1e-6
relative, its gradient with respect to every prediction to 1e-6 of the gradient scale (the reference reduces in fp32,
the kernel in fp64 with a fixed order)."""
import os

import numpy as np
import pytest
import torch

import model_oracle as mo
from sgnn_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'targets_expected.npz')
CASES = ['rect_mask_w5', 'rect_nomask_w1', 'cube64_mask_w5']


def _load(tag):
    z = np.load(GOLD)
    d0, d1, d2, batch, cfg, masking = (int(v) for v in z[tag + '_cfg'])
    return z, (d0, d1, d2), batch, cfg, bool(masking), float(z[tag + '_wgeo'])


@pytest.mark.parametrize('tag', CASES)
def test_fused_targets_equal_reference_and_oracle(tag):
    from sgnn_amd import loss as L
    z, dims, batch, cfg, masking, wgeo = _load(tag)
    data = synth.make_batch(batch, dims, cfg=cfg, occupancy=0.07)
    sdf, hier, known = data['sdf'].cuda(), [h.cuda() for h in data['hierarchy']], data['known'].cuda()
    locs = data['input'][0]
    (ts, occs, hiers), w = L.compute_targets_and_weights(sdf, hier, 4, 3.0, masking, known, wgeo, locs.cuda())
    # the reference's outputs
    assert np.array_equal(ts.cpu().numpy(), z[tag + '_tsdf'])
    for h in range(4):
        assert np.array_equal(occs[h].cpu().numpy(), z['%s_occ%d' % (tag, h)].astype(np.float32)), h
        assert np.array_equal(hiers[h].cpu().numpy(), z['%s_hier%d' % (tag, h)]), h
        if wgeo > 1:
            assert np.array_equal(w[h].cpu().numpy(), z['%s_w%d' % (tag, h)]), h
    if wgeo <= 1:
        assert w is None
    # the oracle's restatement, live
    o_ts, o_occs, o_hiers = mo.compute_targets(data['sdf'].clone(), [h.clone() for h in data['hierarchy']], 4, 3, masking,
                                               data['known'])
    assert torch.equal(ts.cpu(), o_ts)
    for h in range(4):
        assert torch.equal(occs[h].cpu(), o_occs[h]) and torch.equal(hiers[h].cpu(), o_hiers[h])
    if wgeo > 1:
        o_w = mo.compute_weights_missing_geo(wgeo, locs, o_occs, 3)
        for h in range(4):
            assert torch.equal(w[h].cpu(), o_w[h])


@pytest.mark.parametrize('tag', CASES)
def test_fused_loss_equals_reference_and_oracle(tag):
    from sgnn_amd import loss as L
    z, dims, batch, cfg, masking, wgeo = _load(tag)
    data = synth.make_batch(batch, dims, cfg=cfg, occupancy=0.07)
    sdf, hier, known = data['sdf'].cuda(), [h.cuda() for h in data['hierarchy']], data['known'].cuda()
    locs = data['input'][0]
    lw = z[tag + '_lw']
    (ts, occs, hiers), w = L.compute_targets_and_weights(sdf, hier, 4, 3.0, masking, known, wgeo, locs.cuda())
    vals = [torch.from_numpy(z['%s_pred%d_vals' % (tag, h)]).cuda().requires_grad_(True) for h in range(4)]
    plocs = [torch.from_numpy(z['%s_pred%d_locs' % (tag, h)]).cuda() for h in range(4)]
    sv = torch.from_numpy(z[tag + '_sdf_vals']).cuda().requires_grad_(True)
    assert L.FUSED
    loss, losses = L.compute_loss([plocs[3], sv], [[plocs[h], vals[h]] for h in range(4)], ts, occs, hiers, lw, 3.0, True,
                                  wgeo, locs.cuda(), masking, known, weights=w)
    assert isinstance(loss, torch.Tensor) and loss.grad_fn is not None and 'TotalLoss' in type(loss.grad_fn).__name__
    loss.backward()
    want = float(z[tag + '_loss'])
    assert abs(loss.item() - want) <= 1e-6 * abs(want), (loss.item(), want)
    assert np.allclose([float(v) for v in losses], z[tag + '_losses'], rtol=2e-6, atol=0)
    for h in range(4):
        g, ref = vals[h].grad.cpu().numpy(), z['%s_pred%d_grad' % (tag, h)]
        assert np.abs(g - ref).max() <= 1e-6 * np.abs(ref).max() + 1e-12, (h, np.abs(g - ref).max(), np.abs(ref).max())
    g, ref = sv.grad.cpu().numpy(), z[tag + '_sdf_grad']
    assert np.abs(g - ref).max() <= 1e-6 * np.abs(ref).max() + 1e-12
    # the oracle's compute_loss on the same inputs, live (value)
    o_t = mo.compute_targets(data['sdf'].clone(), [h.clone() for h in data['hierarchy']], 4, 3, masking, data['known'])
    o_vals = [torch.from_numpy(z['%s_pred%d_vals' % (tag, h)]) for h in range(4)]
    o_locs = [torch.from_numpy(z['%s_pred%d_locs' % (tag, h)]) for h in range(4)]
    o_loss, _ = mo.compute_loss([o_locs[3], torch.from_numpy(z[tag + '_sdf_vals'])], [[o_locs[h], o_vals[h]] for h in range(4)],
                                o_t[0], o_t[1], o_t[2], lw, 3, True, wgeo, locs, masking, data['known'])
    assert abs(loss.item() - float(o_loss)) <= 2e-6 * abs(float(o_loss))
