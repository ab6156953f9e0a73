"""End-to-end parity of the HIP GenModel (sgnn_amd/model.py) against
  (1) golden vectors produced by the real reference model.py/loss.py (tests/golden), and
  (2) the CPU oracle model on fresh seeded inputs (forward + loss + every parameter gradient).
Site lists (active-site indices) must be bit-identical; logits within 1e-4 (north_star)."""
import os

import numpy as np
import pytest
import torch

import model_oracle as mo
from util import check_grads_vs_fp64_fixture, param_fill
from sgnn_amd import synth
from sgnn_amd.model import named_gradients

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
TOL = 1e-4


def hip_model(dims, cfg, train):
    from sgnn_amd.model import GenModel
    m = GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1)
    param_fill(m, seed=cfg)
    m.train(train)
    return m.cuda()


def _close(got, want, tol, what):
    """fp32 tolerance of the north star (1e-4) applied to the logit scale: max |err| <= tol * max(1, max|ref|)
    and rms err <= tol.  Measured context (scripts/diag_error.py, profiles/r01_error_growth.txt): with these
    weights activations reach |x| ~ 10 after ~60 stacked conv/BN layers, the reference's own fp32 evaluation
    sits 3e-4 (max) / 5e-5 (rms) from the float64 value, the HIP path about half of that."""
    d = np.abs(got.astype(np.float64) - want)
    scale = max(1.0, float(np.abs(want).max()))
    assert d.max() <= tol * scale, '%s: max err %g (scale %g)' % (what, d.max(), scale)
    assert np.sqrt((d ** 2).mean()) <= tol, '%s: rms err %g' % (what, np.sqrt((d ** 2).mean()))


def check_levels(oocc, osdf, g_occ, g_sdf, tol=TOL):
    """Site lists bit-exact; logits within the fp32 tolerance of the expected values."""
    for h in range(4):
        gl, gv = g_occ[h]
        if len(gl) == 0:
            assert len(oocc[h][0]) == 0, 'level %d should be empty' % h
            continue
        assert np.array_equal(oocc[h][0].cpu().numpy(), gl), 'level %d site list differs' % h
        _close(oocc[h][1].detach().cpu().numpy(), gv, tol, 'level %d logits' % h)
    gl, gv = g_sdf
    if len(gl):
        assert np.array_equal(osdf[0].cpu().numpy(), gl)
        _close(osdf[1].detach().cpu().numpy(), gv, tol, 'final sdf')
    else:
        assert len(osdf[0]) == 0


@pytest.mark.parametrize('name', ['genmodel_train_32', 'genmodel_train_rect', 'genmodel_train_empty',
                                  'genmodel_scene_eval'])
def test_hip_model_matches_reference_golden(name):
    from sgnn_amd import loss as L
    g = np.load(os.path.join(GOLD, name + '.npz'))
    dims = tuple(int(d) for d in g['dims'])
    scene = bool(g['scene_mode'])
    m = hip_model((32, 32, 32) if scene else dims, int(g['cfg']), bool(g['train']))
    data = synth.make_batch(int(g['batch']), dims, cfg=int(g['cfg']), occupancy=float(g['occupancy']))
    locs, feats = data['input']
    lw = np.ones(5, dtype=np.float32)
    # float64 evaluation of the reference code = exact value of its function: tolerance 1e-4 (north_star).
    # The reference's own fp32 run is up to ~1e-4 from that (see make_golden.py), so against it 2e-4.
    g_occ = [(g['occ%d_locs' % h], g['occ%d_vals64' % h]) for h in range(4)]
    g_sdf = (g['sdf_locs'], g['sdf_vals64'])
    g_occ32 = [(g['occ%d_locs' % h], g['occ%d_vals' % h]) for h in range(4)]
    g_sdf32 = (g['sdf_locs'], g['sdf_vals'])
    if scene:
        m.update_sizes(np.array(dims), np.array(dims) // 8)
        with torch.no_grad():
            osdf, oocc = m([locs, feats.cuda()], lw)   # coords may stay on the host (test_scene.py:80-82)
        check_levels(oocc, osdf, g_occ, g_sdf)
        check_levels(oocc, osdf, g_occ32, g_sdf32, 2 * TOL)
        return
    sdf, known = data['sdf'].cuda(), data['known'].cuda()
    hier = [h.cuda() for h in data['hierarchy']]
    t_sdf, t_occ, t_hier = L.compute_targets(sdf, hier, 4, 3, True, known)
    osdf, oocc = m([locs.cuda(), feats.cuda()], lw)
    check_levels(oocc, osdf, g_occ, g_sdf)
    check_levels(oocc, osdf, g_occ32, g_sdf32, 2 * TOL)
    loss, losses = L.compute_loss(osdf, oocc, t_sdf, t_occ, t_hier, lw, 3, True, float(g['weight_missing_geo']),
                                  locs.cuda(), True, known)
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) < 1e-4 * max(1.0, abs(float(g['loss'])))
    grads = named_gradients(m)            # reference layout (dense convolutions store (K, Cin, Cout))
    # round 5: against the reference's exact (fp64) gradients with its own fp32 spread as the yardstick
    check_grads_vs_fp64_fixture(g, grads, name)
    for k in g.files:
        if k.startswith('buf::'):
            assert np.abs(dict(m.named_buffers())[k[5:]].cpu().numpy() - g[k]).max() < 1e-5, k


def _oracle_run(data, dims, cfg, dtype):
    locs, feats = data['input']
    lw = np.ones(5, dtype=np.float32)
    om = param_fill(mo.GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1), cfg).train().to(dtype)
    t = mo.compute_targets(data['sdf'].clone().to(dtype), [h.clone().to(dtype) for h in data['hierarchy']], 4, 3, True,
                           data['known'])
    osdf, oocc = om([locs, feats.to(dtype)], lw)
    loss, _ = mo.compute_loss(osdf, oocc, t[0], t[1], t[2], lw, 3, True, 5.0, locs, True, data['known'])
    loss.backward()
    return om, osdf, oocc, loss.item()


def test_hip_model_vs_oracle_fresh_inputs_all_grads():
    """Every parameter gradient.  ReLU masks (and the loss's masks) make the gradient a discontinuous function
    of the activations, so two correct evaluations in different precision/summation order differ by ~1 % on
    parameter gradients (the oracle's own fp32 vs fp64 runs do: median 0.9 %, max 3 % here).  The bar is
    therefore relative: the HIP path must be as close to the float64 oracle as the oracle's own fp32 run is
    (factor 1.5 + small absolute slack), and the scalar loss must agree to 1e-5 relative."""
    from sgnn_amd import loss as L
    dims, cfg, B = (32, 32, 32), 21, 3
    data = synth.make_batch(B, dims, cfg=cfg, occupancy=0.07)
    locs, feats = data['input']
    lw = np.ones(5, dtype=np.float32)
    o64, osdf, oocc, l64 = _oracle_run(data, dims, cfg, torch.float64)
    o32, _, _, l32 = _oracle_run(data, dims, cfg, torch.float32)
    hm = hip_model(dims, cfg, True)
    th = L.compute_targets(data['sdf'].clone().cuda(), [h.clone().cuda() for h in data['hierarchy']], 4, 3, True,
                           data['known'].cuda())
    hsdf, hocc = hm([locs.cuda(), feats.cuda()], lw)
    hloss, _ = L.compute_loss(hsdf, hocc, th[0], th[1], th[2], lw, 3, True, 5.0, locs.cuda(), True,
                              data['known'].cuda())
    hloss.backward()
    check_levels(hocc, hsdf, [(o[0].numpy(), o[1].detach().numpy()) for o in oocc],
                 (osdf[0].numpy(), osdf[1].detach().numpy()))
    assert abs(hloss.item() - l64) < 1e-5 * max(1.0, abs(l64))
    hp, p32 = named_gradients(hm), dict(o32.named_parameters())
    e_hip, e_cpu = [], []
    for n, p in o64.named_parameters():
        assert p.grad is not None and hp[n] is not None, n
        scale = max(1e-12, p.grad.abs().max().item())
        e_hip.append((p.grad - hp[n].cpu().double()).abs().max().item() / scale)
        e_cpu.append((p.grad - p32[n].grad.double()).abs().max().item() / scale)
    e_hip, e_cpu = np.array(e_hip), np.array(e_cpu)
    assert np.median(e_hip) <= 1.5 * np.median(e_cpu) + 1e-4, (np.median(e_hip), np.median(e_cpu))
    assert e_hip.max() <= 2.0 * e_cpu.max() + 1e-3, (e_hip.max(), e_cpu.max())
    ob = dict(o64.named_buffers())
    for n, b in hm.named_buffers():
        if b.dtype.is_floating_point:
            assert (ob[n] - b.cpu().double()).abs().max().item() < 1e-4, n


def test_state_dict_interchangeable_with_oracle_layout():
    from sgnn_amd.model import GenModel
    hm = GenModel(8, (64, 64, 64), 1, 16, 16, 4, True, True, 1, 1)
    om = mo.GenModel(8, (64, 64, 64), 1, 16, 16, 4, True, True, 1, 1)
    assert list(hm.state_dict().keys()) == list(om.state_dict().keys())
    hm.load_state_dict(om.state_dict(), strict=True)
    assert sum(p.numel() for p in hm.parameters()) == 643735


def test_train_step_decreases_loss():
    from sgnn_amd.train import train_step, to_device
    torch.manual_seed(0)
    from sgnn_amd.model import GenModel
    m = GenModel(8, (32, 32, 32), 1, 16, 16, 4, True, True, 1, 1).cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    batch = to_device(synth.make_batch(2, (32, 32, 32), cfg=31, occupancy=0.08), 'cuda')
    lw = np.ones(5, dtype=np.float32)
    first = last = None
    for it in range(8):
        loss, _, _ = train_step(m, opt, batch, lw)
        last = loss.item()
        first = last if first is None else first
    assert np.isfinite(last) and last < first
