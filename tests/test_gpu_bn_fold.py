"""BatchNormReLU folded into the consuming convolution's gather (sgnn_prog_set_bn_fold, csrc/prog.hip make_plan): every
scn.BatchNormReLU whose only reader is a SubmanifoldConvolution / Convolution / the up-sampling convolution
(torch/model.py:37-42, 181, 187, 256) launches no apply pass — the convolution and its weight gradient normalise the rows
they gather.  Same statistics, same per-value arithmetic (sgnn_bn_act), missing rules stay zero: the training step must be
BIT-IDENTICAL to the reference path with the apply pass, and it must launch fewer kernels.  (The fold is not the default:
it measured neutral, see csrc/prog.hip g_bn_fold; these tests keep the selectable path correct.)"""
import numpy as np
import pytest
import torch

from util import param_fill
from sgnn_amd import synth

pytestmark = pytest.mark.gpu


def _step(dims, batch, cfg, occupancy, fold, train=True):
    from sgnn_amd import _lib, loss as L
    from sgnn_amd.model import GenModel, named_gradients
    lib = _lib.load()
    prev = lib.sgnn_prog_set_bn_fold(1 if fold else 2)      # 2 = the exact reference path of the fold
    prev_rows = lib.sgnn_prog_set_bn_fold_rows(0)      # fold on EVERY level (default: levels of >= 40 960 rows only)
    try:
        m = param_fill(GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1), cfg).train(train).cuda()
        data = synth.make_batch(batch, dims, cfg=cfg, occupancy=occupancy)
        locs, feats = data['input'][0].cuda(), data['input'][1].cuda()
        lw = np.ones(5, dtype=np.float32)
        t = L.compute_targets(data['sdf'].clone().cuda(), [h.clone().cuda() for h in data['hierarchy']], 4, 3, True,
                              data['known'].cuda())
        torch.cuda.synchronize()
        n0 = lib.sgnn_launch_count()
        with torch.set_grad_enabled(train):
            osdf, oocc = m([locs, feats], lw, batch_size=batch)
            if train:
                loss, _ = L.compute_loss(osdf, oocc, t[0], t[1], t[2], lw, 3, True, 5.0, locs, True, data['known'].cuda())
                loss.backward()
        torch.cuda.synchronize()
        launches = lib.sgnn_launch_count() - n0
        outs = [o[1].detach().clone() for o in oocc if torch.is_tensor(o[1])] + [osdf[1].detach().clone()]
        sites = [o[0].clone() for o in oocc if torch.is_tensor(o[0])] + [osdf[0].clone()]
        grads = dict((k, v.clone()) for k, v in named_gradients(m).items() if v is not None) if train else {}
        bufs = dict((k, v.clone()) for k, v in m.named_buffers())
        return outs, sites, (float(loss) if train else None), grads, bufs, launches
    finally:
        lib.sgnn_prog_set_bn_fold(prev)
        lib.sgnn_prog_set_bn_fold_rows(prev_rows)


@pytest.mark.parametrize('dims,batch,cfg,occ', [((32, 32, 32), 2, 11, 0.08), ((64, 64, 64), 5, 2, 0.05)])
def test_fold_is_bit_identical_to_the_apply_pass(dims, batch, cfg, occ):
    """32^3: every level runs the small-level kernels; 64^3 x 5: the input level (57 k rows) runs the large-level ones."""
    a = _step(dims, batch, cfg, occ, True)
    b = _step(dims, batch, cfg, occ, False)
    for sa, sb in zip(a[1], b[1]):
        assert torch.equal(sa, sb)
    for va, vb in zip(a[0], b[0]):
        assert torch.equal(va, vb), float((va - vb).abs().max())
    assert a[2] == b[2], (a[2], b[2])
    assert sorted(a[3]) == sorted(b[3]) and len(a[3]) > 150
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), (k, float((a[3][k] - b[3][k]).abs().max()))
    for k in a[4]:
        assert torch.equal(a[4][k], b[4][k]), k
    # 39 of the 57 BatchNorm layers of the five programs sit in front of a convolution: their apply passes are gone
    assert b[5] - a[5] >= 35, (a[5], b[5])


def test_fold_in_eval_mode_uses_the_running_statistics():
    a = _step((32, 32, 32), 2, 11, 0.08, True, train=False)
    b = _step((32, 32, 32), 2, 11, 0.08, False, train=False)
    for sa, sb in zip(a[1], b[1]):
        assert torch.equal(sa, sb)
    for va, vb in zip(a[0], b[0]):
        assert torch.equal(va, vb)
    assert b[5] > a[5]
