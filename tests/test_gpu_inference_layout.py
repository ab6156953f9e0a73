"""Inference layout of the program executor (sgnn_prog_forward with training | 2; HISTORY.md section 2, INTEGRATION.md
"Inference memory"): when nothing asks for a gradient, the buffers of a program share storage by liveness, all programs of
a device share ONE arena and the outputs are copied out.  The arithmetic is the training layout's: outputs must be
bit-identical, in eval and in training mode, and results of an earlier call must survive later calls."""
import numpy as np
import pytest
import torch

from util import param_fill
from sgnn_amd import synth

pytestmark = pytest.mark.gpu


def _model(dims, cfg, train):
    from sgnn_amd.model import GenModel
    m = param_fill(GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1), seed=cfg)
    return m.train(train).cuda()


def _same(a, b):
    (sa, oa), (sb, ob) = a, b
    assert len(oa) == len(ob)
    for (la, va), (lb, vb) in zip(oa, ob):
        assert torch.equal(la, lb) and torch.equal(va, vb)
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1])


@pytest.mark.parametrize('train', [False, True])
def test_inference_layout_is_the_same_arithmetic(train):
    from sgnn_amd.scn import program as P_
    dims, cfg = (32, 32, 32), 11
    data = synth.make_batch(3, dims, cfg=cfg, occupancy=0.08)
    locs, feats = data['input'][0].cuda(), data['input'][1].cuda()
    lw = np.ones(5, dtype=np.float32)
    m = _model(dims, cfg, train)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    feats_g = feats.clone().requires_grad_(True)            # a gradient may be asked for: training layout
    ref = m([locs, feats_g], lw)
    assert ref[1][0][1].requires_grad
    sizes_train = [p.last_arena_floats for p in P_.programs_of(m)]
    m.load_state_dict(state)                                # training mode moved the running statistics
    with torch.no_grad():                                   # nothing can: inference layout
        out = m([locs, feats], lw)
    sizes_infer = [p.last_arena_floats for p in P_.programs_of(m)]
    _same(ref, out)
    assert all(a[0] > b[0] for a, b in zip(sizes_train, sizes_infer))       # every program's forward arena shrank
    assert sum(b[0] for b in sizes_infer) * 3 < sum(a[0] for a in sizes_train) * 2
    # the shared arena is reused by the next call: results handed out before must not change
    keep = [v.clone() for _, v in out[1]] + [out[0][1].clone()]
    data2 = synth.make_batch(2, dims, cfg=cfg + 1, occupancy=0.1)
    m.load_state_dict(state)
    with torch.no_grad():
        m([data2['input'][0].cuda(), data2['input'][1].cuda()], lw)
    for a, b in zip(keep, [v for _, v in out[1]] + [out[0][1]]):
        assert torch.equal(a, b)


def test_requires_grad_parameters_alone_keep_the_training_layout():
    """Parameters that require a gradient make a backward pass possible: grad mode on -> training layout even if the
    input features do not require one; the backward pass then works."""
    dims, cfg = (32, 32, 32), 12
    data = synth.make_batch(2, dims, cfg=cfg, occupancy=0.08)
    m = _model(dims, cfg, True)
    lw = np.ones(5, dtype=np.float32)
    sdf, occ = m([data['input'][0].cuda(), data['input'][1].cuda()], lw)
    loss = sum(v.sum() for _, v in occ if torch.is_tensor(v))        # (a level may predict nothing: [[], []])
    if torch.is_tensor(sdf[1]):
        loss = loss + sdf[1].sum()
    loss.backward()
    got = [p.grad for p in m.encoder.parameters()]
    assert all(g is not None and torch.isfinite(g).all() for g in got)
