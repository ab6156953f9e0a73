"""Capacity mode (device-side row counts, scn/capacity.py) and the graph-captured training step (train.GraphStep).

The classic path sizes every generative level on the host from a read-back of `sigmoid(pred) > 0.5` (torch/model.py:233,
322); capacity mode keeps the counts on the device.  It must be the SAME computation: identical site lists (the live
prefix of every capacity-sized tensor), the same logits / loss / gradients up to the summation order of reductions whose
block partition depends on the launch size, and — replayed from a HIP graph — bit-identical to itself run eagerly."""
import numpy as np
import pytest
import torch

from util import param_fill
from sgnn_amd import synth

pytestmark = pytest.mark.gpu

DIMS, CFG = (32, 32, 32), 23


def _batch(seed, n=2):
    d = synth.make_batch(n, DIMS, cfg=seed, occupancy=0.08)
    return {'input': [d['input'][0].cuda(), d['input'][1].cuda()], 'sdf': d['sdf'].cuda(), 'known': d['known'].cuda(),
            'hierarchy': [h.cuda() for h in d['hierarchy']]}


def _model(seed=CFG):
    from sgnn_amd.model import GenModel
    return param_fill(GenModel(8, DIMS, 1, 16, 16, 4, True, True, 1, 1), seed).train().cuda()


def _classic(model, batch, lw):
    """Forward + loss + backward on the classic path, logging the row counts."""
    from sgnn_amd import loss as L
    from sgnn_amd.scn import metadata as MD
    MD.COUNT_LOG = []
    try:
        (tsdf, toccs, thier), w = L.compute_targets_and_weights(batch['sdf'], batch['hierarchy'], 4, 3.0, True, batch['known'],
                                                                5.0, batch['input'][0])
        osdf, oocc = model(batch['input'], lw, batch_size=int(batch['sdf'].shape[0]))
        loss, _ = L.compute_loss(osdf, oocc, tsdf, toccs, thier, lw, 3.0, True, 5.0, batch['input'][0], True, batch['known'],
                                 weights=w)
        loss.backward()
        return osdf, oocc, loss, MD.COUNT_LOG
    finally:
        MD.COUNT_LOG = None


def _capped(model, batch, lw, cap):
    from sgnn_amd import loss as L
    locs, feats = batch['input']
    n = int(locs.shape[0])
    slocs = torch.zeros(cap.input_rows, 4, dtype=torch.int64, device='cuda')
    sfeats = torch.zeros(cap.input_rows, feats.shape[1], device='cuda')
    slocs[:n], sfeats[:n] = locs, feats
    slocs._sgnn_cnt = cap.input_cnt()
    cap.set_input_rows(n)
    (tsdf, toccs, thier), w = L.compute_targets_and_weights(batch['sdf'], batch['hierarchy'], 4, 3.0, True, batch['known'], 5.0,
                                                            slocs)
    osdf, oocc = model([slocs, sfeats], lw, batch_size=int(batch['sdf'].shape[0]), capacity=cap)
    loss, _ = L.compute_loss(osdf, oocc, tsdf, toccs, thier, lw, 3.0, True, 5.0, slocs, True, batch['known'], weights=w)
    loss.backward()
    return osdf, oocc, loss


def test_capacity_forward_backward_equals_the_classic_path():
    from sgnn_amd.scn.capacity import Capacity, trim
    from sgnn_amd.scn.metadata import runtime
    lw = np.ones(5, dtype=np.float32)
    batch = _batch(3)
    ma, mb = _model(), _model()
    sa, oa, la, log = _classic(ma, batch, lw)
    cap = Capacity.from_log('cuda', log, headroom=1.4)
    rt = runtime(torch.device('cuda', 0))
    syncs = rt.syncs
    sb, ob, lb = _capped(mb, batch, lw, cap)
    assert rt.syncs == syncs, 'capacity mode must not read anything back'
    assert int(rt.state[1].item()) & 4 == 0
    live = cap.read()
    want = [e for e in log if e[0] == 'gen']
    assert live['input'] == batch['input'][0].shape[0]
    assert live['enc'] == [e for e in log if e[0] == 'enc'][0][2]
    # The two passes run the same arithmetic, but reductions whose block partition follows the launch size (BatchNorm
    # statistics of the joined tensors, the small-/large-level convolution choice) sum in a different order, so a
    # logit within ~1e-5 of the threshold may be decided differently: site lists must agree except for the children of
    # such borderline sites (a handful at most), values are compared on the common sites.
    def keys(t):
        t = t.long()
        return (t[:, 3] << 48) | (t[:, 0] << 32) | (t[:, 1] << 16) | t[:, 2]
    mism = 0
    for h in range(5):
        (sa_, va), (sb_, vb) = (oa[h] if h < 4 else sa), (ob[h] if h < 4 else sb)
        sb_ = trim(sb_)
        va, vb = va.detach(), vb.detach()[:sb_.shape[0]]
        if torch.equal(sa_, sb_):
            ia = ib = torch.arange(sa_.shape[0], device='cuda')
        else:
            ka, kb = keys(sa_), keys(sb_)
            common = ka[torch.isin(ka, kb)]
            ia = torch.nonzero(torch.isin(ka, common)).view(-1)
            ib = torch.nonzero(torch.isin(kb, common)).view(-1)
            assert torch.equal(ka[ia], kb[ib]), 'level %d: common sites are not in the same order' % h
            mism += (sa_.shape[0] - ia.numel()) + (sb_.shape[0] - ib.numel())
        d = (va[ia] - vb[ib]).abs()
        tol = 5e-5 * max(1.0, float(va.abs().max()))
        if mism == 0:
            assert float(d.max()) <= tol, (h, float(d.max()))
        else:   # a site decided differently changes its neighbourhood (and, slightly, the level's BatchNorm statistics)
            assert float((d > 100 * tol).float().mean()) <= 0.02, (h, float(d.max()))
    assert mism <= 32, 'site lists differ by %d sites' % mism
    if mism == 0:
        assert [k for k, _ in live['gen']] == [e[1] for e in want]
        for (k, pyr), e in zip(live['gen'], want):
            assert pyr[:len(e[2])] == e[2]
    assert abs(la.item() - lb.item()) <= (2e-5 if mism == 0 else 2e-3) * abs(la.item())
    for (na, pa), (nb, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert pa.grad is not None and pb.grad is not None, na
        scale = float(pa.grad.abs().max()) + 1e-12
        lim = (2e-3 if mism == 0 else 2e-2) * scale + 1e-7
        assert float((pa.grad - pb.grad).abs().max()) <= lim, (na, float((pa.grad - pb.grad).abs().max()), scale)
    for (na, ba), (nb, bb) in zip(ma.named_buffers(), mb.named_buffers()):
        assert torch.allclose(ba.float(), bb.float(), rtol=1e-4, atol=1e-5), na


def test_flat_adam_is_adam():
    from sgnn_amd.train import FlatAdam
    torch.manual_seed(5)
    shapes = [(27, 8, 8), (16,), (3, 5), (1,)]
    pa = [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    ref = torch.optim.Adam(pa, lr=1e-2, weight_decay=1e-3)
    opt = FlatAdam([('a', pb[:2]), ('b', pb[2:])], lr=1e-2, weight_decay=1e-3)
    for it in range(6):
        gs = [torch.randn(s, device='cuda') for s in shapes]
        skip_b = it == 2                      # segment b has no gradient in step 2: Adam leaves it (and its step) alone
        for p, q, g in zip(pa, pb, gs):
            in_b = any(q is x for x in pb[2:])
            p.grad = None if (skip_b and in_b) else g.clone()
            q.grad = None if (skip_b and in_b) else g.clone()
        ref.step()
        opt.step(reached=opt.collect())
        for p, q in zip(pa, pb):
            assert torch.allclose(p, q, rtol=2e-6, atol=1e-7), (it, float((p - q).abs().max()))
    sd = opt.state_dict()
    assert sd['state'][0]['step'].item() == 6 and sd['state'][3]['step'].item() == 5
    assert torch.allclose(sd['state'][1]['exp_avg'], ref.state[pa[1]]['exp_avg'], rtol=1e-5, atol=1e-7)
    ref2 = torch.optim.Adam(pa, lr=1e-2, weight_decay=1e-3)
    ref2.load_state_dict(sd)                  # torch accepts the checkpoint as is


def _train(use_graph, steps, batches, lw, headroom=1.4, **kw):
    from sgnn_amd.train import GraphStep
    m = _model()
    gs = GraphStep(m, lr=1e-3, use_graph=use_graph, headroom=headroom, settle=False, **kw)
    losses = []
    for i in range(steps):
        losses.append(gs(batches[i % len(batches)], lw).clone())
    torch.cuda.synchronize()
    return m, gs, [float(v) for v in losses]


def test_graph_replay_is_bit_identical_to_eager_capacity_steps():
    lw = np.ones(5, dtype=np.float32)
    batches = [_batch(3), _batch(4)]
    ma, ga, la = _train(False, 6, batches, lw)
    mb, gb, lb = _train(True, 6, batches, lw)
    assert gb.stats['captures'] == 1 and gb.stats['replays'] == 4 and gb.stats['overflows'] == 0, gb.stats
    assert ga.stats['eager_steps'] == 5 and ga.stats['probe_steps'] == 1, ga.stats
    assert la == lb, (la, lb)
    for (na, pa), (nb, pb) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert torch.equal(pa, pb), na


def test_pyramid_lane_and_replan_do_not_change_the_training_run():
    """(a) metadata.SIDE_PYRAMID: the stride-2 pyramids and coarse rulebooks built on the side lane and joined inside
    sgnn_prog_forward are the same tables — six captured steps give bit-identical losses and parameters with the lane on
    and off.  (b) GraphStep.replan() re-sizes the capacities from the live counts and re-captures without touching the
    training run (up to the summation order of the weight-gradient partials)."""
    from sgnn_amd.scn import metadata as MD
    lw = np.ones(5, dtype=np.float32)
    batches = [_batch(3), _batch(4)]
    runs = []
    prev = MD.SIDE_PYRAMID
    try:
        for on in (True, False):
            MD.SIDE_PYRAMID = on
            runs.append(_train(True, 6, batches, lw))
    finally:
        MD.SIDE_PYRAMID = prev
    (ma, ga, la), (mb, gb, lb) = runs
    assert ga.stats['captures'] == 1 and gb.stats['captures'] == 1
    assert la == lb, (la, lb)
    for (na, pa), (nb, pb) in zip(ma.state_dict().items(), mb.state_dict().items()):
        assert torch.equal(pa, pb), na
    # (b) the same run with a re-plan in the middle
    from sgnn_amd.train import GraphStep
    m = _model()
    gs = GraphStep(m, lr=1e-3, headroom=1.4, settle=False)
    losses = []
    for i in range(6):
        if i == 3:
            old = gs.capacity.describe()
            gs.replan()
            assert gs.graphs is None and gs.stats['replans'] == 1
        losses.append(float(gs(batches[i % 2], lw)))
    torch.cuda.synchronize()
    assert gs.stats['captures'] == 2 and gs.stats['overflows'] == 0, gs.stats
    new = gs.capacity.describe()
    assert new != old and all(n >= 1024 for n in [new['input']] + new['enc'])
    # (weight-gradient partials are cut by the launch grid, which follows the capacities: same sums, another order)
    assert np.allclose(losses, la, rtol=1e-6, atol=0), (losses, la)
    for (na, pa), (nb, pb) in zip(ma.state_dict().items(), m.state_dict().items()):
        if pa.is_floating_point():
            assert float((pa - pb).abs().max()) <= 1e-6 + 1e-5 * float(pa.abs().max()), na
        else:
            assert torch.equal(pa, pb), na


def test_graph_step_follows_the_classic_training_loop():
    """Same batches, same Adam: the losses of graph-replayed steps track train_step's (different reduction partitions
    only: a few 1e-5 relative after a handful of steps)."""
    from sgnn_amd.train import FlatAdam, genmodel_segments
    from sgnn_amd import loss as L
    lw = np.ones(5, dtype=np.float32)
    batches = [_batch(3), _batch(4)]
    _, _, lg = _train(True, 5, batches, lw)
    m = _model()
    opt = FlatAdam(genmodel_segments(m), lr=1e-3)
    lc = []
    for i in range(5):
        b = batches[i % 2]
        opt.zero_grad()
        _, _, loss, _ = _classic(m, b, lw)
        opt.step(reached=opt.collect())
        lc.append(float(loss))
    assert np.allclose(lg, lc, rtol=5e-4), (lg, lc)
    assert lc[-1] < lc[0]


def test_overflow_discards_the_step_and_recovers():
    from sgnn_amd.train import GraphStep
    lw = np.ones(5, dtype=np.float32)
    small, big = _batch(3, n=2), _batch(9, n=2)
    m = _model()
    gs = GraphStep(m, lr=1e-3, headroom=1.3, settle=False)
    gs(small, lw)                                      # probe
    gs(small, lw)                                      # eager capacity step
    gs(small, lw)                                      # capture + replay
    assert gs.stats['overflows'] == 0
    caps0 = gs.capacity.describe()
    before = [p.detach().clone() for p in m.parameters()]
    # shrink one level's capacity under the live count: the next replay must overflow, leave the parameters alone,
    # and the step after that must have re-run the batch on larger capacities
    live = gs.capacity.read()
    k2 = live['gen'][2][0]
    assert k2 > 64
    from sgnn_amd.scn.capacity import Capacity
    gs._drain()
    shrunk = Capacity('cuda', gs.capacity.input_rows, gs.capacity.enc,
                      [(k if g != 2 else k2 // 2, p) for g, (k, p) in enumerate(gs.capacity.gen)])
    gs.capacity, gs.stage, gs.graphs, gs._live = shrunk, 1, None, None
    before = [p.detach().clone() for p in m.parameters()]
    gs(small, lw)                                      # eager capacity step on the shrunk plan: overflows
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(before, m.parameters())), 'an overflowed step must not update anything'
    gs(small, lw)                                      # detects it (one step late), re-runs the batch, grows
    gs(small, lw)
    gs(small, lw)
    torch.cuda.synchronize()
    assert gs.stats['overflows'] >= 1
    assert gs.capacity.gen[2][0] >= k2
    assert any(not torch.equal(a, b) for a, b in zip(before, m.parameters()))
    assert np.isfinite(float(gs.loss))


def test_empty_level_in_capacity_mode_behaves_like_the_reference_early_return():
    """Nothing predicted occupied after the first refinement: the reference stops there (torch/model.py:211), the later
    stages' parameters get no gradient and Adam skips them.  Capacity mode runs those stages on zero rows: same result."""
    from sgnn_amd.train import GraphStep
    lw = np.ones(5, dtype=np.float32)
    b = _batch(3)

    def build():
        m = _model()
        with torch.no_grad():
            m.refinement[0].linear.weight.zero_()
            m.refinement[0].linear.bias.fill_(-30.0)   # sigmoid(out) <= 0.5 everywhere: level 1 keeps nothing
        return m
    m = build()
    gs = GraphStep(m, lr=1e-3, headroom=1.5, settle=False)
    state0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    for _ in range(4):
        loss = gs(b, lw)
    torch.cuda.synchronize()
    assert gs.stats['replays'] >= 1 and np.isfinite(float(loss))
    live = gs.capacity.read()
    assert live['gen'][0][0] > 0 and live['gen'][1][0] == 0
    sd = m.state_dict()
    for k, v in sd.items():
        later = k.startswith('refinement.1.') or k.startswith('refinement.2.') or k.startswith('surfacepred.')
        if later:
            assert torch.equal(v, state0[k]), 'stage without input sites changed: %s' % k
    assert any(not torch.equal(sd[k], state0[k]) for k in sd if k.startswith('refinement.0.p1'))
    steps = gs.opt.steps.cpu().tolist()
    assert steps[0] == 4 and steps[1] == 4 and steps[2] == 0 and steps[3] == 0 and steps[4] == 0


def test_counts_of_an_overflowed_step_never_size_a_plan():
    """Regression (round 5, found by tests/test_gpu_dp_protocol.py scenario C): the hierarchy dies in the probe step, so the
    first plan has minimum capacities below the dead level; the next capacity steps overflow by a factor of 10 there.  The
    counts that ride back with an overflowed step are CLAMPED to the capacities — they must not become the "live" counts a
    later re-plan is sized from (the plan was re-sized to 2 x 1024 rows for a 10 k-row level and overflowed again)."""
    from sgnn_amd.train import GraphStep
    from sgnn_amd.scn import functions as F_
    from test_gpu_dp_protocol import _kill_level
    lw = np.ones(5, dtype=np.float32)
    batches = [_batch(30 + i) for i in range(6)]
    real = F_.compact_sigmoid_plan
    active = [True]
    calls = _kill_level(active)
    try:
        gs = GraphStep(_model(), lr=1e-3, headroom=2.0, settle=False, teacher_forced=True)
        for i, b in enumerate(batches):
            calls[0] = 0
            gs(b, lw)
            active[0] = False
        torch.cuda.synchronize()
        gs._drain()
    finally:
        F_.compact_sigmoid_plan = real
    assert gs.stats['overflows'] == 2 and gs.stats['replans'] == 0, (gs.stats, gs.overflow_log)
    assert [i for i, _ in gs.overflow_log] == [0, 1], gs.overflow_log      # the two capacity steps on the minimum plan
    assert all(full for _, full in gs.overflow_log)                        # ... each names the levels that were full
    live = gs.capacity.read()
    assert live['gen'][3][0] > 2048 and gs.capacity.gen[3][0] >= live['gen'][3][0]
    assert gs.stats['captures'] == 2 and gs.stats['replays'] >= 2
