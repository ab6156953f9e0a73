"""GPU: device IoU / target-surface L1 (sgnn_iou_counts, sgnn_l1_tgtsurf) against the numbers the reference's
loss.py produced for the same inputs, and against the oracle at training size.  IoU is integer work: exact."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))
import metrics_oracle as mo  # noqa: E402

from sgnn_amd import loss as L, metrics, synth  # noqa: E402

pytestmark = pytest.mark.gpu
B, TRUNC = 3, 3.0


@pytest.fixture(scope='module')
def g():
    return np.load(os.path.join(HERE, 'golden', 'metrics_expected.npz'))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize('h', range(4))
@pytest.mark.parametrize('as_byte', [True, False])
def test_iou_matches_reference(g, h, as_byte):
    occ = g['target_occ%d' % h]
    tgt = dev(occ.astype(np.uint8)) if as_byte else dev(occ.astype(np.float32))
    locs, vals = dev(g['locs%d' % h]), dev(g['vals%d' % h])
    pred = metrics.pred_occs_from_outputs([[locs, vals]], B)[0]
    for masking in (True, False):
        k = 'iou%d_m%d' % (h, int(masking))
        assert metrics.compute_iou_sparse_dense(pred, tgt, masking) == float(g[k])            # list API
        assert np.array_equal(metrics.compute_iou_sparse_dense(pred, tgt, masking, batched=False), g[k + '_per'])
        c = metrics.iou_counts(locs, tgt, masking, logits=vals)                               # training-loop API
        assert metrics.iou_from_counts(c) == float(g[k])
    pred[2] = None
    assert metrics.compute_iou_sparse_dense(pred, tgt, True) == float(g['iou%d_none' % h])


def test_iou_edge_cases(g):
    tgt = dev(g['target_occ3'].astype(np.uint8))
    assert metrics.compute_iou_sparse_dense([None] * B, tgt, True) == -1
    empty = [torch.zeros((0, 3), dtype=torch.long, device='cuda')] * B
    assert metrics.compute_iou_sparse_dense(empty, tgt, True) == 0.0        # union = target voxels, nothing hit
    none_occ = torch.zeros_like(tgt)
    assert metrics.compute_iou_sparse_dense(empty, none_occ, True) == -1    # empty union
    with pytest.raises(RuntimeError, match='GPU only'):
        metrics.iou_counts(torch.zeros((0, 4), dtype=torch.long), tgt.cpu(), True)


def test_level_ious_from_model_output(g):
    outs = [[dev(g['locs%d' % h]), dev(g['vals%d' % h])] for h in range(4)]
    outs[1] = [torch.zeros((0, 4), dtype=torch.long, device='cuda'), torch.zeros((0, 2), device='cuda')]
    tgts = [dev(g['target_occ%d' % h].astype(np.float32)) for h in range(4)]
    got = metrics.level_ious(outs, tgts, True)
    assert got[1] == -1
    for h in (0, 2, 3):
        assert got[h] == float(g['iou%d_m1' % h])


def test_l1_tgtsurf_matches_reference(g):
    sl, sv, tgt, kn = dev(g['sdf_locs']), dev(g['sdf_vals']), dev(g['target_sdf']), dev(g['known'])
    for masking in (True, False):
        for thresh, tag in ((None, 'n'), (1.0, '1')):
            got = metrics.compute_l1_tgtsurf_sparse_dense(sl, sv, tgt, TRUNC, masking, kn, thresh=thresh)
            assert got == pytest.approx(float(g['l1tgt_m%d_t%s' % (int(masking), tag)]), rel=2e-6)   # fp32 mean vs fp64 sums
        l1p = L.compute_l1_predsurf_sparse_dense(sl, sv, tgt, None, False, masking, kn)               # train.py:296
        assert float(l1p) == pytest.approx(float(g['l1pred_m%d' % int(masking)]), rel=2e-6)
    one = sl[:, 3] == 0
    got = metrics.compute_l1_tgtsurf_sparse_dense(sl[one], sv[one], tgt[:1], TRUNC, True, kn[:1], batched=False)
    assert got.shape == (1,) and got[0] == pytest.approx(float(g['l1tgt_single'][0]), rel=2e-6)
    with pytest.raises(ValueError):
        metrics.compute_l1_tgtsurf_sparse_dense(sl, sv, tgt, TRUNC, True, kn, batched=False)


def test_training_size_against_oracle():
    """configs[1] geometry: 32 blocks of 64^3, every voxel of a thick band predicted."""
    data = synth.make_batch(32, 64, cfg=5, occupancy=0.05)
    sdf, known = data['sdf'].cuda(), data['known'].cuda()
    tgt_sdf, tgt_occs, _ = L.compute_targets(sdf.clone(), [h.cuda() for h in data['hierarchy']], 4, TRUNC, True, known)
    rng = np.random.default_rng(3)
    occ = tgt_occs[3].cpu().numpy()
    cand = np.argwhere(np.abs(data['sdf'][:, 0].numpy()) < 4.5)
    cand = cand[rng.random(len(cand)) < 0.8]
    locs = np.concatenate([cand[:, 1:], cand[:, :1]], 1).astype(np.int64)
    logits = rng.normal(0.5, 2.0, (len(locs), 2)).astype(np.float32)
    c = metrics.iou_counts(dev(locs), tgt_occs[3], True, logits=dev(logits))
    keep = 1.0 / (1.0 + np.exp(-logits[:, 0])) > 0.5
    pred = [locs[(locs[:, 3] == b) & keep][:, :3] for b in range(32)]
    occ_u8 = occ.astype(np.int8).astype(np.uint8)
    assert metrics.iou_from_counts(c) == mo.compute_iou_sparse_dense(pred, occ_u8, True)
    assert np.array_equal(metrics.iou_from_counts(c, batched=False), mo.compute_iou_sparse_dense(pred, occ_u8, True, False))
    vals = rng.normal(0, 2, len(locs)).astype(np.float32)
    got = metrics.compute_l1_tgtsurf_sparse_dense(dev(locs), dev(vals), tgt_sdf, TRUNC, True, known)
    want = mo.compute_l1_tgtsurf_sparse_dense(locs, vals, tgt_sdf.cpu().numpy(), TRUNC, True, known.cpu().numpy())
    assert got == pytest.approx(want, rel=1e-9)
