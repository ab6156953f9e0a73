"""CPU: the metrics oracle against numbers produced by the reference's own loss.py
(tests/golden/make_golden_metrics.py)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))
import metrics_oracle as mo  # noqa: E402

B, TRUNC = 3, 3.0


@pytest.fixture(scope='module')
def g():
    return np.load(os.path.join(HERE, 'golden', 'metrics_expected.npz'))


def level_pred(g, h):
    locs, vals = g['locs%d' % h], g['vals%d' % h]
    keep = 1.0 / (1.0 + np.exp(-vals[:, 0].astype(np.float32))) > 0.5
    return [locs[(locs[:, 3] == b) & keep][:, :3] for b in range(B)]


@pytest.mark.parametrize('h', range(4))
def test_iou_matches_reference(g, h):
    tgt = g['target_occ%d' % h].astype(np.uint8)          # -1 -> 255, the reference's .byte()
    pred = level_pred(g, h)
    for masking in (True, False):
        k = 'iou%d_m%d' % (h, int(masking))
        assert mo.compute_iou_sparse_dense(pred, tgt, masking) == float(g[k])
        assert np.array_equal(mo.compute_iou_sparse_dense(pred, tgt, masking, batched=False), g[k + '_per'])
    pred[2] = None
    assert mo.compute_iou_sparse_dense(pred, tgt, True) == float(g['iou%d_none' % h])


def test_iou_without_predictions(g):
    assert mo.compute_iou_sparse_dense([None] * B, g['target_occ3'].astype(np.uint8), True) == -1 == g['iou_allnone']


def test_l1_tgtsurf_matches_reference(g):
    for masking in (True, False):
        for thresh, tag in ((None, 'n'), (1.0, '1')):
            got = mo.compute_l1_tgtsurf_sparse_dense(g['sdf_locs'], g['sdf_vals'], g['target_sdf'], TRUNC, masking,
                                                     g['known'], thresh=thresh)
            assert got == pytest.approx(float(g['l1tgt_m%d_t%s' % (int(masking), tag)]), rel=2e-6)


@pytest.mark.parametrize('seed', range(4))
def test_oracle_matches_live_reference_on_random_predictions(seed):
    """Random targets / predictions through the reference's own loss.py (imported live where /root/reference exists;
    sparseconvnet / plyfile / marching_cubes stubbed for the import only) and through the oracle."""
    ref_dir = '/root/reference/torch'
    if not os.path.isdir(ref_dir):
        pytest.skip('reference sources not present on this machine')
    import types
    import torch
    added = []
    for name in ('sparseconvnet', 'plyfile', 'marching_cubes', 'marching_cubes.marching_cubes'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
            added.append(name)
    sys.modules['marching_cubes'].marching_cubes = sys.modules['marching_cubes.marching_cubes']
    sys.path.insert(0, ref_dir)
    try:
        import loss as ref_loss
    finally:
        sys.path.remove(ref_dir)
        for name in added:
            sys.modules.pop(name, None)
        for name in ('loss', 'data_util'):        # the reference's module names must not shadow anything later
            sys.modules.pop(name, None)
    rng = np.random.default_rng(seed)
    nb, d = 2, int(rng.choice([8, 12]))
    occ = rng.choice([-1, 0, 1], size=(nb, 1, d, d, d), p=[0.15, 0.6, 0.25]).astype(np.int8)
    pred = []
    for b in range(nb):
        cells = rng.permutation(d ** 3)[:int(rng.integers(5, d ** 3 // 2))]
        cells.sort()
        pred.append(np.stack([cells // (d * d), (cells // d) % d, cells % d], 1).astype(np.int64))
    tgt_b = torch.from_numpy(occ.astype(np.uint8))
    tpred = [torch.from_numpy(p) for p in pred]
    for masking in (True, False):
        want = ref_loss.compute_iou_sparse_dense(tpred, tgt_b, masking)
        assert mo.compute_iou_sparse_dense(pred, occ.astype(np.uint8), masking) == want
        want_per = ref_loss.compute_iou_sparse_dense(tpred, tgt_b, masking, batched=False)
        assert np.array_equal(mo.compute_iou_sparse_dense(pred, occ.astype(np.uint8), masking, batched=False), want_per)
    sdf = rng.normal(0, 2.5, (nb, 1, d, d, d)).astype(np.float32).clip(-3, 3)
    known = rng.integers(0, 4, (nb, 1, d, d, d)).astype(np.uint8)
    locs = np.concatenate([np.concatenate([p, np.full((len(p), 1), b)], 1) for b, p in enumerate(pred)])
    vals = rng.normal(0, 2, (len(locs), 1)).astype(np.float32)
    for masking in (True, False):
        for thresh in (None, 1.0):
            want = ref_loss.compute_l1_tgtsurf_sparse_dense(torch.from_numpy(locs), torch.from_numpy(vals),
                                                            torch.from_numpy(sdf), 3.0, masking, torch.from_numpy(known),
                                                            thresh=thresh)
            got = mo.compute_l1_tgtsurf_sparse_dense(locs, vals, sdf, 3.0, masking, known, thresh=thresh)
            assert got == pytest.approx(want, rel=5e-6)
