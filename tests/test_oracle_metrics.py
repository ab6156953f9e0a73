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
