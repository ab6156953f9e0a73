"""TEST INFRASTRUCTURE — numpy restatement of the reference's evaluation metrics.
Only tests/ may import this.

  compute_iou_sparse_dense          torch/loss.py:84-120 (flatten, mask out UNKNOWN target voxels from the
                                    prediction, np.intersect1d / np.union1d per sample)
  compute_l1_tgtsurf_sparse_dense   torch/loss.py:201-231 (dense prediction filled with -truncation, gathers at the
                                    target's surface voxels, optional known mask, mean absolute difference)

Parity status: PINNED — tests/test_oracle_metrics.py holds both functions to tests/golden/metrics_expected.npz,
produced by tests/golden/make_golden_metrics.py from the reference's own loss.py.
"""
import numpy as np

UNK_THRESH = 2


def compute_iou_sparse_dense(sparse_pred_locs, dense_tgts, use_loss_masking, batched=True):
    """dense_tgts: (B,1,D0,D1,D2) uint8 with 255 = unknown (train.py:288 `.byte()` of -1)."""
    dims = dense_tgts.shape[2:]
    nb = dense_tgts.shape[0]
    corr, union = np.zeros(nb, dtype=np.float64), np.zeros(nb, dtype=np.float64)
    for b in range(nb):
        if sparse_pred_locs[b] is None:
            continue
        tgt = dense_tgts[b, 0].reshape(-1)
        p = np.asarray(sparse_pred_locs[b], dtype=np.int64)
        pred = p[:, 0] * dims[1] * dims[2] + p[:, 1] * dims[2] + p[:, 2]
        tgtlocs = np.nonzero(tgt == 1)[0]
        if use_loss_masking:
            pred = np.setdiff1d(pred, np.nonzero(tgt == 255)[0], assume_unique=True)
        corr[b] = len(np.intersect1d(pred, tgtlocs, assume_unique=True))
        union[b] = len(np.union1d(pred, tgtlocs))
    if not batched:
        with np.errstate(divide='ignore', invalid='ignore'):
            return np.divide(corr.astype(np.float32), union.astype(np.float32))
    return corr.sum() / union.sum() if union.sum() > 0 else -1


def compute_l1_tgtsurf_sparse_dense(locs, vals, dense_tgts, truncation, use_loss_masking, known, thresh=None):
    nb, _, d0, d1, d2 = dense_tgts.shape
    pred = np.full(nb * d0 * d1 * d2, -truncation, dtype=np.float32)
    locs = np.asarray(locs, dtype=np.int64)
    pred[((locs[:, 3] * d0 + locs[:, 0]) * d1 + locs[:, 1]) * d2 + locs[:, 2]] = np.asarray(vals, np.float32).reshape(-1)
    t = dense_tgts.reshape(-1)
    sel = (np.abs(t) <= thresh) if thresh is not None else (np.abs(t) < truncation)
    if use_loss_masking:
        sel &= known.reshape(-1) < UNK_THRESH
    return np.abs(pred[sel].astype(np.float64) - t[sel].astype(np.float64)).mean() if sel.any() else float('nan')
