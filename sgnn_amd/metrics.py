"""Evaluation metrics on the device — counterpart of torch/loss.py:84-120 (compute_iou_sparse_dense) and
:201-231 (compute_l1_tgtsurf_sparse_dense), called from torch/train.py:271-297, 353-378.

The reference moves every level's prediction to the host and intersects index sets with numpy; here a level is two
launches (`sgnn_iou_counts`) and the result stays on the device until the caller wants the number.  Same names and
argument meaning as the reference for the list-per-sample API, plus `iou_counts` for the training loop, which
takes a level's unfiltered `[locs, logits]` straight from the model output (no per-sample splitting at all).
"""
import numpy as np
import torch

from . import _lib
from .loss import UNK_ID  # noqa: F401  (re-exported for callers that mirror loss.py)


def _device_only(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.SgnnError('sgnn_amd.metrics runs on the GPU only (got a %s tensor); the CPU restatement lives '
                                 'in oracle/ and is test infrastructure' % t.device)
    _lib.require_gpu()


def _dense_args(dense_tgts):
    assert dense_tgts.dim() == 5 and dense_tgts.shape[1] == 1
    assert dense_tgts.dtype in (torch.float32, torch.uint8)
    nb, _, d0, d1, d2 = (int(v) for v in dense_tgts.shape)
    return dense_tgts.contiguous(), nb, d0, d1, d2


def iou_counts(locs, dense_tgts, use_loss_masking, logits=None, keep=None):
    """Device int64 (B,3) = per sample {P, C, T} (see include/sgnn_hip.h: sgnn_iou_counts).  No host sync.

    locs (M,4) int64 [z,y,x,b]; rows count if keep[r] (uint8/bool), else if sigmoid(logits[r]) > 0.5, else all."""
    _device_only(locs, dense_tgts, logits, keep)
    tgt, nb, d0, d1, d2 = _dense_args(dense_tgts)
    locs = locs.contiguous()
    m = int(locs.shape[0])
    lstride = 0
    if keep is not None:
        keep = keep.to(torch.uint8).contiguous()
    elif logits is not None:
        assert logits.dtype == torch.float32
        if logits.dim() == 2:
            lstride = int(logits.stride(0))
            assert logits.stride(1) == 1
        else:
            logits = logits.contiguous()
            lstride = 1
    counters = torch.empty((nb, 3), dtype=torch.int64, device=tgt.device)
    _lib.call('sgnn_iou_counts', _lib.ptr(locs), _lib.ptr(keep), _lib.ptr(logits), lstride, m, _lib.ptr(tgt),
              int(tgt.dtype == torch.uint8), nb, d0, d1, d2, int(bool(use_loss_masking)), _lib.ptr(counters))
    return counters


def iou_from_counts(counters, batched=True):
    """loss.py:109-119: intersection / union, -1 for an empty union (batched); per-sample array otherwise."""
    c = counters.cpu().numpy().astype(np.float64)
    corr, union = c[:, 1], c[:, 0] + c[:, 2] - c[:, 1]
    if not batched:
        with np.errstate(divide='ignore', invalid='ignore'):
            return np.divide(corr.astype(np.float32), union.astype(np.float32))
    return float(corr.sum() / union.sum()) if union.sum() > 0 else -1


def compute_iou_sparse_dense(sparse_pred_locs, dense_tgts, use_loss_masking, truncation=3, batched=True):
    """loss.py:84-120.  sparse_pred_locs: per sample an (n,3) [z,y,x] tensor of predicted-occupied sites or None."""
    parts = [torch.cat([p.long(), torch.full((p.shape[0], 1), b, dtype=torch.long, device=p.device)], 1)
             for b, p in enumerate(sparse_pred_locs) if p is not None]
    locs = torch.cat(parts) if parts else torch.zeros((0, 4), dtype=torch.long, device=dense_tgts.device)
    counters = iou_counts(locs, dense_tgts, use_loss_masking)
    have = torch.tensor([p is not None for p in sparse_pred_locs], device=counters.device)
    counters = counters * have[:, None]          # loss.py:91-92 `continue`: a skipped sample adds nothing (0/0 unbatched)
    return iou_from_counts(counters, batched=batched)


def pred_occs_from_outputs(output_occs, batch_size):
    """train.py:271-284: per level, per sample, the sites with sigmoid(occ) > 0.5 (reference list layout)."""
    pred = [None] * len(output_occs)
    for h, (locs, vals) in enumerate(output_occs):
        pred[h] = [None] * batch_size
        if len(locs) == 0:
            continue
        keep = torch.sigmoid(vals[:, 0].detach()) > 0.5
        for b in range(batch_size):
            pred[h][b] = locs[(locs[:, -1] == b) & keep][:, :-1]
    return pred


def level_ious(output_occs, target_for_occs, use_loss_masking):
    """IoU of every level straight from the model output (what train.py:271-290 computes through pred_occs):
    one `iou_counts` per level, one host read-back for all levels.  Empty levels give None (train.py:276-277
    leaves pred_occs[h][b] = None and compute_iou then returns -1)."""
    counts = []
    for (locs, vals), tgt in zip(output_occs, target_for_occs):
        counts.append(None if len(locs) == 0 else iou_counts(locs, tgt, use_loss_masking, logits=vals))
    return [-1 if c is None else iou_from_counts(c) for c in counts]


def compute_l1_tgtsurf_sparse_dense(sparse_pred_locs, sparse_pred_vals, dense_tgts, truncation, use_loss_masking,
                                    known, batched=True, thresh=None):
    """loss.py:201-231: mean |pred - target| over the target's surface voxels (prediction = -truncation where
    nothing was predicted)."""
    _device_only(sparse_pred_locs, sparse_pred_vals, dense_tgts, known if use_loss_masking else None)
    assert dense_tgts.dtype == torch.float32
    tgt, nb, d0, d1, d2 = _dense_args(dense_tgts)
    if not batched and nb != 1:
        raise ValueError('unbatched target-surface L1 is defined for batch size 1 only (loss.py:229-232)')
    locs = sparse_pred_locs.contiguous()
    vals = sparse_pred_vals.reshape(-1).float().contiguous()
    kn = known.contiguous() if use_loss_masking else None
    assert kn is None or kn.dtype == torch.uint8
    out = torch.empty(3, dtype=torch.float64, device=tgt.device)
    wsb = _lib.query('sgnn_l1_tgtsurf_ws_bytes')
    ws = torch.empty(wsb, dtype=torch.uint8, device=tgt.device)
    _lib.call('sgnn_l1_tgtsurf', _lib.ptr(locs), _lib.ptr(vals), int(locs.shape[0]), _lib.ptr(tgt), _lib.ptr(kn), nb,
              d0, d1, d2, float(truncation), -1.0 if thresh is None else float(thresh), _lib.ptr(out), _lib.ptr(ws),
              wsb)
    v = float(out[2].item())
    return v if batched else np.array([v], dtype=np.float32)
