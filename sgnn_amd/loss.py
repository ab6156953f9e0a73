"""Targets and hierarchical loss — counterpart of torch/loss.py:15-32 (compute_targets), :35-48
(compute_weights_missing_geo), :58-82 (BCE sparse-vs-dense), :122-157 (log-L1), :160-199 (compute_loss),
and torch/data_util.py:151-154 (preprocess_sdf_pt).  batched=True semantics only (what train.py:262-266
uses).  Device-agnostic torch ops: this is the caller of the hot path (SURVEY.md §8 row a-H); fusing it
into HIP kernels is row f1 ("next").
"""
import torch
import torch.nn.functional as F

UNK_THRESH = 2
UNK_ID = -1


def preprocess_sdf_pt(sdf, truncation):
    return sdf.clamp_(-truncation, truncation)


def compute_targets(target, hierarchy, num_hierarchy_levels, truncation, use_loss_masking, known):
    assert target.dim() == 5
    L = num_hierarchy_levels
    target_for_occs, target_for_hier = [None] * L, [None] * L
    target_for_sdf = preprocess_sdf_pt(target, truncation)
    target_for_hier[-1] = target.clone()
    occ = (torch.abs(target_for_sdf) < truncation).float()
    if use_loss_masking:
        occ = torch.where(known >= UNK_THRESH, torch.full_like(occ, UNK_ID), occ)   # select, not a nonzero + scatter (no sync)
    target_for_occs[-1] = occ
    for h in range(L - 2, -1, -1):
        target_for_occs[h] = F.max_pool3d(target_for_occs[h + 1], kernel_size=2)
        target_for_hier[h] = preprocess_sdf_pt(hierarchy[h], truncation)
    return target_for_sdf, target_for_occs, target_for_hier


def _targets_fusable(target, hierarchy, L, known, use_loss_masking, input_locs=None):
    if not (FUSED and target.is_cuda and target.dtype == torch.float32 and target.dim() == 5 and 1 <= L <= 4):
        return False
    if any(int(d) % (1 << (L - 1)) for d in target.shape[2:]) or target.shape[1] != 1:
        return False
    if use_loss_masking and (known is None or known.dtype != torch.uint8 or not known.is_cuda):
        return False
    if input_locs is not None and (input_locs.dtype != torch.int64 or input_locs.device != target.device):
        return False
    return all(h.is_cuda and h.dtype == torch.float32 for h in hierarchy[:L - 1])


def compute_targets_and_weights(target, hierarchy, num_hierarchy_levels, truncation, use_loss_masking, known,
                                weight_missing_geo=1.0, input_locs=None):
    """compute_targets + compute_weights_missing_geo (loss.py:15-48) -> ((tsdf, occs, hiers), weights or None).
    On the device this is sgnn_loss_targets: three launches, inputs left untouched (no clones needed)."""
    L = num_hierarchy_levels
    want_w = weight_missing_geo > 1
    if not _targets_fusable(target, hierarchy, L, known, use_loss_masking, input_locs if want_w else None):
        t = compute_targets(target.clone(), [h.clone() for h in hierarchy], L, truncation, use_loss_masking, known)
        w = compute_weights_missing_geo(weight_missing_geo, input_locs, t[1], truncation) if want_w else None
        return t, w
    from . import _lib
    import numpy as np
    target = target.contiguous()
    B, _, d0, d1, d2 = (int(v) for v in target.shape)
    dev = target.device
    new = lambda k: torch.empty(B, 1, d0 >> k, d1 >> k, d2 >> k, dtype=torch.float32, device=dev)
    tsdf, occs, hiers = new(0), [None] * L, [None] * L
    weights = [None] * L if want_w else None
    occs[-1], hiers[-1] = new(0), new(0)
    if want_w:
        weights[-1] = new(0)
    arr = {'hin': [], 'occ': [], 'w': [], 'hier': []}
    held = []
    for k in range(1, L):
        h = L - 1 - k
        occs[h], hiers[h] = new(k), new(k)
        hin = hierarchy[h].contiguous()
        assert tuple(hin.shape) == tuple(occs[h].shape), 'hierarchy level %d has shape %s' % (h, tuple(hin.shape))
        held.append(hin)
        arr['hin'].append(hin.data_ptr())
        arr['occ'].append(occs[h].data_ptr())
        arr['hier'].append(hiers[h].data_ptr())
        if want_w:
            weights[h] = new(k)
            arr['w'].append(weights[h].data_ptr())
        else:
            arr['w'].append(0)
    ptrs = dict((k, np.ascontiguousarray(np.array(v + [0], dtype=np.uint64))) for k, v in arr.items())
    locs = input_locs.contiguous() if want_w else None
    kn = known.contiguous() if use_loss_masking else None
    n_cnt = getattr(input_locs, '_sgnn_cnt', None) if want_w else None     # capacity mode: live rows of input_locs
    _lib.call('sgnn_loss_targets', _lib.ptr(target), _lib.ptr(kn), _lib.ptr(locs), 0 if locs is None else int(locs.shape[0]),
              _lib.ptr(n_cnt), B, d0, d1, d2, float(truncation), int(bool(use_loss_masking)), float(weight_missing_geo), L - 1,
              ptrs['hin'].ctypes.data, _lib.ptr(tsdf), _lib.ptr(hiers[-1]), _lib.ptr(occs[-1]),
              _lib.ptr(weights[-1]) if want_w else None, ptrs['occ'].ctypes.data, ptrs['w'].ctypes.data,
              ptrs['hier'].ctypes.data)
    return (tsdf, occs, hiers), weights


def _flat(locs, dims):
    return ((locs[:, 3] * dims[0] + locs[:, 0]) * dims[1] + locs[:, 1]) * dims[2] + locs[:, 2]


def compute_weights_missing_geo(weight_missing_geo, input_locs, target_for_occs, truncation):
    L = len(target_for_occs)
    weights = [None] * L
    dims = target_for_occs[-1].shape[2:]
    w = torch.ones(target_for_occs[-1].shape, dtype=torch.int32, device=target_for_occs[-1].device)
    w.view(-1)[_flat(input_locs.to(w.device), dims)] += 1
    w = w + 3 * (torch.abs(target_for_occs[-1]) <= truncation).to(torch.int32)
    weights[-1] = (w == 4).float() * (weight_missing_geo - 1) + 1
    for h in range(L - 2, -1, -1):
        weights[h] = weights[h + 1][:, :, ::2, ::2, ::2].contiguous()
    return weights


def apply_log_transform(sdf):
    return torch.sign(sdf) * torch.log(torch.abs(sdf) + 1)


def _masked_mean(values, mask):
    """mean(values[mask]) written as a ratio of sums: identical value (0/0 = nan for an empty mask, like the
    mean of an empty selection) but no boolean-mask indexing, i.e. no device->host sync inside the step."""
    m = mask.to(values.dtype)
    return (values * m).sum() / m.sum()


def compute_bce_sparse_dense(sparse_pred_locs, sparse_pred_vals, dense_tgts, weights, use_loss_masking):
    assert dense_tgts.dim() == 5 and dense_tgts.shape[1] == 1
    fl = _flat(sparse_pred_locs, dense_tgts.shape[2:])
    pred, tgt = sparse_pred_vals.reshape(-1), dense_tgts.view(-1)[fl]
    w = None if weights is None else weights.view(-1)[fl]
    if use_loss_masking:   # loss.py:67-72: drop UNK_ID targets
        per = F.binary_cross_entropy_with_logits(pred, tgt.clamp(min=0), weight=w, reduction='none')
        return _masked_mean(per, tgt != UNK_ID)
    tgt = torch.where(tgt == UNK_ID, torch.zeros_like(tgt), tgt)
    return F.binary_cross_entropy_with_logits(pred, tgt, weight=w)


def compute_l1_predsurf_sparse_dense(sparse_pred_locs, sparse_pred_vals, dense_tgts, weights, use_log_transform,
                                     use_loss_masking, known):
    assert dense_tgts.dim() == 5 and dense_tgts.shape[1] == 1
    fl = _flat(sparse_pred_locs, dense_tgts.shape[2:])
    pred, tgt = sparse_pred_vals.reshape(-1), dense_tgts.view(-1)[fl]
    w = None if weights is None else weights.view(-1)[fl]
    if use_log_transform:
        pred, tgt = apply_log_transform(pred), apply_log_transform(tgt)
    d = torch.abs(pred - tgt)
    if w is not None:
        d = d * w
    if use_loss_masking:   # loss.py:132-138
        return _masked_mean(d, (known < UNK_THRESH).view(-1)[fl])
    return torch.mean(d)


FUSED = True   # use the HIP level-loss kernels (sgnn_loss_level_*) for device tensors; False: torch ops below


class _LevelLoss(torch.autograd.Function):
    """(bce_mean, l1_mean) of one level through sgnn_loss_level_fwd/bwd (include/sgnn_hip.h)."""

    @staticmethod
    def forward(ctx, vals, locs, tgt_occ, tgt_sdf, weights, known, occ_col, sdf_col, use_log, mask_mode):
        from . import _lib
        from .scn.metadata import runtime
        vals = vals.contiguous()
        locs = locs.contiguous()
        m, vstride = vals.shape
        dims = tgt_sdf.shape[2:]
        rt = runtime(vals.device)
        sums = torch.empty(3, dtype=torch.float64, device=vals.device)
        out2 = torch.empty(2, dtype=torch.float32, device=vals.device)
        wsb = _lib.query('sgnn_loss_ws_bytes')
        ws = rt.workspace(wsb)
        m_cnt = getattr(locs, '_sgnn_cnt', None)       # capacity mode: live row count (device int64[1])
        args = (_lib.ptr(locs), _lib.ptr(vals), vstride, occ_col, sdf_col, _lib.ptr(tgt_occ), _lib.ptr(tgt_sdf),
                _lib.ptr(weights), _lib.ptr(known), int(dims[0]), int(dims[1]), int(dims[2]), m, int(use_log),
                mask_mode, _lib.ptr(m_cnt))
        _lib.call('sgnn_loss_level_fwd', *args, _lib.ptr(sums), _lib.ptr(out2), _lib.ptr(ws), wsb)
        ctx.args = args
        ctx.keep = (locs, vals, tgt_occ, tgt_sdf, weights, known, sums, m_cnt)
        return out2

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        locs, vals, tgt_occ, tgt_sdf, weights, known, sums, _m_cnt = ctx.keep
        g = g.contiguous()
        dvals = torch.empty_like(vals)
        _lib.call('sgnn_loss_level_bwd', *ctx.args, _lib.ptr(sums), _lib.ptr(g), _lib.ptr(dvals))
        return (dvals,) + (None,) * 9


class _TotalLoss(torch.autograd.Function):
    """The whole hierarchical loss (loss.py:160-199) as one autograd node: per level sgnn_loss_level_fwd, then
    sgnn_loss_combine for  sum_l weight_l * (bce_l + l1_l);  backward fans the incoming gradient out to the levels
    (sgnn_loss_combine_bwd) and runs sgnn_loss_level_bwd per level.  `levels` = list of dicts (locs, tgt_occ, tgt_sdf,
    weights, known, occ_col, sdf_col, mask_mode, coef (bce, l1)); vals_l are the differentiable inputs."""

    @staticmethod
    def forward(ctx, levels, use_log, *vals):
        from . import _lib
        from .scn.metadata import runtime
        import numpy as np
        dev = vals[0].device
        rt = runtime(dev)
        n = len(levels)
        out2s = torch.empty(2 * n, dtype=torch.float32, device=dev)
        sums = torch.empty(n, 3, dtype=torch.float64, device=dev)
        wsb = _lib.query('sgnn_loss_multi_ws_bytes')
        ws = rt.workspace(wsb)
        desc = np.zeros((n, 16), dtype=np.int64)
        coef, held = [], []
        p = lambda t: 0 if t is None else t.data_ptr()
        for l, (lv, v) in enumerate(zip(levels, vals)):
            v = v.contiguous()
            m_cnt = getattr(lv['locs'], '_sgnn_cnt', None)       # capacity mode: live row count (device int64[1])
            locs = lv['locs'].contiguous()
            dims = lv['tgt_sdf'].shape[2:]
            m, vstride = v.shape
            desc[l] = [p(locs), p(v), vstride, lv['occ_col'], lv['sdf_col'], p(lv['tgt_occ']), p(lv['tgt_sdf']),
                       p(lv['weights']), p(lv['known']), int(dims[0]), int(dims[1]), int(dims[2]), m, int(use_log),
                       lv['mask_mode'], p(m_cnt)]
            held.append((locs, v, m_cnt))
            coef += [float(lv['coef'][0]), float(lv['coef'][1])]
        desc = np.ascontiguousarray(desc)
        coef_np = np.ascontiguousarray(np.array(coef, dtype=np.float32))
        total = torch.empty((), dtype=torch.float32, device=dev)
        cur = torch.empty(n, dtype=torch.float32, device=dev)
        _lib.call('sgnn_loss_levels_fwd', desc.ctypes.data, n, coef_np.ctypes.data, sums.data_ptr(), out2s.data_ptr(),
                  total.data_ptr(), cur.data_ptr(), _lib.ptr(ws), wsb)
        ctx.keep = (levels, held, desc, sums, coef_np)
        ctx.mark_non_differentiable(cur)
        return total, cur

    @staticmethod
    def backward(ctx, g, _g_cur):
        from . import _lib
        import numpy as np
        levels, held, desc, sums, coef_np = ctx.keep
        n = len(levels)
        g = g.contiguous().view(1)
        grads = [torch.empty_like(h[1]) for h in held]
        ptrs = np.ascontiguousarray(np.array([t.data_ptr() for t in grads] + [0], dtype=np.uint64))
        _lib.call('sgnn_loss_levels_bwd', desc.ctypes.data, n, coef_np.ctypes.data, sums.data_ptr(), g.data_ptr(),
                  ptrs.ctypes.data)
        return (None, None) + tuple(grads)


def _fusable(vals, *dense):
    return (FUSED and torch.is_tensor(vals) and vals.is_cuda and vals.dtype == torch.float32 and
            all(d is None or (d.is_cuda and d.is_contiguous()) for d in dense))


def _compute_loss_fused(output_sdf, output_occs, target_for_sdf, target_for_occs, target_for_hier, loss_weights,
                        use_log_transform, use_loss_masking, known, weights):
    """compute_loss through _TotalLoss when every active level is fusable; None otherwise."""
    levels, vals, slot = [], [], []
    for h in range(len(output_occs)):
        if len(output_occs[h][0]) == 0 or loss_weights[h] == 0:
            slot.append(None)
            continue
        locs, v = output_occs[h]
        if not (_fusable(v, target_for_occs[h], target_for_hier[h], weights[h]) and
                target_for_occs[h].dtype == torch.float32):
            return None
        lw = float(loss_weights[h])
        levels.append(dict(locs=locs, tgt_occ=target_for_occs[h], tgt_sdf=target_for_hier[h], weights=weights[h],
                           known=None, occ_col=0, sdf_col=1, mask_mode=1 if use_loss_masking else 0, coef=(lw, lw)))
        vals.append(v)
        slot.append(len(levels) - 1)
    if len(output_sdf[0]) > 0 and loss_weights[-1] > 0:
        v = output_sdf[1]
        if not (_fusable(v, target_for_sdf, weights[-1], known if use_loss_masking else None) and
                (not use_loss_masking or known.dtype == torch.uint8)):
            return None
        levels.append(dict(locs=output_sdf[0], tgt_occ=None, tgt_sdf=target_for_sdf, weights=weights[-1],
                           known=known if use_loss_masking else None, occ_col=-1, sdf_col=0,
                           mask_mode=2 if use_loss_masking else 0, coef=(0.0, float(loss_weights[-1]))))
        vals.append(v)
        slot.append(len(levels) - 1)
    else:
        slot.append(None)
    if not levels or len(levels) > 5:
        return None
    total, cur = _TotalLoss.apply(levels, bool(use_log_transform), *vals)
    losses = [-1 if k is None else cur[k] for k in slot]
    return total, losses


def compute_loss(output_sdf, output_occs, target_for_sdf, target_for_occs, target_for_hier, loss_weights, truncation,
                 use_log_transform=True, weight_missing_geo=1, input_locs=None, use_loss_masking=True, known=None,
                 weights=None):
    """Returns (loss tensor, per-level loss tensors or -1).  Unlike loss.py:185 the per-level values are
    left on the device (no .item() sync inside the step); callers convert when they log.
    `weights`: the result of compute_weights_missing_geo if the caller already has it (train_step computes targets
    and weights on a second stream while the encoder runs)."""
    assert len(output_occs) == len(target_for_occs)
    loss, losses = 0.0, []
    if weights is None and weight_missing_geo <= 1:
        weights = [None] * len(target_for_occs)
    if weights is not None:
        fused = _compute_loss_fused(output_sdf, output_occs, target_for_sdf, target_for_occs, target_for_hier,
                                    loss_weights, use_log_transform, use_loss_masking, known, weights)
        if fused is not None:
            return fused
    if weights is None:
        weights = [None] * len(target_for_occs)
        if weight_missing_geo > 1:
            weights = compute_weights_missing_geo(weight_missing_geo, input_locs, target_for_occs, truncation)
    for h in range(len(output_occs)):
        if len(output_occs[h][0]) == 0 or loss_weights[h] == 0:
            losses.append(-1)
            continue
        locs, vals = output_occs[h]
        if _fusable(vals, target_for_occs[h], target_for_hier[h], weights[h]) and target_for_occs[h].dtype == torch.float32:
            both = _LevelLoss.apply(vals, locs, target_for_occs[h], target_for_hier[h], weights[h], None, 0, 1,
                                    use_log_transform, 1 if use_loss_masking else 0)
            cur = both[0] + both[1]
        else:
            l_occ = compute_bce_sparse_dense(locs, vals[:, 0], target_for_occs[h], weights[h], use_loss_masking)
            cur_known = None if not use_loss_masking else (target_for_occs[h] == UNK_ID) * UNK_THRESH
            l_sdf = compute_l1_predsurf_sparse_dense(locs, vals[:, 1], target_for_hier[h], weights[h],
                                                     use_log_transform, use_loss_masking, cur_known)
            cur = l_occ + l_sdf
        loss = loss + float(loss_weights[h]) * cur
        losses.append(cur.detach())
    if len(output_sdf[0]) > 0 and loss_weights[-1] > 0:
        vals = output_sdf[1]
        if _fusable(vals, target_for_sdf, weights[-1], known if use_loss_masking else None) and \
                (not use_loss_masking or known.dtype == torch.uint8):
            cur = _LevelLoss.apply(vals, output_sdf[0], None, target_for_sdf, weights[-1],
                                   known if use_loss_masking else None, -1, 0, use_log_transform,
                                   2 if use_loss_masking else 0)[1]
        else:
            cur = compute_l1_predsurf_sparse_dense(output_sdf[0], vals, target_for_sdf, weights[-1],
                                                   use_log_transform, use_loss_masking, known)
        loss = loss + float(loss_weights[-1]) * cur
        losses.append(cur.detach())
    else:
        losses.append(-1)
    return loss, losses
