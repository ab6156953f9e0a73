"""torch.autograd wrappers around the C ABI (include/sgnn_hip.h).

Each Function allocates outputs with torch (device memory plumbing) and hands raw device
pointers to libsgnn_hip.so on the current HIP stream.  No arithmetic happens in Python/torch.
"""
import torch
from torch.autograd import Function

from .. import _lib
from .._lib import ptr
from .metadata import runtime

import os

CONV_TRANSPOSE_W = 1
CONV_FLIP_K = 2
# Round 5: the glue between two generative stages with fewer launches (kept coordinates written by the compaction's write
# kernel, children + their int64 rows in one pass).  SGNN_FUSED_GLUE=0 restores the separate launches (A/B, parity test).
FUSED_GLUE = os.environ.get('SGNN_FUSED_GLUE', '1') != '0'


_ws_cache = {}


def _ws_bytes(name, *args):
    """Workspace-size queries are pure functions of their arguments: memoised to keep ctypes off the hot path."""
    k = (name,) + args
    v = _ws_cache.get(k)
    if v is None:
        if len(_ws_cache) > 4096:
            _ws_cache.clear()
        v = _ws_cache[k] = _lib.query(name, *args)
    return v


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError('sgnn_amd operators are float32 (got %s)' % t.dtype)
    return t.contiguous()


def conv_fwd_raw(x, cin, w, K, table, ld, n_out, cout, flags=0, in_shift=0):
    y = torch.empty(n_out, cout, dtype=torch.float32, device=x.device)
    _lib.call('sgnn_conv_fwd', ptr(x), x.shape[0], cin, ptr(w), K, ptr(table), ld, n_out, cout, ptr(y), flags,
              in_shift)
    return y


_SPLIT_SHAPES = {(16, 24), (24, 16), (24, 32), (32, 24), (64, 32), (32, 64), (56, 28), (28, 56)}
_arange_cache = {}


def conv_fwd_split(x, cin, w, K, table, ld, n_out, cout, flags=0):
    """Same result as conv_fwd_raw up to summation order, for few rows and many offsets (the dense bottleneck's
    k4s2 convolutions: 256..16 384 rows, 64 taps): the taps are cut into G slices that run as the `groups` of
    sgnn_conv_fwd_ex — G times the workgroups, 1/G of the serial offset walk — and sgnn_sum_groups adds the slices."""
    tiles = (n_out + 63) // 64
    G = 1
    while G * 2 <= K and K % (G * 2) == 0 and tiles * G < 1024:
        G *= 2
    if G == 1 or (cin, cout) not in _SPLIT_SHAPES:
        return conv_fwd_raw(x, cin, w, K, table, ld, n_out, cout, flags, 0)
    key = (str(x.device), K)
    kmap = _arange_cache.get(key)
    if kmap is None:
        kmap = _arange_cache[key] = torch.arange(K, dtype=torch.int32, device=x.device)
    part = torch.empty(n_out, G * cout, dtype=torch.float32, device=x.device)
    _lib.call('sgnn_conv_fwd_ex', ptr(x), x.shape[0], cin, ptr(w), K // G, ptr(table), ld, n_out, cout, ptr(part), flags,
              0, ptr(kmap), None, 1, G, K)
    return sum_groups_raw(part, cout, n_out, G)


def conv_dw_raw(x, cin, dy, cout, table, ld, K, n_out, in_shift=0):
    rt = runtime(x.device)
    dw = torch.empty(K, cin, cout, dtype=torch.float32, device=x.device)
    wsb = _lib.query('sgnn_conv_bwd_weight_ws_bytes', n_out, K, cin, cout)
    ws = rt.workspace(wsb)
    _lib.call('sgnn_conv_bwd_weight', ptr(x), x.shape[0], cin, ptr(dy), cout, ptr(table), ld, K, n_out, ptr(dw),
              in_shift, ptr(ws), wsb)
    return dw


class SparseConv(Function):
    """y[j] = sum_k W[k]^T x[table_f[k][j]]; gradients through table_b (see sgnn_hip.h)."""

    @staticmethod
    def forward(ctx, x, weight, table_f, ld_f, n_out, table_b, ld_b, n_in, flags_b, in_shift):
        x, weight = _f32c(x), _f32c(weight)
        K, cin, cout = weight.shape
        assert x.shape[1] == cin, 'feature width %d != nIn %d' % (x.shape[1], cin)
        if K == 64 and not in_shift:
            y = conv_fwd_split(x, cin, weight, K, table_f, ld_f, n_out, cout, 0)
        else:
            y = conv_fwd_raw(x, cin, weight, K, table_f, ld_f, n_out, cout, 0, in_shift)
        ctx.save_for_backward(x, weight)
        ctx.tables = (table_f, ld_f, n_out, table_b, ld_b, n_in, flags_b, in_shift)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        table_f, ld_f, n_out, table_b, ld_b, n_in, flags_b, in_shift = ctx.tables
        K, cin, cout = weight.shape
        dy = _f32c(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if in_shift:
                # gradient w.r.t. the virtual (replicated) rows, then summed over each group
                dfull = conv_fwd_raw(dy, cout, weight, K, table_b, ld_b, n_in, cin, flags_b, 0)
                dx = sum_groups_raw(dfull, cin, n_in >> in_shift, 1 << in_shift)
            elif K == 64:
                dx = conv_fwd_split(dy, cout, weight, K, table_b, ld_b, n_in, cin, flags_b)
            else:
                dx = conv_fwd_raw(dy, cout, weight, K, table_b, ld_b, n_in, cin, flags_b, 0)
        if ctx.needs_input_grad[1]:
            dw = conv_dw_raw(x, cin, dy, cout, table_f, ld_f, K, n_out, in_shift)
        return dx, dw, None, None, None, None, None, None, None, None


class DenseK4S2(Function):
    """nn.Conv3d(k4,s2,p1) (down) / nn.ConvTranspose3d(k4,s2,p1) (up) between two dense pyramid levels on channel-last rows in
    batch-major raster order (torch/model.py:89-136).  weight: (64, Cin, Cout), the taps in parity-group
    slot order (model.K4S2_TAPS); lvl: model._DenseGeometry.level().

    The coarse side of either layer sees all 64 taps (Conv3d forward, ConvTranspose3d data gradient, Conv3d weight
    gradient): a 64-offset rulebook walk over the coarse rows (tdown), cut into tap slices that run side by side
    (conv_fwd_split).  The FINE side sees 8 taps per voxel — which 8 is decided by the voxel's parity — so walking a
    64-row table per fine voxel (tup) issues 56 empty gathers and MFMA blocks for every 8 that carry data.  With
    `parity` the fine side runs on the COARSE rulebook instead, as 8 parity groups of 8 taps (sgnn_conv_fwd_ex /
    sgnn_conv_bwd_weight_ex with groups = 8 — the walk of the generative up-sampling convolution, ExpandConv): rows come
    out / go in in child order (8*coarse + parity) and one 16-byte-row gather converts to / from raster order.
    Measured (profiles/r05u_*): ConvTranspose3d(56 -> 28) forward at 2048 -> 16384 rows and its weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, lvl, down, parity):
        x, weight = _f32c(x), _f32c(weight)
        K, cin, cout = weight.shape
        assert K == 64 and x.shape[1] == cin, 'k4/s2 layer: weight %s, rows %s' % (tuple(weight.shape), tuple(x.shape))
        if down:
            assert x.shape[0] == lvl.n_f
            y = conv_fwd_split(x, cin, weight, 64, lvl.tdown, lvl.ld_c, lvl.n_c, cout, 0)
        elif parity:
            assert x.shape[0] == lvl.n_c
            y8 = torch.empty(8 * lvl.n_c, cout, dtype=torch.float32, device=x.device)
            _lib.call('sgnn_conv_fwd_ex', ptr(x), lvl.n_c, cin, ptr(weight), 8, ptr(lvl.nbr27), lvl.ld_c, lvl.n_c, cout,
                      ptr(y8), 0, 0, ptr(lvl.slots), None, 1, 8, 27)
            y = gather_rows_raw(y8, cout, lvl.child_of_raster, lvl.n_f)
        else:
            y = conv_fwd_split(x, cin, weight, 64, lvl.tup, lvl.ld_f, lvl.n_f, cout, 0)
        ctx.save_for_backward(x, weight)
        ctx.cfg = (lvl, down, parity)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        lvl, down, parity = ctx.cfg
        _, cin, cout = weight.shape
        dy = _f32c(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if not down:         # coarse side: all 64 taps
                dx = conv_fwd_split(dy, cout, weight, 64, lvl.tdown, lvl.ld_c, lvl.n_c, cin, CONV_TRANSPOSE_W)
            elif parity:
                dx8 = torch.empty(8 * lvl.n_c, cin, dtype=torch.float32, device=dy.device)
                _lib.call('sgnn_conv_fwd_ex', ptr(dy), lvl.n_c, cout, ptr(weight), 8, ptr(lvl.nbr27), lvl.ld_c, lvl.n_c,
                          cin, ptr(dx8), CONV_TRANSPOSE_W, 0, ptr(lvl.slots), None, 1, 8, 27)
                dx = gather_rows_raw(dx8, cin, lvl.child_of_raster, lvl.n_f)
            else:
                dx = conv_fwd_split(dy, cout, weight, 64, lvl.tup, lvl.ld_f, lvl.n_f, cin, CONV_TRANSPOSE_W)
        if ctx.needs_input_grad[1]:
            if down:
                dw = conv_dw_raw(x, cin, dy, cout, lvl.tdown, lvl.ld_c, 64, lvl.n_c)
            elif parity:
                dy8 = gather_rows_raw(dy, cout, lvl.raster_of_child, lvl.n_f)
                rt = runtime(x.device)
                dw = torch.empty_like(weight)
                wsb = _lib.query('sgnn_conv_bwd_weight_ws_bytes', lvl.n_c, 64, cin, cout)
                ws = rt.workspace(wsb)
                _lib.call('sgnn_conv_bwd_weight_ex', ptr(x), lvl.n_c, cin, ptr(dy8), cout, ptr(lvl.nbr27), lvl.ld_c, 8,
                          lvl.n_c, ptr(dw), 0, ptr(lvl.slots), None, 1, 8, 27, ptr(ws), wsb)
            else:
                dw = conv_dw_raw(x, cin, dy, cout, lvl.tup, lvl.ld_f, 64, lvl.n_f)
        return dx, dw, None, None, None


_expand_cache = {}


def expand_maps(device):
    """Constants of the generative up-sampling convolution (see sgnn_hip.h, sgnn_conv_fwd_ex).
    Child parity g = 4*jz+2*jy+jx, parent-level offset slot i = 4*iz+2*iy+ix with parent offset o = i - 1 + j
    per axis.  A (64, 27): Wc[g*8+i] = sum of the 3x3x3 taps whose neighbour child lies in that parent.
    S (64): parent-table row of (g, i);  ST = 26 - S (mirrored row, data gradient);  PAR (64): parity g."""
    key = str(device)
    m = _expand_cache.get(key)
    if m is None:
        taps = {(0, 0): (-1,), (0, 1): (0, 1), (1, 0): (-1, 0), (1, 1): (1,)}
        A = torch.zeros(64, 27)
        S, PAR = [], []
        for g in range(8):
            j = (g >> 2, (g >> 1) & 1, g & 1)
            for i_ in range(8):
                i = (i_ >> 2, (i_ >> 1) & 1, i_ & 1)
                o = [i[a] - 1 + j[a] for a in range(3)]
                S.append((o[0] + 1) * 9 + (o[1] + 1) * 3 + (o[2] + 1))
                PAR.append(g)
                for dz in taps[(j[0], i[0])]:
                    for dy in taps[(j[1], i[1])]:
                        for dx in taps[(j[2], i[2])]:
                            A[g * 8 + i_, (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1)] = 1.0
        assert float(A.sum()) == 8 * 27          # every child sees each of its 27 taps exactly once
        St = torch.tensor(S, dtype=torch.int32)
        m = _expand_cache[key] = (A.to(device), St.to(device), (26 - St).to(device),
                                  torch.tensor(PAR, dtype=torch.int32, device=device))
    return m


EXPAND_DX_SPLIT = 4


class ExpandConv(Function):
    """SubmanifoldConvolution(3x3x3) over the 8-child expansion of a level whose children all carry their parent's
    features (torch/model.py:192-207 followed by n0/n1, :220-222), evaluated on the PARENT rulebook:
        y[8p+g] = sum_i Wc[g][i]^T f[nbr[S[g][i]][p]],   Wc[g][i] = sum of the taps that fall into that parent.
    Same function as the reference's expand -> InputLayer -> SubmanifoldConvolution, with 64 instead of 216
    gathers per parent and no children grid / rulebook."""

    @staticmethod
    def forward(ctx, f, wc, table, ld, n):
        f, wc = _f32c(f), _f32c(wc)
        cin, cout = wc.shape[1], wc.shape[2]
        _, S, ST, PAR = expand_maps(f.device)
        y = torch.empty(8 * n, cout, dtype=torch.float32, device=f.device)
        _lib.call('sgnn_conv_fwd_ex', ptr(f), n, cin, ptr(wc), 8, ptr(table), ld, n, cout, ptr(y), 0, 0, ptr(S), None,
                  1, 8, 27)
        ctx.save_for_backward(f, wc)
        ctx.cfg = (table, ld, n)
        return y

    @staticmethod
    def backward(ctx, dy):
        f, wc = ctx.saved_tensors
        table, ld, n = ctx.cfg
        cin, cout = wc.shape[1], wc.shape[2]
        dy = _f32c(dy)
        _, S, ST, PAR = expand_maps(f.device)
        df = dwc = None
        if ctx.needs_input_grad[0]:
            # 64 offsets per parent row: cut into G slices that run as conv groups (G x the workgroups, 1/G of the
            # serial offset walk per workgroup), slices added by sgnn_sum_groups — as for the dense bottleneck
            G = EXPAND_DX_SPLIT
            part = torch.empty(n, G * cin, dtype=torch.float32, device=f.device)
            _lib.call('sgnn_conv_fwd_ex', ptr(dy), 8 * n, cout, ptr(wc), 64 // G, ptr(table), ld, n, cin, ptr(part),
                      CONV_TRANSPOSE_W, 0, ptr(ST), ptr(PAR), 8, G, 27)
            df = part if G == 1 else sum_groups_raw(part, cin, n, G)
        if ctx.needs_input_grad[1]:
            rt = runtime(f.device)
            dwc = torch.empty_like(wc)
            wsb = _lib.query('sgnn_conv_bwd_weight_ws_bytes', n, 64, cin, cout)
            ws = rt.workspace(wsb)
            _lib.call('sgnn_conv_bwd_weight_ex', ptr(f), n, cin, ptr(dy), cout, ptr(table), ld, 8, n, ptr(dwc), 0,
                      ptr(S), None, 1, 8, 27, ptr(ws), wsb)
        return df, dwc, None, None, None


class ExpandWeights(Function):
    """Wc (64, nIn, nOut) = the 3x3x3 taps pre-summed per (child parity, parent offset slot) — sgnn_expand_weights, the
    same kernel the native stage program uses (so both paths round identically)."""

    @staticmethod
    def forward(ctx, weight):
        weight = _f32c(weight)
        ctx.shape = tuple(weight.shape)
        wc = torch.empty(64, weight.shape[1], weight.shape[2], dtype=torch.float32, device=weight.device)
        _lib.call('sgnn_expand_weights', ptr(weight), weight.shape[1], weight.shape[2], ptr(wc))
        return wc

    @staticmethod
    def backward(ctx, dwc):
        dwc = _f32c(dwc)
        dw = torch.empty(ctx.shape, dtype=torch.float32, device=dwc.device)
        _lib.call('sgnn_expand_weights_bwd', ptr(dwc), ctx.shape[1], ctx.shape[2], ptr(dw))
        return dw


def expand_conv(f, weight, grid):
    """weight: the (27, nIn, nOut) parameter of the reference's n1 layer; grid: the parent level's Grid."""
    return ExpandConv.apply(f, ExpandWeights.apply(weight), grid.subm_table(), grid.ld, grid.n)


class BatchNormLeaky(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, training, leak):
        x = _f32c(x)
        n, c = x.shape
        rt = runtime(x.device)
        y = torch.empty_like(x)
        save = torch.empty(2, c, dtype=torch.float32, device=x.device)
        wsb = _ws_bytes('sgnn_bn_ws_bytes', 0, c)
        ws = rt.workspace(wsb)
        _lib.call('sgnn_bn_fwd', ptr(x), n, c, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                  float(eps), float(momentum), int(bool(training)), float(leak), ptr(save[0]), ptr(save[1]), ptr(y),
                  ptr(ws), wsb)
        ctx.save_for_backward(x, gamma, beta, save)
        ctx.cfg = (bool(training), float(leak))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, save = ctx.saved_tensors
        training, leak = ctx.cfg
        n, c = x.shape
        dy = _f32c(dy)
        rt = runtime(x.device)
        dx = torch.empty_like(x)
        # two tensors (not two rows of one): autograd can hand them to the parameters' .grad without cloning
        dg = torch.empty(c, dtype=torch.float32, device=x.device)
        db = torch.empty(c, dtype=torch.float32, device=x.device)
        wsb = _ws_bytes('sgnn_bn_ws_bytes', 0, c)
        ws = rt.workspace(wsb)
        _lib.call('sgnn_bn_bwd', ptr(x), ptr(dy), n, c, ptr(gamma), ptr(beta), ptr(save[0]), ptr(save[1]),
                  int(training), leak, ptr(dx), ptr(dg), ptr(db), ptr(ws), wsb)
        dgamma = dg if gamma is not None else None
        dbeta = db if beta is not None else None
        return dx, dgamma, dbeta, None, None, None, None, None, None


def gather_rows_raw(src, c, idx, m):
    dst = torch.empty(m, c, dtype=torch.float32, device=src.device)
    _lib.call('sgnn_gather_rows', ptr(src), c, ptr(idx), m, ptr(dst))
    return dst


def scatter_rows_raw(src, c, idx, m, n_dst, m_cnt=None):
    dst = torch.empty(n_dst, c, dtype=torch.float32, device=src.device)
    _lib.call('sgnn_scatter_rows', ptr(src), c, ptr(idx), m, ptr(dst), n_dst, ptr(m_cnt))
    return dst


def sum_groups_raw(src, c, n, rep):
    dst = torch.empty(n, c, dtype=torch.float32, device=src.device)
    _lib.call('sgnn_sum_groups', ptr(src), c, n, rep, ptr(dst))
    return dst


class GatherRows(Function):
    """dst[r] = src[idx[r]] with unique idx (mask compaction)."""

    @staticmethod
    def forward(ctx, src, idx, m):
        src = _f32c(src)
        ctx.save_for_backward(idx)
        ctx.shape = (src.shape[0], src.shape[1], m)
        return gather_rows_raw(src, src.shape[1], idx, m)

    @staticmethod
    def backward(ctx, d):
        (idx,) = ctx.saved_tensors
        n, c, m = ctx.shape
        return scatter_rows_raw(_f32c(d), c, idx, m, n), None, None


class ScatterRows(Function):
    """dst (n_dst rows, zeros elsewhere); dst[idx[r]] = src[r] with unique idx.  m_cnt: device row count of src
    (capacity mode)."""

    @staticmethod
    def forward(ctx, src, idx, n_dst, m_cnt=None):
        src = _f32c(src)
        ctx.save_for_backward(idx)
        ctx.shape = (src.shape[0], src.shape[1])
        ctx.m_cnt = m_cnt
        return scatter_rows_raw(src, src.shape[1], idx, src.shape[0], n_dst, m_cnt)

    @staticmethod
    def backward(ctx, d):
        (idx,) = ctx.saved_tensors
        m, c = ctx.shape
        d = _f32c(d)
        if ctx.m_cnt is None:
            return gather_rows_raw(d, c, idx, m), None, None, None
        out = torch.empty(m, c, dtype=torch.float32, device=d.device)
        _lib.call('sgnn_gather_rows_dn', ptr(d), c, ptr(idx), ptr(ctx.m_cnt), m, ptr(out))
        return out, None, None, None


class RowLinear(Function):
    """y = x W^T + b for the 1-2 output per-site heads (W (cout, cin), b (cout) or None)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w = _f32c(x), _f32c(w)
        n, cin = x.shape
        cout = w.shape[0]
        y = torch.empty(n, cout, dtype=torch.float32, device=x.device)
        _lib.call('sgnn_linear_fwd', ptr(x), n, cin, ptr(w), ptr(b), cout, ptr(y))
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        n, cin = x.shape
        cout = w.shape[0]
        dy = _f32c(dy)
        rt = runtime(x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty(cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
        wsb = _ws_bytes('sgnn_linear_ws_bytes', 0, cin, cout)
        ws = rt.workspace(wsb)
        _lib.call('sgnn_linear_bwd', ptr(x), ptr(dy), n, cin, ptr(w), cout, ptr(dx), ptr(dw), ptr(db), ptr(ws), wsb)
        return dx, dw, db


class UnPool(Function):
    """fine[i] = coarse[parent[i]]; backward = sum over the (<= 8) children."""

    @staticmethod
    def forward(ctx, coarse, parent, nf, children, ldc):
        coarse = _f32c(coarse)
        ctx.save_for_backward(children)
        ctx.shape = (coarse.shape[0], coarse.shape[1], ldc)
        return gather_rows_raw(coarse, coarse.shape[1], parent, nf)

    @staticmethod
    def backward(ctx, d):
        (children,) = ctx.saved_tensors
        nc, c, ldc = ctx.shape
        d = _f32c(d)
        out = torch.empty(nc, c, dtype=torch.float32, device=d.device)
        _lib.call('sgnn_gather_sum', ptr(d), c, ptr(children), ldc, 8, nc, ptr(out))
        return out, None, None, None, None


class RepeatRows(Function):
    @staticmethod
    def forward(ctx, src, rep):
        src = _f32c(src)
        n, c = src.shape
        ctx.cfg = (n, c, rep)
        dst = torch.empty(n * rep, c, dtype=torch.float32, device=src.device)
        _lib.call('sgnn_repeat_rows', ptr(src), c, n, rep, ptr(dst))
        return dst

    @staticmethod
    def backward(ctx, d):
        n, c, rep = ctx.cfg
        return sum_groups_raw(_f32c(d), c, n, rep), None


class ConcatRows(Function):
    """dst[r] = [a[ia[r]] | b[ib[r]]] (None index = identity, negative ib = zeros)."""

    @staticmethod
    def forward(ctx, a, ia, b, ib, m):
        a, b = _f32c(a), _f32c(b)
        ca, cb = a.shape[1], b.shape[1]
        dst = torch.empty(m, ca + cb, dtype=torch.float32, device=a.device)
        _lib.call('sgnn_concat_rows', ptr(a), ca, ptr(ia), ptr(b), cb, ptr(ib), m, ptr(dst))
        ctx.idx = (ia, ib)
        ctx.shape = (a.shape[0], ca, b.shape[0], cb, m)
        return dst

    @staticmethod
    def backward(ctx, d):
        ia, ib = ctx.idx
        na, ca, nb, cb, m = ctx.shape
        d = _f32c(d)
        da = torch.empty(na, ca, dtype=torch.float32, device=d.device) if ctx.needs_input_grad[0] else None
        db = torch.empty(nb, cb, dtype=torch.float32, device=d.device) if ctx.needs_input_grad[2] else None
        _lib.call('sgnn_concat_rows_bwd', ptr(d), ca, ptr(ia), cb, ptr(ib), m, ptr(da), na, ptr(db), nb)
        return da, None, db, None, None


class AddRows(Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _f32c(a), _f32c(b)
        y = torch.empty_like(a)
        _lib.call('sgnn_add', ptr(a), ptr(b), a.numel(), ptr(y))
        return y

    @staticmethod
    def backward(ctx, d):
        return d, d


class SparseToDenseFn(Function):
    @staticmethod
    def forward(ctx, feats, coords, batch, d0, d1, d2):
        feats = _f32c(feats)
        n, c = feats.shape
        dense = torch.empty(batch, c, d0, d1, d2, dtype=torch.float32, device=feats.device)
        _lib.call('sgnn_sparse_to_dense', ptr(feats), ptr(coords), n, c, ptr(dense), batch, d0, d1, d2)
        ctx.save_for_backward(coords)
        ctx.cfg = (n, c, batch, d0, d1, d2)
        return dense

    @staticmethod
    def backward(ctx, d):
        (coords,) = ctx.saved_tensors
        n, c, batch, d0, d1, d2 = ctx.cfg
        d = _f32c(d)
        df = torch.empty(n, c, dtype=torch.float32, device=d.device)
        _lib.call('sgnn_dense_to_sparse', ptr(d), ptr(coords), n, c, ptr(df), batch, d0, d1, d2)
        return df, None, None, None, None, None


class DenseToSparseFn(Function):
    """rows[r][ch] = dense[b][ch][z][y][x] at coords[r] (coords unique)."""

    @staticmethod
    def forward(ctx, dense, coords):
        dense = _f32c(dense)
        batch, c, d0, d1, d2 = dense.shape
        n = coords.shape[0]
        rows = torch.empty(n, c, dtype=torch.float32, device=dense.device)
        _lib.call('sgnn_dense_to_sparse', ptr(dense), ptr(coords), n, c, ptr(rows), batch, d0, d1, d2)
        ctx.save_for_backward(coords)
        ctx.cfg = (n, c, batch, d0, d1, d2)
        return rows

    @staticmethod
    def backward(ctx, d):
        (coords,) = ctx.saved_tensors
        n, c, batch, d0, d1, d2 = ctx.cfg
        d = _f32c(d)
        dd = torch.empty(batch, c, d0, d1, d2, dtype=torch.float32, device=d.device)
        _lib.call('sgnn_sparse_to_dense', ptr(d), ptr(coords), n, c, ptr(dd), batch, d0, d1, d2)
        return dd, None


# ---------------------------------------------------------------------------
# non-differentiable index helpers
# ---------------------------------------------------------------------------
def compact_sigmoid(logits, stride, n):
    """Stable list of rows with sigmoid(logit) > 0.5 -> (sel int32[count], count).  One host sync."""
    rt = runtime(logits.device)
    sel = torch.empty(max(n, 1), dtype=torch.int32, device=logits.device)
    wsb = _lib.query('sgnn_compact_ws_bytes', n)
    ws = rt.workspace(wsb)
    _lib.call('sgnn_compact_sigmoid', ptr(logits), stride, n, ptr(sel), ptr(rt.state), ptr(ws), wsb)
    count = rt.read_count()
    return sel[:count], count


def _compact_call(logits, stride, n, coords_all, sel, rt, ws, wsb, teacher):
    if teacher is None:
        _lib.call('sgnn_compact_sigmoid', ptr(logits), stride, n, ptr(sel), ptr(rt.state), ptr(ws), wsb)
    else:       # masks from a dense target occupancy volume (B,1,d0,d1,d2) instead of the predicted logits
        B, _, d0, d1, d2 = (int(v) for v in teacher.shape)
        _lib.call('sgnn_compact_dense', ptr(coords_all), n, ptr(teacher), B, d0, d1, d2, ptr(sel), ptr(rt.state), ptr(ws),
                  wsb)


def compact_sigmoid_plan(logits, stride, n, coords_all, depth, teacher=None):
    """compact_sigmoid + the kept rows' coordinates + the stride-2 pyramid (`depth` levels) below them, with ONE host
    read-back for all row counts: the coordinate gather and the pyramid kernels read the kept-row count from device
    memory (sgnn_gather_rows_dn, sgnn_down2_chain).  Returns (sel[:count], count, locs (count,4) int32); when a
    pyramid was built, `locs` carries it (`_sgnn_plan`) and the next InputLayer adopts it instead of rebuilding."""
    from . import metadata as MD
    rt = runtime(logits.device)
    if not MD.CHAIN or depth < 1 or n == 0:
        if teacher is None:
            sel, cnt = compact_sigmoid(logits, stride, n)
        else:
            sel = torch.empty(max(n, 1), dtype=torch.int32, device=logits.device)
            wsb = _lib.query('sgnn_compact_ws_bytes', n)
            _compact_call(logits, stride, n, coords_all, sel, rt, rt.workspace(wsb), wsb, teacher)
            cnt = rt.read_count()
            sel = sel[:cnt]
        if MD.COUNT_LOG is not None:
            MD.COUNT_LOG.append(('gen', int(cnt), []))
        return sel, cnt, gather_coords(coords_all, sel, cnt)
    sel = torch.empty(n, dtype=torch.int32, device=logits.device)
    wsb = _lib.query('sgnn_compact_ws_bytes', n)
    ws = rt.workspace(wsb)
    _compact_call(logits, stride, n, coords_all, sel, rt, ws, wsb, teacher)
    locs_cap = torch.empty(n, 4, dtype=torch.int32, device=logits.device)
    _lib.call('sgnn_gather_rows_dn', ptr(coords_all), 4, ptr(sel), ptr(rt.state), n, ptr(locs_cap))
    _inherit_bounds(locs_cap, coords_all)
    chain = MD.PendingChain(locs_cap, 0, True, depth)
    host = rt.read_counts()
    count = int(host[0])
    locs = _inherit_bounds(locs_cap[:count], coords_all)
    if MD.COUNT_LOG is not None:
        MD.COUNT_LOG.append(('gen', count, [int(v) for v in host[2:2 + chain.depth]]))
    if count:
        locs._sgnn_plan = chain.finalize(count, host)
    return sel[:count], count, locs


def compact_capped(logits, stride, n_all, coords_all, depth, capacity, g, teacher=None):
    """Capacity-mode counterpart of compact_sigmoid_plan for generative level g of `capacity` (scn.capacity.Capacity):
    no host read-back.  The candidate count is coords_all._sgnn_cnt (None: n_all is exact, the dense coarse volume);
    returns (sel (n_all), kept capacity K, locs (K,4) int32) where locs carries its live count (`_sgnn_cnt`, and
    `_sgnn_cnt8` = 8 x it) and, for depth >= 1, the stride-2 pyramid below it (`_sgnn_plan`) sized by the capacities."""
    from . import metadata as MD
    dev = coords_all.device
    rt = runtime(dev)
    K, pyr_caps = capacity.gen[g]
    cnt2 = capacity.kept2(g)
    n_cnt = getattr(coords_all, '_sgnn_cnt', None)
    sel = torch.empty(max(n_all, 1), dtype=torch.int32, device=dev)
    wsb = _lib.query('sgnn_compact_ws_bytes', n_all)
    ws = rt.workspace(wsb)
    kept, kept8 = cnt2[0:1], cnt2[1:2]
    locs = _inherit_bounds(torch.empty(K, 4, dtype=torch.int32, device=dev), coords_all)
    if FUSED_GLUE:       # the kept sites' coordinates are written by the compaction's own write kernel
        if teacher is None:
            _lib.call('sgnn_compact_sigmoid_cap_locs', ptr(logits), stride, n_all, ptr(n_cnt), ptr(coords_all), ptr(sel),
                      ptr(locs), ptr(cnt2), K, ptr(rt.status32), ptr(ws), wsb)
        else:
            B, _, d0, d1, d2 = (int(v) for v in teacher.shape)
            _lib.call('sgnn_compact_dense_cap_locs', ptr(coords_all), n_all, ptr(n_cnt), ptr(teacher), B, d0, d1, d2,
                      ptr(sel), ptr(locs), ptr(cnt2), K, ptr(rt.status32), ptr(ws), wsb)
    else:
        if teacher is None:
            _lib.call('sgnn_compact_sigmoid_cap', ptr(logits), stride, n_all, ptr(n_cnt), ptr(sel), ptr(cnt2), K,
                      ptr(rt.status32), ptr(ws), wsb)
        else:
            B, _, d0, d1, d2 = (int(v) for v in teacher.shape)
            _lib.call('sgnn_compact_dense_cap', ptr(coords_all), n_all, ptr(n_cnt), ptr(teacher), B, d0, d1, d2, ptr(sel),
                      ptr(cnt2), K, ptr(rt.status32), ptr(ws), wsb)
        _lib.call('sgnn_gather_rows_dn', ptr(coords_all), 4, ptr(sel), ptr(kept), K, ptr(locs))
    locs._sgnn_cnt, locs._sgnn_cnt8 = kept, kept8
    if depth >= 1:
        depth = min(depth, len(pyr_caps))
        chain = MD.PendingChain(locs, K, True, depth, n0_cnt=kept, counts=capacity.pyr_counts(g, depth),
                                level_caps=pyr_caps)
        locs._sgnn_plan = chain.finalize_capped()
    return sel, K, locs


def compact_mask(mask_u8, n):
    rt = runtime(mask_u8.device)
    sel = torch.empty(max(n, 1), dtype=torch.int32, device=mask_u8.device)
    wsb = _lib.query('sgnn_compact_ws_bytes', n)
    ws = rt.workspace(wsb)
    _lib.call('sgnn_compact_mask', ptr(mask_u8), n, ptr(sel), ptr(rt.state), ptr(ws), wsb)
    count = rt.read_count()
    return sel[:count], count


def _inherit_bounds(coords, parent, scale=1):
    """A subset (scale = 1) or the 8-child expansion (scale = 2) of sites that lie in [0, Z) x [0, Y) x [0, X), b < B by
    construction does so too: the bound travels with the coordinates (Grid.bounds: such a level needs no hash grid)."""
    b = getattr(parent, '_sgnn_bounds', None)
    if b is not None:
        coords._sgnn_bounds = (int(b[0]), int(b[1]) * scale, int(b[2]) * scale, int(b[3]) * scale)
    return coords


def gather_coords(coords32, sel, m):
    """coords rows are 16-byte rows: reuse the fp32 row gather as a pure bit copy."""
    out = torch.empty(m, 4, dtype=torch.int32, device=coords32.device)
    _lib.call('sgnn_gather_rows', ptr(coords32), 4, ptr(sel), m, ptr(out))
    return _inherit_bounds(out, coords32)


def expand8_coords(coords32, with_i64=False):
    """with_i64: the children's int64 rows (what coords_to_i64 would return for them) are written in the same pass and
    travel with the result (`_sgnn_i64`) — the model returns them as the level's `locs`."""
    n = coords32.shape[0]
    out = torch.empty(8 * n, 4, dtype=torch.int32, device=coords32.device)
    cnt = getattr(coords32, '_sgnn_cnt', None)
    if with_i64 and FUSED_GLUE:
        o64 = torch.empty(8 * n, 4, dtype=torch.int64, device=coords32.device)
        _lib.call('sgnn_expand8_coords_i64', ptr(coords32), n, ptr(out), ptr(o64), ptr(cnt))
        out._sgnn_i64 = o64
    else:
        _lib.call('sgnn_expand8_coords', ptr(coords32), n, ptr(out), ptr(cnt))
    if cnt is not None:          # capacity mode: the children's live row count (8 x kept) sits next to the kept count
        out._sgnn_cnt = coords32._sgnn_cnt8
        if getattr(out, '_sgnn_i64', None) is not None:
            out._sgnn_i64._sgnn_cnt = out._sgnn_cnt
    return _inherit_bounds(out, coords32, 2)


def dense_coords(batch, d0, d1, d2, device):
    out = torch.empty(batch * d0 * d1 * d2, 4, dtype=torch.int32, device=device)
    _lib.call('sgnn_dense_coords', batch, d0, d1, d2, ptr(out))
    return out


def coords_to_i64(coords32):
    made = getattr(coords32, '_sgnn_i64', None)       # expand8_coords(with_i64=True) already wrote them
    if made is not None:
        return made
    n = coords32.shape[0]
    out = torch.empty(n, 4, dtype=torch.int64, device=coords32.device)
    cnt = getattr(coords32, '_sgnn_cnt', None)
    _lib.call('sgnn_coords_to_i64', ptr(coords32), n, ptr(out), ptr(cnt))
    if cnt is not None:
        out._sgnn_cnt = cnt
    return out
