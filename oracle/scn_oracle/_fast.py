"""TEST INFRASTRUCTURE — ctypes binding of oracle/_build/libscn_cpu.so (C/OpenMP versions of the oracle's two hot
loops; built by `make -C oracle cpu`, which __graft_entry__.build() runs).  Optional: `available` is False when the
library has not been built, and nothing changes unless scn_oracle.FAST is set (bench.py's cpu_baseline leg does)."""
import ctypes
import os

import numpy as np
import torch

_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), '_build', 'libscn_cpu.so')
_lib = None
if os.path.exists(_PATH):
    try:
        _lib = ctypes.CDLL(_PATH)
        vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
        _lib.scn_conv_fwd.argtypes = [vp, i32, vp, i32, vp, vp, i64, vp]
        _lib.scn_conv_bwd_x.argtypes = [vp, i32, vp, i32, vp, vp, i64, vp]
        _lib.scn_conv_bwd_w.argtypes = [vp, i32, vp, i32, vp, vp, i64, vp]
        _lib.scn_subm_rules.argtypes = [vp, i64, vp, vp, vp]
        _lib.scn_conv_tab.argtypes = [vp, i32, vp, i32, i32, vp, i64, vp]
        _lib.scn_conv_tab_t.argtypes = [vp, i32, vp, i32, i32, vp, i64, i32, vp]
        _lib.scn_conv_tab_w.argtypes = [vp, i32, vp, i32, i32, vp, i64, vp]
        for f in (_lib.scn_conv_tab, _lib.scn_conv_tab_t, _lib.scn_conv_tab_w):
            f.restype = None
        _lib.scn_cpu_threads.restype = i32
        for f in (_lib.scn_conv_fwd, _lib.scn_conv_bwd_x, _lib.scn_conv_bwd_w, _lib.scn_subm_rules):
            f.restype = None
    except OSError:
        _lib = None
available = _lib is not None


def threads():
    return int(_lib.scn_cpu_threads()) if available else 0


class RuleConv(torch.autograd.Function):
    """rule_conv of scn_oracle (per-offset x[in] @ W[k] added into y[out]) through the C kernels, float32 only."""

    @staticmethod
    def forward(ctx, x, weight, pairs, n_out):
        x, w = x.contiguous(), weight.contiguous()
        cin, cout = int(w.shape[1]), int(w.shape[2])
        y = x.new_zeros(n_out, cout)
        pairs = [(i.contiguous(), o.contiguous()) for i, o in pairs]
        for k, (i, o) in enumerate(pairs):
            if i.numel():
                _lib.scn_conv_fwd(x.data_ptr(), cin, w[k].data_ptr(), cout, i.data_ptr(), o.data_ptr(), i.numel(),
                                  y.data_ptr())
        ctx.save_for_backward(x, w)
        ctx.pairs = pairs
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        cin, cout = int(w.shape[1]), int(w.shape[2])
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.zeros_like(w) if ctx.needs_input_grad[1] else None
        for k, (i, o) in enumerate(ctx.pairs):
            if not i.numel():
                continue
            if dx is not None:
                _lib.scn_conv_bwd_x(dy.data_ptr(), cout, w[k].data_ptr(), cin, i.data_ptr(), o.data_ptr(), i.numel(),
                                    dx.data_ptr())
            if dw is not None:
                _lib.scn_conv_bwd_w(x.data_ptr(), cin, dy.data_ptr(), cout, i.data_ptr(), o.data_ptr(), i.numel(),
                                    dw[k].data_ptr())
        return dx, dw, None, None


class TableConv(torch.autograd.Function):
    """The same convolution from its neighbour tables (one OpenMP region per call): tab_f (K, n_out) int64 numpy =
    input row per (offset, output row); tab_b (K, n_in) = output row per (offset, input row); flip_b: the data gradient
    walks the weights in reverse offset order (3x3x3: tab_b is the forward table itself, mirrored)."""

    @staticmethod
    def forward(ctx, x, weight, tab_f, tab_b, flip_b, n_out):
        x, w = x.contiguous(), weight.contiguous()
        K, cin, cout = (int(v) for v in w.shape)
        y = x.new_zeros(n_out, cout)
        if n_out and x.shape[0]:
            _lib.scn_conv_tab(x.data_ptr(), cin, w.data_ptr(), cout, K, tab_f.ctypes.data, n_out, y.data_ptr())
        ctx.save_for_backward(x, w)
        ctx.tabs = (tab_f, tab_b, flip_b, n_out)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        tab_f, tab_b, flip_b, n_out = ctx.tabs
        dy = dy.contiguous()
        K, cin, cout = (int(v) for v in w.shape)
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.zeros_like(w) if ctx.needs_input_grad[1] else None
        if n_out and x.shape[0]:
            if dx is not None:
                _lib.scn_conv_tab_t(dy.data_ptr(), cout, w.data_ptr(), cin, K, tab_b.ctypes.data, x.shape[0], int(flip_b),
                                    dx.data_ptr())
            if dw is not None:
                _lib.scn_conv_tab_w(x.data_ptr(), cin, dy.data_ptr(), cout, K, tab_f.ctypes.data, n_out, dw.data_ptr())
        return dx, dw, None, None, None, None


def subm_rules(coords, sorted_keys, order):
    n = int(coords.shape[0])
    nbr = np.empty((27, n), dtype=np.int64)
    if n:
        c, s, o = (np.ascontiguousarray(a, dtype=np.int64) for a in (coords, sorted_keys, order))
        _lib.scn_subm_rules(c.ctypes.data, n, s.ctypes.data, o.ctypes.data, nbr.ctypes.data)
    return nbr
