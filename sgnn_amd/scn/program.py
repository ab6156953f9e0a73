"""Compile a tree of scn containers into a flat op list and run it through the native executor
(sgnn_prog_forward / sgnn_prog_backward, sgnn_amd/csrc/prog.hip).

The reference composes its sparse sub-networks from scn.Sequential / ConcatTable / AddTable / JoinTable
(torch/model.py:31-47 encoder layer, :178-188 Refinement, :253-257 SurfacePrediction, and upstream's
FullyConvolutionalNet).  Executing those containers layer by layer costs one Python autograd node per layer
(~110 per step); a Program runs the same kernels in the same order from one call per direction.  The
modules stay the parameter holders, so state-dict layout and results are unchanged (bit-identical to the
per-layer path, tests/test_gpu_program.py).
"""
import ctypes

import numpy as np
import torch
from torch.autograd import Function

from .. import _lib
from . import modules as M
from .metadata import runtime

OP_SUBM, OP_DOWN, OP_UNPOOL, OP_BN, OP_ADD, OP_JOIN = range(6)
ENABLED = True   # False: run the containers layer by layer (one autograd node per layer)


class Unsupported(Exception):
    pass


class Program(object):
    """ops over buffers; buffer 0 = input (level 0).  `taps` maps a module to the buffer holding its output."""

    def __init__(self, chain, in_channels, tap_modules=()):
        self.ops, self.opf, self.bufs = [], [], [[0, in_channels]]
        self.slots = []            # tensors in parameter-slot order (conv weight | gamma, beta, rmean, rvar)
        self.grad_slot = []        # True where the slot is a trainable parameter
        self.taps = {}
        self._tap_modules = set(id(m) for m in tap_modules)
        cur = (0, 0, in_channels)
        for m in chain:
            cur = self._emit(m, cur)
        if isinstance(cur, list):
            raise Unsupported('chain ends in a ConcatTable')
        self.out = cur[0]
        self.nlev = 1 + max(b[0] for b in self.bufs)
        self.ops_np = np.ascontiguousarray(np.array(self.ops, dtype=np.int32).reshape(-1, 8))
        self.opf_np = np.ascontiguousarray(np.array(self.opf, dtype=np.float32).reshape(-1, 4))
        self.bufs_np = np.ascontiguousarray(np.array(self.bufs, dtype=np.int32).reshape(-1, 2))
        self.subm_levels = sorted(set(o[5] for o in self.ops if o[0] == OP_SUBM))

    def _new_buf(self, level, ch):
        self.bufs.append([level, ch])
        return len(self.bufs) - 1

    def _slot(self, module, names, trainable):
        first = len(self.slots)
        for nm, g in zip(names, trainable):
            self.slots.append((module, nm))     # resolved at run time: survives .to()/load_state_dict
            self.grad_slot.append(g)
        return first

    def tensors(self):
        return [getattr(m, nm) for m, nm in self.slots]

    def _emit(self, m, cur):
        out = self._emit_inner(m, cur)
        if id(m) in self._tap_modules and not isinstance(out, list):
            self.taps[id(m)] = out
        return out

    def _emit_inner(self, m, cur):
        if isinstance(m, (M.Sequential, torch.nn.Sequential)):
            for c in m._modules.values():
                cur = self._emit(c, cur)
            return cur
        if isinstance(m, M.ConcatTable):
            if isinstance(cur, list):
                raise Unsupported('nested ConcatTable input')
            return [self._emit(c, cur) for c in m._modules.values()]
        if isinstance(m, (M.AddTable, M.JoinTable)):
            if not isinstance(cur, list) or len(cur) < 2:
                raise Unsupported('table op without a ConcatTable in front')
            acc = cur[0]
            for nxt in cur[1:]:
                if acc[1] != nxt[1]:
                    raise Unsupported('table inputs on different levels')
                if isinstance(m, M.AddTable):
                    if acc[2] != nxt[2]:
                        raise Unsupported('AddTable channel mismatch')
                    b = self._new_buf(acc[1], acc[2])
                    self.ops.append([OP_ADD, acc[0], nxt[0], b, -1, acc[1], acc[2], acc[2]])
                    acc = (b, acc[1], acc[2])
                else:
                    b = self._new_buf(acc[1], acc[2] + nxt[2])
                    self.ops.append([OP_JOIN, acc[0], nxt[0], b, -1, acc[1], acc[2], nxt[2]])
                    acc = (b, acc[1], acc[2] + nxt[2])
                self.opf.append([0, 0, 0, 0])
            return acc
        if isinstance(cur, list):
            raise Unsupported('%s applied to a table' % type(m).__name__)
        buf, lev, ch = cur
        if isinstance(m, M.Identity):
            return cur
        if isinstance(m, M.SubmanifoldConvolution):
            if m.bias is not None or m.nIn != ch:
                raise Unsupported('SubmanifoldConvolution with bias / channel mismatch')
            b = self._new_buf(lev, m.nOut)
            self.ops.append([OP_SUBM, buf, -1, b, self._slot(m, ['weight'], [True]), lev, m.nIn, m.nOut])
            self.opf.append([0, 0, 0, 0])
            return (b, lev, m.nOut)
        if isinstance(m, M.Convolution):
            if m.bias is not None or m.nIn != ch:
                raise Unsupported('Convolution with bias / channel mismatch')
            b = self._new_buf(lev + 1, m.nOut)
            self.ops.append([OP_DOWN, buf, -1, b, self._slot(m, ['weight'], [True]), lev, m.nIn, m.nOut])
            self.opf.append([0, 0, 0, 0])
            return (b, lev + 1, m.nOut)
        if isinstance(m, M.UnPooling):
            if lev < 1:
                raise Unsupported('UnPooling above the input level')
            b = self._new_buf(lev - 1, ch)
            self.ops.append([OP_UNPOOL, buf, -1, b, -1, lev - 1, ch, ch])
            self.opf.append([0, 0, 0, 0])
            return (b, lev - 1, ch)
        if isinstance(m, M.BatchNormalization):
            if m.weight is None or m.nPlanes != ch:
                raise Unsupported('non-affine BatchNormalization / channel mismatch')
            b = self._new_buf(lev, ch)
            s = self._slot(m, ['weight', 'bias', 'running_mean', 'running_var'], [True, True, False, False])
            self.ops.append([OP_BN, buf, -1, b, s, lev, ch, ch])
            self.opf.append([m.eps, m.momentum, m.leakiness, 0])
            return (b, lev, ch)
        raise Unsupported('module %s' % type(m).__name__)


def _ptr_array(values):
    return np.ascontiguousarray(np.array(values, dtype=np.uint64))


class _Run(object):
    """Per-forward state: level geometry, arena, host descriptor arrays."""
    pass


def _levels(prog, x):
    """Grids / stride-2 rulebooks of the program's levels, built through the tensor's Metadata (host syncs for
    the coarse row counts, exactly as the per-layer path does)."""
    md, key = x.metadata, x.key
    grids, downs = [x.grid()], []
    for _ in range(prog.nlev - 1):
        if any(v % 2 for v in key):
            raise ValueError('Convolution(2,2): spatial size %s is not even' % list(key))
        nxt = tuple(v // 2 for v in key)
        d = md.down2(key, nxt)
        downs.append(d)
        grids.append(d.coarse)
        key = nxt
    return grids, downs


class _ProgramFn(Function):
    @staticmethod
    def forward(ctx, x, run, *params):
        prog = run.prog
        x = x.contiguous()
        rt = runtime(x.device)
        n_lev = prog.nlev
        lev_n = np.array([g.n for g in run.grids], dtype=np.int64)
        lev_ld = np.array([g.ld for g in run.grids], dtype=np.int64)
        nbr = [0] * n_lev
        for l in prog.subm_levels:
            nbr[l] = run.grids[l].subm_table().data_ptr()
        children = [d.children.data_ptr() for d in run.downs] + [0]
        ptable = [d.ptable.data_ptr() for d in run.downs] + [0]
        parent = [d.parent.data_ptr() if d.parent.numel() else 0 for d in run.downs] + [0]
        run.lev_n, run.lev_ld = lev_n, lev_ld
        run.tabs = [_ptr_array(v) for v in (nbr, children, ptable, parent)]
        run.pptr = _ptr_array([p.data_ptr() for p in params])
        ops, opf, bufs = prog.ops_np, prog.opf_np, prog.bufs_np
        nops, nbuf = ops.shape[0], bufs.shape[0]
        total = _lib.query('sgnn_prog_arena_floats', ops.ctypes.data, nops, bufs.ctypes.data, nbuf,
                           lev_n.ctypes.data, n_lev)
        wsb = _lib.query('sgnn_prog_ws_bytes', ops.ctypes.data, nops, lev_n.ctypes.data, n_lev)
        run.total, run.wsb = total, wsb
        arena = torch.empty(total, dtype=torch.float32, device=x.device)
        ws = rt.workspace(wsb)
        keep = np.zeros(nbuf, dtype=np.int32)          # buffers read after the call: never fused away
        keep[run.out_bufs] = 1
        _lib.call('sgnn_prog_forward', ops.ctypes.data, opf.ctypes.data, nops, bufs.ctypes.data, nbuf,
                  lev_n.ctypes.data, lev_ld.ctypes.data, run.tabs[0].ctypes.data, run.tabs[1].ctypes.data,
                  run.tabs[2].ctypes.data, run.tabs[3].ctypes.data, n_lev, run.pptr.ctypes.data, len(params),
                  x.data_ptr(), arena.data_ptr(), total, keep.ctypes.data, int(run.training), ws.data_ptr(), wsb)
        run.offsets = {}
        outs = []
        for b in run.out_bufs:
            off = _lib.query('sgnn_prog_buffer_offset', ops.ctypes.data, nops, bufs.ctypes.data, nbuf,
                             lev_n.ctypes.data, n_lev, b)
            rows, ch = int(lev_n[bufs[b, 0]]), int(bufs[b, 1])
            run.offsets[b] = (off, rows, ch)
            outs.append(arena[off:off + rows * ch].view(rows, ch))
        ctx.run = run
        ctx.save_for_backward(x, arena, *params)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        run = ctx.run
        prog = run.prog
        x, arena = ctx.saved_tensors[0], ctx.saved_tensors[1]
        params = ctx.saved_tensors[2:]
        rt = runtime(x.device)
        ops, opf, bufs = prog.ops_np, prog.opf_np, prog.bufs_np
        nops, nbuf = ops.shape[0], bufs.shape[0]
        garena = torch.empty(run.total, dtype=torch.float32, device=x.device)
        ginit = np.zeros(nbuf, dtype=np.int32)
        for b, g in zip(run.out_bufs, gouts):
            if g is None:
                continue
            off, rows, ch = run.offsets[b]
            garena[off:off + rows * ch].view(rows, ch).copy_(g)
            ginit[b] = 1
        # one flat gradient tensor for all trainable slots
        sizes = [p.numel() if t else 0 for p, t in zip(params, prog.grad_slot)]
        flat = torch.empty(max(sum(sizes), 1), dtype=torch.float32, device=x.device)
        gptr, views, o = [], [], 0
        for p, s in zip(params, sizes):
            if s:
                v = flat[o:o + s].view_as(p)
                gptr.append(v.data_ptr())
                views.append(v)
                o += s
            else:
                gptr.append(0)
                views.append(None)
        gp = _ptr_array(gptr)
        need_dx = bool(ctx.needs_input_grad[0])
        ws = rt.workspace(run.wsb)
        rt.side_lane(run.wsb)
        _lib.call('sgnn_prog_backward', ops.ctypes.data, opf.ctypes.data, nops, bufs.ctypes.data, nbuf,
                  run.lev_n.ctypes.data, run.lev_ld.ctypes.data, run.tabs[0].ctypes.data, run.tabs[1].ctypes.data,
                  run.tabs[2].ctypes.data, run.tabs[3].ctypes.data, prog.nlev, run.pptr.ctypes.data, gp.ctypes.data,
                  len(params), x.data_ptr(), arena.data_ptr(), garena.data_ptr(), run.total, ginit.ctypes.data,
                  int(need_dx), int(run.training), ws.data_ptr(), run.wsb)
        dx = None
        if need_dx:
            off0 = _lib.query('sgnn_prog_buffer_offset', ops.ctypes.data, nops, bufs.ctypes.data, nbuf,
                              run.lev_n.ctypes.data, prog.nlev, 0)
            dx = garena[off0:off0 + x.numel()].view_as(x)
        return (dx, None) + tuple(views)


def compile_or_none(chain, in_channels, tap_modules=()):
    try:
        return Program(chain, in_channels, tap_modules)
    except Unsupported:
        return None


def run_program(prog, x, training, out_bufs=None):
    """x: SparseConvNetTensor at the program's level 0.  Returns (list of output feature tensors, grids, downs)."""
    run = _Run()
    run.prog, run.training = prog, training
    run.grids, run.downs = _levels(prog, x)
    run.out_bufs = list(out_bufs) if out_bufs is not None else [prog.out]
    outs = _ProgramFn.apply(x.features, run, *prog.tensors())
    return list(outs), run.grids, run.downs
