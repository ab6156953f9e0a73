"""Compile a tree of scn containers — optionally with the tensor glue around it — into a flat op list and run it
through the native executor (sgnn_prog_forward / sgnn_prog_backward, sgnn_amd/csrc/prog.hip).

The reference composes its sparse sub-networks from scn.Sequential / ConcatTable / AddTable / JoinTable
(torch/model.py:31-47 encoder layer, :178-191 Refinement, :253-258 SurfacePrediction, and upstream's
FullyConvolutionalNet) and glues the generative stages together with tensor ops (:209-247 Refinement.forward,
:259-272 SurfacePrediction.forward, :338-355 concat_skip).  Executing that layer by layer costs one Python autograd
node per layer (~110 per step) plus a dozen more per stage for the glue, and the training step is host-bound
(profiles/r02_host_bound.txt); a Program runs the same kernels from ONE call per direction:

    [CONCAT_IN: kept rows of the previous stage | their occ/sdf logits | encoder skip features at those sites]
      -> chain of scn modules (p1, p2, p3) -> [8-child up-sampling convolution n1 -> BatchNormReLU n2 -> linear heads]

The modules stay the parameter holders, so state-dict layout and results are unchanged (tests/test_gpu_program.py).
"""
import numpy as np
import torch
from torch.autograd import Function

from .. import _lib
from . import modules as M
from .metadata import runtime

OP_SUBM, OP_DOWN, OP_UNPOOL, OP_BN, OP_ADD, OP_JOIN, OP_CONCAT_IN, OP_EXPAND, OP_LINEAR = range(9)
OPW = 12
ENABLED = True   # False: run the containers layer by layer (one autograd node per layer)
# True: every Program keeps ONE grow-only arena (x1.25 head-room) and all programs share one gradient arena, instead of
# a fresh torch allocation per call.  Generated level sizes change every step (masks are data dependent): with fresh
# allocations the caching allocator meets a size it has no block for every few steps and falls back to hipMalloc /
# hipFree — 10x step-time spikes on the 128^3 @ 20 % stress configuration (profiles/r01i_config5.txt).  Opt-in,
# because the outputs of a forward call are views of the arena: they are overwritten by the NEXT forward call of the
# same program (fine for a training loop, wrong for code that keeps two forward results alive).
PERSISTENT_ARENAS = False
_garenas = {}
# Deferred lane join (train.GraphStep sets DEFER_JOIN around its backward pass): a program's backward call does not wait
# for its weight-gradient kernels on the side lane; the caller joins the lane once, before the optimizer, and then clears
# `_deferred`.  Until then everything those kernels read — the program's forward arena, its gradient arena (its own one:
# not the shared per-device arena), the incoming output gradients, the level tables — is kept alive here.
DEFER_JOIN = False
_deferred = []
_iarenas = {}      # inference: one forward arena per device, shared by all programs


ARENA_SHRINK_AFTER = 32     # calls in a row that used less than half of a persistent arena before it is re-allocated


def _arena(owner, key, floats, dev, headroom=1.25):
    """Grow-only (x headroom) persistent arena that also SHRINKS: after ARENA_SHRINK_AFTER consecutive calls that needed
    less than half of it (a curriculum switched a level off, a large scene was followed by small ones) it is
    re-allocated at the size now needed — level sizes that merely fluctuate never trigger it."""
    t = owner.get(key)
    idle = owner.get((key, 'idle'), 0)
    idle = idle + 1 if (t is not None and floats * 2 < t.numel()) else 0
    if t is None or t.numel() < floats or t.device != dev or idle >= ARENA_SHRINK_AFTER:
        owner[key] = None
        del t
        t = owner[key] = torch.empty(int(floats * headroom) + 1024, dtype=torch.float32, device=dev)
        idle = 0
    owner[(key, 'idle')] = idle
    return t[:floats]


def arena_bytes():
    """Bytes currently held by persistent arenas: (forward arenas of all programs are reported by their owners, see
    memory_report) gradient arenas, inference arenas."""
    f = lambda d: sum(4 * t.numel() for k, t in d.items() if torch.is_tensor(t))
    return {'gradient': f(_garenas), 'inference': f(_iarenas)}


def release_arenas():
    """Drop the shared persistent arenas (gradient arenas, inference arenas); they are re-allocated on demand.  Forward
    arenas belong to their programs and go with the model."""
    _garenas.clear()
    _iarenas.clear()


class Unsupported(Exception):
    pass


def _op(t, in0=-1, in1=-1, out=-1, par=-1, lev=0, cin=0, cout=0, in2=-1, ia=-1, ib=-1, ic=-1):
    return [t, in0, in1, out, par, lev, cin, cout, in2, ia, ib, ic]


class Program(object):
    """ops over buffers.  Without `sources`, buffer 0 is the one external input (level 0, in_channels).  With
    `sources` = up to three (rows class name, channels, index slot or None) entries (None entries allowed), the
    externals are those tensors and a CONCAT_IN op builds the chain's input rows from them.  `tail` continues after
    the chain: ('expand', SubmanifoldConvolution) | ('bn', BatchNormalization) | ('linear', [nn.Linear, ...]).
    Rows classes: 0..nlev-1 = the stride-2 pyramid below the chain's input level; further named classes ('child' =
    8 x level 0, and the classes of the sources) get the ids nlev.. in `class_ids`; their row counts are given per run.
    `taps` maps a module (or a tail entry's module / first linear) to the buffer holding its output."""

    def __init__(self, chain, in_channels, tap_modules=(), sources=None, tail=()):
        self.ops, self.opf, self.bufs = [], [], []
        self.slots = []            # (module, attribute) in parameter-slot order
        self.grad_slot = []        # True where the slot is a trainable parameter
        self.taps = {}
        self._tap_modules = set(id(m) for m in tap_modules)
        self._classes = []         # named rows classes, in id order after the pyramid levels
        if sources is None:
            self.n_ext = 1
            self.bufs.append([0, in_channels])
            cur = (0, 0, in_channels)
        else:
            if len(sources) != 3 or all(s is None for s in sources):
                raise Unsupported('CONCAT_IN takes three (possibly None) sources')
            ins, idxs, total = [], [], 0
            for s in sources:
                if s is None:
                    ins.append(-1)
                    idxs.append(-1)
                    continue
                cls, ch, slot = s
                ins.append(self._new_buf(self._class(cls), ch))
                idxs.append(-1 if slot is None else int(slot))
                total += ch
            self.n_ext = len(self.bufs)
            if total != in_channels:
                raise Unsupported('sources carry %d channels, the chain expects %d' % (total, in_channels))
            b = self._new_buf(0, total)
            self.ops.append(_op(OP_CONCAT_IN, ins[0], ins[1], b, -1, 0, 0, 0, ins[2], idxs[0], idxs[1], idxs[2]))
            self.opf.append([0, 0, 0, 0])
            cur = (b, 0, total)
        for m in chain:
            cur = self._emit(m, cur)
        if isinstance(cur, list):
            raise Unsupported('chain ends in a ConcatTable')
        self.out = cur[0]
        for kind, m in tail:
            cur = self._emit_tail(kind, m, cur)
        self.tail_out = cur[0]
        self.nlev = 1 + max([b[0] for b in self.bufs if b[0] >= 0] + [0])
        for b in self.bufs:                      # named classes were negative placeholders until nlev was known
            if b[0] < 0:
                b[0] = self.nlev + (-1 - b[0])
        for o in self.ops:
            if o[5] < 0:
                o[5] = self.nlev + (-1 - o[5])
        self.class_ids = dict((name, self.nlev + k) for k, name in enumerate(self._classes))
        self.n_classes = self.nlev + len(self._classes)
        self.ops_np = np.ascontiguousarray(np.array(self.ops, dtype=np.int32).reshape(-1, OPW))
        self.opf_np = np.ascontiguousarray(np.array(self.opf, dtype=np.float32).reshape(-1, 4))
        self.bufs_np = np.ascontiguousarray(np.array(self.bufs, dtype=np.int32).reshape(-1, 2))
        self.subm_levels = sorted(set(o[5] for o in self.ops if o[0] in (OP_SUBM, OP_EXPAND)))
        self.n_idx = 1 + max([max(o[9:12]) for o in self.ops] + [-1])

    def _class(self, name):
        if name not in self._classes:
            self._classes.append(name)
        return -1 - self._classes.index(name)

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

    def _emit_tail(self, kind, m, cur):
        buf, lev, ch = cur
        if kind == 'expand':
            if lev != 0 or not isinstance(m, M.SubmanifoldConvolution) or m.bias is not None or m.nIn != ch:
                raise Unsupported('up-sampling convolution: level-0 input, no bias')
            cls = self._class('child')
            b = self._new_buf(cls, m.nOut)
            self.ops.append(_op(OP_EXPAND, buf, -1, b, self._slot(m, ['weight'], [True]), 0, m.nIn, m.nOut))
            self.opf.append([0, 0, 0, 0])
            out = (b, cls, m.nOut)
        elif kind == 'bn':
            if not isinstance(m, M.BatchNormalization) or m.weight is None or m.nPlanes != ch:
                raise Unsupported('non-affine BatchNormalization / channel mismatch')
            b = self._new_buf(lev, ch)
            s = self._slot(m, ['weight', 'bias', 'running_mean', 'running_var'], [True, True, False, False])
            self.ops.append(_op(OP_BN, buf, -1, b, s, lev, ch, ch))
            self.opf.append([m.eps, m.momentum, m.leakiness, 0])
            out = (b, lev, ch)
        elif kind == 'linear':
            lins = list(m)
            if not 1 <= len(lins) <= 2 or any(l.in_features != ch or l.out_features != 1 for l in lins):
                raise Unsupported('linear heads: one or two nn.Linear(ch, 1)')
            first = len(self.slots)
            for l in lins:
                self._slot(l, ['weight', 'bias'], [True, l.bias is not None])
            b = self._new_buf(lev, len(lins))
            self.ops.append(_op(OP_LINEAR, buf, -1, b, first, lev, ch, len(lins)))
            self.opf.append([0, 0, 0, 0])
            out = (b, lev, len(lins))
            m = lins[0]
        else:
            raise Unsupported('tail entry %r' % (kind,))
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
                    self.ops.append(_op(OP_ADD, acc[0], nxt[0], b, -1, acc[1], acc[2], acc[2]))
                    acc = (b, acc[1], acc[2])
                else:
                    b = self._new_buf(acc[1], acc[2] + nxt[2])
                    self.ops.append(_op(OP_JOIN, acc[0], nxt[0], b, -1, acc[1], acc[2], nxt[2]))
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
            self.ops.append(_op(OP_SUBM, buf, -1, b, self._slot(m, ['weight'], [True]), lev, m.nIn, m.nOut))
            self.opf.append([0, 0, 0, 0])
            return (b, lev, m.nOut)
        if isinstance(m, M.Convolution):
            if m.bias is not None or m.nIn != ch:
                raise Unsupported('Convolution with bias / channel mismatch')
            b = self._new_buf(lev + 1, m.nOut)
            self.ops.append(_op(OP_DOWN, buf, -1, b, self._slot(m, ['weight'], [True]), lev, m.nIn, m.nOut))
            self.opf.append([0, 0, 0, 0])
            return (b, lev + 1, m.nOut)
        if isinstance(m, M.UnPooling):
            if lev < 1:
                raise Unsupported('UnPooling above the input level')
            b = self._new_buf(lev - 1, ch)
            self.ops.append(_op(OP_UNPOOL, buf, -1, b, -1, lev - 1, ch, ch))
            self.opf.append([0, 0, 0, 0])
            return (b, lev - 1, ch)
        if isinstance(m, M.BatchNormalization):
            if m.weight is None or m.nPlanes != ch:
                raise Unsupported('non-affine BatchNormalization / channel mismatch')
            b = self._new_buf(lev, ch)
            s = self._slot(m, ['weight', 'bias', 'running_mean', 'running_var'], [True, True, False, False])
            self.ops.append(_op(OP_BN, buf, -1, b, s, lev, ch, ch))
            self.opf.append([m.eps, m.momentum, m.leakiness, 0])
            return (b, lev, ch)
        raise Unsupported('module %s' % type(m).__name__)


def _ptr_array(values):
    return np.ascontiguousarray(np.array(values, dtype=np.uint64))


class _Run(object):
    """Per-forward state: level geometry, arena, host descriptor arrays."""
    pass


def _levels(prog, md, key, grid0):
    """Grids / stride-2 rulebooks of the program's pyramid, through the Metadata (pre-built by Metadata.prebuild or a
    compaction's PendingChain: no host sync here then)."""
    grids, downs = [grid0], []
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
    def forward(ctx, run, *tensors):
        prog = run.prog
        n_ext = prog.n_ext
        ext = [t.contiguous() for t in tensors[:n_ext]]
        params = tensors[n_ext:]
        dev = ext[0].device
        rt = runtime(dev)
        lev_n = np.zeros(prog.n_classes, dtype=np.int64)
        lev_ld = np.zeros(prog.n_classes, dtype=np.int64)
        for l, g in enumerate(run.grids):
            lev_n[l], lev_ld[l] = g.n, g.ld
        for name, cid in prog.class_ids.items():
            lev_n[cid] = run.extra_rows[name]
        nbr = [0] * prog.n_classes
        # capacity mode: device row count per rows class (0 = lev_n is exact); lev_n then holds the capacities
        cnts = [0] * prog.n_classes
        for l, g in enumerate(run.grids):
            if g.cnt is not None:
                cnts[l] = g.cnt.data_ptr()
        for name, cid in prog.class_ids.items():
            c = run.extra_cnt.get(name)
            if c is not None:
                cnts[cid] = c.data_ptr()
        for l in prog.subm_levels:
            nbr[l] = run.grids[l].subm_table().data_ptr()
        pad = [0] * (prog.n_classes - len(run.downs))
        children = [d.children.data_ptr() for d in run.downs] + pad
        ptable = [d.ptable.data_ptr() for d in run.downs] + pad
        parent = [d.parent.data_ptr() if d.parent.numel() else 0 for d in run.downs] + pad
        run.lev_n, run.lev_ld = lev_n, lev_ld
        prog.last_lev_n = lev_n.copy()          # rows (capacity mode: capacities) per class of the latest run (bench accounting)
        run.tabs = [_ptr_array(v) for v in (nbr, children, ptable, parent, cnts)]
        run.pptr = _ptr_array([0 if p is None else p.data_ptr() for p in params])
        run.eptr = _ptr_array([t.data_ptr() for t in ext])
        run.iptr = _ptr_array([t.data_ptr() for t in run.idx] + [0])
        ops, opf, bufs = prog.ops_np, prog.opf_np, prog.bufs_np
        nops, nbuf, ncls = ops.shape[0], bufs.shape[0], prog.n_classes
        keep = np.zeros(nbuf, dtype=np.int32)          # buffers read after the call: never fused away / never views
        keep[run.out_bufs] = 1
        run.keep = keep
        qa = (ops.ctypes.data, nops, bufs.ctypes.data, nbuf, n_ext, lev_n.ctypes.data, ncls, keep.ctypes.data)
        infer = bool(run.infer)
        # gradient arena (backward: buffers + scratch) / forward arena (buffers only; liveness-packed for inference)
        total = _lib.query('sgnn_prog_arena_floats', *qa, 0)
        fwd_total = _lib.query('sgnn_prog_arena_floats', *qa, 2 if infer else 1)
        wsb = _lib.query('sgnn_prog_ws_bytes', ops.ctypes.data, nops, lev_n.ctypes.data, ncls)
        run.total, run.wsb = total, wsb
        prog.last_arena_floats = (fwd_total, total)
        if infer:           # one arena for ALL programs of the device: the outputs are copied out below
            arena = _arena(_iarenas, str(dev), fwd_total, dev, 1.1)
        elif PERSISTENT_ARENAS:
            arena = _arena(prog.__dict__.setdefault('_arenas', {}), 'fwd', fwd_total, dev)
        else:
            arena = torch.empty(fwd_total, dtype=torch.float32, device=dev)
        ws = rt.workspace(wsb)
        # capacity mode: the pyramid below level 0 may still be in flight on the pyramid lane (metadata.SIDE_PYRAMID)
        ready = next((d.ready for d in run.downs if d.ready is not None), None)
        wait = ready.cuda_event if ready is not None else None
        _lib.stamp('prog<')
        _lib.call('sgnn_prog_forward', ops.ctypes.data, opf.ctypes.data, nops, bufs.ctypes.data, nbuf, n_ext,
                  lev_n.ctypes.data, lev_ld.ctypes.data, run.tabs[0].ctypes.data, run.tabs[1].ctypes.data,
                  run.tabs[2].ctypes.data, run.tabs[3].ctypes.data, run.tabs[4].ctypes.data, ncls,
                  run.pptr.ctypes.data, len(params),
                  run.eptr.ctypes.data, run.iptr.ctypes.data, len(run.idx), arena.data_ptr(), fwd_total,
                  keep.ctypes.data, int(run.training) | (2 if infer else 0), wait, ws.data_ptr(), wsb)
        _lib.stamp('prog>')
        run.offsets = {}
        outs = []
        for b in run.out_bufs:
            off = _lib.query('sgnn_prog_buffer_offset', *qa, int(infer), b)
            assert off >= 0
            rows, ch = int(lev_n[bufs[b, 0]]), int(bufs[b, 1])
            run.offsets[b] = (off, rows, ch)
            o = arena[off:off + rows * ch].view(rows, ch)
            # inference: the arena is dropped (or handed to the next program) right away, only the outputs stay
            outs.append(o.clone() if infer else o)
        if infer:
            ctx.run = None
            return tuple(outs)
        ctx.run = run
        ctx.save_for_backward(arena, *ext, *[p for p in params if p is not None])
        ctx.param_none = [p is None for p in params]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        run = ctx.run
        prog = run.prog
        n_ext = prog.n_ext
        saved = ctx.saved_tensors
        arena, ext = saved[0], saved[1:1 + n_ext]
        it = iter(saved[1 + n_ext:])
        params = [None if none else next(it) for none in ctx.param_none]
        dev = arena.device
        rt = runtime(dev)
        ops, opf, bufs = prog.ops_np, prog.opf_np, prog.bufs_np
        nops, nbuf, ncls = ops.shape[0], bufs.shape[0], prog.n_classes
        if PERSISTENT_ARENAS and DEFER_JOIN:   # the lane may still read this program's gradient buffers: an arena of its own
            garena = _arena(prog.__dict__.setdefault('_arenas', {}), 'bwd', run.total, dev)
        elif PERSISTENT_ARENAS:    # one gradient arena per device: program backward calls never overlap
            garena = _arena(_garenas, str(dev), run.total, dev)
        else:
            garena = torch.empty(run.total, dtype=torch.float32, device=dev)
        gout = [0] * nbuf
        held = []
        for b, g in zip(run.out_bufs, gouts):
            if g is None:
                continue
            g = g.contiguous()
            held.append(g)
            gout[b] = g.data_ptr()
        gout = _ptr_array(gout)
        # one flat gradient tensor for all trainable slots; a parameter bound to an optimizer's persistent flat
        # gradient buffer (train.FlatAdam.bind_programs: `_sgnn_flat_grad`) gets its gradient written THERE and autograd
        # sees None for it — no per-tensor accumulation node, no copy, a fixed address for graph capture
        live = prog.tensors()
        direct = [getattr(lp, '_sgnn_flat_grad', None) if (t and p is not None) else None
                  for lp, p, t in zip(live, params, prog.grad_slot)]
        sizes = [p.numel() if (t and p is not None and d is None) else 0 for p, t, d in zip(params, prog.grad_slot, direct)]
        flat = torch.empty(max(sum(sizes), 1), dtype=torch.float32, device=dev)
        gptr, views, o = [], [], 0
        for p, s, d in zip(params, sizes, direct):
            if d is not None:
                gptr.append(d.data_ptr())
                views.append(None)
            elif s:
                v = flat[o:o + s].view_as(p)
                gptr.append(v.data_ptr())
                views.append(v)
                o += s
            else:
                gptr.append(0)
                views.append(None)
        gp = _ptr_array(gptr)
        gext = [torch.empty_like(t) if ctx.needs_input_grad[1 + i] else None for i, t in enumerate(ext)]
        geptr = _ptr_array([0 if t is None else t.data_ptr() for t in gext])
        ws = rt.workspace(run.wsb)
        rt.side_lane(run.wsb)
        # deferral needs every parameter gradient to land in a buffer nobody reads before the caller's join (the flat
        # gradient buffer of train.FlatAdam); gradients handed back to autograd must be complete when this call returns
        defer = bool(DEFER_JOIN and sum(sizes) == 0)
        prev_defer = _lib.query('sgnn_prog_defer_join', int(defer))
        if defer:
            _deferred.append((arena, garena, held, ext, params, run))
        try:
            _lib.stamp('bwd<')
            _lib.call('sgnn_prog_backward', ops.ctypes.data, opf.ctypes.data, nops, bufs.ctypes.data, nbuf, n_ext,
                      run.lev_n.ctypes.data, run.lev_ld.ctypes.data, run.tabs[0].ctypes.data, run.tabs[1].ctypes.data,
                      run.tabs[2].ctypes.data, run.tabs[3].ctypes.data, run.tabs[4].ctypes.data, ncls,
                      run.pptr.ctypes.data, gp.ctypes.data,
                      len(params), run.eptr.ctypes.data, geptr.ctypes.data, run.iptr.ctypes.data, len(run.idx),
                      arena.data_ptr(), garena.data_ptr(), run.total, gout.ctypes.data, run.keep.ctypes.data,
                      int(run.training), ws.data_ptr(), run.wsb)
        finally:
            _lib.query('sgnn_prog_defer_join', prev_defer)
        _lib.stamp('bwd>')
        return (None,) + tuple(gext) + tuple(views)


def compile_or_none(chain, in_channels, tap_modules=(), sources=None, tail=()):
    try:
        return Program(chain, in_channels, tap_modules, sources, tail)
    except Unsupported:
        return None


def run_program(prog, x, training, out_bufs=None, ext=None, idx=(), extra_rows=None, extra_cnt=None):
    """x: SparseConvNetTensor at the program's level 0 (its features are the external input unless `ext` lists the
    program's source tensors).  Returns (list of output feature tensors, grids, downs).  extra_cnt: device row counts
    (int64[1] tensors) of named rows classes, capacity mode."""
    run = _Run()
    run.prog, run.training = prog, training
    run.grids, run.downs = _levels(prog, x.metadata, x.key, x.grid())
    run.out_bufs = list(out_bufs) if out_bufs is not None else [prog.out]
    run.idx = list(idx)
    run.extra_rows = dict(extra_rows or {})
    run.extra_cnt = dict(extra_cnt or {})
    tensors = [x.features] if ext is None else list(ext)
    # nothing will ask for a gradient: inference layout (buffers share storage by liveness, outputs copied out)
    run.infer = not (torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                                     for t in list(tensors) + list(prog.tensors())))
    if 'child' in prog.class_ids and 'child' not in run.extra_rows:
        run.extra_rows['child'] = 8 * run.grids[0].n
    outs = _ProgramFn.apply(run, *tensors, *prog.tensors())
    return list(outs), run.grids, run.downs


def programs_of(model):
    """Every compiled Program hanging off the modules of `model` (encoder stack: `_prog`; generative stages:
    `_stage_progs`), compiled lazily by the first forward pass."""
    out = []
    for m in model.modules():
        p = m.__dict__.get('_prog')
        if isinstance(p, Program):
            out.append(p)
        for p in (m.__dict__.get('_stage_progs') or {}).values():
            if isinstance(p, Program):
                out.append(p)
    return out
