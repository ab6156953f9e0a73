"""nn.Modules with the names, constructor arguments and state-dict layout the reference uses from
`sparseconvnet` (torch/model.py:31-47, 178-188, 253-257, 296, 380), executing on
the MI355X through libsgnn_hip.so.  Float32 only; tensors must live on the GPU.
"""
import torch
import torch.nn as nn

from .metadata import Metadata, Grid, coords_from_locs
from . import functions as F_


class SparseConvNetTensor(object):
    """features (N,C) + shared Metadata + spatial size.  The size is kept as a tuple of ints (`key`, also the
    Metadata grid key); the LongTensor(3) the reference reads (`.spatial_size`, torch/model.py:380) is built
    lazily so that the per-layer host path stays free of tensor arithmetic."""
    __slots__ = ('features', 'metadata', 'key', '_size', '_grid')

    def __init__(self, features=None, metadata=None, spatial_size=None, grid=None):
        self.features = features
        self.metadata = metadata
        if isinstance(spatial_size, tuple):
            self.key, self._size = spatial_size, None
        elif spatial_size is None:
            self.key, self._size = None, None
        else:
            self.key, self._size = tuple(int(s) for s in spatial_size), spatial_size
        self._grid = grid

    @property
    def spatial_size(self):
        if self._size is None and self.key is not None:
            self._size = torch.LongTensor(list(self.key))
        return self._size

    def get_spatial_locations(self, spatial_size=None):
        return self.metadata.getSpatialLocations(self.key if spatial_size is None else spatial_size)

    def grid(self):
        if self._grid is None:
            self._grid = self.metadata.grids[self.key]
        return self._grid

    def cuda(self):
        return self

    def __repr__(self):
        return 'SparseConvNetTensor<features=%s spatial=%s>' % (tuple(self.features.shape), list(self.key))


def _size3(spatial_size, dimension):
    if hasattr(spatial_size, '__len__'):
        return torch.LongTensor([int(s) for s in spatial_size])
    return torch.LongTensor([int(spatial_size)] * dimension)


class Sequential(nn.Sequential):
    def add(self, module):
        self._modules[str(len(self._modules))] = module
        return self

    def forward(self, x):
        for m in self._modules.values():
            x = m(x)
        return x


class ConcatTable(nn.Module):
    def add(self, module):
        self._modules[str(len(self._modules))] = module
        return self

    def forward(self, x):
        return [m(x) for m in self._modules.values()]


class AddTable(nn.Module):
    def forward(self, xs):
        f = xs[0].features
        for t in xs[1:]:
            f = F_.AddRows.apply(f, t.features)
        return SparseConvNetTensor(f, xs[0].metadata, xs[0].key, xs[0]._grid)


class JoinTable(nn.Module):
    def forward(self, xs):
        f = xs[0].features
        for t in xs[1:]:
            f = F_.ConcatRows.apply(f, None, t.features, None, f.shape[0])
        return SparseConvNetTensor(f, xs[0].metadata, xs[0].key, xs[0]._grid)


class Identity(nn.Module):
    def forward(self, x):
        return x


class InputLayer(nn.Module):
    """mode 0 only (the reference never uses another, torch/model.py:31,178,185,253): active row i is
    input row i.  `spatial_size` is a mutable LongTensor upper bound (torch/model.py:357-369)."""

    def __init__(self, dimension, spatial_size, mode=3):
        nn.Module.__init__(self)
        if dimension != 3:
            raise NotImplementedError('sgnn_amd implements dimension 3')
        if mode != 0:
            raise NotImplementedError('InputLayer mode %d: only mode 0 is on the SG-NN hot path' % mode)
        self.dimension = dimension
        self.mode = mode
        self.spatial_size = _size3(spatial_size, dimension)

    def forward(self, x):
        locs, feats = x[0], x[1]
        if not feats.is_cuda:
            raise RuntimeError('sgnn_amd.scn.InputLayer: features must be on the GPU (no CPU fallback)')
        coords = coords_from_locs(locs, feats.device)
        md = Metadata(self.dimension)
        key = tuple(int(v) for v in self.spatial_size)
        plan = getattr(coords, '_sgnn_plan', None)   # stride-2 pyramid pre-built with the compaction that made coords
        if plan is not None and plan[0].n == coords.shape[0]:
            g = plan[0]
            md.adopt(key, g, plan[1])
        else:
            g = Grid(coords, cnt=getattr(coords, '_sgnn_cnt', None))   # capacity mode: the live row count travels along
            md.set_input(key, g)
        return SparseConvNetTensor(feats, md, key, g)


class OutputLayer(nn.Module):
    def __init__(self, dimension):
        nn.Module.__init__(self)

    def forward(self, x):
        return x.features


def _conv_weight(filter_volume, nIn, nOut):
    std = (2.0 / nIn / filter_volume) ** 0.5
    return nn.Parameter(torch.Tensor(filter_volume, nIn, nOut).normal_(0, std))


class SubmanifoldConvolution(nn.Module):
    def __init__(self, dimension, nIn, nOut, filter_size, bias):
        nn.Module.__init__(self)
        if dimension != 3 or filter_size != 3:
            raise NotImplementedError('SubmanifoldConvolution: 3x3x3 only')
        self.nIn, self.nOut, self.filter_volume = nIn, nOut, 27
        self.weight = _conv_weight(27, nIn, nOut)
        self.bias = nn.Parameter(torch.zeros(nOut)) if bias else None

    def forward(self, x):
        g = x.grid()
        tab = g.subm_table()
        y = F_.SparseConv.apply(x.features, self.weight, tab, g.ld, g.n, tab, g.ld, g.n,
                                F_.CONV_TRANSPOSE_W | F_.CONV_FLIP_K, 0)
        if self.bias is not None:
            y = y + self.bias
        return SparseConvNetTensor(y, x.metadata, x.key, g)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        _accept_grouped_weight(state_dict, prefix)
        return nn.Module._load_from_state_dict(self, state_dict, prefix, *args, **kwargs)


def _accept_grouped_weight(state_dict, prefix):
    # later upstream versions store (K, groups=1, nIn, nOut)
    w = state_dict.get(prefix + 'weight')
    if w is not None and w.dim() == 4 and w.shape[1] == 1:
        state_dict[prefix + 'weight'] = w[:, 0]


class Convolution(nn.Module):
    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias):
        nn.Module.__init__(self)
        if dimension != 3 or filter_size != 2 or filter_stride != 2:
            raise NotImplementedError('Convolution: size 2 / stride 2 only')
        self.nIn, self.nOut, self.filter_volume = nIn, nOut, 8
        self.weight = _conv_weight(8, nIn, nOut)
        self.bias = nn.Parameter(torch.zeros(nOut)) if bias else None

    def forward(self, x):
        if any(v % 2 for v in x.key):
            raise ValueError('Convolution(2,2): spatial size %s is not even' % list(x.key))
        out_size = tuple(v // 2 for v in x.key)
        d = x.metadata.down2(x.key, out_size)
        y = F_.SparseConv.apply(x.features, self.weight, d.children, d.ldc, d.coarse.n, d.ptable, d.ldf, d.fine.n,
                                F_.CONV_TRANSPOSE_W, 0)
        if self.bias is not None:
            y = y + self.bias
        return SparseConvNetTensor(y, x.metadata, out_size, d.coarse)

    _load_from_state_dict = SubmanifoldConvolution._load_from_state_dict


class _DeconvFn(torch.autograd.Function):
    """y_fine[i] = x_coarse[parent[i]] W[off_i]  (transpose of Convolution(2,2) on its rulebook)."""

    @staticmethod
    def forward(ctx, x, weight, d):
        x, weight = F_._f32c(x), F_._f32c(weight)
        K, cin, cout = weight.shape
        y = F_.conv_fwd_raw(x, cin, weight, K, d.ptable, d.ldf, d.fine.n, cout, 0, 0)
        ctx.save_for_backward(x, weight)
        ctx.d = d
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        d = ctx.d
        K, cin, cout = weight.shape
        dy = F_._f32c(dy)
        dx = F_.conv_fwd_raw(dy, cout, weight, K, d.children, d.ldc, d.coarse.n, cin, F_.CONV_TRANSPOSE_W, 0)
        dw = F_.conv_dw_raw(x, cin, dy, cout, d.ptable, d.ldf, K, d.fine.n, 0)
        return dx, dw, None


class Deconvolution(nn.Module):
    """Not reached by the reference; provided for API completeness."""

    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias):
        nn.Module.__init__(self)
        if dimension != 3 or filter_size != 2 or filter_stride != 2:
            raise NotImplementedError('Deconvolution: size 2 / stride 2 only')
        self.nIn, self.nOut, self.filter_volume = nIn, nOut, 8
        self.weight = _conv_weight(8, nIn, nOut)
        self.bias = nn.Parameter(torch.zeros(nOut)) if bias else None

    def forward(self, x):
        out_size = tuple(v * 2 for v in x.key)
        d = x.metadata.down2(out_size, x.key)
        y = _DeconvFn.apply(x.features, self.weight, d)
        if self.bias is not None:
            y = y + self.bias
        return SparseConvNetTensor(y, x.metadata, out_size, d.fine)


class UnPooling(nn.Module):
    def __init__(self, dimension, pool_size, pool_stride):
        nn.Module.__init__(self)
        if dimension != 3 or pool_size != 2 or pool_stride != 2:
            raise NotImplementedError('UnPooling: size 2 / stride 2 only')

    def forward(self, x):
        out_size = tuple(v * 2 for v in x.key)
        d = x.metadata.down2(out_size, x.key)
        y = F_.UnPool.apply(x.features, d.parent, d.fine.n, d.children, d.ldc)
        return SparseConvNetTensor(y, x.metadata, out_size, d.fine)


class BatchNormalization(nn.Module):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9, affine=True, leakiness=1.0):
        nn.Module.__init__(self)
        self.nPlanes, self.eps, self.momentum, self.leakiness = nPlanes, eps, momentum, leakiness
        self.register_buffer('running_mean', torch.zeros(nPlanes))
        self.register_buffer('running_var', torch.ones(nPlanes))
        if affine:
            self.weight = nn.Parameter(torch.ones(nPlanes))
            self.bias = nn.Parameter(torch.zeros(nPlanes))
        else:
            self.weight = self.bias = None

    def forward(self, x):
        y = F_.BatchNormLeaky.apply(x.features, self.weight, self.bias, self.running_mean, self.running_var,
                                    self.eps, self.momentum, self.training, self.leakiness)
        return SparseConvNetTensor(y, x.metadata, x.key, x._grid)


class BatchNormReLU(BatchNormalization):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9):
        BatchNormalization.__init__(self, nPlanes, eps, momentum, True, 0.0)


class NetworkInNetwork(nn.Module):
    """1x1 'convolution' = dense (N,Cin)x(Cin,Cout) product: a plain library GEMM (rocBLAS via torch)."""

    def __init__(self, nIn, nOut, bias):
        nn.Module.__init__(self)
        std = (2.0 / nIn) ** 0.5
        self.weight = nn.Parameter(torch.Tensor(nIn, nOut).normal_(0, std))
        self.bias = nn.Parameter(torch.zeros(nOut)) if bias else None

    def forward(self, x):
        y = x.features @ self.weight
        if self.bias is not None:
            y = y + self.bias
        return SparseConvNetTensor(y, x.metadata, x.key, x._grid)


class SparseToDense(nn.Module):
    def __init__(self, dimension, nPlanes):
        nn.Module.__init__(self)
        self.nPlanes = nPlanes

    def forward(self, x, batch_size=None):
        g = x.grid()
        s = list(x.key)
        if batch_size is None:
            # upstream semantics: B = max batch index + 1 (one device->host read)
            batch_size = int(g.coords[:, 3].max().item()) + 1 if g.n else 0
        return F_.SparseToDenseFn.apply(x.features, g.coords, batch_size, s[0], s[1], s[2])


def FullyConvolutionalNet(dimension, reps, nPlanes, residual_blocks=False, downsample=(2, 2)):
    """The recursive U-net of SURVEY.md §2.2 (Refinement.p2 / SurfacePrediction.p2, torch/model.py:180,255);
    output channels = sum(nPlanes).  Child numbering follows .add() order so state-dict keys line up."""

    def block(m, a, b):
        if residual_blocks:
            m.add(ConcatTable()
                  .add(Identity() if a == b else NetworkInNetwork(a, b, False))
                  .add(Sequential()
                       .add(BatchNormReLU(a))
                       .add(SubmanifoldConvolution(dimension, a, b, 3, False))
                       .add(BatchNormReLU(b))
                       .add(SubmanifoldConvolution(dimension, b, b, 3, False)))
                  ).add(AddTable())
        else:
            m.add(Sequential().add(BatchNormReLU(a)).add(SubmanifoldConvolution(dimension, a, b, 3, False)))

    def U(planes):
        m = Sequential()
        for _ in range(reps):
            block(m, planes[0], planes[0])
        if len(planes) > 1:
            m.add(ConcatTable()
                  .add(Identity())
                  .add(Sequential()
                       .add(BatchNormReLU(planes[0]))
                       .add(Convolution(dimension, planes[0], planes[1], downsample[0], downsample[1], False))
                       .add(U(planes[1:]))
                       .add(UnPooling(dimension, downsample[0], downsample[1]))))
            m.add(JoinTable())
        return m

    return U(list(nPlanes))
