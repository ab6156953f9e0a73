"""CPU ORACLE — test infrastructure, NOT product code.

A plain torch-CPU / numpy restatement of the `sparseconvnet` operator contract
that SG-NN's torch/model.py relies on (reference call sites: torch/model.py:7,
31-47, 178-188, 253-257, 296, 380).  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this package.  The product
(`sgnn_amd`) never does.

PARITY STATUS: **parity unpinned** for the scn arithmetic.  The operator
library the reference imports (facebookresearch/SparseConvNet) is not vendored
under /root/reference, not pinned to a version (README.md:11 only says "uses
SparseConvNet"), and cannot be fetched.  The reference ships no tests or
golden vectors.  What *is* pinned:
  * every op here is proven equivalent to a dense torch op on the same data
    (tests/test_oracle_dense_equiv.py): Subm3 == F.conv3d(pad=1) sampled at
    active sites, Convolution(2,2) == F.conv3d(k=2,s=2), UnPooling == nearest
    upsample masked by the fine active set, BatchNormReLU == F.batch_norm+relu.
  * the reference's own composition logic (torch/model.py, imported here in the
    authoring container with this package registered as `sparseconvnet`) is
    pinned by tests/golden/*.npz (generator: tests/golden/make_golden.py).

Algorithm (same shape as upstream's CPU path): per-sample
coordinate grid -> explicit per-offset rulebook of (in_row, out_row) pairs ->
for each offset: gather rows -> mm -> indexed add.  Autograd comes from the
torch ops used (index_select / mm / index_add_).

Conventions fixed by this restatement:
  * coords are (N,4) int64 [z,y,x,batch] (scene_dataloader.py:13-36).
  * 3x3x3 offset index k = (dz+1)*9 + (dy+1)*3 + (dx+1); cross-correlation.
  * stride-2 offset index k = (z&1)*4 + (y&1)*2 + (x&1).
  * stride-2 output sites are numbered in FIRST-TOUCH order: scanning the fine
    sites in row order, a parent gets the next free index the first time one of
    its children is seen.  (Upstream uses hash-iteration order, which no
    independent implementation can reproduce; sets must be equal.)
  * InputLayer mode 0: active row i == input row i; duplicates are an error.
  * BatchNorm: biased variance to normalise, unbiased for running_var,
    eps=1e-4, momentum=0.9 (fraction of the OLD running value kept).
"""
import numpy as np
import torch
import torch.nn as nn

__all__ = [
    'InputLayer', 'OutputLayer', 'SubmanifoldConvolution', 'Convolution', 'Deconvolution',
    'UnPooling', 'BatchNormReLU', 'BatchNormalization', 'FullyConvolutionalNet', 'Sequential',
    'ConcatTable', 'AddTable', 'JoinTable', 'Identity', 'SparseToDense',
    'SparseConvNetTensor', 'Metadata', 'NetworkInNetwork',
]


# ----------------------------------------------------------------------------
# coordinate bookkeeping
# ----------------------------------------------------------------------------
def pack_keys(coords):
    """(N,4) int64 [z,y,x,b] -> uint-like int64 key b<<48|z<<32|y<<16|x (coords in [0,65535])."""
    c = np.asarray(coords, dtype=np.int64)
    return (c[:, 3] << 48) | (c[:, 0] << 32) | (c[:, 1] << 16) | c[:, 2]


class Grid(object):
    """Active sites of one resolution level: coords (N,4) int64 numpy, row i <-> site i."""

    def __init__(self, coords):
        self.coords = np.ascontiguousarray(coords, dtype=np.int64).reshape(-1, 4)
        self.n = self.coords.shape[0]
        if self.n and (self.coords.min() < 0 or self.coords[:, :3].max() > 65535 or self.coords[:, 3].max() > 32767):
            raise ValueError('coordinate out of the supported [0,65535] range')
        keys = pack_keys(self.coords)
        self.order = np.argsort(keys, kind='stable')
        self.sorted_keys = keys[self.order]
        if self.n > 1 and np.any(self.sorted_keys[1:] == self.sorted_keys[:-1]):
            raise ValueError('InputLayer(mode=0): duplicate coordinates are a caller error')
        self._subm = None

    def lookup(self, coords):
        """coords (M,4) -> row index or -1."""
        coords = np.asarray(coords, dtype=np.int64).reshape(-1, 4)
        out = np.full(coords.shape[0], -1, dtype=np.int64)
        if self.n == 0 or coords.shape[0] == 0:
            return out
        ok = np.all(coords >= 0, axis=1) & np.all(coords[:, :3] <= 65535, axis=1)
        keys = pack_keys(np.where(ok[:, None], coords, 0))
        pos = np.searchsorted(self.sorted_keys, keys)
        pos = np.minimum(pos, self.n - 1)
        hit = ok & (self.sorted_keys[pos] == keys)
        out[hit] = self.order[pos[hit]]
        return out

    def subm_rules(self, filter_size=3):
        """Neighbour table nbr[k, j] = row of site at p_j + d_k, or -1.  k order: z-major."""
        assert filter_size == 3
        if self._subm is None and FAST:
            from . import _fast
            if _fast.available:
                self._subm = _fast.subm_rules(self.coords, self.sorted_keys, self.order)
        if self._subm is None:
            nbr = np.full((27, self.n), -1, dtype=np.int64)
            k = 0
            for dz in (-1, 0, 1):
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        q = self.coords.copy()
                        q[:, 0] += dz
                        q[:, 1] += dy
                        q[:, 2] += dx
                        nbr[k] = self.lookup(q)
                        k += 1
            self._subm = nbr
        return self._subm


def down2_rules(fine):
    """Stride-2/size-2 rulebook.  Returns (coarse Grid, parent[i], offset[i]) — exactly one
    rule per fine site; coarse sites numbered in first-touch order."""
    pc = fine.coords.copy()
    off = ((pc[:, 0] & 1) << 2) | ((pc[:, 1] & 1) << 1) | (pc[:, 2] & 1)
    pc[:, :3] >>= 1
    keys = pack_keys(pc)
    if fine.n == 0:
        return Grid(np.zeros((0, 4), np.int64)), np.zeros(0, np.int64), off
    uniq, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    # first-touch order: rank unique parents by their first occurrence
    rank_of_uniq = np.empty(len(uniq), dtype=np.int64)
    rank_of_uniq[np.argsort(first, kind='stable')] = np.arange(len(uniq))
    parent = rank_of_uniq[inv]
    ccoords = np.empty((len(uniq), 4), dtype=np.int64)
    ccoords[rank_of_uniq] = pc[first]
    return Grid(ccoords), parent, off


class Metadata(object):
    """Per-forward container of grids keyed by spatial size (upstream: Metadata_3)."""

    def __init__(self, dimension=3):
        self.dimension = dimension
        self.grids = {}
        self.down = {}  # (fine_key, coarse_key) -> (parent, off)

    @staticmethod
    def key(spatial_size):
        return tuple(int(s) for s in spatial_size)

    def set_input(self, spatial_size, coords):
        self.grids[self.key(spatial_size)] = Grid(coords)

    def grid(self, spatial_size):
        return self.grids[self.key(spatial_size)]

    def getSpatialLocations(self, spatial_size):
        return torch.from_numpy(self.grid(spatial_size).coords.copy())

    def down2(self, in_size, out_size):
        k = (self.key(in_size), self.key(out_size))
        if k not in self.down:
            coarse, parent, off = down2_rules(self.grid(in_size))
            if self.key(out_size) not in self.grids:
                self.grids[self.key(out_size)] = coarse
            self.down[k] = (parent, off)
        return self.down[k]


class SparseConvNetTensor(object):
    def __init__(self, features=None, metadata=None, spatial_size=None):
        self.features = features
        self.metadata = metadata
        self.spatial_size = spatial_size

    def get_spatial_locations(self, spatial_size=None):
        return self.metadata.getSpatialLocations(self.spatial_size if spatial_size is None else spatial_size)

    def __repr__(self):
        return 'SparseConvNetTensor<oracle features=%s spatial=%s>' % (
            tuple(self.features.shape), self.spatial_size.tolist())


# ----------------------------------------------------------------------------
# functional forms (autograd through torch ops)
# ----------------------------------------------------------------------------
FAST = False   # True: convolutions and 3x3x3 rulebooks through the C/OpenMP kernels (oracle/csrc/scn_cpu.c) — only the
               # cpu_baseline leg of bench.py switches it on; parity tests and fixtures use the torch-op path below


def rule_conv(x, weight, nbr_or_pairs, n_out, tables=None):
    """Per-offset gather -> mm -> index_add (upstream CPU algorithm).
    nbr_or_pairs: list over offsets of (in_idx, out_idx) LongTensors.  tables = (tab_f (K, n_out), tab_b (K, n_in),
    flip_b): the same rules as neighbour tables — only the optional C/OpenMP mode (FAST) uses them."""
    if FAST and x.dtype == torch.float32 and weight.dtype == torch.float32:
        from . import _fast
        if _fast.available:
            if tables is not None and hasattr(_fast, 'TableConv'):
                return _fast.TableConv.apply(x, weight, tables[0], tables[1], tables[2], n_out)
            return _fast.RuleConv.apply(x, weight, nbr_or_pairs, n_out)
    out = x.new_zeros(n_out, weight.shape[2])
    for k, (i_idx, o_idx) in enumerate(nbr_or_pairs):
        if i_idx.numel() == 0:
            continue
        out = out.index_add(0, o_idx, x.index_select(0, i_idx) @ weight[k])
    return out


def pairs_from_nbr(nbr):
    pairs = []
    for k in range(nbr.shape[0]):
        o = np.nonzero(nbr[k] >= 0)[0]
        pairs.append((torch.from_numpy(nbr[k][o].copy()), torch.from_numpy(o.copy())))
    return pairs


def down_tables(parent, off, n_coarse):
    """Stride-2 rules as tables: children (8, n_coarse) = fine row per (offset, coarse row); ptable (8, n_fine) = coarse
    row of fine row i in row off[i], -1 elsewhere."""
    n = parent.shape[0]
    children = np.full((8, n_coarse), -1, dtype=np.int64)
    children[off, parent] = np.arange(n, dtype=np.int64)
    ptable = np.full((8, n), -1, dtype=np.int64)
    ptable[off, np.arange(n)] = parent
    return children, ptable, False


def pairs_from_down(parent, off, transpose=False):
    pairs = []
    for k in range(8):
        i = np.nonzero(off == k)[0]
        a, b = torch.from_numpy(i.copy()), torch.from_numpy(parent[i].copy())
        pairs.append((b, a) if transpose else (a, b))
    return pairs


# ----------------------------------------------------------------------------
# modules (names/ctor args as the reference calls them)
# ----------------------------------------------------------------------------
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
        out = SparseConvNetTensor(None, xs[0].metadata, xs[0].spatial_size)
        f = xs[0].features
        for t in xs[1:]:
            f = f + t.features
        out.features = f
        return out


class JoinTable(nn.Module):
    def forward(self, xs):
        return SparseConvNetTensor(torch.cat([t.features for t in xs], 1), xs[0].metadata, xs[0].spatial_size)


class Identity(nn.Module):
    def forward(self, x):
        return x


class InputLayer(nn.Module):
    def __init__(self, dimension, spatial_size, mode=3):
        nn.Module.__init__(self)
        self.dimension = dimension
        self.spatial_size = torch.LongTensor([int(s) for s in (spatial_size if hasattr(spatial_size, '__len__') else [spatial_size] * dimension)])
        self.mode = mode
        assert mode == 0, 'the reference only uses mode=0 (model.py:31,178,185,253)'

    def forward(self, x):
        coords, feats = x[0], x[1]
        md = Metadata(self.dimension)
        c = coords.detach().cpu().numpy().astype(np.int64)
        if c.shape[0] and np.any(c[:, :3] >= self.spatial_size.numpy()[None, :]):
            raise ValueError('coordinate outside spatial_size')
        md.set_input(self.spatial_size, c)
        return SparseConvNetTensor(feats, md, self.spatial_size.clone())


class OutputLayer(nn.Module):
    def __init__(self, dimension):
        nn.Module.__init__(self)

    def forward(self, x):
        return x.features


class SubmanifoldConvolution(nn.Module):
    def __init__(self, dimension, nIn, nOut, filter_size, bias):
        nn.Module.__init__(self)
        assert dimension == 3 and filter_size == 3
        self.nIn, self.nOut = nIn, nOut
        self.filter_volume = 27
        std = (2.0 / nIn / self.filter_volume) ** 0.5
        self.weight = nn.Parameter(torch.Tensor(self.filter_volume, nIn, nOut).normal_(0, std))
        self.bias = nn.Parameter(torch.zeros(nOut)) if bias else None

    def forward(self, x):
        g = x.metadata.grid(x.spatial_size)
        nbr = g.subm_rules(3)
        if getattr(g, '_pairs3', None) is None:       # pair lists of a grid serve every convolution of its level
            g._pairs3 = pairs_from_nbr(nbr)
        out = rule_conv(x.features, self.weight, g._pairs3, g.n, tables=(nbr, nbr, True))
        if self.bias is not None:
            out = out + self.bias
        return SparseConvNetTensor(out, x.metadata, x.spatial_size)


class Convolution(nn.Module):
    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias):
        nn.Module.__init__(self)
        assert dimension == 3 and filter_size == 2 and filter_stride == 2
        self.nIn, self.nOut = nIn, nOut
        self.filter_volume = 8
        std = (2.0 / nIn / self.filter_volume) ** 0.5
        self.weight = nn.Parameter(torch.Tensor(self.filter_volume, nIn, nOut).normal_(0, std))
        self.bias = nn.Parameter(torch.zeros(nOut)) if bias else None

    def forward(self, x):
        assert int((x.spatial_size % 2).sum()) == 0, 'stride-2 conv needs even spatial size'
        out_size = x.spatial_size // 2
        parent, off = x.metadata.down2(x.spatial_size, out_size)
        n_out = x.metadata.grid(out_size).n
        out = rule_conv(x.features, self.weight, pairs_from_down(parent, off), n_out, tables=down_tables(parent, off, n_out))
        if self.bias is not None:
            out = out + self.bias
        return SparseConvNetTensor(out, x.metadata, out_size)


class Deconvolution(nn.Module):
    """Transpose of Convolution(2,2) reusing its rulebook (not reached by the reference)."""

    def __init__(self, dimension, nIn, nOut, filter_size, filter_stride, bias):
        nn.Module.__init__(self)
        assert dimension == 3 and filter_size == 2 and filter_stride == 2
        self.nIn, self.nOut = nIn, nOut
        self.filter_volume = 8
        std = (2.0 / nIn / self.filter_volume) ** 0.5
        self.weight = nn.Parameter(torch.Tensor(self.filter_volume, nIn, nOut).normal_(0, std))
        self.bias = nn.Parameter(torch.zeros(nOut)) if bias else None

    def forward(self, x):
        out_size = x.spatial_size * 2
        parent, off = x.metadata.down2(out_size, x.spatial_size)
        n_out = x.metadata.grid(out_size).n
        out = rule_conv(x.features, self.weight, pairs_from_down(parent, off, transpose=True), n_out)
        if self.bias is not None:
            out = out + self.bias
        return SparseConvNetTensor(out, x.metadata, out_size)


class UnPooling(nn.Module):
    def __init__(self, dimension, pool_size, pool_stride):
        nn.Module.__init__(self)
        assert dimension == 3 and pool_size == 2 and pool_stride == 2

    def forward(self, x):
        out_size = x.spatial_size * 2
        parent, _ = x.metadata.down2(out_size, x.spatial_size)
        out = x.features.index_select(0, torch.from_numpy(parent.copy()))
        return SparseConvNetTensor(out, x.metadata, out_size)


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
        f = x.features
        if self.training:
            n = f.shape[0]
            mean = f.mean(0) if n else f.new_zeros(self.nPlanes)
            var = ((f - mean) ** 2).mean(0) if n else f.new_zeros(self.nPlanes)
            with torch.no_grad():
                if n:
                    self.running_mean.mul_(self.momentum).add_((1 - self.momentum) * mean.detach().to(self.running_mean.dtype))
                    unb = var.detach() * (n / max(n - 1, 1))
                    self.running_var.mul_(self.momentum).add_((1 - self.momentum) * unb.to(self.running_var.dtype))
        else:
            mean, var = self.running_mean.to(f.dtype), self.running_var.to(f.dtype)
        y = (f - mean) * torch.rsqrt(var + self.eps)
        if self.weight is not None:
            y = y * self.weight + self.bias
        if self.leakiness != 1.0:
            y = torch.where(y > 0, y, y * self.leakiness)
        return SparseConvNetTensor(y, x.metadata, x.spatial_size)


class BatchNormReLU(BatchNormalization):
    def __init__(self, nPlanes, eps=1e-4, momentum=0.9):
        BatchNormalization.__init__(self, nPlanes, eps, momentum, True, 0.0)


class NetworkInNetwork(nn.Module):
    def __init__(self, nIn, nOut, bias):
        nn.Module.__init__(self)
        std = (2.0 / nIn) ** 0.5
        self.weight = nn.Parameter(torch.Tensor(nIn, nOut).normal_(0, std))
        self.bias = nn.Parameter(torch.zeros(nOut)) if bias else None

    def forward(self, x):
        out = x.features @ self.weight
        if self.bias is not None:
            out = out + self.bias
        return SparseConvNetTensor(out, x.metadata, x.spatial_size)


class SparseToDense(nn.Module):
    def __init__(self, dimension, nPlanes):
        nn.Module.__init__(self)
        self.nPlanes = nPlanes

    def forward(self, x):
        c = torch.from_numpy(x.metadata.grid(x.spatial_size).coords.copy())
        s = [int(v) for v in x.spatial_size]
        nb = int(c[:, 3].max()) + 1 if c.shape[0] else 0
        dense = x.features.new_zeros(nb, s[0], s[1], s[2], self.nPlanes)
        if c.shape[0]:
            dense = dense.index_put((c[:, 3], c[:, 0], c[:, 1], c[:, 2]), x.features)
        return dense.permute(0, 4, 1, 2, 3).contiguous()


def FullyConvolutionalNet(dimension, reps, nPlanes, residual_blocks=False, downsample=(2, 2)):
    """Recursive U-net described in SURVEY.md §2.2 (output channels = sum(nPlanes))."""

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
