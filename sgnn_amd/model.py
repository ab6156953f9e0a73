"""GenModel on the MI355X hot path — the build's counterpart of torch/model.py:276 (GenModel) and its
sub-modules (SparseEncoderLayer :21, TSDFEncoder :69, Refinement :169, SurfacePrediction :249).

Same constructor signature, same forward contract
    forward(x=[locs (N,4) long [z,y,x,b], feats (N,C) float], loss_weights)
        -> ([locs, sdf (M,1)], outputs=[[locs_unfilt_h, (occ,sdf)_h] for h in 0..L-1])
and the same module/attribute names, so a reference checkpoint's state_dict loads unchanged
(SURVEY.md App. B).  What differs is how the generative glue runs:
  * coordinates stay on the device as int32 rows; the reference's CPU nonzero/boolean-mask indexing
    (model.py:195-207, 233-247, 319-336) becomes expand8 / ballot-prefix-sum compaction kernels;
  * concat_skip's two dense int64 indicator volumes (model.py:338-355) become a hash-grid lookup on
    the encoder level's existing grid followed by one fused gather+concat;
  * the dense 8^3 bottleneck (model.py:89-136) keeps its nn.Conv3d / BatchNorm3d
    modules as parameter holders (state-dict layout) but executes on the same HIP kernels: a dense volume
    is a fully-active level, the k4/s2 (transposed) convolutions are rulebook walks — 64 offsets seen from the
    coarse side, 8 parity groups of 8 offsets on the coarse level's neighbour table seen from the fine side
    (functions.DenseK4S2, K4S2_TAPS below) — the 1x1x1 convolutions K = 1 walks on the identity table, BatchNorm3d
    is the row BatchNorm.  (MIOpen fell back to naive 3D kernels here: ~45 % of the step in
    profiles/r01a_bench_kernel_stats_first.csv.)
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import scn
from .scn import functions as F_
from .scn import program as P_
from .scn.metadata import coords_from_locs


STAGES = True   # each generative stage (skip join .. heads) as one native program; False: per-module glue
# teacher-forced forward: all generative levels' geometry (and its host read-backs) before the first heavy kernel
TEACHER_GEOMETRY_FIRST = os.environ.get('SGNN_TEACHER_GEOMETRY_FIRST', '1') != '0'


def _k4s2_maps():
    """The 64 taps of a k4/s2/p1 convolution (or transposed convolution) grouped by the PARITY of the fine voxel they touch.
    A fine voxel 2c + j (j = its parity per axis) meets coarse voxel c + o through tap k = j + 1 - 2o on that axis, so per
    axis two taps and two coarse neighbours serve each parity: with i = o + 1 - j in {0, 1}, slot q = g*8 + i_
    (g = 4jz+2jy+jx, i_ = 4iz+2iy+ix) holds tap K4S2_TAPS[q] and reads the coarse level's 3x3x3 neighbour row K4S2_NBR[q]
    — the index structure of the generative up-sampling convolution (functions.expand_maps).  The dense k4/s2 weights
    are STORED in this slot order (DenseConv), so the fine side of every such layer — ConvTranspose3d forward and its weight
    gradient, Conv3d data gradient — runs as 8 parity groups of 8 taps on the coarse rulebook instead of walking 64 taps per
    fine voxel of which 56 are empty."""
    P, S = [], []
    for g in range(8):
        j = (g >> 2, (g >> 1) & 1, g & 1)
        for i_ in range(8):
            i = (i_ >> 2, (i_ >> 1) & 1, i_ & 1)
            k = [3 - 2 * i[a] - j[a] for a in range(3)]
            o = [i[a] - 1 + j[a] for a in range(3)]
            P.append((k[0] * 4 + k[1]) * 4 + k[2])
            S.append((o[0] + 1) * 9 + (o[1] + 1) * 3 + (o[2] + 1))
    assert sorted(P) == list(range(64))
    return P, S


K4S2_TAPS, K4S2_NBR = _k4s2_maps()
K4S2_SLOT = [K4S2_TAPS.index(t) for t in range(64)]     # slot of tap (kz*4+ky)*4+kx
_K4S2_INDEX = {}


def _k4s2_index(which, device):
    """K4S2_TAPS / K4S2_SLOT as an index tensor on `device`, built once per device (torch.tensor(list, device=...) is a
    synchronous pageable H2D copy: a host sync per layer, and illegal inside a stream capture — ADVICE r5)."""
    key = (which, str(device))
    t = _K4S2_INDEX.get(key)
    if t is None:
        t = _K4S2_INDEX[key] = torch.tensor(K4S2_TAPS if which == 'taps' else K4S2_SLOT, device=device)
    return t
# 8 parity groups on the coarse rulebook (default) / 0: the 64-tap walk over the fine rows (A/B measurements, parity test)
DENSE_PARITY = os.environ.get('SGNN_DENSE_PARITY', '1') != '0'


class DenseConv(nn.Module):
    """Parameter holder of an nn.Conv3d / nn.ConvTranspose3d (bias=False, torch/model.py:89-136) whose `weight` lives
    in the layout the rulebook kernels read — (K = k^3 taps, Cin, Cout), the taps of a k4/s2 layer in parity-group order
    (K4S2_TAPS) — instead of torch's (Cout, Cin, k, k, k) /
    (Cin, Cout, k, k, k): the training step no longer re-permutes six weight tensors (and their gradients) every
    iteration.  state_dict() / load_state_dict() speak the torch layout, so reference checkpoints load unchanged and
    saved ones load into the reference; initial values are drawn exactly as the torch module draws them."""

    def __init__(self, cin, cout, k, stride, pad, transposed=False):
        nn.Module.__init__(self)
        self.cin, self.cout, self.k, self.stride, self.pad, self.transposed = cin, cout, k, stride, pad, transposed
        ref = (nn.ConvTranspose3d if transposed else nn.Conv3d)(cin, cout, kernel_size=k, stride=stride, padding=pad,
                                                                 bias=False)
        self.weight = nn.Parameter(self.to_native(ref.weight.detach()))
        self.weight._sgnn_dense = self     # optimizer checkpoints convert their per-parameter state with it too
        self._register_state_dict_hook(DenseConv._save_hook)

    @property
    def parity_order(self):      # k4/s2/p1: the taps are stored in parity-group order (K4S2_TAPS)
        return self.k == 4 and self.stride == 2 and self.pad == 1

    def to_native(self, w):      # torch layout -> (K, Cin, Cout)
        perm = (2, 3, 4, 0, 1) if self.transposed else (2, 3, 4, 1, 0)
        w = w.permute(*perm).reshape(self.k ** 3, self.cin, self.cout)
        if self.parity_order:
            w = w[_k4s2_index('taps', w.device)]
        return w.contiguous()

    def to_torch(self, w):       # (K, Cin, Cout) -> torch layout
        k = self.k
        if self.parity_order:
            w = w[_k4s2_index('slot', w.device)]
        w = w.reshape(k, k, k, self.cin, self.cout)
        return (w.permute(3, 4, 0, 1, 2) if self.transposed else w.permute(4, 3, 0, 1, 2)).contiguous()

    @staticmethod
    def _save_hook(module, state, prefix, local_metadata):
        state[prefix + 'weight'] = module.to_torch(state[prefix + 'weight'])

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        w = state_dict.get(prefix + 'weight')
        want = (self.cin, self.cout) if self.transposed else (self.cout, self.cin)
        if w is not None and w.dim() == 5 and tuple(w.shape[:2]) == want:
            state_dict[prefix + 'weight'] = self.to_native(w)
        return nn.Module._load_from_state_dict(self, state_dict, prefix, *args, **kwargs)


def named_gradients(model):
    """{parameter name: gradient} with every tensor in the REFERENCE's layout (the dense convolutions keep their weights
    as (K, Cin, Cout), see DenseConv) — what a test or a tool that compares against the reference's `.grad`s wants."""
    owners = {}
    for mn, mod in model.named_modules():
        if isinstance(mod, DenseConv):
            owners[(mn + '.' if mn else '') + 'weight'] = mod
    out = {}
    for n, p in model.named_parameters():
        g = p.grad
        out[n] = owners[n].to_torch(g) if (g is not None and n in owners) else g
    return out


def _dense_block(cin, cout, k, stride, pad, transposed=False):
    return nn.Sequential(DenseConv(cin, cout, k, stride, pad, transposed), nn.BatchNorm3d(cout), nn.ReLU(True))


class SparseEncoderLayer(nn.Module):
    def __init__(self, nf_in, nf, input_sparsetensor, return_sparsetensor, max_data_size):
        nn.Module.__init__(self)
        self.nf_in, self.nf = nf_in, nf
        self.input_sparsetensor, self.return_sparsetensor = input_sparsetensor, return_sparsetensor
        self.max_data_size = max_data_size
        if not input_sparsetensor:
            self.p0 = scn.InputLayer(3, max_data_size, mode=0)
        self.p1 = scn.SubmanifoldConvolution(3, nf_in, nf, 3, False)
        body = scn.Sequential()
        for _ in range(2):
            body.add(scn.BatchNormReLU(nf)).add(scn.SubmanifoldConvolution(3, nf, nf, 3, False))
        self.p2 = scn.Sequential().add(scn.ConcatTable().add(scn.Identity()).add(body)).add(scn.AddTable())
        self.p2.add(scn.BatchNormReLU(nf))
        self.p3 = scn.Sequential().add(scn.Convolution(3, nf, nf, 2, 2, False)).add(scn.BatchNormReLU(nf))
        if not return_sparsetensor:
            self.p4 = scn.SparseToDense(3, nf)

    def forward(self, x, batch_size=None, densify=True):
        if not self.input_sparsetensor:
            x = self.p0(x)
        skip = self.p2(self.p1(x))
        x = self.p3(skip)
        if self.return_sparsetensor:
            return x, [skip]
        return (self.p4(x, batch_size) if densify else x), [skip, x]


class TSDFEncoder(nn.Module):
    def __init__(self, nf_in, nf_per_level, nf_out, use_skip_sparse, use_skip_dense, input_volume_size):
        nn.Module.__init__(self)
        assert isinstance(nf_per_level, list)
        self.use_skip_sparse, self.use_skip_dense = use_skip_sparse, use_skip_dense
        layers = []
        for lv, nf in enumerate(nf_per_level):
            size = (np.array(input_volume_size) // (lv + 1)).tolist()  # model.py:79; only level 0's is read
            layers.append(SparseEncoderLayer(nf_in if lv == 0 else nf_per_level[lv - 1], nf, lv > 0,
                                             lv < len(nf_per_level) - 1, size))
        self.process_sparse = nn.Sequential(*layers)
        nf = nf_per_level[-1]
        nf0, nf1 = nf * 3 // 2, nf * 2
        nf2 = nf1
        self.encode_dense0 = _dense_block(nf, nf0, 4, 2, 1)
        self.encode_dense1 = _dense_block(nf0, nf1, 4, 2, 1)
        self.bottleneck_dense2 = _dense_block(nf1, nf2, 1, 1, 0)
        nf3 = nf2 if not use_skip_dense else nf1 + nf2
        nf4 = nf3 // 2
        self.decode_dense3 = _dense_block(nf3, nf4, 4, 2, 1, True)
        if use_skip_dense:
            nf4 += nf0
        nf5 = nf4 // 2
        self.decode_dense4 = _dense_block(nf4, nf5, 4, 2, 1, True)
        self.final = _dense_block(nf5, nf_out, 1, 1, 0)
        self.occpred = nn.Sequential(nn.Conv3d(nf_out, 1, kernel_size=1, bias=False))
        self.sdfpred = nn.Sequential(nn.Conv3d(nf_out, 1, kernel_size=1, bias=False))

    def forward(self, x, batch_size=None, capacity=None):
        """Returns (feat_rows (B*V, nf_out), out_rows (B*V, 2) [occ logit, sdf], skips, dense geometry);
        rows are the dense coarse volume in batch-major raster order (== permute(0,2,3,4,1).view(-1, C)).
        capacity (scn.capacity.Capacity): capacity mode — x[0] carries its live row count, no host read-back."""
        skips = []
        n_layers = len(self.process_sparse)
        prog = self._sparse_program() if P_.ENABLED else None
        if prog is not None:
            # the three sparse encoder levels (p1, p2, p3 each) as one native program; taps = the p2 outputs
            x0 = self.process_sparse[0].p0(x)
            if capacity is not None:                    # stride-2 levels with device-side row counts
                x0.metadata.prebuild(x0.key, n_layers, capacity=(capacity.enc, capacity.enc_counts()))
            else:
                x0.metadata.prebuild(x0.key, n_layers)      # all stride-2 levels, one host read-back
            taps = [prog.taps[id(l.p2)][0] for l in self.process_sparse]
            outs, grids, _ = P_.run_program(prog, x0, self.training, taps + [prog.out])
            keys = [x0.key]
            for _ in range(n_layers):
                keys.append(tuple(v // 2 for v in keys[-1]))
            ts = [scn.SparseConvNetTensor(outs[i], x0.metadata, keys[i], grids[i]) for i in range(n_layers + 1)]
            x = ts[-1]
            if self.use_skip_sparse:
                skips = ts
        else:
            if capacity is not None:
                raise RuntimeError('capacity mode runs on the native program path (sgnn_amd.scn.program.ENABLED)')
            for i, layer in enumerate(self.process_sparse):
                x, ft = layer(x, batch_size, densify=(i < n_layers - 1))
                if self.use_skip_sparse:
                    skips.extend(ft)
        g = x.grid()
        dims = x.key
        if batch_size is None:  # upstream SparseToDense semantics: B = max batch index + 1 (one host read)
            batch_size = int(g.coords[:, 3].max().item()) + 1 if g.n else 0
        geo = dense_geometry(batch_size, dims, x.features.device)
        rows = F_.ScatterRows.apply(x.features, geo.grid.lookup(g.coords, g.cnt), geo.grid.n, g.cnt)
        t0, t1 = geo.level(0), geo.level(1)
        tr = []
        enc0 = _bn3d_relu(self.encode_dense0[1], _dense_conv(rows, self.encode_dense0[0], t0, down=True), tr)
        enc1 = _bn3d_relu(self.encode_dense1[1], _dense_conv(enc0, self.encode_dense1[0], t1, down=True), tr)
        bott = _bn3d_relu(self.bottleneck_dense2[1], _conv1x1(enc1, self.bottleneck_dense2[0]), tr)
        d_in = _join(bott, enc1) if self.use_skip_dense else bott
        dec0 = _bn3d_relu(self.decode_dense3[1], _dense_conv(d_in, self.decode_dense3[0], t1, down=False), tr)
        d_in = _join(dec0, enc0) if self.use_skip_dense else dec0
        xr = _bn3d_relu(self.decode_dense4[1], _dense_conv(d_in, self.decode_dense4[0], t0, down=False), tr)
        xr = _bn3d_relu(self.final[1], _conv1x1(xr, self.final[0]), tr)
        # both 1x1 heads in one pass; column 0 = occupancy logit, 1 = sdf (model.py:163-165)
        if tr:
            torch._foreach_add_(tr, 1)
        w = torch.cat([self.occpred[0].weight.view(1, -1), self.sdfpred[0].weight.view(1, -1)], 0)
        return xr, F_.RowLinear.apply(xr, w, None), skips, geo


    def _sparse_program(self):
        if not hasattr(self, '_prog'):
            chain, taps = [], []
            for l in self.process_sparse:
                chain += [l.p1, l.p2, l.p3]
                taps.append(l.p2)
            object.__setattr__(self, '_prog', P_.compile_or_none(chain, self.process_sparse[0].nf_in, taps))
        return self._prog


# ---- dense bottleneck on the sparse-conv kernels ------------------------------------------------------------
class _DenseGeometry(object):
    """Fully-active grid of a dense (B, d0, d1, d2) volume plus the k4/s2/p1 rulebooks between its pyramid
    levels.  Depends only on the shape, so it is built once and cached."""

    def __init__(self, batch, dims, device):
        self.batch, self.dims, self.device = batch, dims, device
        self.coords = F_.dense_coords(batch, dims[0], dims[1], dims[2], device)
        self.coords._sgnn_bounds = (int(batch), int(dims[0]), int(dims[1]), int(dims[2]))   # travels to every generated level
        self.grid = scn.Grid(self.coords)
        self._levels = {}

    def level(self, lv):
        """Tables between pyramid level lv (fine, dims / 2^lv) and lv+1 (coarse), an object with
          tdown (64 x ld_c): tdown[q][o] = fine row feeding coarse voxel o through the tap of slot q (K4S2_TAPS[q] =
                (kz*4+ky)*4+kx: input index 2*o - 1 + k per axis, zero padding 1);
          tup (64 x ld_f):   tup[q][i] = coarse row that fine voxel i meets through that tap (56 of the 64 are -1);
          nbr27 (27 x ld_c): the coarse level's dense 3x3x3 neighbour table (row (oz+1)*9+(oy+1)*3+(ox+1));
          slots (64):        K4S2_NBR — the nbr27 row of slot q;
          child_of_raster (n_f) / raster_of_child (n_f): fine voxel <-> its position 8*coarse + parity in child order.
        Rows of both levels are batch-major raster order."""
        if lv not in self._levels:
            fd = [d >> lv for d in self.dims]
            if any(d % 2 for d in fd):
                raise ValueError('dense level dims %s are not even' % (fd,))
            cd = [d // 2 for d in fd]
            dev, B = self.device, self.batch
            taps = _k4s2_index('taps', dev)
            kz, ky, kx = (taps // 16).view(64, 1), ((taps // 4) % 4).view(64, 1), (taps % 4).view(64, 1)

            def unravel(n, d):
                r = torch.arange(n, device=dev)
                x_ = r % d[2]
                y_ = (r // d[2]) % d[1]
                z_ = (r // (d[1] * d[2])) % d[0]
                b_ = r // (d[0] * d[1] * d[2])
                return b_.view(1, -1), z_.view(1, -1), y_.view(1, -1), x_.view(1, -1)

            n_c, n_f = B * cd[0] * cd[1] * cd[2], B * fd[0] * fd[1] * fd[2]
            ld_c, ld_f = ((max(n_c, 1) + 255) // 256) * 256, ((max(n_f, 1) + 255) // 256) * 256
            b_, z_, y_, x_ = unravel(n_c, cd)
            iz, iy, ix = 2 * z_ - 1 + kz, 2 * y_ - 1 + ky, 2 * x_ - 1 + kx
            ok = (iz >= 0) & (iz < fd[0]) & (iy >= 0) & (iy < fd[1]) & (ix >= 0) & (ix < fd[2])
            down = torch.full((64, ld_c), -1, dtype=torch.int32, device=dev)
            down[:, :n_c] = torch.where(ok, ((b_ * fd[0] + iz) * fd[1] + iy) * fd[2] + ix, -1).to(torch.int32)
            o = torch.arange(27, device=dev)
            oz, oy, ox = (o // 9 - 1).view(27, 1), ((o // 3) % 3 - 1).view(27, 1), (o % 3 - 1).view(27, 1)
            nz, ny, nx = z_ + oz, y_ + oy, x_ + ox
            ok = (nz >= 0) & (nz < cd[0]) & (ny >= 0) & (ny < cd[1]) & (nx >= 0) & (nx < cd[2])
            nbr27 = torch.full((27, ld_c), -1, dtype=torch.int32, device=dev)
            nbr27[:, :n_c] = torch.where(ok, ((b_ * cd[0] + nz) * cd[1] + ny) * cd[2] + nx, -1).to(torch.int32)
            b_, z_, y_, x_ = unravel(n_f, fd)
            tz, ty, tx = z_ + 1 - kz, y_ + 1 - ky, x_ + 1 - kx
            ok = ((tz >= 0) & (tz % 2 == 0) & (tz // 2 < cd[0]) & (ty >= 0) & (ty % 2 == 0) & (ty // 2 < cd[1]) &
                  (tx >= 0) & (tx % 2 == 0) & (tx // 2 < cd[2]))
            up = torch.full((64, ld_f), -1, dtype=torch.int32, device=dev)
            up[:, :n_f] = torch.where(ok, ((b_ * cd[0] + tz // 2) * cd[1] + ty // 2) * cd[2] + tx // 2, -1).to(torch.int32)
            coarse = ((b_ * cd[0] + z_ // 2) * cd[1] + y_ // 2) * cd[2] + x_ // 2
            child = (8 * coarse + 4 * (z_ % 2) + 2 * (y_ % 2) + (x_ % 2)).view(-1)
            raster = torch.empty_like(child)
            raster[child] = torch.arange(n_f, device=dev)
            t = _K4S2Level()
            t.tdown, t.ld_c, t.n_c = down.contiguous().view(-1), ld_c, n_c
            t.tup, t.ld_f, t.n_f = up.contiguous().view(-1), ld_f, n_f
            t.nbr27 = nbr27.contiguous().view(-1)
            t.slots = torch.tensor(K4S2_NBR, dtype=torch.int32, device=dev)
            t.child_of_raster, t.raster_of_child = child.to(torch.int32).contiguous(), raster.to(torch.int32).contiguous()
            self._levels[lv] = t
        return self._levels[lv]


class _K4S2Level(object):
    """Static index tables of one dense k4/s2 level pair (see _DenseGeometry.level)."""
    __slots__ = ('tdown', 'ld_c', 'n_c', 'tup', 'ld_f', 'n_f', 'nbr27', 'slots', 'child_of_raster', 'raster_of_child')


_dense_cache = {}


def dense_geometry(batch, dims, device):
    key = (int(batch), tuple(int(d) for d in dims), str(device))
    geo = _dense_cache.get(key)
    if geo is None:
        if len(_dense_cache) > 8:
            _dense_cache.clear()
        geo = _dense_cache[key] = _DenseGeometry(key[0], key[1], device)
    return geo


def _dense_conv(rows, conv, tables, down):
    """nn.Conv3d(k4,s2,p1) (down=True) or nn.ConvTranspose3d(k4,s2,p1) (down=False) on channel-last rows."""
    w = conv.weight
    if isinstance(conv, DenseConv):        # already (64, Cin, Cout) in parity-group slot order
        wk = w
    else:
        if down:     # nn.Conv3d weight (Cout, Cin, 4,4,4) -> (64 taps, Cin, Cout)
            wk = w.permute(2, 3, 4, 1, 0).reshape(64, w.shape[1], w.shape[0])
        else:        # nn.ConvTranspose3d (Cin, Cout, 4,4,4) -> (64 taps, Cin, Cout)
            wk = w.permute(2, 3, 4, 0, 1).reshape(64, w.shape[0], w.shape[1])
        wk = wk[_k4s2_index('taps', w.device)]
    return F_.DenseK4S2.apply(rows, wk, tables, bool(down), DENSE_PARITY)


_ident_tables = {}


def _conv1x1(rows, conv):
    """1x1x1 convolution of a dense level = a K = 1 rulebook convolution on the identity table (same MFMA kernels as every
    other convolution; the rocBLAS GEMMs this replaces took 45-66 us each for their tiny shapes)."""
    if not isinstance(conv, DenseConv):
        return rows @ conv.weight.view(conv.weight.shape[0], -1).t()
    n = int(rows.shape[0])
    key = (n, str(rows.device))
    tab = _ident_tables.get(key)
    if tab is None:
        if len(_ident_tables) > 16:
            _ident_tables.clear()
        ld = ((max(n, 1) + 255) // 256) * 256
        t = torch.full((ld,), -1, dtype=torch.int32, device=rows.device)
        t[:n] = torch.arange(n, dtype=torch.int32, device=rows.device)
        tab = _ident_tables[key] = (t, ld)
    t, ld = tab
    return F_.SparseConv.apply(rows, conv.weight, t, ld, n, t, ld, n, F_.CONV_TRANSPOSE_W, 0)


def _bn3d_relu(bn, rows, tracked=None):
    """nn.BatchNorm3d + ReLU on channel-last rows (statistics over all B*V voxels, like BatchNorm3d)."""
    y = F_.BatchNormLeaky.apply(rows, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                1.0 - bn.momentum, bn.training, 0.0)
    if bn.training and bn.num_batches_tracked is not None:
        if tracked is None:
            bn.num_batches_tracked.add_(1)
        else:
            tracked.append(bn.num_batches_tracked)      # the caller advances all its counters in ONE launch
    return y


def _join(a, b):
    return F_.ConcatRows.apply(a, None, b, None, a.shape[0])


def _cached_program(owner, chain, in_channels):
    if not hasattr(owner, '_prog'):
        object.__setattr__(owner, '_prog', P_.compile_or_none(chain, in_channels))
    return owner._prog


class Refinement(nn.Module):
    def __init__(self, nf_in, nf, pass_occ, pass_feats, max_data_size, truncation=3):
        nn.Module.__init__(self)
        self.pass_occ, self.pass_feats = pass_occ, pass_feats
        self.nf_in, self.nf, self.truncation = nf_in, nf, truncation
        self.p0 = scn.InputLayer(3, max_data_size, mode=0)
        self.p1 = scn.SubmanifoldConvolution(3, nf_in, nf, 3, False)
        self.p2 = scn.FullyConvolutionalNet(3, reps=1, nPlanes=[nf, nf, nf], residual_blocks=True)
        self.p3 = scn.BatchNormReLU(nf * 3)
        self.p4 = scn.OutputLayer(3)
        self.n0 = scn.InputLayer(3, max_data_size, mode=0)
        self.n1 = scn.SubmanifoldConvolution(3, nf * 3, nf, 3, False)
        self.n2 = scn.BatchNormReLU(nf)
        self.n3 = scn.OutputLayer(3)
        self.linear = nn.Linear(nf, 1)
        self.linearsdf = nn.Linear(nf, 1)
        self.fused_expand = True   # False: materialise the 8x replicated features like the reference does
        self.plan_depth = 2        # stride-2 levels of the NEXT stage's U-Net (FullyConvolutionalNet nPlanes of 3)

    def forward(self, x):
        coords = x[0]
        if len(coords) == 0:
            return [[], []], [[], []]
        prog = _cached_program(self, [self.p1, self.p2, self.p3], self.nf_in) if P_.ENABLED else None
        if prog is not None:
            x0 = self.p0(x)
            outs, grids, _ = P_.run_program(prog, x0, self.training)
            t = scn.SparseConvNetTensor(outs[0], x0.metadata, x0.key, grids[0])
        else:
            t = self.p3(self.p2(self.p1(self.p0(x))))
        f = self.p4(t)
        # 8-child expansion (model.py:192-207): child row 8i+j, j = 4dz+2dy+dx, features replicated
        coords_next = F_.expand8_coords(self.p0_coords(x))
        if self.fused_expand and self.n1.bias is None:
            # n0 -> n1 on the replicated features == a grouped convolution on the parent rulebook (sgnn_hip.h)
            y_pre = F_.expand_conv(f, self.n1.weight, t.grid())
            y = self.n3(self.n2(scn.SparseConvNetTensor(y_pre, None, None)))
        else:
            feats_next = F_.RepeatRows.apply(f, 8)
            y = self.n3(self.n2(self.n1(self.n0([coords_next, feats_next]))))
        # occupancy + sdf heads as one (nf -> 2) product; column 0 = occ logit, 1 = sdf (model.py:230-231,240)
        out = F_.RowLinear.apply(y, torch.cat([self.linear.weight, self.linearsdf.weight], 0),
                                 torch.cat([self.linear.bias, self.linearsdf.bias], 0))
        n_all = out.shape[0]
        # stable, == torch boolean indexing order; the next stage's U-Net pyramid is built in the same submission
        sel, cnt, locs = F_.compact_sigmoid_plan(out.detach(), 2, n_all, coords_next, self.plan_depth)
        if self.pass_feats and self.pass_occ:
            feats = F_.ConcatRows.apply(y, sel, out, sel, cnt)          # [feats | occ,sdf] (model.py:242)
        elif self.pass_feats:
            feats = F_.GatherRows.apply(y, sel, cnt)
        else:
            feats = F_.GatherRows.apply(out, sel, cnt)
        return [locs, feats], [F_.coords_to_i64(coords_next), out]

    @staticmethod
    def p0_coords(x):
        return coords_from_locs(x[0], x[1].device)

    def stage(self, prev, skip):
        """The whole level as ONE native program (sgnn_amd/scn/program.py): skip join + p1 + p2 + p3 + up-sampling
        convolution n1 + n2 + both heads.  prev = (a, b, sel, locs, cnt): this level's input rows are
        [a[sel] | b[sel]] (what forward() receives as x[1], before concat_skip) at the sites `locs`; skip = (Grid,
        features) of the encoder level or None.  Returns (y (8cnt,nf), out (8cnt,2) [occ, sdf], child coords), or None
        when the configuration is not compilable (the caller then uses forward())."""
        prog, ext, idx, extra = _stage_program(self, prev, skip, [self.p1, self.p2, self.p3], self.nf_in,
                                               [('expand', self.n1), ('bn', self.n2),
                                                ('linear', [self.linear, self.linearsdf])])
        if prog is None:
            return None
        locs = prev[3]
        x0 = self.p0([locs, ext[0]])
        cnt8 = getattr(locs, '_sgnn_cnt8', None)              # capacity mode: live rows of the 8-child expansion
        outs, _, _ = P_.run_program(prog, x0, self.training, [prog.taps[id(self.n2)][0], prog.taps[id(self.linear)][0]],
                                    ext=ext, idx=idx, extra_rows=extra,
                                    extra_cnt=None if cnt8 is None else {'child': cnt8})
        children = getattr(locs, '_sgnn_children', None)     # already made by GenModel._teacher_plans
        return outs[0], outs[1], (children if children is not None else F_.expand8_coords(locs, with_i64=True))


def _stage_program(owner, prev, skip, chain, nf_in, tail):
    """Program + run-time inputs of a generative stage (see Refinement.stage)."""
    a, b, sel, locs, cnt = prev
    srcs, ext = [None, None, None], []
    for k, t in enumerate((a, b)):
        if t is not None:
            srcs[k] = ('prev', int(t.shape[1]), 0)
            ext.append(t)
    idx = [sel]
    extra = {'prev': int(ext[0].shape[0])}
    if skip is not None:
        grid_from, feats_from = skip
        if grid_from.n == 0:
            return None, None, None, None
        srcs[2] = ('skip', int(feats_from.shape[1]), 1)
        ext.append(feats_from)
        idx.append(grid_from.lookup(locs, getattr(locs, '_sgnn_cnt', None)))
        extra['skip'] = grid_from.n
    cache = owner.__dict__.setdefault('_stage_progs', {})
    key = tuple(srcs)
    if key not in cache:
        cache[key] = P_.compile_or_none(chain, nf_in, sources=srcs, tail=tail)
    return cache[key], ext, idx, extra


class SurfacePrediction(nn.Module):
    def __init__(self, nf_in, nf, nf_out, max_data_size):
        nn.Module.__init__(self)
        self.p0 = scn.InputLayer(3, max_data_size, mode=0)
        self.p1 = scn.SubmanifoldConvolution(3, nf_in, nf, 3, False)
        self.p2 = scn.FullyConvolutionalNet(3, reps=1, nPlanes=[nf, nf, nf], residual_blocks=True)
        self.p3 = scn.BatchNormReLU(nf * 3)
        self.p4 = scn.OutputLayer(3)
        self.linear = nn.Linear(nf * 3, nf_out)

    def forward(self, x):
        if len(x[0]) == 0:
            return [], []
        prog = _cached_program(self, [self.p1, self.p2, self.p3], self.p1.nIn) if P_.ENABLED else None
        if prog is not None:
            outs, _, _ = P_.run_program(prog, self.p0(x), self.training)
            f = outs[0]
        else:
            f = self.p4(self.p3(self.p2(self.p1(self.p0(x)))))
        return F_.RowLinear.apply(f, self.linear.weight, self.linear.bias)

    def stage(self, prev, skip):
        """As Refinement.stage: skip join + p1 + p2 + p3 + linear in one native program; returns sdf (cnt, 1) or None."""
        prog, ext, idx, extra = _stage_program(self, prev, skip, [self.p1, self.p2, self.p3], self.p1.nIn,
                                               [('linear', [self.linear])])
        if prog is None:
            return None
        x0 = self.p0([prev[3], ext[0]])
        outs, _, _ = P_.run_program(prog, x0, self.training, [prog.taps[id(self.linear)][0]], ext=ext, idx=idx,
                                    extra_rows=extra)
        return outs[0]


class GenModel(nn.Module):
    def __init__(self, encoder_dim, input_dim, input_nf, nf_coarse, nf, num_hierarchy_levels, pass_occ, pass_feats,
                 use_skip_sparse, use_skip_dense, truncation=3):
        nn.Module.__init__(self)
        self.truncation, self.pass_occ, self.pass_feats = truncation, pass_occ, pass_feats
        self.use_skip_sparse = use_skip_sparse
        L = num_hierarchy_levels
        if not isinstance(input_dim, (list, tuple, np.ndarray)):
            input_dim = [input_dim, input_dim, input_dim]
        if L > 2:
            self.nf_per_level = [int(encoder_dim * (1 + float(k) / (L - 2))) for k in range(L - 1)]
        else:
            self.nf_per_level = [encoder_dim] * (L - 1)
        self.encoder = TSDFEncoder(input_nf, self.nf_per_level, nf_coarse, use_skip_sparse, use_skip_dense,
                                   input_volume_size=input_dim)
        self.refine_sizes = [(np.array(input_dim) // (2 ** k)).tolist() for k in range(L - 1)][::-1]
        self.nf_per_level.append(self.nf_per_level[-1])
        self.data_dim = 3
        self.refinement = scn.Sequential()
        for h in range(1, L):
            c = (self.nf_per_level[L - h] if use_skip_sparse else 0) + (2 if pass_occ else 0)
            if pass_feats:
                c += nf_coarse if h == 1 else nf
            self.refinement.add(Refinement(c, nf, pass_occ, pass_feats, self.refine_sizes[h - 1], truncation))
        self.PRED_SURF = True
        c = (self.nf_per_level[0] if use_skip_sparse else 0) + (2 if pass_occ else 0) + (nf if pass_feats else 0)
        self.surfacepred = SurfacePrediction(c, nf, 1, self.refine_sizes[-1])

    # -- generative glue ---------------------------------------------------------------------------------
    def dense_coarse_to_sparse(self, feat_rows, occ_rows, geo, truncation=3, plan_depth=0):
        """model.py:315-336: every coarse voxel is a candidate; keep sigmoid(occ) > 0.5 in raster order.
        feat_rows / occ_rows are the dense volume as rows (what the reference builds with permute+view)."""
        n_all = occ_rows.shape[0]
        sel, cnt, locs = F_.compact_sigmoid_plan(occ_rows.detach(), 2, n_all, geo.coords, plan_depth)
        if self.pass_occ and self.pass_feats:
            feats = F_.ConcatRows.apply(occ_rows, sel, feat_rows, sel, cnt)  # [occ,sdf | feats] (model.py:330)
        elif self.pass_occ:
            feats = F_.GatherRows.apply(occ_rows, sel, cnt)
        else:
            feats = F_.GatherRows.apply(feat_rows, sel, cnt)
        return locs, feats, [F_.coords_to_i64(geo.coords), occ_rows]

    @staticmethod
    def concat_skip(x_from, x_to, spatial_size=None, batch_size=None):
        """model.py:338-355 as a hash join: x_from = (Grid of the encoder level, its features)."""
        grid_from, feats_from = x_from
        coords_to, feats_to = x_to
        if grid_from.n == 0 or len(coords_to) == 0:
            return x_to
        coords_to = coords_from_locs(coords_to, feats_to.device)
        rows = grid_from.lookup(coords_to)
        return [coords_to, F_.ConcatRows.apply(feats_to, None, feats_from, rows, coords_to.shape[0])]

    def update_sizes(self, input_max_dim, refine_max_dim):
        """model.py:357-369 (called per scene by test_scene.py:78).  Spatial sizes are upper bounds for the
        sparse layers; only the encoder input size changes results (it fixes the dense volume).  The
        reference's in-loop array doubling is not replicated: level h gets refine*2^h
        (p0) and refine*2^(h+1) (n0)."""
        inp = (np.array(input_max_dim).reshape(-1) * np.ones(3, dtype=np.int64)).astype(np.int64)
        ref = (np.array(refine_max_dim).reshape(-1) * np.ones(3, dtype=np.int64)).astype(np.int64)
        self.encoder.process_sparse[0].p0.spatial_size[:] = torch.from_numpy(inp)
        for h in range(len(self.refinement)):
            self.refinement[h].p0.spatial_size[:] = torch.from_numpy(ref * 2 ** h)
            self.refinement[h].n0.spatial_size[:] = torch.from_numpy(ref * 2 ** (h + 1))
        self.surfacepred.p0.spatial_size[:] = torch.from_numpy(ref * 2 ** len(self.refinement))

    # -- forward -----------------------------------------------------------------------------------------
    def _runs(self, loss_weights):
        R = len(self.refinement)
        # which later stage consumes a compaction's sites (its U-Net needs a 2-level stride-2 pyramid)?
        runs = [loss_weights[h + 1] > 0 for h in range(R)] + [bool(self.PRED_SURF and loss_weights[-1] > 0)]
        for h in range(R):     # the pyramid serves the next stage that actually runs (a skipped level does not compact)
            self.refinement[h].plan_depth = 2 if any(runs[h + 1:]) else 0
        return runs

    def plan_geometry(self, locs, loss_weights, batch_size, teacher):
        """Everything a teacher-forced forward() reads back from the device, computed from the batch alone: the input
        coordinates with the encoder's stride-2 pyramid attached, and the site / index lists + pyramids of every
        generative level (_teacher_plans).  Returns `geometry` for forward(..., geometry=...), which then runs without a
        host synchronisation.  A training loop may call this for batch i+1 on a second stream while batch i's backward
        pass runs (train.GeometryPrefetcher)."""
        from .scn import metadata as MD
        dev = teacher[0].device
        coords = coords_from_locs(locs, dev)
        depth = len(self.encoder.process_sparse)
        if MD.CHAIN and depth >= 2 and coords.shape[0]:
            n = int(coords.shape[0])
            chain = MD.PendingChain(coords, n, False, depth)
            plan = chain.finalize(n, MD.runtime(dev).read_counts())
            # the level-0 Grid holds a view of `coords`; hanging the plan on `coords` itself would close a reference
            # cycle through the view's base that the garbage collector cannot see (one leaked pyramid per step)
            coords = coords[:]
            coords._sgnn_plan = plan
        enc = self.encoder
        dims = tuple(int(v) >> depth for v in enc.process_sparse[0].p0.spatial_size)
        plans = self._teacher_plans(dense_geometry(batch_size, dims, dev), self._runs(loss_weights), teacher)
        return coords, plans

    def forward(self, x, loss_weights, batch_size=None, teacher=None, geometry=None, capacity=None):
        """capacity (scn.capacity.Capacity, optional; not in the reference): capacity mode — every level's tensors have
        the plan's capacities, the live row counts stay on the device (results carry them as `_sgnn_cnt`; rows past a
        count are undefined) and the call performs NO host read-back, so a training step can be captured in a HIP
        graph (train.GraphStep).  x[0] may have more rows than the batch (x[0]._sgnn_cnt or capacity.input_cnt()
        holds the live count).  Needs batch_size and the native stage path.
        teacher (optional, not in the reference): list of the L dense target occupancy volumes (loss.compute_targets'
        target_for_occs); when given, every generative mask is `target occupancy == 1` at the candidate site instead of
        sigmoid(predicted occupancy) > 0.5, so the per-level site counts do not depend on the weights (bench.py).
        geometry: result of plan_geometry() for this batch and these loss weights (teacher-forced only)."""
        if geometry is not None:
            if teacher is None or not (P_.ENABLED and STAGES):
                raise ValueError('geometry plans belong to the teacher-forced native stage path')
            x = [geometry[0], x[1]]
        if capacity is not None:
            if geometry is not None or batch_size is None or not (P_.ENABLED and STAGES):
                raise ValueError('capacity mode: pass batch_size, no geometry plan, native stage path on')
            if getattr(x[0], '_sgnn_cnt', None) is None:
                x[0]._sgnn_cnt = capacity.input_cnt()
        x = [coords_from_locs(x[0], x[1].device), x[1]]
        R = len(self.refinement)
        runs = self._runs(loss_weights)
        plans = None if geometry is None else geometry[1]
        if (plans is None and teacher is not None and batch_size is not None and TEACHER_GEOMETRY_FIRST and P_.ENABLED
                and STAGES and capacity is None):
            # teacher-forced masks depend on the data only: build the site lists, index lists and stride-2 pyramids of
            # ALL generative levels now, while the GPU queue is short.  Their row-count read-backs then wait for a few
            # small kernels each instead of for a whole stage's convolutions, and everything after them — encoder,
            # stages, loss, backward, Adam — is issued without a single host synchronisation.
            enc = self.encoder
            dims = tuple(int(v) >> len(enc.process_sparse) for v in enc.process_sparse[0].p0.spatial_size)
            plans = self._teacher_plans(dense_geometry(batch_size, dims, x[1].device), runs, teacher)
        feat_rows, occ_rows, skips, geo = self.encoder(x, batch_size, capacity=capacity)
        if self.use_skip_sparse:
            skips = [(t.grid(), t.features) for t in skips]
        if P_.ENABLED and STAGES:
            res = self._forward_stages(feat_rows, occ_rows, skips, geo, loss_weights, runs, teacher, plans, capacity)
            if res is not None:
                if capacity is not None:
                    # a pyramid built on the pyramid lane that no program consumed (a level whose stage did not run) must
                    # still be joined: the training stream owns every buffer it wrote, and a capture needs all forks closed
                    from .scn.metadata import join_pyramid_lane
                    join_pyramid_lane(feat_rows.device)
                    return res
                if not self.training:     # inference: report input errors (duplicate / out-of-range sites) from THIS call
                    from .scn.metadata import runtime
                    runtime(feat_rows.device).check_status()
                return res
        if teacher is not None:
            raise NotImplementedError('teacher forcing runs on the native stage path only')
        outputs = []
        locs, feats, out0 = self.dense_coarse_to_sparse(feat_rows, occ_rows, geo, truncation=3,
                                                        plan_depth=2 if runs[0] else 0)
        outputs.append(out0)
        xs = [locs, feats]
        for h in range(R):
            if loss_weights[h + 1] > 0:
                if self.use_skip_sparse:
                    xs = self.concat_skip(skips[R - h], xs)
                xs, occ = self.refinement[h](xs)
                outputs.append(occ)
            else:
                outputs.append([[], []])
        locs = xs[0]
        if self.PRED_SURF and loss_weights[-1] > 0:
            if self.use_skip_sparse:
                xs = self.concat_skip(skips[0], xs)
            sdf = self.surfacepred(xs)
            locs_out = F_.coords_to_i64(locs) if len(locs) else locs
            return [locs_out, sdf], outputs
        return [[], []], outputs

    def _teacher_plans(self, geo, runs, teacher):
        """The compactions of _forward_stages for teacher-forced masks, without the features: plans[0] for the coarse
        volume, plans[h + 1] after refinement h (None where _forward_stages would not compact)."""
        R = len(self.refinement)
        plan = F_.compact_sigmoid_plan(geo.coords, 2, int(geo.coords.shape[0]), geo.coords, 2 if runs[0] else 0, teacher[0])
        plans = [plan]
        for h in range(R):
            if not runs[h] or plan[1] == 0:
                plans.append(None)
                continue
            locs = plan[2]
            locs._sgnn_children = F_.expand8_coords(locs)
            plan = F_.compact_sigmoid_plan(locs._sgnn_children, 2, 8 * plan[1], locs._sgnn_children,
                                           self.refinement[h].plan_depth, teacher[h + 1])
            plans.append(plan)
        return plans

    def _forward_stages(self, feat_rows, occ_rows, skips, geo, loss_weights, runs, teacher=None, plans=None,
                        capacity=None):
        """forward() with every generative stage as one native program (Refinement.stage / SurfacePrediction.stage):
        the kept rows of a level are never gathered into their own tensor — the next stage's CONCAT_IN reads them
        through the compaction's index list.  Same results as the per-module path (tests/test_gpu_program.py)."""
        if not (self.pass_occ or self.pass_feats):
            return None
        R = len(self.refinement)
        if getattr(geo, 'coords_i64', None) is None:
            geo.coords_i64 = F_.coords_to_i64(geo.coords)      # depends on the volume shape only (cached with geo)
        outputs = [[geo.coords_i64, occ_rows]]
        n_all = occ_rows.shape[0]
        tv = (lambda h: None) if teacher is None else (lambda h: teacher[h])
        if plans is not None:
            sel, cnt, locs = plans[0]
        elif capacity is not None:      # kept counts stay on the device; `cnt` is the level's capacity from here on
            sel, cnt, locs = F_.compact_capped(occ_rows.detach(), 2, n_all, geo.coords, 2 if runs[0] else 0, capacity, 0,
                                               tv(0))
        else:
            sel, cnt, locs = F_.compact_sigmoid_plan(occ_rows.detach(), 2, n_all, geo.coords, 2 if runs[0] else 0, tv(0))
        # channel order of model.py:330: [occ, sdf | features]
        prev = (occ_rows if self.pass_occ else None, feat_rows if self.pass_feats else None, sel, locs, cnt)
        for h in range(R):
            if not runs[h]:
                outputs.append([[], []])
                continue
            if cnt == 0:                         # nothing predicted occupied: the hierarchy ends here (model.py:211)
                outputs.append([[], []])
                continue
            ref = self.refinement[h]
            got = ref.stage(prev, skips[R - h] if self.use_skip_sparse else None)
            if got is None:
                return None if h == 0 else self._stage_fallback()
            y, out, coords_next = got
            outputs.append([F_.coords_to_i64(coords_next), out])
            if plans is not None:
                sel, cnt, locs = plans[h + 1]
            elif capacity is not None:
                sel, cnt, locs = F_.compact_capped(out.detach(), 2, out.shape[0], coords_next, ref.plan_depth, capacity,
                                                   h + 1, tv(h + 1))
            else:
                sel, cnt, locs = F_.compact_sigmoid_plan(out.detach(), 2, out.shape[0], coords_next, ref.plan_depth, tv(h + 1))
            # channel order of model.py:242: [features | occ, sdf]
            prev = (y if ref.pass_feats else None, out if ref.pass_occ else None, sel, locs, cnt)
        if not runs[R]:
            return [[], []], outputs
        if cnt == 0:
            return [[], []], outputs
        sdf = self.surfacepred.stage(prev, skips[0] if self.use_skip_sparse else None)
        if sdf is None:
            return self._stage_fallback()
        return [F_.coords_to_i64(prev[3]), sdf], outputs

    @staticmethod
    def _stage_fallback():
        raise RuntimeError('sgnn_amd: a generative stage could not be compiled into a native program after an earlier '
                           'one was; set sgnn_amd.model.STAGES = False to run the per-module path')
