"""First-principles pin of the oracle's scn arithmetic: every sparse op equals a dense torch op sampled
at the active sites.  The reference ships no tests for this path, so this is what fixes
offset order, cross-correlation (no flip), stride-2 offsets and BN conventions.  CPU only, fp64."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import scn_oracle as scn
from util import random_sites


def dense_of(locs, feats, B, S):
    d = torch.zeros(B, feats.shape[1], S, S, S, dtype=feats.dtype)
    d[locs[:, 3], :, locs[:, 0], locs[:, 1], locs[:, 2]] = feats
    return d


def at(d, l):
    return d[l[:, 3], :, l[:, 0], l[:, 1], l[:, 2]]


@pytest.mark.parametrize('seed,surface', [(0, False), (1, True)])
def test_subm_equals_masked_dense_conv(seed, surface):
    B, S, cin, cout = 2, 12, 3, 5
    locs = random_sites(B, S, 0.3, seed, surface)
    perm = torch.randperm(locs.shape[0], generator=torch.Generator().manual_seed(seed))
    locs = locs[perm]  # arbitrary row order must not matter
    feats = torch.randn(locs.shape[0], cin, dtype=torch.float64, requires_grad=True)
    conv = scn.SubmanifoldConvolution(3, cin, cout, 3, False).double()
    y = conv(scn.InputLayer(3, [S] * 3, mode=0)([locs, feats])).features
    wd = conv.weight.view(3, 3, 3, cin, cout).permute(4, 3, 0, 1, 2)
    fd = feats.detach().clone().requires_grad_(True)
    yd = at(F.conv3d(dense_of(locs, fd, B, S), wd, padding=1), locs)
    assert (y - yd).abs().max().item() < 1e-12
    g = torch.randn_like(y)
    y.backward(g)
    yd.backward(g)
    assert (feats.grad - fd.grad).abs().max().item() < 1e-12


def test_strided_conv_unpool_equal_dense():
    B, S, c = 2, 8, 4
    locs = random_sites(B, S, 0.3, 3)
    feats = torch.randn(locs.shape[0], c, dtype=torch.float64)
    x = scn.InputLayer(3, [S] * 3, mode=0)([locs, feats])
    conv = scn.Convolution(3, c, 6, 2, 2, False).double()
    z = conv(x)
    zl = z.metadata.getSpatialLocations(z.spatial_size)
    dense = dense_of(locs, feats, B, S)
    zd = F.conv3d(dense, conv.weight.view(2, 2, 2, c, 6).permute(4, 3, 0, 1, 2), stride=2)
    assert (at(zd, zl) - z.features).abs().max().item() < 1e-12
    occ = F.max_pool3d((dense.abs().sum(1, keepdim=True) > 0).double(), 2)
    assert zl.shape[0] == int(occ.sum().item())                       # active set = unique(floor(p/2))
    assert len({tuple(r) for r in zl.tolist()}) == zl.shape[0]
    # first-touch order: parent of fine row 0 is coarse row 0, and ranks only grow on first sight
    parent, off = z.metadata.down2(x.spatial_size, z.spatial_size)
    seen = -1
    for p in parent:
        assert p <= seen + 1
        seen = max(seen, p)
    assert np.array_equal(off, (locs[:, 0] % 2 * 4 + locs[:, 1] % 2 * 2 + locs[:, 2] % 2).numpy())
    u = scn.UnPooling(3, 2, 2)(z)
    ud = F.interpolate(dense_of(zl, z.features.detach(), B, S // 2), scale_factor=2, mode='nearest')
    assert (at(ud, locs) - u.features).abs().max().item() < 1e-12


def test_batchnorm_relu_equals_torch():
    torch.manual_seed(0)
    x = torch.randn(500, 7, dtype=torch.float64) * 3 + 1
    m = scn.BatchNormReLU(7).double()
    rm, rv = m.running_mean.clone(), m.running_var.clone()
    y = m(scn.SparseConvNetTensor(x, None, None)).features
    yt = F.relu(F.batch_norm(x, rm, rv, m.weight, m.bias, True, 0.1, 1e-4))  # torch momentum = 1 - scn momentum
    assert (y - yt).abs().max().item() < 1e-12
    assert (m.running_mean - rm).abs().max().item() < 1e-12 and (m.running_var - rv).abs().max().item() < 1e-12
    m.eval()
    y2 = m(scn.SparseConvNetTensor(x, None, None)).features
    assert (y2 - F.relu(F.batch_norm(x, rm, rv, m.weight, m.bias, False, 0.1, 1e-4))).abs().max().item() < 1e-12


def test_sparse_to_dense_and_duplicates():
    locs = torch.tensor([[0, 1, 2, 0], [3, 3, 3, 1], [1, 0, 0, 1]])
    f = torch.arange(6.).view(3, 2)
    d = scn.SparseToDense(3, 2)(scn.InputLayer(3, [4] * 3, mode=0)([locs, f]))
    assert d.shape == (2, 2, 4, 4, 4) and d[1, 1, 3, 3, 3] == 3 and d.sum() == f.sum()
    with pytest.raises(ValueError):
        scn.InputLayer(3, [4] * 3, mode=0)([torch.cat([locs, locs[:1]]), torch.zeros(4, 2)])
