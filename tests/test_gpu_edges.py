"""GPU edge cases of the sparse operators: coordinate limits, isolated sites, level sizes around the tile and
kernel-variant boundaries, a fully dense level."""
import numpy as np
import pytest
import torch

import scn_oracle as oscn
from util import copy_params

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _hip():
    import sgnn_amd.scn as scn
    return scn


def _conv_pair(locs, feats, size, cin, cout, seed=0):
    scn = _hip()
    torch.manual_seed(seed)
    co = oscn.SubmanifoldConvolution(3, cin, cout, 3, False)
    ch = scn.SubmanifoldConvolution(3, cin, cout, 3, False).cuda()
    copy_params(co, ch)
    xo = oscn.InputLayer(3, size, mode=0)([locs, feats])
    xh = scn.InputLayer(3, size, mode=0)([locs.cuda(), feats.cuda()])
    return co(xo).features, ch(xh).features.cpu()


def test_out_of_range_coordinates_raise():
    scn = _hip()
    from sgnn_amd._lib import SgnnError
    for bad in ([[-1, 2, 3, 0]], [[1, 70000, 3, 0]], [[1, 2, 3, -2]]):
        locs = torch.tensor(bad + [[2, 2, 2, 0]])
        with pytest.raises((SgnnError, ValueError, RuntimeError)):
            x = scn.InputLayer(3, [8] * 3, mode=0)([locs.cuda(), torch.zeros(2, 1).cuda()])
            x.grid().hash()
            scn.runtime().read_count()
    scn.runtime().state.zero_()                      # leave no pending status for the next test


def test_isolated_sites_use_only_the_centre_tap():
    g = torch.arange(0, 30, 3)
    zz, yy, xx = torch.meshgrid(g, g, g, indexing='ij')
    locs = torch.stack([zz.reshape(-1), yy.reshape(-1), xx.reshape(-1), torch.zeros(zz.numel(), dtype=torch.long)], 1)
    feats = torch.randn(locs.shape[0], 16)
    yo, yh = _conv_pair(locs, feats, [32] * 3, 16, 16)
    assert (yo - yh).abs().max().item() < TOL
    scn = _hip()
    xh = scn.InputLayer(3, [32] * 3, mode=0)([locs.cuda(), feats.cuda()])
    tab = xh.grid().subm_table().view(27, -1)[:, :xh.grid().n].cpu()
    assert (tab[13] == torch.arange(xh.grid().n)).all() and (tab[[k for k in range(27) if k != 13]] == -1).all()


def test_sites_at_the_coordinate_limit():
    base = torch.tensor([65533, 65533, 65533])
    offs = torch.tensor([[a, b, c] for a in range(3) for b in range(3) for c in range(3)])
    locs = torch.cat([base + offs, torch.zeros(27, 1, dtype=torch.long)], 1)       # up to 65535 on every axis
    feats = torch.randn(27, 8)
    yo, yh = _conv_pair(locs, feats, [65536] * 3, 8, 8)
    assert (yo - yh).abs().max().item() < TOL


@pytest.mark.parametrize('n', [1, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1023, 1025])
def test_level_sizes_around_tile_boundaries(n):
    rng = np.random.default_rng(n)
    cells = rng.permutation(14 ** 3)[:n]                  # a dense 14^3 cloud: plenty of neighbours at every size
    locs = torch.from_numpy(np.stack([cells // 196, (cells // 14) % 14, cells % 14, np.zeros(n, np.int64)], 1))
    feats = torch.randn(n, 16)
    yo, yh = _conv_pair(locs, feats, [16] * 3, 16, 16, seed=n)
    assert yh.shape == (n, 16) and (yo - yh).abs().max().item() < TOL * max(1.0, yo.abs().max().item())


def test_both_kernel_variants_around_the_row_threshold():
    """40 960 rows is where the 64-row and the 256-row workgroup variants swap (CONV_SMALL_GRID * 256)."""
    for n in (40700, 41200):
        rng = np.random.default_rng(n)
        cells = rng.permutation(40 ** 3)[:n]
        locs = torch.from_numpy(np.stack([cells // 1600, (cells // 40) % 40, cells % 40, np.zeros(n, np.int64)], 1))
        feats = torch.randn(n, 16)
        yo, yh = _conv_pair(locs, feats, [40] * 3, 16, 16, seed=1)
        assert (yo - yh).abs().max().item() < TOL * max(1.0, yo.abs().max().item())


def test_fully_dense_level_gradients():
    scn = _hip()
    d = 10
    zz, yy, xx = torch.meshgrid(torch.arange(d), torch.arange(d), torch.arange(d), indexing='ij')
    locs = torch.stack([zz.reshape(-1), yy.reshape(-1), xx.reshape(-1), torch.zeros(d ** 3, dtype=torch.long)], 1)
    feats = torch.randn(d ** 3, 12)
    torch.manual_seed(3)
    co = oscn.SubmanifoldConvolution(3, 12, 12, 3, False)
    ch = scn.SubmanifoldConvolution(3, 12, 12, 3, False).cuda()
    copy_params(co, ch)
    fo, fh = feats.clone().requires_grad_(), feats.clone().cuda().requires_grad_()
    yo = co(oscn.InputLayer(3, [d] * 3, mode=0)([locs, fo])).features
    yh = ch(scn.InputLayer(3, [d] * 3, mode=0)([locs.cuda(), fh])).features
    w = torch.randn_like(yo)
    (yo * w).sum().backward()
    (yh * w.cuda()).sum().backward()
    assert (yo - yh.cpu()).abs().max().item() < TOL * max(1.0, yo.abs().max().item())
    assert (fo.grad - fh.grad.cpu()).abs().max().item() < TOL * max(1.0, fo.grad.abs().max().item())
    assert (co.weight.grad - ch.weight.grad.cpu()).abs().max().item() < 1e-3 * max(1.0, co.weight.grad.abs().max().item())
    # equals the dense convolution too (zero padding), the property the oracle itself is pinned with
    dense = torch.nn.functional.conv3d(feats.view(1, d, d, d, 12).permute(0, 4, 1, 2, 3),
                                       co.weight.detach().view(3, 3, 3, 12, 12).permute(4, 3, 0, 1, 2), padding=1)
    assert (dense[0].permute(1, 2, 3, 0).reshape(-1, 12) - yh.detach().cpu()).abs().max().item() < 1e-3
