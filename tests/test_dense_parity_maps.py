"""Index structure of the dense k4/s2 layers by parity groups (sgnn_amd/model.py: K4S2_TAPS / K4S2_NBR, _DenseGeometry.level,
DenseConv's slot order; functions.DenseK4S2 runs it on the HIP kernels) — restated here with plain torch indexing on the CPU
and held to torch's own nn.Conv3d / nn.ConvTranspose3d (k4, s2, p1; torch/model.py:89-136) in float64.  No GPU needed: the
kernels the GPU path calls (sgnn_conv_fwd_ex / sgnn_conv_bwd_weight_ex with groups = 8) are the up-sampling convolution's,
tested in tests/test_gpu_ops.py / test_gpu_dense_heads.py."""
import pytest
import torch
import torch.nn as nn


def _geometry(batch, dims):
    from sgnn_amd import model as M
    g = object.__new__(M._DenseGeometry)          # the constructor builds device coordinates: not needed for the tables
    g.batch, g.dims, g.device, g._levels = batch, dims, torch.device('cpu'), {}
    return g.level(0)


def _rows(t):      # (B, C, z, y, x) -> channel-last rows in batch-major raster order
    return t.permute(0, 2, 3, 4, 1).reshape(-1, t.shape[1])


def _walk(x, w, table, ld, n_out, transpose=False):
    """y[j] = sum_q W[q]^T x[table[q][j]] (W[q] applied transposed for a data gradient) — the rulebook walk."""
    t = table.view(-1, ld)[:, :n_out].long()
    y = torch.zeros(n_out, w.shape[1] if transpose else w.shape[2], dtype=x.dtype)
    for q in range(t.shape[0]):
        ok = t[q] >= 0
        rows = x[t[q][ok]]
        y[ok] += rows @ (w[q].t() if transpose else w[q])
    return y


def _walk8(x, w, lvl, transpose=False):
    """Child-order result of the 8-parity-group walk on the coarse rulebook: y8[8c+g] = sum_i W[g*8+i]^T x[nbr27[S[g*8+i]][c]]."""
    nbr = lvl.nbr27.view(27, lvl.ld_c)[:, :lvl.n_c].long()
    y8 = torch.zeros(lvl.n_c, 8, w.shape[1] if transpose else w.shape[2], dtype=x.dtype)
    for q in range(64):
        g = q // 8
        src = nbr[int(lvl.slots[q])]
        ok = src >= 0
        y8[ok, g] += x[src[ok]] @ (w[q].t() if transpose else w[q])
    return y8.view(8 * lvl.n_c, -1)


def test_slot_order_is_a_permutation_and_matches_the_expand_maps():
    from sgnn_amd import model as M
    assert sorted(M.K4S2_TAPS) == list(range(64))
    assert [M.K4S2_TAPS[M.K4S2_SLOT[t]] for t in range(64)] == list(range(64))
    # the coarse neighbour of slot (g, i) is the up-sampling convolution's: o = i - 1 + j per axis
    S = []
    for g in range(8):
        for i_ in range(8):
            o = [((i_ >> s) & 1) - 1 + ((g >> s) & 1) for s in (2, 1, 0)]
            S.append((o[0] + 1) * 9 + (o[1] + 1) * 3 + (o[2] + 1))
    assert M.K4S2_NBR == S


@pytest.mark.parametrize('cin,cout,dims', [(3, 5, (4, 4, 4)), (4, 2, (2, 4, 6))])
def test_parity_group_walks_equal_torch_conv_and_transpose(cin, cout, dims):
    from sgnn_amd import model as M
    torch.manual_seed(cin * 10 + cout)
    B = 2
    lvl = _geometry(B, dims)
    cd = [d // 2 for d in dims]
    assert lvl.n_f == B * dims[0] * dims[1] * dims[2] and lvl.n_c == lvl.n_f // 8
    assert torch.equal(lvl.raster_of_child[lvl.child_of_raster.long()].long(), torch.arange(lvl.n_f))

    # ---- Conv3d(cin -> cout): forward on the coarse side, data gradient by parity groups
    conv = nn.Conv3d(cin, cout, 4, 2, 1, bias=False).double()
    d = M.DenseConv(cin, cout, 4, 2, 1, False)
    d.load_state_dict(conv.state_dict())
    w = d.weight.detach().double()
    assert torch.equal(d.state_dict()['weight'].double(), conv.weight.detach())
    x = torch.randn(B, cin, *dims, dtype=torch.float64, requires_grad=True)
    y_ref = conv(x)
    gy = torch.randn_like(y_ref)
    y_ref.backward(gy)
    xr, gyr = _rows(x.detach()), _rows(gy)
    y = _walk(xr, w, lvl.tdown, lvl.ld_c, lvl.n_c)
    assert torch.allclose(y, _rows(y_ref.detach()), atol=1e-12)
    dx_tup = _walk(gyr, w, lvl.tup, lvl.ld_f, lvl.n_f, transpose=True)                    # the 64-tap walk over the fine rows
    dx_par = _walk8(gyr, w, lvl, transpose=True)[lvl.child_of_raster.long()]              # 8 groups x 8 taps, child -> raster
    assert torch.allclose(dx_tup, _rows(x.grad), atol=1e-12)
    assert torch.allclose(dx_par, _rows(x.grad), atol=1e-12)
    # weight gradient in slot order == to_native of torch's
    t = lvl.tdown.view(64, lvl.ld_c)[:, :lvl.n_c].long()
    dw = torch.stack([(xr[t[q].clamp(min=0)] * (t[q] >= 0).unsqueeze(1)).t() @ gyr for q in range(64)])
    assert torch.allclose(dw, d.to_native(conv.weight.grad), atol=1e-12)

    # ---- ConvTranspose3d(cout -> cin): forward and weight gradient by parity groups, data gradient on the coarse side
    ct = nn.ConvTranspose3d(cout, cin, 4, 2, 1, bias=False).double()
    dt = M.DenseConv(cout, cin, 4, 2, 1, True)
    dt.load_state_dict(ct.state_dict())
    wt = dt.weight.detach().double()
    assert torch.equal(dt.to_torch(dt.weight.detach()).double(), ct.weight.detach())
    xc = torch.randn(B, cout, *cd, dtype=torch.float64, requires_grad=True)
    z_ref = ct(xc)
    gz = torch.randn_like(z_ref)
    z_ref.backward(gz)
    xcr, gzr = _rows(xc.detach()), _rows(gz)
    z_tup = _walk(xcr, wt, lvl.tup, lvl.ld_f, lvl.n_f)
    z_par = _walk8(xcr, wt, lvl)[lvl.child_of_raster.long()]
    assert torch.allclose(z_tup, _rows(z_ref.detach()), atol=1e-12)
    assert torch.allclose(z_par, _rows(z_ref.detach()), atol=1e-12)
    dxc = _walk(gzr, wt, lvl.tdown, lvl.ld_c, lvl.n_c, transpose=True)
    assert torch.allclose(dxc, _rows(xc.grad), atol=1e-12)
    # dW[g*8+i] = sum_c x[nbr27[S][c]]^T dz8[8c+g], dz8 = the gradient rows in child order
    gz8 = gzr[lvl.raster_of_child.long()].view(lvl.n_c, 8, cin)
    nbr = lvl.nbr27.view(27, lvl.ld_c)[:, :lvl.n_c].long()
    dwt = []
    for q in range(64):
        src = nbr[int(lvl.slots[q])]
        dwt.append((xcr[src.clamp(min=0)] * (src >= 0).unsqueeze(1)).t() @ gz8[:, q // 8])
    assert torch.allclose(torch.stack(dwt), dt.to_native(ct.weight.grad), atol=1e-12)
