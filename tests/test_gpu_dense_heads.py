"""The dense 8^3 bottleneck and the per-site heads run on the HIP kernels too (sgnn_amd/model.py):
checked here against torch's own dense ops evaluated in float64 on the CPU."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize('parity', [True, False])
@pytest.mark.parametrize('cin,cout,dims', [(16, 24, (8, 8, 8)), (24, 32, (4, 4, 4)), (16, 24, (4, 8, 4)), (28, 56, (8, 8, 8)),
                                            (32, 64, (4, 4, 4))])
def test_dense_k4s2_conv_and_transpose_match_torch(cin, cout, dims, parity, monkeypatch):
    """torch/model.py:89-136: nn.Conv3d / nn.ConvTranspose3d (k4, s2, p1) as rulebook walks — the fine side of each layer as 8
    parity groups of 8 taps on the coarse rulebook (parity, the default) or as a 64-offset walk over the fine rows.  (28, 56)
    and (32, 64) are the shapes of the model's two ConvTranspose3d layers, read right to left."""
    from sgnn_amd import model as M
    monkeypatch.setattr(M, 'DENSE_PARITY', parity)
    torch.manual_seed(cin + cout)
    B = 3
    x = torch.randn(B, cin, *dims)
    cd = nn.Conv3d(cin, cout, 4, 2, 1, bias=False).double()
    ctd = nn.ConvTranspose3d(cout, cin, 4, 2, 1, bias=False).double()
    xd = x.double().requires_grad_(True)
    y_ref = cd(xd)
    z_ref = ctd(y_ref)
    g = torch.randn_like(z_ref)
    z_ref.backward(g)
    geo = M.dense_geometry(B, dims, torch.device('cuda'))
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, cin).cuda().requires_grad_(True)
    ch = nn.Conv3d(cin, cout, 4, 2, 1, bias=False).cuda()
    cth = nn.ConvTranspose3d(cout, cin, 4, 2, 1, bias=False).cuda()
    ch.weight.data.copy_(cd.weight.data.float())
    cth.weight.data.copy_(ctd.weight.data.float())
    y = M._dense_conv(rows, ch, geo.level(0), down=True)
    z = M._dense_conv(y, cth, geo.level(0), down=False)
    y_want = y_ref.detach().permute(0, 2, 3, 4, 1).reshape(-1, cout)
    z_want = z_ref.detach().permute(0, 2, 3, 4, 1).reshape(-1, cin)
    assert (y.detach().cpu().double() - y_want).abs().max().item() < TOL
    assert (z.detach().cpu().double() - z_want).abs().max().item() < TOL
    z.backward(g.permute(0, 2, 3, 4, 1).reshape(-1, cin).float().cuda())
    gx = xd.grad.permute(0, 2, 3, 4, 1).reshape(-1, cin)
    assert (rows.grad.cpu().double() - gx).abs().max().item() < TOL * max(1.0, gx.abs().max().item())
    for a, b in ((ch.weight.grad, cd.weight.grad), (cth.weight.grad, ctd.weight.grad)):
        assert (a.cpu().double() - b).abs().max().item() < TOL * max(1.0, b.abs().max().item())


@pytest.mark.parametrize('cin,cout,bias', [(16, 2, True), (48, 1, True), (16, 2, False)])
def test_row_linear_heads(cin, cout, bias):
    from sgnn_amd.scn import functions as F_
    torch.manual_seed(cin)
    n = 5000
    x = torch.randn(n, cin, dtype=torch.float64, requires_grad=True)
    w = torch.randn(cout, cin, dtype=torch.float64, requires_grad=True)
    b = torch.randn(cout, dtype=torch.float64, requires_grad=True) if bias else None
    y = torch.nn.functional.linear(x, w, b)
    g = torch.randn_like(y)
    y.backward(g)
    xh = x.detach().float().cuda().requires_grad_(True)
    wh = w.detach().float().cuda().requires_grad_(True)
    bh = b.detach().float().cuda().requires_grad_(True) if bias else None
    yh = F_.RowLinear.apply(xh, wh, bh)
    yh.backward(g.float().cuda())
    assert (yh.detach().cpu().double() - y.detach()).abs().max().item() < TOL
    assert (xh.grad.cpu().double() - x.grad).abs().max().item() < TOL
    assert (wh.grad.cpu().double() - w.grad).abs().max().item() < 1e-3 * max(1.0, w.grad.abs().max().item())
    if bias:
        assert (bh.grad.cpu().double() - b.grad).abs().max().item() < 1e-3 * max(1.0, b.grad.abs().max().item())


def test_batchnorm3d_on_rows_matches_torch():
    from sgnn_amd import model as M
    torch.manual_seed(9)
    x = torch.randn(4, 24, 4, 4, 4) * 2 + 1
    bn_ref = nn.BatchNorm3d(24).double()
    bn_hip = nn.BatchNorm3d(24).cuda()
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5)
        bn_ref.bias.uniform_(-0.3, 0.3)
    bn_hip.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in bn_ref.state_dict().items()})
    xd = x.double().requires_grad_(True)
    y_ref = torch.relu(bn_ref(xd))
    rows = x.permute(0, 2, 3, 4, 1).reshape(-1, 24).cuda().requires_grad_(True)
    y = M._bn3d_relu(bn_hip, rows)
    want = y_ref.detach().permute(0, 2, 3, 4, 1).reshape(-1, 24)
    assert (y.detach().cpu().double() - want).abs().max().item() < TOL
    assert (bn_hip.running_var.cpu().double() - bn_ref.running_var).abs().max().item() < 1e-5
    assert int(bn_hip.num_batches_tracked) == 1
    g = torch.randn_like(y_ref)
    y_ref.backward(g)
    y.backward(g.permute(0, 2, 3, 4, 1).reshape(-1, 24).float().cuda())
    assert (rows.grad.cpu().double() - xd.grad.permute(0, 2, 3, 4, 1).reshape(-1, 24)).abs().max().item() < TOL


@pytest.mark.parametrize('cin,cout', [(48, 16), (12, 8)])
def test_generative_upsampling_conv_equals_expand_then_subm(cin, cout):
    """torch/model.py:192-207 + n0/n1 (:220-222): expand every site into 8 children carrying its features,
    then SubmanifoldConvolution on the children — versus the parent-rulebook formulation (ExpandConv)."""
    import numpy as np
    import scn_oracle as oscn
    import model_oracle as mo
    import sgnn_amd.scn as scn
    from sgnn_amd.scn import functions as F_
    from util import random_sites
    torch.manual_seed(cin)
    locs = random_sites(2, 12, 0.2, 4, surface=True)
    f = torch.randn(locs.shape[0], cin)
    conv_o = oscn.SubmanifoldConvolution(3, cin, cout, 3, False)
    fo = f.clone().double().requires_grad_(True)
    co = conv_o.double()
    locs_c, feats_c = mo.expand_children(locs, fo)
    yo = co(oscn.InputLayer(3, [24] * 3, mode=0)([locs_c, feats_c])).features
    g = torch.randn_like(yo)
    yo.backward(g)
    fh = f.clone().cuda().requires_grad_(True)
    wh = conv_o.weight.detach().float().cuda().requires_grad_(True)
    grid = scn.InputLayer(3, [12] * 3, mode=0)([locs.cuda(), fh]).grid()
    yh = F_.expand_conv(fh, wh, grid)
    assert yh.shape == yo.shape
    assert (yh.detach().cpu().double() - yo.detach()).abs().max().item() < TOL
    yh.backward(g.float().cuda())
    assert (fh.grad.cpu().double() - fo.grad).abs().max().item() < TOL * max(1.0, fo.grad.abs().max().item())
    gw = co.weight.grad
    assert (wh.grad.cpu().double() - gw).abs().max().item() < TOL * max(1.0, gw.abs().max().item())
