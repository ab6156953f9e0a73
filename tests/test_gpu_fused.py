"""Fused convolution epilogues and strided rows (sgnn_conv_fwd_epi, sgnn_bn_fwd_ex / sgnn_bn_bwd_ex) against the
unfused kernels (themselves held to the oracle in test_gpu_ops.py): residual add in the store, BatchNorm statistics
out of the accumulator tile, rows that live in a column range of a wider buffer — serves the AddTable / BatchNormReLU /
JoinTable around every convolution of torch/model.py:33-42 and the FullyConvolutionalNet blocks (:180, :255)."""
import numpy as np
import pytest
import torch

from util import random_sites

pytestmark = pytest.mark.gpu


def _grid(batch, dim, seed):
    from sgnn_amd.scn.metadata import Grid, coords_from_locs
    locs = random_sites(batch, dim, 0.1, seed, surface=True)
    g = Grid(coords_from_locs(locs, torch.device('cuda')))
    return g, g.subm_table()


def _view(n, c, ld, col0, gen):
    """(contiguous copy, wide buffer, pointer of the view) of random rows living in columns [col0, col0+c) of ld."""
    wide = torch.randn(n, ld, device='cuda', generator=gen)
    return wide[:, col0:col0 + c].contiguous(), wide, wide.data_ptr() + 4 * col0


@pytest.mark.parametrize('batch,dim', [(2, 24), (5, 64)])   # 64-row workgroups / 256-row workgroups (>= 41 k rows)
@pytest.mark.parametrize('cin,cout', [(16, 16), (8, 12), (34, 16)])
def test_conv_epilogue_add_stats_strided(batch, dim, cin, cout):
    from sgnn_amd import _lib
    from sgnn_amd.scn import functions as F_
    g, tab = _grid(batch, dim, 7)
    n = g.n
    gen = torch.Generator(device='cuda').manual_seed(cin * 10 + cout)
    xc, xw, xp = _view(n, cin, cin + 8, 4, gen)
    ac, aw, ap = _view(n, cout, cout + 4, 4, gen)
    w = torch.randn(27, cin, cout, device='cuda', generator=gen) * 0.2
    ywide = torch.full((n, cout + 24), 7.0, device='cuda')
    nblk = _lib.query('sgnn_conv_stats_blocks', n)
    assert nblk == (-(-n // 16) if -(-n // 256) < 160 else -(-n // 256))    # 16-row workgroups on small levels
    part = torch.zeros(nblk, 2, cout, dtype=torch.float64, device='cuda')
    _lib.call('sgnn_conv_fwd_epi', xp, n, cin, cin + 8, w.data_ptr(), 27, tab.data_ptr(), g.ld, n, cout,
              ywide.data_ptr() + 4 * 8, cout + 24, 0, ap, cout + 4, 1, part.data_ptr(), None, 0, None, None, None, None, 0.0)
    want = F_.conv_fwd_raw(xc, cin, w, 27, tab, g.ld, n, cout) + ac
    got = ywide[:, 8:8 + cout]
    assert torch.equal(got, want)                                  # same accumulation order, same fp32 add
    assert (ywide[:, :8] == 7).all() and (ywide[:, 8 + cout:] == 7).all()   # neighbours of the column range untouched
    s = part.sum(0)
    w64 = want.double()
    assert torch.allclose(s[0], w64.sum(0), rtol=1e-6, atol=1e-6 * float(w64.abs().sum(0).max()))
    assert torch.allclose(s[1], (w64 * w64).sum(0), rtol=1e-6)
    # in-place accumulation: addend == y
    acc = ac.clone()
    _lib.call('sgnn_conv_fwd_epi', xp, n, cin, cin + 8, w.data_ptr(), 27, tab.data_ptr(), g.ld, n, cout,
              acc.data_ptr(), 0, 0, acc.data_ptr(), 0, 0, None, None, 0, None, None, None, None, 0.0)
    assert torch.equal(acc, want)
    # BatchNorm on those partials == BatchNorm with its own statistics pass (1e-6: summation order only)
    from sgnn_amd.scn.metadata import runtime
    rt = runtime(torch.device('cuda'))
    wsb = _lib.query('sgnn_bn_ws_bytes', n, cout)
    ws = rt.workspace(wsb)
    gamma = torch.rand(cout, device='cuda', generator=gen) + 0.5
    beta = torch.randn(cout, device='cuda', generator=gen) * 0.1
    outs = []
    for pre in (True, False):
        rm, rv = torch.zeros(cout, device='cuda'), torch.ones(cout, device='cuda')
        save = torch.empty(2, cout, device='cuda')
        y = torch.empty(n, cout + 2, device='cuda')
        _lib.call('sgnn_bn_fwd_ex', ywide.data_ptr() + 4 * 8, cout + 24, n, cout, gamma.data_ptr(), beta.data_ptr(),
                  rm.data_ptr(), rv.data_ptr(), 1e-4, 0.9, 1, 0.0, save[0].data_ptr(), save[1].data_ptr(), y.data_ptr(),
                  cout + 2, part.data_ptr() if pre else None, nblk if pre else 0, ws.data_ptr(), wsb)
        outs.append((y[:, :cout].clone(), rm, rv, save))
    for a, b in zip(outs[0], outs[1]):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    # and equals the contiguous entry point
    rm, rv = torch.zeros(cout, device='cuda'), torch.ones(cout, device='cuda')
    save = torch.empty(2, cout, device='cuda')
    y = torch.empty(n, cout, device='cuda')
    _lib.call('sgnn_bn_fwd', want.data_ptr(), n, cout, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(),
              1e-4, 0.9, 1, 0.0, save[0].data_ptr(), save[1].data_ptr(), y.data_ptr(), ws.data_ptr(), wsb)
    assert torch.allclose(y, outs[1][0], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('batch,dim', [(2, 24), (5, 64)])
@pytest.mark.parametrize('c', [16, 12])
def test_data_gradient_epilogue_feeds_batchnorm_backward(batch, dim, c):
    """dX convolution with stats = 2 + sgnn_bn_bwd_ex(pre_partial) == dX convolution, then sgnn_bn_bwd."""
    from sgnn_amd import _lib
    from sgnn_amd.scn import functions as F_
    from sgnn_amd.scn.metadata import runtime
    g, tab = _grid(batch, dim, 11)
    n, cout = g.n, 16
    gen = torch.Generator(device='cuda').manual_seed(c)
    dy_conv = torch.randn(n, cout, device='cuda', generator=gen)         # gradient of the convolution output
    w = torch.randn(27, c, cout, device='cuda', generator=gen) * 0.2     # layer weight (K, cin = c, cout)
    bn_x = torch.randn(n, c, device='cuda', generator=gen) * 2 + 0.3     # BatchNorm input
    gamma = torch.rand(c, device='cuda', generator=gen) + 0.5
    beta = torch.randn(c, device='cuda', generator=gen) * 0.3
    prior = torch.randn(n, c, device='cuda', generator=gen)              # gradient the BN output already carries
    rt = runtime(torch.device('cuda'))
    wsb = _lib.query('sgnn_bn_ws_bytes', n, c)
    ws = rt.workspace(wsb)
    rm, rv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
    save = torch.empty(2, c, device='cuda')
    bn_y = torch.empty(n, c, device='cuda')
    _lib.call('sgnn_bn_fwd', bn_x.data_ptr(), n, c, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 1e-4,
              0.9, 1, 0.0, save[0].data_ptr(), save[1].data_ptr(), bn_y.data_ptr(), ws.data_ptr(), wsb)
    flags = F_.CONV_TRANSPOSE_W | F_.CONV_FLIP_K
    # unfused
    d_bn_out = F_.conv_fwd_raw(dy_conv, cout, w, 27, tab, g.ld, n, c, flags) + prior
    dx0, dgb0 = torch.empty(n, c, device='cuda'), torch.empty(2, c, device='cuda')
    _lib.call('sgnn_bn_bwd', bn_x.data_ptr(), d_bn_out.data_ptr(), n, c, gamma.data_ptr(), beta.data_ptr(),
              save[0].data_ptr(), save[1].data_ptr(), 1, 0.0, dx0.data_ptr(), dgb0[0].data_ptr(), dgb0[1].data_ptr(),
              ws.data_ptr(), wsb)
    # fused: in-place accumulation onto `prior` + statistics in the epilogue
    nblk = _lib.query('sgnn_conv_stats_blocks', n)
    part = torch.zeros(nblk, 2, c, dtype=torch.float64, device='cuda')
    gbuf = prior.clone()
    _lib.call('sgnn_conv_fwd_epi', dy_conv.data_ptr(), n, cout, 0, w.data_ptr(), 27, tab.data_ptr(), g.ld, n, c,
              gbuf.data_ptr(), 0, flags, gbuf.data_ptr(), 0, 2, part.data_ptr(), bn_x.data_ptr(), 0, save[0].data_ptr(),
              save[1].data_ptr(), gamma.data_ptr(), beta.data_ptr(), 0.0)
    assert torch.equal(gbuf, d_bn_out)
    dz = torch.where(bn_y > 0, d_bn_out, torch.zeros_like(d_bn_out)).double()
    xh = ((bn_x - save[0]) * save[1]).double()
    s = part.sum(0)
    assert torch.allclose(s[0], dz.sum(0), rtol=1e-5, atol=1e-6 * float(dz.abs().sum(0).max()))
    assert torch.allclose(s[1], (dz * xh).sum(0), rtol=1e-5, atol=1e-6 * float((dz * xh).abs().sum(0).max()))
    dx1, dgb1 = torch.empty(n, c, device='cuda'), torch.empty(2, c, device='cuda')
    _lib.call('sgnn_bn_bwd_ex', bn_x.data_ptr(), 0, gbuf.data_ptr(), 0, n, c, gamma.data_ptr(), beta.data_ptr(),
              save[0].data_ptr(), save[1].data_ptr(), 1, 0.0, None, 0, dx1.data_ptr(), 0, dgb1[0].data_ptr(),
              dgb1[1].data_ptr(), part.data_ptr(), nblk, ws.data_ptr(), wsb)
    scale = max(1.0, float(dx0.abs().max()))
    assert (dx1 - dx0).abs().max().item() <= 1e-5 * scale
    assert torch.allclose(dgb1, dgb0, rtol=1e-5, atol=1e-5 * float(dgb0.abs().max()))


def test_epilogue_rejects_uncompiled_shapes():
    from sgnn_amd import _lib
    g, tab = _grid(1, 16, 3)
    x = torch.randn(g.n, 5, device='cuda')
    w = torch.randn(27, 5, 7, device='cuda')
    y = torch.empty(g.n, 7, device='cuda')
    with pytest.raises(_lib.SgnnError, match='compiled'):
        _lib.call('sgnn_conv_fwd_epi', x.data_ptr(), g.n, 5, 0, w.data_ptr(), 27, tab.data_ptr(), g.ld, g.n, 7,
                  y.data_ptr(), 0, 0, y.data_ptr(), 0, 0, None, None, 0, None, None, None, None, 0.0)


@pytest.mark.parametrize('cin,cout', [(16, 16), (26, 16)])
def test_one_round_tiling_gives_the_same_rows_and_statistics(cin, cout):
    """Large levels run as ONE round of workgroups, each taking J consecutive 256-row tiles (sgnn_tune.conv_one_round,
    k_conv_fwd / k_conv_fwd_u).  Against one tile per workgroup on a level large enough for J = 2: output rows bit-identical
    (same arithmetic per row), BatchNorm statistics equal to fp64 round-off (the partial rows are grouped differently), and
    the partial blocks past the live workgroups hold exact zeros."""
    from sgnn_amd import synth, _lib
    from sgnn_amd.scn.metadata import Grid, coords_from_locs
    DEV = torch.device('cuda')
    data = synth.make_batch(32, (64, 64, 64), cfg=2, occupancy=0.05)
    g = Grid(coords_from_locs(data['input'][0], DEV))
    tab, n = g.subm_table(), g.n
    assert n > 1280 * 256, 'the level must exceed one round of 256-row workgroups (5 resident per CU)'
    gen = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(n, cin, device=DEV, generator=gen)
    w = torch.randn(27, cin, cout, device=DEV, generator=gen) * 0.2
    nblk = _lib.query('sgnn_conv_stats_blocks', n)
    lib = _lib.load()

    def run(one_round):
        prev = _lib.tune('conv_one_round', one_round)
        try:
            y = torch.empty(n, cout, device=DEV)
            partial = torch.full((nblk, 2, cout), float('nan'), dtype=torch.float64, device=DEV)
            _lib.call('sgnn_conv_fwd_epi', x.data_ptr(), n, cin, cin, w.data_ptr(), 27, tab.data_ptr(), g.ld, n, cout,
                      y.data_ptr(), cout, 0, None, 0, 1, partial.data_ptr(), None, 0, None, None, None, None, 0.0)
            torch.cuda.synchronize()
        finally:
            _lib.tune('conv_one_round', prev)
        return y, partial
    y1, p1 = run(1)
    y0, p0 = run(0)
    assert torch.equal(y1, y0)
    assert torch.isfinite(p1).all() and torch.isfinite(p0).all()
    live1 = int((p1.abs().sum(dim=(1, 2)) > 0).sum().item())
    assert live1 < nblk and (p1[live1:] == 0).all(), 'workgroups past the live ones must write zero partials'
    s1, s0 = p1.sum(0), p0.sum(0)
    assert ((s1 - s0).abs() <= 1e-12 * s0.abs().clamp_min(1.0)).all()
    want = torch.stack([y0.double().sum(0), (y0.double() ** 2).sum(0)])
    assert ((s1 - want).abs() <= 1e-9 * want.abs().clamp_min(1.0)).all()


@pytest.mark.parametrize('c', [16, 8, 12])
def test_wide_epilogue_is_bit_identical_and_falls_back_on_odd_strides(c):
    """Round 5: the wide (quad-transposed, 16-byte) epilogue of the 256-row kernels (sgnn_tune.conv_wide_epi) against the
    element-wise one — forward with residual + statistics and data gradient with in-place accumulation + BatchNorm-backward
    statistics: rows AND fp64 statistics partials bit-identical (the arithmetic stays in the MFMA layout); with a row stride
    that is not a multiple of four floats the dispatcher must fall back to the element-wise form by itself."""
    from sgnn_amd import _lib
    from sgnn_amd.scn import functions as F_
    lib = _lib.load()
    g, tab = _grid(5, 64, 13)
    n = g.n
    assert -(-n // 256) >= 160                       # the 256-row kernels
    gen = torch.Generator(device='cuda').manual_seed(100 + c)
    x = torch.randn(n, c, device='cuda', generator=gen)
    w = torch.randn(27, c, c, device='cuda', generator=gen) * 0.2
    add = torch.randn(n, c, device='cuda', generator=gen)
    bn_x = torch.randn(n, c, device='cuda', generator=gen)
    mean, inv = torch.randn(c, device='cuda', generator=gen) * 0.1, torch.rand(c, device='cuda', generator=gen) + 0.5
    gamma, beta = torch.rand(c, device='cuda', generator=gen) + 0.5, torch.randn(c, device='cuda', generator=gen) * 0.3
    nblk = _lib.query('sgnn_conv_stats_blocks', n)
    flags = F_.CONV_TRANSPOSE_W | F_.CONV_FLIP_K

    def run(ldy, col0, stats, fl):
        y = torch.full((n, ldy), 3.0, device='cuda')
        y[:, col0:col0 + c] = add
        part = torch.zeros(nblk, 2, c, dtype=torch.float64, device='cuda')
        yp = y.data_ptr() + 4 * col0
        _lib.call('sgnn_conv_fwd_epi', x.data_ptr(), n, c, 0, w.data_ptr(), 27, tab.data_ptr(), g.ld, n, c, yp, ldy, fl,
                  yp, ldy, stats, part.data_ptr(), bn_x.data_ptr() if stats == 2 else None, 0,
                  mean.data_ptr() if stats == 2 else None, inv.data_ptr() if stats == 2 else None,
                  gamma.data_ptr() if stats == 2 else None, beta.data_ptr() if stats == 2 else None, 0.0)
        return y, part

    prev = _lib.tune('conv_wide_epi', 1)
    try:
        for ldy, col0 in ((c, 0), (c + 8, 4), (c + 1, 0), (c + 3, 2)):       # aligned, aligned view, odd strides (fallback)
            for stats, fl in ((1, 0), (2, flags)):
                _lib.tune('conv_wide_epi', 0)
                y0, p0 = run(ldy, col0, stats, fl)
                _lib.tune('conv_wide_epi', 1)
                y1, p1 = run(ldy, col0, stats, fl)
                assert torch.equal(y0, y1), (ldy, col0, stats)
                assert torch.equal(p0, p1), (ldy, col0, stats)
                assert (y1[:, :col0] == 3).all() and (y1[:, col0 + c:] == 3).all()
                if stats == 1:      # and it is the convolution + residual it claims to be
                    want = F_.conv_fwd_raw(x, c, w, 27, tab, g.ld, n, c) + add
                    assert torch.equal(y1[:, col0:col0 + c], want)
    finally:
        _lib.tune('conv_wide_epi', prev)
