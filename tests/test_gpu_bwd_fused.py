"""Fused backward of a 3x3x3 SubmanifoldConvolution (csrc/conv_bwd_fused.hip; reference torch/model.py:38,40,180,255 under
train.py:262): data gradient AND weight gradient from one gather of dy.  Held to the CPU oracle (1e-4, north_star), to the
two-kernel path (dX rows bit-identical; dW to fp32 summation order), to NaN-poisoned surroundings, to capacity mode, and
through the program executor with the switch on and off."""
import numpy as np
import pytest
import torch

import scn_oracle as oscn
from util import random_sites, copy_params

pytestmark = pytest.mark.gpu
TOL = 1e-4
C, K = 16, 27


@pytest.fixture
def small_levels():
    """the fused kernel serves levels >= 40 960 rows by default; let it run on test-sized levels"""
    from sgnn_amd import _lib
    prev = _lib.tune('conv_bwd_fused_rows', 256)
    prev_on = _lib.tune('conv_bwd_fused', 1)
    # ... and hold it against the 256-row kernel it shares its walk with (k_conv_small, which test-sized levels would get,
    # sums the offsets in another order)
    prev_small = _lib.tune('conv_small_rows', 0)
    yield
    _lib.tune('conv_small_rows', prev_small)
    _lib.tune('conv_bwd_fused', prev_on)
    _lib.tune('conv_bwd_fused_rows', prev)


def _level(batch, dim, occ, seed, surface=True):
    from sgnn_amd.scn.metadata import Grid, coords_from_locs
    locs = random_sites(batch, dim, occ, seed, surface=surface)
    g = Grid(coords_from_locs(locs, torch.device('cuda')))
    return locs, g, g.subm_table()


def _fused(dy, x, w, tab, g, n, addend=None, stats=None, n_dev=None, dx=None):
    """-> dx, dw (+ statistics partials).  stats: (bn_x, mean, invstd, gamma, beta)."""
    from sgnn_amd import _lib
    dx = torch.empty(n, C, device='cuda') if dx is None else dx
    dw = torch.full((K, C, C), float('nan'), device='cuda')
    wsb = _lib.query('sgnn_conv_bwd_fused_ws_bytes', n, C, C)
    ws = torch.empty(wsb // 4 + 64, device='cuda')
    part = None
    if stats is not None:
        part = torch.full((_lib.query('sgnn_conv_stats_blocks', n), 2, C), float('nan'), dtype=torch.float64, device='cuda')
    p = lambda t: None if t is None else t.data_ptr()
    _lib.call('sgnn_conv_bwd_fused', dy.data_ptr(), n, C, dy.stride(0), x.data_ptr(), C, x.stride(0), w.data_ptr(),
              tab.data_ptr(), g.ld, dx.data_ptr(), dx.stride(0), p(addend), 0 if addend is None else addend.stride(0),
              2 if stats is not None else 0, p(part), p(stats[0]) if stats else None, stats[0].stride(0) if stats else 0,
              p(stats[1]) if stats else None, p(stats[2]) if stats else None, p(stats[3]) if stats else None,
              p(stats[4]) if stats else None, 0.0, dw.data_ptr(), ws.data_ptr(), ws.numel() * 4, p(n_dev))
    return dx, dw, part


def _two_kernels(dy, x, w, tab, g, n, addend=None, stats=None):
    from sgnn_amd import _lib
    from sgnn_amd.scn import functions as F_
    FL = F_.CONV_TRANSPOSE_W | F_.CONV_FLIP_K
    dx = torch.empty(n, C, device='cuda')
    part = None
    if stats is not None:
        part = torch.zeros(_lib.query('sgnn_conv_stats_blocks', n), 2, C, dtype=torch.float64, device='cuda')
    p = lambda t: None if t is None else t.data_ptr()
    _lib.call('sgnn_conv_fwd_epi', dy.data_ptr(), n, C, dy.stride(0), w.data_ptr(), K, tab.data_ptr(), g.ld, n, C, dx.data_ptr(),
              0, FL, p(addend), 0 if addend is None else addend.stride(0), 2 if stats is not None else 0, p(part),
              p(stats[0]) if stats else None, 0, p(stats[1]) if stats else None, p(stats[2]) if stats else None,
              p(stats[3]) if stats else None, p(stats[4]) if stats else None, 0.0)
    dw = F_.conv_dw_raw(x, C, dy, C, tab, g.ld, K, n)
    return dx, dw, part


def test_fused_backward_matches_the_oracle(small_levels):
    """the reference algorithm (per-offset gather -> mm -> index_add and its autograd) on the same sites and weights"""
    import sgnn_amd.scn as scn
    torch.manual_seed(7)
    locs = random_sites(2, 20, 0.2, 31, surface=True)
    feats = torch.randn(locs.shape[0], C)
    fo = feats.clone().requires_grad_(True)
    mo = oscn.SubmanifoldConvolution(3, C, C, 3, False)
    yo = mo(oscn.InputLayer(3, [20] * 3, mode=0)([locs, fo])).features
    go = torch.randn_like(yo)
    yo.backward(go)
    xh = scn.InputLayer(3, [20] * 3, mode=0)([locs.cuda(), feats.cuda()])
    g = xh.grid()
    from sgnn_amd import _lib
    assert _lib.query('sgnn_conv_bwd_fused_supported', g.n, C, C, K) == 1
    dx, dw, _ = _fused(go.cuda(), feats.cuda(), mo.weight.detach().cuda().contiguous(), g.subm_table(), g, g.n)
    assert (fo.grad - dx.cpu()).abs().max().item() < TOL
    scale = max(1.0, mo.weight.grad.abs().max().item())
    assert (mo.weight.grad - dw.cpu()).abs().max().item() < TOL * scale


@pytest.mark.parametrize('batch,dim,occ', [(2, 24, 0.1), (8, 48, 0.08)])
def test_fused_backward_equals_the_two_kernel_path(small_levels, batch, dim, occ):
    """dX rows bit-identical to the data-gradient kernel (same walk, same epilogue), statistics partials to fp64 round-off,
    dW within fp32 summation order of the weight-gradient kernel and of the exact (fp64) value"""
    locs, g, tab = _level(batch, dim, occ, 3)
    n = g.n
    gen = torch.Generator(device='cuda').manual_seed(n)
    dy = torch.randn(n, C, device='cuda', generator=gen)
    x = torch.relu(torch.randn(n, C, device='cuda', generator=gen))
    w = torch.randn(K, C, C, device='cuda', generator=gen) * 0.2
    bn_x = torch.randn(n, C, device='cuda', generator=gen)
    stats = (bn_x, torch.randn(C, device='cuda', generator=gen) * 0.1, torch.rand(C, device='cuda', generator=gen) + 0.5,
             torch.rand(C, device='cuda', generator=gen) + 0.5, torch.randn(C, device='cuda', generator=gen) * 0.1)
    acc = torch.randn(n, C, device='cuda', generator=gen)
    for addend, st in ((None, None), (acc, stats)):
        want_dx, want_dw, want_p = _two_kernels(dy, x, w, tab, g, n, addend, st)
        got_dx, got_dw, got_p = _fused(dy, x, w, tab, g, n, addend, st)
        assert torch.equal(got_dx, want_dx)
        t = tab.view(K, g.ld)[:, :n].long()
        ref = torch.zeros(K, C, C, dtype=torch.float64, device='cuda')
        for k in range(K):
            ok = t[k] >= 0
            ref[k] = x.double()[t[k][ok]].t() @ dy.double()[ok]
        scale = float(ref.abs().max())
        assert torch.isfinite(got_dw).all()
        assert float((got_dw.double() - ref).abs().max()) < 2e-6 * scale
        assert float((got_dw - want_dw).abs().max()) < 4e-6 * scale
        if st is not None:
            a, b = want_p.sum(0), got_p.sum(0)
            assert torch.isfinite(got_p).all()
            assert float(((a - b).abs() / a.abs().clamp_min(1e-30)).max()) < 1e-9


def test_fused_backward_in_place_addend_and_strided_rows(small_levels):
    """gradient accumulation in place (addend aliases dx) and rows inside wider buffers (JoinTable column ranges)"""
    locs, g, tab = _level(3, 24, 0.12, 9)
    n = g.n
    gen = torch.Generator(device='cuda').manual_seed(5)
    wide_dy = torch.randn(n, 48, device='cuda', generator=gen)
    wide_x = torch.randn(n, 32, device='cuda', generator=gen)
    dy, x = wide_dy[:, 16:32], wide_x[:, 16:32]
    w = torch.randn(K, C, C, device='cuda', generator=gen) * 0.2
    acc = torch.randn(n, C, device='cuda', generator=gen)
    want_dx, want_dw, _ = _two_kernels(dy.contiguous(), x.contiguous(), w, tab, g, n, acc.clone(), None)
    buf = acc.clone()
    got_dx, got_dw, _ = _fused(dy, x, w, tab, g, n, buf, None, dx=buf)
    assert torch.equal(got_dx, want_dx)
    assert float((got_dw - want_dw).abs().max()) < 4e-6 * float(want_dw.abs().max())


def test_fused_backward_never_reads_outside_its_slabs(small_levels):
    """the NaN-slab test of the convolution kernels, for the fused kernel — dy, x, the addend and the
    BatchNorm input in the middle of NaN-filled allocations, a row count that is no multiple of the 256-row tiles"""
    locs, g, tab = _level(3, 24, 0.1, 5)
    n = g.n
    assert n % 256 != 0
    gen = torch.Generator(device='cuda').manual_seed(11)
    dy = torch.randn(n, C, device='cuda', generator=gen)
    x = torch.randn(n, C, device='cuda', generator=gen)
    w = torch.randn(K, C, C, device='cuda', generator=gen) * 0.2
    bn_x = torch.randn(n, C, device='cuda', generator=gen)
    acc = torch.randn(n, C, device='cuda', generator=gen)
    cst = (torch.zeros(C, device='cuda'), torch.ones(C, device='cuda'), torch.ones(C, device='cuda'), torch.zeros(C, device='cuda'))
    pad = 4096

    def poisoned(t):
        big = torch.full((t.numel() + 2 * pad,), float('nan'), device='cuda')
        big[pad:pad + t.numel()] = t.reshape(-1)
        return big, big[pad:pad + t.numel()].view_as(t)
    want = _fused(dy, x, w, tab, g, n, acc, (bn_x,) + cst)
    (bdy, pdy), (bx, px), (bb, pb), (ba, pa) = poisoned(dy), poisoned(x), poisoned(bn_x), poisoned(acc)
    got = _fused(pdy, px, w, tab, g, n, pa, (pb,) + cst)
    for a, b in zip(got, want):
        assert torch.isfinite(a).all(), 'the fused backward kernel read outside its slabs'
        assert torch.equal(a, b)
    assert torch.isnan(bdy[:pad]).all() and torch.isnan(bx[-pad:]).all()


def test_fused_backward_capacity_mode(small_levels):
    """row count on the device: a capacity-sized launch over the live prefix returns what the exact launch returns"""
    locs, g, tab = _level(4, 32, 0.1, 13)
    n = g.n
    live = n - 777
    gen = torch.Generator(device='cuda').manual_seed(2)
    dy = torch.randn(n, C, device='cuda', generator=gen)
    x = torch.randn(n, C, device='cuda', generator=gen)
    w = torch.randn(K, C, C, device='cuda', generator=gen) * 0.2
    # the exact launch over the first `live` rows needs a table whose entries >= live are absent: rebuild it from the prefix
    from sgnn_amd.scn.metadata import Grid
    gl = Grid(g.coords[:live].contiguous())
    tl = gl.subm_table()
    want_dx, want_dw, _ = _fused(dy[:live].contiguous(), x[:live].contiguous(), w, tl, gl, live)
    # capacity-sized table of the same sites: rows >= live do not exist, their entries and references to them are -1
    tcap = torch.full((K, g.ld), -1, dtype=torch.int32, device='cuda')
    tcap[:, :live] = tl.view(K, gl.ld)[:, :live]
    n_dev = torch.tensor([live], dtype=torch.int64, device='cuda')
    dx = torch.full((n, C), float('nan'), device='cuda')
    got_dx, got_dw, _ = _fused(dy, x, w, tcap.view(-1), g, n, None, None, n_dev=n_dev, dx=dx)
    assert torch.equal(got_dx[:live], want_dx)
    assert torch.isnan(got_dx[((live + 255) // 256) * 256:]).all()          # nothing written past the live tiles
    assert float((got_dw - want_dw).abs().max()) < 4e-6 * float(want_dw.abs().max())


def test_training_step_with_and_without_the_fused_kernel(small_levels):
    """the whole model through the program executor (sgnn_prog_backward): loss identical, every parameter gradient agrees
    between the fused and the two-kernel backward (dX rows are bit-identical, so everything upstream of a fused layer sees
    the same gradient; dW differs by fp32 summation order only), and the fused kernel really ran"""
    from sgnn_amd import _lib, synth, loss as L
    from sgnn_amd.model import GenModel
    from util import param_fill
    lib = _lib.load()
    dims, cfg = (32, 32, 32), 17
    data = synth.make_batch(2, dims, cfg=cfg, occupancy=0.08)
    lw = np.ones(5, dtype=np.float32)
    outs = []
    for on in (1, 0):
        prev = _lib.tune('conv_bwd_fused', on)
        try:
            m = param_fill(GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1), cfg).train().cuda()
            t = L.compute_targets(data['sdf'].clone().cuda(), [h.clone().cuda() for h in data['hierarchy']], 4, 3, True,
                                  data['known'].cuda())
            osdf, oocc = m([data['input'][0].cuda(), data['input'][1].cuda()], lw)
            loss, _ = L.compute_loss(osdf, oocc, t[0], t[1], t[2], lw, 3, True, 5.0, data['input'][0].cuda(), True,
                                     data['known'].cuda())
            lib.sgnn_prof_enable(4096)
            loss.backward()
            torch.cuda.synchronize()
            kinds = []
            kind, cin, cout, kk, flags = (__import__('ctypes').c_int() for _ in range(5))
            n_out, ms = __import__('ctypes').c_int64(), __import__('ctypes').c_float()
            for i in range(lib.sgnn_prof_count()):
                B = __import__('ctypes').byref
                if lib.sgnn_prof_get(i, B(kind), B(n_out), B(cin), B(cout), B(kk), B(flags), B(ms)) == 0:
                    kinds.append(kind.value)
            lib.sgnn_prof_disable()
            outs.append((loss.item(), {n: p.grad.clone() for n, p in m.named_parameters()}, kinds.count(2)))
        finally:
            _lib.tune('conv_bwd_fused', prev)
    (la, ga, na), (lb, gb, nb) = outs
    assert na > 0 and nb == 0, 'fused launches: %d with the switch on, %d with it off' % (na, nb)
    assert la == lb
    for n in ga:
        scale = max(1.0, float(gb[n].abs().max()))
        assert float((ga[n] - gb[n]).abs().max()) <= 1e-5 * scale, n
