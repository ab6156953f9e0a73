"""HIP operators vs the CPU oracle on identical seeded inputs.
Indices / rulebooks must be bit-identical; fp32 features within 1e-4 (north_star tolerance)."""
import numpy as np
import pytest
import torch

import scn_oracle as oscn
from util import random_sites, copy_params, triples_from_table

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _hip():
    import sgnn_amd.scn as scn
    return scn


def _pair(locs, feats, size):
    scn = _hip()
    xo = oscn.InputLayer(3, size, mode=0)([locs, feats])
    xh = scn.InputLayer(3, size, mode=0)([locs.cuda(), feats.cuda()])
    return xo, xh


@pytest.mark.parametrize('surface', [False, True])
def test_subm_rulebook_bit_identical(surface):
    locs = random_sites(3, 24, 0.08, 1, surface)
    feats = torch.zeros(locs.shape[0], 1)
    xo, xh = _pair(locs, feats, [24, 24, 24])
    g = xh.grid()
    got = triples_from_table(g.subm_table(), 27, g.ld, g.n)
    nbr = xo.metadata.grid(xo.spatial_size).subm_rules(3)
    k, j = np.nonzero(nbr >= 0)
    want = np.stack([k, nbr[k, j], j], 1)
    want = want[np.lexsort((want[:, 2], want[:, 1], want[:, 0]))]
    assert got.shape == want.shape and np.array_equal(got, want)


def test_down2_sites_and_rules_bit_identical():
    locs = random_sites(2, 16, 0.15, 2)
    feats = torch.zeros(locs.shape[0], 1)
    xo, xh = _pair(locs, feats, [16, 16, 16])
    d = xh.metadata.down2(xh.spatial_size, xh.spatial_size // 2)
    parent_o, off_o = xo.metadata.down2(xo.spatial_size, xo.spatial_size // 2)
    coarse_o = xo.metadata.grid(xo.spatial_size // 2).coords
    assert d.coarse.n == coarse_o.shape[0]
    assert np.array_equal(d.coarse.coords.cpu().numpy().astype(np.int64), coarse_o)  # first-touch order, exact
    assert np.array_equal(d.parent.cpu().numpy().astype(np.int64), parent_o)
    ch = d.children.view(8, d.ldc)[:, :d.coarse.n].cpu().numpy()
    want = np.full_like(ch, -1)
    want[off_o, parent_o] = np.arange(len(parent_o))
    assert np.array_equal(ch, want)
    # the coarse hash must resolve every coarse site to its own row
    rows = d.coarse.lookup(d.coarse.coords)
    assert np.array_equal(rows.cpu().numpy(), np.arange(d.coarse.n))


CONV_CASES = [(1, 8), (8, 8), (8, 12), (12, 12), (12, 16), (16, 16), (34, 16), (30, 16), (26, 16), (48, 16), (5, 7)]


@pytest.mark.parametrize('cin,cout', CONV_CASES)
def test_subm_conv_fwd_bwd(cin, cout):
    scn = _hip()
    torch.manual_seed(cin * 100 + cout)
    locs = random_sites(2, 20, 0.2, cin + cout, surface=(cin % 2 == 0))
    feats = torch.randn(locs.shape[0], cin)
    fo = feats.clone().requires_grad_(True)
    fh = feats.clone().cuda().requires_grad_(True)
    mo = oscn.SubmanifoldConvolution(3, cin, cout, 3, False)
    mh = scn.SubmanifoldConvolution(3, cin, cout, 3, False).cuda()
    copy_params(mo, mh)
    yo = mo(oscn.InputLayer(3, [20] * 3, mode=0)([locs, fo])).features
    yh = mh(scn.InputLayer(3, [20] * 3, mode=0)([locs.cuda(), fh])).features
    assert (yo - yh.cpu()).abs().max().item() < TOL
    go = torch.randn_like(yo)
    yo.backward(go)
    yh.backward(go.cuda())
    assert (fo.grad - fh.grad.cpu()).abs().max().item() < TOL
    scale = max(1.0, mo.weight.grad.abs().max().item())
    assert (mo.weight.grad - mh.weight.grad.cpu()).abs().max().item() < TOL * scale


@pytest.mark.parametrize('c', [8, 12, 16, 7])
def test_strided_conv_unpool_fwd_bwd(c):
    scn = _hip()
    torch.manual_seed(c)
    locs = random_sites(2, 16, 0.2, 7 + c)
    feats = torch.randn(locs.shape[0], c)
    fo = feats.clone().requires_grad_(True)
    fh = feats.clone().cuda().requires_grad_(True)
    mo, mh = oscn.Convolution(3, c, c, 2, 2, False), scn.Convolution(3, c, c, 2, 2, False).cuda()
    copy_params(mo, mh)
    xo = oscn.InputLayer(3, [16] * 3, mode=0)([locs, fo])
    xh = scn.InputLayer(3, [16] * 3, mode=0)([locs.cuda(), fh])
    zo, zh = mo(xo), mh(xh)
    assert zo.features.shape == zh.features.shape
    assert (zo.features - zh.features.cpu()).abs().max().item() < TOL
    uo, uh = oscn.UnPooling(3, 2, 2)(zo).features, scn.UnPooling(3, 2, 2)(zh).features
    assert torch.equal(uo, uo) and (uo - uh.cpu()).abs().max().item() < TOL
    go = torch.randn_like(uo)
    (uo * go).sum().backward()
    (uh * go.cuda()).sum().backward()
    assert (fo.grad - fh.grad.cpu()).abs().max().item() < TOL
    assert (mo.weight.grad - mh.weight.grad.cpu()).abs().max().item() < TOL * max(1.0, mo.weight.grad.abs().max().item())


def test_deconvolution_matches_oracle():
    scn = _hip()
    torch.manual_seed(3)
    locs = random_sites(2, 16, 0.2, 11)
    feats = torch.randn(locs.shape[0], 8)
    fo, fh = feats.clone().requires_grad_(True), feats.clone().cuda().requires_grad_(True)
    co, ch = oscn.Convolution(3, 8, 8, 2, 2, False), scn.Convolution(3, 8, 8, 2, 2, False).cuda()
    do, dh = oscn.Deconvolution(3, 8, 12, 2, 2, False), scn.Deconvolution(3, 8, 12, 2, 2, False).cuda()
    copy_params(co, ch)
    copy_params(do, dh)
    yo = do(co(oscn.InputLayer(3, [16] * 3, mode=0)([locs, fo]))).features
    yh = dh(ch(scn.InputLayer(3, [16] * 3, mode=0)([locs.cuda(), fh]))).features
    assert (yo - yh.cpu()).abs().max().item() < TOL
    g = torch.randn_like(yo)
    yo.backward(g)
    yh.backward(g.cuda())
    assert (fo.grad - fh.grad.cpu()).abs().max().item() < TOL
    assert (do.weight.grad - dh.weight.grad.cpu()).abs().max().item() < TOL * max(1.0, do.weight.grad.abs().max().item())


@pytest.mark.parametrize('c,train', [(8, True), (12, True), (16, True), (48, True), (16, False), (5, True)])
def test_batchnorm_relu(c, train):
    scn = _hip()
    torch.manual_seed(c)
    locs = random_sites(2, 16, 0.3, c)
    feats = torch.randn(locs.shape[0], c) * 2 + 0.5
    fo, fh = feats.clone().requires_grad_(True), feats.clone().cuda().requires_grad_(True)
    mo, mh = oscn.BatchNormReLU(c), scn.BatchNormReLU(c).cuda()
    with torch.no_grad():
        mo.weight.uniform_(0.5, 1.5)
        mo.bias.uniform_(-0.5, 0.5)
        mo.running_mean.uniform_(-0.2, 0.7)
        mo.running_var.uniform_(2.0, 5.0)
    copy_params(mo, mh)
    mo.train(train)
    mh.train(train)
    yo = mo(oscn.InputLayer(3, [16] * 3, mode=0)([locs, fo])).features
    yh = mh(scn.InputLayer(3, [16] * 3, mode=0)([locs.cuda(), fh])).features
    assert (yo - yh.cpu()).abs().max().item() < TOL
    assert (mo.running_mean - mh.running_mean.cpu()).abs().max().item() < 1e-5
    assert (mo.running_var - mh.running_var.cpu()).abs().max().item() < 1e-5
    g = torch.randn_like(yo)
    yo.backward(g)
    yh.backward(g.cuda())
    assert (fo.grad - fh.grad.cpu()).abs().max().item() < TOL
    assert (mo.weight.grad - mh.weight.grad.cpu()).abs().max().item() < 1e-3
    assert (mo.bias.grad - mh.bias.grad.cpu()).abs().max().item() < 1e-3


def test_fully_convolutional_net_and_sparse_to_dense():
    scn = _hip()
    torch.manual_seed(5)
    locs = random_sites(2, 16, 0.25, 5, surface=True)
    feats = torch.randn(locs.shape[0], 16)
    fo, fh = feats.clone().requires_grad_(True), feats.clone().cuda().requires_grad_(True)
    mo = oscn.Sequential().add(oscn.FullyConvolutionalNet(3, 1, [16, 16, 16], True)).add(oscn.BatchNormReLU(48))
    mh = scn.Sequential().add(scn.FullyConvolutionalNet(3, 1, [16, 16, 16], True)).add(scn.BatchNormReLU(48)).cuda()
    assert list(mo.state_dict().keys()) == list(mh.state_dict().keys())
    copy_params(mo, mh)
    xo = mo(oscn.InputLayer(3, [16] * 3, mode=0)([locs, fo]))
    xh = mh(scn.InputLayer(3, [16] * 3, mode=0)([locs.cuda(), fh]))
    assert (xo.features - xh.features.cpu()).abs().max().item() < TOL
    do, dh = oscn.SparseToDense(3, 48)(xo), scn.SparseToDense(3, 48)(xh)
    assert do.shape == dh.shape and (do - dh.cpu()).abs().max().item() < TOL
    g = torch.randn_like(do)
    (do * g).sum().backward()
    (dh * g.cuda()).sum().backward()
    assert (fo.grad - fh.grad.cpu()).abs().max().item() < 5 * TOL
    for (ko, po), (kh, ph) in zip(mo.named_parameters(), mh.named_parameters()):
        assert ko == kh
        assert (po.grad - ph.grad.cpu()).abs().max().item() < 1e-3 * max(1.0, po.grad.abs().max().item()), ko


def test_duplicate_sites_raise():
    scn = _hip()
    from sgnn_amd._lib import SgnnError
    locs = torch.tensor([[1, 2, 3, 0], [1, 2, 3, 0], [2, 2, 2, 0]])
    x = scn.InputLayer(3, [8] * 3, mode=0)([locs.cuda(), torch.zeros(3, 1).cuda()])
    x.grid().hash()
    with pytest.raises(SgnnError):
        scn.runtime().read_count()


def test_empty_input_is_legal():
    scn = _hip()
    x = scn.InputLayer(3, [8] * 3, mode=0)([torch.zeros(0, 4, dtype=torch.long).cuda(), torch.zeros(0, 4).cuda()])
    y = scn.SubmanifoldConvolution(3, 4, 16, 3, False).cuda()(x)
    assert y.features.shape == (0, 16)
    z = scn.Convolution(3, 16, 16, 2, 2, False).cuda()(y)
    assert z.features.shape == (0, 16)


@pytest.mark.parametrize('order', ['raster', 'shuffled', 'children'])
def test_lds_window_rulebook_equals_global_probe_rulebook(order):
    """k_rulebook_subm3_lds (voxel index of a row window in LDS, global table only for misses) must produce the very
    table of the global-probe kernel for every site order: sorted raster blocks, a random permutation (nothing is
    near by: all look-ups fall through to the global table) and the 8-children-per-parent order of generated levels."""
    from sgnn_amd import synth, _lib
    from sgnn_amd.scn import functions as F_
    from sgnn_amd.scn.metadata import Grid, coords_from_locs
    lib = _lib.load()
    locs = synth.make_batch(3, (32, 32, 32), cfg=9, occupancy=0.1)['input'][0]
    if order == 'shuffled':
        locs = locs[torch.randperm(locs.shape[0], generator=torch.Generator().manual_seed(0))]
    coords = coords_from_locs(locs, torch.device('cuda'))
    if order == 'children':
        coords = F_.expand8_coords(coords)
    tabs = []
    for on in (1, 0):
        prev = _lib.tune('rulebook_lds', on)
        try:
            g = Grid(coords)
            tabs.append(g.subm_table().clone())
        finally:
            _lib.tune('rulebook_lds', prev)
    assert torch.equal(tabs[0], tabs[1])
    assert int((tabs[0].view(27, -1)[:, :coords.shape[0]] >= 0).sum()) > 27 * coords.shape[0] // 4


@pytest.mark.parametrize('order', ['raster', 'shuffled', 'children'])
@pytest.mark.parametrize('volume_blocks', [8, 1, 0])
def test_dense_volume_rulebook_equals_hash_rulebook(order, volume_blocks):
    """sgnn_rulebook_subm3_dense (neighbours read from a dense index volume) must produce the hash rulebook's table for
    every input: all site orders, sites on the volume boundary, sites outside the declared spatial size, and batch
    indices the volume does not cover (volume_blocks = 1: blocks 1, 2 go through the hash; 0: everything does).  The
    volume must be all -1 again afterwards."""
    from sgnn_amd import synth, _lib
    from sgnn_amd.scn import functions as F_
    from sgnn_amd.scn.metadata import Grid, coords_from_locs
    dev = torch.device('cuda')
    dims = (32, 32, 32)
    locs = synth.make_batch(3, dims, cfg=9, occupancy=0.1)['input'][0]
    # boundary faces / corners and a few sites outside the declared size (their neighbours must still find them)
    extra = torch.tensor([[0, 0, 0, 0], [31, 31, 31, 1], [0, 31, 0, 2], [32, 5, 5, 0], [31, 5, 5, 0], [33, 5, 5, 0],
                          [5, 32, 31, 1], [5, 31, 31, 1]], dtype=locs.dtype)
    locs = torch.unique(torch.cat([locs, extra]), dim=0)
    if order == 'shuffled':
        locs = locs[torch.randperm(locs.shape[0], generator=torch.Generator().manual_seed(0))]
    coords = coords_from_locs(locs, dev)
    if order == 'children':
        coords = F_.expand8_coords(coords)
        dims = (64, 64, 64)
    g = Grid(coords)
    ref = g.subm_table().clone()
    keys, vals, cap = g.hash()
    entries = volume_blocks * dims[0] * dims[1] * dims[2]
    vol = torch.full((max(entries, 1),), -1, dtype=torch.int32, device=dev)
    out = torch.full_like(ref, 12345)
    _lib.call('sgnn_rulebook_subm3_dense', keys.data_ptr(), vals.data_ptr(), cap, coords.data_ptr(), g.n, dims[0],
              dims[1], dims[2], vol.data_ptr(), entries, out.data_ptr(), g.ld, None)
    assert torch.equal(out, ref)
    assert int((vol != -1).sum()) == 0


@pytest.mark.parametrize('live', ['all', 'fewer'])
def test_multi_level_rulebook_equals_per_level_rulebooks(live):
    """sgnn_rulebook_subm3_multi (the coarse levels of a hierarchy in one pre-fill + one builder launch) against one
    sgnn_rulebook_subm3 per level: every table entry for entry, entries beyond the padded live range untouched, with an
    empty level in the list, ld > n, and device-resident counts below the capacity."""
    import numpy as np
    from sgnn_amd import synth, _lib
    from sgnn_amd.scn.metadata import Grid, coords_from_locs
    dev = torch.device('cuda')
    levels = []
    for i, (dims, occ) in enumerate([((32, 32, 32), 0.12), ((16, 16, 16), 0.3), ((8, 8, 8), 0.0), ((8, 8, 8), 0.5)]):
        if occ == 0.0:
            levels.append(None)
            continue
        locs = synth.make_batch(3, dims, cfg=20 + i, occupancy=occ)['input'][0]
        g = Grid(coords_from_locs(locs, dev))
        keys, vals, cap = g.hash()
        n_live = g.n if live == 'all' else max(1, (g.n * 2) // 3)
        levels.append((g, keys, vals, cap, torch.tensor([n_live], dtype=torch.int64, device=dev)))
    some = next(l for l in levels if l is not None)

    def build(multi):
        outs = [torch.full((27 * (l[0].ld + 256 if l else 256),), 12345, dtype=torch.int32, device=dev) for l in levels]
        a = lambda f, z=0: np.array([f(l) if l else z for l in levels], dtype=np.int64)
        arrs = [a(lambda l: l[1].data_ptr(), some[1].data_ptr()), a(lambda l: l[2].data_ptr(), some[2].data_ptr()),
                a(lambda l: l[3], 2), a(lambda l: l[0].coords.data_ptr()), a(lambda l: l[0].n),
                np.array([o.data_ptr() for o in outs], dtype=np.int64), a(lambda l: l[0].ld + 256, 256),
                a(lambda l: l[4].data_ptr())]
        old = _lib.tune('rulebook_multi', multi)
        try:
            _lib.call('sgnn_rulebook_subm3_multi', len(levels), *[x.ctypes.data for x in arrs])
            torch.cuda.synchronize()
        finally:
            _lib.tune('rulebook_multi', old)
        return outs
    one, many = build(0), build(1)
    for l, o, m in zip(levels, one, many):
        assert torch.equal(o, m)
        if l is None:
            assert int((m != 12345).sum()) == 0
        else:
            g, n_live = l[0], int(l[4])
            t = m.view(27, g.ld + 256)
            assert torch.equal(t[13, :n_live], torch.arange(n_live, dtype=torch.int32, device=dev))
            if live == 'all':        # (a grid hashed beyond the live count hands out rows >= n_live: same in both builds)
                assert torch.equal(t[:, :g.n], g.subm_table().view(27, g.ld)[:, :g.n])
                assert int((t[:, g.n:(g.n + 255) // 256 * 256] != -1).sum()) == 0
    with pytest.raises(_lib.SgnnError):
        _lib.call('sgnn_rulebook_subm3_multi', 5, *[np.zeros(5, dtype=np.int64).ctypes.data] * 8)


def test_registered_levels_use_the_dense_rulebook_and_match():
    """Grids registered in a Metadata know their spatial size and build large rulebooks through the volume."""
    from sgnn_amd import synth
    import sgnn_amd.scn as scn
    from sgnn_amd.scn import metadata as MD
    locs = synth.make_batch(4, (64, 64, 64), cfg=2)['input'][0].cuda()
    assert locs.shape[0] >= MD.DENSE_RULEBOOK_MIN_ROWS
    feats = torch.zeros(locs.shape[0], 1, device='cuda')
    x = scn.InputLayer(3, (64, 64, 64), mode=0)([locs, feats])
    g = x.grid()
    assert g.dims == (64, 64, 64)
    tab = g.subm_table()
    ref = MD.Grid(g.coords).subm_table()       # unregistered: hash path
    assert torch.equal(tab, ref)
    assert int((MD.runtime(locs.device).index_volume() != -1).sum()) == 0


@pytest.mark.parametrize('cin,cout', [(16, 16), (8, 8), (8, 12), (26, 16), (1, 8), (48, 16)])
@pytest.mark.parametrize('batch,dim', [(2, 24), (5, 64)])     # one-offset-per-workgroup variant / 256-row blocks x offset groups
def test_convolution_kernels_never_read_outside_their_slabs(cin, cout, batch, dim):
    """the weight-gradient kernel once issued dy tail loads whose 32-bit offsets wrapped INTO the buffer
    (harmless only while the matching x rows were exact zeros).  Here x, dy and the forward output live in the middle of
    NaN-filled allocations — the table's padding entries are -1, the row count is not a multiple of the 256-row blocks — so
    any load that leaves its slab (rows >= n, the row tail, a wrapped offset) poisons the result.  Forward, data gradient
    and weight gradient must be finite and equal to the same calls on clean, exactly-sized tensors."""
    from sgnn_amd.scn import functions as F_
    from sgnn_amd.scn.metadata import Grid, coords_from_locs
    locs = random_sites(batch, dim, 0.1, 5, surface=True)
    g = Grid(coords_from_locs(locs, torch.device('cuda')))
    tab, n = g.subm_table(), g.n
    assert n % 256 != 0
    gen = torch.Generator(device='cuda').manual_seed(cin + 100 * cout)
    x = torch.randn(n, cin, device='cuda', generator=gen)
    dy = torch.randn(n, cout, device='cuda', generator=gen)
    w = torch.randn(27, cin, cout, device='cuda', generator=gen) * 0.2
    pad = 4096

    def poisoned(t):
        big = torch.full((t.numel() + 2 * pad,), float('nan'), device='cuda')
        big[pad:pad + t.numel()] = t.reshape(-1)
        return big, big[pad:pad + t.numel()].view_as(t)
    bx, px = poisoned(x)
    bdy, pdy = poisoned(dy)
    want_y = F_.conv_fwd_raw(x, cin, w, 27, tab, g.ld, n, cout)
    got_y = F_.conv_fwd_raw(px, cin, w, 27, tab, g.ld, n, cout)
    assert torch.isfinite(got_y).all() and torch.equal(got_y, want_y)
    flags = F_.CONV_TRANSPOSE_W | F_.CONV_FLIP_K
    want_dx = F_.conv_fwd_raw(dy, cout, w, 27, tab, g.ld, n, cin, flags)
    got_dx = F_.conv_fwd_raw(pdy, cout, w, 27, tab, g.ld, n, cin, flags)
    assert torch.isfinite(got_dx).all() and torch.equal(got_dx, want_dx)
    want_dw = F_.conv_dw_raw(x, cin, dy, cout, tab, g.ld, 27, n)
    got_dw = F_.conv_dw_raw(px, cin, pdy, cout, tab, g.ld, 27, n)
    assert torch.isfinite(got_dw).all(), 'weight gradient read outside its slabs'
    assert torch.equal(got_dw, want_dw)
    assert torch.isnan(bx[:pad]).all() and torch.isnan(bdy[-pad:]).all()      # (the guard regions are intact)
