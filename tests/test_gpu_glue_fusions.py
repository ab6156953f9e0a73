"""Round 5 launch fusions of the geometry / glue kernels (csrc/grid_rules.hip): every one of them has a switch, and the
switch must not change a single bit of any table, site list or count —

* sgnn_tune.scan_inline      the write kernels of the compactions / stride-2 levels sum the raw block counts themselves
                            (no scan launch between the count and the write kernel);
* sgnn_tune.chain_merged     tables pass of level l + hash insertion of level l + 1 in one launch;
* SGNN_FUSED_GLUE / functions.FUSED_GLUE   kept coordinates written by the compaction's write kernel, children and
                            their int64 rows in one pass;
* SGNN_VOLUME_ONLY / metadata.VOLUME_ONLY  generated levels build their 3x3x3 rulebook from the dense index volume alone
                            and never build a hash grid of their own.

The reference computes all of this on the host inside SparseConvNet's Metadata (torch/model.py:192-207, 233-243, 322-336
are its call sites); the oracle parity of the default path is tests/test_gpu_configs.py's."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from sgnn_amd import _lib as L
    return L.load()


@pytest.mark.parametrize('keep_cap', [1 << 20, 37])
def test_compaction_with_locs_equals_compaction_plus_gather(keep_cap):
    """sgnn_compact_sigmoid_cap_locs == sgnn_compact_sigmoid_cap + sgnn_gather_rows_dn, with and without the scan launch,
    also when the kept count is clamped (overflow flagged, counts clamped, locs = the first keep_cap kept rows)."""
    from sgnn_amd import _lib as L
    lib = _lib()
    dev = torch.device('cuda')
    g = torch.Generator().manual_seed(5)
    n_cap, n_live = 9000, 7777                      # 5 scan blocks, the last one ragged, live count on the device
    logits = torch.randn(n_cap, 2, generator=g).to(dev)
    coords = torch.randint(0, 60, (n_cap, 4), generator=g, dtype=torch.int32).to(dev)
    n_dev = torch.tensor([n_live], dtype=torch.int64, device=dev)
    wsb = L.query('sgnn_compact_ws_bytes', n_cap)
    want_all = torch.nonzero(logits[:n_live, 0] > 0)[:, 0].to(torch.int32)   # sigmoid(x) > 0.5  <=>  x > 0 (x == 0: 0.5, dropped)
    K = min(keep_cap, n_cap)
    results = []
    for inline in (1, 0):
        prev = L.tune('scan_inline', inline)
        try:
            for fused in (True, False):
                sel = torch.full((n_cap,), -7, dtype=torch.int32, device=dev)
                locs = torch.full((K, 4), -7, dtype=torch.int32, device=dev)
                cnt2 = torch.full((2,), -7, dtype=torch.int64, device=dev)
                status = torch.zeros(1, dtype=torch.int32, device=dev)
                ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
                if fused:
                    L.call('sgnn_compact_sigmoid_cap_locs', logits.data_ptr(), 2, n_cap, n_dev.data_ptr(), coords.data_ptr(),
                           sel.data_ptr(), locs.data_ptr(), cnt2.data_ptr(), K, status.data_ptr(), ws.data_ptr(), wsb)
                else:
                    L.call('sgnn_compact_sigmoid_cap', logits.data_ptr(), 2, n_cap, n_dev.data_ptr(), sel.data_ptr(),
                           cnt2.data_ptr(), K, status.data_ptr(), ws.data_ptr(), wsb)
                    L.call('sgnn_gather_rows_dn', coords.data_ptr(), 4, sel.data_ptr(), cnt2.data_ptr(), K, locs.data_ptr())
                torch.cuda.synchronize()
                kept = int(cnt2[0])
                results.append((sel[:want_all.shape[0]].clone(), locs[:kept].clone(), cnt2.clone(), int(status[0])))
        finally:
            L.tune('scan_inline', prev)
    total = int(want_all.shape[0])
    for sel, locs, cnt2, status in results:
        assert torch.equal(sel.cpu(), want_all.cpu())
        assert int(cnt2[0]) == min(total, K) and int(cnt2[1]) == 8 * min(total, K)
        assert (status != 0) == (total > K)
        assert torch.equal(locs.cpu(), coords[want_all.long()[:min(total, K)]].cpu())


def test_expand8_with_int64_rows_equals_the_two_launches():
    from sgnn_amd.scn import functions as F_
    dev = torch.device('cuda')
    c = torch.randint(0, 500, (1234, 4), generator=torch.Generator().manual_seed(1), dtype=torch.int32).to(dev)
    was = F_.FUSED_GLUE
    try:
        F_.FUSED_GLUE = True
        a = F_.expand8_coords(c, with_i64=True)
        a64 = F_.coords_to_i64(a)
        assert a64 is a._sgnn_i64
        F_.FUSED_GLUE = False
        b = F_.expand8_coords(c, with_i64=True)
        assert getattr(b, '_sgnn_i64', None) is None
        b64 = F_.coords_to_i64(b)
    finally:
        F_.FUSED_GLUE = was
    assert torch.equal(a, b) and torch.equal(a64, b64) and a64.dtype == torch.int64
    assert torch.equal(a64.cpu(), a.cpu().to(torch.int64))


@pytest.mark.parametrize('order', ['raster', 'shuffled', 'children'])
def test_volume_only_rulebook_equals_hash_rulebook_and_flags_uncovered_sites(order):
    """sgnn_rulebook_subm3_volume (generated levels: every site inside the index volume by construction, no hash grid of
    the level) builds the hash rulebook's table; a site outside the volume raises SGNN_STATUS_COORD_RANGE instead."""
    from sgnn_amd import synth, _lib as L
    from sgnn_amd.scn import functions as F_
    from sgnn_amd.scn.metadata import Grid, coords_from_locs
    dev = torch.device('cuda')
    dims = (32, 32, 32)
    locs = synth.make_batch(3, dims, cfg=9, occupancy=0.1)['input'][0]
    extra = torch.tensor([[0, 0, 0, 0], [31, 31, 31, 1], [0, 31, 0, 2], [31, 0, 31, 0]], dtype=locs.dtype)
    locs = torch.unique(torch.cat([locs, extra]), dim=0)
    if order == 'shuffled':
        locs = locs[torch.randperm(locs.shape[0], generator=torch.Generator().manual_seed(0))]
    coords = coords_from_locs(locs, dev)
    if order == 'children':
        coords = F_.expand8_coords(coords)
        dims = (64, 64, 64)
    g = Grid(coords)
    ref = g.subm_table().clone()                      # hash grid + mirrored probes
    entries = 3 * dims[0] * dims[1] * dims[2]
    vol = torch.full((entries,), -1, dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    out = torch.full_like(ref, 12345)
    L.call('sgnn_rulebook_subm3_volume', coords.data_ptr(), g.n, dims[0], dims[1], dims[2], vol.data_ptr(), entries,
           out.data_ptr(), g.ld, None, status.data_ptr())
    assert torch.equal(out, ref)
    assert int((vol != -1).sum()) == 0 and int(status[0]) == 0
    # one block too few: the sites of block 2 are not covered -> flagged
    L.call('sgnn_rulebook_subm3_volume', coords.data_ptr(), g.n, dims[0], dims[1], dims[2], vol.data_ptr(), entries * 2 // 3,
           out.data_ptr(), g.ld, None, status.data_ptr())
    torch.cuda.synchronize()
    assert int(status[0]) & 1 and int((vol != -1).sum()) == 0


def test_generated_levels_take_the_volume_only_path_and_nothing_changes(monkeypatch):
    """Coordinates of generated levels carry their bound (dense coarse volume -> x2 per level); a level large enough for the
    dense rulebook builder then never builds a hash grid — same site lists, logits, loss, gradients."""
    from test_gpu_capacity import _batch, _model, _classic, _capped
    from sgnn_amd.scn.capacity import Capacity, trim
    from sgnn_amd.scn import metadata as MD
    lw = np.ones(5, dtype=np.float32)
    batch = _batch(3)
    _, _, _, log = _classic(_model(), batch, lw)
    monkeypatch.setattr(MD, 'DENSE_RULEBOOK_MIN_ROWS', 256)
    built = []
    real_hash = MD.Grid.hash

    def counting_hash(self):
        if self.keys is None:
            built.append(self.n)
        return real_hash(self)
    monkeypatch.setattr(MD.Grid, 'hash', counting_hash)
    runs = []
    for on in (True, False):
        monkeypatch.setattr(MD, 'VOLUME_ONLY', on)
        del built[:]
        m = _model()
        cap = Capacity.from_log('cuda', log, headroom=1.4)
        osdf, oocc, loss = _capped(m, batch, lw, cap)
        torch.cuda.synchronize()
        assert int(MD.runtime(torch.device('cuda', 0)).state[1].item()) & 1 == 0
        runs.append((osdf, oocc, loss.detach().clone(), [p.grad.clone() for p in m.parameters()], len(built)))
    (sa, oa, la, ga, ha), (sb, ob, lb, gb, hb) = runs
    assert ha < hb, 'the volume-only path did not replace any hash build (%d vs %d)' % (ha, hb)
    assert torch.equal(la, lb)
    for (ca, xa), (cb, xb) in list(zip(oa, ob)) + [(sa, sb)]:
        ta, tb = trim(ca), trim(cb)
        assert torch.equal(ta, tb)
        assert torch.equal(xa.detach()[:ta.shape[0]], xb.detach()[:tb.shape[0]])
    for a, b in zip(ga, gb):
        assert torch.equal(a, b)


def test_capacity_forward_is_bit_identical_with_the_fusions_off():
    """The whole capacity-mode forward + backward (compactions, stride-2 chains with their tables, rulebooks, every level's
    site list, logits, loss, gradients) with the fusions on (the default) and with all of them off."""
    from test_gpu_capacity import _batch, _model, _classic, _capped
    from sgnn_amd.scn.capacity import Capacity
    from sgnn_amd.scn import functions as F_
    from sgnn_amd import _lib as L
    lib = _lib()
    lw = np.ones(5, dtype=np.float32)
    batch = _batch(3)
    _, _, _, log = _classic(_model(), batch, lw)
    runs = []
    for on in (True, False):
        prev = (L.tune('scan_inline', int(on)), L.tune('chain_merged', int(on)), F_.FUSED_GLUE)
        F_.FUSED_GLUE = on
        try:
            m = _model()
            cap = Capacity.from_log('cuda', log, headroom=1.4)
            osdf, oocc, loss = _capped(m, batch, lw, cap)
            torch.cuda.synchronize()
            live = cap.read()
            grads = [p.grad.clone() for p in m.parameters()]
            runs.append((osdf, oocc, loss.detach().clone(), live, grads))
        finally:
            L.tune('scan_inline', prev[0])
            L.tune('chain_merged', prev[1])
            F_.FUSED_GLUE = prev[2]
    from sgnn_amd.scn.capacity import trim
    (sa, oa, la, lva, ga), (sb, ob, lb, lvb, gb) = runs
    assert lva == lvb
    assert torch.equal(la, lb)
    for (ca, xa), (cb, xb) in list(zip(oa, ob)) + [(sa, sb)]:
        ta, tb = trim(ca), trim(cb)                         # rows past the live count of a capacity-sized tensor are undefined
        assert torch.equal(ta, tb)
        assert torch.equal(xa.detach()[:ta.shape[0]], xb.detach()[:tb.shape[0]])
    for a, b in zip(ga, gb):
        assert torch.equal(a, b)
