"""sgnn_down2_chain (stride-2 pyramid of several levels with device-side row counts, one host read-back) must give
exactly what level-by-level sgnn_rulebook_down2 gives — same first-touch coarse order, parents, hashes, tables —
and the model must use 5 host read-backs per forward instead of 15 (scn.Convolution(2,2) rulebooks, torch/model.py:44
and the FullyConvolutionalNet bodies; mask compactions :233-247, :322-336)."""
import numpy as np
import pytest
import torch

from util import random_sites, param_fill

pytestmark = pytest.mark.gpu


def _levels(md, key, depth):
    out = []
    for _ in range(depth):
        nxt = tuple(v // 2 for v in key)
        out.append(md.down2(key, nxt))
        key = nxt
    return out


@pytest.mark.parametrize('surface', [False, True])
def test_chain_equals_level_by_level(surface):
    from sgnn_amd.scn import metadata as MD
    locs = random_sites(3, 32, 0.1, 5, surface)
    res = []
    for chain in (True, False):
        MD.CHAIN = chain
        try:
            rt = MD.runtime(torch.device('cuda'))
            g = MD.Grid(MD.coords_from_locs(locs, torch.device('cuda')))
            md = MD.Metadata()
            md.grids[(32, 32, 32)] = g
            s0 = rt.syncs
            md.prebuild((32, 32, 32), 3)
            ds = _levels(md, (32, 32, 32), 3)
            res.append((ds, rt.syncs - s0))
        finally:
            MD.CHAIN = True
    (a, sa), (b, sb) = res
    assert sa == 1 and sb == 3
    for da, db in zip(a, b):
        assert da.coarse.n == db.coarse.n and da.coarse.n > 0
        assert torch.equal(da.coarse.coords, db.coarse.coords)
        assert torch.equal(da.parent, db.parent)
        assert torch.equal(da.children.view(8, -1)[:, :da.coarse.n], db.children.view(8, -1)[:, :db.coarse.n])
        assert torch.equal(da.ptable.view(8, -1)[:, :da.fine.n], db.ptable.view(8, -1)[:, :db.fine.n])
        rows = da.coarse.lookup(db.coarse.coords)        # the chain's coarse hash resolves every coarse site
        assert torch.equal(rows.cpu(), torch.arange(db.coarse.n, dtype=torch.int32))
        # 3x3x3 rulebooks of the coarse levels agree as well (they are built from that hash)
        assert torch.equal(da.coarse.subm_table().view(27, -1)[:, :da.coarse.n],
                           db.coarse.subm_table().view(27, -1)[:, :db.coarse.n])


def test_model_forward_uses_five_readbacks_and_same_results():
    from sgnn_amd import synth
    from sgnn_amd.model import GenModel
    from sgnn_amd.scn import metadata as MD
    dims, cfg = (32, 32, 32), 23
    data = synth.make_batch(2, dims, cfg=cfg, occupancy=0.08)
    lw = np.ones(5, dtype=np.float32)
    m = param_fill(GenModel(8, dims, 1, 16, 16, 4, True, True, 1, 1), cfg).train().cuda()
    rt = MD.runtime(torch.device('cuda'))
    outs = []
    for chain in (True, False):
        MD.CHAIN = chain
        try:
            s0 = rt.syncs
            with torch.no_grad():
                sdf, occ = m([data['input'][0].cuda(), data['input'][1].cuda()], lw, batch_size=2)
            outs.append((sdf, occ, rt.syncs - s0))
        finally:
            MD.CHAIN = True
    (sa, oa, na), (sb, ob, nb) = outs
    assert na == 5 and nb == 15, (na, nb)
    for h in range(4):
        assert torch.equal(oa[h][0], ob[h][0]) and torch.equal(oa[h][1], ob[h][1])
    assert torch.equal(sa[0], sb[0]) and torch.equal(sa[1], sb[1])
