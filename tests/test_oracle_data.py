"""CPU: the data oracle and the host loader against fixtures decoded by the reference's own loader
(tests/golden/make_golden_data.py ran /root/reference/torch/data_util.py + scene_dataloader.py)."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))
import data_oracle  # noqa: E402

from sgnn_amd import data  # noqa: E402

DATA = os.path.join(HERE, 'golden', 'data')
CHUNKS = [os.path.join(DATA, 'chunk_%d.sdfs' % i) for i in range(3)]
S_IN = os.path.join(DATA, 'scene_in', 'scene0.sdf')
S_TGT_DIR = os.path.join(DATA, 'scene_tgt')
S_TGT = os.path.join(S_TGT_DIR, 'scene0.sdf')
TRUNC = 3.0


@pytest.fixture(scope='module')
def exp():
    return np.load(os.path.join(HERE, 'golden', 'data_expected.npz'))


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    assert np.array_equal(a, b, equal_nan=True)          # bit-exact, -inf included


@pytest.mark.parametrize('mod', [data_oracle, data], ids=['oracle', 'host'])
def test_file_readers_match_reference(exp, mod):
    for i, p in enumerate(CHUNKS):
        (il, iv), tgt, dims, w2g, known, hier = mod.load_train_file(p)
        same(il, exp['c%d_in_locs' % i])
        same(iv, exp['c%d_in_vals' % i])
        same(tgt, exp['c%d_target' % i])
        assert list(dims) == list(exp['c%d_dims' % i])
        same(w2g, exp['c%d_w2g' % i])
        same(known, exp['c%d_known' % i])
        for h in range(3):
            same(hier[h], exp['c%d_hier%d' % (i, h)])
    (sl, sv), sdims, sw2g = mod.load_scene(S_IN)
    same(sl, exp['s_in_locs'])
    same(sv, exp['s_in_vals'])
    assert list(sdims) == list(exp['s_dims'])
    same(sw2g, exp['s_w2g'])
    same(mod.load_scene_known(os.path.splitext(S_TGT)[0] + '.knw'), exp['s_known'])


def check_batch(b, exp, k, hier_levels):
    t = lambda v: v.numpy() if torch.is_tensor(v) else v
    same(t(b['input'][0]), exp[k + 'locs'])
    same(t(b['input'][1]), exp[k + 'feats'])
    same(t(b['sdf']), exp[k + 'sdf'])
    same(t(b['known']), exp[k + 'known'])
    same(t(b['orig_dims']), exp[k + 'orig_dims'])
    if hier_levels:
        assert len(b['hierarchy']) == hier_levels
        for h in range(hier_levels):
            same(t(b['hierarchy'][h]), exp[k + 'hier%d' % h])
    else:
        assert b['hierarchy'] is None


@pytest.mark.parametrize('levels', [4, 3])
def test_chunk_batches_match_reference(exp, levels):
    k = 'b%d_' % levels
    ob = data_oracle.collate([data_oracle.sample_chunk(p, TRUNC, levels) for p in CHUNKS])
    check_batch(ob, exp, k, levels - 1)
    same(ob['world2grid'], exp[k + 'w2g'])
    assert ob['name'] == list(exp[k + 'names'])
    ds = data.SceneDataset(CHUNKS, 16, TRUNC, levels, 0)
    hb = data.collate([ds[i] for i in range(len(ds))])
    check_batch(hb, exp, k, levels - 1)
    same(hb['world2grid'].numpy(), exp[k + 'w2g'])


@pytest.mark.parametrize('height', [16, 0, 128])
def test_scene_batches_match_reference(exp, height):
    k = 's%d_' % height
    check_batch(data_oracle.collate([data_oracle.sample_scene(S_IN, S_TGT, TRUNC, 4, height)]), exp, k, 0)
    ds = data.SceneDataset([S_IN], 0, TRUNC, 4, height, target_path=S_TGT_DIR)
    check_batch(data.collate([ds[0]]), exp, k, 0)


def test_layout_rejects_damaged_files(tmp_path):
    raw = np.fromfile(CHUNKS[0], dtype=np.uint8)
    with pytest.raises(RuntimeError, match='truncated'):
        data.Layout(raw[:len(raw) - 5], data.KIND_CHUNK)
    with pytest.raises(RuntimeError, match='header'):
        data.Layout(raw[:40], data.KIND_CHUNK)
    bad = raw.copy()
    bad[0:8] = 0                                             # dimx = 0
    with pytest.raises(RuntimeError, match='dimensions'):
        data.Layout(bad, data.KIND_CHUNK)
    lay = data.Layout(raw, data.KIND_CHUNK)
    assert int(lay.t[21]) == raw.size and (lay.dimz, lay.dimy, lay.dimx) == (16, 16, 16)
    wrong = raw.copy()                                       # known-mask count must equal the volume
    off = int(lay.t[11]) - 8
    wrong[off:off + 8] = np.array([4095], dtype='<u8').view(np.uint8)
    with pytest.raises(RuntimeError, match='known-mask'):
        data.Layout(wrong, data.KIND_CHUNK)


def test_writer_reader_round_trip(tmp_path):
    from sgnn_amd import synth
    p = str(tmp_path / 'c.sdfs')
    a = synth.write_chunk(p, (8, 16, 24), 5, occupancy=0.2)
    (il, iv), tgt, dims, w2g, known, hier = data.load_train_file(p)
    assert dims == [8, 16, 24] and np.array_equal(known, a['known']) and np.array_equal(w2g, a['world2grid'])
    assert np.array_equal(il, a['input'][0].astype(np.int32))
    assert np.array_equal(iv, a['input'][1] / a['voxelsize'])
    assert [h.shape for h in hier] == [(1, 2, 3), (2, 4, 6), (4, 8, 12)]


@pytest.mark.parametrize('seed', range(4))
def test_readers_match_live_reference_on_random_files(tmp_path, seed):
    """Random chunk / scene files decoded by the reference's own data_util.py (imported live where /root/reference
    exists; plyfile / marching_cubes stubbed as in tests/golden/make_golden_data.py), the oracle and the host loader."""
    ref_dir = '/root/reference/torch'
    if not os.path.isdir(ref_dir):
        pytest.skip('reference sources not present on this machine')
    import types
    # stubs only for the duration of the import (other tests import the REAL marching_cubes_cpp of oracle/_ref)
    added = []
    for name in ('plyfile', 'marching_cubes', 'marching_cubes.marching_cubes'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
            added.append(name)
    sys.modules['marching_cubes'].marching_cubes = sys.modules['marching_cubes.marching_cubes']
    sys.path.insert(0, ref_dir)
    try:
        import data_util as ref_data
    finally:
        sys.path.remove(ref_dir)
        for name in added:
            sys.modules.pop(name, None)
    from sgnn_amd import synth
    rng = np.random.default_rng(seed)
    dims = tuple(int(8 * v) for v in rng.integers(1, 4, 3))
    vs = float(rng.choice([0.02, 0.046875, 0.1]))
    p = str(tmp_path / 'c.sdfs')
    synth.write_chunk(p, dims, 40 + seed, occupancy=float(rng.uniform(0.05, 0.4)), voxelsize=vs)
    want = ref_data.load_train_file(p)
    for mod in (data_oracle, data):
        got = mod.load_train_file(p)
        same(got[0][0], want[0][0])
        same(got[0][1], want[0][1])
        same(got[1], want[1])
        assert list(got[2]) == list(want[2])
        same(got[3], want[3])
        same(got[4], want[4])
        for h in range(3):
            same(got[5][h], want[5][h])
    s_in, s_tgt = str(tmp_path / 'a.sdf'), str(tmp_path / 'b.sdf')
    synth.write_scene_triple(s_in, s_tgt, (int(rng.integers(9, 30)), int(rng.integers(9, 30)), int(rng.integers(9, 30))),
                             60 + seed, occupancy=0.3, voxelsize=vs)
    for mod in (data_oracle, data):
        got, want = mod.load_scene(s_tgt), ref_data.load_scene(s_tgt)
        same(got[0][0], want[0][0])
        same(got[0][1], want[0][1])
        assert list(got[1]) == list(want[1])
        same(mod.load_scene_known(s_tgt[:-4] + '.knw'), ref_data.load_scene_known(s_tgt[:-4] + '.knw'))
