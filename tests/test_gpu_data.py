"""GPU: DeviceBatchLoader (packed file images decoded by the io kernels) against the reference-decoded
fixtures and, at training sizes, against the data oracle.  Bit-exact: integer coordinates, one float32 division."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))
import data_oracle  # noqa: E402

from sgnn_amd import data, synth  # noqa: E402

pytestmark = pytest.mark.gpu

DATA = os.path.join(HERE, 'golden', 'data')
CHUNKS = [os.path.join(DATA, 'chunk_%d.sdfs' % i) for i in range(3)]
S_IN = os.path.join(DATA, 'scene_in', 'scene0.sdf')
S_TGT_DIR = os.path.join(DATA, 'scene_tgt')
TRUNC = 3.0


def same(a, b):
    a = a.cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    assert np.array_equal(a, b, equal_nan=True)


def check(b, e, hier_levels):
    assert b['input'][0].is_cuda and b['sdf'].is_cuda and b['known'].is_cuda
    same(b['input'][0], e['locs'])
    same(b['input'][1], e['feats'])
    same(b['sdf'], e['sdf'])
    same(b['known'], e['known'])
    same(b['orig_dims'], e['orig_dims'])
    if hier_levels:
        assert len(b['hierarchy']) == hier_levels
        for h in range(hier_levels):
            same(b['hierarchy'][h], e['hier%d' % h])
    else:
        assert b['hierarchy'] is None


@pytest.fixture(scope='module')
def exp():
    return np.load(os.path.join(HERE, 'golden', 'data_expected.npz'))


def sub(exp, k):
    return {n[len(k):]: exp[n] for n in exp.files if n.startswith(k)}


@pytest.mark.parametrize('levels', [4, 3])
def test_chunk_batch_matches_reference_fixture(exp, levels):
    batches = list(data.DeviceBatchLoader(CHUNKS, 3, TRUNC, num_hierarchy_levels=levels))
    assert len(batches) == 1
    e = sub(exp, 'b%d_' % levels)
    check(batches[0], e, levels - 1)
    same(batches[0]['world2grid'], e['w2g'])
    assert batches[0]['name'] == list(e['names'])


@pytest.mark.parametrize('height', [16, 0, 128])
def test_scene_batch_matches_reference_fixture(exp, height):
    batches = list(data.DeviceBatchLoader([S_IN], 1, TRUNC, max_input_height=height, target_path=S_TGT_DIR))
    assert len(batches) == 1
    check(batches[0], sub(exp, 's%d_' % height), 0)


def test_training_size_batches_match_oracle(tmp_path):
    files = []
    for i in range(10):                      # 64^3 chunks, 2 full batches of 4 + a dropped remainder
        p = str(tmp_path / ('blk%02d.sdfs' % i))
        synth.write_chunk(p, (64, 64, 64), 300 + i, occupancy=0.05, voxelsize=0.02 + 0.001 * (i % 3))
        files.append(p)
    loader = data.DeviceBatchLoader(files, 4, TRUNC)
    assert len(loader) == 2
    for bi, b in enumerate(loader):
        ob = data_oracle.collate([data_oracle.sample_chunk(p, TRUNC, 4) for p in files[4 * bi:4 * bi + 4]])
        same(b['input'][0], ob['input'][0])
        same(b['input'][1], ob['input'][1])
        same(b['sdf'], ob['sdf'])
        same(b['known'], ob['known'])
        same(b['world2grid'], ob['world2grid'])
        for h in range(3):
            same(b['hierarchy'][h], ob['hierarchy'][h])
    assert bi == 1
    # a second pass reuses the pinned staging buffers and yields the same bytes
    again = next(iter(loader))
    ob = data_oracle.collate([data_oracle.sample_chunk(p, TRUNC, 4) for p in files[:4]])
    same(again['input'][0], ob['input'][0])
    same(again['sdf'], ob['sdf'])


def test_loaded_batch_trains(tmp_path):
    """The decoded batch is what train_step consumes (same keys, dtypes, device)."""
    from sgnn_amd import model as M, train
    files = []
    for i in range(2):
        p = str(tmp_path / ('t%d.sdfs' % i))
        synth.write_chunk(p, (32, 32, 32), 900 + i, occupancy=0.08)
        files.append(p)
    batch = next(iter(data.DeviceBatchLoader(files, 2, TRUNC)))
    torch.manual_seed(0)
    net = M.GenModel(8, (32, 32, 32), 1, 16, 16, 4, True, True, 1, 1).cuda()
    opt = train.make_optimizer(net.parameters())
    loss, losses, _ = train.train_step(net, opt, batch, np.ones(5, dtype=np.float32))
    assert np.isfinite(loss.item())


def test_mixed_chunk_sizes_rejected(tmp_path):
    a, b = str(tmp_path / 'a.sdfs'), str(tmp_path / 'b.sdfs')
    synth.write_chunk(a, (16, 16, 16), 1, occupancy=0.2)
    synth.write_chunk(b, (16, 16, 32), 2, occupancy=0.2)
    with pytest.raises(ValueError, match='dimensions'):
        list(data.DeviceBatchLoader([a, b], 2, TRUNC))
