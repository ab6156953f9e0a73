"""TEST INFRASTRUCTURE — CPU restatement of the reference's data-file readers and batch assembly
(SURVEY.md §8 row f2).  Only tests/, __graft_entry__.smoke() and bench legs marked cpu_baseline may import this.

Follows, function by function:
  read_header/read_block   torch/data_util.py:64-79 (struct.unpack of the header, counts, triples, values)
  load_train_file          torch/data_util.py:63-117
  load_scene               torch/data_util.py:121-139
  load_scene_known         torch/data_util.py:142-155
  sample_chunk/sample_scene  torch/scene_dataloader.py:60-116 (SceneDataset.__getitem__)
  collate                  torch/scene_dataloader.py:13-36

Parity status: PINNED — tests/test_oracle_data.py checks every function here against
tests/golden/data_expected.npz, which tests/golden/make_golden_data.py produced by running the reference's own
data_util.py / scene_dataloader.py on the fixture files under tests/golden/data/.
Like the reference it decodes scalar by scalar with `struct` (that cost is what the cpu_baseline leg times).
"""
import os
import struct

import numpy as np


def read_header(f):
    dimx, dimy, dimz = struct.unpack('<QQQ', f.read(24))
    voxelsize = struct.unpack('<f', f.read(4))[0]
    world2grid = np.asarray(struct.unpack('<16f', f.read(64)), dtype=np.float32).reshape(4, 4)
    return dimx, dimy, dimz, voxelsize, world2grid


def read_block(f, voxelsize):
    num = struct.unpack('<Q', f.read(8))[0]
    locs = np.asarray(struct.unpack('<%dI' % (3 * num), f.read(12 * num)), dtype=np.int32).reshape(num, 3)
    locs = np.flip(locs, 1).copy()                                   # x,y,z on disk -> z,y,x
    vals = np.asarray(struct.unpack('<%df' % num, f.read(4 * num)), dtype=np.float32)
    vals /= voxelsize                                                # float32 / python float -> float32
    return locs, vals


def to_dense(locs, vals, dimx, dimy, dimz, fill):
    dense = np.full((dimz, dimy, dimx), fill, dtype=vals.dtype)
    dense[locs[:, 0], locs[:, 1], locs[:, 2]] = vals
    return dense


def load_train_file(path):
    with open(path, 'rb') as f:
        dimx, dimy, dimz, vs, w2g = read_header(f)
        inputs = read_block(f, vs)
        tl, tv = read_block(f, vs)
        target = to_dense(tl, tv, dimx, dimy, dimz, -np.inf)
        num = struct.unpack('<Q', f.read(8))[0]
        assert num == dimx * dimy * dimz
        known = np.frombuffer(f.read(num), dtype=np.uint8).reshape(dimz, dimy, dimx).copy()
        hierarchy, factor = [], 2
        for _ in range(3):
            hl, hv = read_block(f, vs)
            hierarchy.append(to_dense(hl, hv, dimx // factor, dimy // factor, dimz // factor, -np.inf))
            factor *= 2
        hierarchy.reverse()
    return list(inputs), target, [dimz, dimy, dimx], w2g, known, hierarchy


def load_scene(path):
    with open(path, 'rb') as f:
        dimx, dimy, dimz, vs, w2g = read_header(f)
        locs, vals = read_block(f, vs)
    return [locs, vals], [dimz, dimy, dimx], w2g


def load_scene_known(path):
    with open(path, 'rb') as f:
        dimx, dimy, dimz, _, _ = read_header(f)
        return np.frombuffer(f.read(dimx * dimy * dimz), dtype=np.uint8).reshape(dimz, dimy, dimx).copy()


def _finish(name, inputs, targets, w2g, known, hierarchy, orig_dims, truncation):
    keep = np.abs(inputs[1]) < truncation
    return {'name': name, 'input': [inputs[0][keep].astype(np.int64), inputs[1][keep][:, None]],
            'sdf': targets[None], 'world2grid': w2g, 'known': known[None],
            'hierarchy': None if hierarchy is None else [g[None] for g in hierarchy],
            'orig_dims': np.array(orig_dims, dtype=np.int64)}


def sample_chunk(path, truncation, num_hierarchy_levels):
    inputs, targets, dims, w2g, known, hierarchy = load_train_file(path)
    if num_hierarchy_levels < 4:
        hierarchy = hierarchy[4 - num_hierarchy_levels:]
    return _finish(os.path.splitext(os.path.basename(path))[0], inputs, targets, w2g, known, hierarchy,
                   targets.shape, truncation)


def sample_scene(in_path, tgt_path, truncation, num_hierarchy_levels, max_input_height):
    inputs, _, _ = load_scene(in_path)
    tgt, dims, w2g = load_scene(tgt_path)
    known = load_scene_known(os.path.splitext(tgt_path)[0] + '.knw')
    targets = to_dense(tgt[0], tgt[1], dims[2], dims[1], dims[0], -np.inf)
    orig = targets.shape
    h = max_input_height
    q = 4 * 2 ** (num_hierarchy_levels - 1)
    pd = np.array(targets.shape)
    if h > 0 and pd[0] > h:
        pd[0] = h
        keep = inputs[0][:, 0] < h
        inputs = [inputs[0][keep], inputs[1][keep]]
    pd = ((pd + q - 1) // q) * q
    padded = np.full(tuple(pd), -np.inf, dtype=np.float32)
    padded[:min(h, targets.shape[0]), :targets.shape[1], :targets.shape[2]] = targets[:h]
    kpad = np.full(tuple(pd), 255, dtype=np.uint8)
    kpad[:min(h, known.shape[0]), :known.shape[1], :known.shape[2]] = known[:h]
    return _finish(os.path.splitext(os.path.basename(in_path))[0], inputs, padded, w2g, kpad, None, orig, truncation)


def collate(samples):
    locs = np.concatenate([np.concatenate([s['input'][0], np.full((len(s['input'][0]), 1), b, np.int64)], 1)
                           for b, s in enumerate(samples)])
    out = {'name': [s['name'] for s in samples], 'input': [locs, np.concatenate([s['input'][1] for s in samples])],
           'sdf': np.stack([s['sdf'] for s in samples]), 'world2grid': np.stack([s['world2grid'] for s in samples]),
           'known': np.stack([s['known'] for s in samples]), 'orig_dims': np.stack([s['orig_dims'] for s in samples]),
           'hierarchy': None}
    if samples[0]['hierarchy'] is not None:
        out['hierarchy'] = [np.stack([s['hierarchy'][h] for s in samples]) for h in range(len(samples[0]['hierarchy']))]
    return out
