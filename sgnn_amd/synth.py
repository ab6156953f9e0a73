"""Synthetic TSDF blocks in exactly the layout `scene_dataloader.collate` emits
(torch/scene_dataloader.py:13-36, 60-116; file semantics torch/data_util.py:63-108):

  input      [locs (sumN,4) int64 [z,y,x,b] batch-major raster order, feats (sumN,1) f32 TSDF in voxels]
  sdf        (B,1,D0,D1,D2) f32, -inf where no target surface was stored
  known      (B,1,D0,D1,D2) u8: 0 known-empty, 1 known-occupied, >=2 unknown (depth behind surface + 1)
  hierarchy  [3] dense (B,1,D/8..), (B,1,D/4..), (B,1,D/2..) coarse -> fine, -inf off-surface

Primary distribution "surface" (SURVEY.md §8d): per block a union of 1..3 random spheres plus
optionally a plane, signed distance in voxels, active iff |sdf| < truncation; the input is the
target with a random half-space removed (self-supervised completion flavour).  Seeds:
numpy default_rng(1000*cfg + block_idx).  Secondary "iid": Bernoulli(p) sites, sdf ~ U(-3,3).
There is no network in the build environment, so real Matterport chunks are never used.
"""
import numpy as np
import torch


def _block_sdf(dims, rng, occupancy):
    zz, yy, xx = np.meshgrid(np.arange(dims[0]), np.arange(dims[1]), np.arange(dims[2]), indexing='ij')
    pts = np.stack([zz, yy, xx], -1).astype(np.float32)
    dmin = float(min(dims))
    # a single sphere of r ~ 0.206*D gives ~5 % of |sdf|<3 sites at 64^3; scale r with the requested occupancy
    base_r = 0.206 * dmin * np.sqrt(max(occupancy, 1e-3) / 0.05)
    k = int(rng.integers(1, 4))
    sdf = np.full(dims, 1e6, dtype=np.float32)
    for _ in range(k):
        c = rng.uniform(0.25, 0.75, 3) * np.array(dims)
        r = base_r / np.sqrt(k) * rng.uniform(0.85, 1.15)
        d = np.sqrt(((pts - c) ** 2).sum(-1)) - r
        sdf = np.minimum(sdf, d)
    return sdf


def make_block(dims, seed, occupancy=0.05, truncation=3.0, dist='surface'):
    """One block -> (input_locs (N,3) int64 zyx, input_sdf (N,), target dense, known dense, hierarchy[3])."""
    rng = np.random.default_rng(seed)
    dims = tuple(int(d) for d in dims)
    if dist == 'iid':
        occ = rng.random(dims) < occupancy
        sdf = np.where(occ, rng.uniform(-truncation, truncation, dims), 1e6).astype(np.float32)
    else:
        sdf = _block_sdf(dims, rng, occupancy)
    band = np.abs(sdf) < truncation
    target = np.where(band, sdf, -np.inf).astype(np.float32)
    known = np.where(sdf > 0, 0, np.where(band, 1, np.minimum(255, np.ceil(-sdf) + 1))).astype(np.uint8)
    # input: drop everything behind a random plane through the block
    nrm = rng.normal(size=3)
    nrm /= np.linalg.norm(nrm)
    zz, yy, xx = np.meshgrid(np.arange(dims[0]), np.arange(dims[1]), np.arange(dims[2]), indexing='ij')
    side = ((zz - dims[0] / 2) * nrm[0] + (yy - dims[1] / 2) * nrm[1] + (xx - dims[2] / 2) * nrm[2]) < 0.25 * min(dims)
    keep = band & side
    z, y, x = np.nonzero(keep)
    in_locs = np.stack([z, y, x], 1).astype(np.int64)
    in_sdf = sdf[z, y, x].astype(np.float32)
    hierarchy = []
    for f in (8, 4, 2):  # coarse -> fine, values in that level's voxel units
        sub = sdf[f // 2::f, f // 2::f, f // 2::f] / f
        hierarchy.append(np.where(np.abs(sub) < truncation, sub, -np.inf).astype(np.float32))
    return in_locs, in_sdf, target, known, hierarchy


def make_batch(batch_size, dims=(64, 64, 64), cfg=2, first_block=0, occupancy=0.05, truncation=3.0, dist='surface'):
    """Collated batch (CPU tensors), same keys/dtypes as scene_dataloader.collate."""
    if not hasattr(dims, '__len__'):
        dims = (dims, dims, dims)
    locs, feats, sdfs, knowns, hier = [], [], [], [], [[], [], []]
    for b in range(batch_size):
        il, isdf, tgt, knw, hr = make_block(dims, 1000 * cfg + first_block + b, occupancy, truncation, dist)
        locs.append(np.concatenate([il, np.full((il.shape[0], 1), b, np.int64)], 1))
        feats.append(isdf[:, None])
        sdfs.append(tgt[None])
        knowns.append(knw[None])
        for h in range(3):
            hier[h].append(hr[h][None])
    return {
        'input': [torch.from_numpy(np.concatenate(locs, 0)), torch.from_numpy(np.concatenate(feats, 0))],
        'sdf': torch.from_numpy(np.stack(sdfs, 0)),
        'known': torch.from_numpy(np.stack(knowns, 0)),
        'hierarchy': [torch.from_numpy(np.stack(h, 0)) for h in hier],
    }


def _sparse_of(dense_vox, band, voxelsize):
    z, y, x = np.nonzero(band)
    return np.stack([z, y, x], 1).astype(np.int64), (dense_vox[z, y, x] * np.float32(voxelsize)).astype(np.float32)


def block_arrays(dims, seed, occupancy=0.05, stored_band=5.0, voxelsize=0.02):
    """Everything one .sdfs chunk stores (data_util.py:63-117), values in metres, bands wider than the training
    truncation so the loader's |sdf| < truncation mask has something to remove."""
    rng = np.random.default_rng(seed)
    dims = tuple(int(d) for d in dims)
    sdf = _block_sdf(dims, rng, occupancy)
    band = np.abs(sdf) < stored_band
    known = np.where(sdf > 0, 0, np.where(np.abs(sdf) < 3.0, 1, np.minimum(255, np.ceil(-sdf) + 1))).astype(np.uint8)
    nrm = rng.normal(size=3)
    nrm /= np.linalg.norm(nrm)
    zz, yy, xx = np.meshgrid(np.arange(dims[0]), np.arange(dims[1]), np.arange(dims[2]), indexing='ij')
    side = ((zz - dims[0] / 2) * nrm[0] + (yy - dims[1] / 2) * nrm[1] + (xx - dims[2] / 2) * nrm[2]) < 0.25 * min(dims)
    hier = []
    for f in (2, 4, 8):                              # file order: 1/2, 1/4, 1/8
        sub = (sdf[f // 2::f, f // 2::f, f // 2::f] / f)[:dims[0] // f, :dims[1] // f, :dims[2] // f]
        hier.append(_sparse_of(sub, np.abs(sub) < stored_band, voxelsize))
    world2grid = np.eye(4, dtype=np.float32)
    world2grid[:3, 3] = rng.uniform(-4, 4, 3).astype(np.float32)
    world2grid[:3, :3] /= np.float32(voxelsize)
    return {'dims': dims, 'voxelsize': np.float32(voxelsize), 'world2grid': world2grid,
            'input': _sparse_of(sdf, band & side, voxelsize), 'target': _sparse_of(sdf, band, voxelsize),
            'known': known, 'hierarchy': hier}


def write_chunk(path, dims, seed, **kw):
    """Write one synthetic .sdfs training chunk (see sgnn_amd/data.py for the layout)."""
    from . import data
    a = block_arrays(dims, seed, **kw)
    data.write_train_file(path, a['dims'], a['voxelsize'], a['world2grid'], a['input'], a['target'], a['known'],
                          a['hierarchy'])
    return a


def write_scene_triple(in_path, tgt_path, dims, seed, **kw):
    """Synthetic whole-scene input .sdf, target .sdf and its .knw (test_scene.py:60-70 data layout)."""
    import os
    from . import data
    a = block_arrays(dims, seed, **kw)
    data.write_scene(in_path, a['dims'], a['voxelsize'], a['world2grid'], a['input'])
    data.write_scene(tgt_path, a['dims'], a['voxelsize'], a['world2grid'], a['target'])
    data.write_known(os.path.splitext(tgt_path)[0] + '.knw', a['dims'], a['voxelsize'], a['world2grid'], a['known'])
    return a


def make_scene(dims=(128, 512, 512), cfg=4, occupancy=0.05, truncation=3.0, tile=64):
    """One whole-scene input [locs (N,4) int64 [z,y,x,0], feats (N,1)] for BASELINE configs[3]: the volume is tiled
    with independent synthetic surface blocks (make_block), ~N = 13 k sites per 64^3 tile -> ~1.7 M at (128,512,512)."""
    dims = tuple(int(d) for d in dims)
    locs, feats, t = [], [], 0
    for z0 in range(0, dims[0], tile):
        for y0 in range(0, dims[1], tile):
            for x0 in range(0, dims[2], tile):
                td = (min(tile, dims[0] - z0), min(tile, dims[1] - y0), min(tile, dims[2] - x0))
                il, isdf, _, _, _ = make_block(td, 1000 * cfg + t, occupancy, truncation)
                il = il + np.array([z0, y0, x0], dtype=np.int64)
                locs.append(np.concatenate([il, np.zeros((len(il), 1), np.int64)], 1))
                feats.append(isdf[:, None])
                t += 1
    locs, feats = np.concatenate(locs), np.concatenate(feats)
    order = np.lexsort((locs[:, 2], locs[:, 1], locs[:, 0]))          # file order: z-major raster (VoxelGrid.h:133-143)
    return [torch.from_numpy(locs[order]), torch.from_numpy(feats[order].astype(np.float32))]
