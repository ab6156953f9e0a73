"""Seeded TSDF volumes for the marching-cubes tests (shared by tests/golden/make_golden_mc.py and the tests)."""
import numpy as np
import torch

from sgnn_amd import synth

CASES = {
    # name: dims (z,y,x), seed, occupancy, iso, trunc, thresh, flavour
    'sphere32': dict(dims=(32, 32, 32), seed=3, occ=0.08, iso=0.0, trunc=3.0, thresh=10.0, kind='block'),
    'rect': dict(dims=(24, 40, 56), seed=4, occ=0.10, iso=0.0, trunc=3.0, thresh=10.0, kind='block'),
    'iso1': dict(dims=(32, 32, 32), seed=5, occ=0.08, iso=1.0, trunc=3.0, thresh=10.0, kind='block'),   # --vis_dfs path
    'jumps': dict(dims=(32, 32, 32), seed=6, occ=0.08, iso=0.0, trunc=3.0, thresh=1.1, kind='noisy'),   # threshold rejections
    'snap': dict(dims=(24, 24, 24), seed=7, occ=0.10, iso=0.0, trunc=3.0, thresh=10.0, kind='quantised'),  # exact hits, near-coincident vertices
    'colors': dict(dims=(24, 24, 24), seed=8, occ=0.10, iso=0.0, trunc=3.0, thresh=10.0, kind='block', colors=True),
    'dense': dict(dims=(20, 20, 20), seed=9, occ=0.3, iso=0.0, trunc=100.0, thresh=200.0, kind='full'),   # every voxel valid, border cubes
    'empty': dict(dims=(16, 16, 16), seed=1, occ=0.05, iso=0.0, trunc=3.0, thresh=10.0, kind='empty'),
    'block64': dict(dims=(64, 64, 64), seed=5, occ=0.05, iso=0.0, trunc=3.0, thresh=10.0, kind='block'),
}


def make_volume(spec):
    """-> (tsdf (z,y,x) float32 CPU tensor with -inf where nothing is stored, colors (z,y,x,3) uint8 or None)."""
    dims, rng = spec['dims'], np.random.default_rng(1000 + spec['seed'])
    sdf = synth._block_sdf(dims, np.random.default_rng(spec['seed']), spec['occ']).astype(np.float32)
    kind = spec['kind']
    if kind == 'full':
        vol = sdf
    elif kind == 'empty':
        vol = np.full(dims, -np.inf, dtype=np.float32)
    else:
        if kind == 'noisy':
            sdf = sdf + rng.normal(0, 0.6, dims).astype(np.float32)
        if kind == 'quantised':
            sdf = np.round(sdf * 2) / 2          # many corner averages land exactly on the iso value or within 1e-5
            sdf = sdf + (rng.random(dims) < 0.05) * np.float32(4e-6)
        vol = np.where(np.abs(sdf) < 4.0, sdf, -np.inf).astype(np.float32)   # band wider than the truncation
    colors = None
    if spec.get('colors'):
        colors = torch.from_numpy(rng.integers(0, 256, dims + (3,), dtype=np.uint8))
    return torch.from_numpy(np.ascontiguousarray(vol, dtype=np.float32)), colors


def scene_prediction():
    """Inputs of a test_scene.py:96-98 style save_predictions call: one scene's input sites [z,y,x,b] + tsdf values and
    a sparse sdf prediction (coordinates + values), numpy on the host as the reference passes them."""
    a = synth.block_arrays((40, 48, 56), 33, occupancy=0.12, stored_band=2.85, voxelsize=1.0)
    il, iv = a['input']
    tl, tv = a['target']
    rng = np.random.default_rng(77)
    keep = rng.random(len(tl)) < 0.93
    inputs = [np.concatenate([il, np.zeros((len(il), 1), np.int64)], 1), iv[:, None].astype(np.float32)]
    pred = [[np.concatenate([tl[keep], np.zeros((int(keep.sum()), 1), np.int64)], 1),
             (tv[keep] + rng.normal(0, 0.05, int(keep.sum()))).astype(np.float32)]]
    return ['scene0_'], inputs, pred
