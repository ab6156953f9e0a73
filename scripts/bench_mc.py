"""Mesh extraction cost at scene size: device marching cubes + clean-up vs the reference's
own extension (oracle/_ref/marching_cubes_cpp.so, cpu_baseline kind "reference") on the same volume.
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mc_cases import make_volume  # noqa: E402
from sgnn_amd import marching_cubes as mc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--dims', default='128,384,384')
ap.add_argument('--iters', type=int, default=10)
args = ap.parse_args()
dims = tuple(int(v) for v in args.dims.split(','))
tsdf, _ = make_volume(dict(dims=dims, seed=21, occ=0.10, kind='block'))
dev = tsdf.cuda()


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n, out


t_soup, (verts, vcols) = timed(lambda: mc.triangle_soup(dev, None, 0.0, 3.0, 10.0), args.iters)
t_clean, mesh = timed(lambda: mc.clean_mesh(verts, vcols), args.iters)
t_all, _ = timed(lambda: mc.run_marching_cubes(dev, None, 0.0, 3.0, 10.0), args.iters)
vol = int(np.prod(dims))
res = {'workload': 'dense %s tsdf, %d active voxels' % (dims, int(torch.isfinite(tsdf).sum())),
       'triangles_raw': int(verts.shape[0] // 3), 'vertices': int(mesh[0].shape[0]), 'faces': int(mesh[2].shape[0]),
       'soup_ms': round(t_soup * 1e3, 3), 'clean_ms': round(t_clean * 1e3, 3), 'total_ms': round(t_all * 1e3, 3),
       'soup_alg_GBps': round((vol * 4 * 2 + vol * 2 + verts.numel() * 4 + vcols.numel()) / t_soup / 1e9, 1),
       'Mvoxels_per_s': round(vol / t_all / 1e6, 1), 'cpu_baseline': None}
sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))
try:
    import marching_cubes_cpp as ref
    col = torch.ones(dims + (3,), dtype=torch.uint8) * 220
    t0 = time.perf_counter()
    rv, rc, rf = ref.run_marching_cubes(tsdf, col, 0.0, 3.0, 10.0)
    dt = time.perf_counter() - t0
    same = (np.array_equal(rv.numpy(), mesh[0].cpu().numpy()) and np.array_equal(rf.numpy(), mesh[2].cpu().numpy()))
    res['cpu_baseline'] = {'value': round(dt * 1e3, 1), 'unit': 'ms', 'Mvoxels_per_s': round(vol / dt / 1e6, 2),
                           'cores': 1, 'kind': 'reference', 'identical_output': bool(same),
                           'sample': 'the same volume through the reference extension, one run'}
except ImportError:
    pass
print(json.dumps(res))
