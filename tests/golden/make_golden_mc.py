"""Generates tests/golden/mc_expected.npz with the REFERENCE's own marching-cubes extension,
built from its sources by oracle/Makefile into oracle/_ref/marching_cubes_cpp.so (`make -C oracle ref`).
Volumes are regenerated from seeds by the tests (sgnn_amd.synth), so only the reference's outputs are stored.
Authoring container only.

Usage:  make -C oracle ref && python tests/golden/make_golden_mc.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import marching_cubes_cpp as ref  # noqa: E402

from mc_cases import CASES, make_volume  # noqa: E402


def save_predictions_case():
    """The reference's data_util.save_predictions (data_util.py:249-284) exactly as test_scene.py:98 calls it
    (targets and occupancy levels None): its own marching_cubes.py wrapper + the compiled extension; `plyfile` is a
    stub (only the point-cloud branches, not taken here, use it)."""
    import tempfile
    import types
    sys.modules.setdefault('plyfile', types.ModuleType('plyfile'))
    sys.path.insert(0, '/root/reference/torch')
    import data_util as ref_data
    from mc_cases import scene_prediction
    names, inputs, pred = scene_prediction()
    tmp = tempfile.mkdtemp()
    ref_data.save_predictions(tmp, names, inputs, None, None, pred, None, None, 3.0)
    res = {}
    for f in sorted(os.listdir(tmp)):
        res['pred_' + f.replace('-', '_').replace('.', '_')] = np.fromfile(os.path.join(tmp, f), dtype=np.uint8)
    print('save_predictions wrote', sorted(os.listdir(tmp)))
    return res


def main():
    out = {}
    for name, spec in CASES.items():
        tsdf, colors = make_volume(spec)
        col = colors if colors is not None else torch.ones(tuple(tsdf.shape) + (3,), dtype=torch.uint8) * 220
        v, c, f = ref.run_marching_cubes(tsdf, col, spec['iso'], spec['trunc'], spec['thresh'])
        out[name + '_v'], out[name + '_c'], out[name + '_f'] = v.numpy(), c.numpy(), f.numpy()
        print(name, tuple(tsdf.shape), 'verts', len(v), 'faces', len(f))
    p = os.path.join(HERE, '_tmp.ply')
    tsdf, colors = make_volume(CASES['sphere32'])
    ref.export_marching_cubes(tsdf, torch.ones(tuple(tsdf.shape) + (3,), dtype=torch.uint8) * 220, 0.0, 3.0, 10.0, p)
    out['sphere32_ply'] = np.fromfile(p, dtype=np.uint8)
    os.remove(p)
    out.update(save_predictions_case())
    np.savez_compressed(os.path.join(HERE, 'mc_expected.npz'), **out)


if __name__ == '__main__':
    main()
