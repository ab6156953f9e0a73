"""Generates the data-file fixtures and their expected decodes by running the REAL
reference loader — /root/reference/torch/data_util.py (load_train_file, load_scene, load_scene_known) and
/root/reference/torch/scene_dataloader.py (SceneDataset, collate), unmodified — on files written by this build's
writer.  Authoring container only; the fixtures (tests/golden/data/*, data_expected.npz) travel, the reference
does not.

Substituted at import time: `plyfile`, `marching_cubes` (+ `.marching_cubes`), `marching_cubes_cpp` := empty
stubs (visualisation only; data_util.py:7,9, scene_dataloader.py:8,11).

Usage:  python tests/golden/make_golden_data.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

for name in ('plyfile', 'marching_cubes_cpp'):
    sys.modules[name] = types.ModuleType(name)
pkg = types.ModuleType('marching_cubes')
pkg.marching_cubes = types.ModuleType('marching_cubes.marching_cubes')
sys.modules['marching_cubes'] = pkg
sys.modules['marching_cubes.marching_cubes'] = pkg.marching_cubes
sys.path.insert(0, '/root/reference/torch')
import data_util as ref_data        # noqa: E402
import scene_dataloader as ref_dl   # noqa: E402

from sgnn_amd import synth          # noqa: E402

DATA = os.path.join(HERE, 'data')
CHUNK_DIMS = (16, 16, 16)
SCENE_DIMS = (21, 18, 27)           # z,y,x: odd volume, so sections after the u8 volume are unaligned
TRUNC = 3.0


def np_(t):
    return t.numpy() if hasattr(t, 'numpy') else np.asarray(t)


def main():
    os.makedirs(os.path.join(DATA, 'scene_in'), exist_ok=True)
    os.makedirs(os.path.join(DATA, 'scene_tgt'), exist_ok=True)
    chunks = []
    for i, (seed, vs) in enumerate([(11, 0.02), (12, 0.02), (13, 0.046875)]):
        p = os.path.join(DATA, 'chunk_%d.sdfs' % i)
        synth.write_chunk(p, CHUNK_DIMS, seed, occupancy=0.25, voxelsize=vs)
        chunks.append(p)
    s_in, s_tgt = os.path.join(DATA, 'scene_in', 'scene0.sdf'), os.path.join(DATA, 'scene_tgt', 'scene0.sdf')
    synth.write_scene_triple(s_in, s_tgt, SCENE_DIMS, 21, occupancy=0.3, voxelsize=0.03)

    out = {}
    for i, p in enumerate(chunks):           # data_util.load_train_file, field by field
        (il, iv), tgt, dims, w2g, known, hier = ref_data.load_train_file(p)
        out.update({'c%d_in_locs' % i: il, 'c%d_in_vals' % i: iv, 'c%d_target' % i: tgt, 'c%d_dims' % i: np.array(dims),
                    'c%d_w2g' % i: w2g, 'c%d_known' % i: known})
        for h in range(3):
            out['c%d_hier%d' % (i, h)] = hier[h]
    (sl, sv), sdims, sw2g = ref_data.load_scene(s_in)
    out.update({'s_in_locs': sl, 's_in_vals': sv, 's_dims': np.array(sdims), 's_w2g': sw2g,
                's_known': ref_data.load_scene_known(os.path.splitext(s_tgt)[0] + '.knw')})

    for levels in (4, 3):                    # SceneDataset + collate on the chunks
        ds = ref_dl.SceneDataset(chunks, 16, TRUNC, levels, 0)
        b = ref_dl.collate([ds[i] for i in range(len(ds))])
        k = 'b%d_' % levels
        out.update({k + 'locs': np_(b['input'][0]), k + 'feats': np_(b['input'][1]), k + 'sdf': np_(b['sdf']),
                    k + 'known': np_(b['known']), k + 'w2g': np_(b['world2grid']), k + 'orig_dims': np_(b['orig_dims']),
                    k + 'names': np.array(b['name'])})
        for h, g in enumerate(b['hierarchy']):
            out[k + 'hier%d' % h] = np_(g)
    for height in (16, 0, 128):              # whole-scene mode with the up-axis clamp (taken / degenerate / not taken)
        ds = ref_dl.SceneDataset([s_in], 0, TRUNC, 4, height, target_path=os.path.join(DATA, 'scene_tgt'))
        b = ref_dl.collate([ds[0]])
        k = 's%d_' % height
        out.update({k + 'locs': np_(b['input'][0]), k + 'feats': np_(b['input'][1]), k + 'sdf': np_(b['sdf']),
                    k + 'known': np_(b['known']), k + 'orig_dims': np_(b['orig_dims'])})
        assert b['hierarchy'] is None
    np.savez_compressed(os.path.join(HERE, 'data_expected.npz'), **out)
    print('wrote', len(out), 'arrays;', sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(DATA)
                                           for f in fs), 'fixture bytes')


if __name__ == '__main__':
    main()
