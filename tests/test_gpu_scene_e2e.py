"""GPU: the chain of test_scene.py:66-104 on this build's pieces — scene files (.sdf/.sdf/.knw) -> DeviceBatchLoader
(scene mode) -> GenModel.update_sizes + eval forward -> padding removal -> save_predictions meshes — runs end to end
and its intermediate results agree with the host-side pieces."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))
import data_oracle  # noqa: E402

from sgnn_amd import data, marching_cubes as mc, synth  # noqa: E402
from sgnn_amd.model import GenModel  # noqa: E402

pytestmark = pytest.mark.gpu


def test_scene_files_to_meshes(tmp_path):
    dims = (40, 72, 88)                                   # not multiples of 32: exercises the padding
    ind, tgd = tmp_path / 'in', tmp_path / 'tgt'
    ind.mkdir()
    tgd.mkdir()
    s_in, s_tgt = str(ind / 'scene7.sdf'), str(tgd / 'scene7.sdf')
    synth.write_scene_triple(s_in, s_tgt, dims, 5, occupancy=0.2, stored_band=2.9, voxelsize=0.02)
    sample = next(iter(data.DeviceBatchLoader([s_in], 1, 3.0, max_input_height=128, target_path=str(tgd))))
    ref = data_oracle.collate([data_oracle.sample_scene(s_in, s_tgt, 3.0, 4, 128)])
    assert np.array_equal(sample['input'][0].cpu().numpy(), ref['input'][0])
    assert tuple(sample['sdf'].shape[2:]) == (64, 96, 96)

    torch.manual_seed(0)
    model = GenModel(8, (64, 64, 64), 1, 16, 16, 4, True, True, 1, 1).cuda()
    input_dim = np.array(sample['sdf'].shape[2:])
    model.update_sizes(input_dim, input_dim // 8)          # test_scene.py:77-78
    lw = np.ones(5, dtype=np.float32)
    with torch.no_grad():
        model.train()                                       # random init: batch statistics, so levels are populated
        output_sdf, output_occs = model(sample['input'], lw)
    assert len(output_sdf[0]) > 0
    od = sample['orig_dims'][0]
    keep = (output_sdf[0][:, 0] < od[0]) & (output_sdf[0][:, 1] < od[1]) & (output_sdf[0][:, 2] < od[2])   # :89-91
    pred = [[output_sdf[0][keep].cpu().numpy(), output_sdf[1][keep].squeeze(1).cpu().numpy()]]
    inputs = [sample['input'][0].cpu().numpy(), sample['input'][1].cpu().numpy()]
    out = tmp_path / 'vis'
    mc.save_predictions(str(out), sample['name'], inputs, None, None, pred, None, sample['world2grid'], 3.0)
    files = sorted(os.listdir(out))
    assert files == ['scene7input-mesh.ply', 'scene7pred-mesh.ply']
    head = open(out / 'scene7input-mesh.ply', 'rb').read(200).decode('ascii', 'ignore')
    nv = int(head.split('element vertex ')[1].split('\n')[0])
    nf = int(head.split('element face ')[1].split('\n')[0])
    assert nv > 500 and nf > 500                            # the input surface is meshed


def test_save_predictions_writes_level_point_clouds(tmp_path):
    """train.py's visualisation call passes the per-level occupancies (data_util.py:264-270): the level point clouds
    must be written next to the meshes instead of aborting."""
    batch = synth.make_batch(1, (32, 32, 32), cfg=3, occupancy=0.1)
    from sgnn_amd import loss as L
    t = L.compute_targets(batch['sdf'].clone(), [h.clone() for h in batch['hierarchy']], 4, 3.0, True, batch['known'])
    locs = batch['input'][0].numpy()
    pred_locs = [[locs[::7][:, :3] // f] for f in (8, 4, 2, 1)]
    pred_sdf = [[locs[:, :3], batch['input'][1].numpy()[:, 0]]]
    out = tmp_path / 'vis'
    mc.save_predictions(str(out), ['b0'], [locs, batch['input'][1].numpy()], t[0].numpy(), [o.numpy() for o in t[1]],
                        pred_sdf, pred_locs, None, 3.0)
    files = sorted(os.listdir(out))
    assert files == sorted(['b0input-mesh.ply', 'b0pred-mesh.ply', 'b0target-mesh.ply'] +
                           ['b0pred-%d.ply' % h for h in range(4)] + ['b0target-%d.ply' % h for h in range(4)])
    raw = open(out / 'b0pred-3.ply', 'rb').read()
    n = int(raw.split(b'element vertex ')[1].split(b'\n')[0])
    pts = np.frombuffer(raw.split(b'end_header\n')[1], dtype='<f4').reshape(-1, 3)
    assert n == pts.shape[0] == pred_locs[3][0].shape[0]
    assert np.array_equal(pts, pred_locs[3][0][:, ::-1].astype(np.float32) + 0.5)      # x,y,z voxel centres, factor 1
    raw0 = open(out / 'b0target-0.ply', 'rb').read()
    pts0 = np.frombuffer(raw0.split(b'end_header\n')[1], dtype='<f4').reshape(-1, 3)
    assert (pts0 % 8 == 4).all()                                                       # level-0 centres scaled by 8
