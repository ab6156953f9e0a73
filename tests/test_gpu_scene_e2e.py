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
