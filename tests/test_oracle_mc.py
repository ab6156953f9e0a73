"""CPU: the marching-cubes restatement (and, when built, the reference's own extension) against the stored
reference outputs."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import mc_oracle  # noqa: E402
from mc_cases import CASES, make_volume  # noqa: E402

SMALL = ['snap', 'jumps', 'colors', 'dense', 'empty', 'iso1']


@pytest.fixture(scope='module')
def g():
    return np.load(os.path.join(HERE, 'golden', 'mc_expected.npz'))


@pytest.mark.parametrize('name', SMALL)
def test_restatement_matches_reference(g, name):
    spec = CASES[name]
    tsdf, colors = make_volume(spec)
    v, c, f = mc_oracle.run_marching_cubes(tsdf.numpy(), None if colors is None else colors.numpy(), spec['iso'],
                                           spec['trunc'], spec['thresh'])
    assert np.array_equal(v.view(np.int32), g[name + '_v'].view(np.int32))
    assert np.array_equal(c, g[name + '_c']) and np.array_equal(f, g[name + '_f'])


def test_reference_module_reproduces_the_fixture(g):
    sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))
    try:
        import marching_cubes_cpp as ref
    except ImportError:
        pytest.skip('oracle/_ref not built (make -C oracle ref needs /root/reference)')
    for name, spec in CASES.items():
        tsdf, colors = make_volume(spec)
        col = colors if colors is not None else torch.ones(tuple(tsdf.shape) + (3,), dtype=torch.uint8) * 220
        v, c, f = ref.run_marching_cubes(tsdf, col, spec['iso'], spec['trunc'], spec['thresh'])
        assert np.array_equal(v.numpy(), g[name + '_v']) and np.array_equal(f.numpy(), g[name + '_f'])


@pytest.mark.parametrize('seed', range(6))
def test_restatement_matches_live_reference_on_random_volumes(seed):
    """Random small volumes (noisy, partially invalid, random iso / thresholds): the restatement against the
    reference's own extension run live."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle', '_ref'))
    try:
        import marching_cubes_cpp as ref
    except ImportError:
        pytest.skip('oracle/_ref not built (make -C oracle ref needs /root/reference)')
    rng = np.random.default_rng(100 + seed)
    dims = tuple(int(v) for v in rng.integers(7, 13, 3))
    zz, yy, xx = np.meshgrid(*[np.arange(d) for d in dims], indexing='ij')
    c = rng.uniform(0.3, 0.7, 3) * np.array(dims)
    sdf = np.sqrt((zz - c[0]) ** 2 + (yy - c[1]) ** 2 + (xx - c[2]) ** 2) - rng.uniform(2.0, 4.0)
    sdf = sdf + rng.normal(0, rng.choice([0.0, 0.2, 0.8]), dims)
    if seed % 2:
        sdf = np.round(sdf * 4) / 4                                       # exact iso hits and ties
    vol = np.where(rng.random(dims) < 0.9, sdf, -np.inf).astype(np.float32)   # holes
    iso = float(rng.choice([0.0, 0.25, 1.0]))
    trunc = float(rng.choice([3.0, 2.5, 50.0]))
    thresh = float(rng.choice([10.0, 1.2]))
    colors = rng.integers(0, 256, dims + (3,), dtype=np.uint8)
    rv, rc, rf = ref.run_marching_cubes(torch.from_numpy(vol), torch.from_numpy(colors), iso, trunc, thresh)
    v, c_, f = mc_oracle.run_marching_cubes(vol, colors, iso, trunc, thresh)
    assert np.array_equal(v.view(np.int32), rv.numpy().view(np.int32))
    assert np.array_equal(c_, rc.numpy()) and np.array_equal(f, rf.numpy())
