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
