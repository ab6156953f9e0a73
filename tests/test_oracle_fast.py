"""CPU: the oracle's optional C/OpenMP kernels (oracle/csrc/scn_cpu.c, used by bench.py's cpu_baseline leg only)
against the default torch-op path that all parity tests anchor on."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), 'oracle'))
import scn_oracle  # noqa: E402
from scn_oracle import _fast  # noqa: E402

from util import random_sites  # noqa: E402

pytestmark = pytest.mark.skipif(not _fast.available, reason='oracle/_build/libscn_cpu.so not built (make -C oracle cpu)')


@pytest.fixture(autouse=True)
def _restore():
    yield
    scn_oracle.FAST = False


def test_rulebook_identical():
    locs = random_sites(2, 20, 0.1, 3, True)
    g0 = scn_oracle.Grid(locs.numpy())
    ref = g0.subm_rules(3).copy()
    scn_oracle.FAST = True
    g1 = scn_oracle.Grid(locs.numpy())
    assert np.array_equal(g1.subm_rules(3), ref)


@pytest.mark.parametrize('cin,cout', [(16, 16), (26, 16), (1, 8)])
def test_rule_conv_forward_backward(cin, cout):
    locs = random_sites(2, 16, 0.15, 5, True)
    g = scn_oracle.Grid(locs.numpy())
    pairs = scn_oracle.pairs_from_nbr(g.subm_rules(3))
    torch.manual_seed(1)
    res = []
    for fast in (False, True):
        scn_oracle.FAST = fast
        x = torch.randn(g.n, cin).requires_grad_()
        w = torch.randn(27, cin, cout).requires_grad_()
        torch.manual_seed(1)
        x.data.copy_(torch.randn(g.n, cin))
        w.data.copy_(torch.randn(27, cin, cout))
        y = scn_oracle.rule_conv(x, w, pairs, g.n)
        (y * torch.linspace(-1, 1, cout)).sum().backward()
        res.append((y.detach(), x.grad, w.grad))
    for a, b in zip(*res):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * max(1.0, a.abs().max().item()))


def test_float64_keeps_the_torch_path():
    scn_oracle.FAST = True
    locs = random_sites(1, 8, 0.3, 2, False)
    g = scn_oracle.Grid(locs.numpy())
    pairs = scn_oracle.pairs_from_nbr(g.subm_rules(3))
    y = scn_oracle.rule_conv(torch.randn(g.n, 4, dtype=torch.float64), torch.randn(27, 4, 4, dtype=torch.float64), pairs, g.n)
    assert y.dtype == torch.float64


@pytest.mark.parametrize('kind', ['subm', 'down', 'deconv'])
def test_table_convolutions_equal_the_torch_path(kind):
    """The table-form kernels (one OpenMP region per convolution: what bench.py's cpu_baseline leg times) through the
    oracle's modules, forward and both gradients, against the per-offset torch-op path."""
    locs = random_sites(2, 16, 0.2, 9, True)
    res = []
    for fast in (False, True):
        scn_oracle.FAST = fast
        torch.manual_seed(4)
        x = torch.randn(locs.shape[0], 12).requires_grad_()
        t = scn_oracle.InputLayer(3, [16, 16, 16], mode=0)([locs, x])
        if kind == 'subm':
            m = scn_oracle.SubmanifoldConvolution(3, 12, 16, 3, False)
            y = m(t).features
        elif kind == 'down':
            m = scn_oracle.Convolution(3, 12, 16, 2, 2, False)
            y = m(t).features
        else:
            d = scn_oracle.Convolution(3, 12, 12, 2, 2, False)
            m = scn_oracle.Deconvolution(3, 12, 16, 2, 2, False)
            torch.manual_seed(5)
            with torch.no_grad():
                d.weight.copy_(torch.randn_like(d.weight) * 0.2)
            y = m(d(t)).features
        torch.manual_seed(6)
        with torch.no_grad():
            m.weight.copy_(torch.randn_like(m.weight) * 0.2)
        if kind == 'deconv':
            y = m(d(t)).features
        else:
            y = m(t).features
        (y * torch.linspace(-1, 1, y.shape[1])).sum().backward()
        res.append((y.detach().clone(), x.grad.clone(), m.weight.grad.clone()))
    for a, b in zip(*res):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * max(1.0, a.abs().max().item()))
