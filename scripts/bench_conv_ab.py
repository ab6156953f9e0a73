"""A/B of the large-level forward convolution (needs GPU): looped kernel (conv.hip k_conv_fwd) vs straight-line kernel
(conv_unrolled.hip k_conv_fwd_u) on the same level, same inputs — results must be bit-identical.
  python scripts/bench_conv_ab.py [--batch 32] [--dim 64] [--iters 30]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgnn_amd import synth, _lib
from sgnn_amd.scn import functions as F_
from sgnn_amd.scn.metadata import Grid, coords_from_locs, build_down2

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--dim', type=int, default=64)
ap.add_argument('--iters', type=int, default=30)
args = ap.parse_args()
dev = torch.device('cuda')
lib = _lib.load()
data = synth.make_batch(args.batch, (args.dim,) * 3, cfg=2, occupancy=0.05)
g = Grid(coords_from_locs(data['input'][0], dev))
tab = g.subm_table()
rules = int((tab.view(27, g.ld)[:, :g.n] >= 0).sum().item())
print('sites %d  rules %d (R/N %.2f)' % (g.n, rules, rules / g.n))


def timeit(fn, iters=args.iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


torch.manual_seed(0)
for cin, cout in ((16, 16), (8, 8), (12, 12), (26, 16), (16, 26), (48, 16), (16, 48), (1, 8)):
    x = torch.randn(g.n, cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.1
    outs = {}
    for flags in (0, 3):     # forward; data gradient (transposed weights, flipped offsets)
        for on in (0, 1):
            _lib.tune('conv_unrolled', on)
            ww = w if flags == 0 else w.transpose(1, 2).contiguous()   # (K, Cout, Cin) of the forward weights
            if flags and cin != cout:
                continue
            y = F_.conv_fwd_raw(x, cin, ww, 27, tab, g.ld, g.n, cout, flags, 0)
            us = timeit(lambda: F_.conv_fwd_raw(x, cin, ww, 27, tab, g.ld, g.n, cout, flags, 0))
            outs[on] = y.clone()
            fl = 2.0 * rules * cin * cout
            print('conv_fwd<%d,%d> K27 flags %d  %-16s %7.1f us   %5.1f TFLOP/s rules-only = %.3f of fp32 MFMA peak'
                  % (cin, cout, flags, 'unrolled kernel' if on else 'looped kernel', us, fl / us / 1e6, fl / us / 1e6 / 157.3))
        same = torch.equal(outs[0], outs[1])
        print('   bit-identical: %s   max |diff| %.3e' % (same, float((outs[0] - outs[1]).abs().max())))
        assert same
# stride-2 table (8 offsets): forward = children table, data gradient = parent table
d = build_down2(g)
for cin, cout in ((8, 12), (16, 16)):
    x = torch.randn(g.n, cin, device=dev)
    w = torch.randn(8, cin, cout, device=dev) * 0.1
    outs = {}
    for on in (0, 1):
        _lib.tune('conv_unrolled', on)
        y = F_.conv_fwd_raw(x, cin, w, 8, d.children, d.ldc, d.coarse.n, cout, 0, 0)
        us = timeit(lambda: F_.conv_fwd_raw(x, cin, w, 8, d.children, d.ldc, d.coarse.n, cout, 0, 0))
        outs[on] = y.clone()
        print('conv_down<%d,%d> K8 (%d -> %d rows)  %-16s %7.1f us' % (cin, cout, g.n, d.coarse.n, 'unrolled kernel' if on else 'looped kernel', us))
    assert torch.equal(outs[0], outs[1]), float((outs[0] - outs[1]).abs().max())
    print('   bit-identical: True')
_lib.tune('conv_unrolled', 1)
