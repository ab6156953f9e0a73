"""Forward vs data-gradient launches of the large-level convolution, with and without their fused epilogues (needs GPU).
VERDICT r4 item 2: inside the step the dX launches of k_conv_fwd run 25-45 % slower than the forward launches of the same
size; the difference is the fused backward epilogue (BatchNorm input re-read + addend + fp64 statistics per accumulator
element).  This script times, stand-alone on one level and per (cin, cout):
   fwd plain | fwd + statistics (stats = 1) | dX plain | dX + in-place addend + BatchNorm-backward statistics (stats = 2)
and, when the library has the switch, A/Bs the wide (row-contiguous, 16-byte) epilogue against the element-wise one.
  python scripts/bench_epi.py [--batch 32] [--dim 64] [--iters 50]"""
import argparse, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgnn_amd import synth, _lib
from sgnn_amd.scn import functions as F_
from sgnn_amd.scn.metadata import Grid, coords_from_locs

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--dim', type=int, default=64)
ap.add_argument('--iters', type=int, default=50)
args = ap.parse_args()
dev = torch.device('cuda')
lib = _lib.load()
data = synth.make_batch(args.batch, (args.dim,) * 3, cfg=2, occupancy=0.05)
g = Grid(coords_from_locs(data['input'][0], dev))
tab = g.subm_table()
n = g.n
print('sites %d' % n)
has_switch = hasattr(lib, 'sgnn_tune_set')


def timeit(fn, iters=args.iters):
    for _ in range(5):
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
FL = F_.CONV_TRANSPOSE_W | F_.CONV_FLIP_K
for c in (16, 8, 12):
    x = torch.randn(n, c, device=dev)
    w = torch.randn(27, c, c, device=dev) * 0.1
    bn_x = torch.randn(n, c, device=dev)
    mean, inv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
    nblk = _lib.query('sgnn_conv_stats_blocks', n)
    part = torch.zeros(nblk, 2, c, dtype=torch.float64, device=dev)
    y = torch.empty(n, c, device=dev)
    acc = torch.randn(n, c, device=dev)

    def run(flags, addend, stats):
        _lib.call('sgnn_conv_fwd_epi', x.data_ptr(), n, c, 0, w.data_ptr(), 27, tab.data_ptr(), g.ld, n, c,
                  (acc if addend else y).data_ptr(), 0, flags, acc.data_ptr() if addend else None, 0, stats,
                  part.data_ptr() if stats else None, bn_x.data_ptr() if stats == 2 else None, 0,
                  mean.data_ptr() if stats == 2 else None, inv.data_ptr() if stats == 2 else None,
                  gamma.data_ptr() if stats == 2 else None, beta.data_ptr() if stats == 2 else None, 0.0)

    for wide in ((0, 1) if has_switch else (None,)):
        if wide is not None:
            _lib.tune('conv_wide_epi', wide)
        t = [timeit(lambda: run(0, False, 0)), timeit(lambda: run(0, False, 1)), timeit(lambda: run(FL, False, 0)),
             timeit(lambda: run(FL, True, 2))]
        print('<%d,%d> K27 %-12s fwd plain %6.1f us | fwd+stats %6.1f | dX plain %6.1f | dX+add+stats2 %6.1f   (dX full / fwd+stats = %.2f)'
              % (c, c, {None: '', 0: 'element-wise', 1: 'wide'}[wide], t[0], t[1], t[2], t[3], t[3] / t[1]))
    if has_switch:      # the two epilogues store the same rows and agree on the statistics to summation order
        outs = []
        for wide in (0, 1):
            _lib.tune('conv_wide_epi', wide)
            acc2 = torch.full((n, c), 0.5, device=dev)
            part.zero_()
            _lib.call('sgnn_conv_fwd_epi', x.data_ptr(), n, c, 0, w.data_ptr(), 27, tab.data_ptr(), g.ld, n, c,
                      acc2.data_ptr(), 0, FL, acc2.data_ptr(), 0, 2, part.data_ptr(), bn_x.data_ptr(), 0, mean.data_ptr(),
                      inv.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 0.0)
            outs.append((acc2, part.sum(0).clone()))
        same = torch.equal(outs[0][0], outs[1][0])
        rel = float(((outs[0][1] - outs[1][1]).abs() / outs[0][1].abs().clamp_min(1e-30)).max())
        print('   rows bit-identical: %s   statistics max rel diff %.2e' % (same, rel))
        assert same and rel < 1e-9
