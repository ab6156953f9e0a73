"""Crossover of the small-level conv kernel vs the pipelined big kernel by level size (needs GPU)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sgnn_amd import synth, _lib
from sgnn_amd.scn import functions as F_
from sgnn_amd.scn.metadata import Grid, coords_from_locs
lib = _lib.load()
dev = torch.device('cuda')
def timeit(fn, iters=200):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for nb, dim in ((1, 32), (4, 32), (1, 64), (2, 64), (4, 64), (8, 64), (16, 64), (32, 64)):
    data = synth.make_batch(nb, (dim,) * 3, cfg=2)
    g = Grid(coords_from_locs(data['input'][0], dev)); tab = g.subm_table()
    for cin, cout in ((16, 16), (48, 16), (16, 48)):
        x = torch.randn(g.n, cin, device=dev); w = torch.randn(27, cin, cout, device=dev) * 0.1
        y = torch.empty(g.n, cout, device=dev)
        args = (x.data_ptr(), g.n, cin, w.data_ptr(), 27, tab.data_ptr(), g.ld, g.n, cout, y.data_ptr(), 0, 0)
        res = []
        for thr in (1 << 30, 0):
            _lib.tune('conv_small_rows', thr)
            res.append(timeit(lambda: _lib.call('sgnn_conv_fwd', *args)))
        _lib.tune('conv_small_rows', 160 * 256)
        fl = 2.0 * int((tab.view(27, g.ld)[:, :g.n] >= 0).sum()) * cin * cout
        print('rows %7d <%d,%d>  small %7.1f us (%5.1f TF)   big %7.1f us (%5.1f TF)' % (g.n, cin, cout, res[0], fl / res[0] / 1e6, res[1], fl / res[1] / 1e6))
