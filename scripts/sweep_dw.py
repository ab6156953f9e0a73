"""Weight-gradient kernel on small levels (needs GPU): one offset per workgroup (default below 16 k rows) against the
9-offsets-per-workgroup layout of the large levels.  <16,16>, K = 27, surface sites; sgnn_tune.conv_small(0) = A/B."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sgnn_amd import synth, _lib
from sgnn_amd.scn.metadata import Grid, coords_from_locs
dev = torch.device('cuda'); lib = _lib.load()
def timeit(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for batch, dim in ((1, 8), (1, 16), (2, 24), (2, 32), (4, 32), (2, 64), (4, 64)):
    locs = synth.make_batch(batch, (dim,) * 3, cfg=2, occupancy=0.1)['input'][0]
    g = Grid(coords_from_locs(locs, dev)); tab = g.subm_table()
    for cin, cout in ((16, 16), (8, 8)):
        x = torch.randn(g.n, cin, device=dev); dy = torch.randn(g.n, cout, device=dev)
        dw = torch.empty(27, cin, cout, device=dev)
        wsb = _lib.query('sgnn_conv_bwd_weight_ws_bytes', g.n, 27, cin, cout)
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=dev)
        args = (x.data_ptr(), g.n, cin, dy.data_ptr(), cout, tab.data_ptr(), g.ld, 27, g.n, dw.data_ptr(), 0, ws.data_ptr(), wsb)
        res = []
        for on in (0, 1):
            _lib.tune('conv_small', on)
            t = timeit(lambda: _lib.call('sgnn_conv_bwd_weight', *args))
            res.append((t, dw.clone()))
        _lib.tune('conv_small', 1)
        print('N %6d <%d,%d>  9 offsets/WG %.1f us   1 offset/WG %.1f us   identical %s' %
              (g.n, cin, cout, res[0][0], res[1][0], bool(torch.equal(res[0][1], res[1][1]))))
