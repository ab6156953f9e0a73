"""Big-level conv kernel under alternative builds (SGNN_LIB=...): fwd <16,16>, <8,8>, <48,16> at N = 366 k."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sgnn_amd import synth, _lib
from sgnn_amd.scn.metadata import Grid, coords_from_locs
dev = torch.device('cuda')
def timeit(fn, iters=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
data = synth.make_batch(32, (64,) * 3, cfg=2)
g = Grid(coords_from_locs(data['input'][0], dev)); tab = g.subm_table()
for cin, cout in ((16, 16), (8, 8), (48, 16), (16, 48)):
    x = torch.randn(g.n, cin, device=dev); w = torch.randn(27, cin, cout, device=dev) * 0.1
    y = torch.empty(g.n, cout, device=dev)
    args = (x.data_ptr(), g.n, cin, w.data_ptr(), 27, tab.data_ptr(), g.ld, g.n, cout, y.data_ptr(), 0, 0)
    print('%s <%d,%d> %.1f us' % (os.path.basename(_lib.LIB_PATH), cin, cout, timeit(lambda: _lib.call('sgnn_conv_fwd', *args))))
# tile kernel (unique rows through LDS) against the gather kernel, and the cost of building the tile index
idx = torch.empty(_lib.query('sgnn_tile_index_bytes', g.ld), dtype=torch.uint8, device=dev)
print('tile index build %.1f us (%d rows, %.1f MB)' % (timeit(lambda: _lib.call('sgnn_tile_index', tab.data_ptr(), g.ld, idx.data_ptr())), g.n, idx.numel() / 1e6))
cnt = idx[: 4 * (g.ld // 128)].view(torch.int32)[: (g.n + 127) // 128]
print('unique rows per tile: mean %.0f max %d overflow %d' % (float(cnt.clamp(min=0).float().mean()), int(cnt.max()), int((cnt < 0).sum())))
for cin, cout in ((16, 16), (8, 8), (12, 12), (8, 12), (12, 16)):
    x = torch.randn(g.n, cin, device=dev); w = torch.randn(27, cin, cout, device=dev) * 0.1
    y = torch.empty(g.n, cout, device=dev)
    a0 = (x.data_ptr(), g.n, cin, w.data_ptr(), 27, tab.data_ptr(), g.ld, g.n, cout, y.data_ptr(), 0, 0)
    a1 = (x.data_ptr(), g.n, cin, w.data_ptr(), tab.data_ptr(), g.ld, g.n, cout, y.data_ptr(), 0, idx.data_ptr())
    print('<%d,%d> gather %.1f us  tile %.1f us' % (cin, cout, timeit(lambda: _lib.call('sgnn_conv_fwd', *a0)),
                                                   timeit(lambda: _lib.call('sgnn_conv_fwd_tiled', *a1))))
lib = _lib.load()
for grid in (256, 512, 1024):
    lib.sgnn_conv_set_tiled(grid)
    cin = cout = 16
    x = torch.randn(g.n, cin, device=dev); w = torch.randn(27, cin, cout, device=dev) * 0.1
    y = torch.empty(g.n, cout, device=dev)
    a1 = (x.data_ptr(), g.n, cin, w.data_ptr(), tab.data_ptr(), g.ld, g.n, cout, y.data_ptr(), 0, idx.data_ptr())
    print('<16,16> tile, %d persistent workgroups: %.1f us' % (grid, timeit(lambda: _lib.call('sgnn_conv_fwd_tiled', *a1))))
lib.sgnn_conv_set_tiled(512)
