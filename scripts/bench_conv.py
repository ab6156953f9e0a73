"""Kernel-tuning harness (needs GPU): times the sparse-conv entry points on a realistic level.
  python scripts/bench_conv.py [--batch 32] [--dim 64] [--kind surface|gen]
surface: synthetic TSDF surface blocks (the encoder's level-0 grid, ~13k sites/block);
gen:     8-child expansion of a surface grid at dim/2 (what the generative levels look like)."""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgnn_amd import synth, _lib
from sgnn_amd.scn import functions as F_
from sgnn_amd.scn.metadata import Grid, coords_from_locs, build_down2

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--dim', type=int, default=64)
ap.add_argument('--kind', default='surface')
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--cases', default='16x16,48x16,16x48,8x8,26x16')
args = ap.parse_args()
dev = torch.device('cuda')
d = args.dim if args.kind == 'surface' else args.dim // 2
data = synth.make_batch(args.batch, (d,) * 3, cfg=2, occupancy=0.05 if args.kind == 'surface' else 0.2)
coords = coords_from_locs(data['input'][0], dev)
if args.kind == 'gen':
    coords = F_.expand8_coords(coords)
g = Grid(coords)
torch.cuda.synchronize()
t0 = time.perf_counter(); g.hash(); torch.cuda.synchronize(); t1 = time.perf_counter()
tab = g.subm_table(); torch.cuda.synchronize(); t2 = time.perf_counter()
valid = int((tab.view(27, g.ld)[:, :g.n] >= 0).sum().item())
print('sites %d  rules %d (R/N %.1f)  hash %.0f us  rulebook(first call) %.0f us' % (g.n, valid, valid / g.n, 1e6 * (t1 - t0), 1e6 * (t2 - t1)))

def timeit(fn, iters=args.iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us

lib = _lib.load()
nbr = torch.empty(27 * g.ld, dtype=torch.int32, device=dev)
keys, vals, cap = g.hash()
for on in (1, 0):
    _lib.tune('rulebook_lds', on)
    us = timeit(lambda: _lib.call('sgnn_rulebook_subm3', keys.data_ptr(), vals.data_ptr(), cap, g.coords.data_ptr(), g.n,
                                  nbr.data_ptr(), g.ld, None), 20)
    print('rulebook_subm3 (%s): %.1f us  (%.1f GB/s on 16N+108N bytes)' % ('LDS window' if on else 'global probes', us, g.n * 124 / us / 1e3))
_lib.tune('rulebook_lds', 0)
dv = d if args.kind == 'surface' else 2 * d      # children live on the next finer lattice
vol = torch.full((dv ** 3 * args.batch,), -1, dtype=torch.int32, device=dev)
us = timeit(lambda: _lib.call('sgnn_rulebook_subm3_dense', keys.data_ptr(), vals.data_ptr(), cap, g.coords.data_ptr(), g.n,
                              dv, dv, dv, vol.data_ptr(), vol.numel(), nbr.data_ptr(), g.ld, None), 20)
print('rulebook_subm3_dense (index volume, mark + table + clear): %.1f us' % us)
assert torch.equal(nbr, tab)
if os.environ.get('RULEBOOK_ONLY'): sys.exit(0)
for case in args.cases.split(','):
    cin, cout = (int(v) for v in case.split('x'))
    x = torch.randn(g.n, cin, device=dev); w = torch.randn(27, cin, cout, device=dev) * 0.1
    dy = torch.randn(g.n, cout, device=dev)
    flops = 2.0 * g.n * 27 * cin * cout
    byts = 4.0 * g.n * (cin + cout) + 4.0 * 27 * g.n
    us = timeit(lambda: F_.conv_fwd_raw(x, cin, w, 27, tab, g.ld, g.n, cout, 0, 0))
    print('conv_fwd<%d,%d>  %8.1f us  %6.1f TF  %7.1f GB/s alg' % (cin, cout, us, flops / us / 1e6, byts / us / 1e3))
    us = timeit(lambda: F_.conv_dw_raw(x, cin, dy, cout, tab, g.ld, 27, g.n, 0))
    print('conv_dw <%d,%d>  %8.1f us  %6.1f TF  %7.1f GB/s alg' % (cin, cout, us, flops / us / 1e6, byts / us / 1e3))
