"""Gather-instruction micro-benchmark on a real level (needs GPU): scripts/kernels/gather_bench.hip.
  python scripts/bench_gather.py [--batch 32] [--dim 64] [--iters 20]
Prints, per lane mapping, the time of one pass over the level's 3x3x3 rulebook and the rate in gathered bytes."""
import argparse, ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgnn_amd import synth
from sgnn_amd.scn.metadata import Grid, coords_from_locs

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--dim', type=int, default=64)
ap.add_argument('--iters', type=int, default=20)
args = ap.parse_args()
lib = ctypes.CDLL(os.path.join(ROOT, 'scripts', 'kernels', 'libgather_bench.so'))
lib.gather_bench.restype = ctypes.c_int
lib.gather_bench.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int,
                             ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = torch.device('cuda')
data = synth.make_batch(args.batch, (args.dim,) * 3, cfg=2, occupancy=0.05)
g = Grid(coords_from_locs(data['input'][0], dev))
tab = g.subm_table()
rules = int((tab.view(27, g.ld)[:, :g.n] >= 0).sum().item())
print('sites %d  rules %d (R/N %.2f)' % (g.n, rules, rules / g.n))
out = torch.zeros(17 * (g.ld + 512) + 1024, device=dev)
names = {0: '64 B rows, conv (MFMA-operand) mapping, b128', 1: '64 B rows, row-contiguous mapping, b128',
         2: '64 B rows, row-contiguous + 4 ds_bpermute', 3: '64 B rows, row-contiguous + LDS write/read b128',
         4: '32 B rows, conv mapping, b64', 5: '32 B rows, row-contiguous mapping, b128 (2 instr / 64 rows)',
         6: '64 B rows, row-contiguous LDS-DMA + ds_read_b128, 4 waves/WG', 7: 'the same, 8 waves/WG',
         10: 'conv loop: 1 offset in flight, no MFMA, 20 waves/CU', 11: 'conv loop: 2 offsets in flight, no MFMA',
         12: 'conv loop: 3 offsets in flight, no MFMA', 13: 'conv loop: 1 offset in flight + 16 MFMA/offset',
         14: 'conv loop: 2 offsets in flight + 16 MFMA/offset', 15: 'conv loop: 3 offsets in flight + 16 MFMA/offset',
         16: 'conv loop: 4 offsets in flight + 16 MFMA/offset',
         17: 'conv loop: 1 in flight + 16 MFMA + weights staged in LDS', 18: 'conv loop: 1 in flight + 16 MFMA + 64 B/row store',
         19: 'conv loop: 1 in flight + 16 MFMA + LDS weights + store',
         8: '64 B rows, conv mapping, offsets walked dx-major (no shared lines in flight)',
         20: 'per (dz,dy): three full gathers (baseline of modes 21/22)', 21: 'per (dz,dy): 1 full + 2 fallback gathers + DPP shifts',
         22: 'mode 21, fallback skipped when no lane needs it'}
ref = {}
xs = {16: torch.randn(g.n, 16, device=dev), 8: torch.randn(g.n, 8, device=dev)}
ap_modes = [int(v) for v in os.environ.get('GATHER_MODES', '10,13,17,18,19').split(',')]
t = tab.view(27, g.ld)[:, :g.n]
hit = 0
for k in range(0, 27, 3):
    j = torch.arange(g.n - 1, device=dev)
    okp = (t[k + 2][:-1] >= 0) & (t[k + 2][:-1] == t[k + 1][1:]) & ((j % 16) != 15)
    okm = (t[k][1:] >= 0) & (t[k][1:] == t[k + 1][:-1]) & (((j + 1) % 16) != 0)
    hit += int(okp.sum()) + int(okm.sum())
side = int((t[0::3] >= 0).sum()) + int((t[2::3] >= 0).sum())
print('dx = -1 / +1 rules served by a lane shift of the dx = 0 gather: %.1f %% of %d' % (100.0 * hit / side, side))
for mode in ap_modes:
    c = 8 if mode in (4, 5) else 16
    x = xs[c]
    stream = torch.cuda.current_stream().cuda_stream
    call = lambda: lib.gather_bench(x.data_ptr(), g.n, c, tab.data_ptr(), g.ld, 27, g.n, out.data_ptr(), mode, stream)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / args.iters * 1e3
    s = float(out.double().sum())
    key = c
    key = (c, mode >= 20)
    if key in ref and (mode < 10 or mode >= 20):
        assert abs(s - ref[key]) <= 1e-4 * max(1.0, abs(ref[key])), (mode, s, ref[key])   # every mapping loads the same rows
    ref.setdefault(key, s)
    print('mode %d  %-62s %7.1f us   %6.2f TB/s gathered (rules only: %6.2f)   %5.1f cycles/gather-instr/CU @2.4GHz'
          % (mode, names[mode], us, 27.0 * g.n * c * 4 / us / 1e6, rules * c * 4.0 / us / 1e6,
             us * 1e-6 * 2.4e9 * 256 / (g.ld / 64 * 27 * (4 if mode != 5 else 2))))
