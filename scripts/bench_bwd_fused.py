"""Fused backward (dX + dW from one gather of dy, csrc/conv_bwd_fused.hip) against the two-kernel path, stand-alone on one
level (needs GPU).  VERDICT r5 item 1.  Per level: the data-gradient launch with its full backward epilogue (in-place addend +
BatchNorm-backward statistics), the weight-gradient launch + reduce, and the fused launch + reduce; dX rows must be
bit-identical, dW equal to fp32 summation order.
  python scripts/bench_bwd_fused.py [--batch 32] [--dim 64] [--iters 50] [--occ 0.05]"""
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
ap.add_argument('--occ', type=float, default=0.05)
ap.add_argument('--no-parity', action='store_true', help='timing only (ablation builds of the kernel)')
args = ap.parse_args()
dev = torch.device('cuda')
lib = _lib.load()
data = synth.make_batch(args.batch, (args.dim,) * 3, cfg=2, occupancy=args.occ)
g = Grid(coords_from_locs(data['input'][0], dev))
tab = g.subm_table()
n = g.n
tab2 = tab.view(27, g.ld)
rules = int((tab2[:, :n] >= 0).sum())
print('sites %d  rules %d (%.2f per row)' % (n, rules, rules / n))


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
c = 16
dy = torch.randn(n, c, device=dev)
x = torch.relu(torch.randn(n, c, device=dev))
w = torch.randn(27, c, c, device=dev) * 0.1
bn_x = torch.randn(n, c, device=dev)
mean, inv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
nblk = _lib.query('sgnn_conv_stats_blocks', n)
acc0 = torch.randn(n, c, device=dev)
ws_dw = torch.empty(_lib.query('sgnn_conv_bwd_weight_ws_bytes', n, 27, c, c) // 4 + 64, device=dev)
ws_f = torch.empty(_lib.query('sgnn_conv_bwd_fused_ws_bytes', n, c, c) // 4 + 64, device=dev)
assert _lib.query('sgnn_conv_bwd_fused_supported', n, c, c, 27) == 1, 'level too small for the fused kernel'


def run_dx(acc, part):
    _lib.call('sgnn_conv_fwd_epi', dy.data_ptr(), n, c, 0, w.data_ptr(), 27, tab.data_ptr(), g.ld, n, c, acc.data_ptr(), 0, FL,
              acc.data_ptr(), 0, 2, part.data_ptr(), bn_x.data_ptr(), 0, mean.data_ptr(), inv.data_ptr(), gamma.data_ptr(),
              beta.data_ptr(), 0.0)


def run_dw(dw):
    _lib.call('sgnn_conv_bwd_weight', x.data_ptr(), n, c, dy.data_ptr(), c, tab.data_ptr(), g.ld, 27, n, dw.data_ptr(), 0,
              ws_dw.data_ptr(), ws_dw.numel() * 4)


def run_fused(acc, part, dw):
    _lib.call('sgnn_conv_bwd_fused', dy.data_ptr(), n, c, 0, x.data_ptr(), c, c, w.data_ptr(), tab.data_ptr(), g.ld,
              acc.data_ptr(), 0, acc.data_ptr(), 0, 2, part.data_ptr(), bn_x.data_ptr(), 0, mean.data_ptr(), inv.data_ptr(),
              gamma.data_ptr(), beta.data_ptr(), 0.0, dw.data_ptr(), ws_f.data_ptr(), ws_f.numel() * 4, None)


# parity first
acc_a, acc_b = acc0.clone(), acc0.clone()
part_a = torch.zeros(nblk, 2, c, dtype=torch.float64, device=dev)
part_b = torch.zeros(nblk, 2, c, dtype=torch.float64, device=dev)
dw_a, dw_b = torch.zeros(27, c, c, device=dev), torch.zeros(27, c, c, device=dev)
run_dx(acc_a, part_a)
run_dw(dw_a)
run_fused(acc_b, part_b, dw_b)
torch.cuda.synchronize()
same = torch.equal(acc_a, acc_b)
srel = float(((part_a.sum(0) - part_b.sum(0)).abs() / part_a.sum(0).abs().clamp_min(1e-30)).max())
# exact weight gradient in fp64 (dW[k] = sum_j x[table[k][j]]^T dy[j]) as the yardstick of both kernels
t = tab2[:, :n].long()
ref = torch.zeros(27, c, c, dtype=torch.float64, device=dev)
xd, dyd = x.double(), dy.double()
for k in range(27):
    ok = t[k] >= 0
    ref[k] = xd[t[k][ok]].t() @ dyd[ok]
scale = float(ref.abs().max())
ea, eb = float((dw_a.double() - ref).abs().max()) / scale, float((dw_b.double() - ref).abs().max()) / scale
print('dX rows bit-identical: %s   statistics max rel diff %.2e   dW max err / max|dW|: two-kernel %.2e  fused %.2e'
      % (same, srel, ea, eb))
assert args.no_parity or (same and srel < 1e-9 and eb < 1e-5), 'parity failed'

part = torch.zeros(nblk, 2, c, dtype=torch.float64, device=dev)
acc = acc0.clone()
dw = torch.zeros(27, c, c, device=dev)
t_dx = timeit(lambda: run_dx(acc, part))
t_dw = timeit(lambda: run_dw(dw))
t_f = timeit(lambda: run_fused(acc, part, dw))
flops = 2.0 * rules * c * c
print('<16,16> K27  dX + addend + stats2 %6.1f us | dW + reduce %6.1f us | sum %6.1f | fused dX + dW + reduce %6.1f us  '
      '(%.2fx the sum; %.3f of fp32 MFMA peak over both products, rules only)'
      % (t_dx, t_dw, t_dx + t_dw, t_f, t_f / (t_dx + t_dw), 2 * flops / (t_f * 1e-6) / 157.3e12))
