"""BatchNorm kernel timings per level size (forward = partial + finalize + apply, backward = partial + finalize +
apply), with their HBM roofline: forward 12*N*C algorithmic bytes, backward 20*N*C (HISTORY.md §4)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgnn_amd import _lib  # noqa: E402
from sgnn_amd._lib import ptr  # noqa: E402

dev = torch.device('cuda:0')
C = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for n in (2432, 12032, 67648, 366085):
    x = torch.randn(n, C, device=dev)
    dy = torch.randn(n, C, device=dev)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    save, dgb = torch.empty(2, C, device=dev), torch.empty(2, C, device=dev)
    wsb = _lib.query('sgnn_bn_ws_bytes', n, C)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)

    def fwd():
        _lib.call('sgnn_bn_fwd', ptr(x), n, C, ptr(g), ptr(b), ptr(rm), ptr(rv), 1e-4, 0.9, 1, 0.0, ptr(save[0]),
                  ptr(save[1]), ptr(y), ptr(ws), wsb)

    def bwd():
        _lib.call('sgnn_bn_bwd', ptr(x), ptr(dy), n, C, ptr(g), ptr(b), ptr(save[0]), ptr(save[1]), 1, 0.0, ptr(dx),
                  ptr(dgb[0]), ptr(dgb[1]), ptr(ws), wsb)

    res = []
    for f, byts in ((fwd, 12 * n * C), (bwd, 20 * n * C)):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        res.append('%6.1f us %6.0f GB/s' % (us, byts / us / 1e3))
    print('N=%7d C=%d  fwd %s | bwd %s' % (n, C, res[0], res[1]))
