"""Capacity mode: row counts stay on the device.

The reference decides the size of every generative level from predicted logits (`mask = sigmoid(out) > 0.5`,
torch/model.py:233, 322) and sizes the next level's tensors from it on the host — five device->host read-backs per
training step in this build's classic path, and a launch sequence that changes with every batch.  In capacity mode
every level gets a fixed CAPACITY (rows its buffers are sized for); the live row counts live in one small device
block and every kernel reads its count from there (include/sgnn_hip.h, "capacity mode").  A step then is a fixed
sequence of launches with fixed arguments: no read-back, and capturable in a HIP graph (train.GraphStep).

Count block layout (int64[64], one per Capacity):
    [0]            rows of the input level
    [1 .. 6]       rows of the encoder's stride-2 pyramid levels 1..
    [8 + 8g + 0]   generative level g (0 = kept sites of the dense coarse volume, g = 1.. after Refinement g-1):
                   kept rows;  [+1] = 8 x kept (their 8-child expansion, torch/model.py:192-207);
                   [+2 ..] rows of the kept sites' stride-2 pyramid (the next stage's U-Net levels 1, 2)
A count that exceeds its capacity is clamped by the producing kernel and SGNN_STATUS_OVERFLOW is raised in the
runtime's status word: nothing is written out of bounds, sgnn_adam_flat skips the update, and the caller re-runs the
batch with larger capacities (`grown()`).
"""
import numpy as np
import torch

SLOTS = 64
ENC0 = 1
GEN0 = 8
GEN_STRIDE = 8


def _round(n, m=256):
    n = int(n)
    return max(m, ((n + m - 1) // m) * m)


class Capacity(object):
    """input_rows: capacity of the input level; enc: capacities of the encoder pyramid levels 1..; gen: per
    generative level (kept capacity, [pyramid capacities])."""

    def __init__(self, device, input_rows, enc, gen):
        self.device = torch.device(device)
        self.input_rows = int(input_rows)
        self.enc = [min(int(c), self.input_rows) for c in enc]
        self.gen = [(int(k), [min(int(c), int(k)) for c in pyr]) for k, pyr in gen]
        assert len(self.enc) <= 6 and len(self.gen) <= (SLOTS - GEN0) // GEN_STRIDE
        self.counts = torch.zeros(SLOTS, dtype=torch.int64, device=self.device)
        self._pin = None

    # -- device views ----------------------------------------------------------------------------------
    def input_cnt(self):
        return self.counts[0:1]

    def enc_counts(self):
        return self.counts[ENC0:ENC0 + len(self.enc)]

    def gen_base(self, g):
        return GEN0 + GEN_STRIDE * g

    def kept2(self, g):
        """int64[2] view: kept rows, 8 x kept rows."""
        b = self.gen_base(g)
        return self.counts[b:b + 2]

    def pyr_counts(self, g, depth):
        b = self.gen_base(g) + 2
        return self.counts[b:b + depth]

    def set_input_rows(self, n):
        """Host-known row count of the batch's input level -> device (async copy from a pinned word; stream-ordered
        before the step's kernels, outside any captured graph)."""
        if n > self.input_rows:
            from .metadata import CapacityOverflow
            raise CapacityOverflow('input level has %d rows, capacity is %d' % (n, self.input_rows))
        if self._pin is None:
            self._pin = [torch.zeros(1, dtype=torch.int64).pin_memory() for _ in range(4)]
            self._k = 0
        pin = self._pin[self._k % len(self._pin)]
        self._k += 1
        pin[0] = int(n)
        self.counts[0:1].copy_(pin, non_blocking=True)

    # -- sizing ----------------------------------------------------------------------------------------
    @classmethod
    def from_log(cls, device, log, headroom=1.3, input_rows=None, min_rows=1024, n_gen=None, pyr_depth=2):
        """Capacities from the row counts an exact-mode forward pass logged (metadata.COUNT_LOG): every count times
        `headroom`, rounded up to 256 rows; a level that was empty gets `min_rows`.  n_gen: generative levels the
        capacity-mode forward will run (the classic pass stops at an empty level and logs fewer: the rest get
        `min_rows` and grow on demand)."""
        enc, gen = None, []
        n_in = 0
        for tag, n0, pyr in log:
            if tag == 'enc':
                n_in, enc = n0, list(pyr)
            elif tag == 'gen':
                gen.append((n0, list(pyr)))
        if enc is None:
            raise ValueError('the count log has no encoder entry')
        r = lambda n: _round(max(int(n * headroom), min_rows))
        while n_gen is not None and len(gen) < n_gen:
            gen.append((0, [0] * pyr_depth))
        return cls(device, input_rows if input_rows is not None else r(n_in), [r(c) for c in enc],
                   [(r(k), [r(c) for c in pyr]) for k, pyr in gen])

    def grown(self, factor=1.5):
        """A larger plan after an overflow."""
        g = lambda n: _round(int(n * factor))
        return Capacity(self.device, g(self.input_rows), [g(c) for c in self.enc],
                        [(g(k), [g(c) for c in pyr]) for k, pyr in self.gen])

    def describe(self):
        return {'input': self.input_rows, 'enc': list(self.enc), 'gen': [(k, list(p)) for k, p in self.gen]}

    def read(self):
        """Live counts as Python ints (ONE read-back; diagnostics, tests, trimming of inference results)."""
        h = self.counts.cpu().tolist()
        return {'input': h[0], 'enc': h[ENC0:ENC0 + len(self.enc)],
                'gen': [(h[self.gen_base(g)], h[self.gen_base(g) + 2:self.gen_base(g) + 2 + len(self.gen[g][1])])
                        for g in range(len(self.gen))]}


def trim(t):
    """Rows [0, live count) of a capacity-mode tensor (one read-back per distinct count; not for the training loop)."""
    cnt = getattr(t, '_sgnn_cnt', None)
    return t if cnt is None else t[:int(cnt.item())]
