"""The data-parallel protocol of train.GraphStep on the CPU: two SIMULATED ranks (threads) drive the REAL slot logic —
GraphStep.__call__, _check, _overflow, _maybe_replan, _resize, _drain — with the device work replaced by bookkeeping: a "step"
compares the batch's row counts with the rank's plan (overflow = a count above its capacity, counts ride back clamped like the
kernels clamp them), the all-reduce is a barrier that exchanges (batch index, step kind, local overflow bit) and returns the
merged bit.  What the GPU test (tests/test_gpu_dp_protocol.py) shows with real kernels on the device, this one pins in the
`-m "not gpu"` suite: every slot sums the same batch on both ranks, overflowed slots are re-run once per rank after the same
slot, whatever rank-local event (re-plan, input outgrowing the plan) happens in between — and round 4's rank-local drain is
caught.  The step being replicated: torch/train.py:245-268; data parallelism is this build's addition."""
import threading

import numpy as np
import pytest
import torch

from sgnn_amd.scn.capacity import Capacity, ENC0
from sgnn_amd.train import GraphStep


class SlotMismatch(AssertionError):
    pass


class Bus(object):
    """A two-party all-reduce: every call is one collective; ranks meet at a barrier, see what the peer brought."""

    def __init__(self, world):
        self.world, self.box = world, [None] * world
        self.barrier = threading.Barrier(world, timeout=20)
        self.slots = []

    def exchange(self, rank, item):
        self.box[rank] = item
        self.barrier.wait()
        every = list(self.box)
        self.barrier.wait()
        if rank == 0:
            self.slots.append(every)
        if len(set(e[0] for e in every)) != 1:
            raise SlotMismatch('a collective sums different batches: %r' % (every,))
        return any(e[2] for e in every)


class _Done(object):
    def synchronize(self):
        pass


class _Opt(object):
    def bind_programs(self, model):
        pass


def _counts_of(batch):
    """live rows per count slot of the plan (scn/capacity.py layout) for a simulated batch"""
    v = [0] * 64
    v[0] = batch['n_in']
    for l, n in enumerate(batch['enc']):
        v[ENC0 + l] = n
    for g, (k, pyr) in enumerate(batch['gen']):
        b = 8 + 8 * g
        v[b], v[b + 1] = k, 8 * k
        for l, n in enumerate(pyr):
            v[b + 2 + l] = n
    return v


class SimStep(GraphStep):
    """GraphStep with the device work replaced by bookkeeping (see the module docstring)."""

    def __init__(self, rank, bus, headroom):
        self.rank, self.bus = rank, bus
        self.grad_sync, self.world_size = (lambda flat: None), bus.world      # (only `is not None` is looked at)
        self.model, self.opt = None, _Opt()
        self.headroom, self.use_graph, self.settle = float(headroom), True, False
        self.capacity = self.key = self.weights = self.graphs = self.static = None
        self.stage, self.pending, self._bound = 0, [], False
        self.outputs, self.keep_outputs, self.loss, self.losses = None, False, None, None
        self.slot_batch = self.slot_kind = None
        self.overflow_log, self._n_issued = [], 0
        self.stats = {'probe_steps': 0, 'eager_steps': 0, 'captures': 0, 'replays': 0, 'overflows': 0, 'replans': 0,
                      'replay_host_ms': 0.0}
        self.applied = []            # (batch id, kind) of every slot whose update this rank applied

    # -- what a slot is, here ---------------------------------------------------------------------------------
    def _slot(self, batch, kind, counts, local_overflow):
        self.slot_batch, self.slot_kind = batch, kind
        merged = self.bus.exchange(self.rank, (batch['id'], kind, bool(local_overflow)))
        if not merged:
            self.applied.append((batch['id'], kind))
        self._word, self._counts = (4 if merged else 0), counts
        return merged

    def _plan_for(self, batch):
        r = lambda n: max(256, -(-int(max(n * self.headroom, 1024)) // 256) * 256)
        return Capacity('cpu', r(batch['n_in']), [r(n) for n in batch['enc']],
                        [(r(k), [r(n) for n in pyr]) for k, pyr in batch['gen']])

    def _probe(self, batch, loss_weights):
        self._slot(batch, 'probe', None, False)
        self._issue_status(batch, loss_weights, probe=True)
        new = self._plan_for(batch)
        if self.capacity is not None:        # never below what an earlier batch needed (as GraphStep._probe)
            old = self.capacity
            new = Capacity('cpu', max(new.input_rows, old.input_rows), [max(a, b) for a, b in zip(new.enc, old.enc)],
                           [(max(k, ko), [max(a, b) for a, b in zip(p, po)]) for (k, p), (ko, po) in zip(new.gen, old.gen)])
        self.capacity = new
        self._live, self._hist = None, []
        self.stats['probe_steps'] += 1
        self.loss = 0.0
        return self.loss

    def _make_static(self, batch):
        self.static = object()

    def _load(self, batch):
        pass

    def _capacity_step(self, kind):
        batch, cap = self.slot_batch_next, self.capacity
        want = _counts_of(batch)
        # capacities per count slot, in the layout of _counts_of
        capv = _counts_of({'n_in': cap.input_rows, 'enc': cap.enc, 'gen': cap.gen})
        over = any(w > c for w, c in zip(want, capv) if c)
        clamped = [min(w, c) if c else w for w, c in zip(want, capv)]
        self._slot(batch, kind, clamped, over)

    def _capacity_step_eager(self, loss_weights):
        self._capacity_step('eager')
        return 0.0, None, None

    def _capture(self, loss_weights):
        self.graphs = (object(),)
        self.stats['captures'] += 1

    def _replay(self):
        self._capacity_step('replay')
        self.stats['replays'] += 1
        return 0.0, None

    def _issue_status(self, batch, loss_weights, probe=False):
        pin = torch.zeros(65, dtype=torch.int64)
        pin[0] = self._word
        if not probe:
            pin[1:] = torch.tensor(self._counts, dtype=torch.int64)
        self._n_issued += 1
        self.pending.append((_Done(), pin, batch, loss_weights, None if probe else self.capacity, self._n_issued - 1, None))

    def __call__(self, batch, loss_weights):
        self.slot_batch_next = batch
        return GraphStep.__call__(self, batch, loss_weights)


class LegacySimStep(SimStep):
    """Round 4's order: a rank-local re-plan first retired everything in flight (including the newest step's word)."""

    def _resize(self, live, grow=1.0):
        self._drain()
        if self.stage < 2:
            return
        SimStep._resize(self, live, grow)


def _batch(i, rank, scale=1.0):
    rng = np.random.default_rng(100 * rank + i)
    s = lambda n: int(n * scale * rng.uniform(0.9, 1.1))
    return {'id': i, 'n_in': s(8000), 'enc': [s(1400), s(300), s(80)],
            'gen': [(s(128), [s(40), s(16)]), (s(700), [s(128), s(40)]), (s(2700), [s(700), s(128)]), (s(10000), [s(2700), s(700)])],
            'sdf': torch.empty(2, 1, 4, 4, 4)}


def _fix_input(b):
    b['input'] = [torch.empty(b['n_in'], 4)]
    return b


def _shrink(step, g):
    cap = step.capacity
    k = cap.gen[g][0]
    step.capacity = Capacity('cpu', cap.input_rows, cap.enc, [(kk if i != g else 256, p) for i, (kk, p) in enumerate(cap.gen)])
    step.graphs, step.stage, step._live, step._hist = None, 1, None, []
    return k


def _run(cls, scenario, n_calls):
    bus = Bus(2)
    out, errs = [None, None], [None, None]

    def rank_main(rank):
        try:
            lw = np.ones(5, dtype=np.float32)
            step = cls(rank, bus, 2.0 if rank == 0 else (10.0 if scenario == 'A' else 1.5))
            for it in range(n_calls):
                big = scenario == 'B' and rank == 1 and it == 4
                b = _fix_input(_batch(it, rank, 3.0 if big else 1.0))
                if it == 3 and rank == 0:
                    _shrink(step, 2)
                if it == 3 and rank == 1 and scenario == 'A':          # on the brink of the "plan too loose" re-plan
                    step.headroom, step._loose = 2.0, 4
                if big:
                    assert b['n_in'] > step.capacity.input_rows
                step(b, lw)
            out[rank] = step
        except BaseException as e:      # a broken barrier in the peer follows from the first error
            errs[rank] = e
            bus.barrier.abort()

    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    return bus, out, errs


def _ids(bus):
    return [e[0][0] for e in bus.slots]


def test_one_rank_overflow_with_a_rank_local_replan_in_the_same_slot():
    bus, steps, errs = _run(SimStep, 'A', 8)
    assert errs == [None, None], errs
    assert _ids(bus) == [0, 1, 2, 3, 4, 3, 4, 5, 6, 7], bus.slots
    assert bus.slots[3] == [(3, 'eager', True), (3, 'replay', False)]                # only rank 0 overflowed ...
    assert steps[1].stats['replans'] == 1 and steps[0].stats['replans'] == 0         # ... rank 1 re-planned in that slot
    assert [k for _, k, _ in bus.slots[5]] == ['probe', 'probe'] and [k for _, k, _ in bus.slots[6]] == ['probe', 'probe']
    for s in steps:
        assert s.stats['overflows'] == 2
        # slots 3 and 4 were skipped by BOTH ranks and each batch's update was applied exactly once
        assert [i for i, _ in s.applied] == [0, 1, 2, 3, 4, 5, 6, 7], s.applied
    assert all(full for _, full in steps[0].overflow_log) and not any(full for _, full in steps[1].overflow_log)


def test_input_outgrows_the_plan_on_one_rank_after_an_overflow_on_the_other():
    bus, steps, errs = _run(SimStep, 'B', 7)
    assert errs == [None, None], errs
    assert _ids(bus) == [0, 1, 2, 3, 4, 3, 4, 5, 6], bus.slots
    assert bus.slots[4] == [(4, 'replay', True), (4, 'probe', False)]       # rank 0 replays (and overflows again), rank 1 probes
    for s in steps:
        assert s.stats['overflows'] == 2 and [i for i, _ in s.applied] == [0, 1, 2, 3, 4, 5, 6], (s.stats, s.applied)


def test_round_4_rank_local_drain_is_caught():
    bus, steps, errs = _run(LegacySimStep, 'A', 8)
    bad = [e for e in errs if isinstance(e, SlotMismatch)]
    assert bad, errs
    # the disagreement: rank 1 re-runs batch 3 (it saw the merged overflow inside its re-plan) opposite rank 0's batch 4
    assert "(4, 'replay'" in str(bad[0]) and "(3, 'probe'" in str(bad[0]), str(bad[0])
    assert _ids(bus)[:4] == [0, 1, 2, 3], bus.slots        # everything before it was in step (the bad slot may be logged too)


def test_counts_of_an_overflowed_slot_never_size_a_plan():
    """single simulated rank: the plan is far too small for two levels; the clamped counts must not drive a re-plan"""
    bus = Bus(1)
    step = SimStep(0, bus, 2.0)
    lw = np.ones(5, dtype=np.float32)
    first = _fix_input(_batch(0, 0))
    first['gen'] = first['gen'][:2] + [(0, [0, 0]), (0, [0, 0])]            # the hierarchy died in the probe step
    step(first, lw)
    for it in range(1, 6):
        step(_fix_input(_batch(it, 0)), lw)
    step._drain()
    assert step.stats['overflows'] == 2 and step.stats['replans'] == 0, (step.stats, step.overflow_log)
    assert _ids(bus) == [0, 1, 2, 1, 2, 3, 4, 5]
    assert step.capacity.gen[3][0] >= 10000
