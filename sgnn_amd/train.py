"""One training step and the data-parallel wrapper — counterpart of torch/train.py:245-268 (H2D copies,
compute_targets, forward, compute_loss, backward, optimizer.step).

Multi-GPU (an addition of this build; the reference is single-process): one process
per GPU, each rank owns `batch_size` independent TSDF blocks (batch index is part of every voxel key, so
blocks never interact in a sparse op), BatchNorm statistics stay per replica, and the only exchange is one
all-reduce of a flat fp32 gradient buffer (643 735 floats = 2.57 MB) over RCCL/xGMI after backward.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from . import loss as loss_util


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_device_numa(device=None):
    """One process per GPU: run it on the CPUs of the NUMA node that GPU hangs off (sysfs local_cpulist of its PCI
    function).  The training step is a stream of ~1 350 kernel launches; issued from the remote socket every launch
    and every count read-back crosses the inter-socket link (measured on a 2-socket MI355X host: 9.4 vs 10.4 ms per
    step, at random, until pinned).  Returns the CPU set, or None when the topology cannot be read (then nothing is
    changed).  SGNN_NO_BIND=1 disables it."""
    import os
    if os.environ.get('SGNN_NO_BIND') == '1' or not torch.cuda.is_available() or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        idx = torch.cuda.current_device() if device is None else torch.device(device).index
        p = torch.cuda.get_device_properties(idx)
        bdf = '%04x:%02x:%02x.0' % (getattr(p, 'pci_domain_id', 0), p.pci_bus_id, p.pci_device_id)
        with open('/sys/bus/pci/devices/%s/local_cpulist' % bdf) as f:
            cpus = _parse_cpulist(f.read()) & os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return cpus
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


class FlatGradAllReduce(object):
    """Averages gradients across ranks with a single collective on one flat buffer.

    The buffer carries one extra float per parameter tensor: 1 where this rank produced a gradient.  A parameter no
    rank reached this step (a generative level whose loss weight is still 0, train.py:203-231) keeps ``grad = None``
    on every rank, so Adam skips it exactly as the single-process run does (no step-counter advance, no weight
    decay).  A parameter only SOME ranks reached (rank-local empty level) receives sum / world_size — the mean over
    all replicas, the missing ones contributing zero — on every rank."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self.flat = None

    def __call__(self):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return
        ps = self.params
        dev, dt = ps[0].device, ps[0].dtype
        have_local = [p.grad is not None for p in ps]
        zeros = None
        if not all(have_local):
            zeros = torch.zeros(max(p.numel() for p, h in zip(ps, have_local) if not h), dtype=dt, device=dev)
        flags = torch.tensor([1.0 if h else 0.0 for h in have_local], dtype=dt).to(dev, non_blocking=True)
        grads = [p.grad.reshape(-1) if h else zeros[:p.numel()] for p, h in zip(ps, have_local)]
        self.flat = torch.cat(grads + [flags])           # one gather kernel instead of one copy per tensor
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        world = dist.get_world_size(self.group)
        if all(have_local):
            have_any = have_local                          # every rank sees >= 1 contributor: no read-back needed
        else:
            have_any = (self.flat[self.numel:] > 0).tolist()   # one small D2H, only on steps with a local gap
        self.flat[:self.numel].div_(world)
        pieces = self.flat[:self.numel].split([p.numel() for p in ps])
        dst, src = [], []
        for p, v, hl, ha in zip(ps, pieces, have_local, have_any):
            if not ha:
                p.grad = None
            elif hl:
                dst.append(p.grad)
                src.append(v.view_as(p))
            else:
                p.grad = v.view_as(p).clone()
        if dst:
            torch._foreach_copy_(dst, src)


class FastAdam(torch.optim.Adam):
    """torch.optim.Adam(fused=True) with the per-step Python bookkeeping removed: same state layout (state_dict /
    load_state_dict / lr schedulers work unchanged) and the same fused multi-tensor kernel, but the per-parameter
    Python loop of Adam.step (_init_group over 307 tensors, grouping by device/dtype: ~1.2 ms of a 10 ms step) runs
    once; a step is then one _foreach_add_ on the step counters and one _fused_adam_ call."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=True)
        self._fast = None

    def _lists(self, group):
        """Parameters that carry a gradient this step (the others are skipped, as Adam.step does), with their state
        tensors (created like Adam._init_group does for fused=True: device step counter, zero moments)."""
        ps, avgs, sqs, steps = [], [], [], []
        for p in group['params']:
            if p.grad is None:
                continue
            st = self.state[p]
            if len(st) == 0:
                st['step'] = torch.zeros((), dtype=torch.float32, device=p.device)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            ps.append(p)
            avgs.append(st['exp_avg'])
            sqs.append(st['exp_avg_sq'])
            steps.append(st['step'])
        return ps, avgs, sqs, steps

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._fast = None                 # the cached moment / step tensors were replaced

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._fast = None

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        if any(g['amsgrad'] or g['maximize'] or g.get('capturable') or g.get('differentiable')
               for g in self.param_groups):
            return super().step()         # decided before any group was touched: no double step
        for gi, group in enumerate(self.param_groups):
            n_with_grad = sum(p.grad is not None for p in group['params'])
            cache = self._fast.get(gi) if self._fast else None
            if cache is None or cache[0] != n_with_grad or any(p.grad is None for p in cache[1][0]) or \
                    any(self.state[p].get('exp_avg') is not a for p, a in zip(cache[1][0][:1], cache[1][1][:1])):
                cache = (n_with_grad, self._lists(group))      # first step, or a level (dis)appeared
                self._fast = dict(self._fast or {})
                self._fast[gi] = cache
            ps, avgs, sqs, steps = cache[1]
            if not ps:
                continue
            grads = [p.grad for p in ps]
            beta1, beta2 = group['betas']
            torch._foreach_add_(steps, 1)
            torch._fused_adam_(ps, grads, avgs, sqs, [], steps, amsgrad=False, lr=group['lr'], beta1=beta1, beta2=beta2,
                               weight_decay=group['weight_decay'], eps=group['eps'], maximize=False, grad_scale=None,
                               found_inf=None)


def make_optimizer(params, lr=1e-3, weight_decay=0.0):
    """Adam as train.py:81 (fused multi-tensor kernel, see FastAdam)."""
    return FastAdam(params, lr=lr, weight_decay=weight_decay)


def to_device(batch, device):
    """train.py:256-262: move one collated sample to the GPU."""
    out = dict(batch)
    out['input'] = [batch['input'][0].to(device), batch['input'][1].to(device)]
    out['sdf'] = batch['sdf'].to(device)
    out['known'] = batch['known'].to(device) if batch.get('known') is not None else None
    out['hierarchy'] = [h.to(device) for h in batch['hierarchy']] if batch.get('hierarchy') is not None else None
    return out


OVERLAP_TARGETS = True
_target_streams = {}


def _target_stream(dev):
    s = _target_streams.get(dev)
    if s is None:
        s = _target_streams[dev] = torch.cuda.Stream(device=dev)
    return s


class StepPlan(object):
    """What a teacher-forced step needs from its batch alone: targets + loss weights, input coordinates with the
    encoder pyramid, site / index lists of every generative level.  `ready` is recorded on the stream that built it."""
    __slots__ = ('batch', 'loss_weights', 'targets', 'weights', 'geometry', 'ready')


class GeometryPrefetcher(object):
    """Teacher-forced training, one batch ahead: the geometry of a step (every host read-back it has) is a function of
    the batch, so batch i+1's StepPlan is built on a second stream while batch i occupies the GPU, and the main stream
    never drains at a step boundary.

        pre = GeometryPrefetcher(model)
        for i, batch in enumerate(batches):
            train_step(model, opt, batch, lw, teacher_forced=True, prefetch=pre, next_batch=batches[i + 1])

    threaded=False (default) builds the next plan after the optimizer step has been issued (1.6 ms of host time per
    step at the BASELINE batch); threaded=True builds it on a worker thread started at the head of the step — correct and
    tested, but measured slower on this workload (7.8 vs 7.3–7.4 ms/step: the builder is Python-bound, its 3.3 ms under
    the GIL slow the issuing thread by more than the five read-back waits it hides).

    Memory discipline: plan tensors are allocated on the prefetch stream's pool and read by main-stream kernels of
    their step.  A plan is retained until the host has seen its step finish on the GPU (end-of-step events, `_retire`),
    only then may the pool hand its blocks to a new plan; the same wait keeps the host at most two steps ahead of the
    device.  The prefetch stream uses scratch lane 1 of the runtime (scn.metadata.lane): its workspace and count block
    are not shared with main-stream kernels."""

    def __init__(self, model, num_hierarchy_levels=4, truncation=3.0, weight_missing_geo=5.0, use_loss_masking=True,
                 threaded=None):
        self.model = model
        self.args = (num_hierarchy_levels, truncation, use_loss_masking, weight_missing_geo)
        self.threaded = (os.environ.get('SGNN_PREFETCH_THREAD', '0') == '1') if threaded is None else bool(threaded)
        self.stream = None
        self.pending = None            # the announced next batch's plan: {'plan' | 'error', 'thread'}
        self.issued = []               # [(end-of-step event on the main stream, plan of that step, status copy)]
        self.t_build = self.t_throttle = 0.0   # host seconds spent building plans / waiting for old steps
        self._status, self._n, self._rt = [], 0, None   # pinned copies of the training lane's status word, per step

    def build(self, batch, loss_weights):
        """StepPlan of `batch` on the prefetch stream (the calling thread blocks for its five read-backs only)."""
        from .scn.metadata import lane
        dev = batch['sdf'].device
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)    # (a high-priority stream measured no different)
        nl, trunc, masking, wgeo = self.args
        plan = StepPlan()
        plan.batch, plan.loss_weights = batch, np.array(loss_weights, copy=True)
        known = batch['known'] if masking else None
        with torch.cuda.device(dev), torch.cuda.stream(self.stream), lane(1):
            plan.targets, plan.weights = loss_util.compute_targets_and_weights(
                batch['sdf'], batch['hierarchy'], nl, trunc, masking, known, wgeo, batch['input'][0])
            plan.geometry = self.model.plan_geometry(batch['input'][0], loss_weights, int(batch['sdf'].shape[0]),
                                                     plan.targets[1])
            plan.ready = torch.cuda.Event()
            plan.ready.record(self.stream)
        return plan

    def _retire(self, keep):
        """Wait for all but the `keep` most recently issued steps to leave the GPU and let go of their plans."""
        import time
        t0 = time.perf_counter()
        while len(self.issued) > keep:
            ev, _plan, status = self.issued.pop(0)
            ev.synchronize()
            if status is not None and int(status[0]):
                # the training stream has no read-back of its own any more: input errors its kernels flagged during
                # that step (duplicate / out-of-range sites) surface here, two steps later at most
                self._rt.raise_status(int(status[0]))
        self.t_throttle += time.perf_counter() - t0

    def _start(self, batch, loss_weights):
        import threading
        import time
        box = {}

        def work():
            t0 = time.perf_counter()
            try:
                box['plan'] = self.build(batch, loss_weights)
            except BaseException as e:          # re-raised by take() on the training thread
                box['error'] = e
            self.t_build += time.perf_counter() - t0

        if not self.threaded:
            work()
            return box
        th = threading.Thread(target=work, name='sgnn-geometry-prefetch', daemon=True)
        box['thread'] = th
        th.start()
        return box

    @staticmethod
    def _result(box):
        if box.get('thread') is not None:
            box['thread'].join()
        if 'error' in box:
            raise box['error']
        return box['plan']

    def take(self, batch, loss_weights):
        """The plan announced for `batch` (built now when it was not announced); the main stream waits for it."""
        box, self.pending = self.pending, None
        plan = self._result(box) if box is not None else None
        if plan is None or plan.batch is not batch or not np.array_equal(plan.loss_weights, loss_weights):
            plan = self._result(self._start(batch, loss_weights))
        torch.cuda.current_stream(batch['sdf'].device).wait_event(plan.ready)
        return plan

    def announce(self, next_batch, loss_weights):
        """Head of a step, right after take(): start building `next_batch`'s plan on the worker thread."""
        if self.threaded and next_batch is not None:
            self._retire(1)     # the pool may recycle plans older than the previous step's from here on
            self.pending = self._start(next_batch, loss_weights)

    def step_issued(self, dev, plan, next_batch, loss_weights):
        """Call after the optimizer step of a step has been issued: remember when it ends, keep its plan until then."""
        from .scn.metadata import runtime
        self._rt = runtime(dev)
        if not self._status:
            self._status = [torch.zeros(1, dtype=torch.int64).pin_memory() for _ in range(4)]
        status = self._status[self._n % len(self._status)]
        self._n += 1
        status.copy_(self._rt.state[1:2], non_blocking=True)       # rides the training stream, no synchronisation
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self.issued.append((ev, plan, status))
        if not self.threaded and next_batch is not None:
            self._retire(1)
            self.pending = self._start(next_batch, loss_weights)
        self._retire(2)


def train_step(model, optimizer, batch, loss_weights, num_hierarchy_levels=4, truncation=3.0,
               use_log_transform=True, weight_missing_geo=5.0, use_loss_masking=True, grad_sync=None,
               teacher_forced=False, prefetch=None, next_batch=None):
    """batch: device-resident dict in scene_dataloader.collate layout.  Returns (loss, losses, outputs).
    prefetch / next_batch (teacher-forced only): a GeometryPrefetcher and the batch of the following step."""
    inputs = batch['input']
    known = batch['known'] if use_loss_masking else None
    dev = batch['sdf'].device

    def targets():    # three launches on the device (sgnn_loss_targets); the batch tensors stay untouched
        return loss_util.compute_targets_and_weights(batch['sdf'], batch['hierarchy'], num_hierarchy_levels, truncation,
                                                     use_loss_masking, known, weight_missing_geo, inputs[0])

    optimizer.zero_grad(set_to_none=True)
    if teacher_forced and prefetch is not None:
        plan = prefetch.take(batch, loss_weights)
        prefetch.announce(next_batch, loss_weights)
        (tgt_sdf, tgt_occs, tgt_hier), weights = plan.targets, plan.weights
        output_sdf, output_occs = model(inputs, loss_weights, batch_size=int(batch['sdf'].shape[0]), teacher=tgt_occs,
                                        geometry=plan.geometry)
    elif teacher_forced:
        # generative masks come from the target occupancy pyramid: the targets are needed before the model runs
        (tgt_sdf, tgt_occs, tgt_hier), weights = targets()
        output_sdf, output_occs = model(inputs, loss_weights, batch_size=int(batch['sdf'].shape[0]), teacher=tgt_occs)
    elif dev.type == 'cuda' and OVERLAP_TARGETS:
        # targets and loss weights depend on the batch only (a dozen dense element-wise / pooling passes over the
        # (B,1,D,D,D) volumes): they run on a second stream underneath the encoder, whose small launches leave the
        # memory system idle; the loss waits for them
        side = _target_stream(dev)
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            (tgt_sdf, tgt_occs, tgt_hier), weights = targets()
        output_sdf, output_occs = model(inputs, loss_weights, batch_size=int(batch['sdf'].shape[0]))
        main.wait_stream(side)
    else:
        (tgt_sdf, tgt_occs, tgt_hier), weights = targets()
        output_sdf, output_occs = model(inputs, loss_weights, batch_size=int(batch['sdf'].shape[0]))
    loss, losses = loss_util.compute_loss(output_sdf, output_occs, tgt_sdf, tgt_occs, tgt_hier, loss_weights,
                                          truncation, use_log_transform, weight_missing_geo, inputs[0],
                                          use_loss_masking, known, weights=weights)
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    optimizer.step()
    if teacher_forced and prefetch is not None:
        prefetch.step_issued(dev, plan, next_batch, loss_weights)
    return loss, losses, (output_sdf, output_occs)


def get_loss_weights(it, num_hierarchy_levels, num_iters_per_level, factor_l1_loss):
    """Curriculum of train.py:203-231 (level k fades in every num_iters_per_level iterations)."""
    w = np.zeros(num_hierarchy_levels + 1, dtype=np.float32)
    cur_level = it // num_iters_per_level
    if cur_level > num_hierarchy_levels:
        w.fill(1)
        w[-1] = factor_l1_loss
        return w
    w[:cur_level + 1] = 1.0
    step_factor = 20
    fade_amount = max(1.0, min(100, num_iters_per_level // step_factor))
    fade_level = it % num_iters_per_level
    cur_weight = 0.0
    if fade_level >= num_iters_per_level - fade_amount + step_factor:
        fade_level_step = (fade_level - num_iters_per_level + fade_amount) // step_factor
        cur_weight = float(fade_level_step) / float(fade_amount // step_factor)
    l1_weight = 0.0
    if cur_level + 1 < num_hierarchy_levels:
        w[cur_level + 1] = cur_weight
    elif cur_level < num_hierarchy_levels:
        l1_weight = factor_l1_loss * cur_weight
    else:
        l1_weight = 1.0
    w[-1] = l1_weight
    return w
