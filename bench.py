#!/usr/bin/env python
"""bench.py — headline metric of BASELINE.json on MI355X.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of synthetic input: compute_targets + GenModel forward +
hierarchical loss + backward + Adam step (train.py:245-268) on 32 synthetic 64^3 TSDF blocks at ~5 %
occupancy PER GPU (BASELINE.json configs[1]; weak scaling: N GPUs process N*32 independent blocks, the
only exchange is one flat fp32 gradient all-reduce over RCCL).  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.

This file holds the driver's contract: argument parsing, the process group, the timed headline leg and the JSON line.
The measurement legs around it live in benchlib/: cpu_leg.py (cpu_baseline), roofline.py (HIP-event roofline accounting),
pmc.py (rocprofv3 counter passes), legs.py (the same workload through the other execution modes).
"""
import argparse
import json
import os
import sys
import time

# Hardware queues per process (read at HIP start-up): the ROCm default, 4.  With more than 4 queues two concurrently
# active branches of the replayed graph can land on the same hardware pipe, which then time-slices them with 40-70 us
# stalls at every switch (17-19 instead of 5.8 ms per step with 8 or 16 queues, profiles/r03w_hw_queues.txt).
os.environ['GPU_MAX_HW_QUEUES'] = os.environ.get('SGNN_BENCH_HW_QUEUES', '4')   # (A/B: SGNN_BENCH_HW_QUEUES=8)

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib import HBM_PEAK_GBS, FP32_MFMA_PEAK_TF          # noqa: E402
from benchlib import roofline as RL                           # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--batch', type=int, default=32, help='blocks per GPU')
    ap.add_argument('--dim', type=int, default=64)
    ap.add_argument('--occupancy', type=float, default=0.05)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-baseline-only', action='store_true', help='internal: run the CPU leg alone, print its JSON')
    ap.add_argument('--cpu-torch-only', action='store_true',
                    help='CPU leg: torch-op oracle only (default: its C/OpenMP kernels for convolutions + rulebooks)')
    ap.add_argument('--cpu-blocks', type=int, default=8, help='blocks in one replica of the CPU-baseline sample')
    ap.add_argument('--cpu-threads', type=int, default=0, help='internal: thread count of a CPU-baseline child process')
    ap.add_argument('--cpu-first-block', type=int, default=0, help='internal: first synthetic block of a CPU replica')
    ap.add_argument('--cpu-barrier-dir', default='', help='internal: file barrier of the CPU replicas')
    ap.add_argument('--cpu-replicas', type=int, default=1, help='internal: replicas meeting at that barrier')
    ap.add_argument('--teacher-forced', action='store_true',
                    help='generative masks from the target hierarchy instead of the predicted occupancy (row counts then do '
                         'not depend on the weights).  Default: the reference\'s sigmoid(pred) > 0.5 masks')
    ap.add_argument('--free-running', action='store_true', help='(default; kept for older command lines)')
    ap.add_argument('--classic', action='store_true',
                    help='the classic eager step (host read-backs of the level sizes, one launch at a time from Python) '
                         'instead of the capacity-mode step replayed from a HIP graph')
    ap.add_argument('--headroom', type=float, default=1.3, help='capacity = measured rows x headroom (graph mode)')
    ap.add_argument('--converged-headroom', type=float, default=1.15,
                    help='head-room of the plan the timed region runs with: after --settle steps the row counts move less '
                         'than 0.1 %% per step, so the capacities can sit closer to the live counts')
    ap.add_argument('--settle', type=int, default=250,
                    help='untimed training steps BEFORE the warm-up steps: with the reference\'s masks the per-level row '
                         'counts follow the weights (several-fold changes in the first dozen steps of a fresh model, then '
                         '+1 %% per step until they reach the data\'s own occupancy after ~250 steps, scripts/diag_drift.py); '
                         'the timed region should see that converged workload, not a point on the transient')
    ap.add_argument('--no-prefetch', action='store_true', help='classic teacher-forced steps build their own geometry')
    ap.add_argument('--no-other-mode', action='store_true', help='skip the comparison legs (profiling runs)')
    ap.add_argument('--no-traffic', action='store_true', help='skip the rocprofv3 --pmc passes (roofline.traffic, counters_in_step)')
    ap.add_argument('--traffic-probe', action='store_true', help='internal: the isolated PMC probe (run under rocprofv3 --pmc)')
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run with one rank per GPU
    (the same command line the driver uses); rank 0's JSON line goes to this process's stdout."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and os.environ.get('SGNN_BENCH_SHARE_GPU') != '1':
        raise SystemExit('bench.py: --gpus %d but only %d GPU(s) are visible' % (args.gpus, n_dev))
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_child(args):
    """--cpu-baseline-only: one replica of the CPU leg on its own cores."""
    from benchlib import cpu_leg
    if hasattr(os, 'sched_setaffinity'):
        try:
            cpus = os.environ.get('SGNN_CPU_LEG_CPUS')
            os.sched_setaffinity(0, [int(c) for c in cpus.split(',')] if cpus else range(os.cpu_count()))
        except (OSError, ValueError):
            pass
    print(json.dumps(cpu_leg.cpu_baseline(args)))


def main():
    args = parse()
    if args.traffic_probe:
        from benchlib import pmc
        pmc.traffic_probe(args)
        return
    if args.cpu_baseline_only:
        cpu_child(args)
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # SGNN_BENCH_FORCE_DIST=1: take the data-parallel code path (process group, flat-gradient all-reduce between the two
    # halves of the replayed step) even with ONE rank — on a 1-GPU box this executes the RCCL branch the 8-GPU run takes
    force_dist = os.environ.get('SGNN_BENCH_FORCE_DIST') == '1' and 'RANK' in os.environ
    dist_on = world > 1 or force_dist
    share = os.environ.get('SGNN_BENCH_SHARE_GPU') == '1' and world > 1
    if dist_on:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if os.environ.get('SGNN_BENCH_SHARE_GPU') == '1':
            # test hook for 1-GPU boxes: every rank on a visible GPU (round robin), gradients exchanged through gloo
            # (RCCL refuses two ranks on one device); exercises the launch + lock-step path, not xGMI
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group('gloo')
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    elif torch.cuda.is_available():
        torch.cuda.set_device(0)
    dev = torch.device('cuda', torch.cuda.current_device())

    from sgnn_amd import _lib, synth
    from sgnn_amd.model import GenModel
    from sgnn_amd.scn import program as P_
    from sgnn_amd.train import (train_step, to_device, FlatGradAllReduce, make_optimizer, bind_to_device_numa,
                                GeometryPrefetcher, GraphStep)
    bound = bind_to_device_numa(dev)          # one process per GPU, on that GPU's NUMA node
    lib = _lib.load()
    _lib.require_gpu()
    lw = np.ones(5, dtype=np.float32)

    def make_batches(nb):
        # two distinct resident batches per rank, alternated, so no step sees cached results
        return [to_device(synth.make_batch(nb, (args.dim,) * 3, cfg=2, first_block=(rank * 2 + j) * nb,
                                           occupancy=args.occupancy), dev) for j in range(2)]

    def make_model():
        torch.manual_seed(1234)  # same initial weights on every rank and in every leg
        return GenModel(8, (args.dim,) * 3, 1, 16, 16, 4, True, True, 1, 1).to(dev)

    def timed(step, warmup, steps, hook=None):
        """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize brackets; max over ranks."""
        for i in range(warmup):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            if hook is not None:
                hook(i, True)
            step(warmup + i)
            if hook is not None:
                hook(i, False)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    def flat_sync(flat):        # data parallel: ONE all-reduce of the flat gradient buffer (+ its reached-flags tail)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    backend = dist.get_backend() if dist_on else None

    batches = make_batches(args.batch)
    n_sites = [int(b['input'][0].shape[0]) for b in batches]
    teacher, classic = bool(args.teacher_forced), bool(args.classic)
    n_prof_steps, graph_info, row_map, pre = 0, None, {}, None

    # ---- the headline leg ------------------------------------------------------------------------------------
    if not classic:
        # capacity mode: every level's row count stays on the device, the whole step (targets, forward, loss, backward,
        # Adam) is captured once in a HIP graph and replayed; masks = sigmoid(predicted occupancy) > 0.5 like the
        # reference (torch/model.py:233, 322) unless --teacher-forced
        model = make_model()
        gs = GraphStep(model, lr=1e-3, teacher_forced=teacher, headroom=args.headroom,
                       grad_sync=flat_sync if dist_on else None, world_size=world)

        def step(i):
            return gs(batches[i % 2], lw)
        for i in range(args.settle):
            step(i)
        # the masks keep growing while the fresh weights train: give the plan its full head-room over the CURRENT counts
        # and let the step be captured again before the measurement, so that no re-plan (a few eager steps + a 0.3 s
        # capture) falls into the W + K steps below
        gs.headroom = min(gs.headroom, max(1.05, args.converged_headroom)) if args.settle >= 200 else gs.headroom
        gs.replan()
        extra = 12          # fixed (every rank must take the same number of steps): eager, three stable snapshots, capture
        for i in range(extra):
            step(i)
        host0 = gs.stats['replay_host_ms']
        replays0 = gs.stats['replays']
        torch.cuda.reset_peak_memory_stats(dev)
        elapsed = timed(step, args.warmup, args.steps)
        graph_info = dict(gs.stats)
        # device memory of the replayed steps: `reserved` is what the process holds (the graph's private pool — every
        # intermediate of the captured step — static buffers, parameters / Adam state, workspaces); `allocated` counts only
        # tensors with a live owner outside the graph
        graph_info['reserved_gb'] = round(torch.cuda.memory_reserved(dev) / 1e9, 2)
        graph_info['peak_allocated_gb'] = round(torch.cuda.max_memory_allocated(dev) / 1e9, 2)
        graph_info.pop('replay_host_ms')
        # host time inside hipGraphLaunch per replay, over the warm-up + timed steps of THIS leg (hidden behind the GPU)
        graph_info['replay_host_ms_per_step'] = round((gs.stats['replay_host_ms'] - host0) /
                                                      max(gs.stats['replays'] - replays0, 1), 3)
        graph_info['preconditioning_steps'] = args.settle + extra
        graph_info['headroom'] = round(float(gs.headroom), 3)
        graph_info['capacity'] = gs.capacity.describe()
        graph_info['live_rows'] = live = gs.capacity.read()
        row_map = RL.capacity_row_map(graph_info['capacity'], live)
        from benchlib.legs import live_sites
        levels = live_sites(args, live)
        # roofline leg: the same capacity-mode steps issued eagerly (same kernels, same sizes) with HIP events around
        # every convolution launch — events cannot sit inside a replayed graph
        gs._drain()
        gs.stage, gs.use_graph = 2, False
        launches0 = lib.sgnn_launch_count()
        step(0)
        torch.cuda.synchronize()
        graph_info['library_launches_per_step'] = int(lib.sgnn_launch_count() - launches0)
        lib.sgnn_prof_enable(1 << 15)
        for i in range(4):
            step(i)
            n_prof_steps += 1
        torch.cuda.synchronize()
        lib.sgnn_prof_disable()
        valid = RL.measure_valid_ratios(step, 0)
        timing_note = ('HIP events around every convolution launch of %d capacity-mode steps issued eagerly right after the '
                       'timed region (same kernels and sizes as the replayed graph; events cannot be recorded inside a replay)'
                       % n_prof_steps)
    else:
        P_.PERSISTENT_ARENAS = True          # grow-only program arenas (a training loop never keeps two forward results)
        model = make_model()
        opt = make_optimizer(model.parameters(), lr=1e-3)
        sync = FlatGradAllReduce(model.parameters()) if dist_on else None
        pre = GeometryPrefetcher(model) if (teacher and not args.no_prefetch and not share) else None

        def step(i):
            return train_step(model, opt, batches[i % 2], lw, grad_sync=sync, teacher_forced=teacher, prefetch=pre,
                              next_batch=batches[(i + 1) % 2] if pre is not None else None)
        lib.sgnn_prof_enable(1 << 15)
        lib.sgnn_prof_disable()
        holder, outs_box = {}, {}

        def hook(i, before):   # HIP events around every conv launch of every 4th timed step
            if i % 4:
                return
            if before:
                lib.sgnn_prof_resume()
            else:
                lib.sgnn_prof_disable()
                holder['n'] = holder.get('n', 0) + 1

        def step_keep(i):
            outs_box['o'] = step(i)[2]
        elapsed = timed(step_keep, args.warmup, args.steps, hook)
        n_prof_steps = holder.get('n', 0)
        from benchlib.legs import sites
        levels = sites(outs_box['o'])
        valid = RL.measure_valid_ratios(step, args.warmup + args.steps)
        timing_note = 'HIP events around every convolution launch of every 4th timed step'

    # ---- comparison legs, same process, same box ---------------------------------------------------------------
    legs = None
    if args.steps >= 20 and not args.no_other_mode and not share and not classic:
        from benchlib.legs import comparison_legs
        P_.PERSISTENT_ARENAS = True
        gs._drain()
        torch.cuda.synchronize()
        state = dict((k, v.detach().clone()) for k, v in model.state_dict().items())
        legs = comparison_legs({'args': args, 'world': world, 'dev': dev, 'lw': lw, 'batches': batches,
                                'make_batches': make_batches, 'fresh_model': make_model, 'state': state, 'timed': timed,
                                'flat_sync': flat_sync, 'dist_on': dist_on, 'teacher': teacher})

    if rank == 0:
        agg = RL.collect_prof(lib, valid, row_map if not classic else None)
        dom_key, roof = RL.roofline_record(agg, n_prof_steps, timing_note)
        step_ms = 1e3 * elapsed / args.steps
        if roof is not None:
            if world == 1 and not args.no_traffic:
                from benchlib import pmc
                # traffic: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over a probe that launches the dominant
                # kernel shape on this batch's input level; counters_in_step: three passes over the replayed step itself
                traffic = pmc.measure_traffic(args, dom_key, os.path.abspath(__file__))
                roof['traffic'] = (traffic or {}).get('bytes_per_launch')
                roof['traffic_detail'] = traffic
                roof['counters_in_step'] = pmc.instep_counters(args)
            alg = RL.algorithmic_step(model, agg, n_prof_steps, row_map)
            roof['step'] = {
                'algorithmic_MB': round(alg['bytes'] / 1e6, 1), 'algorithmic_GFLOP': round(alg['flops'] / 1e9, 2),
                'hbm_frac': round(alg['bytes'] / (step_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
                'fp32_mfma_frac': round(alg['flops'] / (step_ms * 1e-3) / (FP32_MFMA_PEAK_TF * 1e12), 4),
                'sparse_ops_counted': alg['sparse_ops'],
                'definition': 'SURVEY 8d: sum over the sparse operators of one step (convolutions fwd + dX + dW from the '
                              'profiled launches with live rows and rules-only flops, BatchNorm 32 N C, row movement) / '
                              'ms_per_step / peak; dense 8^3 bottleneck excluded'}
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            from benchlib import cpu_leg
            cpu = cpu_leg.cpu_baseline_subprocess(args, os.path.abspath(__file__))
        masks = ('generative masks teacher-forced from the target hierarchy (row counts independent of the weights)' if teacher
                 else 'generative masks = sigmoid(predicted occupancy) > 0.5 as in the reference (torch/model.py:233,322)')
        mode = ('classic eager step (host read-backs of the level sizes)' if classic else
                'capacity mode: row counts stay on the device, the whole step is ONE replayed HIP graph (train.GraphStep)')
        res = {
            'metric': 'TSDF blocks/sec fwd+bwd (64^3@5% occ, bs32)',
            'value': round(args.batch * world * args.steps / elapsed, 2),
            'unit': 'blocks/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(step_ms, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'configs[1]: full SG-NN 4-level GenModel (643735 params, random init), %d synthetic %d^3 TSDF '
                                   'surface blocks per GPU at ~%.0f%% occupancy, compute_targets+fwd+loss+bwd+Adam; %s; %s'
                                   % (args.batch, args.dim, 100 * args.occupancy, masks, mode),
                       'host_cpus_bound': (len(bound) if bound else None), 'global_batch': args.batch * world,
                       'input_sites_per_batch': n_sites, 'generated_sites_per_level': levels, 'parallelism': 'dp%d' % world,
                       'geometry': ('built one batch ahead on a second stream during the previous step (once per step; '
                                    'train.GeometryPrefetcher)' if pre is not None else
                                    'built inside its own step' + ('' if classic else ' (inside the graph)')),
                       'graph': graph_info,
                       'ranks_in_process_group': (dist.get_world_size() if dist_on else 1),
                       'collective': ('%s all-reduce of the flat gradient buffer (%d floats + segment flags) between the two '
                                      'graph halves' % (backend, 643735)) if (dist_on and not classic) else backend},
            'roofline': roof, 'cpu_baseline': cpu, 'other_legs': legs,
            'launches_per_step': (graph_info or {}).get('library_launches_per_step'),
            'host_graph_launch_ms': (graph_info or {}).get('replay_host_ms_per_step'),
            'batch1_ms': ((legs or {}).get('batch1') or {}).get('ms_per_step'),
        }
        if cpu:
            res['gpu_over_cpu'] = round(res['value'] / cpu['value'], 1)
        print(json.dumps(res))
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
