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
"""
import argparse
import ctypes
import json
import os
import sys
import time

# Hardware queues per process (read at HIP start-up): the ROCm default, 4.  Round 2 asked for 8 (an own queue for the
# geometry-prefetch lane); round 3 measured that with more than 4 queues two concurrently active branches of the replayed
# graph can land on the same hardware pipe, which then time-slices them with 40-70 us stalls at every switch (17-19 instead
# of 5.8 ms per step with 8 or 16 queues, profiles/r03w_hw_queues.txt).  With 4 queues no branch placement showed it.
os.environ['GPU_MAX_HW_QUEUES'] = os.environ.get('SGNN_BENCH_HW_QUEUES', '4')   # (A/B: SGNN_BENCH_HW_QUEUES=8)

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable)
FP32_MFMA_PEAK_TF = 157.3    # v_mfma_f32_16x16x4_f32 (MI355X_MICROARCH.md)


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
                    help='CPU leg: torch-op oracle only (default: its C/OpenMP kernels for convolutions + rulebooks when built)')
    ap.add_argument('--cpu-blocks', type=int, default=8, help='blocks in the CPU-baseline sample')
    ap.add_argument('--cpu-threads', type=int, default=0, help='internal: thread count of a CPU-baseline child process')
    ap.add_argument('--teacher-forced', action='store_true',
                    help='generative masks from the target hierarchy instead of the predicted occupancy (per-level row counts '
                         'then do not depend on the random weights).  Default: the reference\'s sigmoid(pred) > 0.5 masks')
    ap.add_argument('--free-running', action='store_true', help='(default now; kept for older command lines)')
    ap.add_argument('--classic', action='store_true',
                    help='the classic eager step (host read-backs of the level sizes, one launch at a time from Python) '
                         'instead of the capacity-mode step replayed from a HIP graph')
    ap.add_argument('--headroom', type=float, default=1.3, help='capacity = measured rows x headroom (graph mode)')
    ap.add_argument('--converged-headroom', type=float, default=1.15,
                    help='head-room of the plan the timed region runs with: after --settle steps the row counts move less '
                         'than 0.1 %% per step, so the capacities can sit closer to the live counts than while the weights '
                         'are fresh (kernels are launched for the capacities: 6.65 -> 6.58 ms per step)')
    ap.add_argument('--settle', type=int, default=250,
                    help='untimed training steps BEFORE the warm-up steps (set-up, like building the model): with the '
                         'reference\'s masks the per-level row counts follow the weights — several-fold changes within '
                         'the first dozen optimizer steps of a fresh model, then +1 %% per step until they reach the '
                         'data\'s own occupancy (410 k of 424 k final sites) after ~250 steps (scripts/diag_drift.py); '
                         'the timed region should see that converged workload, not a point on the transient')
    ap.add_argument('--no-prefetch', action='store_true',
                    help='teacher-forced steps build their own geometry (5 read-backs at the head of the step) instead of '
                         'having it built one batch ahead on a second stream (train.GeometryPrefetcher)')
    ap.add_argument('--no-other-mode', action='store_true', help='skip the extra steps in the other mask mode (profiling runs)')
    ap.add_argument('--no-traffic', action='store_true', help='skip the rocprofv3 --pmc passes behind roofline.traffic')
    ap.add_argument('--traffic-probe', action='store_true', help='internal: launch the dominant kernel a few times (run under rocprofv3 --pmc)')
    return ap.parse_args()


def conv_alg_bytes(kind, n_out, cin, cout, K):
    """Algorithmic HBM bytes of one conv launch: feature slab read once, output written once,
    the K x n_out int32 rule table, the weights.  dW reads x and dy and writes K*cin*cout."""
    if kind == 0:
        return 4 * n_out * (cin + cout) + 4 * K * n_out + 4 * K * cin * cout
    return 4 * n_out * (cin + cout) + 4 * K * n_out + 4 * K * cin * cout


def collect_prof(lib, valid_ratio=None, row_map=None):
    """valid_ratio: {(n_out, K): fraction of the K x n_out table entries that are rules} measured on the run's own
    rulebooks; `flops` counts the rules only (F = 2 R Cin Cout), `flops_exec` every table entry.
    row_map (capacity mode): launch rows (= capacities) -> live rows; bytes and executed flops count the live rows."""
    valid_ratio = valid_ratio or {}
    row_map = row_map or {}
    n = lib.sgnn_prof_count()
    kind, cin, cout, K, flags = (ctypes.c_int() for _ in range(5))
    n_out = ctypes.c_int64()
    ms = ctypes.c_float()
    agg = {}
    for i in range(n):
        rc = lib.sgnn_prof_get(i, ctypes.byref(kind), ctypes.byref(n_out), ctypes.byref(cin), ctypes.byref(cout),
                               ctypes.byref(K), ctypes.byref(flags), ctypes.byref(ms))
        if rc != 0:
            continue
        key = (kind.value, cin.value, cout.value, K.value)
        a = agg.setdefault(key, {'launches': 0, 'ms': 0.0, 'bytes': 0.0, 'flops': 0.0, 'by_size': {}})
        live = row_map.get(n_out.value, n_out.value)
        # rules = ratio x (table entries of the launch); a stride-2 table (K = 8) holds one rule per fine row, i.e. the
        # fraction of real entries is unknown here and left at 1 (as before)
        fl = 2.0 * n_out.value * K.value * cin.value * cout.value * valid_ratio.get((n_out.value, K.value), float(live) / max(n_out.value, 1))
        fl_exec = 2.0 * live * K.value * cin.value * cout.value
        a['launches'] += 1
        a['ms'] += ms.value
        a['bytes'] += conv_alg_bytes(kind.value, live, cin.value, cout.value, K.value)
        a['flops'] += fl
        a['flops_exec'] = a.get('flops_exec', 0.0) + fl_exec
        # the same kernel serves levels of very different size: keep the launches apart by output rows (powers of 4)
        bucket = 0 if n_out.value <= 0 else int(np.floor(np.log(max(n_out.value, 1)) / np.log(4.0)))
        b = a['by_size'].setdefault(bucket, {'launches': 0, 'ms': 0.0, 'flops': 0.0, 'rows': 0})
        b['launches'] += 1
        b['ms'] += ms.value
        b['flops'] += fl
        b['rows'] += live
    return agg


def capacity_row_map(cap, live):
    """launch rows (capacities) -> live rows of every level a capacity-mode step touches."""
    m = {cap['input']: live['input']}
    for c, n in zip(cap['enc'], live['enc']):
        m.setdefault(c, n)
    for (k, pyr), (nk, npyr) in zip(cap['gen'], live['gen']):
        m.setdefault(k, nk)
        m.setdefault(8 * k, 8 * nk)
        for c, n in zip(pyr, npyr):
            m.setdefault(c, n)
    return m


def algorithmic_step(model, agg, n_prof_steps, row_map, valid_ratio):
    """sum of the ALGORITHMIC bytes and flops of one training step over all sparse operators with the
    run's own row counts N_l and rule counts R_l.  Convolutions (forward, data gradient, weight gradient) come from
    the profiled launch records; BatchNormReLU 12 N C forward + 20 N C backward; UnPooling / AddTable / JoinTable /
    skip-join / linear heads as row movement, forward + backward.  The dense 8^3 bottleneck is excluded."""
    from sgnn_amd.scn import program as P_
    conv_b = sum(a['bytes'] for a in agg.values()) / max(n_prof_steps, 1)
    conv_f = sum(a['flops'] for a in agg.values()) / max(n_prof_steps, 1)
    other_b, n_ops = 0.0, 0
    for prog in P_.programs_of(model):
        lev = getattr(prog, 'last_lev_n', None)
        if lev is None:
            continue
        rows = lambda b: float(row_map.get(int(lev[prog.bufs[b][0]]), int(lev[prog.bufs[b][0]])))
        ch = lambda b: prog.bufs[b][1] if b >= 0 else 0
        for o in prog.ops:
            t, in0, in1, out = o[0], o[1], o[2], o[3]
            n_ops += 1
            if t == P_.OP_BN:
                other_b += 32.0 * rows(out) * ch(out)
            elif t in (P_.OP_UNPOOL, P_.OP_ADD, P_.OP_JOIN):
                srcs = [b for b in (in0, in1) if b >= 0]
                other_b += 2 * (4.0 * sum(rows(b) * ch(b) for b in srcs) + 4.0 * rows(out) * ch(out) + 8.0 * rows(out))
            elif t == P_.OP_CONCAT_IN:
                other_b += 2 * (8.0 * rows(out) * ch(out) + 8.0 * rows(out))
            elif t == P_.OP_LINEAR:
                other_b += 2 * 4.0 * rows(out) * (ch(in0) + ch(out))
    return {'bytes': conv_b + other_b, 'flops': conv_f, 'conv_bytes': conv_b, 'other_bytes': other_b, 'sparse_ops': n_ops}


def cpu_info():
    model, cores = 'unknown', set()
    try:
        phys = core = None
        for line in open('/proc/cpuinfo'):
            k, _, v = line.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'model name':
                model = v
            elif k == 'physical id':
                phys = v
            elif k == 'core id':
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return model, (len(cores) or (os.cpu_count() or 1))


def cpu_baseline(args):
    """The oracle (CPU restatement of the reference algorithm: explicit rulebook, per-offset gather -> small GEMM ->
    scatter-add) timed on this host on a bounded sample of the same workload: one child process per thread count (all
    physical cores, and 1), median of 5 steps after 2 warm-up steps."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    nthreads = args.cpu_threads or torch.get_num_threads()
    torch.set_num_threads(nthreads)
    import model_oracle as mo
    import scn_oracle
    from scn_oracle import _fast
    from sgnn_amd import synth
    torch.manual_seed(0)
    nb = args.cpu_blocks
    m = mo.GenModel(8, (args.dim,) * 3, 1, 16, 16, 4, True, True, 1, 1)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    data = synth.make_batch(nb, (args.dim,) * 3, cfg=2, occupancy=args.occupancy)
    lw = np.ones(5, dtype=np.float32)
    # convolutions and 3x3x3 rulebooks through the oracle's C/OpenMP kernels when they are built (same algorithm as
    # the torch-op mode and held to it by tests/test_oracle_fast.py; ~80 % of the oracle's step is inside them)
    scn_oracle.FAST = bool(_fast.available and not args.cpu_torch_only)

    def step():
        t0 = time.time()
        t = mo.compute_targets(data['sdf'].clone(), [h.clone() for h in data['hierarchy']], 4, 3, True, data['known'])
        opt.zero_grad()
        osdf, oocc = m(data['input'], lw)
        loss, _ = mo.compute_loss(osdf, oocc, t[0], t[1], t[2], lw, 3, True, 5.0, data['input'][0], True, data['known'])
        loss.backward()
        opt.step()
        return time.time() - t0

    n_warm, n_timed = (1, 3) if nthreads == 1 else (2, 5)        # the single-thread leg is ~10 s per step
    for _ in range(n_warm):
        step()
    times = sorted(step() for _ in range(n_timed))
    med = times[len(times) // 2]
    how = ('convolutions (neighbour-table form, ONE OpenMP region per convolution over the output rows, pair / table '
           'lists cached per grid) + 3x3x3 rulebooks in C/OpenMP (oracle/csrc/scn_cpu.c, %d threads), BatchNorm / stride-2 '
           'rulebooks / glue / loss torch-CPU (%d threads)' % (_fast.threads(), nthreads)) if scn_oracle.FAST else \
        'torch-CPU ops only (%d threads)' % nthreads
    return {'value': nb / med, 'unit': 'blocks/s', 'cores': nthreads, 'kind': 'port', 's_per_step': round(med, 3),
            'sample': '%d synthetic %d^3 blocks (cfg 2 seeds), full GenModel targets+fwd+loss+bwd+Adam on the CPU oracle [%s], '
                      'median of %d timed steps after %d warm-up steps' % (nb, args.dim, how, n_timed, n_warm)}


def cpu_baseline_subprocess(args):
    """The CPU leg in its own processes (the GPU process is pinned to its GPU's NUMA node and OpenMP pools are sized
    at start-up): once on all physical cores — the reported baseline — and once on 1 thread."""
    import subprocess
    model, phys = cpu_info()
    res = {}
    runs = [('all_cores', phys), ('one_thread', 1)] + ([('16_threads', 16)] if phys > 16 else [])
    for tag, nt in runs:
        env = dict(os.environ)
        env['OMP_NUM_THREADS'] = str(nt)
        env['MKL_NUM_THREADS'] = str(nt)
        cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-only', '--dim', str(args.dim), '--occupancy',
               str(args.occupancy), '--cpu-blocks', str(args.cpu_blocks), '--cpu-threads', str(nt)] + \
              (['--cpu-torch-only'] if args.cpu_torch_only else [])
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
        got = None
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith('{'):
                got = json.loads(line)
                break
        if got is None:
            raise RuntimeError('cpu baseline leg failed: %s' % out.stderr[-400:])
        res[tag] = got
    # headline = the fastest of the measured thread counts (fork/join over 128 cores costs more than it buys on loops
    # this short: all-cores is reported next to it, as is the single thread)
    best = max(res, key=lambda k: res[k]['value'])
    cpu = dict(res[best])
    cpu.update({'cpu_model': model, 'physical_cores': phys, 'threads_of_headline': res[best]['cores'],
                'by_threads': dict((k, {'threads': v['cores'], 'value': round(v['value'], 4), 's_per_step': v['s_per_step']})
                                   for k, v in res.items())})
    return cpu


def measure_valid_ratios(step, i):
    """One extra (untimed) step with a hook on the rulebook builder: fraction of real rules per table, keyed by
    (rows, K).  A stride-2 table holds exactly one rule per fine site."""
    from sgnn_amd.scn import metadata as MD
    ratios, real = {}, MD.Grid.subm_table

    def hooked(self):
        fresh = self._nbr is None
        tab = real(self)
        if fresh and self.n:
            ratios[(self.n, 27)] = float((tab.view(27, self.ld)[:, :self.n] >= 0).sum().item()) / (27.0 * self.n)
        return tab
    MD.Grid.subm_table = hooked
    try:
        step(i)
        torch.cuda.synchronize()
    finally:
        MD.Grid.subm_table = real
    return ratios


def traffic_probe(args):
    """Internal (--traffic-probe, run under rocprofv3 --pmc): the dominant conv shape on this batch's input level."""
    from sgnn_amd import synth
    from sgnn_amd.scn import functions as F_
    from sgnn_amd.scn.metadata import Grid, coords_from_locs
    dev = torch.device('cuda', 0)
    cin, cout = (int(v) for v in os.environ.get('SGNN_PROBE_SHAPE', '16,16').split(','))
    data = synth.make_batch(args.batch, (args.dim,) * 3, cfg=2, occupancy=args.occupancy)
    g = Grid(coords_from_locs(data['input'][0], dev))
    tab = g.subm_table()
    x = torch.randn(g.n, cin, device=dev)
    w = torch.randn(27, cin, cout, device=dev) * 0.1
    for _ in range(6):
        F_.conv_fwd_raw(x, cin, w, 27, tab, g.ld, g.n, cout, 0, 0)
    torch.cuda.synchronize()
    print(json.dumps({'rows': g.n, 'rules': int((tab.view(27, g.ld)[:, :g.n] >= 0).sum().item())}))


def measure_traffic(args, dom_key):
    """HBM bytes per launch of the dominant conv class from PMC counters: two rocprofv3 passes (FETCH_SIZE / WRITE_SIZE,
    KiB) over the probe; gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide reads (MI355X_MICROARCH.md, HBM),
    so traffic = 2 * FETCH + WRITE.  Returns None when rocprofv3 is not usable."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which('rocprofv3') is None or dom_key is None:
        return None
    cin, cout = dom_key[1], dom_key[2]
    vals, rows, rules = {}, None, None
    env = dict(os.environ)
    env.update({'SGNN_PROBE_SHAPE': '%d,%d' % (cin, cout), 'TMPDIR': '/tmp', 'SGNN_NO_BIND': '1'})
    for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
        d = tempfile.mkdtemp(prefix='sgnn_pmc_', dir='/tmp')
        try:
            cmd = ['rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'p', '--',
                   sys.executable, os.path.abspath(__file__), '--traffic-probe', '--batch', str(args.batch), '--dim',
                   str(args.dim), '--occupancy', str(args.occupancy)]
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd='/tmp')
            for line in out.stdout.splitlines():
                if line.startswith('{'):
                    info = json.loads(line)
                    rows, rules = info['rows'], info['rules']
            got = []
            for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get('Counter_Name') == counter and 'k_conv_fwd' in r.get('Kernel_Name', ''):
                        got.append(float(r['Counter_Value']))
            if not got:
                return None
            vals[counter] = sum(got) / len(got)
        except (subprocess.SubprocessError, OSError, ValueError, KeyError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if rows is None:
        return None
    byts = (2.0 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024.0
    alg = 4.0 * rows * (cin + cout) + 4.0 * 27 * rows + 4.0 * 27 * cin * cout
    return {'bytes_per_launch': round(byts), 'algorithmic_bytes': round(alg), 'ratio': round(byts / alg, 3),
            'kernel': 'conv_fwd<%d,%d>K27' % (cin, cout), 'rows': rows, 'rules': rules,
            'FETCH_SIZE_KiB': round(vals['FETCH_SIZE'], 1), 'WRITE_SIZE_KiB': round(vals['WRITE_SIZE'], 1),
            'method': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes, kernel trace only), mean per launch of the '
                      'kernel on the batch\'s input level; traffic = 2*FETCH + WRITE (gfx950 FETCH_SIZE correction)'}


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


def main():
    args = parse()
    if args.traffic_probe:
        traffic_probe(args)
        return
    if args.cpu_baseline_only:
        if hasattr(os, 'sched_setaffinity'):
            try:
                os.sched_setaffinity(0, range(os.cpu_count()))     # undo an inherited NUMA pin
            except OSError:
                pass
        print(json.dumps(cpu_baseline(args)))
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # SGNN_BENCH_FORCE_DIST=1: take the data-parallel code path (process group, flat-gradient all-reduce between the
    # two halves of the replayed step) even with ONE rank — on a 1-GPU box this executes the RCCL branch the 8-GPU run
    # takes (tests/test_gpu_distributed.py)
    force_dist = os.environ.get('SGNN_BENCH_FORCE_DIST') == '1' and 'RANK' in os.environ
    dist_on = world > 1 or force_dist
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
    share = os.environ.get('SGNN_BENCH_SHARE_GPU') == '1' and world > 1

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
    teacher = bool(args.teacher_forced)
    classic = bool(args.classic)
    n_prof_steps = 0
    graph_info = None
    row_map = {}

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
        # the masks keep growing while the fresh weights train (310 k -> 430 k final sites over the first 200 steps): give
        # the plan its full head-room over the CURRENT counts and let the step be captured again before the measurement,
        # so that no re-plan (a few eager steps + a 0.3 s capture) falls into the W + K steps below
        gs.headroom = min(gs.headroom, max(1.05, args.converged_headroom)) if args.settle >= 200 else gs.headroom
        gs.replan()
        extra = 12          # fixed (every rank must take the same number of steps): eager, three stable snapshots, capture
        for i in range(extra):
            step(i)
        elapsed = timed(step, args.warmup, args.steps)
        graph_info = dict(gs.stats)
        graph_info['replay_host_ms_per_step'] = round(graph_info.pop('replay_host_ms') / max(gs.stats['replays'], 1), 3)
        graph_info['preconditioning_steps'] = args.settle + extra
        graph_info['headroom'] = round(float(gs.headroom), 3)
        graph_info['capacity'] = gs.capacity.describe()
        graph_info['live_rows'] = gs.capacity.read()
        live = graph_info['live_rows']
        row_map = capacity_row_map(graph_info['capacity'], live)
        levels = [args.batch * (args.dim // 8) ** 3] + [8 * k for k, _ in live['gen'][:-1]] + [live['gen'][-1][0]]
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
        valid = measure_valid_ratios(step, 0)
        pre = None
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
        holder = {}

        def hook(i, before):   # HIP events around every conv launch of every 4th timed step
            if i % 4:
                return
            if before:
                lib.sgnn_prof_resume()
            else:
                lib.sgnn_prof_disable()
                holder['n'] = holder.get('n', 0) + 1
        outs_box = {}

        def step_keep(i):
            outs_box['o'] = step(i)[2]
        elapsed = timed(step_keep, args.warmup, args.steps, hook)
        n_prof_steps = holder.get('n', 0)
        outs = outs_box['o']
        levels = [int(o[0].shape[0]) if len(o[0]) else 0 for o in outs[1]] + [int(outs[0][0].shape[0]) if len(outs[0][0]) else 0]
        valid = measure_valid_ratios(step, args.warmup + args.steps)

    # ---- comparison legs, same process, same box (fresh model each: same initial weights) ------------------------
    legs = {}
    if args.steps >= 20 and not args.no_other_mode and not share:
        k2 = max(10, args.steps // 3)
        P_.PERSISTENT_ARENAS = True
        if not classic:
            gs._drain()
            torch.cuda.synchronize()
            state = dict((k, v.detach().clone()) for k, v in model.state_dict().items())
            fresh_model = make_model

            def make_model():       # every comparison leg starts from the headline leg's weights: the same masks
                m = GenModel(8, (args.dim,) * 3, 1, 16, 16, 4, True, True, 1, 1).to(dev)
                m.load_state_dict(state)
                return m
            # (a) the classic eager path with the reference's masks: five host read-backs per step, ~680 launches issued
            #     from Python (what BENCH_r01 / BENCH_r02's `other_mask_mode` measured)
            m2 = make_model()
            o2 = make_optimizer(m2.parameters(), lr=1e-3)
            s2 = FlatGradAllReduce(m2.parameters()) if dist_on else None
            box2 = {}

            def step2(i):
                box2['o'] = train_step(m2, o2, batches[i % 2], lw, grad_sync=s2, teacher_forced=False)[2]
            el = timed(step2, 8, k2)
            o_ = box2['o']
            legs['classic_eager_free_running'] = {
                'steps': k2, 'value': round(args.batch * world * k2 / el, 2), 'ms_per_step': round(1e3 * el / k2, 3),
                'generated_sites_per_level': [int(o[0].shape[0]) if len(o[0]) else 0 for o in o_[1]] +
                                             [int(o_[0][0].shape[0]) if len(o_[0][0]) else 0]}
            # (b) BENCH_r02's headline: teacher-forced masks + geometry built one batch ahead on a second stream
            m3 = make_model()
            o3 = make_optimizer(m3.parameters(), lr=1e-3)
            s3 = FlatGradAllReduce(m3.parameters()) if dist_on else None
            p3 = GeometryPrefetcher(m3)
            el = timed(lambda i: train_step(m3, o3, batches[i % 2], lw, grad_sync=s3, teacher_forced=True, prefetch=p3,
                                            next_batch=batches[(i + 1) % 2]), 8, k2)
            legs['classic_eager_teacher_forced_prefetch'] = {'steps': k2, 'value': round(args.batch * world * k2 / el, 2),
                                                             'ms_per_step': round(1e3 * el / k2, 3)}
            del m2, o2, m3, o3, p3
            # (c) graph replay with teacher-forced masks (row counts independent of the weights)
            m4 = make_model()
            g4 = GraphStep(m4, lr=1e-3, teacher_forced=not teacher, headroom=args.headroom, settle=teacher,
                           grad_sync=flat_sync if dist_on else None, world_size=world)
            el = timed(lambda i: g4(batches[i % 2], lw), 8, k2)
            live4 = g4.capacity.read()
            legs['graph_teacher_forced' if not teacher else 'graph_free_running'] = {
                'steps': k2, 'value': round(args.batch * world * k2 / el, 2), 'ms_per_step': round(1e3 * el / k2, 3),
                'generated_sites_per_level': [args.batch * (args.dim // 8) ** 3] + [8 * k for k, _ in live4['gen'][:-1]] +
                                             [live4['gen'][-1][0]], 'stats': dict(g4.stats)}
            del m4, g4
            # (e) a point on the transient, for comparison with earlier rounds' free-running numbers: fresh weights,
            #     40 + 12 untimed steps, then k2 timed ones (final level ~260-320 k sites; BENCH_r02: 212 k)
            if not teacher:
                m6 = fresh_model()
                g6 = GraphStep(m6, lr=1e-3, headroom=max(args.headroom, 1.6), grad_sync=flat_sync if dist_on else None,
                               world_size=world)
                for i in range(40):
                    g6(batches[i % 2], lw)
                g6.replan()
                el = timed(lambda i: g6(batches[i % 2], lw), 12, k2)
                live6 = g6.capacity.read()
                legs['graph_free_running_early_in_training'] = {
                    'steps': k2, 'value': round(args.batch * world * k2 / el, 2), 'ms_per_step': round(1e3 * el / k2, 3),
                    'generated_sites_per_level': [args.batch * (args.dim // 8) ** 3] + [8 * k for k, _ in live6['gen'][:-1]] +
                                                 [live6['gen'][-1][0]], 'stats': dict(g6.stats)}
                del m6, g6
            # (d) the fixed cost of a step: the same graph-replayed step on ONE block per GPU
            if args.batch > 1:
                b1 = make_batches(1)
                m5 = make_model()
                g5 = GraphStep(m5, lr=1e-3, teacher_forced=teacher, headroom=max(args.headroom, 1.6), settle=False,
                               grad_sync=flat_sync if dist_on else None, world_size=world)
                el = timed(lambda i: g5(b1[i % 2], lw), 12, k2)
                legs['batch1'] = {'steps': k2, 'ms_per_step': round(1e3 * el / k2, 3), 'stats': dict(g5.stats)}
                del m5, g5, b1
    other = legs or None
    unprefetched = None
    if rank == 0:
        agg = collect_prof(lib, valid, row_map if not classic else None)
        dom_key, dom = max(agg.items(), key=lambda kv: kv[1]['ms']) if agg else (None, None)
        roof = None
        kernels = []
        for key, a in sorted(agg.items(), key=lambda kv: -kv[1]['ms']):
            kernels.append({'kernel': '%s<%d,%d>K%d' % ('conv_fwd' if key[0] == 0 else 'conv_dw', key[1], key[2], key[3]),
                            'launches': a['launches'], 'ms_total': round(a['ms'], 3),
                            'GBps': round(a['bytes'] / (a['ms'] * 1e-3) / 1e9, 1) if a['ms'] > 0 else None,
                            'TFLOPs': round(a['flops'] / (a['ms'] * 1e-3) / 1e12, 2) if a['ms'] > 0 else None})
        if dom is not None and dom['ms'] > 0:
            # the binding roof of the dominant conv class: time at the HBM roof (algorithmic bytes / 8 TB/s) vs time
            # at the fp32-MFMA roof (flops / 157.3 TFLOP/s); C >= 16 convolutions sit above the fp32 ridge
            gbs = dom['bytes'] / (dom['ms'] * 1e-3) / 1e9
            tfs = dom['flops'] / (dom['ms'] * 1e-3) / 1e12
            t_hbm, t_mfma = dom['bytes'] / (HBM_PEAK_GBS * 1e9), dom['flops'] / (FP32_MFMA_PEAK_TF * 1e12)
            if t_mfma >= t_hbm:
                roof = {'bound': 'mfma', 'achieved': round(tfs, 2), 'peak': FP32_MFMA_PEAK_TF, 'unit': 'TFLOP/s',
                        'frac': round(tfs / FP32_MFMA_PEAK_TF, 4)}
            else:
                roof = {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                        'frac': round(gbs / HBM_PEAK_GBS, 4)}
            # traffic: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; their own runs, kernel trace only) over a probe
            # that launches the dominant kernel shape on this batch's input level; per launch, gfx950 correction applied
            traffic = None
            if world == 1 and not args.no_traffic:
                traffic = measure_traffic(args, dom_key)
            roof.update({'traffic': (traffic or {}).get('bytes_per_launch'), 'traffic_detail': traffic,
                         'flops_counted': 'rules only (2 R Cin Cout); executed incl. empty table entries: %.2f TFLOP/s'
                                          % (dom.get('flops_exec', dom['flops']) / (dom['ms'] * 1e-3) / 1e12),
                         'kernel': kernels[0]['kernel'],
                         'avg_launch_us': round(1e3 * dom['ms'] / dom['launches'], 2), 'launches': dom['launches'],
                         'alg_GBps': round(gbs, 1), 'alg_frac_of_hbm_peak': round(gbs / HBM_PEAK_GBS, 4),
                         'TFLOPs': round(tfs, 2), 'frac_of_fp32_mfma_peak': round(tfs / FP32_MFMA_PEAK_TF, 4),
                         'conv_ms_per_step': round(sum(a['ms'] for a in agg.values()) / max(n_prof_steps, 1), 3),
                         'profiled_steps': n_prof_steps,
                         'timing': ('HIP events around every convolution launch of %d capacity-mode steps issued eagerly '
                                    'right after the timed region (same kernels and sizes as the replayed graph; events '
                                    'cannot be recorded inside a replay)' % n_prof_steps) if not classic else
                                   'HIP events around every convolution launch of every 4th timed step',
                         # `frac` above is over ALL launches of the kernel; split by level size it is throughput-bound
                         # only on the big levels and launch/latency-bound on the small ones
                         'by_level_size': [
                             {'mean_rows': int(b['rows'] / b['launches']), 'launches': b['launches'],
                              'avg_us': round(1e3 * b['ms'] / b['launches'], 1),
                              'share_of_kernel_time': round(b['ms'] / dom['ms'], 3),
                              'TFLOPs': round(b['flops'] / (b['ms'] * 1e-3) / 1e12, 2),
                              'frac_of_fp32_mfma_peak': round(b['flops'] / (b['ms'] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF, 4)}
                             for _, b in sorted(dom['by_size'].items(), reverse=True) if b['ms'] > 0],
                         'top_kernels': kernels[:6]})
        step_ms = 1e3 * elapsed / args.steps
        if roof is not None:
            alg = algorithmic_step(model, agg, n_prof_steps, row_map, valid)
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
            cpu = cpu_baseline_subprocess(args)
        total_blocks = args.batch * world * args.steps
        res = {
            'metric': 'TSDF blocks/sec fwd+bwd (64^3@5% occ, bs32)', 'value': round(total_blocks / elapsed, 2),
            'unit': 'blocks/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'configs[1]: full SG-NN 4-level GenModel (643735 params, random init), %d synthetic '
                                   '%d^3 TSDF surface blocks per GPU at ~%.0f%% occupancy, compute_targets+fwd+loss+bwd+Adam; %s; %s'
                                   % (args.batch, args.dim, 100 * args.occupancy,
                                      'generative masks teacher-forced from the target hierarchy (row counts independent of '
                                      'the random weights)' if teacher else
                                      'generative masks = sigmoid(predicted occupancy) > 0.5 as in the reference (torch/model.py:233,322)',
                                      'classic eager step (host read-backs of the level sizes)' if classic else
                                      'capacity mode: row counts stay on the device, the whole step is ONE replayed HIP graph '
                                      '(train.GraphStep)'),
                       'host_cpus_bound': (len(bound) if bound else None), 'global_batch': args.batch * world, 'input_sites_per_batch': n_sites,
                       'generated_sites_per_level': levels, 'parallelism': 'dp%d' % world,
                       'geometry': ('built one batch ahead on a second stream during the previous step (once per step; '
                                    'train.GeometryPrefetcher)' if pre is not None else
                                    'built inside its own step' + ('' if classic else ' (inside the graph)')),
                       'graph': graph_info,
                       'ranks_in_process_group': (dist.get_world_size() if dist_on else 1),
                       'collective': ('%s all-reduce of the flat gradient buffer (%d floats + segment flags) between the two '
                                      'graph halves' % (backend, 643735)) if (dist_on and not classic) else backend},
            'roofline': roof, 'cpu_baseline': cpu, 'other_legs': other,
            'launches_per_step': (graph_info or {}).get('library_launches_per_step'),
            'batch1_ms': ((other or {}).get('batch1') or {}).get('ms_per_step'),
        }
        if cpu:
            res['gpu_over_cpu'] = round(res['value'] / cpu['value'], 1)
        print(json.dumps(res))
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
