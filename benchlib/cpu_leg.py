"""cpu_baseline of bench.py: the oracle (CPU restatement of the reference algorithm — explicit rulebook, per-offset gather ->
small GEMM -> scatter-add; kind "port", the real SparseConvNet CPU build is not available) timed on this host's cores on a
bounded sample of the same workload.

How it uses "all cores": the step's BatchNorm statistics, rulebook builds and glue are serial sections between OpenMP
regions, so ONE step does not scale past ~16 threads (round 3: 12.6 blocks/s at 16 threads, 1.95 at 128).  The reference
parallelises over samples; the unit that is independent end to end here is a REPLICA — a step over its own blocks with its own
BatchNorm statistics, exactly the partitioning the multi-GPU run uses (HISTORY.md section 6).  The all-cores figure therefore
runs R = cores / T replicas side by side (T threads each, pinned to disjoint cores, idle OpenMP threads sleeping) and reports
the aggregate blocks/s.  Measured on the 2 x 64-core EPYC 9575F of the GPU boxes (profiles/r04j_cpu_leg.txt): the aggregate
saturates at ~20 blocks/s from 8 replicas on (4 x 32 threads 16.0, 8 x 16: 18.5, 16 x 8: 19.9, 32 x 4: 20.1) while every
replica slows down 5-8x — the oracle's step is bound by the host's memory system (gathers over 92 k-row levels from 128
cores), not by core count; T = 8 is the default."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    """One replica: GenModel targets + forward + loss + backward + Adam on the CPU oracle, args.cpu_blocks blocks, on
    args.cpu_threads threads; median of the timed steps.  Prints / returns one JSON-able dict."""
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
    data = synth.make_batch(nb, (args.dim,) * 3, cfg=2, first_block=args.cpu_first_block, occupancy=args.occupancy)
    lw = np.ones(5, dtype=np.float32)
    # convolutions and 3x3x3 rulebooks through the oracle's C/OpenMP kernels when they are built (same algorithm as the
    # torch-op mode and held to it by tests/test_oracle_fast.py; ~80 % of the oracle's step is inside them)
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
    if args.cpu_barrier_dir:                                     # replicas of one measurement start their timed steps together
        open(os.path.join(args.cpu_barrier_dir, 'ready.%d' % os.getpid()), 'w').close()
        deadline = time.time() + 600.0
        while len(os.listdir(args.cpu_barrier_dir)) < args.cpu_replicas and time.time() < deadline:
            time.sleep(0.01)
        n_timed = 8
    t_begin = time.time()
    times = [step() for _ in range(n_timed)]
    wall = time.time() - t_begin
    med = sorted(times)[len(times) // 2]
    how = ('convolutions (neighbour-table form, ONE OpenMP region per convolution over the output rows, pair / table lists '
           'cached per grid) + 3x3x3 rulebooks in C/OpenMP (oracle/csrc/scn_cpu.c, %d threads), BatchNorm / stride-2 rulebooks / '
           'glue / loss torch-CPU (%d threads)' % (_fast.threads(), nthreads)) if scn_oracle.FAST else \
        'torch-CPU ops only (%d threads)' % nthreads
    return {'value': nb / med, 'unit': 'blocks/s', 'cores': nthreads, 'kind': 'port', 's_per_step': round(med, 3),
            'timed_steps': n_timed, 'timed_wall_s': round(wall, 3), 't_begin': t_begin, 't_end': t_begin + wall,
            'sample': '%d synthetic %d^3 blocks (cfg 2 seeds), full GenModel targets+fwd+loss+bwd+Adam on the CPU oracle [%s], '
                      'median of %d timed steps after %d warm-up steps' % (nb, args.dim, how, n_timed, n_warm)}


def _child(args, bench_path, nthreads, cpus=None, first_block=0, barrier_dir='', replicas=1):
    env = dict(os.environ)
    env['OMP_NUM_THREADS'] = str(nthreads)
    env['MKL_NUM_THREADS'] = str(nthreads)
    if cpus is not None:
        env['SGNN_CPU_LEG_CPUS'] = ','.join(str(c) for c in cpus)
    if replicas > 1:        # replicas side by side: idle OpenMP threads must sleep, not spin (two pools per process)
        env['GOMP_SPINCOUNT'] = '0'
        env['OMP_WAIT_POLICY'] = 'passive'
    cmd = [sys.executable, bench_path, '--cpu-baseline-only', '--dim', str(args.dim), '--occupancy', str(args.occupancy),
           '--cpu-blocks', str(args.cpu_blocks), '--cpu-threads', str(nthreads), '--cpu-first-block', str(first_block),
           '--cpu-barrier-dir', barrier_dir, '--cpu-replicas', str(replicas)] + \
          (['--cpu-torch-only'] if args.cpu_torch_only else [])
    return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)


def _result(p, timeout=900):
    out, err = p.communicate(timeout=timeout)
    for line in reversed(out.strip().splitlines()):
        if line.startswith('{'):
            return json.loads(line)
    raise RuntimeError('cpu baseline leg failed: %s' % err[-400:])


def cpu_baseline_subprocess(args, bench_path):
    """The CPU leg in its own processes (the GPU process is pinned to its GPU's NUMA node and OpenMP pools are sized at
    start-up): one replica on 1 thread and on T = min(16, cores) threads, then R = cores // T replicas side by side on
    disjoint cores — the all-cores figure and the headline."""
    model, phys = cpu_info()
    ncpu = os.cpu_count() or phys
    T = min(int(os.environ.get('SGNN_CPU_LEG_THREADS', '8')), phys)       # threads per replica
    R = max(1, phys // T)
    res = {}
    legs = [('one_thread', 1), ('one_replica_%d_threads' % T, T)] + ([('one_replica_16_threads', 16)] if phys >= 32 and T != 16 else [])
    for tag, nt in legs:
        if tag == 'one_thread' and os.environ.get('SGNN_CPU_LEG_SKIP_ONE') == '1':
            continue
        res[tag] = _result(_child(args, bench_path, nt))
    one = res['one_replica_%d_threads' % T]
    cpu = dict(one)
    if R > 1:
        # replicas pinned to disjoint core blocks (logical CPU ids 0..: cores first on this image's hosts), different blocks
        import shutil
        import tempfile
        bdir = tempfile.mkdtemp(prefix='sgnn_cpu_leg_')                  # file barrier: every replica starts its timed steps
        try:                                                             # when all of them are warmed up
            pin = os.environ.get('SGNN_CPU_LEG_PIN', '1') == '1'
            kids = [_child(args, bench_path, T, cpus=range(r * T, (r + 1) * T) if (pin and (r + 1) * T <= ncpu) else None,
                           first_block=r * args.cpu_blocks, barrier_dir=bdir, replicas=R) for r in range(R)]
            outs = [_result(k, 1500) for k in kids]
        finally:
            shutil.rmtree(bdir, ignore_errors=True)
        t0, t1 = min(o['t_begin'] for o in outs), max(o['t_end'] for o in outs)
        blocks = sum(args.cpu_blocks * o['timed_steps'] for o in outs)
        agg = blocks / (t1 - t0)
        res['all_cores'] = {'threads': R * T, 'replicas': R, 'threads_per_replica': T, 'value': round(agg, 3),
                            's_per_step': round(float(np.median([o['s_per_step'] for o in outs])), 3),
                            'window_s': round(t1 - t0, 2)}
        cpu = dict(outs[0])
        cpu.update({'value': agg, 'cores': R * T, 's_per_step': res['all_cores']['s_per_step'],
                    'sample': '%d replicas side by side, each: %s; aggregate = all blocks of all replicas\' timed steps / the '
                              'window from the first replica\'s first timed step to the last one\'s end (%.1f s); replicas '
                              'keep their own BatchNorm statistics, as the ranks of the multi-GPU run do'
                              % (R, outs[0]['sample'], t1 - t0)})
    for k in ('t_begin', 't_end', 'timed_wall_s', 'timed_steps'):
        cpu.pop(k, None)
    by = {}
    for k, v in res.items():
        by[k] = {'threads': v.get('threads', v.get('cores')), 'value': round(v['value'], 4), 's_per_step': v['s_per_step']}
        if 'replicas' in v:
            by[k]['replicas'] = v['replicas']
    cpu.update({'cpu_model': model, 'physical_cores': phys, 'threads_of_headline': cpu['cores'], 'by_threads': by})
    return cpu
