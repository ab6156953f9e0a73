"""The data-parallel training step on the device: two ranks run the REAL train_step (GenModel on the
HIP kernels + FlatGradAllReduce + FastAdam) on different TSDF blocks; in step 1 one rank's generative hierarchy dies
early (no site predicted occupied at one refinement level), so the deeper levels produce no gradient there.  The
replicas must be bit-identical after every step, and the all-reduced gradient must equal the mean of the two
single-process gradients.  Both ranks share cuda:0 when only one GPU is visible (gloo carries the exchange; the
driver's multi-GPU bench runs the same code over RCCL), and `bench.py --gpus 2` must really start two ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS = (32, 32, 32)


def _setup(rank):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from util import param_fill
    from sgnn_amd import synth
    from sgnn_amd.model import GenModel
    from sgnn_amd.train import make_optimizer, to_device
    dev = torch.device('cuda', rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    model = param_fill(GenModel(8, DIMS, 1, 16, 16, 4, True, True, 1, 1), 5).train().to(dev)
    opt = make_optimizer(model.parameters(), lr=1e-3)
    batches = [to_device(synth.make_batch(2, DIMS, cfg=7, first_block=10 * it + 2 * rank, occupancy=0.08), dev)
               for it in range(3)]
    return dev, model, opt, batches


def _kill_level(active):
    """While `active[0]`, the second mask compaction of a forward (the first Refinement's) keeps nothing."""
    from sgnn_amd.scn import functions as F_
    real = F_.compact_sigmoid_plan
    calls = [0]

    def patched(logits, stride, n, coords_all, depth, teacher=None):
        sel, cnt, locs = real(logits, stride, n, coords_all, depth, teacher)
        calls[0] += 1
        if active[0] and calls[0] == 2:
            return sel[:0], 0, locs[:0]
        return sel, cnt, locs
    F_.compact_sigmoid_plan = patched
    return calls


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    dev, model, opt, batches = _setup(rank)
    from sgnn_amd.train import FlatGradAllReduce, train_step
    sync = FlatGradAllReduce(model.parameters())
    lw = np.ones(5, dtype=np.float32)
    active = [False]
    calls = _kill_level(active)
    grads_step0, reached = None, []
    for it in range(3):
        active[0] = (rank == 1 and it == 1)
        calls[0] = 0
        _, _, outs = train_step(model, opt, batches[it], lw, grad_sync=sync)
        reached.append([int(len(o[0])) for o in outs[1]])
        if it == 0:
            grads_step0 = [p.grad.detach().cpu().clone() for p in model.parameters()]
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu()
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1]), 'replicas diverged after step %d' % it
        assert torch.isfinite(flat).all()
    if rank == 0:
        torch.save(grads_step0, out)
    res = [None] * world
    dist.all_gather_object(res, reached)
    if rank == 0:
        torch.save({'grads': grads_step0, 'reached': res}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_train_step_lock_step(tmp_path):
    out = str(tmp_path / 'dp.pt')
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    # rank 1, step 1: the hierarchy stopped after the first refinement level (levels 2, 3 never ran)
    assert r['reached'][1][1][2:] == [0, 0] and all(v > 0 for v in r['reached'][0][1])
    assert all(v > 0 for v in r['reached'][1][0]) and all(v > 0 for v in r['reached'][1][2])
    # step 0: the synchronised gradient equals the mean of the two single-process gradients
    from sgnn_amd.train import train_step
    lw = np.ones(5, dtype=np.float32)
    singles = []
    for rank in range(2):
        _, model, opt, batches = _setup(rank)
        opt.zero_grad(set_to_none=True)
        train_step(model, opt, batches[0], lw, grad_sync=lambda: singles.append(
            [p.grad.detach().cpu().clone() for p in model.parameters()]))
    for g, a, b in zip(r['grads'], singles[0], singles[1]):
        want = (a + b) / 2
        assert torch.allclose(g, want, rtol=1e-5, atol=1e-7 * max(1.0, float(want.abs().max())))


def test_bench_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` (no launcher) must spawn two ranks and report n_gpus = 2 from the process group."""
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env['SGNN_BENCH_SHARE_GPU'] = '1'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2', '--batch', '2',
           '--dim', '32', '--no-cpu-baseline']
    # Two ranks SHARING one device is a functional stand-in, not a supported layout.  Round 3 saw one such run in ~10 die
    # inside the HIP runtime; round 4 ran this command 69 times in a row without a failure (profiles/r04k_flake_hunt.txt) —
    # which still allows a true failure rate of a few per cent.  So: ONE retry, only for a death by signal
    # (negative return code; a Python error fails at once), every failed attempt leaves its stderr under
    # gpurun_out/flake/, and two crashes in a row fail the test loudly.
    for attempt in range(2):
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        if p.returncode == 0:
            break
        d = os.path.join(ROOT, 'gpurun_out', 'flake')
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, 'test_bench_gpus_2.attempt%d.err' % attempt), 'w') as f:
            f.write('returncode %d\n' % p.returncode + p.stderr)
        if p.returncode > 0 and 'Signals.SIG' not in p.stderr and 'exitcode  : -' not in p.stderr:
            break
    assert p.returncode == 0, 'attempt %d: rc %d\n%s' % (attempt, p.returncode, p.stderr[-3000:])
    line = [l for l in p.stdout.splitlines() if l.startswith('{')][-1]
    res = json.loads(line)
    assert res['n_gpus'] == 2 and res['config']['ranks_in_process_group'] == 2
    assert res['config']['global_batch'] == 4 and res['value'] > 0


def test_bench_rccl_branch_runs_with_one_rank():
    """the code the 8-GPU driver run takes — init_process_group('nccl', device_id=...), the flat
    gradient buffer all-reduced through RCCL between the two halves of the replayed step — executed here with ONE rank
    under the driver's launcher form."""
    env = dict(os.environ)
    env['SGNN_BENCH_FORCE_DIST'] = '1'
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    port = 36500 + (os.getpid() % 2000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '6', '--warmup', '4',
           '--batch', '2', '--dim', '32', '--no-cpu-baseline', '--no-traffic', '--no-other-mode']
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][-1])
    assert res['n_gpus'] == 1 and res['config']['ranks_in_process_group'] == 1
    assert res['config']['collective'].startswith('nccl'), res['config']['collective']
    g = res['config']['graph']
    assert g['captures'] >= 1 and g['replays'] >= 1 and g['overflows'] == 0 and res['value'] > 0


def _worker_graph(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from util import param_fill
    from sgnn_amd import synth
    from sgnn_amd.model import GenModel
    from sgnn_amd.train import GraphStep, to_device
    dev = torch.device('cuda', rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    model = param_fill(GenModel(8, DIMS, 1, 16, 16, 4, True, True, 1, 1), 5).train().to(dev)
    batches = [to_device(synth.make_batch(2, DIMS, cfg=7, first_block=10 * it + 2 * rank, occupancy=0.08), dev)
               for it in range(2)]
    step = GraphStep(model, lr=1e-3, headroom=1.6, settle=False, grad_sync=lambda flat: dist.all_reduce(flat), world_size=world)
    lw = np.ones(5, dtype=np.float32)
    for it in range(6):
        loss = step(batches[it % 2], lw)
        torch.cuda.synchronize()
        flat = step.opt.flat_p.detach().cpu()
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        assert torch.equal(both[0], both[1]), 'replicas diverged after step %d' % it
        assert torch.isfinite(flat).all() and np.isfinite(float(loss))
    res = [None] * world
    dist.all_gather_object(res, dict(step.stats))
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_graph_step_lock_step(tmp_path):
    """Data-parallel GraphStep: each rank replays [targets .. backward] and [Adam] as two graphs around ONE all-reduce of
    the flat gradient buffer (+ segment flags); replicas stay bit-identical, every rank replays from the third step on."""
    out = str(tmp_path / 'dpg.pt')
    port = 37500 + (os.getpid() % 2000)
    mp.spawn(_worker_graph, args=(2, port, out), nprocs=2, join=True)
    stats = torch.load(out)
    for s in stats:
        assert s['captures'] >= 1 and s['replays'] >= 3, s
