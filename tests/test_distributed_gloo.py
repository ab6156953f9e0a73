"""N>1 path on CPU: two gloo ranks, each with its own TSDF blocks, one flat gradient all-reduce
(sgnn_amd.train.FlatGradAllReduce) — result must equal the mean of the per-rank gradients, including a
parameter that one rank never reached (empty generative level)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3), torch.nn.Linear(3, 2))


def _local_grads(rank):
    m = _make_model()
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(7 + rank, 6, generator=g)
    h = m[2](m[1](m[0](x)))
    loss = (h ** 2).mean() if rank == 1 else (m[3](h) ** 2).mean()   # rank 1 never touches layer 3
    loss.backward()
    return m


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sgnn_amd.train import FlatGradAllReduce
    m = _local_grads(rank)
    FlatGradAllReduce(m.parameters())()
    if rank == 0:
        torch.save([p.grad.clone() for p in m.parameters()], out)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_two_ranks(tmp_path):
    out = str(tmp_path / 'g.pt')
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    ms = [_local_grads(r) for r in range(2)]
    for i, g in enumerate(got):
        parts = [list(m.parameters())[i].grad for m in ms]
        want = sum(torch.zeros_like(g) if p is None else p for p in parts) / 2
        assert torch.allclose(g, want, atol=1e-7), i


def test_single_process_is_a_noop():
    sys.path.insert(0, ROOT)
    from sgnn_amd.train import FlatGradAllReduce
    m = _local_grads(0)
    before = [p.grad.clone() for p in m.parameters()]
    FlatGradAllReduce(m.parameters())()
    for a, b in zip(before, (p.grad for p in m.parameters())):
        assert torch.equal(a, b)


def _worker_step(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sgnn_amd.train import FlatGradAllReduce, make_optimizer
    m = _make_model()
    opt = make_optimizer(m.parameters(), lr=1e-2)
    sync = FlatGradAllReduce(m.parameters())
    for it in range(3):                                   # step 1: rank 1 skips the last layer
        opt.zero_grad(set_to_none=True)
        g = torch.Generator().manual_seed(1000 * it + rank)
        h = m[2](m[1](m[0](torch.randn(5 + rank, 6, generator=g))))
        ((h ** 2).mean() if (rank == 1 and it == 1) else (m[3](h) ** 2).mean()).backward()
        sync()
        opt.step()
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    both = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    if rank == 0:
        torch.save(both, out)
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_stay_identical_after_optimizer_steps(tmp_path):
    """Data parallelism as bench.py runs it: per-rank batches, flat gradient all-reduce, FastAdam — the replicas'
    parameters must be bit-identical after every step, also when a rank did not reach a layer."""
    out = str(tmp_path / 'p.pt')
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_step, args=(2, port, out), nprocs=2, join=True)
    a, b = torch.load(out)
    assert torch.equal(a, b) and torch.isfinite(a).all()


def _worker_unreached(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sgnn_amd.train import FlatGradAllReduce, make_optimizer
    m = _make_model()
    opt = make_optimizer(m.parameters(), lr=1e-2, weight_decay=0.1)
    sync = FlatGradAllReduce(m.parameters())
    w3_before = m[3].weight.detach().clone()
    none_after_sync, steps = [], []
    for it in range(4):                       # curriculum: the last layer's loss weight is 0 for the first 3 steps
        opt.zero_grad(set_to_none=True)
        g = torch.Generator().manual_seed(1000 * it + rank)
        h = m[2](m[1](m[0](torch.randn(5 + rank, 6, generator=g))))
        ((h ** 2).mean() if it < 3 else (m[3](h) ** 2).mean()).backward()
        sync()
        none_after_sync.append(all(p.grad is None for p in m[3].parameters()))
        opt.step()
        st = opt.state.get(m[3].weight, {})
        steps.append(float(st['step']) if 'step' in st else 0.0)
        if it == 2:
            same = torch.equal(m[3].weight.detach(), w3_before)
    if rank == 0:
        torch.save({'none': none_after_sync, 'steps': steps, 'untouched': same}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_level_no_rank_reached_is_skipped_by_adam(tmp_path):
    """a parameter no rank produced a gradient for must keep grad=None after the all-reduce, so
    Adam's step counter and weight decay do not run for it — exactly like the single-process reference loop."""
    out = str(tmp_path / 'u.pt')
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_worker_unreached, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r['none'] == [True, True, True, False]
    assert r['steps'] == [0.0, 0.0, 0.0, 1.0]
    assert r['untouched']


def test_fast_adam_survives_load_state_dict():
    """the cached moment lists must be dropped when load_state_dict replaces the state tensors."""
    sys.path.insert(0, ROOT)
    from sgnn_amd.train import make_optimizer
    torch.manual_seed(0)
    m, ref = _make_model(), _make_model()
    o1, o2 = make_optimizer(m.parameters(), lr=1e-2), torch.optim.Adam(ref.parameters(), lr=1e-2)

    def run(model, opt, seed):
        opt.zero_grad(set_to_none=True)
        g = torch.Generator().manual_seed(seed)
        (model(torch.randn(4, 6, generator=g)) ** 2).mean().backward()
        opt.step()

    run(m, o1, 1)
    run(ref, o2, 1)
    o1.load_state_dict(o1.state_dict())        # replaces every state tensor with a copy
    run(m, o1, 2)
    run(ref, o2, 2)
    for a, b in zip(m.parameters(), ref.parameters()):
        assert torch.allclose(a, b, atol=1e-7)
    sd = o1.state_dict()['state']
    assert all(float(v['step']) == 2.0 for v in sd.values())
    assert all(v['exp_avg'].abs().sum() > 0 for v in sd.values())
