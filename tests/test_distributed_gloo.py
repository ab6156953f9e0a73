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
