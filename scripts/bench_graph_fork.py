"""How does a replayed HIP graph execute a fork?  (round 6: every experiment that re-timed the weight-gradient forks, added one
cross-stream edge per program, or moved the lane's join behind Adam cost the step the same +0.8 ms — the size of losing the
lane's overlap altogether.)  A chain of N small dependent kernels on the capture stream; ONE long kernel forked onto a second
stream behind chain kernel `fork_at` and joined behind chain kernel `join_at`.  A DAG executor takes max(chain, long kernel);
an executor that synchronises per topological level takes their sum.   python scripts/bench_graph_fork.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sgnn_amd import _lib
from sgnn_amd._lib import ptr

dev = torch.device('cuda', 0)
_lib.require_gpu()
N = 200
small = torch.zeros(256, device=dev); one = torch.full((256,), 1e-6, device=dev)
big_n = 1 << 26                                              # 256 MB in + 256 MB out per launch of the long kernel
big = torch.zeros(big_n, device=dev); big1 = torch.full((big_n,), 1e-6, device=dev)


def chain(k):
    for _ in range(k):
        _lib.call('sgnn_add', ptr(small), ptr(one), 256, ptr(small))


def long_kernel(reps):
    for _ in range(reps):
        _lib.call('sgnn_add', ptr(big), ptr(big1), big_n, ptr(big))


def timed(g, reps=50):
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def build(fork_at, join_at, long_reps, with_long=True, with_chain=True):
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        chain(4); long_kernel(1)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        if with_chain:
            chain(fork_at)
        if with_long:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                long_kernel(long_reps)
        if with_chain:
            chain(join_at - fork_at)
        if with_long:
            torch.cuda.current_stream().wait_stream(side)
        if with_chain:
            chain(N - join_at)
    return g


t_chain = timed(build(0, 0, 0, with_long=False))
t_long = timed(build(0, 0, 1, with_chain=False))
print('chain of %d dependent small kernels alone: %.0f us;  the long kernel alone: %.0f us' % (N, t_chain, t_long))
for fork_at, join_at in ((1, N), (1, N // 2), (N // 2, N), (1, 2), (N // 2, N // 2 + 1)):
    t = timed(build(fork_at, join_at, 1))
    print('fork behind chain kernel %3d, join behind %3d: %.0f us   (max %.0f, sum %.0f)'
          % (fork_at, join_at, t, max(t_chain, t_long), t_chain + t_long))
# many forks: one long kernel cut in 8 pieces, each forked behind its own chain kernel, ONE join at the end (the weight-gradient lane)
side = torch.cuda.Stream(device=dev)
small_big = big[:big_n // 8]; small_big1 = big1[:big_n // 8]
g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(device=dev)
with torch.cuda.graph(g, stream=s):
    for i in range(8):
        chain(N // 8)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            _lib.call('sgnn_add', ptr(small_big), ptr(small_big1), big_n // 8, ptr(small_big))
    torch.cuda.current_stream().wait_stream(side)
    chain(1)
print('8 forks (one eighth of the long kernel each, every %d chain kernels), one join at the end: %.0f us' % (N // 8, timed(g)))

# structural cost alone: the forked kernels are as small as the chain's (nothing to overlap, nothing to contend for)
def tiny_forks(nforks, join_each):
    side = torch.cuda.Stream(device=dev)
    t2 = torch.zeros(256, device=dev)
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(device=dev)
    with torch.cuda.graph(g, stream=s):
        per = N // max(nforks, 1)
        for i in range(nforks):
            chain(per)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _lib.call('sgnn_add', ptr(t2), ptr(one), 256, ptr(t2))
            if join_each:
                torch.cuda.current_stream().wait_stream(side)
        chain(N - per * nforks)
        if nforks and not join_each:
            torch.cuda.current_stream().wait_stream(side)
        chain(1)
    return timed(g)


print('chain of %d + 1 small kernels, no fork: %.0f us' % (N, tiny_forks(0, False)))
for nf in (1, 8, 50):
    print('  + %2d forks of ONE small kernel each, one join at the end: %.0f us;  joined right after each fork: %.0f us'
          % (nf, tiny_forks(nf, False), tiny_forks(nf, True)))


# the weight-gradient lane's shape and its alternatives: fork i's kernel must be finished (a) at the end, (b) before fork i + 1
# is taken (the side stream then never holds more than one pending kernel), (c) `lag` chain kernels after its fork
def lane(nforks, mode, lag=0, side_elems=256):
    side = torch.cuda.Stream(device=dev)
    t2 = torch.zeros(max(side_elems, 256), device=dev); o2 = torch.full((max(side_elems, 256),), 1e-6, device=dev)
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(device=dev)
    per = N // nforks
    with torch.cuda.graph(g, stream=s):
        pending = False
        for i in range(nforks):
            if mode == 'before_next_fork' and pending:
                torch.cuda.current_stream().wait_stream(side)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _lib.call('sgnn_add', ptr(t2), ptr(o2), side_elems, ptr(t2))
            pending = True
            if mode == 'lag':
                chain(lag)
                torch.cuda.current_stream().wait_stream(side)
                chain(per - lag)
            else:
                chain(per)
        torch.cuda.current_stream().wait_stream(side)
        chain(1)
    return timed(g)


for nf in (8, 50):
    for elems, what in ((256, 'tiny side kernels'), (1 << 22, 'side kernels of 48 MB traffic (~12 us)')):
        print('%2d forks, %s: join at the end %.0f us | before the next fork %.0f us | 2 chain kernels after the fork %.0f us'
              % (nf, what, lane(nf, 'end', side_elems=elems), lane(nf, 'before_next_fork', side_elems=elems),
                 lane(nf, 'lag', 2, side_elems=elems)))


# the structures that cost the real step +0.8 ms (profiles/r06l_ab_endfork.txt, r06t): P "programs" of F forks each
def programs(P, F, variant, side_elems=1 << 20):
    side = torch.cuda.Stream(device=dev)
    t2 = torch.zeros(side_elems, device=dev); o2 = torch.full((side_elems,), 1e-6, device=dev)
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(device=dev)
    per = max(N // (P * F), 1)
    with torch.cuda.graph(g, stream=s):
        for p in range(P):
            for i in range(F):
                side.wait_stream(torch.cuda.current_stream())                      # fork
                with torch.cuda.stream(side):
                    _lib.call('sgnn_add', ptr(t2), ptr(o2), side_elems, ptr(t2))   # "weight gradient"
                chain(per)
            if variant == 'end_edge':                                              # the lane's reduce waits for the program's last kernel
                side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _lib.call('sgnn_add', ptr(t2), ptr(o2), 256, ptr(t2))              # "reduce"
            if variant == 'join_per_program':
                torch.cuda.current_stream().wait_stream(side)
        if variant == 'late_join':
            chain(2)                                                                # "Adam" before the join
            torch.cuda.current_stream().wait_stream(side)
        else:
            torch.cuda.current_stream().wait_stream(side)
            chain(2)
    return timed(g)


for P, F in ((5, 4), (5, 10)):
    print('%d programs x %d forks: one join at the end %.0f us | + lane waits for each program\'s last kernel %.0f us | '
          'join per program %.0f us | join behind the last two chain kernels %.0f us'
          % (P, F, programs(P, F, 'base'), programs(P, F, 'end_edge'), programs(P, F, 'join_per_program'), programs(P, F, 'late_join')))
