"""Diagnostic (needs GPU): step time, allocator and GC state over a long teacher-forced run with the geometry prefetcher."""
import os, sys, time, gc
os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from sgnn_amd import synth
from sgnn_amd.model import GenModel
from sgnn_amd.scn import program as P_
from sgnn_amd.train import train_step, to_device, make_optimizer, GeometryPrefetcher
P_.PERSISTENT_ARENAS = True
torch.manual_seed(1234)
dev = torch.device('cuda', 0)
use = '--no-prefetch' not in sys.argv
if '--no-gc' in sys.argv: gc.disable()
m = GenModel(8, (64,) * 3, 1, 16, 16, 4, True, True, 1, 1).to(dev)
opt = make_optimizer(m.parameters(), lr=1e-3)
batches = [to_device(synth.make_batch(32, (64,) * 3, cfg=2, first_block=j * 32), dev) for j in range(2)]
lw = np.ones(5, dtype=np.float32)
pre = GeometryPrefetcher(m) if use else None
N = int(os.environ.get('STEPS', '200'))
torch.cuda.synchronize(); t0 = time.perf_counter(); tb = 0.0
for i in range(N):
    train_step(m, opt, batches[i % 2], lw, teacher_forced=True, prefetch=pre, next_batch=batches[(i + 1) % 2] if use else None)
    if (i + 1) % 20 == 0:
        torch.cuda.synchronize(); t1 = time.perf_counter()
        st = torch.cuda.memory_stats()
        print('steps %3d-%3d: %.3f ms/step  build %.2f ms/step  alloc %.2f GB reserved %.2f GB segments %d cudaMalloc %d  gc %s'
              % (i - 19, i, 1e3 * (t1 - t0) / 20, 1e3 * ((pre.t_build - tb) / 20 if pre else 0), st['allocated_bytes.all.current'] / 2**30,
                 st['reserved_bytes.all.current'] / 2**30, st['segment.all.current'], st['num_device_alloc'], gc.get_count()), flush=True)
        tb = pre.t_build if pre else 0
        t0 = time.perf_counter()
if '--referrers' in sys.argv:
    plans = [o for o in gc.get_objects() if type(o).__name__ == 'StepPlan']
    print('live StepPlan objects:', len(plans))
    import types
    for p in plans[:2]:
        for r in gc.get_referrers(p):
            if r is plans: continue
            print('  referrer:', type(r).__name__, (list(r.keys())[:12] if isinstance(r, dict) else (getattr(r, 'f_code', None) and r.f_code.co_name) or str(r)[:120]))
    big = [o for o in gc.get_objects() if torch.is_tensor(o) and o.is_cuda and o.numel() >= 32 * 64 ** 3]
    print('live big cuda tensors:', len(big))
    seen = 0
    for t in big[4:10]:
        for r in gc.get_referrers(t):
            if r is big: continue
            print('  tensor', tuple(t.shape), 'referrer:', type(r).__name__, (list(r.keys())[:8] if isinstance(r, dict) else str(r)[:100]))
