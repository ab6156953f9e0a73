"""Loader throughput: chunks/s from .sdfs files on local disk to a device-resident collated
batch, DeviceBatchLoader vs the reference-style struct.unpack loader (oracle/data_oracle.py) on the host cores.

  python scripts/bench_loader.py [--files 64] [--batch 32] [--dim 64]
Prints one JSON line.  File images come from the page cache after the first pass (the disk is not the subject)."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
from sgnn_amd import data, synth, _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--files', type=int, default=64)
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--dim', type=int, default=64)
ap.add_argument('--passes', type=int, default=5)
ap.add_argument('--cpu_files', type=int, default=4)
args = ap.parse_args()

tmp = tempfile.mkdtemp(prefix='sgnn_loader_')
files = []
for i in range(args.files):
    p = os.path.join(tmp, 'c%03d.sdfs' % i)
    synth.write_chunk(p, (args.dim,) * 3, 5000 + i, occupancy=0.05)
    files.append(p)
file_bytes = sum(os.path.getsize(p) for p in files)

loader = data.DeviceBatchLoader(files, args.batch, 3.0)
for _ in loader:                      # warm-up: page cache, pinned buffers, kernels
    pass
torch.cuda.synchronize()
t0 = time.perf_counter()
nb = 0
for _ in range(args.passes):
    for b in loader:
        nb += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
chunks = nb * args.batch

# decode only (staged batch already pinned): the device part on its own
staged = loader._stage(files[:args.batch])
for _ in range(3):
    loader._decode(*staged)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(20):
    loader._decode(*staged)
torch.cuda.synchronize()
dec = (time.perf_counter() - t1) / 20
out_bytes = sum(t.numel() * t.element_size() for t in
                [loader._decode(*staged)[k] for k in ('sdf', 'known')] + loader._decode(*staged)['hierarchy'] +
                loader._decode(*staged)['input'])

import data_oracle  # noqa: E402  (cpu_baseline leg)
t2 = time.perf_counter()
data_oracle.collate([data_oracle.sample_chunk(p, 3.0, 4) for p in files[:args.cpu_files]])
cpu = (time.perf_counter() - t2) / args.cpu_files

print(json.dumps({
    'metric': 'chunks/s, .sdfs file image -> device-resident collated batch', 'value': round(chunks / dt, 1),
    'unit': 'chunks/s', 'batch': args.batch, 'dim': args.dim, 'mean_file_bytes': file_bytes // args.files,
    'decode_ms_per_batch': round(dec * 1e3, 3),
    'decode_GBps': round((staged[1]['bytes'] + out_bytes) / dec / 1e9, 1),
    'staged_bytes_per_batch': staged[1]['bytes'], 'device_bytes_written_per_batch': out_bytes,
    'cpu_baseline': {'value': round(1.0 / cpu, 2), 'unit': 'chunks/s', 'cores': 1, 'kind': 'port',
                     'sample': '%d chunks through oracle/data_oracle.py (struct.unpack per scalar, as '
                               'data_util.py:63-117)' % args.cpu_files}}))
