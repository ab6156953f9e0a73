"""Workload of the IN-STEP counter passes (run under rocprofv3 --pmc by benchlib/pmc.py or scripts/pmc_step.sh):
the graph-replayed training step of bench.py's headline (configs[1], capacity mode, reference masks) — `--settle` untimed
steps, a re-plan, then `--replays` replayed steps.  Prints one JSON line with the live row counts.  rocprofv3's counter
collection serialises the dispatches; counters are per dispatch, so that does not matter for them."""
import argparse, json, os, sys
os.environ.setdefault('GPU_MAX_HW_QUEUES', '4')
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgnn_amd import synth
from sgnn_amd.model import GenModel
from sgnn_amd.train import GraphStep, to_device

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--dim', type=int, default=64)
ap.add_argument('--occupancy', type=float, default=0.05)
ap.add_argument('--settle', type=int, default=60)
ap.add_argument('--replays', type=int, default=8)
args = ap.parse_args()
dev = torch.device('cuda', 0)
torch.manual_seed(1234)
model = GenModel(8, (args.dim,) * 3, 1, 16, 16, 4, True, True, 1, 1).to(dev)
batches = [to_device(synth.make_batch(args.batch, (args.dim,) * 3, cfg=2, first_block=j * args.batch, occupancy=args.occupancy), dev)
           for j in range(2)]
lw = np.ones(5, dtype=np.float32)
gs = GraphStep(model, lr=1e-3, headroom=1.3)
for i in range(args.settle):
    gs(batches[i % 2], lw)
gs.replan()
for i in range(12 + args.replays):
    gs(batches[i % 2], lw)
torch.cuda.synchronize()
print(json.dumps({'stats': dict((k, v) for k, v in gs.stats.items() if k != 'replay_host_ms'), 'live_rows': gs.capacity.read()}))
