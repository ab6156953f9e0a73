"""The captured step against the HIP graph executor's stream assignment (profiles/r06y_graph_executor.txt).

A replayed step overlaps its weight gradients with the backward chain only while no training-stream kernel sits on the
lane's stream: ONE such node (captured after the last fork, in front of the join) holds the whole lane back until the chain
is done — 5.7 -> 6.7 ms at bench size, the state rounds 3-6 kept falling into.  The runtime writes its assignment out
(DEBUG_HIP_GRAPH_DOT_PRINT=1), so the structure can be checked without timing anything."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SNIPPET = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from sgnn_amd import _lib, synth
from sgnn_amd.model import GenModel
from sgnn_amd.train import GraphStep, to_device
dev = torch.device('cuda', 0)
if %r:
    buf = torch.zeros(64, dtype=torch.int64, device=dev)
    _lib.load().sgnn_stamp_enable(buf.data_ptr(), 64)
    _lib.STAMPS = True
torch.manual_seed(3)
model = GenModel(8, (32, 32, 32), 1, 16, 16, 4, True, True, 1, 1).to(dev)
batches = [to_device(synth.make_batch(2, (32, 32, 32), cfg=7, first_block=2 * j, occupancy=0.08), dev) for j in range(2)]
gs = GraphStep(model, lr=1e-3, headroom=1.6, settle=False)
lw = np.ones(5, dtype=np.float32)
for i in range(6):
    gs(batches[i %% 2], lw)
torch.cuda.synchronize()
assert gs.stats['captures'] >= 1 and gs.stats['replays'] >= 1, gs.stats
'''


def _dump(tmp_path, stamp_only):
    from benchlib import graphdot
    env = dict(os.environ, DEBUG_HIP_GRAPH_DOT_PRINT='1')
    if stamp_only:
        env['SGNN_STAMP_ONLY'] = stamp_only
    p = subprocess.run([sys.executable, '-c', SNIPPET % (ROOT, bool(stamp_only))], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    dumps = glob.glob(os.path.join(str(tmp_path), 'graph_*_dot_print_*'))
    if not dumps:
        pytest.skip('this HIP runtime wrote no graph dump (DEBUG_HIP_GRAPH_DOT_PRINT)')
    parsed = [graphdot.parse(d) for d in dumps]
    nodes, edges = max(parsed, key=lambda ne: len(ne[0]))           # the step (the other dump is torch's warm-up graph)
    if not nodes:
        pytest.skip('graph dump in an unknown format')
    return nodes, edges


@pytest.mark.gpu
def test_no_training_stream_kernel_sits_on_the_weight_gradient_lane(tmp_path):
    from benchlib import graphdot
    nodes, edges = _dump(tmp_path, None)
    lane, chain, offenders = graphdot.backward_lane_report(nodes, edges)
    assert lane is not None, 'no forked weight gradient in the captured step: the lane is off'
    assert chain is not None and lane != chain
    assert len({n.stream for n in nodes.values()}) == 2          # one lane: a third stream was measured to collide (r03t)
    assert not offenders, [(n.id, graphdot.short(n.name)) for n in offenders]
    # every BatchNorm backward of the chain on ONE stream, every forked weight gradient on the other
    assert {n.stream for n in nodes.values() if graphdot.short(n.name).startswith('k_bn_bwd_apply')} == {chain}


@pytest.mark.gpu
def test_the_check_sees_the_node_that_costs_the_overlap(tmp_path):
    """Negative control: the stamp in front of the final join (4 bytes written by one thread on the training stream) is the
    node that costs a bench-size step 0.9 ms; the reader must flag it."""
    from benchlib import graphdot
    nodes, edges = _dump(tmp_path, 'join<')
    lane, chain, offenders = graphdot.backward_lane_report(nodes, edges)
    assert lane is not None and [graphdot.short(n.name) for n in offenders] == ['k_stamp']
