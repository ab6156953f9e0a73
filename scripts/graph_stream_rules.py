"""Which stream does the HIP graph executor give a node?  Tiny captured graphs (near-empty kernels: sgnn_add over 256 floats on
distinct buffers so that every node is identifiable by its position), dumped by the runtime itself (DEBUG_HIP_GRAPH_DOT_PRINT=1,
set here before HIP initialises) and read back with benchlib/graphdot.py; each structure is also timed (200 replays).

  chain(a) = a dependent kernels on the capture stream ("main");  lane(b) = b dependent kernels on the side stream
  fork order: 'lane-first' = side.wait_stream(main); lane kernels; then main's next kernel      (the pyramid lane's capture order)
              'main-first' = event recorded on main; main's next kernel; THEN the lane kernels    (the weight-gradient lane's order)

    python scripts/graph_stream_rules.py
"""
import glob
import os
import sys
import tempfile
import time

os.environ['DEBUG_HIP_GRAPH_DOT_PRINT'] = '1'
work = tempfile.mkdtemp(prefix='graphdot_')
os.chdir(work)
import torch                                                         # noqa: E402
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sgnn_amd import _lib                                            # noqa: E402
from sgnn_amd._lib import ptr                                        # noqa: E402
from benchlib import graphdot                                        # noqa: E402

dev = torch.device('cuda', 0)
_lib.require_gpu()
one = torch.full((256,), 1e-6, device=dev)
bufs = [torch.zeros(256, device=dev) for _ in range(2)]


def k(which):
    _lib.call('sgnn_add', ptr(bufs[which]), ptr(one), 256, ptr(bufs[which]))


def build(structure):
    """structure: list of ('main', n) | ('fork-lane-first', nlane) | ('fork-main-first', nlane, nmain_before_lane) | ('join',)"""
    side = torch.cuda.Stream(device=dev)
    s = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    roles = []
    with torch.cuda.stream(s):
        k(0); k(1)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        main = torch.cuda.current_stream()
        for step in structure:
            if step[0] == 'main':
                for _ in range(step[1]):
                    k(0); roles.append('M')
            elif step[0] == 'fork-lane-first':
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    for _ in range(step[1]):
                        k(1); roles.append('L')
            elif step[0] == 'fork-main-first':
                ev = torch.cuda.Event(); ev.record(main)
                for _ in range(step[2]):
                    k(0); roles.append('M')
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    for _ in range(step[1]):
                        k(1); roles.append('L')
            elif step[0] == 'join':
                main.wait_stream(side)
    return g, roles


def timed(g, reps=200):
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def show(tag, structure):
    before = set(glob.glob('graph_*_dot_print_*'))
    g, roles = build(structure)
    new = sorted(set(glob.glob('graph_*_dot_print_*')) - before, key=os.path.getmtime)
    us = timed(g)
    line = '?'
    if new:
        nodes, edges = graphdot.parse(new[-1])
        order = sorted(nodes.values())
        if len(order) == len(roles):
            line = ' '.join('%s%d' % (r, n.stream) for r, n in zip(roles, order))
        else:
            line = 'dump has %d nodes, capture %d' % (len(order), len(roles))
    print('%-64s %7.1f us   %s' % (tag, us, line))


N = 6
print('# role+stream per node in capture order (M = main chain, L = lane); time of one replay')
show('chain of 24', [('main', 24)])
for nl in (1, 4):
    show('lane-first fork of %d, 6 main, join, 6 main' % nl, [('main', N), ('fork-lane-first', nl), ('main', N), ('join',), ('main', N)])
    show('main-first fork of %d (1 main before lane), 5 main, join, 6 main' % nl,
         [('main', N), ('fork-main-first', nl, 1), ('main', N - 1), ('join',), ('main', N)])
show('lane-first x3 (fork 4 / 6 main / join)', [('main', 2)] + [('fork-lane-first', 4), ('main', N), ('join',)] * 3 + [('main', 2)])
show('main-first x3 (fork 4 / 6 main / join)', [('main', 2)] + [('fork-main-first', 4, 1), ('main', N - 1), ('join',)] * 3 + [('main', 2)])
show('lane-first x3, no join until the end', [('main', 2)] + [('fork-lane-first', 2), ('main', N)] * 3 + [('join',), ('main', 2)])
show('main-first x3, no join until the end', [('main', 2)] + [('fork-main-first', 2, 1), ('main', N - 1)] * 3 + [('join',), ('main', 2)])
show('main-first x3, no join until the end, 1 main before the join', [('main', 2)] + [('fork-main-first', 2, 1), ('main', N - 1)] * 3 + [('main', 1), ('join',), ('main', 2)])
show('main-first x3, LAST fork lane-only (no main after it), join', [('main', 2)] + [('fork-main-first', 2, 1), ('main', N - 1)] * 2 + [('fork-lane-first', 2), ('join',), ('main', 2)])
show('… the same + 1 main between that fork and the join', [('main', 2)] + [('fork-main-first', 2, 1), ('main', N - 1)] * 2 + [('fork-lane-first', 2), ('main', 1), ('join',), ('main', 2)])
