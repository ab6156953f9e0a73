"""Summarise a HIP graph dump (DEBUG_HIP_GRAPH_DOT_PRINT=1 writes graph_<pid>_dot_print_<k> at instantiation; reader:
benchlib/graphdot.py): which stream the executor assigned every node to, in node order, as runs of equal stream; the
cross-stream edges; per-stream node counts; the weight-gradient lane check of tests/test_gpu_graph_streams.py.

    python scripts/graph_dot_summary.py gpurun_out/dot/graph_210_dot_print_2 [--runs]
"""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from benchlib.graphdot import parse, short, backward_lane_report      # noqa: E402

nodes, edges = parse(sys.argv[1])
order = sorted(nodes.values())
per = collections.Counter(n.stream for n in order)
cross = [(a, b) for a, b in edges if nodes[a].stream != nodes[b].stream]
print('# %d nodes, %d edges, %d cross-stream edges, %d nodes that signal; nodes per stream: %s' % (
    len(order), len(edges), len(cross), sum(1 for n in order if n.signals), dict(sorted(per.items()))))
lane, chain, bad = backward_lane_report(nodes, edges)
print('# backward pass: weight-gradient lane on stream %s, chain on stream %s, chain kernels on the lane\'s stream: %s'
      % (lane, chain, [(n.id, short(n.name)) for n in bad] or 'none'))
if '--runs' in sys.argv:
    print('# runs of consecutive nodes (capture order) on one stream: first_id  stream  count  first kernel .. last kernel')
    i = 0
    while i < len(order):
        j = i
        while j + 1 < len(order) and order[j + 1].stream == order[i].stream:
            j += 1
        print('%6d  s%-2d %4d  %s .. %s' % (order[i].id, order[i].stream, j - i + 1, short(order[i].name), short(order[j].name)))
        i = j + 1
fam = collections.defaultdict(collections.Counter)
for n in order:
    fam[short(n.name)][n.stream] += 1
print('# kernel -> nodes per stream')
for k in sorted(fam, key=lambda k: -sum(fam[k].values()))[:40]:
    print('  %-28s %s' % (k, dict(sorted(fam[k].items()))))
