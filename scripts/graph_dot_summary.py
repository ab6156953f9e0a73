"""Summarise a HIP graph dump (DEBUG_HIP_GRAPH_DOT_PRINT=1 writes graph_<pid>_dot_print_<k> at instantiation): which stream
the executor assigned every node to, in node order, as runs of equal stream; the cross-stream edges; per-stream node counts.

    python scripts/graph_dot_summary.py gpurun_out/dot/graph_210_dot_print_2 [--runs]
"""
import re
import sys
import collections

text = open(sys.argv[1]).read()
nodes = {}
for m in re.finditer(r'"(graph_\d+_node_(\d+))"\[[^\]]*?label="\d+\n([^\n]*)\nStreamId:(\d+)\nSignalIsRequired: (\w+)', text):
    nodes[m.group(1)] = (int(m.group(2)), m.group(3), int(m.group(4)), m.group(5) == 'true')
edges = re.findall(r'"(graph_\d+_node_\d+)"\s*->\s*"(graph_\d+_node_\d+)"', text)


def short(name):
    m = re.match(r'_Z\d+(k_[a-z0-9_]+)', name)
    if m:
        return m.group(1)
    m = re.search(r'(k_[a-z0-9_]+)', name)
    return m.group(1) if m else name[:40]


order = sorted(nodes.values())
per = collections.Counter(s for _, _, s, _ in order)
cross = [(a, b) for a, b in edges if a in nodes and b in nodes and nodes[a][2] != nodes[b][2]]
print('# %d nodes, %d edges, %d cross-stream edges, %d nodes that signal; nodes per stream: %s' % (
    len(order), len(edges), len(cross), sum(1 for n in order if n[3]), dict(sorted(per.items()))))
if '--runs' in sys.argv:
    print('# runs of consecutive nodes (capture order) on one stream: first_id  stream  count  first kernel .. last kernel')
    i = 0
    while i < len(order):
        j = i
        while j + 1 < len(order) and order[j + 1][2] == order[i][2]:
            j += 1
        print('%6d  s%-2d %4d  %s .. %s' % (order[i][0], order[i][2], j - i + 1, short(order[i][1]), short(order[j][1])))
        i = j + 1
fam = collections.defaultdict(collections.Counter)
for _, name, s, _ in order:
    fam[short(name)][s] += 1
print('# kernel -> nodes per stream')
for k in sorted(fam, key=lambda k: -sum(fam[k].values()))[:40]:
    print('  %-28s %s' % (k, dict(sorted(fam[k].items()))))
