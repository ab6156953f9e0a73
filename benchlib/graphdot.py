"""Reader for the HIP runtime's graph dumps.  With DEBUG_HIP_GRAPH_DOT_PRINT=1 in the environment (before HIP initialises)
hipGraphInstantiate writes `graph_<pid>_dot_print_<k>` into the working directory: one DOT node per graph node with the
kernel name, the executor's StreamId for it and whether it must signal, plus the edges.  That assignment decides what overlaps
in a replayed step (profiles/r06y_graph_executor.txt): the FIRST-captured child of a node inherits its stream, every other
child goes to another stream, and each stream runs its nodes in order — so a training-stream kernel captured right after a
fork whose first child was a lane kernel queues behind everything the lane has been given so far."""
import collections
import re

Node = collections.namedtuple('Node', 'id name stream signals')


def short(name):
    """Kernel family of a (mangled) node label: `_Z13k_chain_countPKi...` -> `k_chain_count`."""
    m = re.match(r'_Z\d+(k_[a-z0-9_]+)', name)
    if m:
        return re.sub(r'\d+$', '', m.group(1))       # (a by-value struct argument leaves its length digits behind the name)
    m = re.search(r'(k_[a-z0-9_]+)', name)
    return m.group(1) if m else name[:40]


def parse(path):
    """-> (nodes by id, edges as (parent id, child id))"""
    text = open(path).read()
    nodes = {}
    for m in re.finditer(r'"graph_\d+_node_(\d+)"\[[^\]]*?label="\d+\n([^\n]*)\nStreamId:(\d+)\nSignalIsRequired: (\w+)', text):
        nodes[int(m.group(1))] = Node(int(m.group(1)), m.group(2), int(m.group(3)), m.group(4) == 'true')
    edges = [(int(a), int(b)) for a, b in re.findall(r'node_(\d+)"\s*->\s*"graph_\d+_node_(\d+)"', text)]
    return nodes, [(a, b) for a, b in edges if a in nodes and b in nodes]


LANE_KERNELS = ('k_conv_dw', 'k_dw_reduce', 'k_expand_weights_bwd')


def backward_lane_report(nodes, edges):
    """The weight-gradient lane of the backward pass: lane kernels = k_conv_dw* with two parents (a fork) and what follows
    them on their stream.  Returns (lane stream, training stream, offenders): offenders are the nodes from the first fork
    to the end of the graph that sit on the lane's stream without being lane kernels.  One such node — a training-stream
    kernel captured after the last fork and in front of the join — keeps the whole lane from starting until the training
    stream has finished (the +0.8 ms state of a replayed step, profiles/r06y_graph_executor.txt)."""
    parents = collections.defaultdict(list)
    for a, b in edges:
        parents[b].append(a)
    forks = [n for n in nodes.values() if short(n.name).startswith('k_conv_dw') and len(parents[n.id]) >= 2]
    if not forks:
        return None, None, []
    lane = collections.Counter(n.stream for n in forks).most_common(1)[0][0]
    chain = collections.Counter(n.stream for n in nodes.values() if short(n.name).startswith('k_bn_bwd_apply')).most_common(1)
    first = min(n.id for n in forks)
    bad = [n for n in nodes.values() if n.id >= first and n.stream == lane and not short(n.name).startswith(LANE_KERNELS)]
    return lane, (chain[0][0] if chain else None), bad
