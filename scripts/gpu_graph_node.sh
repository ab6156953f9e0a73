#!/bin/bash
# the cost of a near-empty dependent kernel in a replayed graph: un-profiled (wall clock) and as rocprofv3 reports it
ROOT=$(pwd); mkdir -p gpurun_out
python scripts/bench_graph_node.py 512 200 | tee gpurun_out/r05_graph_node.txt
export TMPDIR=/tmp; D=/tmp/prof_node; rm -rf $D
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python $ROOT/scripts/bench_graph_node.py 512 20 > /tmp/node_prof.out 2>&1)
echo "--- the same program under rocprofv3 --kernel-trace --stats:" | tee -a gpurun_out/r05_graph_node.txt
cat /tmp/node_prof.out | tee -a gpurun_out/r05_graph_node.txt
F=$(find $D -name '*kernel_stats.csv' | head -1)
[ -n "$F" ] && head -4 $F | cut -c1-200 | tee -a gpurun_out/r05_graph_node.txt
