#!/bin/bash
# kernel trace of the graph-replayed step: summary + per-launch table of one step.  Usage: gpu_r3_trace.sh TAG [bench flags]
TAG=$1; shift
mkdir -p gpurun_out
export TMPDIR=/tmp; D=/tmp/prof_$TAG; rm -rf $D; ROOT=$(pwd)
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python $ROOT/bench.py --steps 16 --warmup 12 --no-cpu-baseline --no-traffic --no-other-mode "$@" > $ROOT/gpurun_out/${TAG}_prof.out 2> $ROOT/gpurun_out/${TAG}_prof.err)
tail -2 $ROOT/gpurun_out/${TAG}_prof.err
F=$(find $D -name '*kernel_stats.csv' | head -1); T=$(find $D -name '*kernel_trace.csv' | head -1)
[ -n "$F" ] && cp $F gpurun_out/${TAG}_kernel_stats.csv
[ -n "$T" ] && python scripts/trace_graph.py $T ${WHICH:--9} gpurun_out/${TAG}_step_launches.csv > gpurun_out/${TAG}_trace_summary.txt 2>&1
head -12 gpurun_out/${TAG}_trace_summary.txt
head -3 $T
