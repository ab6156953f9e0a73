#!/bin/bash
# kernel trace of the default bench's headline: per-launch table of one replayed step of the converged workload
TAG=$1
mkdir -p gpurun_out
export TMPDIR=/tmp; D=/tmp/prof_$TAG; rm -rf $D; ROOT=$(pwd)
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python $ROOT/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-traffic --no-other-mode > $ROOT/gpurun_out/${TAG}_prof.out 2> $ROOT/gpurun_out/${TAG}_prof.err)
T=$(find $D -name '*kernel_trace.csv' | head -1)
[ -n "$T" ] && python scripts/trace_graph.py $T 290 gpurun_out/${TAG}_step_launches.csv > gpurun_out/${TAG}_trace_summary.txt 2>&1
head -70 gpurun_out/${TAG}_trace_summary.txt
