#!/bin/bash
# per-dispatch kernel trace of a few bench steps -> gpurun_out/${TAG}_trace.csv (+ memory copies)
TAG=$1; STEPS=${2:-6}
export TMPDIR=/tmp; D=/tmp/trace_$TAG; rm -rf $D; ROOT=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $D -o t -- python $ROOT/bench.py --steps $STEPS --warmup 6 --no-cpu-baseline --no-traffic --no-other-mode > $ROOT/gpurun_out/${TAG}_trace.out 2> $ROOT/gpurun_out/${TAG}_trace.err)
F=$(find $D -name '*kernel_trace.csv' | head -1); cp $F gpurun_out/${TAG}_trace.csv
M=$(find $D -name '*memory_copy_trace.csv' | head -1); [ -n "$M" ] && cp $M gpurun_out/${TAG}_memcpy.csv
ls -la gpurun_out/${TAG}_*
head -2 gpurun_out/${TAG}_trace.csv
