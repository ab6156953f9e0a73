export TMPDIR=/tmp; D=/tmp/prof_nl; rm -rf $D; ROOT=$(pwd)
(cd /tmp && SGNN_SIDE_LANE=0 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $D -o r -- python $ROOT/bench.py --no-cpu-baseline --no-traffic --no-other-mode --steps 20 --warmup 5 > $ROOT/gpurun_out/r05q_nolane_prof.out 2> $ROOT/gpurun_out/r05q_nolane_prof.err)
T=$(find $D -name '*kernel_trace.csv' | head -1)
[ -n "$T" ] && python scripts/trace_graph.py $T 280 gpurun_out/r05q_nolane_step_launches.csv > gpurun_out/r05q_nolane_trace_summary.txt 2>&1
head -6 gpurun_out/r05q_nolane_trace_summary.txt
tail -2 gpurun_out/r05q_nolane_prof.out | cut -c1-200
