#!/bin/bash
# Round-3 experiments: graph replay under different HIP-graph / side-lane settings, and a kernel trace of the replay.
TAG=$1
mkdir -p gpurun_out
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 30 --warmup 12 --no-cpu-baseline --no-traffic --no-other-mode --teacher-forced $EXTRA \
      > gpurun_out/${TAG}_${name}.json 2> gpurun_out/${TAG}_${name}.err
  python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_${name}.json').read().strip().splitlines()[-1])
    print('%-28s %8.3f ms/step  %s' % ('${name}', d['ms_per_step'], (d['config'].get('graph') or {}).get('live_rows',{}).get('gen')))
except Exception as e:
    print('${name} failed', e); print(open('gpurun_out/${TAG}_${name}.err').read()[-600:])
PY
}
timeout -k 10 300 python -m pytest tests/test_gpu_capacity.py -x -q > gpurun_out/${TAG}_cap_tests.log 2>&1; echo "cap tests rc $?"; tail -15 gpurun_out/${TAG}_cap_tests.log
run graph_default A=1
run graph_noside SGNN_SIDE_LANE=0
EXTRA="--headroom 1.25" run graph_h125 A=1
EXTRA="--classic --no-prefetch" run classic_tf A=1
EXTRA="" 
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_graph_default.json').read().strip().splitlines()[-1])
print('graph stats', d['config']['graph'])
PY
# kernel trace of the graph replay
export TMPDIR=/tmp; D=/tmp/prof_$TAG; rm -rf $D; ROOT=$(pwd)
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python $ROOT/bench.py --steps 20 --warmup 12 --no-cpu-baseline --no-traffic --no-other-mode --teacher-forced > $ROOT/gpurun_out/${TAG}_prof.out 2> $ROOT/gpurun_out/${TAG}_prof.err)
tail -2 $ROOT/gpurun_out/${TAG}_prof.err
F=$(find $D -name '*kernel_stats.csv' | head -1); T=$(find $D -name '*kernel_trace.csv' | head -1)
[ -n "$F" ] && cp $F gpurun_out/${TAG}_kernel_stats.csv
if [ -n "$T" ]; then
  python scripts/trace_graph.py $T > gpurun_out/${TAG}_trace_summary.txt 2>&1; head -60 gpurun_out/${TAG}_trace_summary.txt
fi
