#!/bin/bash
# Round-3 closing run: complete GPU suite, smoke, the default bench (CPU leg + PMC traffic), kernel trace + stats of the
# bench command, memory report.   Usage: scripts/gpu_r3_final.sh TAG
TAG=$1
mkdir -p gpurun_out
timeout -k 10 1200 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_tests.log 2>&1
echo "tests rc $?"; grep -n "^E  \|^FAILED\|passed\|failed" gpurun_out/${TAG}_tests.log | head -20
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc $?"; tail -3 gpurun_out/${TAG}_smoke.log
T0=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_full.err
echo "default bench rc $? in $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/${TAG}_bench_full.err
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_full.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['unit'], d['ms_per_step'], 'ms/step; roofline frac', d['roofline']['frac'], 'traffic', d['roofline'].get('traffic'), 'step', d['roofline'].get('step'))
print('cpu', d.get('cpu_baseline'))
print('legs', {k:(v.get('ms_per_step')) for k,v in (d.get('other_legs') or {}).items()}, 'launches', d.get('launches_per_step'), 'batch1', d.get('batch1_ms'))
PY
export TMPDIR=/tmp; D=/tmp/prof_$TAG; rm -rf $D; ROOT=$(pwd)
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python $ROOT/bench.py --no-cpu-baseline --no-traffic > $ROOT/gpurun_out/${TAG}_prof.out 2> $ROOT/gpurun_out/${TAG}_prof.err)
F=$(find $D -name '*kernel_stats.csv' | head -1); T=$(find $D -name '*kernel_trace.csv' | head -1)
[ -n "$F" ] && cp $F gpurun_out/${TAG}_bench_kernel_stats.csv && head -12 $F
[ -n "$T" ] && python scripts/trace_graph.py $T 350 gpurun_out/${TAG}_step_launches.csv > gpurun_out/${TAG}_trace_summary.txt 2>&1
head -8 gpurun_out/${TAG}_trace_summary.txt
for c in c1 c3 c4; do
  timeout -k 10 300 python -u scripts/memory_report.py $c 2>&1 | grep -v "amdgpu.ids\|UserWarning\|run_backward" >> gpurun_out/${TAG}_memory.txt
done
cat gpurun_out/${TAG}_memory.txt | cut -c1-400
