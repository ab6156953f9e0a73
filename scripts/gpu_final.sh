#!/bin/bash
# Closing run of a round (round 4 on): smoke, the default bench (CPU leg, isolated + in-step PMC passes), kernel trace + stats of the bench
# command.   Usage: scripts/gpu_final.sh TAG      (the full `pytest -m gpu` suite is run separately)
TAG=$1
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc $?"; tail -2 gpurun_out/${TAG}_smoke.log
T0=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/${TAG}_bench_full.json 2> gpurun_out/${TAG}_bench_full.err
echo "default bench rc $? in $(( $(date +%s) - T0 )) s"; tail -3 gpurun_out/${TAG}_bench_full.err
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_full.json').read().strip().splitlines()[-1])
r=d['roofline']
print('bench', d['value'], d['unit'], d['ms_per_step'], 'ms/step; roofline frac', r['frac'], 'traffic', r.get('traffic'), 'step', r.get('step'))
print('by size', [(b['mean_rows'], b['rules_per_row'], b['avg_us'], b['frac_of_fp32_mfma_peak']) for b in r['by_level_size']])
print('top', [(k['kernel'], k['frac']) for k in r['top_kernels']])
print('in-step', [(k['kernel'][:30], k.get('mfma_util'), k.get('ta_busy'), k.get('hbm_MB_per_dispatch')) for k in (r.get('counters_in_step') or {}).get('kernels', [])])
c=d.get('cpu_baseline') or {}
print('cpu', c.get('value'), c.get('cores'), c.get('by_threads'), d.get('gpu_over_cpu'))
print('legs', {k:(v.get('ms_per_step')) for k,v in (d.get('other_legs') or {}).items()}, 'launches', d.get('launches_per_step'), 'batch1', d.get('batch1_ms'), 'host graph launch ms', d.get('host_graph_launch_ms'))
PY
export TMPDIR=/tmp; D=/tmp/prof_$TAG; rm -rf $D; ROOT=$(pwd)
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python $ROOT/bench.py --no-cpu-baseline --no-traffic --no-other-mode > $ROOT/gpurun_out/${TAG}_prof.out 2> $ROOT/gpurun_out/${TAG}_prof.err)
F=$(find $D -name '*kernel_stats.csv' | head -1); T=$(find $D -name '*kernel_trace.csv' | head -1)
[ -n "$F" ] && cp $F gpurun_out/${TAG}_bench_kernel_stats.csv && head -8 $F
[ -n "$T" ] && python scripts/trace_graph.py $T 380 gpurun_out/${TAG}_step_launches.csv > gpurun_out/${TAG}_trace_summary.txt 2>&1
head -8 gpurun_out/${TAG}_trace_summary.txt
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_prof.out').read().strip().splitlines()[-1])
g=d['config']['graph']
n=g['probe_steps']+g['eager_steps']+g['replays']+5+1
print('steps under the profiler', n)
open('gpurun_out/${TAG}_steps.txt','w').write(str(n))
PY
[ -n "$F" ] && python scripts/prof_categories.py gpurun_out/${TAG}_bench_kernel_stats.csv $(cat gpurun_out/${TAG}_steps.txt) > gpurun_out/${TAG}_bench_categories.txt && cat gpurun_out/${TAG}_bench_categories.txt
