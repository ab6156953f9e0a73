#!/bin/bash
# One GPU-box visit: [tests] + bench + rocprofv3 kernel-trace summary.  Usage: scripts/gpu_round.sh TAG "pytest args" [bench steps]
# Writes gpurun_out/${TAG}_{tests.log,bench.json,bench_kernel_stats.csv,bench_kernel_summary.txt}
TAG=$1; PYT=$2; STEPS=${3:-30}
mkdir -p gpurun_out
if [ -n "$PYT" ]; then
  timeout -k 10 ${TEST_TIMEOUT:-600} python -m pytest $PYT -x -q > gpurun_out/${TAG}_tests.log 2>&1
  RC=$?
  tail -5 gpurun_out/${TAG}_tests.log
  if [ $RC -ne 0 ]; then echo "tests failed (rc $RC): skipping bench/profile"; grep -n "Error\|error\|Abort" gpurun_out/${TAG}_tests.log | head -10; exit 1; fi
fi
timeout -k 10 200 python bench.py --steps $STEPS --warmup 10 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
[ -s gpurun_out/${TAG}_bench.json ] || { echo "bench failed"; tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
cat gpurun_out/${TAG}_bench.json | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['unit'], d['ms_per_step'], 'ms/step', 'frac', d['roofline']['frac'], d['config']['generated_sites_per_level'])"
export TMPDIR=/tmp; D=/tmp/prof_$TAG; rm -rf $D
ROOT=$(pwd)
(cd /tmp && timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python $ROOT/bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-traffic --no-other-mode > $ROOT/gpurun_out/${TAG}_prof.out 2> $ROOT/gpurun_out/${TAG}_prof.err)
tail -3 $ROOT/gpurun_out/${TAG}_prof.err
F=$(find $D -name '*kernel_stats.csv' | head -1)
if [ -n "$F" ]; then
  cp $F gpurun_out/${TAG}_bench_kernel_stats.csv
  python scripts/prof_summary.py $F $((STEPS+11)) 70 > gpurun_out/${TAG}_bench_kernel_summary.txt
  python scripts/prof_categories.py $F $((STEPS+11)) > gpurun_out/${TAG}_bench_categories.txt 2>/dev/null
  head -30 gpurun_out/${TAG}_bench_kernel_summary.txt
  python - <<PY
import csv
rows=list(csv.DictReader(open("$F")))
print('launches/step: %.0f' % (sum(int(r['Calls']) for r in rows)/float($STEPS+11)))
PY
fi
