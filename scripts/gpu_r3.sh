#!/bin/bash
# Round-3 GPU visit: capacity/graph tests first (fail fast), then the whole GPU suite, then bench (+ optional rocprof).
# Usage: scripts/gpu_r3.sh TAG [full|quick|bench] [bench steps]
TAG=$1; MODE=${2:-full}; STEPS=${3:-40}
mkdir -p gpurun_out
if [ "$MODE" != "bench" ]; then
  timeout -k 10 400 python -m pytest tests/test_gpu_capacity.py -x -q > gpurun_out/${TAG}_cap_tests.log 2>&1
  RC=$?
  tail -25 gpurun_out/${TAG}_cap_tests.log
  if [ $RC -ne 0 ]; then echo "capacity tests failed (rc $RC)"; grep -n "Error\|error\|assert" gpurun_out/${TAG}_cap_tests.log | head -30; fi
  if [ "$MODE" = "full" ]; then
    timeout -k 10 700 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_capacity.py > gpurun_out/${TAG}_tests.log 2>&1
    echo "full suite rc $?"; tail -8 gpurun_out/${TAG}_tests.log
  fi
fi
timeout -k 10 300 python bench.py --steps $STEPS --warmup 10 --no-cpu-baseline --no-traffic > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench rc $?"; tail -5 gpurun_out/${TAG}_bench.err
[ -s gpurun_out/${TAG}_bench.json ] && python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['unit'], d['ms_per_step'], 'ms/step', 'frac', (d.get('roofline') or {}).get('frac'))
print('graph', d['config'].get('graph'))
print('legs', json.dumps(d.get('other_legs')))
PY
