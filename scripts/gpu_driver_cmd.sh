#!/bin/bash
# what the driver runs at round end, on the working tree: smoke, the bench command line of the contract, the GPU suite
TAG=$1; mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/${TAG}_smoke.log
T0=$(date +%s)
timeout -k 10 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_driver_cmd.json 2> gpurun_out/${TAG}_driver_cmd.err
echo "driver bench rc $? in $(( $(date +%s) - T0 )) s"
python -c "import json; d=json.loads(open('gpurun_out/${TAG}_driver_cmd.json').read().strip().splitlines()[-1]); print(d['metric'], d['value'], d['unit'], d['ms_per_step'], 'roofline', d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'])"
timeout -k 10 1500 python -m pytest tests -m gpu -q -x > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/${TAG}_pytest_gpu.log
