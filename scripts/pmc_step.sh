#!/bin/bash
# In-step counters of the kernels that carry the training step -> gpurun_out/${TAG}_pmc_instep.json (copy into profiles/)
TAG=${1:-r04}
mkdir -p gpurun_out
python - <<PY > gpurun_out/${TAG}_pmc_instep.json
import json, sys, types
sys.path.insert(0, '.')
from benchlib import pmc
args = types.SimpleNamespace(batch=32, dim=64, occupancy=0.05)
print(json.dumps(pmc.instep_counters(args), indent=1))
PY
head -c 3000 gpurun_out/${TAG}_pmc_instep.json
