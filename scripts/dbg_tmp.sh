mkdir -p gpurun_out
SGNN_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 2 --batch 2 --dim 32 --no-cpu-baseline > gpurun_out/dbg_dp.out 2> gpurun_out/dbg_dp.err
echo "rc $?"; grep -v "amdgpu.ids\|Setting OMP" gpurun_out/dbg_dp.err | grep -B2 -A25 "Traceback" | head -80
python - <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from util import param_fill
from sgnn_amd import synth
from sgnn_amd.model import GenModel
from sgnn_amd.train import GraphStep
import numpy as np
d = synth.make_batch(2, (32,32,32), cfg=3, occupancy=0.08)
b = {'input': [d['input'][0].cuda(), d['input'][1].cuda()], 'sdf': d['sdf'].cuda(), 'known': d['known'].cuda(), 'hierarchy': [h.cuda() for h in d['hierarchy']]}
m = param_fill(GenModel(8, (32,32,32), 1, 16, 16, 4, True, True, 1, 1), 3).train().cuda()
gs = GraphStep(m, settle=False); gs.count_nodes = True
import sgnn_amd.train as T
orig = T._count_graph_nodes
def dump(g):
    g.debug_dump('gpurun_out/dbg_graph.dot'); return orig(g)
T._count_graph_nodes = dump
for i in range(4): gs(b, np.ones(5, dtype=np.float32))
torch.cuda.synchronize()
print(gs.stats)
t = open('gpurun_out/dbg_graph.dot').read()
print(len(t)); print(t[:1500])
PY
