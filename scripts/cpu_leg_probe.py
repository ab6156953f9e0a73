"""CPU-leg experiments on the GPU box's host (no GPU needed): python scripts/cpu_leg_probe.py [pin|nopin] [threads per replica]"""
import json, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from benchlib import cpu_leg
mode, T = sys.argv[1], int(sys.argv[2])
os.environ['SGNN_CPU_LEG_THREADS'] = str(T)
os.environ['SGNN_CPU_LEG_PIN'] = '1' if mode == 'pin' else '0'
os.environ['SGNN_CPU_LEG_SKIP_ONE'] = '1'
args = types.SimpleNamespace(dim=64, occupancy=0.05, cpu_blocks=8, cpu_torch_only=False)
r = cpu_leg.cpu_baseline_subprocess(args, os.path.join(ROOT, 'bench.py'))
print(mode, T, json.dumps(r['by_threads']))
