"""s_memtime stamps of one workgroup of the fused backward kernel (measurement build: scripts/build_variant.sh trace
conv_bwd_fused.hip -DFUSED_TRACE=0, loaded through SGNN_LIB).  Prints where wave 0 of that workgroup spends its cycles (100 MHz
s_memtime ticks converted at the reported shader clock are approximate; ratios matter)."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], '--no-parity', '--iters', '2']
exec(open(os.path.join(ROOT, 'scripts', 'bench_bwd_fused.py')).read())
buf = (ctypes.c_ulonglong * 512)()
lib.sgnn_debug_fused_trace.argtypes = [ctypes.c_void_p]
assert lib.sgnn_debug_fused_trace(buf) == 0
t = np.array(buf[:], dtype=np.int64)
print('stamps (ticks): start %d, weights staged +%d' % (t[0], t[1] - t[0]))
for j in range(8):
    b = 2 + j * 32
    if t[b] == 0 or b + 29 >= 500:
        break
    st = t[b + 2:b + 29]
    d = np.diff(np.concatenate([[t[b + 1]], st]))
    print('tile %d: x tile + transposition %d | 27 stages: total %d, mean %.1f, min %d, max %d, first 3 %s | epilogue %d'
          % (j, t[b + 1] - t[b], st[-1] - t[b + 1], d.mean(), d.min(), d.max(), d[:3].tolist(), t[b + 29] - st[-1]))
print('statistics partial %d | dW combine + store %d | whole workgroup %d' % (t[501] - t[500], t[502] - t[501], t[502] - t[0]))
