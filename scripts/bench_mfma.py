"""Sustained v_mfma_f32_16x16x4_f32 rate on this GPU (needs GPU): the ceiling the convolution kernels' MFMA phase can reach.
  python scripts/bench_mfma.py
Prints TFLOP/s and the shader clock the rate implies (one such MFMA occupies a SIMD's matrix pipe for 32 cycles)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, 'scripts', 'kernels', 'libmfma_peak.so'))
lib.mfma_peak.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
out = torch.zeros(1 << 22, device='cuda')
cus = torch.cuda.get_device_properties(0).multi_processor_count
st = torch.cuda.current_stream().cuda_stream
for mode, what in ((0, '8 independent accumulators'), (1, 'one accumulator (dependent chain)')):
    for wgs_per_cu in (1, 2, 4):
        blocks, iters = cus * wgs_per_cu, 20000 // wgs_per_cu
        lib.mfma_peak(out.data_ptr(), mode, blocks, 100, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.mfma_peak(out.data_ptr(), mode, blocks, iters, st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        n = blocks * 4 * iters * 16                     # wave-level MFMAs
        tf = n * 2048 / ms / 1e9
        per_simd = cus * 4
        ghz = n / per_simd * 32 / ms / 1e6               # if the pipe were never idle
        print('%-36s %d waves/SIMD  %7.2f ms  %6.1f TFLOP/s  -> %.2f GHz x 32 cycles/MFMA' % (what, wgs_per_cu, ms, tf, ghz))

# round 6: does the rate depend on the operand VALUES?  16 different operand pairs per iteration, four accumulators
lib.mfma_peak_data.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
torch.manual_seed(0)
for what, data, zero in (('random normal operands', torch.randn(32 * 256, device='cuda'), 0),
                         ('small-magnitude operands (1e-3 * normal)', torch.randn(32 * 256, device='cuda') * 1e-3, 0),
                         ('operands = 1.0', torch.ones(32 * 256, device='cuda'), 0),
                         ('operands = 0.0', torch.zeros(32 * 256, device='cuda'), 1)):
    for wgs_per_cu in (1, 2, 4):
        blocks, iters = cus * wgs_per_cu, 20000 // wgs_per_cu
        lib.mfma_peak_data(out.data_ptr(), data.data_ptr(), zero, blocks, 100, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.mfma_peak_data(out.data_ptr(), data.data_ptr(), zero, blocks, iters, st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        n = blocks * 4 * iters * 16
        print('%-42s %d waves/SIMD  %7.2f ms  %6.1f TFLOP/s  -> %.2f GHz x 32 cycles/MFMA'
              % (what, wgs_per_cu, ms, n * 2048 / ms / 1e9, n / (cus * 4) * 32 / ms / 1e6))

# round 6: the MFMA stream of the fused backward kernel's offset walk, looped (small code) and unrolled over 27 offsets
lib.mfma_peak_stage.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
data = torch.randn(64 * 256, device='cuda')
for nk in (1, 27):
    for wgs_per_cu in (1, 2):
        blocks, iters = cus * wgs_per_cu, (27 * 400 // nk) // wgs_per_cu
        lib.mfma_peak_stage(out.data_ptr(), data.data_ptr(), nk, blocks, 2, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.mfma_peak_stage(out.data_ptr(), data.data_ptr(), nk, blocks, iters, st); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        n = blocks * 4 * iters * nk * 32
        print('fused-backward stage stream, %2d offsets unrolled   %d waves/SIMD  %7.2f ms  %6.1f TFLOP/s  -> %.2f GHz x 32 cycles/MFMA'
              % (nk, wgs_per_cu, ms, n * 2048 / ms / 1e9, n / (cus * 4) * 32 / ms / 1e6))
