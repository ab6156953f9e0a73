// Measurement kernel (not part of libsgnn_hip.so): sustained rate of v_mfma_f32_16x16x4_f32, the instruction every
// convolution kernel of the library is built on.  scripts/bench_mfma.py loads it.
//   mode 0: 8 independent accumulators per wave (the pipe never waits for a result)
//   mode 1: ONE accumulator (16 dependent MFMAs in a row, the weight-gradient kernel's pattern)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k_mfma(float *out, int iters, float seed) {
  const int lane = threadIdx.x & 63;
  float a = seed + lane * 1e-3f, b = seed - lane * 1e-3f;
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0)
        acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 7], 0, 0, 0);
      else
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[0], 0, 0, 0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;   // keeps the loop alive, never true in practice
}

extern "C" __attribute__((visibility("default"))) int mfma_peak(float *out, int mode, int blocks, int iters, void *stream) {
  if (mode == 0)
    hipLaunchKernelGGL(k_mfma<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 1.0f);
  else
    hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 1.0f);
  return (int)hipGetLastError();
}
