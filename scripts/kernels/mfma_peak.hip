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

// mode 2 / 3 (round 6): 16 DIFFERENT operand pairs per iteration read from memory (mode 2: whatever `in` holds — random
// normal data in scripts/bench_mfma.py; mode 3: the same registers zeroed), four accumulators: does the sustained rate depend
// on the operand values / on operands changing from one instruction to the next?
template <int ZERO>
__global__ __launch_bounds__(256) void k_mfma_data(float *out, const float *in, int iters) {
  float a[16], b[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    a[u] = ZERO ? 0.f : in[(u * 2 + 0) * 256 + threadIdx.x];
    b[u] = ZERO ? 0.f : in[(u * 2 + 1) * 256 + threadIdx.x];
  }
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[u & 3], 0, 0, 0);
    asm volatile("" : "+v"(a[0]), "+v"(b[0]));
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

// mode "stage" (round 6): the MFMA stream of one offset of the fused backward kernel (csrc/conv_bwd_fused.hip): 4 dX MFMAs on
// four accumulators, then 12 x (dW on one accumulator, dX), then 4 dW; operands from 32 + 8 different registers; two waves per
// SIMD.  NK = 27: the 27 offsets unrolled with 27 dW accumulators (the kernel's shape, ~25 KB of code); NK = 1: one stage in a
// loop (same instruction mix, a few hundred bytes of code).
template <int NK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_mfma_stage(float *out, const float *in, int iters) {
  float a[16], xa[16], t[16], b[4];
#pragma unroll
  for (int u = 0; u < 16; ++u) {
    a[u] = in[(u * 3 + 0) * 256 + threadIdx.x];
    xa[u] = in[(u * 3 + 1) * 256 + threadIdx.x];
    t[u] = in[(u * 3 + 2) * 256 + threadIdx.x];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) b[u] = in[(48 + u) * 256 + threadIdx.x];
  f32x4 acc[4], ad[NK];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NK; ++i) ad[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m * 4], b[0], acc[m], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 1; s < 4; ++s)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          ad[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[(s - 1) * 4 + m], t[(s - 1) * 4 + m], ad[k], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m * 4 + s], b[s], acc[m], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) ad[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[12 + i], t[12 + i], ad[k], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+v"(a[0]), "+v"(t[0]), "+v"(xa[0]), "+v"(b[0]));
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < NK; ++i) s += ad[i][0] + ad[i][1] + ad[i][2] + ad[i][3];
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" __attribute__((visibility("default"))) int mfma_peak_stage(float *out, const float *in, int nk, int blocks, int iters,
                                                                      void *stream) {
  if (nk == 27)
    hipLaunchKernelGGL(k_mfma_stage<27>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, in, iters);
  else
    hipLaunchKernelGGL(k_mfma_stage<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, in, iters);
  return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default"))) int mfma_peak_data(float *out, const float *in, int zero, int blocks, int iters,
                                                                     void *stream) {
  if (zero)
    hipLaunchKernelGGL(k_mfma_data<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, in, iters);
  else
    hipLaunchKernelGGL(k_mfma_data<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, in, iters);
  return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default"))) int mfma_peak(float *out, int mode, int blocks, int iters, void *stream) {
  if (mode == 0)
    hipLaunchKernelGGL(k_mfma<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 1.0f);
  else
    hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 1.0f);
  return (int)hipGetLastError();
}
