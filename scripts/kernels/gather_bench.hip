// Micro-benchmark (measurement tool, not product code): how fast can a CU gather 64-byte (or 32-byte) feature rows
// through a 3x3x3 neighbour table, as a function of the lane -> (row, chunk) mapping of the gather instruction?
//
// The convolution kernels (sgnn_amd/csrc/conv.hip) use the MFMA A-operand mapping directly: lane (r = lane & 15,
// q = lane >> 4) loads chunk q of row r, i.e. the 16 lanes of a quarter-wave touch 16 DIFFERENT rows and each row is
// touched by four quarter-waves.  Mode 1 loads the same bytes with lanes 4g .. 4g+3 covering one whole row.
//   mode 0: conv mapping, 16 B / lane            mode 1: row-contiguous mapping, 16 B / lane
//   mode 2: mode 1 + 4 x ds_bpermute per gather (what it costs to get back to the MFMA layout in registers)
//   mode 3: mode 1 + ds_write_b128 / ds_read_b128 through a per-wave LDS tile (the other way back)
//   mode 4: 32-byte rows (8 channels), conv mapping, 8 B / lane
//   mode 5: 32-byte rows, row-contiguous mapping, 16 B / lane (2 lanes per row, 32 rows per instruction)
// Every mode walks K offsets over 64 output rows per wave and adds up what it loaded (one float per lane is stored).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const float *__restrict__ x, int64_t n_in, int c,
                                               const int32_t *__restrict__ table, int64_t ld, int K, int64_t n_out,
                                               float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float tile[4][64 * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (uint32_t)(n_in * c * 4));
  const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(table, (uint32_t)((int64_t)K * ld * 4));
  const uint32_t lane_off = (uint32_t)(row0 + lane) * 4u, ld4 = (uint32_t)ld * 4u;
  const uint32_t rowb = (uint32_t)c * 4u;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  float *mine = tile[wave];
  auto one = [&](int k) {
    const int32_t iv = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_off, k * ld4, 0);
    if constexpr (MODE == 0) {
      const int r = lane & 15, q = lane >> 4;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int32_t id = __builtin_amdgcn_ds_bpermute((m * 16 + r) * 4, iv);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (uint32_t)id * rowb + q * 16, 0, 0);
        acc0 += __uint_as_float(v.x); acc1 += __uint_as_float(v.y); acc2 += __uint_as_float(v.z); acc3 += __uint_as_float(v.w);
      }
    } else if constexpr (MODE == 1 || MODE == 2 || MODE == 3) {
      const int rr = lane >> 2, ch = lane & 3;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int32_t id = __builtin_amdgcn_ds_bpermute((m * 16 + rr) * 4, iv);
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (uint32_t)id * rowb + ch * 16, 0, 0);
        if constexpr (MODE == 2) {   // lane (r, q) of the MFMA layout wants what lane 4r + q loaded
          const int src = (((lane & 15) << 2) | (lane >> 4)) * 4;
          v.x = __builtin_amdgcn_ds_bpermute(src, v.x);
          v.y = __builtin_amdgcn_ds_bpermute(src, v.y);
          v.z = __builtin_amdgcn_ds_bpermute(src, v.z);
          v.w = __builtin_amdgcn_ds_bpermute(src, v.w);
        }
        if constexpr (MODE == 3) {   // row-major [16 rows][16 floats] tile, chunk position rotated by row >> 2
          float *p = mine + m * 256 + rr * 16 + (((ch + (rr >> 2)) & 3) << 2);
          *reinterpret_cast<u32x4 *>(p) = v;
          const int r = lane & 15, q = lane >> 4;
          const float *g = mine + m * 256 + r * 16 + (((q + (r >> 2)) & 3) << 2);
          v = *reinterpret_cast<const u32x4 *>(g);
        }
        acc0 += __uint_as_float(v.x); acc1 += __uint_as_float(v.y); acc2 += __uint_as_float(v.z); acc3 += __uint_as_float(v.w);
      }
    } else if constexpr (MODE == 4) {
      const int r = lane & 15, q = lane >> 4;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int32_t id = __builtin_amdgcn_ds_bpermute((m * 16 + r) * 4, iv);
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_x, (uint32_t)id * rowb + q * 8, 0, 0);
        acc0 += __uint_as_float(v.x); acc1 += __uint_as_float(v.y);
      }
    } else {
      const int rr = lane >> 1, ch = lane & 1;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int32_t id = __builtin_amdgcn_ds_bpermute((m * 32 + rr) * 4, iv);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (uint32_t)id * rowb + ch * 16, 0, 0);
        acc0 += __uint_as_float(v.x); acc1 += __uint_as_float(v.y); acc2 += __uint_as_float(v.z); acc3 += __uint_as_float(v.w);
      }
    }
  };
  for (int k = 0; k + 2 < K; k += 3) {   // three offsets per trip: their 12 (6) gathers are in flight together
    one(k);
    one(k + 1);
    one(k + 2);
  }
  (void)n_out;
  out[row0 + lane] = (acc0 + acc1) + (acc2 + acc3);   // out holds ld + 256 floats: every lane stores (sums are compared)
}

extern "C" __attribute__((visibility("default"))) int gather_bench(const float *x, int64_t n_in, int c,
                                                                    const int32_t *table, int64_t ld, int K,
                                                                    int64_t n_out, float *out, int mode, void *stream) {
  const dim3 grid((unsigned)((n_out + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(k_gather<0>, grid, block, 0, s, x, n_in, c, table, ld, K, n_out, out); break;
    case 1: hipLaunchKernelGGL(k_gather<1>, grid, block, 0, s, x, n_in, c, table, ld, K, n_out, out); break;
    case 2: hipLaunchKernelGGL(k_gather<2>, grid, block, 0, s, x, n_in, c, table, ld, K, n_out, out); break;
    case 3: hipLaunchKernelGGL(k_gather<3>, grid, block, 0, s, x, n_in, c, table, ld, K, n_out, out); break;
    case 4: hipLaunchKernelGGL(k_gather<4>, grid, block, 0, s, x, n_in, c, table, ld, K, n_out, out); break;
    case 5: hipLaunchKernelGGL(k_gather<5>, grid, block, 0, s, x, n_in, c, table, ld, K, n_out, out); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
