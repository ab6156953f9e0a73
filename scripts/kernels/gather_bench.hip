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
  auto one = [&](int kk) {
    // MODE 8: mode 0 with the offsets walked dx-major (k = 3 * (kk % 9) + kk / 9): the three offsets in flight together no
    // longer are the dx = -1, 0, +1 neighbours of one (dz, dy) row, i.e. they share no cache lines
    const int k = (MODE == 8) ? (3 * (kk % 9) + kk / 9) : kk;
    const int32_t iv = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_off, k * ld4, 0);
    if constexpr (MODE == 0 || MODE == 8) {
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

// mode 6: the same rows by LDS-DMA (buffer_load_dwordx4 ... lds) in row-contiguous order into a three-slot ring per
// wave, each lane then reads its MFMA fragment back with ds_read_b128 — the gather path of conv_lds.hip without the MFMAs
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, float *dst, uint32_t voff) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)dst, 16, voff, 0, 0, 0);
#endif
}
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_gather_dma(const float *__restrict__ x, int64_t n_in,
                                                         const int32_t *__restrict__ table, int64_t ld, int64_t n_out,
                                                         float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float ring0[WAVES][1024];
  __shared__ __attribute__((aligned(16))) float ring1[WAVES][1024];
  __shared__ __attribute__((aligned(16))) float ring2[WAVES][1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * WAVES + wave) * 64;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (uint32_t)(n_in * 64));
  const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(table, (uint32_t)(27 * ld * 4));
  const uint32_t lane_off = (uint32_t)(row0 + lane) * 4u, ld4 = (uint32_t)ld * 4u;
  int32_t idx[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) idx[k] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_off, k * ld4, 0);
  const int rl = lane >> 2, cpos = lane & 3, r = lane & 15, q = lane >> 4;
  auto swz = [](int row) { return (0x78 >> (((row & 15) >> 2) * 2)) & 3; };
  auto slot = [&](int s) -> float * { return s == 0 ? ring0[wave] : (s == 1 ? ring1[wave] : ring2[wave]); };
  auto issue = [&](int k) {
    float *dst = slot(k % 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = j * 16 + rl;
      const int32_t id = __builtin_amdgcn_ds_bpermute(row * 4, idx[k]);
      lds_dma16(rs_x, dst + j * 256, (uint32_t)id * 64u + (uint32_t)((cpos ^ swz(row)) * 16));
    }
  };
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  issue(0);
  issue(1);
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    if (k + 2 < 27) issue(k + 2);
    __builtin_amdgcn_sched_barrier(0);
    const float *src = slot(k % 3);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int row = m * 16 + r;
      const u32x4 v = *reinterpret_cast<const u32x4 *>(src + (row * 4 + (q ^ swz(row))) * 4);
      acc0 += __uint_as_float(v.x); acc1 += __uint_as_float(v.y); acc2 += __uint_as_float(v.z); acc3 += __uint_as_float(v.w);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  (void)n_out;
  out[row0 + lane] = (acc0 + acc1) + (acc2 + acc3);
}

// Response surface of the convolution's inner loop: conv-mapping gathers, DEPTH offsets in flight per wave, NMFMA MFMAs per
// offset on the gathered registers (16 = the <16,16> convolution), occupancy limited to what 32 KiB of LDS per workgroup allows
// (5 workgroups = 20 waves per CU, the convolution kernel's).  Straight-line code (27 offsets unrolled): exact s_waitcnt counts.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int DEPTH, int NMFMA, int FEAT = 0>   // FEAT bit 0: weight tile staged into LDS, B fragments by ds_read_b128; bit 1: 64 B / row output store
__global__ __launch_bounds__(256) void k_loop(const float *__restrict__ x, int64_t n_in, const int32_t *__restrict__ table,
                                             int64_t ld, int64_t n_out, float *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) float pad[8192 - 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (uint32_t)(n_in * 64));
  const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(table, (uint32_t)(27 * ld * 4));
  const uint32_t lane_off = (uint32_t)(row0 + lane) * 4u, ld4 = (uint32_t)ld * 4u;
  if (n_out < 0) pad[tid] = 1.f;    // keeps the array
  if constexpr (FEAT & 1) {          // 27 x 16 x 16 "weights" (the first rows of x) -> LDS, like the convolution's stage()
    for (int e = tid; e < 27 * 256; e += 256) pad[e] = x[e];
    __syncthreads();
  }
  int32_t idx[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) idx[k] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_off, k * ld4, 0);
  const int r = lane & 15, q = lane >> 4;
  u32x4 a[DEPTH + 1][4];
  auto issue = [&](int k) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int32_t id = __builtin_amdgcn_ds_bpermute((m * 16 + r) * 4, idx[k]);
      a[k % (DEPTH + 1)][m] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (uint32_t)id * 64u + q * 16, 0, 0);
    }
  };
  f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
  float s0 = 0.f;
#pragma unroll
  for (int k = 0; k < DEPTH; ++k) issue(k);
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    if (k + DEPTH < 27) issue(k + DEPTH);
    __builtin_amdgcn_sched_barrier(0);
    const u32x4 *v = a[k % (DEPTH + 1)];
    if constexpr (NMFMA > 0) {
#pragma unroll
      for (int i = 0; i < NMFMA; ++i) {
        const int m = i & 3, c = (i >> 2) & 3;
        const float av = __uint_as_float(c == 0 ? v[m].x : (c == 1 ? v[m].y : (c == 2 ? v[m].z : v[m].w)));
        float bv = 1.0f;
        if constexpr (FEAT & 1) bv = pad[(k * 16 + r) * 16 + q * 4 + c];
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[m], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int m = 0; m < 4; ++m) s0 += __uint_as_float(v[m].x) + __uint_as_float(v[m].y) + __uint_as_float(v[m].z) + __uint_as_float(v[m].w);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (FEAT & 2) {          // the convolution's epilogue store: 16 columns x 64 rows per wave, 4-byte stores
    float *y = out + 1024 + row0 * 16;     // (out is sized for it by the caller)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) y[(m * 16 + q * 4 + i) * 16 + r] = acc[m][i] + s0;
  } else {
    out[row0 + lane] = s0 + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (n_out < 0 ? pad[lane] : 0.f);
  }
}


// Modes 20-22 (round 5; VERDICT r4 item 5, DESIGN.md section 8 "open 2"): fewer gathers per rule.  On a raster-ordered level the
// dx = -1, 0, +1 neighbours of consecutive output rows are consecutive input rows: nbr[(dz,dy,+1)][j] == nbr[(dz,dy,0)][j+1]
// wherever both exist.  In the MFMA-operand mapping lane (r, q) holds chunk q of the row of output row r of a 16-row tile, so the
// dx = +1 rows of a tile are the dx = 0 rows one lane further (DPP row_shl:1), the dx = -1 rows one lane back (row_shr:1) — except
// at the tile's edge lane and wherever the table disagrees (run boundaries, generated levels in child order), where a per-lane
// FALLBACK gather fetches the row (all other lanes out of range: the instruction still issues, but touches few rows).
//   mode 20: three full gathers per (dz, dy) pair (the baseline in this loop structure)
//   mode 21: one full gather + two fallback gathers + DPP shifts
//   mode 22: mode 21, fallback instruction skipped (wave-uniform branch) when no lane of the wave needs it
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_row(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
template <int MODE>
__global__ __launch_bounds__(256) void k_gather3(const float *__restrict__ x, int64_t n_in, const int32_t *__restrict__ table,
                                                int64_t ld, int64_t n_out, float *__restrict__ out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (uint32_t)(n_in * 64));
  const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(table, (uint32_t)(27 * ld * 4));
  const uint32_t lane_off = (uint32_t)(row0 + lane) * 4u, ld4 = (uint32_t)ld * 4u;
  const int r = lane & 15, q = lane >> 4;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  auto add = [&](const u32x4 &v) {
    acc0 += __uint_as_float(v.x); acc1 += __uint_as_float(v.y); acc2 += __uint_as_float(v.z); acc3 += __uint_as_float(v.w);
  };
  for (int t = 0; t < 9; ++t) {
    const int32_t ivm = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_off, (3 * t) * ld4, 0);
    const int32_t iv0 = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_off, (3 * t + 1) * ld4, 0);
    const int32_t ivp = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_off, (3 * t + 2) * ld4, 0);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int32_t idm = __builtin_amdgcn_ds_bpermute((m * 16 + r) * 4, ivm);
      const int32_t id0 = __builtin_amdgcn_ds_bpermute((m * 16 + r) * 4, iv0);
      const int32_t idp = __builtin_amdgcn_ds_bpermute((m * 16 + r) * 4, ivp);
      const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (uint32_t)id0 * 64u + q * 16, 0, 0);
      add(v0);
      if constexpr (MODE == 20) {
        add(__builtin_amdgcn_raw_buffer_load_b128(rs_x, (uint32_t)idm * 64u + q * 16, 0, 0));
        add(__builtin_amdgcn_raw_buffer_load_b128(rs_x, (uint32_t)idp * 64u + q * 16, 0, 0));
      } else {
        const int32_t id0n = (int32_t)dpp_row<0x101>((uint32_t)id0), id0p = (int32_t)dpp_row<0x111>((uint32_t)id0);
        const bool hitp = r != 15 && idp >= 0 && idp == id0n, hitm = r != 0 && idm >= 0 && idm == id0p;   // (edge lanes of the 16-row tile: no neighbour lane)
        const uint32_t offp = hitp ? 0xFFFFF800u : (uint32_t)idp * 64u + q * 16, offm = hitm ? 0xFFFFF800u : (uint32_t)idm * 64u + q * 16;
        u32x4 fp = {0, 0, 0, 0}, fm = {0, 0, 0, 0};
        if constexpr (MODE == 21) {
          fp = __builtin_amdgcn_raw_buffer_load_b128(rs_x, offp, 0, 0);
          fm = __builtin_amdgcn_raw_buffer_load_b128(rs_x, offm, 0, 0);
        } else {
          if (__ballot(!hitp && idp >= 0)) fp = __builtin_amdgcn_raw_buffer_load_b128(rs_x, offp, 0, 0);
          if (__ballot(!hitm && idm >= 0)) fm = __builtin_amdgcn_raw_buffer_load_b128(rs_x, offm, 0, 0);
        }
        u32x4 sp, sm;
        sp.x = dpp_row<0x101>(v0.x); sp.y = dpp_row<0x101>(v0.y); sp.z = dpp_row<0x101>(v0.z); sp.w = dpp_row<0x101>(v0.w);
        sm.x = dpp_row<0x111>(v0.x); sm.y = dpp_row<0x111>(v0.y); sm.z = dpp_row<0x111>(v0.z); sm.w = dpp_row<0x111>(v0.w);
        add(hitp ? sp : fp);
        add(hitm ? sm : fm);
      }
    }
  }
  (void)n_out;
  out[row0 + lane] = (acc0 + acc1) + (acc2 + acc3);
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
    case 10: hipLaunchKernelGGL((k_loop<1, 0>), grid, block, 0, s, x, n_in, table, ld, n_out, out); break;
    case 11: hipLaunchKernelGGL((k_loop<2, 0>), grid, block, 0, s, x, n_in, table, ld, n_out, out); break;
    case 12: hipLaunchKernelGGL((k_loop<3, 0>), grid, block, 0, s, x, n_in, table, ld, n_out, out); break;
    case 13: hipLaunchKernelGGL((k_loop<1, 16>), grid, block, 0, s, x, n_in, table, ld, n_out, out); break;
    case 14: hipLaunchKernelGGL((k_loop<2, 16>), grid, block, 0, s, x, n_in, table, ld, n_out, out); break;
    case 15: hipLaunchKernelGGL((k_loop<3, 16>), grid, block, 0, s, x, n_in, table, ld, n_out, out); break;
    case 17: hipLaunchKernelGGL((k_loop<1, 16, 1>), grid, block, 0, s, x, n_in, table, ld, n_out, out); break;
    case 18: hipLaunchKernelGGL((k_loop<1, 16, 2>), grid, block, 0, s, x, n_in, table, ld, n_out, out); break;
    case 19: hipLaunchKernelGGL((k_loop<1, 16, 3>), grid, block, 0, s, x, n_in, table, ld, n_out, out); break;
    case 16: hipLaunchKernelGGL((k_loop<4, 16>), grid, block, 0, s, x, n_in, table, ld, n_out, out); break;
    case 8: hipLaunchKernelGGL(k_gather<8>, grid, block, 0, s, x, n_in, c, table, ld, K, n_out, out); break;
    case 20: hipLaunchKernelGGL(k_gather3<20>, grid, block, 0, s, x, n_in, table, ld, n_out, out); break;
    case 21: hipLaunchKernelGGL(k_gather3<21>, grid, block, 0, s, x, n_in, table, ld, n_out, out); break;
    case 22: hipLaunchKernelGGL(k_gather3<22>, grid, block, 0, s, x, n_in, table, ld, n_out, out); break;
    case 6: hipLaunchKernelGGL(k_gather_dma<4>, grid, block, 0, s, x, n_in, table, ld, n_out, out); break;
    case 7: hipLaunchKernelGGL(k_gather_dma<8>, dim3((unsigned)((n_out + 511) / 512)), dim3(512), 0, s, x, n_in, table, ld, n_out, out); break;
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
