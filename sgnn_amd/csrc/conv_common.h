// Pieces shared by the convolution kernels (conv.hip: looped register-gather kernels, small-level kernel, weight gradients;
// conv_unrolled.hip: the straight-line variants).  (The LDS-DMA gather kernel that also used this header was deleted in round 4.)
#pragma once
#include <atomic>
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CONV_MREP 4                        // 16-row MFMA tiles per wave in the large-grid variant
#define CONV_ROWS_PER_WAVE (16 * CONV_MREP)
#define CONV_ROWS_PER_BLOCK (4 * CONV_ROWS_PER_WAVE)   // also the required multiple of the table's ld

template <int CIN, int COUT>
struct ConvCfg {
  static constexpr int V = (CIN + 3) / 4;    // channels per lane quarter
  static constexpr int CINP = 4 * V;         // padded input channels
  static constexpr int NT = (COUT + 15) / 16;  // 16-wide output column tiles
  static constexpr int PER_K = NT * 16 * CINP;  // LDS floats per offset
#ifndef CONV_LDS_FLOATS
#define CONV_LDS_FLOATS 8192                  // weight tile budget: 32 KiB of LDS
#endif
  static constexpr int KC_RAW = CONV_LDS_FLOATS / PER_K;
  static constexpr int KC = KC_RAW < 1 ? 1 : (KC_RAW > 27 ? 27 : KC_RAW);
  // widest load that is aligned for every (row, q)
  static constexpr int ALIGN = ((CIN % 4 == 0) && (V % 4 == 0)) ? 16 : (((CIN % 2 == 0) && (V % 2 == 0)) ? 8 : 4);
};

// optional generalisation of the rulebook walk (sgnn_conv_*_ex): offset k of group g reads table row
// kmap[g*K + k] (NULL: k), gathers feature row idx*in_mul + kadd[g*K + k] (NULL: +0) and group g owns output rows
// row*groups + g and the weight block g.  Plain convolutions use {NULL, NULL, 1, 1}.
struct ConvEx {
  const int32_t *kmap;
  const int32_t *kadd;
  int in_mul;
  int groups;
  int table_rows;   // rows of `table` (K unless kmap selects rows of a larger table)
};

// resident workgroups of a 256-thread kernel on the whole device (occupancy x CUs), cached per instantiation: the large-level
// kernels size their row tiles so that ONE round of workgroups covers the level (k_conv_fwd; sgnn_conv_set_one_round)
// Per DEVICE (a process may drive several) and thread-safe: relaxed atomics, every thread that races computes the same value.
// A failed query is NOT cached (0 = "one-round mode off" would otherwise stick for the life of the process).  The J-tile
// decomposition of a level — and with it the fp64 grouping of the BatchNorm statistics partials — follows this value, so
// statistics are bit-reproducible per (device model, driver, compiler), not across them (include/sgnn_hip.h says so).
// hipOccupancyMaxActiveBlocksPerMultiprocessor is a host-side query: legal during a stream capture.
template <auto KERNEL>
static int conv_wg_capacity() {
  static std::atomic<int> cache[16];   // zero-initialised; value + 1 is stored so that 0 means "not computed yet"
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  const int slot = dev & 15;
  const int have = cache[slot].load(std::memory_order_relaxed);
  if (have > 0 && (dev < 16)) return have - 1;
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, KERNEL, 256, 0) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return 0;
  if (dev < 16) cache[slot].store(per_cu * cus + 1, std::memory_order_relaxed);
  return per_cu * cus;
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}

// V consecutive floats at byte offset `off` of a raw buffer; an out-of-range offset (rule entry -1 maps to
// 0xFFFFFFxx) returns zeros, so missing neighbours need neither a branch nor a select.  Multi-dword
// buffer loads only need dword alignment on gfx950.
template <int V>
__device__ __forceinline__ void buf_load_floats(__amdgpu_buffer_rsrc_t rs, uint32_t off, float (&a)[V]) {
  int s = 0;
#pragma unroll
  for (; s + 4 <= V; s += 4) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 4 * s, 0, 0);
    a[s] = __uint_as_float(v.x); a[s + 1] = __uint_as_float(v.y);
    a[s + 2] = __uint_as_float(v.z); a[s + 3] = __uint_as_float(v.w);
  }
  if constexpr (V % 4 == 3) {
    const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, off + 4 * (V - 3), 0, 0);
    a[V - 3] = __uint_as_float(v.x); a[V - 2] = __uint_as_float(v.y); a[V - 1] = __uint_as_float(v.z);
  } else if constexpr (V % 4 == 2) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off + 4 * (V - 2), 0, 0);
    a[V - 2] = __uint_as_float(v.x); a[V - 1] = __uint_as_float(v.y);
  } else if constexpr (V % 4 == 1) {
    a[V - 1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off + 4 * (V - 1), 0, 0));
  }
}

// Weight slices k0 .. k0 + kc - 1 into LDS as wl[kk][n][c] (the B-fragment layout: n = output column, c = input channel,
// NT*16 x CINP floats per offset, padding zero).  The global tensor is (K, CIN, COUT) — or (K, COUT, CIN) for the data
// gradient (transpose) — so a slice is one contiguous run: 16-byte loads in global order (coalesced; dword alignment
// suffices for raw-buffer loads), four LDS writes each.  Round 4: the element-wise form before walked the LDS image and
// read the forward layout with a stride of COUT floats per thread, 27 dependent iterations for a 16 x 16 x 27 tile
// (scripts/kernels/gather_bench.hip: the weight tile was +7 us of a 62 us launch).  Contains the barriers around the tile.
template <int CIN, int COUT>
__device__ __forceinline__ void conv_stage_weights(float *wl, const float *__restrict__ w, int K, int k0, int kc,
                                                   bool transpose, bool flip) {
  using C = ConvCfg<CIN, COUT>;
  constexpr int CINP = C::CINP, NT = C::NT, SL = CIN * COUT;
  const int tid = threadIdx.x;
  __syncthreads();                                   // every wave is done with the previous contents
  if constexpr (CINP != CIN || NT * 16 != COUT) {    // padded image: zero it first (LDS only)
    for (int e = tid; e < kc * C::PER_K; e += 256) wl[e] = 0.f;
    __syncthreads();
  }
  const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(w, (uint32_t)((int64_t)K * SL * 4));
  auto put = [&](int ko, int idx, float v) {         // idx: element of the slice in global order
    const int c = transpose ? idx % CIN : idx / COUT, n = transpose ? idx / CIN : idx % COUT;
    wl[(ko * NT * 16 + n) * CINP + c] = v;
  };
  if constexpr (SL % 4 == 0) {
    for (int g = tid; g < kc * (SL / 4); g += 256) {
      const int ko = g / (SL / 4), r = (g - ko * (SL / 4)) * 4;
      const int ks = flip ? (K - 1 - (k0 + ko)) : (k0 + ko);
      float v[4];
      buf_load_floats<4>(rs_w, (uint32_t)(ks * SL + r) * 4u, v);
#pragma unroll
      for (int j = 0; j < 4; ++j) put(ko, r + j, v[j]);
    }
  } else {
    for (int g = tid; g < kc * SL; g += 256) {
      const int ko = g / SL, r = g - ko * SL;
      const int ks = flip ? (K - 1 - (k0 + ko)) : (k0 + ko);
      put(ko, r, w[(int64_t)ks * SL + r]);
    }
  }
  __syncthreads();
}

// Shared epilogue of the MFMA kernels: optional residual addend, strided store, optional BatchNorm statistics
// (see ConvEpi).  acc[m][nt] holds the wave's rows row0 + m*16 + (lane>>4)*4 + i, column nt*16 + (lane&15).
// Contains barriers when statistics are on: call it from all threads.  sred: >= 4*2*NT*16 doubles of LDS.
// (Round 4: passing each 16-row tile through LDS to store whole 64-byte rows per four lanes — 4x fewer, full-line store
//  instructions — was built and measured: 6.34 / 6.35 / 6.41 ms per step without / on launches with statistics / on every
//  launch; the barrier it needs before re-using the weight tile costs more than the scattered 4-byte stores.  Dropped.)
template <int COUT, int M, int NT>
__device__ __forceinline__ void conv_epilogue_rows(f32x4 (&acc)[M][NT], int64_t row0, int64_t n_out, unsigned groups,
                                                   unsigned grp, float *y, const ConvEpi &epi, int stats,
                                                   const float *any_ptr, double (&s1)[NT], double (&s2)[NT]) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int r = lane & 15, q = lane >> 4;
  const uint32_t ldy4 = (uint32_t)epi.ldy * 4u;
  const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(y, (uint32_t)(((n_out * groups - 1) * epi.ldy + COUT) * 4));
  const bool has_add = epi.addend != nullptr;
  const __amdgpu_buffer_rsrc_t rs_a =
      make_rsrc(has_add ? epi.addend : any_ptr, has_add ? (uint32_t)(((n_out * groups - 1) * epi.ld_add + COUT) * 4) : 0u);
  const uint32_t lda4 = (uint32_t)epi.ld_add * 4u;
  const __amdgpu_buffer_rsrc_t rs_b =
      make_rsrc(stats == 2 ? epi.bn_x : any_ptr, stats == 2 ? (uint32_t)(((n_out - 1) * epi.ld_bnx + COUT) * 4) : 0u);
  const uint32_t ldb4 = (uint32_t)epi.ld_bnx * 4u;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = nt * 16 + r;
    float cm = 0.f, ci = 0.f, cg = 1.f, cb = 0.f;
    if (stats == 2 && col < COUT) {
      cm = epi.mean[col];
      ci = epi.invstd[col];
      cg = epi.gamma ? epi.gamma[col] : 1.f;
      cb = epi.beta ? epi.beta[col] : 0.f;
    }
    // fp64 from the first addition on: E[x^2] - E[x]^2 downstream must stay exact to fp32 resolution
    double f1 = 0.0, f2 = 0.0;
#pragma unroll
    for (int m = 0; m < M; ++m) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = row0 + m * 16 + q * 4 + i;
        const bool ok = col < COUT && row < n_out;
        const uint32_t orow = (uint32_t)(row * groups + grp);
        float v = acc[m][nt][i];
        if (has_add) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_a, ok ? orow * lda4 + col * 4u : 0xFFFFFFFFu, 0, 0));
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_y, ok ? orow * ldy4 + col * 4u : 0xFFFFFFFFu, 0, 0);
        if (stats == 1) {
          const double vm = ok ? (double)v : 0.0;
          f1 += vm;
          f2 = fma(vm, vm, f2);
        } else if (stats == 2) {
          const float xb = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_b, ok ? orow * ldb4 + col * 4u : 0xFFFFFFFFu, 0, 0));
          const float xh = (xb - cm) * ci;
          const float t = fmaf(xh, cg, cb);
          const float dz = ok ? (t > 0.f ? v : v * epi.leak) : 0.f;
          f1 += (double)dz;
          f2 = fma((double)dz, (double)xh, f2);
        }
      }
    }
    s1[nt] += f1;
    s2[nt] += f2;
  }
}

// the workgroup's statistics partial from the lanes' running sums (conv_epilogue_rows, one or several row tiles)
template <int COUT, int NT>
__device__ __forceinline__ void conv_epilogue_stats(const double (&s1)[NT], const double (&s2)[NT], const ConvEpi &epi,
                                                    int stats, double *sred, size_t pblock) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  if (stats) {
    // lanes r, r+16, r+32, r+48 hold the same column: fold them, then the four waves in fixed order through LDS
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      double a = s1[nt], b = s2[nt];
      a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
      a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
      if (q == 0) {
        sred[(wave * 2 + 0) * NT * 16 + nt * 16 + r] = a;
        sred[(wave * 2 + 1) * NT * 16 + nt * 16 + r] = b;
      }
    }
    __syncthreads();
    for (int o = tid; o < 2 * NT * 16; o += 256) {
      const int which = o / (NT * 16), col = o % (NT * 16);
      if (col < COUT) {
        double t = 0.0;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) t += sred[(wv * 2 + which) * NT * 16 + col];
        epi.partial[(pblock * 2 + which) * COUT + col] = t;
      }
    }
  }
}

template <int COUT, int M, int NT>
__device__ __forceinline__ void conv_epilogue(f32x4 (&acc)[M][NT], int64_t row0, int64_t n_out, unsigned groups,
                                              unsigned grp, float *y, const ConvEpi &epi, int stats, double *sred,
                                              const float *any_ptr, size_t pblock) {
  double s1[NT], s2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) s1[nt] = s2[nt] = 0.0;
  conv_epilogue_rows<COUT, M, NT>(acc, row0, n_out, groups, grp, y, epi, stats, any_ptr, s1, s2);
  conv_epilogue_stats<COUT, NT>(s1, s2, epi, stats, sred, pblock);
}

// ---------------------------------------------------------------------------------------------------------------------
// Wide epilogue (round 5).  In the MFMA C layout a lane holds FOUR ROWS of one column, so the
// element-wise epilogue above issues one 4-byte memory instruction per accumulator element and operand: 16 stores per wave
// and 64-row tile in the forward pass, 16 + 16 loads (addend, BatchNorm input) on top in the data-gradient pass — 48
// instructions against the 108 gathers of the 27-offset walk, at the tail of the tile with nothing to overlap them.  In the
// step the dX launches of k_conv_fwd ran 25-45 % slower than forward launches of the same size for it
// (profiles/r04p_step_launches.csv).  Here the four lanes of a quad (columns 4c .. 4c+3, same rows) transpose their 4 x 4
// block with DPP quad permutes (16 VALU operations, no LDS): lane j of the quad then holds row q*4 + j, columns 4c .. 4c+3 —
// one 16-byte row-contiguous access per operand and 16-row tile, 4x fewer memory instructions — and the operand loads are
// issued BEFORE the last MFMA block of the tile (conv_epi_wide_prefetch), under the wait for the last gather.  The
// arithmetic stays in the MFMA layout (operands are transposed INTO it, results out of it): rows and statistics partials
// are bit-identical to the element-wise epilogue.
// Needs COUT % 4 == 0, one column tile (COUT <= 16), row strides that are multiples of 4 floats (the dispatcher checks).
// ---------------------------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float sgnn_quad_perm(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// in: lane j of a quad holds a[i] = B[i][j] (i = 0..3);  out: a[t] = B[j][t]
__device__ __forceinline__ void sgnn_quad_transpose4(float (&a)[4], int lane) {
  const bool odd = (lane & 1) != 0, hi = (lane & 2) != 0;
  float s, r;
  s = odd ? a[0] : a[1]; r = sgnn_quad_perm<0xB1>(s); a[0] = odd ? r : a[0]; a[1] = odd ? a[1] : r;   // lanes j <-> j ^ 1
  s = odd ? a[2] : a[3]; r = sgnn_quad_perm<0xB1>(s); a[2] = odd ? r : a[2]; a[3] = odd ? a[3] : r;
  s = hi ? a[0] : a[2];  r = sgnn_quad_perm<0x4E>(s); a[0] = hi ? r : a[0];  a[2] = hi ? a[2] : r;    // lanes j <-> j ^ 2
  s = hi ? a[1] : a[3];  r = sgnn_quad_perm<0x4E>(s); a[1] = hi ? r : a[1];  a[3] = hi ? a[3] : r;
}

template <int M>
struct EpiRows {          // the tile's epilogue operands in the transposed layout: [16-row tile][4 columns]
  float add[M][4], bnx[M][4];
};
#define SGNN_EPI_OOB 0xFFFFF800u   // slabs are limited to 0xFFFFF000 bytes: out of range, and +16 does not wrap

// the per-column BatchNorm constants of the backward statistics (stats == 2) into LDS: cst[which][col], which = mean,
// invstd, gamma, beta; call from all threads BEFORE a barrier that precedes the first epilogue
template <int COUT>
__device__ __forceinline__ void conv_epi_wide_constants(float *cst, const ConvEpi &epi, int stats) {
  const int tid = threadIdx.x;
  if (stats == 2 && tid < 64) {
    const int which = tid >> 4, col = tid & 15;
    const float *src = which == 0 ? epi.mean : (which == 1 ? epi.invstd : (which == 2 ? epi.gamma : epi.beta));
    float v = which == 2 ? 1.f : 0.f;
    if (src && col < COUT) v = src[col];
    cst[tid] = v;
  }
}

// which: bit 0 = the addend rows, bit 1 = the BatchNorm-input rows (SGNN_WEPI_EARLY of them are fetched under the last
// offset, the rest at the head of conv_epi_wide_finish)
#ifndef SGNN_WEPI_EARLY
#define SGNN_WEPI_EARLY 3
#endif
template <int COUT, int M>
__device__ __forceinline__ void conv_epi_wide_prefetch(EpiRows<M> &p, int64_t row0, int64_t n_out, const ConvEpi &epi,
                                                       int stats, const float *any_ptr, int which = 3) {
  const bool has_add = epi.addend != nullptr && (which & 1);
  if (!(which & 2)) stats = 0;
  if (!has_add && stats != 2) return;          // uniform over the launch
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const uint32_t c4 = (uint32_t)(r >> 2) * 4u;
  const int jr = r & 3;
  const __amdgpu_buffer_rsrc_t rs_a =
      make_rsrc(has_add ? epi.addend : any_ptr, has_add ? (uint32_t)(((n_out - 1) * epi.ld_add + COUT) * 4) : 0u);
  const __amdgpu_buffer_rsrc_t rs_b =
      make_rsrc(stats == 2 ? epi.bn_x : any_ptr, stats == 2 ? (uint32_t)(((n_out - 1) * epi.ld_bnx + COUT) * 4) : 0u);
  const uint32_t lda4 = (uint32_t)epi.ld_add * 4u, ldb4 = (uint32_t)epi.ld_bnx * 4u;
  // (the empty asm pins the address arithmetic below HERE: it is invariant in the offset loop, and hoisted in front of that
  //  loop it cost the whole walk ~25 VGPRs — one wave per SIMD of occupancy)
  uint32_t rbase = (uint32_t)row0 + (uint32_t)(q * 4 + jr);
  asm volatile("" : "+v"(rbase));
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int64_t row = (int64_t)(rbase + (uint32_t)(m * 16));
    const bool ok = c4 < (uint32_t)COUT && row < n_out;
    if (has_add) buf_load_floats<4>(rs_a, ok ? (uint32_t)row * lda4 + c4 * 4u : SGNN_EPI_OOB, p.add[m]);
    if (stats == 2) buf_load_floats<4>(rs_b, ok ? (uint32_t)row * ldb4 + c4 * 4u : SGNN_EPI_OOB, p.bnx[m]);
  }
}

// The tile's epilogue.  Arithmetic and statistics happen in the ORIGINAL MFMA layout (lane = one column, four rows), exactly
// as in conv_epilogue_rows — same fp32 operations per element, same fp64 summation order, so rows AND statistics partials are
// bit-identical to the element-wise epilogue; only the memory side is transposed: the operand rows arrive as 16-byte chunks
// and are quad-transposed into the MFMA layout, the result rows are quad-transposed back and leave as 16-byte chunks.
template <int COUT, int M>
__device__ __forceinline__ void conv_epi_wide_finish(f32x4 (&acc)[M][1], EpiRows<M> &p, int64_t row0, int64_t n_out,
                                                     float *y, const ConvEpi &epi, int stats, const float *cst,
                                                     double (&s1)[1], double (&s2)[1], const float *any_ptr) {
  if constexpr (SGNN_WEPI_EARLY != 3) conv_epi_wide_prefetch<COUT, M>(p, row0, n_out, epi, stats, any_ptr, 3 & ~SGNN_WEPI_EARLY);
  const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
  const uint32_t c4 = (uint32_t)(r >> 2) * 4u;
  const int jr = r & 3;
  const bool has_add = epi.addend != nullptr;
  const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(y, (uint32_t)(((n_out - 1) * epi.ldy + COUT) * 4));
  const uint32_t ldy4 = (uint32_t)epi.ldy * 4u;
  float cm = 0.f, ci = 0.f, cg = 1.f, cb = 0.f;
  if (stats == 2) {
    cm = cst[0 * 16 + r];
    ci = cst[1 * 16 + r];
    cg = cst[2 * 16 + r];
    cb = cst[3 * 16 + r];
  }
  double f1 = 0.0, f2 = 0.0;
  uint32_t rq = (uint32_t)row0 + (uint32_t)(q * 4);     // (pinned below the offset loop like the prefetch's addresses)
  asm volatile("" : "+v"(rq));
#pragma unroll
  for (int m = 0; m < M; ++m) {
    float v[4] = {acc[m][0][0], acc[m][0][1], acc[m][0][2], acc[m][0][3]};   // rows m*16 + q*4 + i, column r
    if (has_add) {
      float a[4] = {p.add[m][0], p.add[m][1], p.add[m][2], p.add[m][3]};
      sgnn_quad_transpose4(a, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += a[i];
    }
    if (stats == 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = r < COUT && (int64_t)(rq + (uint32_t)(m * 16 + i)) < n_out;
        const double vm = ok ? (double)v[i] : 0.0;
        f1 += vm;
        f2 = fma(vm, vm, f2);
      }
    } else if (stats == 2) {
      float xb[4] = {p.bnx[m][0], p.bnx[m][1], p.bnx[m][2], p.bnx[m][3]};
      sgnn_quad_transpose4(xb, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const bool ok = r < COUT && (int64_t)(rq + (uint32_t)(m * 16 + i)) < n_out;
        const float xh = (xb[i] - cm) * ci;
        const float t = fmaf(xh, cg, cb);
        const float dz = ok ? (t > 0.f ? v[i] : v[i] * epi.leak) : 0.f;
        f1 += (double)dz;
        f2 = fma((double)dz, (double)xh, f2);
      }
    }
    sgnn_quad_transpose4(v, lane);                                           // row m*16 + q*4 + jr, columns c4 .. c4+3
    const int64_t row = (int64_t)(rq + (uint32_t)(m * 16 + jr));
    const bool okt = c4 < (uint32_t)COUT && row < n_out;
    u32x4 o;
    o.x = __float_as_uint(v[0]); o.y = __float_as_uint(v[1]); o.z = __float_as_uint(v[2]); o.w = __float_as_uint(v[3]);
    __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, okt ? (uint32_t)row * ldy4 + c4 * 4u : SGNN_EPI_OOB, 0, 0);
    // one 16-row tile at a time: interleaving the four tiles' transposes and fp64 chains (what the scheduler does with the
    // unrolled loop) needs 4x the temporaries and costs the whole kernel a wave per SIMD of occupancy
    __builtin_amdgcn_sched_barrier(0);
  }
  s1[0] += f1;
  s2[0] += f2;
}
