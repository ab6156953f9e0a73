// The 256-row rulebook walk with ROW-CONTIGUOUS gathers (round 6): forward and data gradient of the 16-channel 3x3x3 layers
// on large levels (torch/model.py:38,40,180,255 — the FullyConvolutionalNet bodies and residual blocks).
//
// k_conv_fwd_w gathers straight into the MFMA A-fragment layout: lane (r = lane & 15, q = lane >> 4) fetches the q-th 16-byte
// chunk of row r, so every 16-lane pass of a gather instruction touches 16 different rows — 41 texture-path cycles per KiB
// (HISTORY.md 4a: 6 + 0.28 x rows per pass), the unit that binds that kernel (texture addresser busy 0.67, MFMA 0.52).  With
// lanes 4g .. 4g+3 covering ONE 64-byte row the same KiB costs 28 cycles.  Here the gathers use that mapping and the rows reach
// the fragment layout through a wave-private 2 KiB LDS tile (32 rows at a time): 4 ds_write_b128 in lane order + 4 ds_read_b128, chunk index XOR-ed
// with 2 for rows 8..15 of a 16-row tile so that the reads are conflict-free for the instruction's 16-lane groups (the writes
// stay whole 128-byte lines per 8 lanes).  The LDS was nearly idle in this kernel (one B-fragment read per offset).  Per
// wave-offset and CU: texture path 164 -> 112 cycles, LDS ~20 -> ~90, MFMA 128.  The LDS round trip of offset k + 1 runs under
// the MFMAs of offset k (two fragment sets), rows are gathered two offsets ahead (two landing sets); straight-line over the 27
// offsets; four waves per SIMD (LDS: 27.6 KB weights + 8 KB tiles per workgroup).  Same arithmetic in the same order as k_conv_fwd_w: rows bit-identical;
// statistics partials grouped by this kernel's own one-round tiling.  Measurements: profiles/r06q_*.
#include "common.h"

#include "conv_common.h"

#ifndef RC_UNROLL
#define RC_UNROLL 0     // 1: the 27 offsets as straight-line code (measured slower for <16,16>, like k_conv_fwd_u on the narrow layers)
#endif

template <int C, int K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_conv_fwd_rc(const float *__restrict__ x, int64_t n_in, const float *__restrict__ w,
                                                    const int32_t *__restrict__ table, int64_t ld, int64_t n_out,
                                                    float *y, int flags, ConvEpi epi, int wg_cap) {
  static_assert(C == 16, "16-channel rows: four 16-byte chunks");
  constexpr int M = 4, V = 4, RPW = 64;
  __shared__ __attribute__((aligned(16))) float wl[K * 256];
  __shared__ double sred[4 * 2 * 16];
  __shared__ float ecst[64];
  __shared__ __attribute__((aligned(16))) float gl[4 * 32 * C];        // wave-private row tiles: 32 rows (2 KiB) at a time

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  if (epi.n_dev) n_out = sgnn_dyn_n(n_out, epi.n_dev);
  int J = 1;                                  // one round of workgroups (k_conv_fwd explains)
  if (wg_cap > 0) {
    const int64_t w1 = (n_out + 4 * RPW - 1) / (4 * RPW);
    J = (int)((w1 + wg_cap - 1) / wg_cap);
    if (J < 1) J = 1;
  }
  unsigned nwg = gridDim.x;
  if (epi.n_dev || J > 1) {
    const int64_t rows_wg = (int64_t)4 * RPW * J;
    nwg = (unsigned)((n_out + rows_wg - 1) / rows_wg);
    if (blockIdx.x >= nwg) {
      if (epi.stats)
        for (int o = tid; o < 2 * C; o += 256) epi.partial[(size_t)blockIdx.x * 2 * C + o] = 0.0;
      return;
    }
  }
  const unsigned tile = sgnn_xcd_tile(blockIdx.x, nwg);
  int64_t row0 = ((int64_t)tile * J * 4 + wave) * RPW;
  const bool transpose = flags & SGNN_CONV_TRANSPOSE_W, flip = flags & SGNN_CONV_FLIP_K;

  const uint32_t ldx4 = (uint32_t)epi.ldx * 4u;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (uint32_t)(((n_in - 1) * epi.ldx + C) * 4));
  const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(table, (uint32_t)((int64_t)K * ld * 4));
  uint32_t lane_off = (uint32_t)(row0 + lane) * 4u;
  const uint32_t ld4 = (uint32_t)ld * 4u;
  // row-contiguous mapping of the gathers: instruction i covers rows 16 i + (lane >> 2), lane & 3 = the 16-byte chunk
  const int rrow = lane >> 2, rchunk = lane & 3;
  const int perm0 = rrow * 4;                                        // ds_bpermute byte address of the row's rule entry (+ 64 i)
  float *gw = gl + wave * (32 * C);
  float *g_wr = gw + rrow * C + ((rchunk ^ ((rrow >> 3) << 1)) << 2);      // + 256 (i & 1)   (rows 8..15 of a tile: chunk ^ 2)
  const float *g_rd = gw + r * C + ((q ^ ((r >> 3) << 1)) << 2);           // + 256 (m & 1)   (fragment: row 16 m + r, chunk q)

  f32x4 acc[M][1];
  double s1[1] = {0.0}, s2[1] = {0.0};
  conv_epi_wide_constants<C>(ecst, epi, epi.stats);
  conv_stage_weights<C, C>(wl, w, K, 0, K, transpose, flip);         // one chunk holds the whole filter (contains the barriers)

  auto idx_at = [&](int k) -> int32_t { return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_off, k * ld4, 0); };
  auto gather = [&](int32_t iv, float(&g)[M][V]) {
#pragma unroll
    for (int i = 0; i < M; ++i) {
      const int32_t id = __builtin_amdgcn_ds_bpermute(perm0 + 64 * i, iv);
      buf_load_floats<V>(rs_x, (uint32_t)id * ldx4 + (uint32_t)(rchunk * 16), g[i]);
    }
  };
  auto wave_sync = [] {      // lanes exchange data through the wave's tile; the LDS executes one wave's operations in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // landing layout -> LDS tile -> MFMA A fragments, 32 rows at a time through the wave's 2 KiB (the LDS executes one wave's
  // operations in order: the second half's writes land after the first half's reads)
  auto to_fragments = [&](const float(&g)[M][V], float(&a)[M][V]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      wave_sync();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        f32x4 v = {g[2 * h + i][0], g[2 * h + i][1], g[2 * h + i][2], g[2 * h + i][3]};
        *reinterpret_cast<f32x4 *>(g_wr + 256 * i) = v;
      }
      wave_sync();
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(g_rd + 256 * m);
        a[2 * h + m][0] = v[0]; a[2 * h + m][1] = v[1]; a[2 * h + m][2] = v[2]; a[2 * h + m][3] = v[3];
      }
    }
  };
  auto load_b = [&](int kk, float(&b)[V]) {
    const float *bp = wl + (kk * 16 + r) * 16 + q * V;
#pragma unroll
    for (int s = 0; s < V; ++s) b[s] = bp[s];
  };
  auto mma = [&](const float(&a)[M][V], const float(&b)[V]) {
#pragma unroll
    for (int s = 0; s < V; ++s)
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b[s], acc[m][0], 0, 0, 0);
  };
  const bool has_add = epi.addend != nullptr, has_bnx = epi.stats == 2;
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(has_add ? epi.addend : x, has_add ? (uint32_t)(((n_out - 1) * epi.ld_add + C) * 4) : 0u);
  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(has_bnx ? epi.bn_x : x, has_bnx ? (uint32_t)(((n_out - 1) * epi.ld_bnx + C) * 4) : 0u);
  const uint32_t lda4 = (uint32_t)epi.ld_add * 4u, ldb4 = (uint32_t)epi.ld_bnx * 4u;
  auto epi_prefetch = [&](EpiRows<M> &p) {     // branch-free (an absent operand is an out-of-range load = zeros)
    const uint32_t c4 = (uint32_t)(r >> 2) * 16u;
    const uint32_t rbase = (uint32_t)row0 + (uint32_t)(q * 4 + (r & 3));
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const uint32_t row = rbase + (uint32_t)(m * 16);
      const bool ok = (int64_t)row < n_out;
      buf_load_floats<4>(rs_a, ok ? row * lda4 + c4 : SGNN_EPI_OOB, p.add[m]);
      buf_load_floats<4>(rs_b, ok ? row * ldb4 + c4 : SGNN_EPI_OOB, p.bnx[m]);
    }
  };

  for (int j = 0; j < J; ++j) {
    if (j > 0) {
      const int64_t wg_row0 = ((int64_t)tile * J + j) * 4 * RPW;
      if (wg_row0 >= n_out) break;               // uniform over the workgroup
      row0 = wg_row0 + wave * RPW;
      lane_off = (uint32_t)(row0 + lane) * 4u;
    }
    EpiRows<M> erows;
#pragma unroll
    for (int m = 0; m < M; ++m) acc[m][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    // two landing sets (rows of offsets k + 1 and k + 2 in flight), two fragment sets (the LDS round trip of offset k + 1 under
    // the MFMAs of offset k)
    float g0[M][V], g1[M][V], a0[M][V], a1[M][V], b0[V], b1[V];
#if RC_UNROLL
    gather(idx_at(0), g0);
    gather(idx_at(1), g1);
    int32_t iv2 = K > 2 ? idx_at(2) : 0, iv3 = K > 3 ? idx_at(3) : 0;
    to_fragments(g0, a0);
    load_b(0, b0);
    if (K > 2) gather(iv2, g0);                   // rows of offset 2
    if (K > 4) iv2 = idx_at(4);
    // straight-line over the offsets (K is a template parameter): counted waits, nothing issued for offsets past the end
#pragma unroll
    for (int kk = 0; kk < K; kk += 2) {
      if (kk + 1 < K) {
        to_fragments(g1, a1);                     // offset kk + 1
        load_b(kk + 1, b1);
        if (kk + 3 < K) gather(iv3, g1);          // rows of offset kk + 3
        if (kk + 5 < K) iv3 = idx_at(kk + 5);
      } else {
        epi_prefetch(erows);                      // last offset: the epilogue's operand rows under its MFMAs
      }
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      if (kk + 1 < K) {
        if (kk + 2 < K) {
          to_fragments(g0, a0);                   // offset kk + 2
          load_b(kk + 2, b0);
          if (kk + 4 < K) gather(iv2, g0);        // rows of offset kk + 4
          if (kk + 6 < K) iv2 = idx_at(kk + 6);
        } else {
          epi_prefetch(erows);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#else
    // looped, branch-free (offsets past the end are clamped and their rows ignored), like k_conv_fwd
    constexpr int klast = K - 1;
    auto idx_c = [&](int k) -> int32_t {
      const int soff = __builtin_amdgcn_readfirstlane((int)((uint32_t)(k < klast ? k : klast) * ld4));
      return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_off, soff, 0);
    };
    gather(idx_c(0), g0);
    gather(idx_c(1), g1);
    int32_t iv2 = idx_c(2), iv3 = idx_c(3);
    to_fragments(g0, a0);
    load_b(0, b0);
    gather(iv2, g0);                              // rows of offset 2
    iv2 = idx_c(4);
    int kk = 0;
    for (; kk + 1 < K; kk += 2) {
      to_fragments(g1, a1);                       // offset kk + 1
      load_b(kk + 1, b1);
      gather(iv3, g1);                            // rows of offset kk + 3
      iv3 = idx_c(kk + 5);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, b0);
      __builtin_amdgcn_sched_barrier(0);
      to_fragments(g0, a0);                       // offset kk + 2
      load_b(kk + 2 < K ? kk + 2 : klast, b0);
      gather(iv2, g0);                            // rows of offset kk + 4
      iv2 = idx_c(kk + 6);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, b1);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    epi_prefetch(erows);                          // the epilogue's operand rows under the last offset
    __builtin_amdgcn_sched_barrier(0);
    if (kk < K) mma(a0, b0);
#endif
    conv_epi_wide_finish<C, M>(acc, erows, row0, n_out, y, epi, epi.stats, ecst, s1, s2, x);
  }
  conv_epilogue_stats<C, 1>(s1, s2, epi, epi.stats, sred, blockIdx.x);
}

// launch over n_out rows (256-row workgroups; capacity mode through epi.n_dev); false: not served (shape, strides, switch)
bool conv_wide_epi_ok(const ConvEpi &epi, const float *y, int K);   // conv.hip
bool sgnn_conv_rc_launch(const float *x, int64_t n_in, int cin, const float *w, int K, const int32_t *table, int64_t ld,
                         int64_t n_out, int cout, float *y, int flags, const ConvEpi &epi, hipStream_t s) {
  if (!g_tune.conv_row_gather || cin != 16 || cout != 16 || K != 27 || !g_tune.conv_one_round) return false;
  if (!conv_wide_epi_ok(epi, y, K)) return false;
  const unsigned grid4 = (unsigned)((n_out + CONV_ROWS_PER_BLOCK - 1) / CONV_ROWS_PER_BLOCK);
  SGNN_LAUNCH((k_conv_fwd_rc<16, 27>), dim3(grid4), dim3(256), 0, s, x, n_in, w, table, ld, n_out, y, flags, epi,
              conv_wg_capacity<k_conv_fwd_rc<16, 27>>());
  return true;
}
