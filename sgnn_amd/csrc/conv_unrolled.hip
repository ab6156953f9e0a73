// Sparse convolution, large levels, plain rulebook walk: the output-stationary MFMA kernel of conv.hip (k_conv_fwd) as
// STRAIGHT-LINE code — K (27 or 8 offsets) is a template parameter and the offset loop is fully unrolled.
//
// Serves scn.SubmanifoldConvolution / scn.Convolution forward and both data gradients on levels above ~40 k rows
// (torch/model.py:32,38,40,44,179,186,254).  Same arithmetic, same summation order as k_conv_fwd: bit-identical results.
//
// Why a second form of the same kernel (profiles/r04b_gather3.txt, scripts/kernels/gather_bench.hip k_loop): the looped
// kernel spends 68-75 us on the 366 k-row <16,16> level although its gathers alone take 42-46 us and its MFMAs 28-32 us,
// and a loop that does nothing but those gathers and 16 MFMAs per offset runs in 49 us.  What the loop form adds:
//  * every loop back-edge is a merge point for hipcc's s_waitcnt insertion: the first load use after it waits vmcnt(0), i.e.
//    the rows gathered for the NEXT offsets are drained once per trip (2 offsets);
//  * the wave's rule entries are loaded inside the loop, three offsets ahead, interleaved with the gathers they feed;
//  * tail handling (clamped offsets, dropped gathers) costs loads that are thrown away.
// Straight-line code has no merge points: every wait is a counted vmcnt(n) that covers exactly the register set about to
// be used.  The wave's K rule entries are loaded up front (K VGPRs, one coalesced 256-byte load each), the rows of offset
// k + 1 are gathered while the MFMAs of offset k run.  Weights are staged KC offsets at a time (<= 32 KiB of LDS) between
// two barriers at compile-time positions of the unrolled sequence; ordinary loads in flight survive a barrier.
#include "conv_common.h"

// WEPI: the wide (row-contiguous, 16-byte) epilogue of conv_common.h (output rows of 8 / 12 / 16 channels)
// (the body of k_conv_fwd_u and k_conv_fwd_uw below)
template <int CIN, int COUT, int K, bool WEPI>
__device__ __forceinline__ void conv_fwd_u_body(const float *__restrict__ x, int64_t n_in, const float *__restrict__ w,
                                                const int32_t *__restrict__ table, int64_t ld, int64_t n_out, float *y,
                                                int flags, int in_shift, const ConvEpi &epi, int wg_cap) {
  using C = ConvCfg<CIN, COUT>;
  constexpr int V = C::V, CINP = C::CINP, NT = C::NT, M = 4;
  constexpr int KC = C::KC < K ? C::KC : K;            // offsets per staged weight chunk
  __shared__ __attribute__((aligned(16))) float wl[KC * C::PER_K];
  __shared__ double sred[4 * 2 * NT * 16];             // statistics scratch (the weight tile stays live across row tiles)
  static_assert(!WEPI || (NT == 1 && COUT % 4 == 0), "wide epilogue: one column tile of whole 16-byte chunks");
  __shared__ float ecst[WEPI ? 64 : 1];                // WEPI: per-column BatchNorm constants of the backward statistics

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  unsigned nwg = gridDim.x;
  if (epi.n_dev) n_out = sgnn_dyn_n(n_out, epi.n_dev);   // capacity mode (see k_conv_fwd)
  // one round of workgroups: J consecutive 256-row tiles per live workgroup (k_conv_fwd explains)
  int J = 1;
  if (wg_cap > 0) {
    J = (int)(((n_out + 255) / 256 + wg_cap - 1) / wg_cap);
    if (J < 1) J = 1;
  }
  if (epi.n_dev || J > 1) {   // the live workgroups share the tiles like an exact-size launch
    nwg = (unsigned)((n_out + 256 * (int64_t)J - 1) / (256 * (int64_t)J));
    if (blockIdx.x >= nwg) {
      if (epi.stats)
        for (int o = tid; o < 2 * COUT; o += 256) epi.partial[(size_t)blockIdx.x * 2 * COUT + o] = 0.0;
      return;
    }
  }
  const unsigned tile = sgnn_xcd_tile(blockIdx.x, nwg);
  const bool transpose = flags & SGNN_CONV_TRANSPOSE_W, flip = flags & SGNN_CONV_FLIP_K;

  const uint32_t ldx4 = (uint32_t)epi.ldx * 4u;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (uint32_t)(((n_in - 1) * epi.ldx + CIN) * 4));
  const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(table, (uint32_t)((int64_t)K * ld * 4));
  const uint32_t ld4 = (uint32_t)ld * 4u;
  int perm[M];
#pragma unroll
  for (int m = 0; m < M; ++m) perm[m] = (m * 16 + r) * 4;
  double s1[NT], s2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) s1[nt] = s2[nt] = 0.0;
  int32_t idx[K];
  f32x4 acc[M][NT];
  if constexpr (WEPI) conv_epi_wide_constants<COUT>(ecst, epi, epi.stats);   // (visible after the barriers of the first stage())

  float a[2][M][V];               // ping-pong register sets: rows of offset k in a[k & 1]
  auto gather = [&](int k) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int32_t id = __builtin_amdgcn_ds_bpermute(perm[m], idx[k]);
      buf_load_floats<V>(rs_x, (uint32_t)id * ldx4 + (uint32_t)(q * V * 4), a[k & 1][m]);
    }
  };
  auto stage = [&](int k0) {   // weight slices k0 .. k0 + KC - 1 as wl[kk][n][c]
    conv_stage_weights<CIN, COUT>(wl, w, K, k0, (K - k0) < KC ? (K - k0) : KC, transpose, flip);
  };
  float bf[2][NT][V];          // B fragments, one offset ahead like the rows (bf[k & 1])
  auto load_b = [&](int k) {
    const int kk = k % KC;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float *bp = wl + (kk * NT * 16 + nt * 16 + r) * CINP + q * V;
#pragma unroll
      for (int s = 0; s < V; ++s) bf[k & 1][nt][s] = bp[s];
    }
  };
  auto mma = [&](int k) {
    const float(&b)[NT][V] = bf[k & 1];
#pragma unroll
    for (int s = 0; s < V; ++s)
#pragma unroll
      for (int m = 0; m < M; ++m) {
        float av = a[k & 1][m][s];
        if constexpr (CINP != CIN)       // the last quarter reads past the row end: those slots must be exact zeros
          if (3 * V + s >= CIN) av = (q == 3) ? 0.f : av;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[nt][s], acc[m][nt], 0, 0, 0);
      }
  };

  for (int j = 0; j < J; ++j) {
  const int64_t wg_row0 = ((int64_t)tile * J + j) * 256;
  if (j > 0 && wg_row0 >= n_out) break;       // uniform over the workgroup
  const int64_t row0 = wg_row0 + wave * 64;
  const uint32_t lane_off = (uint32_t)(row0 + lane) * 4u;
  // the wave's rule entries of every offset, loaded up front (padding rows of the table hold -1)
#pragma unroll
  for (int k = 0; k < K; ++k)
    idx[k] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_off, k * ld4, 0) >> in_shift;
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  gather(0);
  EpiRows<WEPI ? M : 1> erows;                 // WEPI: addend / BatchNorm-input rows of this tile, loaded under the last offset
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (k % KC == 0) {                        // compile-time positions (the loop is fully unrolled)
      if (KC < K || j == 0) stage(k);         // a tile that holds all K offsets is staged once
      load_b(k);                              // first offset of a freshly staged chunk: nothing to prefetch from
    }
    if (k + 1 < K) gather(k + 1);
    if (k + 1 < K && (k + 1) % KC != 0) load_b(k + 1);
    if constexpr (WEPI) {
      if (k == K - 1) conv_epi_wide_prefetch<COUT, M>(erows, row0, n_out, epi, epi.stats, x);
    }
    __builtin_amdgcn_sched_barrier(0);
    mma(k);
    __builtin_amdgcn_sched_barrier(0);
  }

  if constexpr (WEPI)
    conv_epi_wide_finish<COUT, M>(acc, erows, row0, n_out, y, epi, epi.stats, ecst, s1, s2, x);
  else
    conv_epilogue_rows<COUT, M, NT>(acc, row0, n_out, 1u, 0u, y, epi, epi.stats, x, s1, s2);
  }
  conv_epilogue_stats<COUT, NT>(s1, s2, epi, epi.stats, sred, blockIdx.x);
}

template <int CIN, int COUT, int K>
__global__ __launch_bounds__(256) void k_conv_fwd_u(const float *__restrict__ x, int64_t n_in, const float *__restrict__ w,
                                                   const int32_t *__restrict__ table, int64_t ld, int64_t n_out, float *y,
                                                   int flags, int in_shift, ConvEpi epi, int wg_cap) {
  conv_fwd_u_body<CIN, COUT, K, false>(x, n_in, w, table, ld, n_out, y, flags, in_shift, epi, wg_cap);
}

// the stride-2 (8-offset) walks with the wide epilogue: with 32 gathers per 64-row tile the element-wise epilogue's 16-48
// memory instructions weigh more than anywhere else.  Occupancy pinned like k_conv_fwd_w (conv.hip explains).  The 27-offset
// wide-row shapes keep the element-wise form: their walk already needs 190-240 registers.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_conv_fwd_uw(
    const float *__restrict__ x, int64_t n_in, const float *__restrict__ w, const int32_t *__restrict__ table, int64_t ld,
    int64_t n_out, float *y, int flags, int in_shift, ConvEpi epi, int wg_cap) {
  conv_fwd_u_body<CIN, COUT, 8, true>(x, n_in, w, table, ld, n_out, y, flags, in_shift, epi, wg_cap);
}

// Shapes: where the straight-line form wins on the 366 k-row level (profiles/r04c_conv_ab.txt, same box, bit-identical
// outputs): the wide-row layers <26,16> 132 -> 114 us, <16,26> 142 -> 119 us (and their 30 / 34-channel siblings), every
// stride-2 (8-offset) walk 5-10 %.  It LOSES on the narrow square layers (<16,16> 63-75 -> 72-90 us, <8,8>, <12,12>) whatever
// the register budget (K rule registers: 4 waves per SIMD; a rolling window: 5 waves, slower still) — those keep the looped kernel.
#define CONV_U_CASES_27(X) X(34, 16) X(16, 34) X(30, 16) X(16, 30) X(26, 16) X(16, 26)
#define CONV_U_CASES_8(X) X(8, 8) X(8, 12) X(12, 8) X(12, 12) X(12, 16) X(16, 12) X(16, 16)

bool sgnn_conv_u_supported(int cin, int cout, int K) {
#define X(CI, CO) \
  if (K == 27 && cin == CI && cout == CO) return true;
  CONV_U_CASES_27(X)
#undef X
#define X(CI, CO) \
  if (K == 8 && cin == CI && cout == CO) return true;
  CONV_U_CASES_8(X)
#undef X
  return false;
}

// launch over n_out rows (256-row workgroups; capacity mode through epi.n_dev); false: shape not compiled
bool conv_wide_epi_ok(const ConvEpi &epi, const float *y, int K);   // conv.hip

template <int CI, int CO, int KK>
static void conv_u_launch_plain(unsigned grid, hipStream_t s, const float *x, int64_t n_in, const float *w, const int32_t *table,
                                int64_t ld, int64_t n_out, float *y, int flags, int in_shift, const ConvEpi &epi) {
  if constexpr (KK == 8 && CO % 4 == 0 && CO <= 16) {
    if (conv_wide_epi_ok(epi, y, KK)) {
      SGNN_LAUNCH((k_conv_fwd_uw<CI, CO>), dim3(grid), dim3(256), 0, s, x, n_in, w, table, ld, n_out, y, flags, in_shift, epi,
                  g_tune.conv_one_round ? conv_wg_capacity<k_conv_fwd_uw<CI, CO>>() : 0);
      return;
    }
  }
  SGNN_LAUNCH((k_conv_fwd_u<CI, CO, KK>), dim3(grid), dim3(256), 0, s, x, n_in, w, table, ld, n_out, y, flags, in_shift,
              epi, g_tune.conv_one_round ? conv_wg_capacity<k_conv_fwd_u<CI, CO, KK>>() : 0);
}

bool sgnn_conv_u_launch(const float *x, int64_t n_in, int cin, const float *w, int K, const int32_t *table, int64_t ld,
                        int64_t n_out, int cout, float *y, int flags, int in_shift, const ConvEpi &epi, hipStream_t s) {
  const unsigned grid = (unsigned)((n_out + 255) / 256);
#define X(CI, CO)                                                                                    \
  if (K == 27 && cin == CI && cout == CO) {                                                          \
    conv_u_launch_plain<CI, CO, 27>(grid, s, x, n_in, w, table, ld, n_out, y, flags, in_shift, epi); \
    return true;                                                                                     \
  }
  CONV_U_CASES_27(X)
#undef X
#define X(CI, CO)                                                                                   \
  if (K == 8 && cin == CI && cout == CO) {                                                          \
    conv_u_launch_plain<CI, CO, 8>(grid, s, x, n_in, w, table, ld, n_out, y, flags, in_shift, epi); \
    return true;                                                                                    \
  }
  CONV_U_CASES_8(X)
#undef X
  return false;
}
