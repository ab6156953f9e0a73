// Sparse convolution on gfx950: output-stationary gather -> fp32 MFMA -> store.
//
// Serves scn.SubmanifoldConvolution (27 offsets) and scn.Convolution(...,2,2) (8 offsets)
// forward and both gradients (reference call sites torch/model.py:32,38,40,44,179,186,254 and
// the FullyConvolutionalNet bodies at :180,255).
//
// Design (MI355X-first, not upstream's per-offset gather/GEMM/scatter-add):
//  * the rulebook is an offset-major neighbour table table[k][j] (int32, -1 = no rule), so a
//    wave reads 64 consecutive rule entries of one offset as one coalesced 256-B run;
//  * each wave owns 16*MREP output rows and walks all K offsets keeping the 16x16 output tiles
//    in MFMA accumulators — no atomics, no scatter, deterministic summation order;
//  * the active-site feature slab is row-major; lane (r = lane&15, q = lane>>4) fetches the
//    q-th quarter of input row table[k][row r] with the widest aligned load, so the four lanes
//    of a row cover one contiguous feature row;
//  * the per-offset contraction is v_mfma_f32_16x16x4_f32: step s contracts the channel set
//    {q*V+s | q=0..3}; the weight slice W[k] is staged in LDS transposed ([n][c]) so the B
//    fragment of lane (n, q) is the same V contiguous floats;
//  * weights are staged KC offsets at a time to keep LDS <= 32 KiB (>= 4 workgroups per CU).
// Exact fp32 (MFMA f32 == fmaf chain), required for the 1e-4 logit tolerance.
#include <stdlib.h>
#include "common.h"

#include "conv_common.h"

// below this many 256-row workgroups the latency-oriented small-level kernel runs (sgnn_tune.conv_small_rows, ~40 k rows)
#define CONV_SMALL_GRID ((int64_t)((g_tune.conv_small_rows + CONV_ROWS_PER_BLOCK - 1) / CONV_ROWS_PER_BLOCK))

// EX = false: plain rulebook walk (ex is ignored; keeps the register budget of the hot instantiations)
// WEPI: the wide (row-contiguous, 16-byte) epilogue of conv_common.h instead of the element-wise one
// (the body of k_conv_fwd and k_conv_fwd_w below)
template <int CIN, int COUT, int M, bool EX, bool WEPI>
__device__ __forceinline__ void conv_fwd_body(const float *__restrict__ x, int64_t n_in,
                                              const float *__restrict__ w, const int32_t *__restrict__ table,
                                              int64_t ld, int K, int64_t n_out, float *y, int flags,
                                              int in_shift, const ConvEx &ex, const ConvEpi &epi, int wg_cap) {
  using C = ConvCfg<CIN, COUT>;
  constexpr int V = C::V, CINP = C::CINP, NT = C::NT, KC = C::KC;
  constexpr int RPW = 16 * M;   // rows per wave: M = 4 normally, 1 when the level is too small to fill the chip
  __shared__ __attribute__((aligned(16))) float wl[KC * C::PER_K];
  __shared__ double sred[4 * 2 * NT * 16];     // statistics scratch (the weight tile stays live across row tiles)
  static_assert(!WEPI || (NT == 1 && M == 4 && !EX && COUT % 4 == 0), "wide epilogue: one column tile of whole 16-byte chunks");
  __shared__ float ecst[WEPI ? 64 : 1];        // WEPI: per-column BatchNorm constants of the backward statistics

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  // all groups of a row tile run next to each other on one XCD (they gather the same feature rows)
  const unsigned groups = EX ? (unsigned)ex.groups : 1u;
  unsigned nwg = gridDim.x;
  if (epi.n_dev) n_out = sgnn_dyn_n(n_out, epi.n_dev);   // capacity mode: the live row count is on the device
  // One round of workgroups (round 4).  A workgroup walks all K offsets of its rows serially, so a launch with MORE
  // workgroups than the chip holds at once (wg_cap = resident workgroups of this instantiation x CUs) ends in a partial
  // second round whose workgroups run almost alone, at the latency of 27 dependent gather round trips instead of at MFMA
  // rate: 1641 workgroups on 1280 slots took 58 us where the MFMA work is 37 us.  Instead every live workgroup takes J
  // consecutive 256-row tiles, J the smallest count for which all of them are resident together; the weight tile is staged
  // once for all J.  The decomposition depends on the LIVE row count and wg_cap only, so a capacity-sized launch and an
  // exact one produce the same statistics partials.
  int J = 1;
  if constexpr (!EX) {
    if (wg_cap > 0) {
      const int64_t w1 = (n_out + 4 * RPW - 1) / (4 * RPW);
      J = (int)((w1 + wg_cap - 1) / wg_cap);
      if (J < 1) J = 1;
    }
  }
  if (epi.n_dev || J > 1) {
    // the LIVE workgroups are the first nwg in dispatch order (round robin over the XCDs) and share the tiles among
    // themselves exactly as an exact-size launch would: every XCD stays busy whatever the capacity's head-room
    const int64_t rows_wg = (int64_t)4 * RPW * J;
    nwg = (unsigned)((n_out + rows_wg - 1) / rows_wg) * groups;
    if (blockIdx.x >= nwg) {   // workgroup past the end: nothing to compute, zero statistics partials
      if (!EX && epi.stats)
        for (int o = tid; o < 2 * COUT; o += 256) epi.partial[(size_t)blockIdx.x * 2 * COUT + o] = 0.0;
      return;
    }
  }
  const unsigned lin = sgnn_xcd_tile(blockIdx.x, nwg);
  const unsigned tile = lin / groups, grp = lin % groups;
  int64_t row0 = ((int64_t)tile * J * 4 + wave) * RPW;  // < ld (ld is a multiple of 256)
  const int32_t *kmap = nullptr, *kadd_g = nullptr;
  if constexpr (EX) {
    w += (int64_t)grp * K * CIN * COUT;
    kmap = ex.kmap ? ex.kmap + grp * K : nullptr;
    kadd_g = ex.kadd ? ex.kadd + grp * K : nullptr;      // like kmap: one entry per (group, offset)
  }
  const int table_rows = EX ? ex.table_rows : K;
  const bool transpose = flags & SGNN_CONV_TRANSPOSE_W, flip = flags & SGNN_CONV_FLIP_K;

  const uint32_t ldx4 = (uint32_t)epi.ldx * 4u;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (uint32_t)(((n_in - 1) * epi.ldx + CIN) * 4));
  const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(table, (uint32_t)((int64_t)table_rows * ld * 4));
  uint32_t lane_off = (uint32_t)(row0 + (lane & (RPW - 1))) * 4u;   // this lane's rule entry in an offset row
  const uint32_t ld4 = (uint32_t)ld * 4u;
  int perm[M];                                               // ds_bpermute byte address of tile m's entry
#pragma unroll
  for (int m = 0; m < M; ++m) perm[m] = (m * 16 + r) * 4;

  f32x4 acc[M][NT];
  double s1[NT], s2[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) s1[nt] = s2[nt] = 0.0;
  if constexpr (WEPI) conv_epi_wide_constants<COUT>(ecst, epi, epi.stats);   // (visible after the barriers of the first stage())

  // one coalesced load fetches the wave's rule entries of an offset; lanes pick theirs with ds_bpermute
  // (the texture addresser, not HBM, is the scarce unit here: profiles/r01c_conv_pmc.txt)
  auto load_idx = [&](int k) -> int32_t {   // padding rows of the table hold -1
    if constexpr (EX) {
      const int trow = kmap ? kmap[k] : k;
      int32_t id = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_off, trow * ld4, 0);
      // fold the row transform in here (once per rule entry): -1 stays negative -> out of range -> zeros
      return (id >> in_shift) * ex.in_mul + ((kadd_g && id >= 0) ? kadd_g[k] : 0);
    } else {
      return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_off, k * ld4, 0) >> in_shift;
    }
  };
  auto gather = [&](int32_t iv, float(&a)[M][V]) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int32_t id = __builtin_amdgcn_ds_bpermute(perm[m], iv);
      buf_load_floats<V>(rs_x, (uint32_t)id * ldx4 + (uint32_t)(q * V * 4), a[m]);
    }
  };
  auto stage = [&](int k) {  // (re)stage KC weight slices as wl[kk][n][c]; in-flight global loads stay in flight
    conv_stage_weights<CIN, COUT>(wl, w, K, k, (K - k) < KC ? (K - k) : KC, transpose, flip);
  };
  auto load_b = [&](int kk, float(&b)[NT][V]) {     // B fragments of staged offset kk: W[kk][q*V .. q*V+V-1][n = nt*16 + r]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const float *bp = wl + (kk * NT * 16 + nt * 16 + r) * CINP + q * V;
#pragma unroll
      for (int s = 0; s < V; ++s) b[nt][s] = bp[s];
    }
  };
  auto mma_b = [&](float(&a)[M][V], const float(&b)[NT][V]) {
    if constexpr (CINP != CIN) {  // the last quarter reads past the row end: those slots must be exact zeros
#pragma unroll
      for (int m = 0; m < M; ++m)
#pragma unroll
        for (int s = 0; s < V; ++s)
          if (3 * V + s >= CIN) a[m][s] = (q == 3) ? 0.f : a[m][s];
    }
#pragma unroll
    for (int s = 0; s < V; ++s)
#pragma unroll
      for (int m = 0; m < M; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b[nt][s], acc[m][nt], 0, 0, 0);
  };
  auto mma = [&](int kk, float(&a)[M][V]) {
    float b[NT][V];
    load_b(kk, b);
    mma_b(a, b);
  };

  // software pipeline, unrolled by two with ping-pong registers: rule entries run three offsets ahead, gathered rows
  // one offset ahead of the MFMAs that consume them.  The loop body is branch-free on purpose: every load is issued
  // unconditionally (offsets past the end are clamped to the last one and their rows ignored), so the compiler can
  // give every use a COUNTED s_waitcnt vmcnt(n).  With loads under uniform branches (tail checks, weight restaging
  // inside the loop) it has to fall back to vmcnt(0) at the merge points, which drained the freshly issued gathers
  // of the next offset in every second step.
  const int klast = K - 1;
  auto idx_at = [&](int k) { return load_idx(k < klast ? k : klast); };
  for (int j = 0; j < J; ++j) {
  if (j > 0) {
    const int64_t wg_row0 = ((int64_t)tile * J + j) * 4 * RPW;
    if (wg_row0 >= n_out) break;               // uniform over the workgroup
    row0 = wg_row0 + wave * RPW;
    lane_off = (uint32_t)(row0 + (lane & (RPW - 1))) * 4u;
  }
  const bool restage = j == 0 || K > KC;       // a weight tile that holds all K offsets is staged once
  EpiRows<WEPI ? M : 1> erows;                 // WEPI: addend / BatchNorm-input rows of this tile, loaded under the last offset
#pragma unroll
  for (int m = 0; m < M; ++m)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (V <= 4 && (NT >= 2 || M == 1)) {
    // narrow rows, several output tiles (long MFMA phase per offset): three register sets, rows gathered TWO offsets
    // ahead of their MFMAs.  Measured at N = 366 k: <16,48> 228 -> 187 us; <16,16> (NT = 1) is 3 % faster with two sets
    float a0[M][V], a1[M][V], a2[M][V];
    for (int k0 = 0; k0 < K; k0 += KC) {      // one pass per staged weight chunk (a single one for the 3x3x3 16->16 layers)
      const int kc = (K - k0) < KC ? (K - k0) : KC;
      if (restage) stage(k0);
      gather(idx_at(k0), a0);
      gather(idx_at(k0 + 1), a1);
      int32_t iv2 = idx_at(k0 + 2), iv3 = idx_at(k0 + 3), iv4 = idx_at(k0 + 4);
      int kk = 0;
      for (; kk + 2 < kc; kk += 3) {
        // sched_barrier: the machine scheduler otherwise sinks the gathers to half an offset before their use
        gather(iv2, a2);                        // rows of offset k0+kk+2
        iv2 = idx_at(k0 + kk + 5);
        __builtin_amdgcn_sched_barrier(0);
        mma(kk, a0);
        __builtin_amdgcn_sched_barrier(0);
        gather(iv3, a0);                        // rows of offset k0+kk+3 (dropped if that is past this chunk)
        iv3 = idx_at(k0 + kk + 6);
        __builtin_amdgcn_sched_barrier(0);
        mma(kk + 1, a1);
        __builtin_amdgcn_sched_barrier(0);
        gather(iv4, a1);                        // rows of offset k0+kk+4
        iv4 = idx_at(k0 + kk + 7);
        __builtin_amdgcn_sched_barrier(0);
        mma(kk + 2, a2);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kk < kc) mma(kk, a0);
      if (kk + 1 < kc) mma(kk + 1, a1);
    }
  } else {
    // (round 4: the B fragments run one offset ahead too — their ds_read is issued before the MFMAs of the current offset
    //  instead of right in front of its own, where every offset paid the LDS latency behind an lgkmcnt(0))
    float a0[M][V], a1[M][V], b0[NT][V], b1[NT][V];
    // (WEPI launches have K <= KC — the dispatcher checks: ONE chunk, so that the epilogue operands loaded under its last
    //  offset are not values carried around a loop, which would keep their 32 registers allocated across the whole walk)
    for (int k0 = 0; k0 < (WEPI ? 1 : K); k0 += KC) {
      const int kc = (K - k0) < KC ? (K - k0) : KC;
      if (restage) stage(k0);
      int32_t iv1 = idx_at(k0 + 1), iv2 = idx_at(k0 + 2);
      gather(idx_at(k0), a0);
      load_b(0, b0);
      int kk = 0;
      for (; kk + 1 < kc; kk += 2) {
        gather(iv1, a1);                        // rows of offset k0+kk+1
        const int32_t iv3 = idx_at(k0 + kk + 3);
        load_b(kk + 1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mma_b(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        gather(iv2, a0);                        // rows of offset k0+kk+2 (dropped if that is past this chunk)
        iv1 = iv3;
        iv2 = idx_at(k0 + kk + 4);
        load_b(kk + 2 < kc ? kk + 2 : kc - 1, b0);
        __builtin_amdgcn_sched_barrier(0);
        mma_b(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (WEPI) {
        // the epilogue's operand rows, issued while the rows of the last offset are still in flight
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SGNN_WEPI_EARLY != 0) conv_epi_wide_prefetch<COUT, M>(erows, row0, n_out, epi, epi.stats, x, SGNN_WEPI_EARLY);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (kk < kc) mma_b(a0, b0);
    }
  }

  if constexpr (WEPI)
    conv_epi_wide_finish<COUT, M>(acc, erows, row0, n_out, y, epi, epi.stats, ecst, s1, s2, x);
  else
    conv_epilogue_rows<COUT, M, NT>(acc, row0, n_out, groups, grp, y, epi, EX ? 0 : epi.stats, x, s1, s2);
  }
  conv_epilogue_stats<COUT, NT>(s1, s2, epi, EX ? 0 : epi.stats, sred, blockIdx.x);
}

template <int CIN, int COUT, int M, bool EX>
__global__ __launch_bounds__(256) void k_conv_fwd(const float *__restrict__ x, int64_t n_in,
                                                 const float *__restrict__ w, const int32_t *__restrict__ table,
                                                 int64_t ld, int K, int64_t n_out, float *y, int flags,
                                                 int in_shift, ConvEx ex, ConvEpi epi, int wg_cap) {
  conv_fwd_body<CIN, COUT, M, EX, false>(x, n_in, w, table, ld, K, n_out, y, flags, in_shift, ex, epi, wg_cap);
}

// The 256-row kernel with the wide epilogue (plain walk, <= 16 channels either side, K <= 27).  Four waves per SIMD like the
// element-wise kernel (92 + 16 registers): left to itself the register allocator takes 120 + 24 for the same loop — the
// epilogue's operand rows raise the kernel's peak past 128, and once a wave per SIMD is gone anyway it relaxes the schedule of
// the offset walk as well — so the occupancy is pinned and the allocator has to fit the walk into the 128 it needs there.
template <int CIN, int COUT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_conv_fwd_w(
    const float *__restrict__ x, int64_t n_in, const float *__restrict__ w, const int32_t *__restrict__ table, int64_t ld,
    int K, int64_t n_out, float *y, int flags, int in_shift, ConvEx ex, ConvEpi epi, int wg_cap) {
  conv_fwd_body<CIN, COUT, CONV_MREP, false, true>(x, n_in, w, table, ld, K, n_out, y, flags, in_shift, ex, epi, wg_cap);
}

// ---------------------------------------------------------------------------
// Small levels (< ~40 k rows: every coarse U-Net level, ~60 % of all convolution launches of a step).  The kernel
// above walks the offsets serially with a short prefetch distance: on a level that cannot fill the chip its time is
// K dependent gather round trips (~0.4 us each, 11 us per launch at K = 27), not bandwidth and not MFMA.  Here a
// workgroup owns only 16 output rows and its four waves split the K offsets: every wave issues the rule loads, then
// ALL its gathers and weight fragments at once (one memory round trip each), runs its ~7 x V MFMAs on two independent
// accumulators, and the four partial tiles are summed through LDS.  Same arithmetic per (row, offset); the summation
// order over offsets differs from the big kernel (fp32 round-off only).  Plain rulebook walk only.
// ---------------------------------------------------------------------------
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_conv_small(const float *__restrict__ x, int64_t n_in,
                                                   const float *__restrict__ w, const int32_t *__restrict__ table,
                                                   int64_t ld, int K, int64_t n_out, float *y, int flags, int in_shift,
                                                   ConvEpi epi) {
  using C = ConvCfg<CIN, COUT>;
  constexpr int V = C::V, CINP = C::CINP, NT = C::NT;
  constexpr int KW = 7;                       // offsets per wave (K <= 28)
  __shared__ float red[4][NT * 256];          // the four waves' partial 16 x (NT*16) tiles
  __shared__ double sred[4][2][NT * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  if (epi.n_dev) {   // capacity mode (see k_conv_fwd)
    n_out = sgnn_dyn_n(n_out, epi.n_dev);
    if (row0 >= n_out) {
      if (epi.stats)
        for (int o = tid; o < 2 * COUT; o += 256) epi.partial[(size_t)blockIdx.x * 2 * COUT + o] = 0.0;
      return;
    }
  }
  const bool transpose = flags & SGNN_CONV_TRANSPOSE_W, flip = flags & SGNN_CONV_FLIP_K;
  const int per = (K + 3) >> 2;               // offsets of this wave: [k0, k0 + nk)
  const int k0 = wave * per;
  const int nk = (K - k0) < per ? ((K - k0) > 0 ? (K - k0) : 0) : per;

  const uint32_t ldx4 = (uint32_t)epi.ldx * 4u, ldy4 = (uint32_t)epi.ldy * 4u;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (uint32_t)(((n_in - 1) * epi.ldx + CIN) * 4));
  const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(table, (uint32_t)((int64_t)K * ld * 4));
  const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(y, (uint32_t)(((n_out - 1) * epi.ldy + COUT) * 4));
  const uint32_t lane_off = (uint32_t)(row0 + r) * 4u;
  const uint32_t ld4 = (uint32_t)ld * 4u;

  // one round trip: the rule entries of all offsets of this wave (offsets past nk are clamped and ignored)
  int32_t id[KW];
#pragma unroll
  for (int kk = 0; kk < KW; ++kk) {
    const int k = k0 + (kk < nk ? kk : (nk > 0 ? nk - 1 : 0));
    id[kk] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_off, (k < K ? k : K - 1) * ld4, 0) >> in_shift;
  }
  // second round trip: every gathered row quarter + (independent of the rules) the weight fragments of these offsets
  float a[KW][V];
#pragma unroll
  for (int kk = 0; kk < KW; ++kk) buf_load_floats<V>(rs_x, (uint32_t)id[kk] * ldx4 + (uint32_t)(q * V * 4), a[kk]);
  float b[KW][NT][V];
  if (transpose) {
    // data gradient: the weights are read as (K, COUT, CIN), so a lane's V channels are CONTIGUOUS — one wide load per
    // (offset, column tile) instead of V four-byte loads (round 4: the weight fragments were 28 of the 35 memory instructions
    // a wave of the <16,16> kernel issues, each as expensive for the texture path as a gather)
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(w, (uint32_t)((int64_t)K * CIN * COUT * 4));
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      const int k = k0 + kk;
      const int ks = flip ? (K - 1 - k) : k;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + r;
        const bool okw = kk < nk && n < COUT;
        buf_load_floats<V>(rs_w, okw ? (uint32_t)((ks * COUT + n) * CIN + q * V) * 4u : 0xFFFFF800u, b[kk][nt]);
        if constexpr (CINP != CIN) {   // the last quarter runs into the next column's weights: those slots are zero
#pragma unroll
          for (int s = 0; s < V; ++s)
            if (3 * V + s >= CIN) b[kk][nt][s] = (q == 3) ? 0.f : b[kk][nt][s];
        }
      }
    }
  } else if constexpr (V == 4 && CIN <= 16 && (CIN * COUT) % 4 == 0) {
    // forward layout (K, CIN, COUT): a lane's V channels are COUT floats apart, so the fragments cannot be loaded wide.  The
    // slice is fetched whole instead — 16 bytes per lane in global order, fully coalesced — and turned into fragments through
    // the wave's own corner of the reduction buffer (LDS operations of one wave complete in order: no barrier; the buffer is
    // not used for the partial tiles until after the MFMAs).  7 wide loads per wave where there were 28 strided ones.
    // Measured on a 14.5 k-row level: <16,16> 7.7 -> 6.9 us; the narrower layers (V = 2, 3: 14 / 21 strided loads) lose 0.4 us
    // to the LDS hop and keep the direct loads below.
    constexpr int SL = CIN * COUT, G4 = SL / 4, NL = (G4 + 63) / 64;
    static_assert(SL <= NT * 256, "the weight slice must fit the wave's partial-tile buffer");
    float *ws = red[wave];
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(w, (uint32_t)((int64_t)K * SL * 4));
    float g[KW][NL][4];
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      const int k = k0 + kk;
      const int ks = flip ? (K - 1 - k) : k;
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const int e4 = l * 64 + lane;
        buf_load_floats<4>(rs_w, (kk < nk && e4 < G4) ? (uint32_t)(ks * SL + e4 * 4) * 4u : 0xFFFFF800u, g[kk][l]);
      }
    }
    // (the lanes exchange data through ws: wave-scope fences tell the compiler so — without them it keeps a lane that
    //  wrote nothing from re-reading.  They cost no instruction: the LDS executes a wave's operations in order)
    auto wave_sync = [] {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      wave_sync();                               // the fragment reads of the previous offset are ordered before these writes
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        const int e4 = l * 64 + lane;
        if (e4 < G4) {
#pragma unroll
          for (int j = 0; j < 4; ++j) ws[e4 * 4 + j] = g[kk][l][j];
        }
      }
      wave_sync();
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + r;
#pragma unroll
        for (int s = 0; s < V; ++s) {
          const int c = q * V + s;
          b[kk][nt][s] = (c < CIN && n < COUT) ? ws[c * COUT + n] : 0.f;     // offsets past nk: zeros were loaded
        }
      }
    }
  } else {
#pragma unroll
    for (int kk = 0; kk < KW; ++kk) {
      const int k = k0 + kk;
      const int ks = flip ? (K - 1 - k) : k;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = nt * 16 + r;
#pragma unroll
        for (int s = 0; s < V; ++s) {
          const int c = q * V + s;
          float v = 0.f;
          if (kk < nk && c < CIN && n < COUT) v = w[((int64_t)ks * CIN + c) * COUT + n];
          b[kk][nt][s] = v;
        }
      }
    }
  }
  if constexpr (CINP != CIN) {
#pragma unroll
    for (int kk = 0; kk < KW; ++kk)
#pragma unroll
      for (int s = 0; s < V; ++s)
        if (3 * V + s >= CIN) a[kk][s] = (q == 3) ? 0.f : a[kk][s];
  }
  f32x4 acc[2][NT];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[h][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < KW; ++kk) {
    if (kk < nk) {   // wave-uniform
#pragma unroll
      for (int s = 0; s < V; ++s)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[kk & 1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk][s], b[kk][nt][s], acc[kk & 1][nt], 0, 0, 0);
    }
  }
  // C/D layout: col = lane&15, row = (lane>>4)*4 + reg  ->  red[wave][nt][row][col]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][nt * 256 + (q * 4 + i) * 16 + r] = acc[0][nt][i] + acc[1][nt][i];
  __syncthreads();

  // one output element per thread and column tile: row = tid / 16, col = nt*16 + tid % 16
  const bool has_add = epi.addend != nullptr;
  const __amdgpu_buffer_rsrc_t rs_a =
      make_rsrc(has_add ? epi.addend : x, has_add ? (uint32_t)(((n_out - 1) * epi.ld_add + COUT) * 4) : 0u);
  const uint32_t lda4 = (uint32_t)epi.ld_add * 4u;
  const int stats = epi.stats;
  const __amdgpu_buffer_rsrc_t rs_b =
      make_rsrc(stats == 2 ? epi.bn_x : x, stats == 2 ? (uint32_t)(((n_out - 1) * epi.ld_bnx + COUT) * 4) : 0u);
  const uint32_t ldb4 = (uint32_t)epi.ld_bnx * 4u;
  const int orow_l = tid >> 4, ocol_l = tid & 15;
  const int64_t row = row0 + orow_l;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = nt * 16 + ocol_l;
    const bool ok = col < COUT && row < n_out;
    const int e = nt * 256 + orow_l * 16 + ocol_l;
    float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    const uint32_t orow = (uint32_t)row;
    if (has_add) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_a, ok ? orow * lda4 + col * 4u : 0xFFFFFFFFu, 0, 0));
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_y, ok ? orow * ldy4 + col * 4u : 0xFFFFFFFFu, 0, 0);
    if (stats) {
      double f1 = 0.0, f2 = 0.0;
      if (stats == 1) {
        f1 = ok ? (double)v : 0.0;
        f2 = f1 * f1;
      } else {
        float cm = 0.f, ci = 0.f, cg = 1.f, cb = 0.f;
        if (col < COUT) {
          cm = epi.mean[col];
          ci = epi.invstd[col];
          cg = epi.gamma ? epi.gamma[col] : 1.f;
          cb = epi.beta ? epi.beta[col] : 0.f;
        }
        const float xb = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_b, ok ? orow * ldb4 + col * 4u : 0xFFFFFFFFu, 0, 0));
        const float xh = (xb - cm) * ci;
        const float t = fmaf(xh, cg, cb);
        const float dz = ok ? (t > 0.f ? v : v * epi.leak) : 0.f;
        f1 = (double)dz;
        f2 = (double)dz * (double)xh;
      }
      // a wave holds rows 4*wave .. 4*wave+3 x 16 columns: fold the 4 rows (lanes c, c+16, c+32, c+48), then the waves
      f1 += __shfl_xor(f1, 16); f2 += __shfl_xor(f2, 16);
      f1 += __shfl_xor(f1, 32); f2 += __shfl_xor(f2, 32);
      if (lane < 16) {
        sred[wave][0][nt * 16 + lane] = f1;
        sred[wave][1][nt * 16 + lane] = f2;
      }
    }
  }
  if (stats) {
    __syncthreads();
    for (int o = tid; o < 2 * NT * 16; o += 256) {
      const int which = o / (NT * 16), col = o % (NT * 16);
      if (col < COUT)
        epi.partial[((size_t)blockIdx.x * 2 + which) * COUT + col] =
            (sred[0][which][col] + sred[1][which][col]) + (sred[2][which][col] + sred[3][which][col]);
    }
  }
}

// any (cin, cout): one thread per output element, plain FMA.  Correctness fallback for layer
// widths outside the SG-NN set; not a performance path.
__global__ __launch_bounds__(256) void k_conv_fwd_generic(const float *__restrict__ x, int cin,
                                                         const float *__restrict__ w, int K,
                                                         const int32_t *__restrict__ table, int64_t ld,
                                                         int64_t n_out, int cout, float *__restrict__ y,
                                                         int flags, int in_shift, ConvEx ex, const int64_t *n_dev) {
  n_out = sgnn_dyn_n(n_out, n_dev);
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_out * ex.groups * cout) return;
  const int64_t orow = t / cout;
  const int n = (int)(t - orow * cout);
  const int64_t row = orow / ex.groups;
  const int grp = (int)(orow - row * ex.groups);
  w += (int64_t)grp * K * cin * cout;
  const bool transpose = flags & SGNN_CONV_TRANSPOSE_W, flip = flags & SGNN_CONV_FLIP_K;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    const int32_t idx = table[(int64_t)(ex.kmap ? ex.kmap[grp * K + k] : k) * ld + row];
    if (idx < 0) continue;
    const float *xr = x + ((int64_t)(idx >> in_shift) * ex.in_mul + (ex.kadd ? ex.kadd[grp * K + k] : 0)) * cin;
    const int ks = flip ? (K - 1 - k) : k;
    for (int c = 0; c < cin; ++c) {
      const float wv = transpose ? w[((int64_t)ks * cout + n) * cin + c] : w[((int64_t)ks * cin + c) * cout + n];
      acc = fmaf(xr[c], wv, acc);
    }
  }
  y[t] = acc;
}

// shapes of the generative up-sampling convolution (grouped / remapped walk): forward and data gradient
// + the dense-bottleneck shapes: their k4s2 convolutions run offset-split (groups = slices of the 64 taps, partial
// outputs summed by sgnn_sum_groups) because a 2 048-row level would otherwise be 32 workgroups walking 64 offsets
#define CONV_EX_CASES(X) \
  X(48, 16) X(16, 48) X(24, 8) X(8, 24) X(16, 24) X(24, 16) X(24, 32) X(32, 24) X(64, 32) X(32, 64) X(56, 28) X(28, 56)

#define CONV_FWD_CASES(X) \
  X(1, 8) X(8, 8) X(8, 12) X(12, 12) X(12, 16) X(16, 16) X(34, 16) X(30, 16) X(26, 16) X(48, 16) \
  X(8, 1) X(12, 8) X(16, 12) X(16, 34) X(16, 30) X(16, 26) X(16, 48) X(32, 16) X(16, 32) X(4, 16) X(16, 4) \
  X(16, 24) X(24, 16) X(24, 32) X(32, 24) X(64, 32) X(32, 64) X(56, 28) X(28, 56) X(32, 32) X(28, 16) X(16, 28)

// number of workgroups (= statistics partial blocks) a plain launch over n_out rows uses
int64_t sgnn_conv_grid_blocks(int64_t n_out, int cin, int cout, int K) {
  const int64_t grid4 = (n_out + CONV_ROWS_PER_BLOCK - 1) / CONV_ROWS_PER_BLOCK;
  if (grid4 < CONV_SMALL_GRID) return (n_out + 15) / 16;         // k_conv_small: 16 rows per workgroup
  return grid4;
}


// conv_unrolled.hip: the large-level kernel as straight-line code (plain rulebook walk, K = 27 / 8)
bool sgnn_conv_u_supported(int cin, int cout, int K);
bool sgnn_conv_u_launch(const float *x, int64_t n_in, int cin, const float *w, int K, const int32_t *table, int64_t ld,
                        int64_t n_out, int cout, float *y, int flags, int in_shift, const ConvEpi &epi, hipStream_t s);

// the wide epilogue moves whole 16-byte chunks of a row: every row stride it touches must be a multiple of 4 floats and
// the bases 16-byte aligned (views into a JoinTable buffer start at multiples of 16 columns; odd test strides fall back)
bool conv_wide_epi_ok(const ConvEpi &epi, const float *y, int K) {
  if (!g_tune.conv_wide_epi || K > 27 || (epi.ldy & 3) || ((uintptr_t)y & 15)) return false;   // (27 = ConvCfg::KC for <= 16 channels)
  if (epi.addend && ((epi.ld_add & 3) || ((uintptr_t)epi.addend & 15))) return false;
  if (epi.stats == 2 && ((epi.ld_bnx & 3) || ((uintptr_t)epi.bn_x & 15))) return false;
  return true;
}
// the 256-row kernel of a level: wide epilogue where the shape and the strides allow it
template <int CI, int CO, bool EXV>
static void conv_launch_big(unsigned grid4, hipStream_t s, const float *x, int64_t n_in, const float *w, const int32_t *table,
                            int64_t ld, int K, int64_t n_out, float *y, int flags, int in_shift, const ConvEx &ex,
                            const ConvEpi &epi) {
  if constexpr (!EXV && CO % 4 == 0 && CO <= 16 && CI <= 16) {
    static_assert(ConvCfg<CI, CO>::KC >= 27, "one weight chunk holds a 3x3x3 filter");
    if (conv_wide_epi_ok(epi, y, K)) {
      SGNN_LAUNCH((k_conv_fwd_w<CI, CO>), dim3(grid4), dim3(256), 0, s, x, n_in, w, table, ld, K, n_out, y, flags, in_shift,
                  ex, epi, !g_tune.conv_one_round ? 0 : conv_wg_capacity<k_conv_fwd_w<CI, CO>>());
      return;
    }
  }
  SGNN_LAUNCH((k_conv_fwd<CI, CO, CONV_MREP, EXV>), dim3(grid4), dim3(256), 0, s, x, n_in, w, table, ld, K, n_out, y,
              flags, in_shift, ex, epi,
              (EXV || !g_tune.conv_one_round) ? 0 : conv_wg_capacity<k_conv_fwd<CI, CO, CONV_MREP, EXV>>());
}

bool sgnn_conv_epi_supported(int cin, int cout) {
#define X(CI, CO) \
  if (cin == CI && cout == CO) return true;
  CONV_FWD_CASES(X)
#undef X
  return false;
}

// the one implementation behind sgnn_conv_fwd / _ex / _epi (epi == NULL: contiguous rows, plain store)
int sgnn_conv_fwd_impl(const float *x, int64_t n_in, int cin, const float *w, int K, const int32_t *table, int64_t ld,
                       int64_t n_out, int cout, float *y, int flags, int in_shift, const int32_t *kmap,
                       const int32_t *kadd, int in_mul, int groups, int table_rows, const ConvEpi *epi_in,
                       sgnn_stream_t stream) {
  SGNN_CHECK_ARG(cin >= 1 && cout >= 1 && K >= 1 && K <= 64 && n_out >= 0 && ld >= n_out && in_shift >= 0 &&
                 in_shift < 31 && in_mul >= 1 && groups >= 1 && groups <= 64 && table_rows >= 1 &&
                 table_rows <= 64 && (kmap || table_rows >= K));
  if (n_out == 0) return SGNN_OK;
  SGNN_CHECK_ARG(x && w && table && y && n_in >= 1);
  SGNN_CHECK_ARG(ld % CONV_ROWS_PER_BLOCK == 0);  // and table[k][n_out..ld) must be -1 (see sgnn_hip.h)
  ConvEpi epi{};
  if (epi_in) epi = *epi_in;
  if (epi.ldx <= 0) epi.ldx = cin;
  if (epi.ldy <= 0) epi.ldy = cout;
  if (epi.ld_add <= 0) epi.ld_add = cout;
  if (epi.ld_bnx <= 0) epi.ld_bnx = cout;
  const bool plain = !kmap && !kadd && in_mul == 1 && groups == 1 && table_rows == K;
  const bool has_epi = epi.ldx != cin || epi.ldy != cout || epi.addend || epi.stats;
  SGNN_CHECK_ARG(epi.ldx >= cin && epi.ldx <= 1024 && epi.ldy >= cout && epi.ldy <= 1024 && epi.ld_add >= cout &&
                 epi.ld_add <= 1024 && epi.ld_bnx >= cout && epi.ld_bnx <= 1024);
  SGNN_CHECK_ARG(epi.stats >= 0 && epi.stats <= 2 && (!epi.stats || (plain && epi.partial)));
  SGNN_CHECK_ARG(epi.stats != 2 || (epi.bn_x && epi.mean && epi.invstd));
  SGNN_CHECK_ARG(!epi.stats || K <= 28);   // the statistics partial count assumes the small-level kernel below ~40 k rows
  const int64_t lmax = epi.ldy > epi.ld_add ? (epi.ldy > epi.ld_bnx ? epi.ldy : epi.ld_bnx)
                                            : (epi.ld_add > epi.ld_bnx ? epi.ld_add : epi.ld_bnx);
  if (n_in * epi.ldx * 4 > 0xFFFFF000ll || n_out * groups * lmax * 4 > 0xFFFFF000ll ||
      (int64_t)table_rows * ld * 4 > 0xFFFFF000ll) {
    sgnn_set_error("sgnn_conv_fwd: a slab exceeds the 4 GiB raw-buffer window (n_in=%lld cin=%d n_out=%lld cout=%d)",
                   (long long)n_in, cin, (long long)n_out, cout);
    return SGNN_EOVERFLOW;
  }
  hipStream_t s = (hipStream_t)stream;
  const ConvEx ex{kmap, kadd, in_mul, groups, table_rows};
  const unsigned grid4 = (unsigned)((n_out + CONV_ROWS_PER_BLOCK - 1) / CONV_ROWS_PER_BLOCK) * (unsigned)groups;
  const unsigned grid1 = (unsigned)((n_out + 63) / 64) * (unsigned)groups;
  const bool small = grid4 < CONV_SMALL_GRID;   // too few 256-row workgroups for 256 CUs: 64-row workgroups
  bool done = false;
  const int prof = sgnn_prof_begin_launch(0, n_out * groups, cin, cout, K, flags, s);
#define LAUNCH_FWD(CI, CO, EXV)                                                                         \
  do {                                                                                                  \
    if (small && !EXV && K <= 28 && (g_tune.conv_small || epi.stats))                                      \
      SGNN_LAUNCH((k_conv_small<CI, CO>), dim3((unsigned)((n_out + 15) / 16)), dim3(256), 0, s,         \
                         x, n_in, w, table, ld, K, n_out, y, flags, in_shift, epi);                     \
    else if (small)                                                                                     \
      SGNN_LAUNCH((k_conv_fwd<CI, CO, 1, EXV>), dim3(grid1), dim3(256), 0, s, x, n_in, w, table,        \
                         ld, K, n_out, y, flags, in_shift, ex, epi, 0);                                 \
    else                                                                                                \
      conv_launch_big<CI, CO, EXV>(grid4, s, x, n_in, w, table, ld, K, n_out, y, flags, in_shift, ex, epi); \
    done = true;                                                                                        \
  } while (0)
  if (plain && !small && g_tune.conv_unrolled && sgnn_conv_u_supported(cin, cout, K))
    done = sgnn_conv_u_launch(x, n_in, cin, w, K, table, ld, n_out, cout, y, flags, in_shift, epi, s);
#define X(CI, CO) \
  if (!done && plain && cin == CI && cout == CO) LAUNCH_FWD(CI, CO, false);
  CONV_FWD_CASES(X)
#undef X
#define X(CI, CO) \
  if (!done && !plain && cin == CI && cout == CO) LAUNCH_FWD(CI, CO, true);
  CONV_EX_CASES(X)
#undef X
  if (!done) {
    if (has_epi) {
      sgnn_set_error("sgnn_conv_fwd_epi: strided / fused epilogues need one of the compiled (cin, cout) shapes, got (%d, %d)", cin, cout);
      return SGNN_EINVAL;
    }
    const int64_t total = n_out * groups * cout;
    SGNN_LAUNCH(k_conv_fwd_generic, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, cin, w, K,
                       table, ld, n_out, cout, y, flags, in_shift, ex, epi.n_dev);
  }
  sgnn_prof_end_launch(prof, s);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_conv_fwd_ex(const float *x, int64_t n_in, int cin, const float *w, int K, const int32_t *table,
                                 int64_t ld, int64_t n_out, int cout, float *y, int flags, int in_shift,
                                 const int32_t *kmap, const int32_t *kadd, int in_mul, int groups,
                                 int table_rows, sgnn_stream_t stream) {
  return sgnn_conv_fwd_impl(x, n_in, cin, w, K, table, ld, n_out, cout, y, flags, in_shift, kmap, kadd, in_mul,
                            groups, table_rows, nullptr, stream);
}

SGNN_EXPORT int sgnn_conv_fwd(const float *x, int64_t n_in, int cin, const float *w, int K, const int32_t *table,
                              int64_t ld, int64_t n_out, int cout, float *y, int flags, int in_shift,
                              sgnn_stream_t stream) {
  return sgnn_conv_fwd_impl(x, n_in, cin, w, K, table, ld, n_out, cout, y, flags, in_shift, nullptr, nullptr, 1, 1,
                            K, nullptr, stream);
}

SGNN_EXPORT int64_t sgnn_conv_stats_blocks(int64_t n_out) { return n_out > 0 ? sgnn_conv_grid_blocks(n_out, 0, 0, 0) : 0; }

// plain rulebook walk with strided rows and a fused epilogue (see ConvEpi; sgnn_hip.h)
SGNN_EXPORT int sgnn_conv_fwd_epi(const float *x, int64_t n_in, int cin, int64_t ldx, const float *w, int K,
                                  const int32_t *table, int64_t ld, int64_t n_out, int cout, float *y, int64_t ldy,
                                  int flags, const float *addend, int64_t ld_add, int stats, double *partial,
                                  const float *bn_x, int64_t ld_bnx, const float *mean, const float *invstd,
                                  const float *gamma, const float *beta, float leak, sgnn_stream_t stream) {
  ConvEpi epi{ldx, ldy, ld_add, addend, stats, partial, bn_x, ld_bnx, mean, invstd, gamma, beta, leak, nullptr};
  return sgnn_conv_fwd_impl(x, n_in, cin, w, K, table, ld, n_out, cout, y, flags, 0, nullptr, nullptr, 1, 1, K, &epi,
                            stream);
}

// ---------------------------------------------------------------------------
// Generative up-sampling convolution (sgnn_hip.h, sgnn_conv_fwd_ex): constants and weight transforms.
// Child parity g = 4*jz+2*jy+jx, parent-level offset slot i = 4*iz+2*iy+ix; per axis the parent offset is
// o = i - 1 + j and the 3x3x3 taps d of a child that fall into that parent are
//   (j,i) = (0,0): {-1}   (0,1): {0,+1}   (1,0): {-1,0}   (1,1): {+1}.
// S[g*8+i] = parent-table row of (g,i), ST = 26 - S (mirrored row: data gradient), PAR = g.
// ---------------------------------------------------------------------------
__device__ int32_t g_expand_maps[192];

// device pointers to S, ST, PAR (64 ints each); the table is uploaded on first use
int sgnn_expand_maps(const int32_t **S, const int32_t **ST, const int32_t **PAR) {
  static const int32_t *base = nullptr;
  if (!base) {
    int32_t host[192];
    for (int g = 0; g < 8; ++g)
      for (int i = 0; i < 8; ++i) {
        const int o0 = ((i >> 2) & 1) - 1 + ((g >> 2) & 1), o1 = ((i >> 1) & 1) - 1 + ((g >> 1) & 1),
                  o2 = (i & 1) - 1 + (g & 1);
        const int srow = (o0 + 1) * 9 + (o1 + 1) * 3 + (o2 + 1);
        host[g * 8 + i] = srow;
        host[64 + g * 8 + i] = 26 - srow;
        host[128 + g * 8 + i] = g;
      }
    SGNN_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_expand_maps), host, sizeof(host)));
    void *p = nullptr;
    SGNN_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(g_expand_maps)));
    base = (const int32_t *)p;
  }
  *S = base;
  *ST = base + 64;
  *PAR = base + 128;
  return SGNN_OK;
}

__device__ __forceinline__ void axis_taps(int j, int i, int &lo, int &hi) {   // taps d in [lo, hi]
  if (j == 0) { lo = i ? 0 : -1; hi = i ? 1 : -1; }
  else        { lo = i ? 1 : -1; hi = i ? 1 : 0; }
}

// Wc[g*8+i] = sum of the 3x3x3 taps of child parity g that fall into parent offset slot i   (64 x cin*cout)
__global__ __launch_bounds__(256) void k_expand_weights(const float *__restrict__ w, int cc, float *__restrict__ wc) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 64 * cc) return;
  const int gi = t / cc, e = t - gi * cc, g = gi >> 3, i = gi & 7;
  int l0, h0, l1, h1, l2, h2;
  axis_taps((g >> 2) & 1, (i >> 2) & 1, l0, h0);
  axis_taps((g >> 1) & 1, (i >> 1) & 1, l1, h1);
  axis_taps(g & 1, i & 1, l2, h2);
  float acc = 0.f;
  for (int dz = l0; dz <= h0; ++dz)
    for (int dy = l1; dy <= h1; ++dy)
      for (int dx = l2; dx <= h2; ++dx) acc += w[(size_t)((dz + 1) * 9 + (dy + 1) * 3 + (dx + 1)) * cc + e];
  wc[t] = acc;
}

// dW[tap] = sum over the 8 parities of dWc[g*8 + i(g, tap)]   (27 x cin*cout)
__global__ __launch_bounds__(256) void k_expand_weights_bwd(const float *__restrict__ dwc, int cc,
                                                           float *__restrict__ dw) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= 27 * cc) return;
  const int tap = t / cc, e = t - tap * cc;
  const int d0 = tap / 9 - 1, d1 = (tap / 3) % 3 - 1, d2 = tap % 3 - 1;
  float acc = 0.f;
  for (int g = 0; g < 8; ++g) {
    const int j0 = (g >> 2) & 1, j1 = (g >> 1) & 1, j2 = g & 1;
    const int i0 = j0 ? (d0 > 0) : (d0 >= 0), i1 = j1 ? (d1 > 0) : (d1 >= 0), i2 = j2 ? (d2 > 0) : (d2 >= 0);
    acc += dwc[(size_t)(g * 8 + i0 * 4 + i1 * 2 + i2) * cc + e];
  }
  dw[t] = acc;
}

SGNN_EXPORT int sgnn_expand_weights(const float *w, int cin, int cout, float *wc, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(w && wc && cin >= 1 && cout >= 1);
  const int cc = cin * cout;
  SGNN_LAUNCH(k_expand_weights, dim3((64 * cc + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, cc, wc);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_expand_weights_bwd(const float *dwc, int cin, int cout, float *dw, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(dwc && dw && cin >= 1 && cout >= 1);
  const int cc = cin * cout;
  SGNN_LAUNCH(k_expand_weights_bwd, dim3((27 * cc + 255) / 256), dim3(256), 0, (hipStream_t)stream, dwc, cc, dw);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// ---------------------------------------------------------------------------
// weight gradient: dW[k][ci][co] = sum_j x[table[k][j]][ci] * dy[j][co]
// MFMA with the site index on the contraction axis (4 rows per instruction):
//   A[i=ci][kslot=q] = x[table[k][R+q]][ci],  B[kslot=q][j=co] = dy[R+q][co]
// grid = (row blocks, offset groups of DW_KPB); every wave keeps DW_KPB*MT*NT accumulators.
// Workgroup partials go to the workspace and are summed in fixed order by k_dw_reduce
// (deterministic, no float atomics).
// ---------------------------------------------------------------------------
// offsets per workgroup: bounded by accumulator registers (KPB*MT*NT*4 VGPRs) and the LDS combine buffer
template <int CIN, int COUT>
struct DwCfg {
  static constexpr int MT = (CIN + 15) / 16, NT = (COUT + 15) / 16;
  static constexpr int KPB = (MT * NT == 1) ? 9 : ((MT * NT <= 3) ? 5 : 3);
};

#ifndef DW_PAIR
#define DW_PAIR 1
#endif
template <int CIN, int COUT, bool EX, int KPBT = 0>
__global__ __launch_bounds__(256) void k_conv_dw(const float *__restrict__ x, int64_t n_in,
                                                const float *__restrict__ dy, const int32_t *__restrict__ table,
                                                int64_t ld, int K, int64_t n_out, float *__restrict__ partial,
                                                int64_t rows_per_block, int in_shift, ConvEx ex, int64_t ldx,
                                                int64_t ld_dy, const int64_t *n_dev) {
  if (n_dev) {   // capacity mode: spread the LIVE rows over all row blocks of the (capacity-sized) launch
    n_out = sgnn_dyn_n(n_out, n_dev);
    const int64_t per = (n_out + gridDim.x - 1) / gridDim.x;
    rows_per_block = ((per + 255) / 256) * 256;
    if (rows_per_block < 256) rows_per_block = 256;
  }
  constexpr int MT = (CIN + 15) / 16, NT = (COUT + 15) / 16;
  constexpr int DW_KPB = KPBT > 0 ? KPBT : DwCfg<CIN, COUT>::KPB;
  constexpr int V = (CIN + 3) / 4, CINP = 4 * V;        // x quarter-row width (as in the forward kernel)
  constexpr int W = (COUT + 3) / 4, COUTP = 4 * W;      // dy quarter-row width
  constexpr int XS = 64 * CINP, YS = 64 * COUTP;        // per-wave LDS tiles (64 rows)
  // PAIR (round 4): rows of at most 8 channels fill half of the 16 A rows of an MFMA tile — two offsets share one tile
  // (A rows 0..7 = the channels of offset 2p, rows 8..15 = those of offset 2p+1, from two LDS tiles; B = dy for both):
  // half the MFMAs and half the fragment reads for the <8,8> / <8,12> gradients.  Same products, same order of
  // additions per (offset, channel pair): bit-identical to the unpaired form.
  constexpr bool PAIR = DW_PAIR && MT * NT == 1 && CIN == 8 && KPBT == 0 && !EX;
  constexpr int XT = PAIR ? 2 : 1;                       // x tiles per wave
  constexpr int RED = DW_KPB * MT * 16 * NT * 16;
  constexpr int LDS_FLOATS = (4 * (XT * XS + YS) > RED) ? 4 * (XT * XS + YS) : RED;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i16 = lane & 15, q = lane >> 4;
  // linear workgroup id -> (row block, offset group) with all offset groups of a row block and neighbouring
  // row blocks on the same XCD (they gather the same feature rows)
  const unsigned lin = sgnn_xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const unsigned bx = lin / gridDim.y, by = lin % gridDim.y;
  const unsigned groups = EX ? (unsigned)ex.groups : 1u;
  const unsigned kgroups = gridDim.y / groups;       // offset groups per weight group
  const unsigned grp = by / kgroups;
  const int k0 = (by % kgroups) * DW_KPB;
  const int32_t *kmap = (EX && ex.kmap) ? ex.kmap + grp * K : nullptr;
  const int table_rows = EX ? ex.table_rows : K;
  const int kc = (K - k0) < DW_KPB ? (K - k0) : DW_KPB;
  const int64_t blk_row0 = (int64_t)bx * rows_per_block;
  int64_t blk_row1 = blk_row0 + rows_per_block;
  if (blk_row1 > n_out) blk_row1 = n_out;

  const uint32_t ldx4 = (uint32_t)ldx * 4u;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (uint32_t)(((n_in - 1) * ldx + CIN) * 4));
  const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(table, (uint32_t)((int64_t)table_rows * ld * 4));
  const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(dy, (uint32_t)(((n_out * groups - 1) * ld_dy + COUT) * 4));
  const uint32_t ld4 = (uint32_t)ld * 4u;
  float *xs = lds + wave * (XT * XS + YS);   // [64][CINP]  gathered feature rows of the current offset (PAIR: of two offsets)
  float *ys = xs + XT * XS;                  // [64][COUTP] output-gradient rows of the chunk

  constexpr int NP = (DW_KPB + 1) / 2;                   // PAIR: offset pairs (the last one may be half empty)
  f32x4 accp[PAIR ? NP : 1];
#pragma unroll
  for (int p = 0; p < (PAIR ? NP : 1); ++p) accp[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 acc[DW_KPB][MT][NT];
#pragma unroll
  for (int kk = 0; kk < DW_KPB; ++kk)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[kk][mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // wide (16 B / lane) gathers like the forward kernel, transposed through LDS so that the site index lands on
  // the MFMA contraction axis: A[i = ci][kslot = j] = x[table[k][R + j]][ci], B[kslot = j][co] = dy[R + j][co]
  // Row-contiguous gathers (round 4, scripts/kernels/gather_bench.hip): a gather instruction whose 16-lane quarter-waves
  // each touch 16 different rows (the MFMA-fragment mapping: lane (row = lane & 15, quarter = lane >> 4)) costs the texture
  // path 41 cycles per KiB; with lanes 4g .. 4g+3 covering one whole 64-byte row it costs 28 (32-byte rows: 2 lanes per
  // row, 17 instead of 25 us per pass over the 366 k-row level).  The weight gradient stages its rows through LDS anyway
  // (the site index must land on the MFMA contraction axis), so for 8- and 16-channel rows the gathers use that mapping:
  // instruction j covers rows j*RPI + lane / LPR, lane % LPR is the 16-byte chunk; the LDS image is the same.
  constexpr bool RCX = (CIN == 16 || CIN == 8);
  constexpr int LPR = RCX ? CIN / 4 : 1, RPI = 64 / LPR, NI = RCX ? LPR : 4, GW = RCX ? 4 : V;   // NI loads of GW floats
  const int rc_row = lane / LPR, rc_chunk = lane % LPR;
  auto gather = [&](int32_t iv, float(&g)[NI][GW]) {
#pragma unroll
    for (int m = 0; m < NI; ++m) {
      const int32_t id = __builtin_amdgcn_ds_bpermute((RCX ? m * RPI + rc_row : m * 16 + i16) * 4, iv);
      buf_load_floats<GW>(rs_x, (uint32_t)id * ldx4 + (uint32_t)(RCX ? rc_chunk * 16 : q * V * 4), g[m]);
    }
  };
  auto norm_rows = [&](float(&g)[NI][GW]) {   // the last quarter reads past the row end: those slots must be exact zeros
    if constexpr (CINP != CIN) {
#pragma unroll
      for (int m = 0; m < NI; ++m)
#pragma unroll
        for (int s = 0; s < GW; ++s)
          if (3 * V + s >= CIN) g[m][s] = (q == 3) ? 0.f : g[m][s];
    }
  };
  auto store_rows = [&](const float(&g)[NI][GW]) {   // the wave's 64 gathered rows -> xs[row][CINP]
#pragma unroll
    for (int m = 0; m < NI; ++m) {
      float *p = RCX ? xs + (m * RPI + rc_row) * CINP + rc_chunk * 4 : xs + (m * 16 + i16) * CINP + q * V;
#pragma unroll
      for (int s = 0; s < GW; ++s) p[s] = g[m][s];
    }
  };
  auto store_rows_to = [&](const float(&g)[NI][GW], float *tile) {   // RCX shapes only (PAIR)
#pragma unroll
    for (int m = 0; m < NI; ++m) {
      float *p = tile + (m * RPI + rc_row) * CINP + rc_chunk * 4;
#pragma unroll
      for (int s = 0; s < GW; ++s) p[s] = g[m][s];
    }
  };
  auto mma_pair = [&](int p, const float(&b)[16][NT]) {   // lanes i16 < 8: tile 0 (offset 2p), i16 >= 8: tile 1 (offset 2p+1)
    const float *src = xs + (i16 >> 3) * XS + (i16 & 7);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float a = ((i16 & 7) < CINP) ? src[(4 * t + q) * CINP] : 0.f;
      accp[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[t][0], accp[p], 0, 0, 0);
    }
  };
  auto mma_chunk = [&](int kk, const float(&b)[16][NT]) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      float a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int ch = mt * 16 + i16;
        a[mt] = (ch < CINP) ? xs[(4 * t + q) * CINP + ch] : 0.f;
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[kk][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[t][nt], acc[kk][mt][nt], 0, 0, 0);
    }
  };

  for (int64_t base = blk_row0 + wave * 64; base < blk_row1; base += 256) {
    // rule entries of this wave's 64 rows for every offset of the group (table padding rows hold -1)
    // (branch-free: offsets past the group's end load a clamped row and are then forced to -1 = "no rule", so every
    // load below is unconditional and the compiler can use counted s_waitcnt instead of draining at merge points)
    int32_t idxv[DW_KPB];
#pragma unroll
    for (int kk = 0; kk < DW_KPB; ++kk) {
      const int ko = k0 + (kk < kc ? kk : kc - 1);
      int32_t id;
      if constexpr (EX) {
        const int trow = kmap ? kmap[ko] : ko;
        id = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, (uint32_t)(base + lane) * 4u, trow * ld4, 0);
        id = (id >> in_shift) * ex.in_mul + ((ex.kadd && id >= 0) ? ex.kadd[grp * K + ko] : 0);
      } else {
        id = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, (uint32_t)(base + lane) * 4u, ko * ld4, 0) >> in_shift;
      }
      idxv[kk] = (kk < kc) ? id : -1;
    }
    // dy tile -> LDS -> B fragments kept in registers for all offsets (rows >= n_out read as zeros)
    if constexpr (COUT == 16 || COUT == 8) {   // row-contiguous (see RCX above): COUT / 4 lanes per row, 16-byte chunks
      constexpr int LY = COUT / 4, RY = 64 / LY;
      const int yr = lane / LY, yc = lane % LY;
      float g[LY][4];
#pragma unroll
      for (int m = 0; m < LY; ++m) {
        const int64_t row = base + m * RY + yr;
        const uint32_t off = (row < blk_row1) ? (uint32_t)((row * groups + grp) * ld_dy + yc * 4) * 4u : 0xFFFFF800u;
        buf_load_floats<4>(rs_dy, off, g[m]);
      }
#pragma unroll
      for (int m = 0; m < LY; ++m) {
        float *p = ys + (m * RY + yr) * COUTP + yc * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) p[s] = g[m][s];
      }
    } else {
      float g[4][W];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int64_t row = base + m * 16 + i16;
        const uint32_t off =
            (row < blk_row1) ? (uint32_t)((row * groups + grp) * ld_dy + q * W) * 4u : 0xFFFFF800u;
        // (not 0xFFFFFFFF: buf_load_floats issues its tail loads at off + 16, +32: a 32-bit wrap would land them INSIDE the
        //  buffer at a misaligned address — garbage that is multiplied by the zero x row of a missing rule, i.e. harmless
        //  unless it happens to be NaN / Inf; slabs are limited to 0xFFFFF000 bytes, so this offset is out of range)
        buf_load_floats<W>(rs_dy, off, g[m]);
        if constexpr (COUTP != COUT) {
#pragma unroll
          for (int s = 0; s < W; ++s)
            if (3 * W + s >= COUT) g[m][s] = (q == 3) ? 0.f : g[m][s];
        }
      }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float *p = ys + (m * 16 + i16) * COUTP + q * W;
#pragma unroll
        for (int s = 0; s < W; ++s) p[s] = g[m][s];
      }
    }
    float b[16][NT];
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int co = nt * 16 + i16;
        b[t][nt] = (co < COUTP) ? ys[(4 * t + q) * COUTP + co] : 0.f;
      }

    if constexpr (PAIR) {
      static_assert(RCX, "the paired path stores row-contiguous gathers");
      // two offsets per MFMA tile; the rows of the next pair are gathered before the MFMA block of the current one
      float ga[2][NI][GW], gb[2][NI][GW];
      auto idx_of = [&](int kk) { return kk < DW_KPB ? idxv[kk < DW_KPB ? kk : 0] : -1; };   // past the group: no rule
      gather(idx_of(0), ga[0]);
      gather(idx_of(1), gb[0]);
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (p + 1 < NP) {
          gather(idx_of(2 * p + 2), ga[(p + 1) & 1]);
          gather(idx_of(2 * p + 3), gb[(p + 1) & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        store_rows_to(ga[p & 1], xs);
        store_rows_to(gb[p & 1], xs + XS);
        mma_pair(p, b);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (MT * NT == 1) {
      // narrow layers: little MFMA work per gather -> prefetch the next offset's rows (ping-pong registers)
      float g0[NI][GW], g1[NI][GW];
      gather(idxv[0], g0);
#pragma unroll
      for (int kk = 0; kk < DW_KPB; kk += 2) {
        if (kk + 1 < DW_KPB) gather(idxv[kk + 1], g1);               // compile-time conditions only
        __builtin_amdgcn_sched_barrier(0);
        norm_rows(g0);
        store_rows(g0);
        mma_chunk(kk, b);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 1 < DW_KPB) {
          if (kk + 2 < DW_KPB) gather(idxv[kk + 2], g0);
          __builtin_amdgcn_sched_barrier(0);
          norm_rows(g1);
          store_rows(g1);
          mma_chunk(kk + 1, b);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      // wide layers: 3+ MFMA tiles per gathered row hide the latency; one register set keeps occupancy up
      float g0[NI][GW];
#pragma unroll
      for (int kk = 0; kk < DW_KPB; ++kk) {
        gather(idxv[kk], g0);
        norm_rows(g0);
        store_rows(g0);
        mma_chunk(kk, b);
      }
    }
  }

  // combine the four waves in fixed order through LDS (the per-wave tiles are dead by now)
  __syncthreads();
  float *red = lds;
  for (int wv = 0; wv < 4; ++wv) {
    if constexpr (PAIR) {
      if (wave == wv) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = q * 4 + i, kk = 2 * p + (row >> 3), ci = row & 7, co = i16;   // A row -> (offset, channel)
            if (kk < DW_KPB) {
              float *pr = &red[(kk * MT * 16 + ci) * NT * 16 + co];
              if (wv == 0)
                *pr = accp[p][i];
              else
                *pr += accp[p][i];
            }
          }
      }
    } else if (wave == wv) {
#pragma unroll
      for (int kk = 0; kk < DW_KPB; ++kk)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int ci = mt * 16 + q * 4 + i, co = nt * 16 + i16;
              float *p = &red[(kk * MT * 16 + ci) * NT * 16 + co];
              if (wv == 0)
                *p = acc[kk][mt][nt][i];
              else
                *p += acc[kk][mt][nt][i];
            }
    }
    __syncthreads();
  }
  float *out = partial + (((int64_t)bx * groups + grp) * K + k0) * CIN * COUT;
  for (int e = tid; e < kc * CIN * COUT; e += 256) {
    const int co = e % COUT, ci = (e / COUT) % CIN, kk = e / (CIN * COUT);
    out[e] = red[(kk * MT * 16 + ci) * NT * 16 + co];
  }
}

// Weight gradient of a ONE-channel input (the network's first convolution, 1 -> 8: `model.py:178`), round 4.  The MFMA kernel
// above spends a 16 x 16 x 4 tile per four rows on a 1 x 8 outer product and pushes the single input channel through the same
// LDS transposition as a 16-channel row: 75 us on the 380 k-row input level, as long as the <16,16> gradient.  Here a thread
// owns rows (row block base + tid, + 256, ...) and keeps the KP x COUT sums of its offset group in registers:
// dw[k][0][co] += x[table[k][j]] * dy[j][co] — coalesced rule entries, 4-byte gathers that mostly hit neighbouring rows, two
// 16-byte dy loads per row.  Same workspace layout and reduce kernel as k_conv_dw; the workgroup's sum is formed in a fixed
// order (waves 0..3, then lanes 0..63): deterministic for a given launch.
template <int COUT, int KP>
__global__ __launch_bounds__(256) void k_conv_dw_c1(const float *__restrict__ x, int64_t n_in, const float *__restrict__ dy,
                                                   const int32_t *__restrict__ table, int64_t ld, int K, int64_t n_out,
                                                   float *__restrict__ partial, int64_t rows_per_block, int in_shift,
                                                   int64_t ldx, int64_t ld_dy, const int64_t *n_dev) {
  static_assert(COUT % 4 == 0, "dy rows are loaded 16 bytes at a time");
  if (n_dev) {   // capacity mode: spread the LIVE rows over all row blocks of the (capacity-sized) launch (as k_conv_dw)
    n_out = sgnn_dyn_n(n_out, n_dev);
    const int64_t per = (n_out + gridDim.x - 1) / gridDim.x;
    rows_per_block = ((per + 255) / 256) * 256;
    if (rows_per_block < 256) rows_per_block = 256;
  }
  constexpr int NV = KP * COUT;
  __shared__ float red[NV][65];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned lin = sgnn_xcd_tile(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
  const unsigned bx = lin / gridDim.y, by = lin % gridDim.y;
  const int k0 = by * KP;
  const int kc = (K - k0) < KP ? (K - k0) : KP;
  const int64_t blk_row0 = (int64_t)bx * rows_per_block;
  int64_t blk_row1 = blk_row0 + rows_per_block;
  if (blk_row1 > n_out) blk_row1 = n_out;
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(x, (uint32_t)(((n_in - 1) * ldx + 1) * 4));
  const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(table, (uint32_t)((int64_t)K * ld * 4));
  const __amdgpu_buffer_rsrc_t rs_dy = make_rsrc(dy, (uint32_t)(((n_out - 1) * ld_dy + COUT) * 4));
  const uint32_t ld4 = (uint32_t)ld * 4u, ldx4 = (uint32_t)ldx * 4u;
  float acc[KP][COUT];
#pragma unroll
  for (int kk = 0; kk < KP; ++kk)
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[kk][c] = 0.f;
  for (int64_t row = blk_row0 + tid; row < blk_row1; row += 256) {
    int32_t id[KP];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {   // offsets past the group's end: a clamped entry, dropped below
      const int ko = k0 + (kk < kc ? kk : kc - 1);
      id[kk] = (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, (uint32_t)row * 4u, ko * ld4, 0) >> in_shift;
    }
    float g[COUT];
    buf_load_floats<COUT>(rs_dy, (uint32_t)(row * ld_dy) * 4u, g);
    float xv[KP];
#pragma unroll
    for (int kk = 0; kk < KP; ++kk)     // rule -1 -> offset 0xFFFFFFFC * ldx -> out of range -> 0
      xv[kk] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, (uint32_t)id[kk] * ldx4, 0, 0));
#pragma unroll
    for (int kk = 0; kk < KP; ++kk) {
      const float v = (kk < kc) ? xv[kk] : 0.f;
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc[kk][c] = fmaf(v, g[c], acc[kk][c]);
    }
  }
  // the four waves in fixed order into red[value][lane], then the 64 lanes of every value in order
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int kk = 0; kk < KP; ++kk)
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
          float *p = &red[kk * COUT + c][lane];
          if (wv == 0)
            *p = acc[kk][c];
          else
            *p += acc[kk][c];
        }
    }
    __syncthreads();
  }
  float *out = partial + ((int64_t)bx * K + k0) * COUT;
  if (tid < kc * COUT) {
    float t = 0.f;
    for (int l = 0; l < 64; ++l) t += red[tid][l];
    out[tid] = t;
  }
}

// dw[e] = sum_b partial[b][e]: 32 consecutive elements per workgroup x 8 interleaved partial streams,
// combined in fixed order through LDS (deterministic)
__global__ __launch_bounds__(256) void k_dw_reduce(const float *__restrict__ partial, int64_t nblk,
                                                  int64_t elems, float *__restrict__ dw) {
  __shared__ float red[8][32];
  const int part = threadIdx.x >> 5, le = threadIdx.x & 31;
  const int64_t e = (int64_t)blockIdx.x * 32 + le;
  float s = 0.f;
  if (e < elems)
    for (int64_t b = part; b < nblk; b += 8) s += partial[b * elems + e];
  red[part][le] = s;
  __syncthreads();
  if (part == 0 && e < elems) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) t += red[p][le];
    dw[e] = t;
  }
}

// The reduce launches of many weight gradients as ONE launch (prog.hip: every convolution of a program's backward pass
// keeps its partials in its own workspace slice; ~50 launches per training step become 5)
__global__ __launch_bounds__(256) void k_dw_reduce_batch(DwBatch b) {
  __shared__ float red[8][32];
  int i = 0;
  while (i + 1 < b.n && (int)blockIdx.x >= b.d[i + 1].blk0) ++i;
  const DwDesc d = b.d[i];
  const int part = threadIdx.x >> 5, le = threadIdx.x & 31;
  const int64_t e = (int64_t)((int)blockIdx.x - d.blk0) * 32 + le;
  float s = 0.f;
  if (e < d.elems)
    for (int64_t k = part; k < d.nblk; k += 8) s += d.partial[k * d.elems + e];
  red[part][le] = s;
  __syncthreads();
  if (part == 0 && e < d.elems) {
    float t = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) t += red[p][le];
    d.dw[e] = t;
  }
}

DwBatch *sgnn_dw_batch = nullptr;   // non-NULL: sgnn_conv_bwd_weight* defer their reduce into it (set by prog.hip only)

int sgnn_dw_batch_flush(DwBatch *b, hipStream_t s) {
  if (!b || b->n == 0) return SGNN_OK;
  int blocks = 0;
  for (int i = 0; i < b->n; ++i) {
    b->d[i].blk0 = blocks;
    blocks += (int)((b->d[i].elems + 31) / 32);
  }
  SGNN_LAUNCH(k_dw_reduce_batch, dim3((unsigned)blocks), dim3(256), 0, s, *b);
  b->n = 0;
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// conv_bwd_fused.hip: the reduce of a fused backward launch's workgroup partials — deferred into the program's batch like
// every other weight gradient of the call, or launched at once
int sgnn_dw_reduce_or_defer(const float *partial, float *dw, int64_t nblk, int64_t elems, hipStream_t s) {
  if (sgnn_dw_batch && sgnn_dw_batch->n < DW_BATCH_MAX) {
    sgnn_dw_batch->d[sgnn_dw_batch->n++] = DwDesc{partial, dw, nblk, elems, 0};
    return SGNN_OK;
  }
  SGNN_LAUNCH(k_dw_reduce, dim3((unsigned)((elems + 31) / 32)), dim3(256), 0, s, partial, nblk, elems, dw);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// generic fallback: one workgroup per (k, ci, co) triple would be wasteful; instead one thread per
// weight element loops over all rows (slow, correctness only)
__global__ __launch_bounds__(256) void k_conv_dw_generic(const float *__restrict__ x, int cin,
                                                        const float *__restrict__ dy, int cout,
                                                        const int32_t *__restrict__ table, int64_t ld, int K,
                                                        int64_t n_out, float *__restrict__ dw, int in_shift,
                                                        ConvEx ex, const int64_t *n_dev) {
  n_out = sgnn_dyn_n(n_out, n_dev);
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (int64_t)ex.groups * K * cin * cout) return;
  const int co = (int)(e % cout), ci = (int)((e / cout) % cin), k = (int)((e / ((int64_t)cin * cout)) % K);
  const int grp = (int)(e / ((int64_t)K * cin * cout));
  float s = 0.f;
  for (int64_t j = 0; j < n_out; ++j) {
    const int32_t id = table[(int64_t)(ex.kmap ? ex.kmap[grp * K + k] : k) * ld + j];
    if (id >= 0) {
      const float xv = x[((int64_t)(id >> in_shift) * ex.in_mul + (ex.kadd ? ex.kadd[grp * K + k] : 0)) * cin + ci];
      s = fmaf(xv, dy[(j * ex.groups + grp) * cout + co], s);
    }
  }
  dw[e] = s;
}

#define DW_FINE_ROWS 16384   // below: one offset per weight-gradient workgroup
static int64_t dw_rows_per_block(int64_t n_out) {
  int64_t rpb = (n_out + g_tune.conv_dw_blocks - 1) / g_tune.conv_dw_blocks;   // <= g_tune.conv_dw_blocks row blocks
  rpb = ((rpb + 255) / 256) * 256;             // whole 256-row wave rounds
  if (rpb < 256) rpb = 256;
  return rpb;
}

bool dw_shape_ok(int cin, int cout) {   // compiled weight-gradient shapes (CONV_DW_CASES)
  static const int shapes[][2] = {{1, 8}, {8, 8}, {8, 12}, {12, 12}, {12, 16}, {16, 16}, {34, 16}, {30, 16}, {26, 16}, {48, 16},
                                  {32, 16}, {4, 16}, {16, 24}, {24, 32}, {64, 32}, {56, 28}, {32, 32}, {28, 16}};
  for (const auto &sh : shapes)
    if (sh[0] == cin && sh[1] == cout) return true;
  return false;
}

SGNN_EXPORT int64_t sgnn_conv_bwd_weight_ws_bytes(int64_t n_out, int K, int cin, int cout) {
  if (n_out <= 0) return 0;
  const int64_t rpb = dw_rows_per_block(n_out);
  const int64_t nblk = (n_out + rpb - 1) / rpb;
  return nblk * (int64_t)K * cin * cout * (int64_t)sizeof(float);   // K counts every group's offsets (groups*K)
}

#define CONV_DW_CASES(X) \
  X(1, 8) X(8, 8) X(8, 12) X(12, 12) X(12, 16) X(16, 16) X(34, 16) X(30, 16) X(26, 16) X(48, 16) X(32, 16) X(4, 16) \
  X(16, 24) X(24, 32) X(64, 32) X(56, 28) X(32, 32) X(28, 16)

SGNN_EXPORT int sgnn_conv_bwd_weight_ex(const float *x, int64_t n_in, int cin, const float *dy, int cout,
                                        const int32_t *table, int64_t ld, int K, int64_t n_out, float *dw,
                                        int in_shift, const int32_t *kmap, const int32_t *kadd, int in_mul,
                                        int groups, int table_rows, void *ws, int64_t ws_bytes,
                                        sgnn_stream_t stream) {
  return sgnn_conv_bwd_weight_impl(x, n_in, cin, cin, dy, cout, cout, table, ld, K, n_out, dw, in_shift, kmap, kadd,
                                   in_mul, groups, table_rows, ws, ws_bytes, stream);
}

// ldx / ld_dy: row strides (floats) of x and dy (rows inside a wider buffer); the generic fallback needs contiguous rows
int sgnn_conv_bwd_weight_impl(const float *x, int64_t n_in, int cin, int64_t ldx, const float *dy, int cout, int64_t ld_dy,
                              const int32_t *table, int64_t ld, int K, int64_t n_out, float *dw, int in_shift,
                              const int32_t *kmap, const int32_t *kadd, int in_mul, int groups, int table_rows, void *ws,
                              int64_t ws_bytes, sgnn_stream_t stream, const int64_t *n_dev) {
  SGNN_CHECK_ARG(ldx >= cin && ldx <= 1024 && ld_dy >= cout && ld_dy <= 1024);
  SGNN_CHECK_ARG(cin >= 1 && cout >= 1 && K >= 1 && K <= 64 && n_out >= 0 && ld >= n_out && dw &&
                 in_shift >= 0 && in_shift < 31 && in_mul >= 1 && groups >= 1 && groups <= 64 && table_rows >= 1 &&
                 table_rows <= 64 && (kmap || table_rows >= K));
  hipStream_t s = (hipStream_t)stream;
  const int64_t elems = (int64_t)groups * K * cin * cout;
  if (n_out == 0) {
    if (sgnn_fill32(dw, 0u, elems, s) != SGNN_OK) return SGNN_EHIP;
    return SGNN_OK;
  }
  SGNN_CHECK_ARG(x && dy && table && n_in >= 1);
  SGNN_CHECK_ARG(ld % CONV_ROWS_PER_BLOCK == 0);
  if (n_in * ldx * 4 > 0xFFFFF000ll || n_out * groups * ld_dy * 4 > 0xFFFFF000ll ||
      (int64_t)table_rows * ld * 4 > 0xFFFFF000ll) {
    sgnn_set_error("sgnn_conv_bwd_weight: a slab exceeds the 4 GiB raw-buffer window");
    return SGNN_EOVERFLOW;
  }
  const ConvEx ex{kmap, kadd, in_mul, groups, table_rows};
  bool done = false;
  const int64_t rpb = dw_rows_per_block(n_out);
  const int64_t nblk = (n_out + rpb - 1) / rpb;
  const bool plain = !kmap && !kadd && in_mul == 1 && groups == 1 && table_rows == K;
#define LAUNCH_DW(CI, CO, EXV)                                                                             \
  do {                                                                                                     \
    if (!ws || ws_bytes < sgnn_conv_bwd_weight_ws_bytes(n_out, groups * K, cin, cout)) {                   \
      sgnn_set_error("sgnn_conv_bwd_weight: workspace too small");                                         \
      return SGNN_ENOWS;                                                                                   \
    }                                                                                                      \
    constexpr int kpb_ = DwCfg<CI, CO>::KPB;                                                               \
    const int prof = sgnn_prof_begin_launch(1, n_out * groups, cin, cout, K, 0, s);                        \
    if constexpr (!EXV && kpb_ == 9) {                                                                     \
      /* small level of a narrow layer: 256 rows x 9 offsets per workgroup leaves most CUs idle and makes */ \
      /* every wave walk 9 dependent gather rounds -> one offset per workgroup (same sums, same order)     */ \
      if (n_out < DW_FINE_ROWS && g_tune.conv_small)                                                          \
        SGNN_LAUNCH((k_conv_dw<CI, CO, false, 1>), dim3((unsigned)nblk, (unsigned)K), dim3(256), 0, s, \
                           x, n_in, dy, table, ld, K, n_out, (float *)ws, rpb, in_shift, ex, ldx, ld_dy, n_dev); \
      else                                                                                                 \
        SGNN_LAUNCH((k_conv_dw<CI, CO, false, 0>), dim3((unsigned)nblk, (unsigned)((K + kpb_ - 1) / kpb_)), \
                           dim3(256), 0, s, x, n_in, dy, table, ld, K, n_out, (float *)ws, rpb, in_shift,  \
                           ex, ldx, ld_dy, n_dev);                                                         \
    } else {                                                                                               \
      SGNN_LAUNCH((k_conv_dw<CI, CO, EXV, 0>),                                                \
                         dim3((unsigned)nblk, (unsigned)(groups * ((K + kpb_ - 1) / kpb_))), dim3(256), 0, \
                         s, x, n_in, dy, table, ld, K, n_out, (float *)ws, rpb, in_shift, ex, ldx, ld_dy, n_dev); \
    }                                                                                                      \
    sgnn_prof_end_launch(prof, s);                                                                         \
    if (sgnn_dw_batch && sgnn_dw_batch->n < DW_BATCH_MAX) {                                                \
      sgnn_dw_batch->d[sgnn_dw_batch->n++] = DwDesc{(const float *)ws, dw, nblk, elems, 0};                \
    } else {                                                                                               \
      SGNN_LAUNCH(k_dw_reduce, dim3((unsigned)((elems + 31) / 32)), dim3(256), 0, s,                \
                         (const float *)ws, nblk, elems, dw);                                              \
    }                                                                                                      \
    done = true;                                                                                           \
  } while (0)
  if (plain && cin == 1 && cout == 8 && g_tune.conv_dw_c1) {   // one-channel input: the VALU kernel (k_conv_dw_c1)
    if (!ws || ws_bytes < sgnn_conv_bwd_weight_ws_bytes(n_out, K, cin, cout)) {
      sgnn_set_error("sgnn_conv_bwd_weight: workspace too small");
      return SGNN_ENOWS;
    }
    const int prof = sgnn_prof_begin_launch(1, n_out, cin, cout, K, 0, s);
    SGNN_LAUNCH((k_conv_dw_c1<8, 9>), dim3((unsigned)nblk, (unsigned)((K + 8) / 9)), dim3(256), 0, s, x, n_in, dy, table, ld,
                K, n_out, (float *)ws, rpb, in_shift, ldx, ld_dy, n_dev);
    sgnn_prof_end_launch(prof, s);
    if (sgnn_dw_batch && sgnn_dw_batch->n < DW_BATCH_MAX)
      sgnn_dw_batch->d[sgnn_dw_batch->n++] = DwDesc{(const float *)ws, dw, nblk, elems, 0};
    else
      SGNN_LAUNCH(k_dw_reduce, dim3((unsigned)((elems + 31) / 32)), dim3(256), 0, s, (const float *)ws, nblk, elems, dw);
    done = true;
  }
#define X(CI, CO) \
  if (!done && plain && cin == CI && cout == CO) LAUNCH_DW(CI, CO, false);
  CONV_DW_CASES(X)
#undef X
#define X(CI, CO) \
  if (!done && !plain && cin == CI && cout == CO) LAUNCH_DW(CI, CO, true);
  X(48, 16) X(24, 8) X(64, 32) X(56, 28)   // up-sampling convolution; dense ConvTranspose3d(k4,s2) by parity groups (model.DenseK4S2)
#undef X
  if (!done) {
    if (ldx != cin || ld_dy != cout) {
      sgnn_set_error("sgnn_conv_bwd_weight: strided rows need one of the compiled (cin, cout) shapes, got (%d, %d)", cin, cout);
      return SGNN_EINVAL;
    }
    SGNN_LAUNCH(k_conv_dw_generic, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s, x, cin, dy,
                       cout, table, ld, K, n_out, dw, in_shift, ex, n_dev);
  }
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_conv_bwd_weight(const float *x, int64_t n_in, int cin, const float *dy, int cout,
                                     const int32_t *table, int64_t ld, int K, int64_t n_out, float *dw,
                                     int in_shift, void *ws, int64_t ws_bytes, sgnn_stream_t stream) {
  return sgnn_conv_bwd_weight_ex(x, n_in, cin, dy, cout, table, ld, K, n_out, dw, in_shift, nullptr, nullptr, 1, 1,
                                 K, ws, ws_bytes, stream);
}
