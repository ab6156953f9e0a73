// Backward of a 3x3x3 SubmanifoldConvolution with square 16-channel rows as ONE kernel: data gradient AND weight gradient
// from a single gather of dy (round 6).  Reference call sites: the FullyConvolutionalNet bodies and the
// residual blocks of torch/model.py:38,40,180,255, whose backward pass is train.py:262.
//
//   dx[i]  = sum_k dy[nbr(i, k)] . W[26 - k]^T                 (the walk k_conv_fwd_w already does for the data gradient)
//   dW[26 - k] = sum_i x[i]^T . dy[nbr(i, k)]                  (the SAME gathered rows; x[i] is the wave's own row tile)
//
// Until round 5 the weight gradient was a second kernel family (k_conv_dw) that gathered x through the forward table on a
// side stream and contended with the chain's convolutions for the texture path.  Here the dX walk keeps its gathers, its
// MFMAs and its wide epilogue (rows and BatchNorm-backward statistics partials are produced by the same code as in
// k_conv_fwd_w: bit-identical rows) and adds, per offset and 64-row wave tile:
//   * one transposition of the gathered 64 x 16 tile through a wave-private LDS buffer: the dX product wants the row index
//     on the MFMA's free axis (lane & 15), the dW product wants it on the contraction axis (lane >> 4, register).  16
//     ds_write_b32 (lanes consecutive: conflict-free) + 4 ds_read_b128 (images skewed by {0, 8, 32, 40} floats: conflict-free
//     for the instruction's 16-lane groups), ~80 LDS cycles per wave-offset against 256 CU-cycles of MFMA;
//   * 16 more v_mfma_f32_16x16x4_f32 with A = x^T (transposed ONCE per tile, kept in registers for all 27 offsets) and
//     B = the transposed gathered tile, accumulated per offset in 27 x 4 accumulator registers that live across all row tiles
//     of the workgroup (straight-line walk: every accumulator index is a compile-time constant);
//   * at the end the four waves' accumulators are summed through LDS in fixed order and leave as ONE partial per workgroup;
//     the fixed-order reduce over workgroups is the existing k_dw_reduce(_batch).  No atomics: deterministic.
// The MFMA pipe becomes the binding unit (32 instead of 16 instructions per wave-offset) where the texture path was; two
// waves per SIMD (108 + 16 accumulators).
//
// Measured (profiles/r06_fused_backward.txt), 366 k rows: 131 us against 62 (dX with its backward epilogue) + 77 (dW + reduce)
// = 0.95x the two kernels, dX rows bit-identical, dW to fp32 summation order.  In the step: 5.78 -> 5.93 ms with the six
// launches of >= 40 960 rows fused, 5.83 with only the two of the 450 k-row level — the summed convolution time falls
// (5.13 -> 4.73 ms) but the launch puts twice the MFMA work on the training stream and takes work off a lane that was hiding
// it.  sgnn_prog_backward therefore uses it only on request (sgnn_tune.conv_bwd_fused = 1).
#include "common.h"

#include "conv_common.h"

#define FUSED_K 27
#define FUSED_TSTRIDE 296                    // floats per 16-row tile image in the transposition buffer
#define FUSED_TB (4 * FUSED_TSTRIDE)         // per wave


// float offset of channel-phase j's 64-float image inside a tile image (skew: see the header comment)
__device__ __forceinline__ constexpr int fused_img(int j) { return j * 64 + (j & 1) * 8 + (j >> 1) * 32; }

template <int C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_conv_bwd_fused(
    const float *__restrict__ dy, int64_t n_dy, const float *__restrict__ w, const int32_t *__restrict__ table, int64_t ld,
    int64_t n_out, float *dx, ConvEpi epi, int wg_cap, const float *__restrict__ xin, int64_t ldxin,
    float *__restrict__ dwp, int pblocks) {
  static_assert(C == 16, "square 16-channel rows");
  constexpr int K = FUSED_K, M = 4, V = 4, RPW = 64;
  __shared__ __attribute__((aligned(16))) float wl[K * 256];            // W[k]^T tiles; the dW combine buffer at the end
  __shared__ double sred[4 * 2 * 16];
  __shared__ float ecst[64];
  __shared__ __attribute__((aligned(16))) float tbuf[4 * FUSED_TB];     // wave-private transposition buffers (gathered dy tile)
  __shared__ __attribute__((aligned(16))) float xbuf[4 * FUSED_TB];     // ... and the wave's own x tile, kept for all 27 offsets

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  if (epi.n_dev) n_out = sgnn_dyn_n(n_out, epi.n_dev);
  // one round of workgroups, as k_conv_fwd: every live workgroup takes J consecutive 256-row tiles
  int J = 1;
  if (wg_cap > 0) {
    const int64_t w1 = (n_out + 4 * RPW - 1) / (4 * RPW);
    J = (int)((w1 + wg_cap - 1) / wg_cap);
    if (J < 1) J = 1;
  }
  const int64_t rows_wg = (int64_t)4 * RPW * J;
  const unsigned nwg = (unsigned)((n_out + rows_wg - 1) / rows_wg);
  float *dw_out = dwp + (size_t)blockIdx.x * (K * C * C);
  if (blockIdx.x >= nwg) {   // nothing to compute: zero statistics partials, zero weight-gradient partial
    if (epi.stats)
      for (int o = tid; o < 2 * C; o += 256) epi.partial[(size_t)blockIdx.x * 2 * C + o] = 0.0;
    if ((int)blockIdx.x < pblocks)
      for (int e = tid; e < K * C * C; e += 256) dw_out[e] = 0.f;
    return;
  }
  const unsigned tile = sgnn_xcd_tile(blockIdx.x, nwg);
  int64_t row0 = ((int64_t)tile * J * 4 + wave) * RPW;

  const uint32_t ldg4 = (uint32_t)epi.ldx * 4u;
  const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(dy, (uint32_t)(((n_dy - 1) * epi.ldx + C) * 4));
  const __amdgpu_buffer_rsrc_t rs_t = make_rsrc(table, (uint32_t)((int64_t)K * ld * 4));
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(xin, (uint32_t)(((n_out - 1) * ldxin + C) * 4));
  const uint32_t ldx4 = (uint32_t)ldxin * 4u;
  uint32_t lane_off = (uint32_t)(row0 + lane) * 4u;
  const uint32_t ld4 = (uint32_t)ld * 4u;
  int perm[M];
#pragma unroll
  for (int m = 0; m < M; ++m) perm[m] = (m * 16 + r) * 4;

  float *tw = tbuf + wave * FUSED_TB;
  float *tw_wr = tw + lane;                                              // + m * TSTRIDE + img(j)
  const int rd_off = fused_img(r & 3) + 16 * (r >> 2) + 4 * q;           // + m * TSTRIDE: rows 16 m + 4 q + (0..3), channel r
  const float *tw_rd = tw + rd_off;
  float *xw_wr = xbuf + wave * FUSED_TB + lane;
  const float *xw_rd = xbuf + wave * FUSED_TB + rd_off;

  f32x4 acc[M][1];
  f32x4 accd[K];
#pragma unroll
  for (int k = 0; k < K; ++k) accd[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  double s1[1] = {0.0}, s2[1] = {0.0};
  conv_epi_wide_constants<C>(ecst, epi, epi.stats);
  conv_stage_weights<C, C>(wl, w, K, 0, K, true, true);                 // W[26 - k]^T as wl[k][n][c] (contains the barriers)

  auto load_idx = [&](int k) -> int32_t { return (int32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_t, lane_off, k * ld4, 0); };
  auto gather = [&](int32_t iv, float(&a)[M][V]) {
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int32_t id = __builtin_amdgcn_ds_bpermute(perm[m], iv);
      buf_load_floats<V>(rs_g, (uint32_t)id * ldg4 + (uint32_t)(q * V * 4), a[m]);
    }
  };
  auto load_b = [&](int kk, float(&b)[V]) {   // B fragment of the data-gradient product: W[26 - kk]^T, one ds_read_b128
    const float *bp = wl + (kk * 16 + r) * 16 + q * V;
#pragma unroll
    for (int s = 0; s < V; ++s) b[s] = bp[s];
  };
  // lanes exchange data through the wave's buffer: wave-scope fences order the compiler's view; the LDS executes one wave's
  // operations in order, so they cost no instruction
  auto wave_sync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // in: lane (r, q) holds a[m][j] = T[16 m + r][4 q + j];  out (tr_read): lane (r, q) holds t[i] = T[16 m + 4 q + i][r]
  auto tr_write = [&](float *dst, const float(&a)[M][V]) {
    wave_sync();
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int j = 0; j < V; ++j) dst[m * FUSED_TSTRIDE + fused_img(j)] = a[m][j];
    wave_sync();
  };
  auto tr_read = [&](const float *src, int m, float(&t)[4]) {
    const f32x4 v = *reinterpret_cast<const f32x4 *>(src + m * FUSED_TSTRIDE);
    t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
  };

  const bool has_add = epi.addend != nullptr, has_bnx = epi.stats == 2;
  const __amdgpu_buffer_rsrc_t rs_a = make_rsrc(has_add ? epi.addend : dy, has_add ? (uint32_t)(((n_out - 1) * epi.ld_add + C) * 4) : 0u);
  const __amdgpu_buffer_rsrc_t rs_b = make_rsrc(has_bnx ? epi.bn_x : dy, has_bnx ? (uint32_t)(((n_out - 1) * epi.ld_bnx + C) * 4) : 0u);
  const uint32_t lda4 = (uint32_t)epi.ld_add * 4u, ldb4 = (uint32_t)epi.ld_bnx * 4u;
  auto epi_prefetch = [&](EpiRows<M> &p) {     // the operand layout of conv_epi_wide_prefetch: row q*4 + (r & 3), columns 4 (r >> 2) ..
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
    // the wave's own rows of the convolution's INPUT, transposed once per tile into the wave's x buffer:
    // xa(m)[i] = x[row0 + 16 m + 4 q + i][r] is read back per offset (registers: the kernel sits at the 256-register line)
    {
      float xr[M][V];
#pragma unroll
      for (int m = 0; m < M; ++m)
        buf_load_floats<V>(rs_x, (uint32_t)(row0 + m * 16 + r) * ldx4 + (uint32_t)(q * V * 4), xr[m]);
      tr_write(xw_wr, xr);
    }
    // three register sets: rows gathered TWO offsets ahead of their MFMAs.  With two waves per SIMD the gathers in flight per
    // CU, not the texture path's rate, bound the walk (one offset ahead: 131 us at 366 k rows, profiles/r06b_fused.txt)
    float a0[M][V], a1[M][V], a2[M][V];
    gather(load_idx(0), a0);
    gather(load_idx(1), a1);
    int32_t iv2 = load_idx(2), iv3 = load_idx(3), iv4 = load_idx(4);

    // one offset: transposition + 16 dX MFMAs (acc[m], steps s in order: the summation order of k_conv_fwd) + 16 dW MFMAs on
    // accd[k], interleaved so that no two neighbouring MFMAs share an accumulator (40-cycle dependent latency, 32-cycle issue)
    // (every phase fenced for the machine scheduler: left alone it groups the dW MFMAs into dependent runs and reads the
    //  transposed tiles piecewise right in front of their use, behind an lgkmcnt(0).  The transposed 16-row tiles t[m] / xa[m]
    //  are read back two phases ahead of their MFMAs, two of each live at a time)
    auto stage = [&](int k, f32x4 &ad, float(&a)[M][V]) {
      float b[V], t[M][4], xa[M][4];
      load_b(k, b);
      tr_write(tw_wr, a);
      tr_read(tw_rd, 0, t[0]);
      tr_read(xw_rd, 0, xa[0]);
      tr_read(tw_rd, 1, t[1]);
      tr_read(xw_rd, 1, xa[1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < M; ++m) acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][0], b[0], acc[m][0], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 1; s < V; ++s) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
          ad = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s - 1][m], t[s - 1][m], ad, 0, 0, 0);
          acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b[s], acc[m][0], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (s + 1 < M) {
          tr_read(tw_rd, s + 1, t[s + 1]);
          tr_read(xw_rd, s + 1, xa[s + 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) ad = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[3][i], t[3][i], ad, 0, 0, 0);
    };

    static_assert(K % 3 == 0, "the walk rotates three register sets");
#pragma unroll
    for (int kk = 0; kk < K; kk += 3) {
      gather(iv2, a2);                                     // rows of offset kk + 2
      if (kk + 5 < K) iv2 = load_idx(kk + 5);
      __builtin_amdgcn_sched_barrier(0);
      stage(kk, accd[kk], a0);
      __builtin_amdgcn_sched_barrier(0);
      if (kk + 3 < K) {
        gather(iv3, a0);                                   // rows of offset kk + 3
        if (kk + 6 < K) iv3 = load_idx(kk + 6);
      }
      __builtin_amdgcn_sched_barrier(0);
      stage(kk + 1, accd[kk + 1], a1);
      __builtin_amdgcn_sched_barrier(0);
      if (kk + 4 < K) {
        gather(iv4, a1);                                   // rows of offset kk + 4
        if (kk + 7 < K) iv4 = load_idx(kk + 7);
      } else {
        // last offset: the epilogue's operand rows are fetched under its MFMAs, into the registers of the two row sets that
        // are dead by now.  Branch-free (an absent operand is an out-of-range load = zeros): conv_epi_wide_prefetch's uniform
        // branches made the allocator spill these rows behind a vmcnt(0) here.
        epi_prefetch(erows);
      }
      __builtin_amdgcn_sched_barrier(0);
      stage(kk + 2, accd[kk + 2], a2);
      __builtin_amdgcn_sched_barrier(0);
    }
    conv_epi_wide_finish<C, M>(acc, erows, row0, n_out, dx, epi, epi.stats, ecst, s1, s2, dy);
  }
  conv_epilogue_stats<C, 1>(s1, s2, epi, epi.stats, sred, blockIdx.x);

  // the workgroup's weight-gradient partial: the four waves in fixed order through LDS (the weight tile is dead)
  __syncthreads();
  float *red = wl;
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int k = 0; k < K; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float *p = &red[(k * 16 + q * 4 + i) * 16 + r];   // D[ci = 4 q + i][co = r]
          if (wv == 0)
            *p = accd[k][i];
          else
            *p += accd[k][i];
        }
    }
    __syncthreads();
  }
  for (int e = tid; e < K * C * C; e += 256) {
    const int k = e >> 8;
    dw_out[(K - 1 - k) * (C * C) + (e & 255)] = red[e];   // walk offset k used weight slice 26 - k
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
bool conv_wide_epi_ok(const ConvEpi &epi, const float *y, int K);                      // conv.hip
int sgnn_dw_reduce_or_defer(const float *partial, float *dw, int64_t nblk, int64_t elems, hipStream_t s);   // conv.hip


#define FUSED_MAX_BLOCKS 512         // workspace bound: partial slots a launch may need (resident workgroups of the kernel: 2 per CU)

static int fused_wg_cap() { return conv_wg_capacity<k_conv_bwd_fused<16>>(); }

// the shapes / sizes the fused kernel serves: plain 27-offset walk, 16 -> 16 channels, a level large enough for the 256-row kernel
bool sgnn_conv_bwd_fused_ok(int64_t n, int cin, int cout, int K) {
  if (!g_tune.conv_bwd_fused || !g_tune.conv_one_round || K != FUSED_K || cin != 16 || cout != 16 || n < g_tune.conv_bwd_fused_rows) return false;
  const int cap = fused_wg_cap();
  return cap > 0 && cap <= FUSED_MAX_BLOCKS;
}

// everything sgnn_conv_bwd_fused_impl checks, without raising an error (prog.hip: fall back to the two-kernel path)
bool sgnn_conv_bwd_fused_usable(int64_t n, int cin, int cout, int K, const ConvEpi &epi, const float *dx, const float *x,
                                int64_t ldx) {
  if (!sgnn_conv_bwd_fused_ok(n, cin, cout, K) || (epi.stats != 0 && epi.stats != 2)) return false;
  if ((ldx & 3) || ((uintptr_t)x & 15) || (epi.ldx & 3)) return false;
  return conv_wide_epi_ok(epi, dx, K);
}

int64_t sgnn_conv_bwd_fused_ws(int64_t n, int cin, int cout) {
  if (n <= 0) return 0;
  const int64_t grid4 = (n + CONV_ROWS_PER_BLOCK - 1) / CONV_ROWS_PER_BLOCK;
  const int64_t blocks = grid4 < FUSED_MAX_BLOCKS ? grid4 : FUSED_MAX_BLOCKS;
  return blocks * FUSED_K * (int64_t)cin * cout * (int64_t)sizeof(float);
}

// dy: (n, cout) rows with stride epi.ldx; dx: (n, cin) rows with stride epi.ldy (+ the epilogue options of ConvEpi: addend,
// BatchNorm-backward statistics, device row count); x: the convolution's input rows (n, cin), stride ldx; dw: (27, cin, cout)
int sgnn_conv_bwd_fused_impl(const float *dy, int64_t n, int cout, const float *w, const int32_t *table, int64_t ld, int cin,
                             float *dx, const ConvEpi &epi_in, const float *x, int64_t ldx, float *dw, void *ws,
                             int64_t ws_bytes, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(dy && w && table && dx && x && dw && n > 0 && ld >= n && ld % CONV_ROWS_PER_BLOCK == 0);
  SGNN_CHECK_ARG(sgnn_conv_bwd_fused_ok(n, cin, cout, FUSED_K));
  ConvEpi epi = epi_in;
  if (epi.ldx <= 0) epi.ldx = cout;
  if (epi.ldy <= 0) epi.ldy = cin;
  if (epi.ld_add <= 0) epi.ld_add = cin;
  if (epi.ld_bnx <= 0) epi.ld_bnx = cin;
  SGNN_CHECK_ARG(ldx >= cin && ldx <= 1024 && (ldx & 3) == 0 && ((uintptr_t)x & 15) == 0);
  SGNN_CHECK_ARG(epi.ldx >= cout && epi.ldx <= 1024 && epi.ldy >= cin && epi.ldy <= 1024);
  SGNN_CHECK_ARG(epi.stats == 0 || (epi.stats == 2 && epi.partial && epi.bn_x && epi.mean && epi.invstd));
  SGNN_CHECK_ARG(conv_wide_epi_ok(epi, dx, FUSED_K));
  const int64_t lmax = epi.ldy > epi.ld_add ? (epi.ldy > epi.ld_bnx ? epi.ldy : epi.ld_bnx)
                                            : (epi.ld_add > epi.ld_bnx ? epi.ld_add : epi.ld_bnx);
  if (n * epi.ldx * 4 > 0xFFFFF000ll || n * lmax * 4 > 0xFFFFF000ll || n * ldx * 4 > 0xFFFFF000ll ||
      (int64_t)FUSED_K * ld * 4 > 0xFFFFF000ll) {
    sgnn_set_error("sgnn_conv_bwd_fused: a slab exceeds the 4 GiB raw-buffer window");
    return SGNN_EOVERFLOW;
  }
  const int cap = fused_wg_cap();
  const int64_t grid4 = (n + CONV_ROWS_PER_BLOCK - 1) / CONV_ROWS_PER_BLOCK;
  const int64_t pblocks = grid4 < cap ? grid4 : cap;
  if (!ws || ws_bytes < pblocks * FUSED_K * (int64_t)cin * cout * (int64_t)sizeof(float)) {
    sgnn_set_error("sgnn_conv_bwd_fused: workspace too small");
    return SGNN_ENOWS;
  }
  hipStream_t s = (hipStream_t)stream;
  const int prof = sgnn_prof_begin_launch(2, n, cout, cin, FUSED_K, SGNN_CONV_TRANSPOSE_W | SGNN_CONV_FLIP_K, s);
  SGNN_LAUNCH((k_conv_bwd_fused<16>), dim3((unsigned)grid4), dim3(256), 0, s, dy, n, w, table, ld, n, dx, epi, cap, x, ldx,
              (float *)ws, (int)pblocks);
  sgnn_prof_end_launch(prof, s);
  SGNN_CHECK_LAUNCH();
  // The reduce runs HERE, on the caller's stream, not in the program's deferred batch on the weight-gradient lane: that batch
  // would need one more cross-stream edge per program (lane waits for the training stream's last kernel), and such an edge
  // costs a replayed step 0.15 ms apiece (profiles/r06l_ab_endfork.txt: +0.75 ms for five of them with no kernel moved).
  DwBatch *const batch = sgnn_dw_batch;
  sgnn_dw_batch = nullptr;
  const int rc = sgnn_dw_reduce_or_defer((const float *)ws, dw, pblocks, (int64_t)FUSED_K * cin * cout, s);
  sgnn_dw_batch = batch;
  return rc;
}

SGNN_EXPORT int64_t sgnn_conv_bwd_fused_ws_bytes(int64_t n, int cin, int cout) { return sgnn_conv_bwd_fused_ws(n, cin, cout); }

SGNN_EXPORT int sgnn_conv_bwd_fused_supported(int64_t n, int cin, int cout, int K) {
  return sgnn_conv_bwd_fused_ok(n, cin, cout, K) ? 1 : 0;
}

SGNN_EXPORT int sgnn_conv_bwd_fused(const float *dy, int64_t n, int cout, int64_t ld_dy, const float *x, int cin, int64_t ldx,
                                    const float *w, const int32_t *table, int64_t ld, float *dx, int64_t ld_dx,
                                    const float *addend, int64_t ld_add, int stats, double *partial, const float *bn_x,
                                    int64_t ld_bnx, const float *mean, const float *invstd, const float *gamma,
                                    const float *beta, float leak, float *dw, void *ws, int64_t ws_bytes,
                                    const int64_t *n_dev, sgnn_stream_t stream) {
  ConvEpi epi{ld_dy, ld_dx, ld_add, addend, stats, partial, bn_x, ld_bnx, mean, invstd, gamma, beta, leak, n_dev};
  return sgnn_conv_bwd_fused_impl(dy, n, cout, w, table, ld, cin, dx, epi, x, ldx, dw, ws, ws_bytes, stream);
}
