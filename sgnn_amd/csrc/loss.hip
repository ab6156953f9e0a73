// Fused sparse-prediction vs dense-target loss of one hierarchy level (the caller
// immediately downstream of the path): gathers the dense targets / weights / masks at the predicted
// sites and evaluates
//     mean over kept sites of  w * BCE_with_logits(occ, tgt_occ)      torch/loss.py:58-82
//     mean over kept sites of  w * | logt(sdf) - logt(tgt_sdf) |       torch/loss.py:122-157, :51-55
// in one pass (the reference issues ~25 torch ops and an int32 volume per level and step).
// HBM-bound gather + deterministic fp64 two-stage reduction; backward is a second gather pass.
#include "common.h"

#define LOSS_MAX_BLOCKS 512
#define UNK_ID_F (-1.0f)
#define UNK_THRESH_U8 2

struct LossArgs {
  const int64_t *locs;    // (M,4) z,y,x,b
  const float *vals;      // (M, vstride)
  int vstride, occ_col, sdf_col;   // column of the occupancy logit (-1: none) and of the sdf value (-1: none)
  const float *tgt_occ;   // dense (B,1,d0,d1,d2) occupancy target with UNK_ID = -1 (NULL if occ_col < 0)
  const float *tgt_sdf;   // dense sdf target
  const float *weights;   // dense weights or NULL
  const uint8_t *known;   // dense u8 (final level mask) or NULL
  int d0, d1, d2;
  int64_t m;
  int use_log;            // loss.py:139-141 log transform
  int mask_mode;          // 0: keep all (unknown occupancy targets count as 0), 1: keep tgt_occ != UNK_ID, 2: keep known < 2
};

__device__ __forceinline__ float logt(float v) { return copysignf(logf(fabsf(v) + 1.0f), v) * (v != 0.f); }

// per-site terms; returns whether the site is kept
__device__ __forceinline__ bool loss_site(const LossArgs &a, int64_t r, float &bce, float &l1, float &dbce, float &dl1) {
  const longlong2 p0 = reinterpret_cast<const longlong2 *>(a.locs)[2 * r];
  const longlong2 p1 = reinterpret_cast<const longlong2 *>(a.locs)[2 * r + 1];
  const int64_t fl = ((p1.y * a.d0 + p0.x) * a.d1 + p0.y) * a.d2 + p1.x;
  const float w = a.weights ? a.weights[fl] : 1.0f;
  float to = 0.f;
  bool keep = true;
  if (a.tgt_occ) to = a.tgt_occ[fl];
  if (a.mask_mode == 1) keep = (to != UNK_ID_F);
  if (a.mask_mode == 2) keep = a.known[fl] < UNK_THRESH_U8;
  if (a.mask_mode == 0 && to == UNK_ID_F) to = 0.f;
  bce = l1 = dbce = dl1 = 0.f;
  if (!keep) return false;
  if (a.occ_col >= 0) {
    const float x = a.vals[r * a.vstride + a.occ_col];
    const float t = fmaxf(to, 0.f);
    bce = w * (fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))));
    dbce = w * (1.0f / (1.0f + expf(-x)) - t);
  }
  if (a.sdf_col >= 0) {
    const float p = a.vals[r * a.vstride + a.sdf_col];
    const float t = a.tgt_sdf[fl];
    const float d = a.use_log ? (logt(p) - logt(t)) : (p - t);
    l1 = w * fabsf(d);
    const float sg = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
    dl1 = w * sg * (a.use_log ? 1.0f / (fabsf(p) + 1.0f) : 1.0f);
  }
  return true;
}

__global__ __launch_bounds__(256) void k_loss_partial(LossArgs a, double *__restrict__ partial) {
  __shared__ double sh[3][256];
  double s_b = 0.0, s_l = 0.0, s_n = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < a.m; r += stride) {
    float bce, l1, db, dl;
    if (loss_site(a, r, bce, l1, db, dl)) {
      s_b += (double)bce;
      s_l += (double)l1;
      s_n += 1.0;
    }
  }
  sh[0][threadIdx.x] = s_b;
  sh[1][threadIdx.x] = s_l;
  sh[2][threadIdx.x] = s_n;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d)
      for (int c = 0; c < 3; ++c) sh[c][threadIdx.x] += sh[c][threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x < 3) partial[blockIdx.x * 3 + threadIdx.x] = sh[threadIdx.x][0];
}

// sums[0..2] = {sum bce, sum l1, kept count}; out2 = {bce mean, l1 mean} (0/0 = nan like an empty mean)
__global__ __launch_bounds__(256) void k_loss_finalize(const double *__restrict__ partial, int nblk,
                                                      double *__restrict__ sums, float *__restrict__ out2) {
  __shared__ double sh[3][256];
  double s[3] = {0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < nblk; b += 256)
    for (int c = 0; c < 3; ++c) s[c] += partial[b * 3 + c];
  for (int c = 0; c < 3; ++c) sh[c][threadIdx.x] = s[c];
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d)
      for (int c = 0; c < 3; ++c) sh[c][threadIdx.x] += sh[c][threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sums[0] = sh[0][0];
    sums[1] = sh[1][0];
    sums[2] = sh[2][0];
    out2[0] = (float)(sh[0][0] / sh[2][0]);
    out2[1] = (float)(sh[1][0] / sh[2][0]);
  }
}

// dvals[r][occ_col] = g[0] * dbce / kept,  dvals[r][sdf_col] = g[1] * dl1 / kept; other columns zero
__global__ __launch_bounds__(256) void k_loss_bwd(LossArgs a, const double *__restrict__ sums,
                                                 const float *__restrict__ gout2, float *__restrict__ dvals) {
  const double kept = sums[2];
  const float g0 = (float)((double)gout2[0] / kept), g1 = (float)((double)gout2[1] / kept);
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < a.m; r += stride) {
    float bce, l1, db, dl;
    const bool keep = loss_site(a, r, bce, l1, db, dl);
    for (int c = 0; c < a.vstride; ++c) {
      float v = 0.f;
      if (keep && c == a.occ_col) v = g0 * db;
      if (keep && c == a.sdf_col) v = g1 * dl;
      dvals[r * a.vstride + c] = v;
    }
  }
}

static int loss_blocks(int64_t m) {
  int64_t b = (m + 1023) / 1024;
  if (b < 1) b = 1;
  if (b > LOSS_MAX_BLOCKS) b = LOSS_MAX_BLOCKS;
  return (int)b;
}

SGNN_EXPORT int64_t sgnn_loss_ws_bytes(void) { return (int64_t)LOSS_MAX_BLOCKS * 3 * sizeof(double) + 64; }

static int fill_args(LossArgs &a, const int64_t *locs, const float *vals, int vstride, int occ_col, int sdf_col,
                     const float *tgt_occ, const float *tgt_sdf, const float *weights, const uint8_t *known, int d0,
                     int d1, int d2, int64_t m, int use_log, int mask_mode) {
  if (!(m >= 0 && vstride >= 1 && occ_col < vstride && sdf_col < vstride && (occ_col >= 0 || sdf_col >= 0))) return -1;
  if (m > 0 && (!locs || !vals)) return -1;
  if (occ_col >= 0 && !tgt_occ) return -1;
  if (sdf_col >= 0 && !tgt_sdf) return -1;
  if (mask_mode == 1 && !tgt_occ) return -1;
  if (mask_mode == 2 && !known) return -1;
  if (mask_mode < 0 || mask_mode > 2) return -1;
  a = LossArgs{locs, vals, vstride, occ_col, sdf_col, tgt_occ, tgt_sdf, weights, known, d0, d1, d2, m, use_log, mask_mode};
  return 0;
}

SGNN_EXPORT int sgnn_loss_level_fwd(const int64_t *locs, const float *vals, int vstride, int occ_col, int sdf_col,
                                    const float *tgt_occ, const float *tgt_sdf, const float *weights,
                                    const uint8_t *known, int d0, int d1, int d2, int64_t m, int use_log,
                                    int mask_mode, double *sums, float *out2, void *ws, int64_t ws_bytes,
                                    sgnn_stream_t stream) {
  LossArgs a;
  SGNN_CHECK_ARG(fill_args(a, locs, vals, vstride, occ_col, sdf_col, tgt_occ, tgt_sdf, weights, known, d0, d1, d2, m,
                           use_log, mask_mode) == 0);
  SGNN_CHECK_ARG(sums && out2);
  if (!ws || ws_bytes < sgnn_loss_ws_bytes()) {
    sgnn_set_error("sgnn_loss_level_fwd: workspace too small");
    return SGNN_ENOWS;
  }
  hipStream_t s = (hipStream_t)stream;
  const int nblk = loss_blocks(m);
  hipLaunchKernelGGL(k_loss_partial, dim3(nblk), dim3(256), 0, s, a, (double *)ws);
  hipLaunchKernelGGL(k_loss_finalize, dim3(1), dim3(256), 0, s, (const double *)ws, nblk, sums, out2);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_loss_level_bwd(const int64_t *locs, const float *vals, int vstride, int occ_col, int sdf_col,
                                    const float *tgt_occ, const float *tgt_sdf, const float *weights,
                                    const uint8_t *known, int d0, int d1, int d2, int64_t m, int use_log,
                                    int mask_mode, const double *sums, const float *gout2, float *dvals,
                                    sgnn_stream_t stream) {
  LossArgs a;
  SGNN_CHECK_ARG(fill_args(a, locs, vals, vstride, occ_col, sdf_col, tgt_occ, tgt_sdf, weights, known, d0, d1, d2, m,
                           use_log, mask_mode) == 0);
  if (m == 0) return SGNN_OK;
  SGNN_CHECK_ARG(sums && gout2 && dvals);
  hipLaunchKernelGGL(k_loss_bwd, dim3(sgnn_grid_for(m, 256, 2048)), dim3(256), 0, (hipStream_t)stream, a, sums, gout2,
                     dvals);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}
