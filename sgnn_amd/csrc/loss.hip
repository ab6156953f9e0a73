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
  const int64_t *m_dev;   // capacity mode: the live row count (clamped to m), NULL = m is exact
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
  const int64_t m = sgnn_dyn_n(a.m, a.m_dev);
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < m; r += stride) {
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
                                                      double *__restrict__ sums, float *__restrict__ out2,
                                                      const int64_t *m_dev) {
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
    // capacity mode with no predicted site at all: the reference skips the level (torch/loss.py:166, len(...) == 0)
    const bool empty = m_dev && *m_dev <= 0;
    out2[0] = empty ? 0.f : (float)(sh[0][0] / sh[2][0]);
    out2[1] = empty ? 0.f : (float)(sh[1][0] / sh[2][0]);
  }
}

// dvals[r][occ_col] = g[0] * dbce / kept,  dvals[r][sdf_col] = g[1] * dl1 / kept; other columns zero
__global__ __launch_bounds__(256) void k_loss_bwd(LossArgs a, const double *__restrict__ sums,
                                                 const float *__restrict__ gout2, float *__restrict__ dvals) {
  const double kept = sums[2];
  const float g0 = (float)((double)gout2[0] / kept), g1 = (float)((double)gout2[1] / kept);
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t m = sgnn_dyn_n(a.m, a.m_dev);
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < m; r += stride) {
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
                     int d1, int d2, int64_t m, int use_log, int mask_mode, const int64_t *m_dev) {
  if (!(m >= 0 && vstride >= 1 && occ_col < vstride && sdf_col < vstride && (occ_col >= 0 || sdf_col >= 0))) return -1;
  if (m > 0 && (!locs || !vals)) return -1;
  if (occ_col >= 0 && !tgt_occ) return -1;
  if (sdf_col >= 0 && !tgt_sdf) return -1;
  if (mask_mode == 1 && !tgt_occ) return -1;
  if (mask_mode == 2 && !known) return -1;
  if (mask_mode < 0 || mask_mode > 2) return -1;
  a = LossArgs{locs, vals, vstride, occ_col, sdf_col, tgt_occ, tgt_sdf, weights, known, d0, d1, d2, m, m_dev, use_log, mask_mode};
  return 0;
}

SGNN_EXPORT int sgnn_loss_level_fwd(const int64_t *locs, const float *vals, int vstride, int occ_col, int sdf_col,
                                    const float *tgt_occ, const float *tgt_sdf, const float *weights,
                                    const uint8_t *known, int d0, int d1, int d2, int64_t m, int use_log,
                                    int mask_mode, const int64_t *m_dev, double *sums, float *out2, void *ws,
                                    int64_t ws_bytes, sgnn_stream_t stream) {
  LossArgs a;
  SGNN_CHECK_ARG(fill_args(a, locs, vals, vstride, occ_col, sdf_col, tgt_occ, tgt_sdf, weights, known, d0, d1, d2, m,
                           use_log, mask_mode, m_dev) == 0);
  SGNN_CHECK_ARG(sums && out2);
  if (!ws || ws_bytes < sgnn_loss_ws_bytes()) {
    sgnn_set_error("sgnn_loss_level_fwd: workspace too small");
    return SGNN_ENOWS;
  }
  hipStream_t s = (hipStream_t)stream;
  const int nblk = loss_blocks(m);
  SGNN_LAUNCH(k_loss_partial, dim3(nblk), dim3(256), 0, s, a, (double *)ws);
  SGNN_LAUNCH(k_loss_finalize, dim3(1), dim3(256), 0, s, (const double *)ws, nblk, sums, out2, m_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_loss_level_bwd(const int64_t *locs, const float *vals, int vstride, int occ_col, int sdf_col,
                                    const float *tgt_occ, const float *tgt_sdf, const float *weights,
                                    const uint8_t *known, int d0, int d1, int d2, int64_t m, int use_log,
                                    int mask_mode, const int64_t *m_dev, const double *sums, const float *gout2,
                                    float *dvals, sgnn_stream_t stream) {
  LossArgs a;
  SGNN_CHECK_ARG(fill_args(a, locs, vals, vstride, occ_col, sdf_col, tgt_occ, tgt_sdf, weights, known, d0, d1, d2, m,
                           use_log, mask_mode, m_dev) == 0);
  if (m == 0) return SGNN_OK;
  SGNN_CHECK_ARG(sums && gout2 && dvals);
  SGNN_LAUNCH(k_loss_bwd, dim3(sgnn_grid_for(m, 256, 2048)), dim3(256), 0, (hipStream_t)stream, a, sums, gout2,
                     dvals);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// ---------------------------------------------------------------------------
// All levels of the hierarchical loss in one submission: sgnn_loss_level_fwd x n + sgnn_loss_combine as TWO launches,
// sgnn_loss_combine_bwd + sgnn_loss_level_bwd x n as ONE (17 -> 3 launches per training step).  Per level the same
// block partition, the same summation order and therefore the same bits as the per-level entry points.
// ---------------------------------------------------------------------------
#define LOSS_MAX_LEVELS 5
struct LossMulti {
  LossArgs a[LOSS_MAX_LEVELS];
  float coef[2 * LOSS_MAX_LEVELS];
  int nblk[LOSS_MAX_LEVELS];
  int n;
};
struct LossOut {
  float *dvals[LOSS_MAX_LEVELS];
};

__global__ __launch_bounds__(256) void k_loss_partial_multi(LossMulti m, double *__restrict__ partial) {
  __shared__ double sh[3][256];
  const int l = blockIdx.y;
  const LossArgs &a = m.a[l];
  if ((int)blockIdx.x >= m.nblk[l]) return;
  double s_b = 0.0, s_l = 0.0, s_n = 0.0;
  const int64_t stride = (int64_t)m.nblk[l] * 256;
  const int64_t rows = sgnn_dyn_n(a.m, a.m_dev);
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += stride) {
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
  if (threadIdx.x < 3) partial[((size_t)l * LOSS_MAX_BLOCKS + blockIdx.x) * 3 + threadIdx.x] = sh[threadIdx.x][0];
}

// one workgroup: per-level totals and means, then total = sum coef * out2 and the per-level values for logging
__global__ __launch_bounds__(256) void k_loss_finalize_multi(const double *__restrict__ partial, LossMulti m,
                                                            double *__restrict__ sums, float *__restrict__ out2s,
                                                            float *__restrict__ total, float *__restrict__ cur) {
  __shared__ double sh[3][256];
  for (int l = 0; l < m.n; ++l) {
    double s[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < m.nblk[l]; b += 256)
      for (int c = 0; c < 3; ++c) s[c] += partial[((size_t)l * LOSS_MAX_BLOCKS + b) * 3 + c];
    for (int c = 0; c < 3; ++c) sh[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
      if (threadIdx.x < d)
        for (int c = 0; c < 3; ++c) sh[c][threadIdx.x] += sh[c][threadIdx.x + d];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      sums[3 * l + 0] = sh[0][0];
      sums[3 * l + 1] = sh[1][0];
      sums[3 * l + 2] = sh[2][0];
      const bool empty = m.a[l].m_dev && *m.a[l].m_dev <= 0;
      out2s[2 * l] = empty ? 0.f : (float)(sh[0][0] / sh[2][0]);
      out2s[2 * l + 1] = empty ? 0.f : (float)(sh[1][0] / sh[2][0]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 2 * m.n; ++i)
      if (m.coef[i] != 0.f) t += m.coef[i] * out2s[i];
    *total = t;
    for (int l = 0; l < m.n; ++l)
      cur[l] = (m.coef[2 * l] != 0.f ? out2s[2 * l] : 0.f) + (m.coef[2 * l + 1] != 0.f ? out2s[2 * l + 1] : 0.f);
  }
}

__global__ __launch_bounds__(256) void k_loss_bwd_multi(LossMulti m, const double *__restrict__ sums,
                                                       const float *__restrict__ g, LossOut o) {
  const int l = blockIdx.y;
  const LossArgs &a = m.a[l];
  float *dvals = o.dvals[l];
  const double kept = sums[3 * l + 2];
  const float g0 = (float)((double)(g[0] * m.coef[2 * l]) / kept), g1 = (float)((double)(g[0] * m.coef[2 * l + 1]) / kept);
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t rows = sgnn_dyn_n(a.m, a.m_dev);
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += stride) {
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

SGNN_EXPORT int64_t sgnn_loss_multi_ws_bytes(void) {
  return (int64_t)LOSS_MAX_LEVELS * LOSS_MAX_BLOCKS * 3 * sizeof(double) + 64;
}

// levels: HOST array of n x 16 int64 = {locs, vals, vstride, occ_col, sdf_col, tgt_occ, tgt_sdf, weights, known, d0, d1,
// d2, m, use_log, mask_mode, m_dev} (pointers as integers); coef: HOST 2n floats (bce, l1 weight per level)
static int fill_multi(LossMulti &m, const int64_t *levels, int n, const float *coef) {
  if (n < 1 || n > LOSS_MAX_LEVELS || !levels || !coef) return -1;
  m.n = n;
  for (int l = 0; l < n; ++l) {
    const int64_t *v = levels + 16 * l;
    if (fill_args(m.a[l], (const int64_t *)(uintptr_t)v[0], (const float *)(uintptr_t)v[1], (int)v[2], (int)v[3], (int)v[4],
                  (const float *)(uintptr_t)v[5], (const float *)(uintptr_t)v[6], (const float *)(uintptr_t)v[7],
                  (const uint8_t *)(uintptr_t)v[8], (int)v[9], (int)v[10], (int)v[11], v[12], (int)v[13], (int)v[14],
                  (const int64_t *)(uintptr_t)v[15]) != 0)
      return -1;
    m.nblk[l] = loss_blocks(v[12]);
    m.coef[2 * l] = coef[2 * l];
    m.coef[2 * l + 1] = coef[2 * l + 1];
  }
  return 0;
}

SGNN_EXPORT int sgnn_loss_levels_fwd(const int64_t *levels, int n, const float *coef_host, double *sums, float *out2s,
                                     float *total, float *cur, void *ws, int64_t ws_bytes, sgnn_stream_t stream) {
  LossMulti m{};
  SGNN_CHECK_ARG(fill_multi(m, levels, n, coef_host) == 0);
  SGNN_CHECK_ARG(sums && out2s && total && cur);
  if (!ws || ws_bytes < sgnn_loss_multi_ws_bytes()) {
    sgnn_set_error("sgnn_loss_levels_fwd: workspace too small");
    return SGNN_ENOWS;
  }
  int gx = 1;
  for (int l = 0; l < n; ++l) gx = m.nblk[l] > gx ? m.nblk[l] : gx;
  hipStream_t s = (hipStream_t)stream;
  SGNN_LAUNCH(k_loss_partial_multi, dim3(gx, n), dim3(256), 0, s, m, (double *)ws);
  SGNN_LAUNCH(k_loss_finalize_multi, dim3(1), dim3(256), 0, s, (const double *)ws, m, sums, out2s, total, cur);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// g: device float (d total); dvals: HOST array of n device pointers (m_l x vstride_l each)
SGNN_EXPORT int sgnn_loss_levels_bwd(const int64_t *levels, int n, const float *coef_host, const double *sums,
                                     const float *g, void *const *dvals, sgnn_stream_t stream) {
  LossMulti m{};
  SGNN_CHECK_ARG(fill_multi(m, levels, n, coef_host) == 0);
  SGNN_CHECK_ARG(sums && g && dvals);
  LossOut o{};
  int64_t mmax = 1;
  for (int l = 0; l < n; ++l) {
    SGNN_CHECK_ARG(dvals[l] || m.a[l].m == 0);
    o.dvals[l] = (float *)dvals[l];
    mmax = m.a[l].m > mmax ? m.a[l].m : mmax;
  }
  SGNN_LAUNCH(k_loss_bwd_multi, dim3(sgnn_grid_for(mmax, 256, 2048), n), dim3(256), 0, (hipStream_t)stream, m, sums, g, o);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// ---------------------------------------------------------------------------
// Targets of the hierarchical loss — compute_targets + compute_weights_missing_geo (torch/loss.py:15-32, :35-48) in
// three launches instead of ~35 tensor ops and an int32 volume:
//   fine   : tsdf = clamp(sdf); hier[L-1] = tsdf; occ[L-1] = |tsdf| < trunc (UNK_ID where known >= 2);
//            w[L-1] = (|occ| <= trunc) ? weight_missing_geo : 1           (every voxel treated as "not an input site")
//   sites  : w[L-1] = 1 at the input sites   (loss.py:42-45: 1 + 1 + 3 = 5 != 4)
//   coarse : occ[h] = 2x2x2 max of occ[h+1]; w[h] = w[h+1][::2,::2,::2]; hier[h] = clamp(hierarchy[h]) for the (up to
//            three) coarser levels, one wave per 8^3 fine voxels (4x4x4 voxels of level L-2), maxima by lane shuffles.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_targets_fine(const float *__restrict__ sdf, const uint8_t *__restrict__ known,
                                                     int64_t total, float trunc, int masking, float wmg,
                                                     float *__restrict__ tsdf, float *__restrict__ hier_last,
                                                     float *__restrict__ occ, float *__restrict__ w) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
    const float t = fminf(fmaxf(sdf[i], -trunc), trunc);          // clamp keeps -inf -> -trunc, like torch.clamp_
    tsdf[i] = t;
    hier_last[i] = t;
    float o = fabsf(t) < trunc ? 1.f : 0.f;
    if (masking && known[i] >= UNK_THRESH_U8) o = UNK_ID_F;
    occ[i] = o;
    if (w) w[i] = (fabsf(o) <= trunc) ? wmg : 1.f;
  }
}

__global__ __launch_bounds__(256) void k_targets_sites(const int64_t *__restrict__ locs, int64_t n, int batch, int d0,
                                                      int d1, int d2, float *__restrict__ w, const int64_t *n_dev) {
  n = sgnn_dyn_n(n, n_dev);
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
    const longlong2 p0 = reinterpret_cast<const longlong2 *>(locs)[2 * r];
    const longlong2 p1 = reinterpret_cast<const longlong2 *>(locs)[2 * r + 1];
    if (p0.x < 0 || p0.x >= d0 || p0.y < 0 || p0.y >= d1 || p1.x < 0 || p1.x >= d2 || p1.y < 0 || p1.y >= batch) continue;
    w[((p1.y * d0 + p0.x) * d1 + p0.y) * d2 + p1.x] = 1.f;
  }
}

struct CoarseArgs {
  const float *occ_f, *w_f;          // finest level (B, d0, d1, d2)
  const float *hier_in[3];           // hierarchy inputs of levels L-2, L-3, L-4 (NULL: level absent)
  float *occ[3], *w[3], *hier[3];    // outputs of those levels (w[k] NULL when weights are off)
  int batch, d0, d1, d2;             // FINE dims; level L-2-k has dims d >> (k+1)
  float trunc;
  int nlev;                          // coarse levels to produce (1..3)
};

__global__ __launch_bounds__(256) void k_targets_coarse(CoarseArgs a) {
  // one wave = one 4x4x4 brick of level L-2 voxels; lane = (lz*4 + ly)*4 + lx
  const int lane = threadIdx.x & 63;
  const int lz = lane >> 4, ly = (lane >> 2) & 3, lx = lane & 3;
  const int c0 = a.d0 >> 1, c1 = a.d1 >> 1, c2 = a.d2 >> 1;                       // level L-2 dims
  const int b0 = (c0 + 3) >> 2, b1 = (c1 + 3) >> 2, b2 = (c2 + 3) >> 2;           // bricks per axis
  const int64_t bricks = (int64_t)a.batch * b0 * b1 * b2;
  const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t br = wave0; br < bricks; br += nwaves) {
    int64_t t = br;
    const int bx = (int)(t % b2); t /= b2;
    const int by = (int)(t % b1); t /= b1;
    const int bz = (int)(t % b0);
    const int b = (int)(t / b0);
    const int z = bz * 4 + lz, y = by * 4 + ly, x = bx * 4 + lx;
    const bool in = z < c0 && y < c1 && x < c2;
    float m = -INFINITY;
    if (in) {
      const float *p = a.occ_f + (((int64_t)b * a.d0 + 2 * z) * a.d1 + 2 * y) * a.d2 + 2 * x;
#pragma unroll
      for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const float2 v = *reinterpret_cast<const float2 *>(p + ((int64_t)dz * a.d1 + dy) * a.d2);
          m = fmaxf(m, fmaxf(v.x, v.y));
        }
      const int64_t o = (((int64_t)b * c0 + z) * c1 + y) * c2 + x;
      a.occ[0][o] = m;
      if (a.w[0]) a.w[0][o] = a.w_f[(((int64_t)b * a.d0 + 2 * z) * a.d1 + 2 * y) * a.d2 + 2 * x];
      a.hier[0][o] = fminf(fmaxf(a.hier_in[0][o], -a.trunc), a.trunc);
    }
    if (a.nlev >= 2) {   // level L-3: max over the 2x2x2 lanes that differ in the low bit of lz, ly, lx
      m = fmaxf(m, __shfl_xor(m, 1));
      m = fmaxf(m, __shfl_xor(m, 4));
      m = fmaxf(m, __shfl_xor(m, 16));
      const int e0 = c0 >> 1, e1 = c1 >> 1, e2 = c2 >> 1;
      const int zz = z >> 1, yy = y >> 1, xx = x >> 1;
      if (((lz | ly | lx) & 1) == 0 && zz < e0 && yy < e1 && xx < e2) {
        const int64_t o = (((int64_t)b * e0 + zz) * e1 + yy) * e2 + xx;
        a.occ[1][o] = m;
        if (a.w[1]) a.w[1][o] = a.w_f[(((int64_t)b * a.d0 + 4 * zz) * a.d1 + 4 * yy) * a.d2 + 4 * xx];
        a.hier[1][o] = fminf(fmaxf(a.hier_in[1][o], -a.trunc), a.trunc);
      }
      if (a.nlev >= 3) {   // level L-4: the whole brick
        m = fmaxf(m, __shfl_xor(m, 2));
        m = fmaxf(m, __shfl_xor(m, 8));
        m = fmaxf(m, __shfl_xor(m, 32));
        const int f0 = e0 >> 1, f1 = e1 >> 1, f2 = e2 >> 1;
        if (lane == 0 && bz < f0 && by < f1 && bx < f2) {
          const int64_t o = (((int64_t)b * f0 + bz) * f1 + by) * f2 + bx;
          a.occ[2][o] = m;
          if (a.w[2]) a.w[2][o] = a.w_f[(((int64_t)b * a.d0 + 8 * bz) * a.d1 + 8 * by) * a.d2 + 8 * bx];
          a.hier[2][o] = fminf(fmaxf(a.hier_in[2][o], -a.trunc), a.trunc);
        }
      }
    }
  }
}

// all outputs are caller-allocated; hier_in / occ / w / hier list the coarser levels from L-2 downwards (host arrays of
// device pointers, ncoarse = L - 1 <= 3); w_last NULL (and w entries NULL) switches the weights off.  Every dimension
// must be divisible by 2^ncoarse (MaxPool3d(2) floors otherwise; the caller falls back to tensor ops then).
SGNN_EXPORT int sgnn_loss_targets(const float *sdf, const uint8_t *known, const int64_t *input_locs, int64_t n_locs,
                                  const int64_t *n_locs_dev, int batch, int d0, int d1, int d2, float trunc, int masking, float weight_missing_geo,
                                  int ncoarse, void *const *hier_in, float *tsdf, float *hier_last, float *occ_last,
                                  float *w_last, void *const *occ, void *const *w, void *const *hier,
                                  sgnn_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  SGNN_CHECK_ARG(batch >= 0 && d0 >= 1 && d1 >= 1 && d2 >= 1 && ncoarse >= 0 && ncoarse <= 3 && n_locs >= 0);
  SGNN_CHECK_ARG((d0 % (1 << ncoarse)) == 0 && (d1 % (1 << ncoarse)) == 0 && (d2 % (1 << ncoarse)) == 0);
  const int64_t total = (int64_t)batch * d0 * d1 * d2;
  if (total == 0) return SGNN_OK;
  SGNN_CHECK_ARG(sdf && tsdf && hier_last && occ_last && (!masking || known) && (!w_last || n_locs == 0 || input_locs));
  SGNN_CHECK_ARG(ncoarse == 0 || (hier_in && occ && hier && (!w_last || w)));
  SGNN_LAUNCH(k_targets_fine, dim3(sgnn_grid_for(total, 256, 4096)), dim3(256), 0, s, sdf, known, total, trunc,
                     masking, weight_missing_geo, tsdf, hier_last, occ_last, w_last);
  if (w_last && n_locs > 0)
    SGNN_LAUNCH(k_targets_sites, dim3(sgnn_grid_for(n_locs, 256, 4096)), dim3(256), 0, s, input_locs, n_locs,
                       batch, d0, d1, d2, w_last, n_locs_dev);
  if (ncoarse > 0) {
    CoarseArgs a{};
    a.occ_f = occ_last;
    a.w_f = w_last;
    for (int k = 0; k < ncoarse; ++k) {
      SGNN_CHECK_ARG(hier_in[k] && occ[k] && hier[k] && (!w_last || w[k]));
      a.hier_in[k] = (const float *)hier_in[k];
      a.occ[k] = (float *)occ[k];
      a.w[k] = w_last ? (float *)w[k] : nullptr;
      a.hier[k] = (float *)hier[k];
    }
    a.batch = batch; a.d0 = d0; a.d1 = d1; a.d2 = d2; a.trunc = trunc; a.nlev = ncoarse;
    const int64_t bricks = (int64_t)batch * ((d0 / 2 + 3) / 4) * ((d1 / 2 + 3) / 4) * ((d2 / 2 + 3) / 4);
    SGNN_LAUNCH(k_targets_coarse, dim3(sgnn_grid_for(bricks, 4, 8192)), dim3(256), 0, s, a);
  }
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// ---------------------------------------------------------------------------
// loss = sum_i coef[i] * out2s[i]  over the (bce, l1) pairs of all levels (torch/loss.py:160-199), per-level values
// cur[l] = out2s[2l] + out2s[2l+1] for logging, and the matching gradient fan-out g2[i] = g * coef[i].
// ---------------------------------------------------------------------------
struct Coef10 {
  float c[10];
};

__global__ void k_loss_combine(const float *__restrict__ out2s, Coef10 coef, int n, float *__restrict__ total,
                               float *__restrict__ cur) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < n; ++i)
      if (coef.c[i] != 0.f) t += coef.c[i] * out2s[i];
    *total = t;
    for (int l = 0; 2 * l < n; ++l)
      cur[l] = (coef.c[2 * l] != 0.f ? out2s[2 * l] : 0.f) + (coef.c[2 * l + 1] != 0.f ? out2s[2 * l + 1] : 0.f);
  }
}

__global__ void k_loss_combine_bwd(const float *__restrict__ g, Coef10 coef, int n, float *__restrict__ g2) {
  if (threadIdx.x < n && blockIdx.x == 0) g2[threadIdx.x] = g[0] * coef.c[threadIdx.x];
}

SGNN_EXPORT int sgnn_loss_combine(const float *out2s, const float *coef_host, int n, float *total, float *cur,
                                  sgnn_stream_t stream) {
  SGNN_CHECK_ARG(out2s && coef_host && total && cur && n >= 1 && n <= 10);
  Coef10 c{};
  for (int i = 0; i < n; ++i) c.c[i] = coef_host[i];
  SGNN_LAUNCH(k_loss_combine, dim3(1), dim3(64), 0, (hipStream_t)stream, out2s, c, n, total, cur);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_loss_combine_bwd(const float *g, const float *coef_host, int n, float *g2, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(g && coef_host && g2 && n >= 1 && n <= 10);
  Coef10 c{};
  for (int i = 0; i < n; ++i) c.c[i] = coef_host[i];
  SGNN_LAUNCH(k_loss_combine_bwd, dim3(1), dim3(64), 0, (hipStream_t)stream, g, c, n, g2);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}
