// Evaluation metrics of sparse predictions against dense targets, on the device.
//
// Reference being replaced:
//   torch/loss.py:84-120   compute_iou_sparse_dense      nonzero + .cpu().numpy() + np.intersect1d/union1d per sample
//   torch/loss.py:201-231  compute_l1_tgtsurf_sparse_dense  full dense scatter of the prediction + nonzero + gathers
// (both run every 20th iteration and in validation: torch/train.py:271-297, 353-378).
//
// IoU needs no set machinery: predicted sites are unique, so with
//     P = #{kept predicted sites whose target is not UNKNOWN},  C = #{those with target == 1},
//     T = #{voxels with target == 1}
// intersection = C and union = P + T - C.  Integer counters, so the result does not depend on summation order.
//
// Target-surface L1 without materialising the dense prediction: every target-surface voxel starts from the fill
// value -truncation (loss.py:207-208), so
//     sum |pred - t|  =  sum_{surface voxels} |-trunc - t|  +  sum_{predicted sites on the surface} (|p - t| - |-trunc - t|)
// one dense streaming pass plus one pass over the predicted sites, fp64 partial sums reduced in fixed order.
#include "common.h"

#define MET_MAX_BLOCKS 1024
#define MET_LDS_SAMPLES 512
#define UNK_F (-1.0f)

struct DenseGeom {
  int nb, d0, d1, d2;
};

__device__ __forceinline__ bool met_flat(const int64_t *__restrict__ locs, int64_t r, const DenseGeom &g, int64_t &fl,
                                         int &b) {
  const longlong2 p0 = reinterpret_cast<const longlong2 *>(locs)[2 * r];
  const longlong2 p1 = reinterpret_cast<const longlong2 *>(locs)[2 * r + 1];
  if ((uint64_t)p0.x >= (uint64_t)g.d0 || (uint64_t)p0.y >= (uint64_t)g.d1 || (uint64_t)p1.x >= (uint64_t)g.d2 ||
      (uint64_t)p1.y >= (uint64_t)g.nb)
    return false;
  b = (int)p1.y;
  fl = ((p1.y * g.d0 + p0.x) * g.d1 + p0.y) * g.d2 + p1.x;
  return true;
}

template <typename T>
__device__ __forceinline__ int met_class(T v);   // 0 empty, 1 occupied, 2 unknown
template <>
__device__ __forceinline__ int met_class<float>(float v) { return v == 1.0f ? 1 : (v == UNK_F ? 2 : 0); }
template <>
__device__ __forceinline__ int met_class<uint8_t>(uint8_t v) { return v == 1 ? 1 : (v == 255 ? 2 : 0); }  // byte(-1)

// counters[b] = {P, C, T} (int64).  Sparse part: one thread per predicted row.
template <typename T>
__global__ __launch_bounds__(256) void k_iou_sparse(const int64_t *__restrict__ locs, const uint8_t *__restrict__ keep,
                                                   const float *__restrict__ logits, int64_t lstride, int64_t m,
                                                   const T *__restrict__ tgt, DenseGeom g, int use_mask,
                                                   unsigned long long *__restrict__ counters) {
  __shared__ unsigned int sh[2 * MET_LDS_SAMPLES];
  const bool lds = g.nb <= MET_LDS_SAMPLES;
  if (lds) {
    for (int i = threadIdx.x; i < 2 * g.nb; i += 256) sh[i] = 0;
    __syncthreads();
  }
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < m; r += stride) {
    bool k = true;
    if (keep) k = keep[r] != 0;
    else if (logits) k = 1.0f / (1.0f + expf(-logits[r * lstride])) > 0.5f;   // train.py:279 Sigmoid()(x) > 0.5
    int64_t fl;
    int b;
    if (!k || !met_flat(locs, r, g, fl, b)) continue;
    const int cls = met_class<T>(tgt[fl]);
    if (use_mask && cls == 2) continue;
    if (lds) {
      atomicAdd(&sh[2 * b], 1u);
      if (cls == 1) atomicAdd(&sh[2 * b + 1], 1u);
    } else {
      atomicAdd(&counters[3 * b], 1ull);
      if (cls == 1) atomicAdd(&counters[3 * b + 1], 1ull);
    }
  }
  if (lds) {
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * g.nb; i += 256)
      if (sh[i]) atomicAdd(&counters[3 * (i >> 1) + (i & 1)], (unsigned long long)sh[i]);
  }
}

// dense part: blockIdx.y = sample, grid-stride over its voxels
template <typename T>
__global__ __launch_bounds__(256) void k_iou_dense(const T *__restrict__ tgt, int64_t vol,
                                                  unsigned long long *__restrict__ counters) {
  __shared__ unsigned int sh[256];
  const T *p = tgt + (int64_t)blockIdx.y * vol;
  unsigned int c = 0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < vol; i += stride) c += met_class<T>(p[i]) == 1;
  sh[threadIdx.x] = c;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0 && sh[0]) atomicAdd(&counters[3 * blockIdx.y + 2], (unsigned long long)sh[0]);
}

SGNN_EXPORT int sgnn_iou_counts(const int64_t *locs, const uint8_t *keep, const float *logits, int64_t lstride,
                                int64_t m, const void *tgt, int tgt_is_u8, int nb, int d0, int d1, int d2,
                                int use_mask, int64_t *counters, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(m >= 0 && nb >= 1 && d0 >= 1 && d1 >= 1 && d2 >= 1 && tgt && counters);
  SGNN_CHECK_ARG(m == 0 || locs);
  hipStream_t s = (hipStream_t)stream;
  SGNN_HIP_TRY(hipMemsetAsync(counters, 0, (size_t)nb * 3 * sizeof(int64_t), s));
  const DenseGeom g{nb, d0, d1, d2};
  const int64_t vol = (int64_t)d0 * d1 * d2;
  unsigned long long *c = (unsigned long long *)counters;
  const dim3 dgrid(sgnn_grid_for(vol, 256 * 8, 256), nb);
  if (tgt_is_u8) {
    if (m > 0)
      SGNN_LAUNCH((k_iou_sparse<uint8_t>), dim3(sgnn_grid_for(m, 256, 1024)), dim3(256), 0, s, locs, keep, logits,
                         lstride, m, (const uint8_t *)tgt, g, use_mask, c);
    SGNN_LAUNCH((k_iou_dense<uint8_t>), dgrid, dim3(256), 0, s, (const uint8_t *)tgt, vol, c);
  } else {
    if (m > 0)
      SGNN_LAUNCH((k_iou_sparse<float>), dim3(sgnn_grid_for(m, 256, 1024)), dim3(256), 0, s, locs, keep, logits,
                         lstride, m, (const float *)tgt, g, use_mask, c);
    SGNN_LAUNCH((k_iou_dense<float>), dgrid, dim3(256), 0, s, (const float *)tgt, vol, c);
  }
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// target-surface L1
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool l1_on_surface(float t, float truncation, float thresh) {
  return thresh >= 0.f ? (fabsf(t) <= thresh) : (fabsf(t) < truncation);   // loss.py:210-213
}

// partial[blk] = {sum, count}: blocks [0, nd) stream the dense target, blocks [nd, nd+ns) the predicted sites
__global__ __launch_bounds__(256) void k_l1_tgtsurf_partial(const int64_t *__restrict__ locs,
                                                           const float *__restrict__ vals, int64_t m,
                                                           const float *__restrict__ tgt,
                                                           const uint8_t *__restrict__ known, DenseGeom g,
                                                           float truncation, float thresh, int nd,
                                                           double *__restrict__ partial) {
  __shared__ double sh[2][256];
  double s = 0.0, c = 0.0;
  const float fill = -truncation;
  if ((int)blockIdx.x < nd) {
    const int64_t total = (int64_t)g.nb * g.d0 * g.d1 * g.d2;
    const int64_t stride = (int64_t)nd * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += stride) {
      const float t = tgt[i];
      if (!l1_on_surface(t, truncation, thresh)) continue;
      if (known && known[i] >= 2) continue;                      // loss.py:221-224, UNK_THRESH = 2
      s += (double)fabsf(fill - t);
      c += 1.0;
    }
  } else {
    const int64_t stride = (int64_t)(gridDim.x - nd) * 256;
    for (int64_t r = (int64_t)(blockIdx.x - nd) * 256 + threadIdx.x; r < m; r += stride) {
      int64_t fl;
      int b;
      if (!met_flat(locs, r, g, fl, b)) continue;
      const float t = tgt[fl];
      if (!l1_on_surface(t, truncation, thresh)) continue;
      if (known && known[fl] >= 2) continue;
      s += (double)fabsf(vals[r] - t) - (double)fabsf(fill - t);
    }
  }
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = c;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + d];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + d];
    }
    __syncthreads();
  }
  if (threadIdx.x < 2) partial[2 * blockIdx.x + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void k_l1_tgtsurf_final(const double *__restrict__ partial, int nblk,
                                                         double *__restrict__ out) {
  __shared__ double sh[2][256];
  double s = 0.0, c = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) {
    s += partial[2 * b];
    c += partial[2 * b + 1];
  }
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = c;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d) {
      sh[0][threadIdx.x] += sh[0][threadIdx.x + d];
      sh[1][threadIdx.x] += sh[1][threadIdx.x + d];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sh[0][0];
    out[1] = sh[1][0];
    out[2] = sh[0][0] / sh[1][0];      // 0/0 = nan, like torch.mean of an empty selection
  }
}

SGNN_EXPORT int64_t sgnn_l1_tgtsurf_ws_bytes(void) { return (int64_t)2 * MET_MAX_BLOCKS * 2 * sizeof(double); }

SGNN_EXPORT int sgnn_l1_tgtsurf(const int64_t *locs, const float *vals, int64_t m, const float *tgt_sdf,
                                const uint8_t *known, int nb, int d0, int d1, int d2, float truncation,
                                float thresh, double *out3, void *ws, int64_t ws_bytes, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(m >= 0 && nb >= 1 && d0 >= 1 && d1 >= 1 && d2 >= 1 && tgt_sdf && out3);
  SGNN_CHECK_ARG(m == 0 || (locs && vals));
  if (!ws || ws_bytes < sgnn_l1_tgtsurf_ws_bytes()) {
    sgnn_set_error("sgnn_l1_tgtsurf: workspace too small");
    return SGNN_ENOWS;
  }
  hipStream_t s = (hipStream_t)stream;
  const DenseGeom g{nb, d0, d1, d2};
  const int64_t total = (int64_t)nb * d0 * d1 * d2;
  const int nd = sgnn_grid_for(total, 256 * 8, MET_MAX_BLOCKS);
  const int ns = m > 0 ? sgnn_grid_for(m, 256 * 4, MET_MAX_BLOCKS) : 0;
  SGNN_LAUNCH(k_l1_tgtsurf_partial, dim3(nd + ns), dim3(256), 0, s, locs, vals, m, tgt_sdf, known, g, truncation,
                     thresh, nd, (double *)ws);
  SGNN_LAUNCH(k_l1_tgtsurf_final, dim3(1), dim3(256), 0, s, (const double *)ws, nd + ns, out3);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}
