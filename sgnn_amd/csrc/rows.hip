// Row movement kernels (pure HBM-bound copies / short sums over fp32 feature rows).
//
// Serve scn.UnPooling, scn.JoinTable / AddTable, scn.SparseToDense, scn.OutputLayer-side
// compactions and the generative glue of torch/model.py:192-207 (8-child replication),
// :229-247 and :315-336 (mask compaction + concat) and :338-355 (skip join)
// (SURVEY.md §8 rows a5, a7, a8, a11-a14).  One thread per 4-byte or 16-byte element of the
// DESTINATION so that every store is coalesced; gathers read whole contiguous rows.
#include <initializer_list>
#include "common.h"

// vector width (floats) usable for rows of c floats
static inline int row_vec(int c) { return (c % 4 == 0) ? 4 : 1; }

template <int VEC>
struct VecT;
template <>
struct VecT<4> { typedef float4 T; };
template <>
struct VecT<1> { typedef float T; };

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::T vzero();
template <>
__device__ __forceinline__ float4 vzero<4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <>
__device__ __forceinline__ float vzero<1>() { return 0.f; }

__device__ __forceinline__ float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float vadd(float a, float b) { return a + b; }

#define ROWS_LAUNCH(KERNEL, c, total_groups, stream, ...)                                              \
  do {                                                                                                 \
    const int grid_ = sgnn_grid_for((total_groups), 256, 4096);                                        \
    if (row_vec(c) == 4)                                                                               \
      SGNN_LAUNCH((KERNEL<4>), dim3(grid_), dim3(256), 0, (hipStream_t)(stream), __VA_ARGS__);  \
    else                                                                                               \
      SGNN_LAUNCH((KERNEL<1>), dim3(grid_), dim3(256), 0, (hipStream_t)(stream), __VA_ARGS__);  \
  } while (0)

// ---------------------------------------------------------------------------
// fill / copy kernels (see common.h: memset / memcpy graph nodes stall a replayed step)
// ---------------------------------------------------------------------------
struct FillRegions {
  uint32_t *p[4];
  int64_t words[4];
  int64_t start[5];   // prefix sums of the 16-byte chunk counts
};

__global__ __launch_bounds__(256) void k_fill32_multi(FillRegions r, int n, uint32_t pattern) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < r.start[n]; g += stride) {
    int k = 0;
    while (k + 1 < n && g >= r.start[k + 1]) ++k;
    const int64_t c = g - r.start[k], w = c * 4;
    uint32_t *q = r.p[k] + w;
    if (w + 4 <= r.words[k] && (((uintptr_t)r.p[k]) & 15) == 0) {
      *reinterpret_cast<uint4 *>(q) = make_uint4(pattern, pattern, pattern, pattern);
    } else {
      for (int t = 0; t < 4 && w + t < r.words[k]; ++t) q[t] = pattern;
    }
  }
}

int sgnn_fill32_multi(void *const *p, const int64_t *words, int nregions, uint32_t pattern, hipStream_t s) {
  FillRegions r{};
  int n = 0;
  for (int k = 0; k < nregions && n < 4; ++k) {
    if (!p[k] || words[k] <= 0) continue;
    r.p[n] = (uint32_t *)p[k];
    r.words[n] = words[k];
    r.start[n + 1] = r.start[n] + (words[k] + 3) / 4;
    ++n;
  }
  if (n == 0) return SGNN_OK;
  SGNN_LAUNCH(k_fill32_multi, dim3(sgnn_grid_for(r.start[n], 256, 4096)), dim3(256), 0, s, r, n, pattern);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

int sgnn_fill32(void *p, uint32_t pattern, int64_t words, hipStream_t s) {
  void *ps[1] = {p};
  return sgnn_fill32_multi(ps, &words, 1, pattern, s);
}

__global__ __launch_bounds__(256) void k_copy_words(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src,
                                                   int64_t words, int vec) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  if (vec) {
    const int64_t n4 = words / 4;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n4; g += stride)
      reinterpret_cast<uint4 *>(dst)[g] = reinterpret_cast<const uint4 *>(src)[g];
    if (blockIdx.x == 0 && threadIdx.x < (words & 3)) dst[n4 * 4 + threadIdx.x] = src[n4 * 4 + threadIdx.x];
  } else {
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < words; g += stride) dst[g] = src[g];
  }
}

int sgnn_copy_words(void *dst, const void *src, int64_t words, hipStream_t s) {
  if (words <= 0 || dst == src) return SGNN_OK;
  const int vec = ((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0;
  SGNN_LAUNCH(k_copy_words, dim3(sgnn_grid_for(vec ? words / 4 + 1 : words, 256, 4096)), dim3(256), 0, s,
                     (uint32_t *)dst, (const uint32_t *)src, words, vec);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// Up to 8 device-to-device copies as ONE launch (train.GraphStep: a batch's seven tensors into the replayed graph's static
// input buffers — as separate copy_ calls they are seven 6 us kernels in front of every step).  16-byte units where source
// and destination are 16-byte aligned, bytes otherwise.
struct CopyRegions {
  uint8_t *d[8];
  const uint8_t *s[8];
  int64_t bytes[8];
  int64_t start[9];   // first 16-byte unit of region k in the launch's flat unit index
};

__global__ __launch_bounds__(256) void k_copy_multi(CopyRegions r, int n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < r.start[n]; g += stride) {
    int k = 0;
    while (k + 1 < n && g >= r.start[k + 1]) ++k;
    const int64_t b = (g - r.start[k]) * 16;
    uint8_t *q = r.d[k] + b;
    const uint8_t *p = r.s[k] + b;
    if (b + 16 <= r.bytes[k] && ((((uintptr_t)r.d[k]) | ((uintptr_t)r.s[k])) & 15) == 0) {
      *reinterpret_cast<uint4 *>(q) = *reinterpret_cast<const uint4 *>(p);
    } else {
      for (int t = 0; t < 16 && b + t < r.bytes[k]; ++t) q[t] = p[t];
    }
  }
}

SGNN_EXPORT int sgnn_copy_multi(void *const *dst, const void *const *src, const int64_t *bytes, int nregions,
                                sgnn_stream_t stream) {
  SGNN_CHECK_ARG(nregions >= 0 && nregions <= 8 && (nregions == 0 || (dst && src && bytes)));
  CopyRegions r{};
  int n = 0;
  for (int k = 0; k < nregions; ++k) {
    if (bytes[k] <= 0 || dst[k] == src[k]) continue;
    SGNN_CHECK_ARG(dst[k] && src[k]);
    r.d[n] = (uint8_t *)dst[k];
    r.s[n] = (const uint8_t *)src[k];
    r.bytes[n] = bytes[k];
    r.start[n + 1] = r.start[n] + (bytes[k] + 15) / 16;
    ++n;
  }
  if (n == 0) return SGNN_OK;
  SGNN_LAUNCH(k_copy_multi, dim3(sgnn_grid_for(r.start[n], 256, 8192)), dim3(256), 0, (hipStream_t)stream, r, n);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// dst[r] = src[idx[r]]  (idx < 0 -> zeros)
template <int VEC>
__global__ __launch_bounds__(256) void k_gather_rows(const float *__restrict__ src, int cq,
                                                    const int32_t *__restrict__ idx, int64_t m,
                                                    float *__restrict__ dst) {
  typedef typename VecT<VEC>::T T;
  const int64_t total = m * cq, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int64_t r = g / cq;
    const int col = (int)(g - r * cq);
    const int32_t i = idx[r];
    reinterpret_cast<T *>(dst)[g] = (i >= 0) ? reinterpret_cast<const T *>(src)[(int64_t)i * cq + col] : vzero<VEC>();
  }
}

SGNN_EXPORT int sgnn_gather_rows(const float *src, int c, const int32_t *idx, int64_t m, float *dst,
                                 sgnn_stream_t stream) {
  SGNN_CHECK_ARG(c >= 1 && m >= 0);
  if (m == 0) return SGNN_OK;
  SGNN_CHECK_ARG(src && idx && dst);
  const int cq = c / row_vec(c);
  ROWS_LAUNCH(k_gather_rows, c, m * cq, stream, src, cq, idx, m, dst);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// strided variants (rows that live in a column range of a wider buffer: JoinTable without a copy, prog.hip):
// ld_* = row strides in floats; VEC = 4 needs every stride and base pointer 16-byte aligned
template <int VEC>
__global__ __launch_bounds__(256) void k_gather_rows_ld(const float *__restrict__ src, int64_t ld_src, int cq,
                                                       const int32_t *__restrict__ idx, int64_t m,
                                                       float *__restrict__ dst, int64_t ld_dst, const int64_t *n_dev) {
  typedef typename VecT<VEC>::T T;
  m = sgnn_dyn_n(m, n_dev);
  const int64_t total = m * cq, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int64_t r = g / cq;
    const int col = (int)(g - r * cq);
    const int32_t i = idx[r];
    *reinterpret_cast<T *>(dst + r * ld_dst + col * VEC) =
        (i >= 0) ? *reinterpret_cast<const T *>(src + (int64_t)i * ld_src + col * VEC) : vzero<VEC>();
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_gather_sum_ld(const float *__restrict__ src, int64_t ld_src, int cq,
                                                      const int32_t *__restrict__ table, int64_t ld, int K,
                                                      int64_t n_out, float *__restrict__ dst, int64_t ld_dst,
                                                      const int64_t *n_dev) {
  typedef typename VecT<VEC>::T T;
  n_out = sgnn_dyn_n(n_out, n_dev);
  const int64_t total = n_out * cq, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int64_t r = g / cq;
    const int col = (int)(g - r * cq);
    T acc = vzero<VEC>();
    for (int k = 0; k < K; ++k) {
      const int32_t i = table[(int64_t)k * ld + r];
      if (i >= 0) acc = vadd(acc, *reinterpret_cast<const T *>(src + (int64_t)i * ld_src + col * VEC));
    }
    *reinterpret_cast<T *>(dst + r * ld_dst + col * VEC) = acc;
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_add_ld(const float *__restrict__ a, int64_t lda, const float *__restrict__ b,
                                               int64_t ldb, int64_t n, int cq, float *y, int64_t ldy,
                                               const int64_t *n_dev) {
  typedef typename VecT<VEC>::T T;
  n = sgnn_dyn_n(n, n_dev);
  const int64_t total = n * cq, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int64_t r = g / cq;
    const int col = (int)(g - r * cq);
    *reinterpret_cast<T *>(y + r * ldy + col * VEC) =
        vadd(*reinterpret_cast<const T *>(a + r * lda + col * VEC), *reinterpret_cast<const T *>(b + r * ldb + col * VEC));
  }
}

static inline bool ld_vec4(int c, std::initializer_list<int64_t> lds, std::initializer_list<const void *> ptrs) {
  if (c % 4) return false;
  for (int64_t l : lds)
    if (l % 4) return false;
  for (const void *p : ptrs)
    if ((uintptr_t)p & 15) return false;
  return true;
}

#define ROWS_LAUNCH_V(KERNEL, vec4, total_groups, stream, ...)                                         \
  do {                                                                                                 \
    const int grid_ = sgnn_grid_for((total_groups), 256, 4096);                                        \
    if (vec4)                                                                                          \
      SGNN_LAUNCH((KERNEL<4>), dim3(grid_), dim3(256), 0, (hipStream_t)(stream), __VA_ARGS__);  \
    else                                                                                               \
      SGNN_LAUNCH((KERNEL<1>), dim3(grid_), dim3(256), 0, (hipStream_t)(stream), __VA_ARGS__);  \
  } while (0)

int sgnn_gather_rows_ld(const float *src, int64_t ld_src, int c, const int32_t *idx, int64_t m, float *dst,
                        int64_t ld_dst, sgnn_stream_t stream, const int64_t *n_dev) {
  SGNN_CHECK_ARG(c >= 1 && m >= 0 && ld_src >= c && ld_dst >= c);
  if (m == 0) return SGNN_OK;
  SGNN_CHECK_ARG(src && idx && dst);
  const bool v4 = ld_vec4(c, {ld_src, ld_dst}, {src, dst});
  const int cq = v4 ? c / 4 : c;
  ROWS_LAUNCH_V(k_gather_rows_ld, v4, m * cq, stream, src, ld_src, cq, idx, m, dst, ld_dst, n_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

int sgnn_gather_sum_ld(const float *src, int64_t ld_src, int c, const int32_t *table, int64_t ld, int K, int64_t n_out,
                       float *dst, int64_t ld_dst, sgnn_stream_t stream, const int64_t *n_dev) {
  SGNN_CHECK_ARG(c >= 1 && n_out >= 0 && K >= 1 && ld >= n_out && ld_src >= c && ld_dst >= c);
  if (n_out == 0) return SGNN_OK;
  SGNN_CHECK_ARG(src && table && dst);
  const bool v4 = ld_vec4(c, {ld_src, ld_dst}, {src, dst});
  const int cq = v4 ? c / 4 : c;
  ROWS_LAUNCH_V(k_gather_sum_ld, v4, n_out * cq, stream, src, ld_src, cq, table, ld, K, n_out, dst, ld_dst, n_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

int sgnn_add_ld(const float *a, int64_t lda, const float *b, int64_t ldb, int64_t n, int c, float *y, int64_t ldy,
                sgnn_stream_t stream, const int64_t *n_dev) {
  SGNN_CHECK_ARG(c >= 1 && n >= 0 && lda >= c && ldb >= c && ldy >= c);
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(a && b && y);
  const bool v4 = ld_vec4(c, {lda, ldb, ldy}, {a, b, y});
  const int cq = v4 ? c / 4 : c;
  ROWS_LAUNCH_V(k_add_ld, v4, n * cq, stream, a, lda, b, ldb, n, cq, y, ldy, n_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// the same with the row count taken from device memory (launched for the host-known bound m_cap): lets a mask
// compaction's coordinates feed the next level's rulebook builders before the host has read the count
template <int VEC>
__global__ __launch_bounds__(256) void k_gather_rows_dn(const float *__restrict__ src, int cq,
                                                       const int32_t *__restrict__ idx, const int64_t *m_dev,
                                                       int64_t m_cap, float *__restrict__ dst) {
  typedef typename VecT<VEC>::T T;
  const int64_t total = sgnn_dyn_n(m_cap, m_dev) * cq, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int64_t r = g / cq;
    const int col = (int)(g - r * cq);
    const int32_t i = idx[r];
    reinterpret_cast<T *>(dst)[g] = (i >= 0) ? reinterpret_cast<const T *>(src)[(int64_t)i * cq + col] : vzero<VEC>();
  }
}

SGNN_EXPORT int sgnn_gather_rows_dn(const float *src, int c, const int32_t *idx, const int64_t *m_dev, int64_t m_cap,
                                    float *dst, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(c >= 1 && m_cap >= 0 && m_dev);
  if (m_cap == 0) return SGNN_OK;
  SGNN_CHECK_ARG(src && idx && dst);
  const int cq = c / row_vec(c);
  ROWS_LAUNCH(k_gather_rows_dn, c, m_cap * cq, stream, src, cq, idx, m_dev, m_cap, dst);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// dst[idx[r]] = src[r]
template <int VEC>
__global__ __launch_bounds__(256) void k_scatter_rows(const float *__restrict__ src, int cq,
                                                     const int32_t *__restrict__ idx, int64_t m,
                                                     float *__restrict__ dst, const int64_t *m_dev) {
  typedef typename VecT<VEC>::T T;
  m = sgnn_dyn_n(m, m_dev);
  const int64_t total = m * cq, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int64_t r = g / cq;
    const int col = (int)(g - r * cq);
    const int32_t i = idx[r];
    if (i >= 0) reinterpret_cast<T *>(dst)[(int64_t)i * cq + col] = reinterpret_cast<const T *>(src)[g];
  }
}

SGNN_EXPORT int sgnn_scatter_rows(const float *src, int c, const int32_t *idx, int64_t m, float *dst,
                                  int64_t n_dst, const int64_t *m_dev, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(c >= 1 && m >= 0 && n_dst >= 0);
  if (n_dst > 0) {
    SGNN_CHECK_ARG(dst);
    if (sgnn_fill32(dst, 0u, n_dst * (int64_t)c, (hipStream_t)stream) != SGNN_OK) return SGNN_EHIP;
  }
  if (m == 0 || n_dst == 0) return SGNN_OK;
  SGNN_CHECK_ARG(src && idx);
  const int cq = c / row_vec(c);
  ROWS_LAUNCH(k_scatter_rows, c, m * cq, stream, src, cq, idx, m, dst, m_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// dst[j] = sum_k src[table[k][j]]
template <int VEC>
__global__ __launch_bounds__(256) void k_gather_sum(const float *__restrict__ src, int cq,
                                                   const int32_t *__restrict__ table, int64_t ld, int K,
                                                   int64_t n_out, float *__restrict__ dst) {
  typedef typename VecT<VEC>::T T;
  const int64_t total = n_out * cq, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int64_t r = g / cq;
    const int col = (int)(g - r * cq);
    T acc = vzero<VEC>();
    for (int k = 0; k < K; ++k) {
      const int32_t i = table[(int64_t)k * ld + r];
      if (i >= 0) acc = vadd(acc, reinterpret_cast<const T *>(src)[(int64_t)i * cq + col]);
    }
    reinterpret_cast<T *>(dst)[g] = acc;
  }
}

SGNN_EXPORT int sgnn_gather_sum(const float *src, int c, const int32_t *table, int64_t ld, int K,
                                int64_t n_out, float *dst, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(c >= 1 && n_out >= 0 && K >= 1 && ld >= n_out);
  if (n_out == 0) return SGNN_OK;
  SGNN_CHECK_ARG(src && table && dst);
  const int cq = c / row_vec(c);
  ROWS_LAUNCH(k_gather_sum, c, n_out * cq, stream, src, cq, table, ld, K, n_out, dst);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

template <int VEC>
__global__ __launch_bounds__(256) void k_repeat_rows(const float *__restrict__ src, int cq, int64_t n, int rep,
                                                    float *__restrict__ dst) {
  typedef typename VecT<VEC>::T T;
  const int64_t total = n * rep * cq, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int64_t r = g / cq;
    const int col = (int)(g - r * cq);
    reinterpret_cast<T *>(dst)[g] = reinterpret_cast<const T *>(src)[(r / rep) * cq + col];
  }
}

SGNN_EXPORT int sgnn_repeat_rows(const float *src, int c, int64_t n, int rep, float *dst, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(c >= 1 && n >= 0 && rep >= 1);
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(src && dst);
  const int cq = c / row_vec(c);
  ROWS_LAUNCH(k_repeat_rows, c, n * rep * cq, stream, src, cq, n, rep, dst);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

template <int VEC>
__global__ __launch_bounds__(256) void k_sum_groups(const float *__restrict__ src, int cq, int64_t n, int rep,
                                                   float *__restrict__ dst, const int64_t *n_dev) {
  typedef typename VecT<VEC>::T T;
  n = sgnn_dyn_n(n, n_dev);
  const int64_t total = n * cq, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int64_t r = g / cq;
    const int col = (int)(g - r * cq);
    // four slices in flight (round 5: one load -> add per trip paid a memory round trip per slice — 12-22 us for the 8- and
    // 16-slice sums of the dense bottleneck's data gradients on a few thousand rows); the additions keep their order
    T acc = vzero<VEC>();
    const T *p = reinterpret_cast<const T *>(src) + r * rep * cq + col;
    int t = 0;
    for (; t + 4 <= rep; t += 4) {
      const T v0 = p[(int64_t)t * cq], v1 = p[(int64_t)(t + 1) * cq], v2 = p[(int64_t)(t + 2) * cq], v3 = p[(int64_t)(t + 3) * cq];
      acc = vadd(vadd(vadd(vadd(acc, v0), v1), v2), v3);
    }
    for (; t < rep; ++t) acc = vadd(acc, p[(int64_t)t * cq]);
    reinterpret_cast<T *>(dst)[g] = acc;
  }
}

int sgnn_sum_groups_dn(const float *src, int c, int64_t n, int rep, float *dst, sgnn_stream_t stream,
                       const int64_t *n_dev) {
  SGNN_CHECK_ARG(c >= 1 && n >= 0 && rep >= 1);
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(src && dst);
  const int cq = c / row_vec(c);
  ROWS_LAUNCH(k_sum_groups, c, n * cq, stream, src, cq, n, rep, dst, n_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_sum_groups(const float *src, int c, int64_t n, int rep, float *dst, sgnn_stream_t stream) {
  return sgnn_sum_groups_dn(src, c, n, rep, dst, stream, nullptr);
}

// dst[r] = [a[ia?ia[r]:r] | b[ib?ib[r]:r] (0 if ib[r] < 0)]   — scalar elements (ca, cb arbitrary)
__global__ __launch_bounds__(256) void k_concat_rows(const float *__restrict__ a, int ca,
                                                    const int32_t *__restrict__ ia,
                                                    const float *__restrict__ b, int cb,
                                                    const int32_t *__restrict__ ib, int64_t m,
                                                    float *__restrict__ dst, const int64_t *n_dev) {
  const int c = ca + cb;
  m = sgnn_dyn_n(m, n_dev);
  const int64_t total = m * c, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int64_t r = g / c;
    const int col = (int)(g - r * c);
    float v = 0.f;
    if (col < ca) {
      const int64_t i = ia ? (int64_t)ia[r] : r;
      if (i >= 0) v = a[i * ca + col];
    } else {
      const int64_t i = ib ? (int64_t)ib[r] : r;
      if (i >= 0) v = b[i * cb + (col - ca)];
    }
    dst[g] = v;
  }
}

SGNN_EXPORT int sgnn_concat_rows(const float *a, int ca, const int32_t *ia, const float *b, int cb,
                                 const int32_t *ib, int64_t m, float *dst, sgnn_stream_t stream) {
  return sgnn_concat_rows_dn(a, ca, ia, b, cb, ib, m, dst, stream, nullptr);
}

int sgnn_concat_rows_dn(const float *a, int ca, const int32_t *ia, const float *b, int cb, const int32_t *ib, int64_t m,
                        float *dst, sgnn_stream_t stream, const int64_t *n_dev) {
  SGNN_CHECK_ARG(ca >= 0 && cb >= 0 && ca + cb >= 1 && m >= 0);
  if (m == 0) return SGNN_OK;
  SGNN_CHECK_ARG(dst && (ca == 0 || a) && (cb == 0 || b));
  SGNN_LAUNCH(k_concat_rows, dim3(sgnn_grid_for(m * (ca + cb), 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, a, ca, ia, b, cb, ib, m, dst, n_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

__global__ __launch_bounds__(256) void k_concat_rows_bwd(const float *__restrict__ ddst, int ca,
                                                        const int32_t *__restrict__ ia, int cb,
                                                        const int32_t *__restrict__ ib, int64_t m,
                                                        float *__restrict__ da, float *__restrict__ db,
                                                        const int64_t *n_dev) {
  const int c = ca + cb;
  m = sgnn_dyn_n(m, n_dev);
  const int64_t total = m * c, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int64_t r = g / c;
    const int col = (int)(g - r * c);
    const float v = ddst[g];
    if (col < ca) {
      if (da) {
        const int64_t i = ia ? (int64_t)ia[r] : r;
        if (i >= 0) da[i * ca + col] = v;
      }
    } else if (db) {
      const int64_t i = ib ? (int64_t)ib[r] : r;
      if (i >= 0) db[i * cb + (col - ca)] = v;
    }
  }
}

SGNN_EXPORT int sgnn_concat_rows_bwd(const float *ddst, int ca, const int32_t *ia, int cb, const int32_t *ib,
                                     int64_t m, float *da, int64_t na, float *db, int64_t nb,
                                     sgnn_stream_t stream) {
  return sgnn_concat_rows_bwd_dn(ddst, ca, ia, cb, ib, m, da, na, db, nb, stream, nullptr);
}

int sgnn_concat_rows_bwd_dn(const float *ddst, int ca, const int32_t *ia, int cb, const int32_t *ib, int64_t m, float *da,
                            int64_t na, float *db, int64_t nb, sgnn_stream_t stream, const int64_t *n_dev) {
  hipStream_t s = (hipStream_t)stream;
  SGNN_CHECK_ARG(ca >= 0 && cb >= 0 && ca + cb >= 1 && m >= 0 && na >= 0 && nb >= 0);
  {
    void *zp[2] = {(da && ia && ca > 0) ? da : nullptr, (db && ib && cb > 0) ? db : nullptr};
    const int64_t zw[2] = {na * (int64_t)ca, nb * (int64_t)cb};
    if (sgnn_fill32_multi(zp, zw, 2, 0u, s) != SGNN_OK) return SGNN_EHIP;
  }
  if (m == 0) return SGNN_OK;
  SGNN_CHECK_ARG(ddst);
  SGNN_CHECK_ARG(ia || !da || na >= m);
  SGNN_CHECK_ARG(ib || !db || nb >= m);
  SGNN_LAUNCH(k_concat_rows_bwd, dim3(sgnn_grid_for(m * (ca + cb), 256, 4096)), dim3(256), 0, s, ddst, ca,
                     ia, cb, ib, m, da, db, n_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// dst[r] = [ a[ia?ia[r]:r] | b[ib?ib[r]:r] | c[ic?ic[r]:r] ]  (negative index -> zeros; a source with 0 channels is
// skipped).  One launch builds the input rows of a generative stage: kept features of the previous level, their
// occupancy/sdf logits, and the encoder's skip features at the same sites (torch/model.py:242, 330, 338-355).
struct Cat3 {
  const float *src[3];
  const int32_t *idx[3];
  int c[3];
};
struct Cat3Out {
  float *dst[3];
  const int32_t *idx[3];
  int c[3];
};

__global__ __launch_bounds__(256) void k_concat3(Cat3 s, int64_t m, float *__restrict__ dst, const int64_t *n_dev) {
  m = sgnn_dyn_n(m, n_dev);
  const int c01 = s.c[0] + s.c[1], c = c01 + s.c[2];
  const int64_t total = m * c, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int64_t r = g / c;
    const int col = (int)(g - r * c);
    const int which = col < s.c[0] ? 0 : (col < c01 ? 1 : 2);
    const int lc = col - (which == 0 ? 0 : (which == 1 ? s.c[0] : c01));
    const int64_t i = s.idx[which] ? (int64_t)s.idx[which][r] : r;
    dst[g] = i >= 0 ? s.src[which][i * s.c[which] + lc] : 0.f;
  }
}

__global__ __launch_bounds__(256) void k_concat3_bwd(const float *__restrict__ ddst, int64_t m, Cat3Out o,
                                                    const int64_t *n_dev) {
  m = sgnn_dyn_n(m, n_dev);
  const int c01 = o.c[0] + o.c[1], c = c01 + o.c[2];
  const int64_t total = m * c, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int64_t r = g / c;
    const int col = (int)(g - r * c);
    const int which = col < o.c[0] ? 0 : (col < c01 ? 1 : 2);
    if (!o.dst[which]) continue;
    const int lc = col - (which == 0 ? 0 : (which == 1 ? o.c[0] : c01));
    const int64_t i = o.idx[which] ? (int64_t)o.idx[which][r] : r;
    if (i >= 0) o.dst[which][i * o.c[which] + lc] = ddst[g];
  }
}

// Row-group form of the two kernels above (round 5): T threads own one destination row, four columns each — no 64-bit
// division per element, the three row indices loaded once per thread instead of once per element, 16 bytes per thread in
// and out.  (The element-per-thread form took 67 us for the 517 k x 26 input rows of the surface stage, 1.4 TB/s over
// bytes read + written, at the head of the stage's dependent chain.)  Same values: pure data movement.
template <int T>
__global__ __launch_bounds__(256) void k_concat3_rg(Cat3 s, int64_t m, float *__restrict__ dst, const int64_t *n_dev) {
  m = sgnn_dyn_n(m, n_dev);
  const int c01 = s.c[0] + s.c[1], c = c01 + s.c[2];
  const int t = threadIdx.x % T, col0 = 4 * t;
  const int64_t stride = (int64_t)gridDim.x * (256 / T);
  for (int64_t r = (int64_t)blockIdx.x * (256 / T) + threadIdx.x / T; r < m; r += stride) {
    if (col0 >= c) continue;
    int64_t i[3];
#pragma unroll
    for (int w = 0; w < 3; ++w) i[w] = (s.c[w] > 0 && s.idx[w]) ? (int64_t)s.idx[w][r] : r;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + j;
      const int which = col < s.c[0] ? 0 : (col < c01 ? 1 : 2);
      const int lc = col - (which == 0 ? 0 : (which == 1 ? s.c[0] : c01));
      const int64_t ii = which == 0 ? i[0] : (which == 1 ? i[1] : i[2]);
      v[j] = (col < c && ii >= 0) ? s.src[which][ii * s.c[which] + lc] : 0.f;
    }
    float *d = dst + r * c + col0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (col0 + j < c) d[j] = v[j];
  }
}

template <int T>
__global__ __launch_bounds__(256) void k_concat3_bwd_rg(const float *__restrict__ ddst, int64_t m, Cat3Out o,
                                                       const int64_t *n_dev) {
  m = sgnn_dyn_n(m, n_dev);
  const int c01 = o.c[0] + o.c[1], c = c01 + o.c[2];
  const int t = threadIdx.x % T, col0 = 4 * t;
  const int64_t stride = (int64_t)gridDim.x * (256 / T);
  for (int64_t r = (int64_t)blockIdx.x * (256 / T) + threadIdx.x / T; r < m; r += stride) {
    if (col0 >= c) continue;
    int64_t i[3];
#pragma unroll
    for (int w = 0; w < 3; ++w) i[w] = (o.c[w] > 0 && o.idx[w]) ? (int64_t)o.idx[w][r] : r;
    const float *g = ddst + r * c + col0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = col0 + j;
      if (col >= c) break;
      const int which = col < o.c[0] ? 0 : (col < c01 ? 1 : 2);
      const int lc = col - (which == 0 ? 0 : (which == 1 ? o.c[0] : c01));
      const int64_t ii = which == 0 ? i[0] : (which == 1 ? i[1] : i[2]);
      if (o.dst[which] && ii >= 0) o.dst[which][ii * o.c[which] + lc] = g[j];
    }
  }
}

SGNN_EXPORT int sgnn_concat3_rows(const float *a, int ca, const int32_t *ia, const float *b, int cb, const int32_t *ib,
                                  const float *c, int cc, const int32_t *ic, int64_t m, float *dst,
                                  sgnn_stream_t stream) {
  return sgnn_concat3_rows_dn(a, ca, ia, b, cb, ib, c, cc, ic, m, dst, stream, nullptr);
}

int sgnn_concat3_rows_dn(const float *a, int ca, const int32_t *ia, const float *b, int cb, const int32_t *ib,
                         const float *c, int cc, const int32_t *ic, int64_t m, float *dst, sgnn_stream_t stream,
                         const int64_t *n_dev) {
  SGNN_CHECK_ARG(ca >= 0 && cb >= 0 && cc >= 0 && ca + cb + cc >= 1 && m >= 0);
  if (m == 0) return SGNN_OK;
  SGNN_CHECK_ARG(dst && (ca == 0 || a) && (cb == 0 || b) && (cc == 0 || c));
  const Cat3 s{{a, b, c}, {ia, ib, ic}, {ca, cb, cc}};
  const int ctot = ca + cb + cc;
  if (ctot <= 32)
    SGNN_LAUNCH(k_concat3_rg<8>, dim3(sgnn_grid_for(m * 8, 256, 8192)), dim3(256), 0, (hipStream_t)stream, s, m, dst, n_dev);
  else if (ctot <= 64)
    SGNN_LAUNCH(k_concat3_rg<16>, dim3(sgnn_grid_for(m * 16, 256, 8192)), dim3(256), 0, (hipStream_t)stream, s, m, dst, n_dev);
  else
    SGNN_LAUNCH(k_concat3, dim3(sgnn_grid_for(m * ctot, 256, 4096)), dim3(256), 0, (hipStream_t)stream, s, m, dst, n_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// gradient of sgnn_concat3_rows: d{a,b,c}[idx[r]] = the matching columns of ddst[r] (indices unique).  A destination
// reached through an index array is zero-filled first (n{a,b,c} rows); NULL destinations are skipped.
SGNN_EXPORT int sgnn_concat3_rows_bwd(const float *ddst, int ca, const int32_t *ia, int cb, const int32_t *ib, int cc,
                                      const int32_t *ic, int64_t m, float *da, int64_t na, float *db, int64_t nb,
                                      float *dc, int64_t nc, sgnn_stream_t stream) {
  return sgnn_concat3_rows_bwd_dn(ddst, ca, ia, cb, ib, cc, ic, m, da, na, db, nb, dc, nc, stream, nullptr);
}

int sgnn_concat3_rows_bwd_dn(const float *ddst, int ca, const int32_t *ia, int cb, const int32_t *ib, int cc,
                             const int32_t *ic, int64_t m, float *da, int64_t na, float *db, int64_t nb, float *dc,
                             int64_t nc, sgnn_stream_t stream, const int64_t *n_dev) {
  hipStream_t s = (hipStream_t)stream;
  SGNN_CHECK_ARG(ca >= 0 && cb >= 0 && cc >= 0 && ca + cb + cc >= 1 && m >= 0 && na >= 0 && nb >= 0 && nc >= 0);
  {   // one launch zero-fills every destination that is reached through an index array
    void *zp[3] = {(da && ia && ca > 0) ? da : nullptr, (db && ib && cb > 0) ? db : nullptr, (dc && ic && cc > 0) ? dc : nullptr};
    const int64_t zw[3] = {na * (int64_t)ca, nb * (int64_t)cb, nc * (int64_t)cc};
    if (sgnn_fill32_multi(zp, zw, 3, 0u, s) != SGNN_OK) return SGNN_EHIP;
  }
  if (m == 0) return SGNN_OK;
  SGNN_CHECK_ARG(ddst);
  SGNN_CHECK_ARG((ia || !da || na >= m) && (ib || !db || nb >= m) && (ic || !dc || nc >= m));
  const Cat3Out o{{da, db, dc}, {ia, ib, ic}, {ca, cb, cc}};
  const int ctot = ca + cb + cc;
  if (ctot <= 32)
    SGNN_LAUNCH(k_concat3_bwd_rg<8>, dim3(sgnn_grid_for(m * 8, 256, 8192)), dim3(256), 0, s, ddst, m, o, n_dev);
  else if (ctot <= 64)
    SGNN_LAUNCH(k_concat3_bwd_rg<16>, dim3(sgnn_grid_for(m * 16, 256, 8192)), dim3(256), 0, s, ddst, m, o, n_dev);
  else
    SGNN_LAUNCH(k_concat3_bwd, dim3(sgnn_grid_for(m * ctot, 256, 4096)), dim3(256), 0, s, ddst, m, o, n_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

__global__ __launch_bounds__(256) void k_add(const float *__restrict__ a, const float *__restrict__ b,
                                            int64_t count, float *__restrict__ y) {
  const int64_t n4 = count / 4, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < n4; g += stride) {
    const float4 u = reinterpret_cast<const float4 *>(a)[g], v = reinterpret_cast<const float4 *>(b)[g];
    reinterpret_cast<float4 *>(y)[g] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
  if (blockIdx.x == 0) {
    const int64_t t = n4 * 4 + threadIdx.x;
    if (t < count) y[t] = a[t] + b[t];
  }
}

SGNN_EXPORT int sgnn_add(const float *a, const float *b, int64_t count, float *y, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(count >= 0);
  if (count == 0) return SGNN_OK;
  SGNN_CHECK_ARG(a && b && y);
  SGNN_LAUNCH(k_add, dim3(sgnn_grid_for(count / 4 + 1, 256, 4096)), dim3(256), 0, (hipStream_t)stream, a, b,
                     count, y);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// dense (B, C, d0, d1, d2) <-> rows.  Thread per (row, channel); channel-major over the row
// index so that consecutive lanes touch consecutive x of the same channel plane when the sites
// are raster ordered.
__global__ __launch_bounds__(256) void k_sparse_to_dense(const float *__restrict__ feats,
                                                        const int4 *__restrict__ coords, int64_t n, int c,
                                                        float *__restrict__ dense, int batch, int d0, int d1,
                                                        int d2) {
  const int64_t total = n * c, stride = (int64_t)gridDim.x * 256;
  const int64_t vol = (int64_t)d0 * d1 * d2;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int ch = (int)(g / n);
    const int64_t r = g - (int64_t)ch * n;
    const int4 p = coords[r];
    if ((unsigned)p.x < (unsigned)d0 && (unsigned)p.y < (unsigned)d1 && (unsigned)p.z < (unsigned)d2 &&
        (unsigned)p.w < (unsigned)batch)
      dense[((int64_t)p.w * c + ch) * vol + ((int64_t)p.x * d1 + p.y) * d2 + p.z] = feats[r * c + ch];
  }
}

SGNN_EXPORT int sgnn_sparse_to_dense(const float *feats, const int32_t *coords, int64_t n, int c, float *dense,
                                     int batch, int d0, int d1, int d2, sgnn_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  SGNN_CHECK_ARG(n >= 0 && c >= 1 && batch >= 0 && d0 >= 0 && d1 >= 0 && d2 >= 0);
  const int64_t total = (int64_t)batch * c * d0 * d1 * d2;
  if (total > 0) {
    SGNN_CHECK_ARG(dense);
    if (sgnn_fill32(dense, 0u, total, s) != SGNN_OK) return SGNN_EHIP;
  }
  if (n == 0 || total == 0) return SGNN_OK;
  SGNN_CHECK_ARG(feats && coords);
  SGNN_LAUNCH(k_sparse_to_dense, dim3(sgnn_grid_for(n * c, 256, 4096)), dim3(256), 0, s, feats,
                     (const int4 *)coords, n, c, dense, batch, d0, d1, d2);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

__global__ __launch_bounds__(256) void k_dense_to_sparse(const float *__restrict__ dense,
                                                        const int4 *__restrict__ coords, int64_t n, int c,
                                                        float *__restrict__ feats, int batch, int d0, int d1,
                                                        int d2) {
  const int64_t total = n * c, stride = (int64_t)gridDim.x * 256;
  const int64_t vol = (int64_t)d0 * d1 * d2;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) {
    const int ch = (int)(g / n);
    const int64_t r = g - (int64_t)ch * n;
    const int4 p = coords[r];
    float v = 0.f;
    if ((unsigned)p.x < (unsigned)d0 && (unsigned)p.y < (unsigned)d1 && (unsigned)p.z < (unsigned)d2 &&
        (unsigned)p.w < (unsigned)batch)
      v = dense[((int64_t)p.w * c + ch) * vol + ((int64_t)p.x * d1 + p.y) * d2 + p.z];
    feats[r * c + ch] = v;
  }
}

SGNN_EXPORT int sgnn_dense_to_sparse(const float *dense, const int32_t *coords, int64_t n, int c, float *feats,
                                     int batch, int d0, int d1, int d2, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && c >= 1 && batch >= 0 && d0 >= 0 && d1 >= 0 && d2 >= 0);
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(dense && coords && feats);
  SGNN_LAUNCH(k_dense_to_sparse, dim3(sgnn_grid_for(n * c, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     dense, (const int4 *)coords, n, c, feats, batch, d0, d1, d2);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}
