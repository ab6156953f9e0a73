// Per-site linear heads: y[r] = W x[r] + b with a handful of outputs (occupancy / sdf logits).
//
// Serves nn.Linear(nf,1) x2 of Refinement (torch/model.py:190-191, 230-231) fused into one (nf -> 2)
// product, and SurfacePrediction.linear (nf*3 -> 1, torch/model.py:258, 271).  With 1-2 outputs per
// site this is a pure HBM stream over the feature slab (vendor GEMV kernels took ~1 ms per call on
// 8e5 rows, profiles/r01a_*); one thread per site row, weights in LDS, fp64 block partials for the
// deterministic weight/bias gradient reduction.
#include "common.h"

#define LIN_MAX_OUT 4
#define LIN_MAX_BLOCKS 512

template <int CIN, int VEC>
__device__ __forceinline__ void load_row(const float *__restrict__ p, float (&v)[CIN]) {
  if constexpr (VEC == 4) {
#pragma unroll
    for (int t = 0; t < CIN / 4; ++t) {
      const float4 q = reinterpret_cast<const float4 *>(p)[t];
      v[4 * t] = q.x; v[4 * t + 1] = q.y; v[4 * t + 2] = q.z; v[4 * t + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int t = 0; t < CIN; ++t) v[t] = p[t];
  }
}

// weight row o lives at w[o] (CIN floats), its bias at b[o] (one float, may be NULL): the two heads of a Refinement are
// separate nn.Linear modules (torch/model.py:190-191) and need not be packed first
struct LinW {
  const float *w[LIN_MAX_OUT];
  const float *b[LIN_MAX_OUT];
};
struct LinG {
  float *dw[LIN_MAX_OUT];
  float *db[LIN_MAX_OUT];
};

template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_linear_fwd(const float *__restrict__ x, int64_t n, LinW p,
                                                   float *__restrict__ y, const int64_t *n_dev) {
  __shared__ float ws[COUT * CIN + COUT];
  n = sgnn_dyn_n(n, n_dev);
  for (int e = threadIdx.x; e < COUT * CIN; e += 256) ws[e] = p.w[e / CIN][e % CIN];
  if (threadIdx.x < COUT) ws[COUT * CIN + threadIdx.x] = p.b[threadIdx.x] ? p.b[threadIdx.x][0] : 0.f;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
    float v[CIN];
    load_row<CIN, (CIN % 4 == 0) ? 4 : 1>(x + r * CIN, v);
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      float acc = ws[COUT * CIN + o];
#pragma unroll
      for (int c = 0; c < CIN; ++c) acc = fmaf(v[c], ws[o * CIN + c], acc);
      y[r * COUT + o] = acc;
    }
  }
}

// dx[r][c] = sum_o dy[r][o] w[o][c]; block partials of dW[o][c] = sum_r dy[r][o] x[r][c], db[o] = sum_r dy[r][o]
// addend (may be dx itself): a gradient the input rows already carry — dx = dy w + addend in the same pass instead of an
// add launch over the level behind this one (row stride ld_add floats, a multiple of 4 when CIN % 4 == 0)
template <int CIN, int COUT>
__global__ __launch_bounds__(256) void k_linear_bwd(const float *__restrict__ x, const float *__restrict__ dy,
                                                   int64_t n, LinW p, float *dx, double *__restrict__ partial,
                                                   const int64_t *n_dev, const float *addend, int64_t ld_add) {
  constexpr int NV = COUT * CIN + COUT;
  n = sgnn_dyn_n(n, n_dev);
  __shared__ float ws[COUT * CIN];
  __shared__ float red[4][NV];
  for (int e = threadIdx.x; e < COUT * CIN; e += 256) ws[e] = p.w[e / CIN][e % CIN];
  __syncthreads();
  float acc[NV];
#pragma unroll
  for (int e = 0; e < NV; ++e) acc[e] = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += stride) {
    float v[CIN], g[COUT];
    load_row<CIN, (CIN % 4 == 0) ? 4 : 1>(x + r * CIN, v);
#pragma unroll
    for (int o = 0; o < COUT; ++o) g[o] = dy[r * COUT + o];
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
#pragma unroll
      for (int c = 0; c < CIN; ++c) acc[o * CIN + c] = fmaf(g[o], v[c], acc[o * CIN + c]);
      acc[COUT * CIN + o] += g[o];
    }
    if (dx) {
      float d[CIN];
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        float t = 0.f;
#pragma unroll
        for (int o = 0; o < COUT; ++o) t = fmaf(g[o], ws[o * CIN + c], t);
        d[c] = t;
      }
      if (addend) {       // uniform over the launch
        float a[CIN];
        load_row<CIN, (CIN % 4 == 0) ? 4 : 1>(addend + r * ld_add, a);
#pragma unroll
        for (int c = 0; c < CIN; ++c) d[c] += a[c];
      }
      if constexpr (CIN % 4 == 0) {
#pragma unroll
        for (int t = 0; t < CIN / 4; ++t)
          reinterpret_cast<float4 *>(dx + r * CIN)[t] = make_float4(d[4 * t], d[4 * t + 1], d[4 * t + 2], d[4 * t + 3]);
      } else {
#pragma unroll
        for (int c = 0; c < CIN; ++c) dx[r * CIN + c] = d[c];
      }
    }
  }
  // wave reduction (fixed butterfly order), then the four waves in order
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    float t = acc[e];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d);
    if (lane == 0) red[wave][e] = t;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < NV; e += 256)
    partial[(size_t)blockIdx.x * NV + e] = (double)red[0][e] + (double)red[1][e] + (double)red[2][e] + (double)red[3][e];
}

__global__ __launch_bounds__(256) void k_linear_finalize(const double *__restrict__ partial, int nblk, int nv,
                                                        int ncw, int cin, LinG g) {
  __shared__ double sh[256];
  const int e = blockIdx.x;
  double a = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) a += partial[(size_t)b * nv + e];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (e < ncw) {
      if (g.dw[e / cin]) g.dw[e / cin][e % cin] = (float)sh[0];
    } else if (g.db[e - ncw]) {
      g.db[e - ncw][0] = (float)sh[0];
    }
  }
}

// one row per thread up to LIN_MAX_BLOCKS workgroups (a thread's rows are a serial load -> FMA -> store chain: with four rows per
// thread the heads of a 16 k-row level ran 13-14 us on 16 CUs), several rows per thread only above 131 k rows
static int lin_blocks(int64_t n) {
  int64_t b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > LIN_MAX_BLOCKS) b = LIN_MAX_BLOCKS;
  return (int)b;
}

SGNN_EXPORT int64_t sgnn_linear_ws_bytes(int64_t n, int cin, int cout) {
  (void)n;
  return (int64_t)LIN_MAX_BLOCKS * (cin * cout + cout) * (int64_t)sizeof(double) + 256;
}

#define LIN_CASES(X) X(16, 1) X(16, 2) X(48, 1) X(48, 2) X(8, 1) X(8, 2) X(32, 1) X(32, 2) X(12, 2) X(4, 1)

int sgnn_linear_fwd_rows(const float *x, int64_t n, int cin, const float *const *w, const float *const *b, int cout,
                         float *y, sgnn_stream_t stream, const int64_t *n_dev) {
  SGNN_CHECK_ARG(n >= 0 && cin >= 1 && cout >= 1 && cout <= LIN_MAX_OUT);
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(x && w && y);
  LinW p{};
  for (int o = 0; o < cout; ++o) {
    SGNN_CHECK_ARG(w[o]);
    p.w[o] = w[o];
    p.b[o] = b ? b[o] : nullptr;
  }
  hipStream_t s = (hipStream_t)stream;
  const int grid = sgnn_grid_for(n, 256, 2048);
  bool done = false;
#define X(CI, CO)                                                                                   \
  if (!done && cin == CI && cout == CO) {                                                           \
    SGNN_LAUNCH((k_linear_fwd<CI, CO>), dim3(grid), dim3(256), 0, s, x, n, p, y, n_dev);     \
    done = true;                                                                                    \
  }
  LIN_CASES(X)
#undef X
  if (!done) {
    sgnn_set_error("sgnn_linear_fwd: unsupported head shape %d -> %d", cin, cout);
    return SGNN_EINVAL;
  }
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_linear_fwd(const float *x, int64_t n, int cin, const float *w, const float *bias, int cout,
                                float *y, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(cout >= 1 && cout <= LIN_MAX_OUT && cin >= 1);
  const float *wr[LIN_MAX_OUT] = {}, *br[LIN_MAX_OUT] = {};
  for (int o = 0; o < cout; ++o) {
    wr[o] = w ? w + (size_t)o * cin : nullptr;
    br[o] = bias ? bias + o : nullptr;
  }
  return sgnn_linear_fwd_rows(x, n, cin, wr, br, cout, y, stream);
}

int sgnn_linear_bwd_rows(const float *x, const float *dy, int64_t n, int cin, const float *const *w, int cout,
                         float *dx, float *const *dw, float *const *db, void *ws, int64_t ws_bytes,
                         sgnn_stream_t stream, const int64_t *n_dev, const float *addend, int64_t ld_add) {
  hipStream_t s = (hipStream_t)stream;
  SGNN_CHECK_ARG(n >= 0 && cin >= 1 && cout >= 1 && cout <= LIN_MAX_OUT && w);
  SGNN_CHECK_ARG(!addend || (dx && ld_add >= cin && (cin % 4 != 0 || (ld_add % 4 == 0 && ((uintptr_t)addend & 15) == 0))));
  if (n == 0) {
    for (int o = 0; o < cout; ++o) {
      if (dw && dw[o]) SGNN_HIP_TRY(hipMemsetAsync(dw[o], 0, (size_t)cin * sizeof(float), s));
      if (db && db[o]) SGNN_HIP_TRY(hipMemsetAsync(db[o], 0, sizeof(float), s));
    }
    return SGNN_OK;
  }
  SGNN_CHECK_ARG(x && dy);
  if (!ws || ws_bytes < sgnn_linear_ws_bytes(n, cin, cout)) {
    sgnn_set_error("sgnn_linear_bwd: workspace too small");
    return SGNN_ENOWS;
  }
  LinW p{};
  LinG g{};
  for (int o = 0; o < cout; ++o) {
    SGNN_CHECK_ARG(w[o]);
    p.w[o] = w[o];
    g.dw[o] = dw ? dw[o] : nullptr;
    g.db[o] = db ? db[o] : nullptr;
  }
  const int nblk = lin_blocks(n);
  const int nv = cin * cout + cout;
  bool done = false;
#define X(CI, CO)                                                                                          \
  if (!done && cin == CI && cout == CO) {                                                                  \
    SGNN_LAUNCH((k_linear_bwd<CI, CO>), dim3(nblk), dim3(256), 0, s, x, dy, n, p, dx, (double *)ws, n_dev, addend, ld_add); \
    done = true;                                                                                           \
  }
  LIN_CASES(X)
#undef X
  if (!done) {
    sgnn_set_error("sgnn_linear_bwd: unsupported head shape %d -> %d", cin, cout);
    return SGNN_EINVAL;
  }
  SGNN_LAUNCH(k_linear_finalize, dim3(nv), dim3(256), 0, s, (const double *)ws, nblk, nv, cin * cout, cin, g);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_linear_bwd(const float *x, const float *dy, int64_t n, int cin, const float *w, int cout,
                                float *dx, float *dw, float *dbias, void *ws, int64_t ws_bytes,
                                sgnn_stream_t stream) {
  SGNN_CHECK_ARG(cout >= 1 && cout <= LIN_MAX_OUT && cin >= 1 && w);
  const float *wr[LIN_MAX_OUT] = {};
  float *dwr[LIN_MAX_OUT] = {}, *dbr[LIN_MAX_OUT] = {};
  for (int o = 0; o < cout; ++o) {
    wr[o] = w + (size_t)o * cin;
    dwr[o] = dw ? dw + (size_t)o * cin : nullptr;
    dbr[o] = dbias ? dbias + o : nullptr;
  }
  return sgnn_linear_bwd_rows(x, dy, n, cin, wr, cout, dx, dwr, dbr, ws, ws_bytes, stream);
}
