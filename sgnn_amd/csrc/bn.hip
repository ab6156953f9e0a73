// scn.BatchNormReLU / BatchNormalization over the N active rows (torch/model.py:37,39,42,45,
// 181,187,256 and the FullyConvolutionalNet bodies).
//
// HBM-bound row passes over a row-major (N, C) slab:
//   stats  : one read  — per-thread fp32 partial sums over a short run of rows, flushed into
//            fp64 accumulators (keeps E[x^2]-E[x]^2 exact to fp32 resolution), workgroup
//            tree in LDS, per-workgroup partials reduced in fixed order (deterministic);
//   apply  : one read + one write, float4 when C % 4 == 0;
//   bwd    : reduce pass (dy, x read) + apply pass (dy, x read, dx written).
// Thread mapping keeps a thread on a fixed channel group: blockDim = RPB rows x CQ column
// groups, CQ = C/VEC, so per-channel constants live in registers.
// The per-channel finalisation (mean / invstd / running statistics, or the backward coefficients) runs in the
// LAST workgroup of the reduce pass to finish (ticket counter + __threadfence), summing the block partials in
// fixed order — same result whichever workgroup ends up last, and one launch less per pass.
#include "common.h"
#include <mutex>
#include <unordered_map>

#define BN_MAX_BLOCKS 512
#define BN_FLUSH 16

// one zero-initialised ticket word per stream (launches on one stream are ordered, so the last workgroup's reset
// to zero is visible to the next launch); 256 bytes of library-owned device memory per stream, never freed
static unsigned int *bn_ticket(hipStream_t s) {
  static std::mutex mu;
  static std::unordered_map<hipStream_t, unsigned int *> tickets;
  std::lock_guard<std::mutex> lock(mu);
  auto it = tickets.find(s);
  if (it != tickets.end()) return it->second;
  unsigned int *p = nullptr;
  if (hipMalloc((void **)&p, 256) != hipSuccess) return nullptr;
  if (hipMemset(p, 0, 256) != hipSuccess) return nullptr;
  tickets[s] = p;
  return p;
}

struct BnFinal {
  unsigned int *ticket;
  float eps, momentum;
  float *running_mean, *running_var, *save_mean, *save_invstd;   // MODE 0
  float *dgamma, *dbeta, *coef;                                   // MODE 1
};

struct BnGeom {
  int vec;   // 4 or 1 floats per thread-column
  int cq;    // column groups per row
  int rpb;   // rows per block iteration
};

static BnGeom bn_geom(int c) {
  BnGeom g;
  g.vec = (c % 4 == 0) ? 4 : 1;
  g.cq = c / g.vec;
  g.rpb = 256 / g.cq;
  if (g.rpb < 1) g.rpb = 1;
  return g;
}

static int bn_blocks(int64_t n, const BnGeom &g) {
  int64_t b = (n + (int64_t)g.rpb * BN_FLUSH - 1) / ((int64_t)g.rpb * BN_FLUSH);
  if (b < 1) b = 1;
  if (b > BN_MAX_BLOCKS) b = BN_MAX_BLOCKS;
  return (int)b;
}

static size_t bn_shbytes(const BnGeom &g, int c) {
  const size_t reduce = (size_t)g.rpb * g.cq * 2 * g.vec, fin = 256 + 2 * (size_t)c;
  return (reduce > fin ? reduce : fin) * sizeof(double);
}

SGNN_EXPORT int64_t sgnn_bn_ws_bytes(int64_t n, int c) {
  (void)n;
  return (int64_t)BN_MAX_BLOCKS * 2 * c * (int64_t)sizeof(double) + 4 * (int64_t)c * sizeof(float) + 256;
}

template <int VEC>
__device__ __forceinline__ void load_vec(const float *p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 t = *reinterpret_cast<const float4 *>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    v[0] = *p;
  }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float *p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    *p = v[0];
  }
}

// partial[blk][0][c] = sum_a, partial[blk][1][c] = sum_b over this block's rows where
//   MODE 0 (forward stats):  a = x,            b = x*x
//   MODE 1 (backward):       a = dz,           b = dz * xhat      (dz = dy masked by the ReLU/leak)
template <int VEC, int MODE>
__global__ __launch_bounds__(256) void k_bn_partial(const float *__restrict__ x, const float *__restrict__ dy,
                                                   int64_t n, int c, int cq, int rpb,
                                                   const float *__restrict__ mean,
                                                   const float *__restrict__ invstd,
                                                   const float *__restrict__ gamma,
                                                   const float *__restrict__ beta, float leak,
                                                   double *__restrict__ partial, BnFinal fin) {
  extern __shared__ double sh[];  // [rpb][2][c] would be large; reduce per column group instead
  const int tid = threadIdx.x;
  const int col = tid % cq, rloc = tid / cq;
  const bool active = rloc < rpb;
  double sa[VEC], sb[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) sa[v] = sb[v] = 0.0;
  float m_[VEC], is_[VEC], g_[VEC], b_[VEC];
  if (MODE == 1 && active) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      m_[v] = mean[col * VEC + v];
      is_[v] = invstd[col * VEC + v];
      g_[v] = gamma ? gamma[col * VEC + v] : 1.f;
      b_[v] = beta ? beta[col * VEC + v] : 0.f;
    }
  }
  if (active) {
    const int64_t step = (int64_t)gridDim.x * rpb;
    int64_t row = (int64_t)blockIdx.x * rpb + rloc;
    while (row < n) {
      float fa[VEC], fb[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) fa[v] = fb[v] = 0.f;
#pragma unroll 4
      for (int it = 0; it < BN_FLUSH && row < n; ++it, row += step) {
        float xv[VEC];
        load_vec<VEC>(x + row * c + col * VEC, xv);
        if (MODE == 0) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            fa[v] += xv[v];
            fb[v] = fmaf(xv[v], xv[v], fb[v]);
          }
        } else {
          float dv[VEC];
          load_vec<VEC>(dy + row * c + col * VEC, dv);
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float xh = (xv[v] - m_[v]) * is_[v];
            const float yv = fmaf(xh, g_[v], b_[v]);
            const float dz = yv > 0.f ? dv[v] : dv[v] * leak;
            fa[v] += dz;
            fb[v] = fmaf(dz, xh, fb[v]);
          }
        }
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        sa[v] += (double)fa[v];
        sb[v] += (double)fb[v];
      }
    }
  }
  // reduce over the rpb row-threads of each column group: sh[rloc][col][2*VEC]
  double *mine = sh + ((size_t)rloc * cq + col) * 2 * VEC;
  if (active) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      mine[v] = sa[v];
      mine[VEC + v] = sb[v];
    }
  }
  __syncthreads();
  // threads [0, cq*VEC*2) each own one output scalar and sum over rpb rows in order
  const int outs = cq * VEC * 2;
  for (int o = tid; o < outs; o += 256) {
    const int which = o / (cq * VEC);            // 0: a, 1: b
    const int ch = o % (cq * VEC);
    const int cg = ch / VEC, v = ch % VEC;
    double s = 0.0;
    for (int rr = 0; rr < rpb; ++rr) s += sh[((size_t)rr * cq + cg) * 2 * VEC + which * VEC + v];
    partial[((size_t)blockIdx.x * 2 + which) * c + ch] = s;
  }

  // ---- last workgroup to arrive finalises all channels ----
  __shared__ int is_last;
  __threadfence();
  __syncthreads();
  if (tid == 0) is_last = (atomicAdd(fin.ticket, 1u) == gridDim.x - 1);
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  const int nblk = gridDim.x;
  // scalar o = tid % spp (consecutive lanes read consecutive doubles: coalesced rows of the partial table),
  // `per` = 256/spp lanes share a scalar, each summing every per-th block in ascending order; their pieces are then
  // added in lane order — a fixed order, whichever workgroup happens to be last
  const int spp = outs < 256 ? outs : 256;            // scalars per pass
  const int per = 256 / spp;
  double *tot = sh + 256;                             // [2c]
  for (int base = 0; base < outs; base += spp) {
    const int o = base + tid % spp, piece = tid / spp;
    const bool live = o < outs && piece < per;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;    // four independent chains keep several loads in flight
    if (live) {
      int blk = piece;
      for (; blk + 3 * per < nblk; blk += 4 * per) {
        a0 += partial[(size_t)blk * outs + o];
        a1 += partial[(size_t)(blk + per) * outs + o];
        a2 += partial[(size_t)(blk + 2 * per) * outs + o];
        a3 += partial[(size_t)(blk + 3 * per) * outs + o];
      }
      for (; blk < nblk; blk += per) a0 += partial[(size_t)blk * outs + o];
    }
    sh[tid] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (live && piece == 0) {
      double t = 0.0;
      for (int j = 0; j < per; ++j) t += sh[j * spp + (tid % spp)];
      tot[o] = t;
    }
    __syncthreads();
  }
  for (int ch = tid; ch < c; ch += 256) {
    const double s1 = tot[ch], s2 = tot[c + ch];
    if (MODE == 0) {
      const double mu = s1 / (double)n;
      double var = s2 / (double)n - mu * mu;
      if (var < 0.0) var = 0.0;
      fin.save_mean[ch] = (float)mu;
      fin.save_invstd[ch] = (float)(1.0 / sqrt(var + (double)fin.eps));
      if (fin.running_mean) fin.running_mean[ch] = fin.momentum * fin.running_mean[ch] + (1.f - fin.momentum) * (float)mu;
      if (fin.running_var) {
        const double unb = var * ((double)n / (double)(n > 1 ? n - 1 : 1));
        fin.running_var[ch] = fin.momentum * fin.running_var[ch] + (1.f - fin.momentum) * (float)unb;
      }
    } else {
      if (fin.dbeta) fin.dbeta[ch] = (float)s1;
      if (fin.dgamma) fin.dgamma[ch] = (float)s2;
      fin.coef[ch] = (float)(s1 / (double)n);
      fin.coef[c + ch] = (float)(s2 / (double)n);
    }
  }
  if (tid == 0) *fin.ticket = 0u;
}

__global__ __launch_bounds__(256) void k_bn_eval_stats(const float *__restrict__ running_mean,
                                                      const float *__restrict__ running_var, int c, float eps,
                                                      float *__restrict__ save_mean,
                                                      float *__restrict__ save_invstd) {
  for (int ch = threadIdx.x; ch < c; ch += 256) {
    save_mean[ch] = running_mean[ch];
    save_invstd[ch] = 1.0f / sqrtf(running_var[ch] + eps);
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_bn_apply(const float *__restrict__ x, int64_t n, int c, int cq,
                                                 const float *__restrict__ mean,
                                                 const float *__restrict__ invstd,
                                                 const float *__restrict__ gamma,
                                                 const float *__restrict__ beta, float leak,
                                                 float *__restrict__ y) {
  // flat element-group index; channel group = g % cq
  const int64_t groups = n * cq;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += stride) {
    const int col = (int)(g % cq);
    float xv[VEC], yv[VEC];
    load_vec<VEC>(x + g * VEC, xv);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const int ch = col * VEC + v;
      const float xh = (xv[v] - mean[ch]) * invstd[ch];
      const float t = fmaf(xh, gamma ? gamma[ch] : 1.f, beta ? beta[ch] : 0.f);
      yv[v] = t > 0.f ? t : t * leak;
    }
    store_vec<VEC>(y + g * VEC, yv);
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float *__restrict__ x, const float *__restrict__ dy,
                                                     int64_t n, int c, int cq, const float *__restrict__ mean,
                                                     const float *__restrict__ invstd,
                                                     const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, float leak, int training,
                                                     const float *__restrict__ coef, float *__restrict__ dx) {
  const int64_t groups = n * cq;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += stride) {
    const int col = (int)(g % cq);
    float xv[VEC], dv[VEC], ov[VEC];
    load_vec<VEC>(x + g * VEC, xv);
    load_vec<VEC>(dy + g * VEC, dv);
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const int ch = col * VEC + v;
      const float gm = gamma ? gamma[ch] : 1.f;
      const float xh = (xv[v] - mean[ch]) * invstd[ch];
      const float t = fmaf(xh, gm, beta ? beta[ch] : 0.f);
      const float dz = t > 0.f ? dv[v] : dv[v] * leak;
      float d = dz;
      if (training) d = dz - coef[ch] - xh * coef[c + ch];
      ov[v] = d * gm * invstd[ch];
    }
    store_vec<VEC>(dx + g * VEC, ov);
  }
}

SGNN_EXPORT int sgnn_bn_fwd(const float *x, int64_t n, int c, const float *gamma, const float *beta,
                            float *running_mean, float *running_var, float eps, float momentum, int training,
                            float leak, float *save_mean, float *save_invstd, float *y, void *ws,
                            int64_t ws_bytes, sgnn_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  SGNN_CHECK_ARG(n >= 0 && c >= 1 && (c % 4 == 0 ? c <= 1024 : c <= 256) && save_mean && save_invstd);
  SGNN_CHECK_ARG(training || (running_mean && running_var));
  const BnGeom g = bn_geom(c);
  if (training && n > 0) {
    SGNN_CHECK_ARG(x);
    if (!ws || ws_bytes < sgnn_bn_ws_bytes(n, c)) {
      sgnn_set_error("sgnn_bn_fwd: workspace too small");
      return SGNN_ENOWS;
    }
    const int nblk = bn_blocks(n, g);
    const size_t shbytes = bn_shbytes(g, c);
    BnFinal fin{bn_ticket(s), eps, momentum, running_mean, running_var, save_mean, save_invstd, nullptr, nullptr, nullptr};
    SGNN_CHECK_ARG(fin.ticket != nullptr);
    if (g.vec == 4)
      hipLaunchKernelGGL((k_bn_partial<4, 0>), dim3(nblk), dim3(256), shbytes, s, x, nullptr, n, c, g.cq, g.rpb,
                         nullptr, nullptr, nullptr, nullptr, 0.f, (double *)ws, fin);
    else
      hipLaunchKernelGGL((k_bn_partial<1, 0>), dim3(nblk), dim3(256), shbytes, s, x, nullptr, n, c, g.cq, g.rpb,
                         nullptr, nullptr, nullptr, nullptr, 0.f, (double *)ws, fin);
  } else if (training) {  // empty batch: identity statistics, nothing to normalise
    SGNN_HIP_TRY(hipMemsetAsync(save_mean, 0, c * sizeof(float), s));
    SGNN_HIP_TRY(hipMemsetAsync(save_invstd, 0, c * sizeof(float), s));
  } else {
    hipLaunchKernelGGL(k_bn_eval_stats, dim3(1), dim3(256), 0, s, (const float *)running_mean,
                       (const float *)running_var, c, eps, save_mean, save_invstd);
  }
  if (n > 0) {
    SGNN_CHECK_ARG(x && y);
    const int grid = sgnn_grid_for(n * g.cq, 256, 2048);
    if (g.vec == 4)
      hipLaunchKernelGGL((k_bn_apply<4>), dim3(grid), dim3(256), 0, s, x, n, c, g.cq, (const float *)save_mean,
                         (const float *)save_invstd, gamma, beta, leak, y);
    else
      hipLaunchKernelGGL((k_bn_apply<1>), dim3(grid), dim3(256), 0, s, x, n, c, g.cq, (const float *)save_mean,
                         (const float *)save_invstd, gamma, beta, leak, y);
  }
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_bn_bwd(const float *x, const float *dy, int64_t n, int c, const float *gamma,
                            const float *beta, const float *save_mean, const float *save_invstd, int training,
                            float leak, float *dx, float *dgamma, float *dbeta, void *ws, int64_t ws_bytes,
                            sgnn_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  SGNN_CHECK_ARG(n >= 0 && c >= 1 && (c % 4 == 0 ? c <= 1024 : c <= 256) && save_mean && save_invstd);
  if (n == 0) {
    if (dgamma) SGNN_HIP_TRY(hipMemsetAsync(dgamma, 0, c * sizeof(float), s));
    if (dbeta) SGNN_HIP_TRY(hipMemsetAsync(dbeta, 0, c * sizeof(float), s));
    return SGNN_OK;
  }
  SGNN_CHECK_ARG(x && dy && dx);
  if (!ws || ws_bytes < sgnn_bn_ws_bytes(n, c)) {
    sgnn_set_error("sgnn_bn_bwd: workspace too small");
    return SGNN_ENOWS;
  }
  const BnGeom g = bn_geom(c);
  const int nblk = bn_blocks(n, g);
  const size_t shbytes = bn_shbytes(g, c);
  double *partial = (double *)ws;
  float *coef = (float *)((char *)ws + (size_t)BN_MAX_BLOCKS * 2 * c * sizeof(double));
  BnFinal fin{bn_ticket(s), 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, dgamma, dbeta, coef};
  SGNN_CHECK_ARG(fin.ticket != nullptr);
  if (g.vec == 4)
    hipLaunchKernelGGL((k_bn_partial<4, 1>), dim3(nblk), dim3(256), shbytes, s, x, dy, n, c, g.cq, g.rpb, save_mean,
                       save_invstd, gamma, beta, leak, partial, fin);
  else
    hipLaunchKernelGGL((k_bn_partial<1, 1>), dim3(nblk), dim3(256), shbytes, s, x, dy, n, c, g.cq, g.rpb, save_mean,
                       save_invstd, gamma, beta, leak, partial, fin);
  const int grid = sgnn_grid_for(n * g.cq, 256, 2048);
  if (g.vec == 4)
    hipLaunchKernelGGL((k_bn_bwd_apply<4>), dim3(grid), dim3(256), 0, s, x, dy, n, c, g.cq, save_mean, save_invstd,
                       gamma, beta, leak, training, (const float *)coef, dx);
  else
    hipLaunchKernelGGL((k_bn_bwd_apply<1>), dim3(grid), dim3(256), 0, s, x, dy, n, c, g.cq, save_mean, save_invstd,
                       gamma, beta, leak, training, (const float *)coef, dx);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}
