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
#include "common.h"

#define BN_MAX_BLOCKS 2048
#define BN_FLUSH 2

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
                                                   double *__restrict__ partial) {
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
      // BN_FLUSH rows per pass: all their loads are issued first (row index clamped, contribution masked), so several
      // rows are in flight per thread instead of one load -> wait -> add chain
      float xv[BN_FLUSH][VEC], dv[BN_FLUSH][VEC];
      bool keep[BN_FLUSH];
#pragma unroll
      for (int it = 0; it < BN_FLUSH; ++it) {
        const int64_t rr = row + it * step;
        keep[it] = rr < n;
        const int64_t rc = rr < n ? rr : row;
        load_vec<VEC>(x + rc * c + col * VEC, xv[it]);
        if (MODE == 1) load_vec<VEC>(dy + rc * c + col * VEC, dv[it]);
      }
#pragma unroll
      for (int it = 0; it < BN_FLUSH; ++it) {
        if (MODE == 0) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float xm = keep[it] ? xv[it][v] : 0.f;
            fa[v] += xm;
            fb[v] = fmaf(xm, xm, fb[v]);
          }
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            const float xh = (xv[it][v] - m_[v]) * is_[v];
            const float yv = fmaf(xh, g_[v], b_[v]);
            const float dz = keep[it] ? (yv > 0.f ? dv[it][v] : dv[it][v] * leak) : 0.f;
            fa[v] += dz;
            fb[v] = fmaf(dz, xh, fb[v]);
          }
        }
      }
      row += (int64_t)BN_FLUSH * step;
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
}

// fixed-order tree sum of the block partials of one channel: one workgroup per channel, 256 lanes
__device__ __forceinline__ void bn_reduce_channel(const double *__restrict__ partial, int nblk, int c, int ch,
                                                  double *sh, double &s, double &s2) {
  double a = 0.0, b = 0.0;
  for (int blk = threadIdx.x; blk < nblk; blk += 256) {
    a += partial[((size_t)blk * 2 + 0) * c + ch];
    b += partial[((size_t)blk * 2 + 1) * c + ch];
  }
  sh[threadIdx.x] = a;
  sh[256 + threadIdx.x] = b;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if (threadIdx.x < d) {
      sh[threadIdx.x] += sh[threadIdx.x + d];
      sh[256 + threadIdx.x] += sh[256 + threadIdx.x + d];
    }
    __syncthreads();
  }
  s = sh[0];
  s2 = sh[256];
}

__global__ __launch_bounds__(256) void k_bn_finalize_fwd(const double *__restrict__ partial, int nblk,
                                                        int64_t n, int c, float eps, float momentum,
                                                        float *__restrict__ running_mean,
                                                        float *__restrict__ running_var,
                                                        float *__restrict__ save_mean,
                                                        float *__restrict__ save_invstd) {
  __shared__ double sh[512];
  const int ch = blockIdx.x;
  double s, s2;
  bn_reduce_channel(partial, nblk, c, ch, sh, s, s2);
  if (threadIdx.x == 0) {
    const double mean = s / (double)n;
    double var = s2 / (double)n - mean * mean;
    if (var < 0.0) var = 0.0;
    save_mean[ch] = (float)mean;
    save_invstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[ch] = momentum * running_mean[ch] + (1.f - momentum) * (float)mean;
    if (running_var) {
      const double unb = var * ((double)n / (double)(n > 1 ? n - 1 : 1));
      running_var[ch] = momentum * running_var[ch] + (1.f - momentum) * (float)unb;
    }
  }
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

// Small levels (few block partials, C <= BN_FUSE_MAXC): the apply kernels finalise the statistics themselves — every
// workgroup sums the short partial table in the same fixed order into LDS, workgroup 0 also stores the results the
// later passes / the caller need — and the separate finalize launch disappears.
#define BN_LDS_MAXC 256   // per-channel constants are staged in LDS up to this many channels
#define BN_U 2            // row groups a thread loads before it uses the first one
#define BN_FUSE_BLOCKS 128
#define BN_FUSE_MAXC 64

struct BnFuse {
  const double *partial;   // NULL: statistics come from the mean/invstd (or coef) arrays
  int nblk;
  float eps, momentum;
  float *running_mean, *running_var, *save_mean, *save_invstd;   // forward
  float *dgamma, *dbeta;                                          // backward
};

// Per-channel totals of the block partials, computed by the whole workgroup: 256/c threads share a channel (strided
// partial sums), their pieces are added in thread order — a fixed order, identical in every workgroup.
// tot[0..c) = sum_a, tot[c..2c) = sum_b;  scratch: 512 doubles.  Contains two barriers.
__device__ __forceinline__ void bn_fuse_totals(const BnFuse &f, int c, double *scratch, double *tot) {
  const int tid = threadIdx.x;
  const int tpc = 256 / c;                       // >= 4 for c <= BN_FUSE_MAXC
  const int ch = tid % c, part = tid / c;
  double a = 0.0, b = 0.0;
  if (part < tpc)
    for (int blk = part; blk < f.nblk; blk += tpc) {
      a += f.partial[((size_t)blk * 2 + 0) * c + ch];
      b += f.partial[((size_t)blk * 2 + 1) * c + ch];
    }
  scratch[tid] = a;
  scratch[256 + tid] = b;
  __syncthreads();
  if (tid < c) {
    double s1 = 0.0, s2 = 0.0;
    for (int p = 0; p < tpc; ++p) {
      s1 += scratch[p * c + tid];
      s2 += scratch[256 + p * c + tid];
    }
    tot[tid] = s1;
    tot[c + tid] = s2;
  }
  __syncthreads();
}

template <int VEC>
__global__ __launch_bounds__(256) void k_bn_apply(const float *__restrict__ x, int64_t n, int c, int cq,
                                                 const float *__restrict__ mean,
                                                 const float *__restrict__ invstd,
                                                 const float *__restrict__ gamma,
                                                 const float *__restrict__ beta, float leak,
                                                 float *__restrict__ y, BnFuse fuse) {
  __shared__ float s_mean[BN_FUSE_MAXC], s_inv[BN_FUSE_MAXC];
  __shared__ double s_scratch[512], s_tot[2 * BN_FUSE_MAXC];
  if (fuse.partial) {
    bn_fuse_totals(fuse, c, s_scratch, s_tot);
    for (int ch = threadIdx.x; ch < c; ch += 256) {
      const double s1 = s_tot[ch], s2 = s_tot[c + ch];
      const double mu = s1 / (double)n;
      double var = s2 / (double)n - mu * mu;
      if (var < 0.0) var = 0.0;
      const float fm = (float)mu, fi = (float)(1.0 / sqrt(var + (double)fuse.eps));
      s_mean[ch] = fm;
      s_inv[ch] = fi;
      if (blockIdx.x == 0) {
        fuse.save_mean[ch] = fm;
        fuse.save_invstd[ch] = fi;
        if (fuse.running_mean) fuse.running_mean[ch] = fuse.momentum * fuse.running_mean[ch] + (1.f - fuse.momentum) * fm;
        if (fuse.running_var) {
          const double unb = var * ((double)n / (double)(n > 1 ? n - 1 : 1));
          fuse.running_var[ch] = fuse.momentum * fuse.running_var[ch] + (1.f - fuse.momentum) * (float)unb;
        }
      }
    }
    __syncthreads();
    mean = s_mean;
    invstd = s_inv;
  }
  // per-channel constants live in LDS (a per-lane global load per constant per element made the texture addresser, not
  // HBM, the limit of this kernel); BN_U independent row loads are in flight per thread before the first use
  __shared__ __attribute__((aligned(16))) float s_k[4 * BN_LDS_MAXC];
  const bool lds_k = c <= BN_LDS_MAXC;
  if (lds_k) {
    for (int ch = threadIdx.x; ch < c; ch += 256) {
      s_k[ch] = mean[ch];
      s_k[BN_LDS_MAXC + ch] = invstd[ch];
      s_k[2 * BN_LDS_MAXC + ch] = gamma ? gamma[ch] : 1.f;
      s_k[3 * BN_LDS_MAXC + ch] = beta ? beta[ch] : 0.f;
    }
    __syncthreads();
  }
  auto K_ = [&](int which, int ch) -> float {
    if (lds_k) return s_k[which * BN_LDS_MAXC + ch];
    return which == 0 ? mean[ch] : which == 1 ? invstd[ch] : which == 2 ? (gamma ? gamma[ch] : 1.f) : (beta ? beta[ch] : 0.f);
  };
  // flat element-group index; channel group = g % cq
  const int64_t groups = n * cq;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t g0 = (int64_t)blockIdx.x * 256 + threadIdx.x; g0 < groups; g0 += stride * BN_U) {
    float xv[BN_U][VEC];
#pragma unroll
    for (int u = 0; u < BN_U; ++u) {
      const int64_t g = g0 + u * stride;
      load_vec<VEC>(x + (g < groups ? g : g0) * VEC, xv[u]);      // clamped: the load itself is unconditional
    }
#pragma unroll
    for (int u = 0; u < BN_U; ++u) {
      const int64_t g = g0 + u * stride;
      if (g >= groups) break;
      const int col = (int)(g % cq);
      float yv[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int ch = col * VEC + v;
        const float xh = (xv[u][v] - K_(0, ch)) * K_(1, ch);
        const float t = fmaf(xh, K_(2, ch), K_(3, ch));
        yv[v] = t > 0.f ? t : t * leak;
      }
      store_vec<VEC>(y + g * VEC, yv);
    }
  }
}

// coef[0][c] = mean(dz), coef[1][c] = mean(dz*xhat); also dgamma/dbeta.  One workgroup per channel.
__global__ __launch_bounds__(256) void k_bn_finalize_bwd(const double *__restrict__ partial, int nblk,
                                                        int64_t n, int c, float *__restrict__ dgamma,
                                                        float *__restrict__ dbeta, float *__restrict__ coef) {
  __shared__ double sh[512];
  const int ch = blockIdx.x;
  double s, s2;
  bn_reduce_channel(partial, nblk, c, ch, sh, s, s2);
  if (threadIdx.x == 0) {
    if (dbeta) dbeta[ch] = (float)s;
    if (dgamma) dgamma[ch] = (float)s2;
    coef[ch] = (float)(s / (double)n);
    coef[c + ch] = (float)(s2 / (double)n);
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float *__restrict__ x, const float *__restrict__ dy,
                                                     int64_t n, int c, int cq, const float *__restrict__ mean,
                                                     const float *__restrict__ invstd,
                                                     const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, float leak, int training,
                                                     const float *__restrict__ coef, float *dx, BnFuse fuse,
                                                     const float *addend) {
  __shared__ float s_coef[2 * BN_FUSE_MAXC];
  __shared__ double s_scratch[512], s_tot[2 * BN_FUSE_MAXC];
  if (fuse.partial) {
    bn_fuse_totals(fuse, c, s_scratch, s_tot);
    for (int ch = threadIdx.x; ch < c; ch += 256) {
      const double s1 = s_tot[ch], s2 = s_tot[c + ch];
      s_coef[ch] = (float)(s1 / (double)n);
      s_coef[c + ch] = (float)(s2 / (double)n);
      if (blockIdx.x == 0) {
        if (fuse.dbeta) fuse.dbeta[ch] = (float)s1;
        if (fuse.dgamma) fuse.dgamma[ch] = (float)s2;
      }
    }
    __syncthreads();
    coef = s_coef;
  }
  __shared__ __attribute__((aligned(16))) float s_k[6 * BN_LDS_MAXC];
  const bool lds_k = c <= BN_LDS_MAXC;
  if (lds_k) {
    for (int ch = threadIdx.x; ch < c; ch += 256) {
      s_k[ch] = mean[ch];
      s_k[BN_LDS_MAXC + ch] = invstd[ch];
      s_k[2 * BN_LDS_MAXC + ch] = gamma ? gamma[ch] : 1.f;
      s_k[3 * BN_LDS_MAXC + ch] = beta ? beta[ch] : 0.f;
      s_k[4 * BN_LDS_MAXC + ch] = training ? coef[ch] : 0.f;
      s_k[5 * BN_LDS_MAXC + ch] = training ? coef[c + ch] : 0.f;
    }
    __syncthreads();
  }
  auto K_ = [&](int which, int ch) -> float {
    if (lds_k) return s_k[which * BN_LDS_MAXC + ch];
    switch (which) {
      case 0: return mean[ch];
      case 1: return invstd[ch];
      case 2: return gamma ? gamma[ch] : 1.f;
      case 3: return beta ? beta[ch] : 0.f;
      case 4: return training ? coef[ch] : 0.f;
      default: return training ? coef[c + ch] : 0.f;
    }
  };
  const int64_t groups = n * cq;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t g0 = (int64_t)blockIdx.x * 256 + threadIdx.x; g0 < groups; g0 += stride * BN_U) {
    float xv[BN_U][VEC], dv[BN_U][VEC], av[BN_U][VEC];
#pragma unroll
    for (int u = 0; u < BN_U; ++u) {
      const int64_t g = g0 + u * stride, gc = (g < groups ? g : g0);
      load_vec<VEC>(x + gc * VEC, xv[u]);
      load_vec<VEC>(dy + gc * VEC, dv[u]);
      if (addend) load_vec<VEC>(addend + gc * VEC, av[u]);      // uniform over the launch
    }
#pragma unroll
    for (int u = 0; u < BN_U; ++u) {
      const int64_t g = g0 + u * stride;
      if (g >= groups) break;
      const int col = (int)(g % cq);
      float ov[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const int ch = col * VEC + v;
        const float gm = K_(2, ch), is = K_(1, ch);
        const float xh = (xv[u][v] - K_(0, ch)) * is;
        const float t = fmaf(xh, gm, K_(3, ch));
        const float dz = t > 0.f ? dv[u][v] : dv[u][v] * leak;
        float d = dz;
        if (training) d = dz - K_(4, ch) - xh * K_(5, ch);
        ov[v] = d * gm * is;
        if (addend) ov[v] += av[u][v];   // gradient already accumulated for this buffer (may be dx itself: in place)
      }
      store_vec<VEC>(dx + g * VEC, ov);
    }
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
  BnFuse fuse{nullptr, 0, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (training && n > 0) {
    SGNN_CHECK_ARG(x);
    if (!ws || ws_bytes < sgnn_bn_ws_bytes(n, c)) {
      sgnn_set_error("sgnn_bn_fwd: workspace too small");
      return SGNN_ENOWS;
    }
    const int nblk = bn_blocks(n, g);
    const size_t shbytes = (size_t)g.rpb * g.cq * 2 * g.vec * sizeof(double);
    if (g.vec == 4)
      hipLaunchKernelGGL((k_bn_partial<4, 0>), dim3(nblk), dim3(256), shbytes, s, x, nullptr, n, c, g.cq, g.rpb,
                         nullptr, nullptr, nullptr, nullptr, 0.f, (double *)ws);
    else
      hipLaunchKernelGGL((k_bn_partial<1, 0>), dim3(nblk), dim3(256), shbytes, s, x, nullptr, n, c, g.cq, g.rpb,
                         nullptr, nullptr, nullptr, nullptr, 0.f, (double *)ws);
    if (nblk <= BN_FUSE_BLOCKS && c <= BN_FUSE_MAXC)   // small level: k_bn_apply finalises
      fuse = BnFuse{(const double *)ws, nblk, eps, momentum, running_mean, running_var, save_mean, save_invstd, nullptr, nullptr};
    else
      hipLaunchKernelGGL(k_bn_finalize_fwd, dim3(c), dim3(256), 0, s, (const double *)ws, nblk, n, c, eps, momentum,
                         running_mean, running_var, save_mean, save_invstd);
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
                         (const float *)save_invstd, gamma, beta, leak, y, fuse);
    else
      hipLaunchKernelGGL((k_bn_apply<1>), dim3(grid), dim3(256), 0, s, x, n, c, g.cq, (const float *)save_mean,
                         (const float *)save_invstd, gamma, beta, leak, y, fuse);
  }
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_bn_bwd(const float *x, const float *dy, int64_t n, int c, const float *gamma,
                            const float *beta, const float *save_mean, const float *save_invstd, int training,
                            float leak, float *dx, float *dgamma, float *dbeta, void *ws, int64_t ws_bytes,
                            sgnn_stream_t stream) {
  return sgnn_bn_bwd_add(x, dy, n, c, gamma, beta, save_mean, save_invstd, training, leak, nullptr, dx, dgamma, dbeta,
                         ws, ws_bytes, stream);
}

SGNN_EXPORT int sgnn_bn_bwd_add(const float *x, const float *dy, int64_t n, int c, const float *gamma,
                                const float *beta, const float *save_mean, const float *save_invstd, int training,
                                float leak, const float *addend, float *dx, float *dgamma, float *dbeta, void *ws,
                                int64_t ws_bytes, sgnn_stream_t stream) {
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
  const size_t shbytes = (size_t)g.rpb * g.cq * 2 * g.vec * sizeof(double);
  double *partial = (double *)ws;
  float *coef = (float *)((char *)ws + (size_t)BN_MAX_BLOCKS * 2 * c * sizeof(double));
  if (g.vec == 4)
    hipLaunchKernelGGL((k_bn_partial<4, 1>), dim3(nblk), dim3(256), shbytes, s, x, dy, n, c, g.cq, g.rpb, save_mean,
                       save_invstd, gamma, beta, leak, partial);
  else
    hipLaunchKernelGGL((k_bn_partial<1, 1>), dim3(nblk), dim3(256), shbytes, s, x, dy, n, c, g.cq, g.rpb, save_mean,
                       save_invstd, gamma, beta, leak, partial);
  BnFuse fuse{nullptr, 0, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (nblk <= BN_FUSE_BLOCKS && c <= BN_FUSE_MAXC)
    fuse = BnFuse{(const double *)partial, nblk, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, dgamma, dbeta};
  else
    hipLaunchKernelGGL(k_bn_finalize_bwd, dim3(c), dim3(256), 0, s, (const double *)partial, nblk, n, c, dgamma, dbeta,
                       coef);
  const int grid = sgnn_grid_for(n * g.cq, 256, 2048);
  if (g.vec == 4)
    hipLaunchKernelGGL((k_bn_bwd_apply<4>), dim3(grid), dim3(256), 0, s, x, dy, n, c, g.cq, save_mean, save_invstd,
                       gamma, beta, leak, training, (const float *)coef, dx, fuse, addend);
  else
    hipLaunchKernelGGL((k_bn_bwd_apply<1>), dim3(grid), dim3(256), 0, s, x, dy, n, c, g.cq, save_mean, save_invstd,
                       gamma, beta, leak, training, (const float *)coef, dx, fuse, addend);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}
