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
#ifndef BN_U
#define BN_U 2            // row groups a thread of the apply passes loads before it uses the first one (4 / 8: measured, no gain)
#endif
#define BN_FUSE_BLOCKS 4096
#define BN_FUSE_MAXC 64

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

static BnGeom bn_geom_scalar(int c) {   // unaligned views: one float per thread-column
  BnGeom g;
  g.vec = 1;
  g.cq = c;
  g.rpb = 256 / c;
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
// ldx / ld_dy: row strides in floats (>= c; rows of a column range of a wider buffer)
template <int VEC, int MODE, bool LIN = false>
__global__ __launch_bounds__(256) void k_bn_partial(const float *__restrict__ x, int64_t ldx,
                                                   const float *__restrict__ dy, int64_t ld_dy, int64_t n, int c,
                                                   int cq, int rpb, const float *__restrict__ mean,
                                                   const float *__restrict__ invstd,
                                                   const float *__restrict__ gamma,
                                                   const float *__restrict__ beta, float leak,
                                                   double *__restrict__ partial, const int64_t *n_dev, BnLin lin) {
  extern __shared__ double sh[];  // [rpb][2][c] would be large; reduce per column group instead
  const int tid = threadIdx.x;
  unsigned nblk = gridDim.x;
  if (n_dev) {   // capacity mode: the blocks an exact-size launch would use share the live rows; the rest write zeros
    n = sgnn_dyn_n(n, n_dev);
    int64_t b = (n + (int64_t)rpb * BN_FLUSH - 1) / ((int64_t)rpb * BN_FLUSH);
    if (b < 1) b = 1;
    if (b > BN_MAX_BLOCKS) b = BN_MAX_BLOCKS;
    if (b < (int64_t)nblk) nblk = (unsigned)b;
    if (blockIdx.x >= nblk) {
      for (int o = tid; o < 2 * c; o += 256) partial[(size_t)blockIdx.x * 2 * c + o] = 0.0;
      return;
    }
  }
  const int col = tid % cq, rloc = tid / cq;
  const bool active = rloc < rpb;
  double sa[VEC], sb[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) sa[v] = sb[v] = 0.0;
  float m_[VEC], is_[VEC], g_[VEC], b_[VEC];
  float lw[2][VEC];        // LIN: the head's weight rows for this thread's channels (BnLin)
  if constexpr (LIN) {
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int v = 0; v < VEC; ++v) lw[o][v] = (active && o < lin.nout) ? lin.w[o][col * VEC + v] : 0.f;
  }
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
    const int64_t step = (int64_t)nblk * rpb;
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
        load_vec<VEC>(x + rc * ldx + col * VEC, xv[it]);
        if constexpr (LIN) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) dv[it][v] = 0.f;
#pragma unroll
          for (int o = 0; o < 2; ++o)
            if (o < lin.nout) {
              const float gv = lin.g[rc * lin.ldg + o];
#pragma unroll
              for (int v = 0; v < VEC; ++v) dv[it][v] = fmaf(gv, lw[o][v], dv[it][v]);
            }
        } else if (MODE == 1) {
          load_vec<VEC>(dy + rc * ld_dy + col * VEC, dv[it]);
        }
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
  // eight rows in flight per thread (a load -> add chain pays one L2 round trip per row: 6 of them at 1430 block partials);
  // the additions keep their order, rows past the end add an exact 0.0
  for (int blk = threadIdx.x; blk < nblk; blk += 8 * 256) {
    double va[8], vb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int bi = blk + u * 256;
      const bool ok = bi < nblk;
      va[u] = ok ? partial[((size_t)bi * 2 + 0) * c + ch] : 0.0;
      vb[u] = ok ? partial[((size_t)bi * 2 + 1) * c + ch] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a += va[u];
      b += vb[u];
    }
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
                                                        float *__restrict__ save_invstd, const int64_t *n_dev) {
  __shared__ double sh[512];
  n = sgnn_dyn_n(n, n_dev);
  const int ch = blockIdx.x;
  double s, s2;
  bn_reduce_channel(partial, nblk, c, ch, sh, s, s2);
  if (threadIdx.x == 0) {
    if (n <= 0) {   // capacity mode, empty level: the reference never runs the layer — running statistics untouched
      save_mean[ch] = 0.f;
      save_invstd[ch] = 0.f;
      return;
    }
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
// Round 6: what this costs is the NUMBER of dependent load rounds, ~0.75 us each (the table was written by the producing
// kernel's workgroups on all eight XCDs: every first touch misses this XCD's L2) — 6 rounds for the 684 16-row partials of a
// 11 k-row level with one 8-byte load per value and 8 blocks in flight (k_bn_apply 7.4 us there against 4.5 us on a 60 k-row
// level with 236 partials).  Even channel counts read two channels per 16-byte load and keep 32 loads in flight: the same
// table in 2 rounds.  The kernels that call this are the FUSE instantiations (small levels only), so the 64 + 64 registers
// cost the streaming instantiations nothing.  Any fixed order is a valid order: deterministic per (nblk, c).
#ifndef BN_TOTALS_WIDE
#define BN_TOTALS_WIDE 1     // 0: the one-value-per-load form everywhere (scripts/build_variant.sh A/B builds)
#endif
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void bn_fuse_totals(const BnFuse &f, int c, double *scratch, double *tot) {
  const int tid = threadIdx.x;
  const int tpc = 256 / c;                       // >= 4 for c <= BN_FUSE_MAXC
  if (BN_TOTALS_WIDE && (c & 1) == 0 && (((uintptr_t)f.partial) & 15) == 0) {
    // thread = (part, which, channel pair): sums `which` (0: sum_a, 1: sum_b) of channels 2 p, 2 p + 1 over blocks part, part + tpc, ...
    const int hp = c >> 1;
    const int p2 = tid % hp, which = (tid / hp) & 1, part = tid / c;
    f64x2 acc = {0.0, 0.0};
    if (part < tpc) {
      constexpr int U = 32;
      for (int blk = part; blk < f.nblk; blk += U * tpc) {
        f64x2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int bi = blk + u * tpc;
          v[u] = bi < f.nblk ? *reinterpret_cast<const f64x2 *>(f.partial + ((size_t)bi * 2 + which) * c + 2 * p2) : f64x2{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
      }
      scratch[which * 256 + part * c + 2 * p2] = acc[0];
      scratch[which * 256 + part * c + 2 * p2 + 1] = acc[1];
    }
  } else {
    const int ch = tid % c, part = tid / c;
    double a = 0.0, b = 0.0;
    if (part < tpc) {
      // eight independent loads in flight per thread; the additions keep their order, missing entries add an exact 0.0
      for (int blk = part; blk < f.nblk; blk += 8 * tpc) {
        double va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int bi = blk + u * tpc;
          const bool ok = bi < f.nblk;
          va[u] = ok ? f.partial[((size_t)bi * 2 + 0) * c + ch] : 0.0;
          vb[u] = ok ? f.partial[((size_t)bi * 2 + 1) * c + ch] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a += va[u];
          b += vb[u];
        }
      }
      scratch[part * c + ch] = a;
      scratch[256 + part * c + ch] = b;
    }
  }
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

// Thread mapping of the apply passes = the one of the statistics pass: a thread owns one column group (VEC channels, its
// constants live in registers) and walks rows row0 + i*step, BN_U rows in flight; ldx / ldy = row strides (floats).
template <int VEC, bool FUSE>
__global__ __launch_bounds__(256) void k_bn_apply(const float *__restrict__ x, int64_t ldx, int64_t n, int c, int cq,
                                                 int rpb, const float *__restrict__ mean,
                                                 const float *__restrict__ invstd,
                                                 const float *__restrict__ gamma,
                                                 const float *__restrict__ beta, float leak,
                                                 float *__restrict__ y, int64_t ldy, BnFuse fuse,
                                                 const int64_t *n_dev) {
  __shared__ float s_mean[BN_FUSE_MAXC], s_inv[BN_FUSE_MAXC];
  __shared__ double s_scratch[512], s_tot[2 * BN_FUSE_MAXC];
  n = sgnn_dyn_n(n, n_dev);
  if (n <= 0) return;        // capacity mode, empty level (whole grid: no barrier is skipped by a part of a workgroup)
  if constexpr (FUSE) {
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
  const int tid = threadIdx.x;
  const int col = tid % cq, rloc = tid / cq;
  if (rloc >= rpb) return;
  float km[VEC], ki[VEC], kg[VEC], kb[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int ch = col * VEC + v;
    km[v] = mean[ch];
    ki[v] = invstd[ch];
    kg[v] = gamma ? gamma[ch] : 1.f;
    kb[v] = beta ? beta[ch] : 0.f;
  }
  const int64_t step = (int64_t)gridDim.x * rpb;
  for (int64_t r0 = (int64_t)blockIdx.x * rpb + rloc; r0 < n; r0 += step * BN_U) {
    float xv[BN_U][VEC];
#pragma unroll
    for (int u = 0; u < BN_U; ++u) {
      const int64_t r = r0 + u * step;
      load_vec<VEC>(x + (r < n ? r : r0) * ldx + col * VEC, xv[u]);      // clamped: the load itself is unconditional
    }
#pragma unroll
    for (int u = 0; u < BN_U; ++u) {
      const int64_t r = r0 + u * step;
      if (r >= n) break;
      float yv[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) yv[v] = sgnn_bn_act(xv[u][v], km[v], ki[v], kg[v], kb[v], leak);
      store_vec<VEC>(y + r * ldy + col * VEC, yv);
    }
  }
}

// coef[0][c] = mean(dz), coef[1][c] = mean(dz*xhat); also dgamma/dbeta.  One workgroup per channel.
__global__ __launch_bounds__(256) void k_bn_finalize_bwd(const double *__restrict__ partial, int nblk,
                                                        int64_t n, int c, float *__restrict__ dgamma,
                                                        float *__restrict__ dbeta, float *__restrict__ coef,
                                                        const int64_t *n_dev) {
  __shared__ double sh[512];
  n = sgnn_dyn_n(n, n_dev);
  const int ch = blockIdx.x;
  double s, s2;
  bn_reduce_channel(partial, nblk, c, ch, sh, s, s2);
  if (threadIdx.x == 0) {
    if (dbeta) dbeta[ch] = (float)s;
    if (dgamma) dgamma[ch] = (float)s2;
    coef[ch] = n > 0 ? (float)(s / (double)n) : 0.f;
    coef[c + ch] = n > 0 ? (float)(s2 / (double)n) : 0.f;
  }
}

template <int VEC, bool LIN, bool FUSE>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float *__restrict__ x, int64_t ldx,
                                                     const float *__restrict__ dy, int64_t ld_dy, int64_t n, int c,
                                                     int cq, int rpb, const float *__restrict__ mean,
                                                     const float *__restrict__ invstd,
                                                     const float *__restrict__ gamma,
                                                     const float *__restrict__ beta, float leak, int training,
                                                     const float *__restrict__ coef, float *dx, int64_t ld_dx,
                                                     BnFuse fuse, const float *addend, int64_t ld_add,
                                                     const int64_t *n_dev, BnLin lin) {
  __shared__ float s_coef[2 * BN_FUSE_MAXC];
  __shared__ double s_scratch[512], s_tot[2 * BN_FUSE_MAXC];
  n = sgnn_dyn_n(n, n_dev);
  if constexpr (FUSE) {
    bn_fuse_totals(fuse, c, s_scratch, s_tot);
    for (int ch = threadIdx.x; ch < c; ch += 256) {
      const double s1 = s_tot[ch], s2 = s_tot[c + ch];
      s_coef[ch] = n > 0 ? (float)(s1 / (double)n) : 0.f;
      s_coef[c + ch] = n > 0 ? (float)(s2 / (double)n) : 0.f;
      if (blockIdx.x == 0) {
        if (fuse.dbeta) fuse.dbeta[ch] = (float)s1;
        if (fuse.dgamma) fuse.dgamma[ch] = (float)s2;
      }
    }
    __syncthreads();
    coef = s_coef;
  }
  const int tid = threadIdx.x;
  const int col = tid % cq, rloc = tid / cq;
  if (rloc >= rpb) return;
  float km[VEC], ki[VEC], kg[VEC], kb[VEC], k4[VEC], k5[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    const int ch = col * VEC + v;
    km[v] = mean[ch];
    ki[v] = invstd[ch];
    kg[v] = gamma ? gamma[ch] : 1.f;
    kb[v] = beta ? beta[ch] : 0.f;
    k4[v] = training ? coef[ch] : 0.f;
    k5[v] = training ? coef[c + ch] : 0.f;
  }
  float lw[2][VEC];
  if constexpr (LIN) {
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int v = 0; v < VEC; ++v) lw[o][v] = o < lin.nout ? lin.w[o][col * VEC + v] : 0.f;
  }
  const int64_t step = (int64_t)gridDim.x * rpb;
  for (int64_t r0 = (int64_t)blockIdx.x * rpb + rloc; r0 < n; r0 += step * BN_U) {
    float xv[BN_U][VEC], dv[BN_U][VEC], av[BN_U][VEC];
#pragma unroll
    for (int u = 0; u < BN_U; ++u) {
      const int64_t r = r0 + u * step, rc = (r < n ? r : r0);
      load_vec<VEC>(x + rc * ldx + col * VEC, xv[u]);
      if constexpr (LIN) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) dv[u][v] = 0.f;
#pragma unroll
        for (int o = 0; o < 2; ++o)
          if (o < lin.nout) {
            const float gv = lin.g[rc * lin.ldg + o];
#pragma unroll
            for (int v = 0; v < VEC; ++v) dv[u][v] = fmaf(gv, lw[o][v], dv[u][v]);
          }
      } else {
        load_vec<VEC>(dy + rc * ld_dy + col * VEC, dv[u]);
      }
      if (addend) load_vec<VEC>(addend + rc * ld_add + col * VEC, av[u]);      // uniform over the launch
    }
#pragma unroll
    for (int u = 0; u < BN_U; ++u) {
      const int64_t r = r0 + u * step;
      if (r >= n) break;
      float ov[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float xh = (xv[u][v] - km[v]) * ki[v];
        const float t = fmaf(xh, kg[v], kb[v]);
        const float dz = t > 0.f ? dv[u][v] : dv[u][v] * leak;
        float d = dz;
        if (training) d = dz - k4[v] - xh * k5[v];
        ov[v] = d * kg[v] * ki[v];
        if (addend) ov[v] += av[u][v];   // gradient already accumulated for this buffer (may be dx itself: in place)
      }
      store_vec<VEC>(dx + r * ld_dx + col * VEC, ov);
    }
  }
}

// Finalising inside the apply kernels makes EVERY apply workgroup re-sum the partial table (L2-resident): worth it as
// long as that re-read stays small next to a separate launch (~5 us of GPU time + ~3 us of host time per finalize, ~70
// of them per training step).  Budget: 48 MB of partial reads per apply launch.
// (A finalise kernel made of ONE 1024-thread workgroup that reads the partial table in coalesced rows was measured against
//  the c single-channel workgroups of k_bn_finalize_fwd / _bwd: +3.8 us per launch inside the step, 6.64 vs 6.51 ms per step
//  — one CU cannot pull a 366 KB table as fast as 16 can; dropped.)
static bool bn_fuse_ok(int64_t nblk, int c, int apply_grid) {
  return c <= BN_FUSE_MAXC && nblk <= BN_FUSE_BLOCKS && nblk * (int64_t)apply_grid * 2 * c * (int64_t)sizeof(double) <= (48ll << 20);
}

static int bn_apply_grid(int64_t n, const BnGeom &g) {
  int64_t b = (n + (int64_t)g.rpb * BN_U - 1) / ((int64_t)g.rpb * BN_U);
  if (b < 1) b = 1;
  if (b > 4096) b = 4096;
  return (int)b;
}

// pre_partial / pre_nblk: statistics partials already produced by the convolution that wrote x (ConvEpi.stats = 1):
// the statistics pass is skipped.  ldx / ldy: row strides (0 = c).
int sgnn_bn_fwd_impl(const float *x, int64_t ldx, int64_t n, int c, const float *gamma, const float *beta,
                     float *running_mean, float *running_var, float eps, float momentum, int training, float leak,
                     float *save_mean, float *save_invstd, float *y, int64_t ldy, const double *pre_partial,
                     int64_t pre_nblk, void *ws, int64_t ws_bytes, sgnn_stream_t stream, const int64_t *n_dev) {
  hipStream_t s = (hipStream_t)stream;
  SGNN_CHECK_ARG(n >= 0 && c >= 1 && (c % 4 == 0 ? c <= 1024 : c <= 256) && save_mean && save_invstd);
  SGNN_CHECK_ARG(training || (running_mean && running_var));
  if (ldx <= 0) ldx = c;
  if (ldy <= 0) ldy = c;
  SGNN_CHECK_ARG(ldx >= c && ldy >= c);
  BnGeom g = bn_geom(c);
  if (g.vec == 4 && (ldx % 4 || ldy % 4 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15))) g = bn_geom_scalar(c);
  BnFuse fuse{nullptr, 0, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (training && n > 0) {
    SGNN_CHECK_ARG(x);
    const double *partial = pre_partial;
    int64_t nblk = pre_nblk;
    if (!partial) {
      if (!ws || ws_bytes < sgnn_bn_ws_bytes(n, c)) {
        sgnn_set_error("sgnn_bn_fwd: workspace too small");
        return SGNN_ENOWS;
      }
      nblk = bn_blocks(n, g);
      const size_t shbytes = (size_t)g.rpb * g.cq * 2 * g.vec * sizeof(double);
      if (g.vec == 4)
        SGNN_LAUNCH((k_bn_partial<4, 0>), dim3((unsigned)nblk), dim3(256), shbytes, s, x, ldx, nullptr, 0, n, c,
                           g.cq, g.rpb, nullptr, nullptr, nullptr, nullptr, 0.f, (double *)ws, n_dev, BnLin{});
      else
        SGNN_LAUNCH((k_bn_partial<1, 0>), dim3((unsigned)nblk), dim3(256), shbytes, s, x, ldx, nullptr, 0, n, c,
                           g.cq, g.rpb, nullptr, nullptr, nullptr, nullptr, 0.f, (double *)ws, n_dev, BnLin{});
      partial = (const double *)ws;
    }
    if (bn_fuse_ok(nblk, c, bn_apply_grid(n, g)))   // small level: k_bn_apply finalises
      fuse = BnFuse{partial, (int)nblk, eps, momentum, running_mean, running_var, save_mean, save_invstd, nullptr, nullptr};
    else
      SGNN_LAUNCH(k_bn_finalize_fwd, dim3(c), dim3(256), 0, s, partial, (int)nblk, n, c, eps, momentum,
                         running_mean, running_var, save_mean, save_invstd, n_dev);
  } else if (training) {  // empty batch: identity statistics, nothing to normalise
    SGNN_HIP_TRY(hipMemsetAsync(save_mean, 0, c * sizeof(float), s));
    SGNN_HIP_TRY(hipMemsetAsync(save_invstd, 0, c * sizeof(float), s));
  } else {
    SGNN_LAUNCH(k_bn_eval_stats, dim3(1), dim3(256), 0, s, (const float *)running_mean,
                       (const float *)running_var, c, eps, save_mean, save_invstd);
  }
  if (n > 0) {
    SGNN_CHECK_ARG(x && y);
    const int grid = bn_apply_grid(n, g);
#define BN_APPLY(VECV, FUSEV)                                                                                 \
  SGNN_LAUNCH((k_bn_apply<VECV, FUSEV>), dim3(grid), dim3(256), 0, s, x, ldx, n, c, g.cq, g.rpb,              \
              (const float *)save_mean, (const float *)save_invstd, gamma, beta, leak, y, ldy, fuse, n_dev)
    if (g.vec == 4 && fuse.partial) BN_APPLY(4, true);
    else if (g.vec == 4) BN_APPLY(4, false);
    else if (fuse.partial) BN_APPLY(1, true);
    else BN_APPLY(1, false);
#undef BN_APPLY
  }
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_bn_fwd(const float *x, int64_t n, int c, const float *gamma, const float *beta,
                            float *running_mean, float *running_var, float eps, float momentum, int training,
                            float leak, float *save_mean, float *save_invstd, float *y, void *ws,
                            int64_t ws_bytes, sgnn_stream_t stream) {
  return sgnn_bn_fwd_impl(x, c, n, c, gamma, beta, running_mean, running_var, eps, momentum, training, leak, save_mean,
                          save_invstd, y, c, nullptr, 0, ws, ws_bytes, stream);
}

SGNN_EXPORT int sgnn_bn_fwd_ex(const float *x, int64_t ldx, int64_t n, int c, const float *gamma, const float *beta,
                               float *running_mean, float *running_var, float eps, float momentum, int training,
                               float leak, float *save_mean, float *save_invstd, float *y, int64_t ldy,
                               const double *pre_partial, int64_t pre_nblk, void *ws, int64_t ws_bytes,
                               sgnn_stream_t stream) {
  return sgnn_bn_fwd_impl(x, ldx, n, c, gamma, beta, running_mean, running_var, eps, momentum, training, leak,
                          save_mean, save_invstd, y, ldy, pre_partial, pre_nblk, ws, ws_bytes, stream);
}

SGNN_EXPORT int sgnn_bn_bwd(const float *x, const float *dy, int64_t n, int c, const float *gamma,
                            const float *beta, const float *save_mean, const float *save_invstd, int training,
                            float leak, float *dx, float *dgamma, float *dbeta, void *ws, int64_t ws_bytes,
                            sgnn_stream_t stream) {
  return sgnn_bn_bwd_impl(x, c, dy, c, n, c, gamma, beta, save_mean, save_invstd, training, leak, nullptr, c, dx, c,
                          dgamma, dbeta, nullptr, 0, ws, ws_bytes, stream);
}

SGNN_EXPORT int sgnn_bn_bwd_add(const float *x, const float *dy, int64_t n, int c, const float *gamma,
                                const float *beta, const float *save_mean, const float *save_invstd, int training,
                                float leak, const float *addend, float *dx, float *dgamma, float *dbeta, void *ws,
                                int64_t ws_bytes, sgnn_stream_t stream) {
  return sgnn_bn_bwd_impl(x, c, dy, c, n, c, gamma, beta, save_mean, save_invstd, training, leak, addend, c, dx, c,
                          dgamma, dbeta, nullptr, 0, ws, ws_bytes, stream);
}

SGNN_EXPORT int sgnn_bn_bwd_ex(const float *x, int64_t ldx, const float *dy, int64_t ld_dy, int64_t n, int c,
                               const float *gamma, const float *beta, const float *save_mean,
                               const float *save_invstd, int training, float leak, const float *addend,
                               int64_t ld_add, float *dx, int64_t ld_dx, float *dgamma, float *dbeta,
                               const double *pre_partial, int64_t pre_nblk, void *ws, int64_t ws_bytes,
                               sgnn_stream_t stream) {
  return sgnn_bn_bwd_impl(x, ldx, dy, ld_dy, n, c, gamma, beta, save_mean, save_invstd, training, leak, addend, ld_add,
                          dx, ld_dx, dgamma, dbeta, pre_partial, pre_nblk, ws, ws_bytes, stream);
}

// pre_partial: sum dz / sum dz*xhat partials produced by the data-gradient convolution that wrote dy (ConvEpi.stats = 2)
int sgnn_bn_bwd_impl(const float *x, int64_t ldx, const float *dy, int64_t ld_dy, int64_t n, int c, const float *gamma,
                     const float *beta, const float *save_mean, const float *save_invstd, int training, float leak,
                     const float *addend, int64_t ld_add, float *dx, int64_t ld_dx, float *dgamma, float *dbeta,
                     const double *pre_partial, int64_t pre_nblk, void *ws, int64_t ws_bytes, sgnn_stream_t stream,
                     const int64_t *n_dev, const BnLin *lin) {
  hipStream_t s = (hipStream_t)stream;
  SGNN_CHECK_ARG(n >= 0 && c >= 1 && (c % 4 == 0 ? c <= 1024 : c <= 256) && save_mean && save_invstd);
  SGNN_CHECK_ARG(!lin || (lin->g && lin->nout >= 1 && lin->nout <= 2 && lin->w[0] && (lin->nout < 2 || lin->w[1]) && !pre_partial));
  if (n == 0) {
    if (dgamma) SGNN_HIP_TRY(hipMemsetAsync(dgamma, 0, c * sizeof(float), s));
    if (dbeta) SGNN_HIP_TRY(hipMemsetAsync(dbeta, 0, c * sizeof(float), s));
    return SGNN_OK;
  }
  SGNN_CHECK_ARG(x && (dy || lin) && dx);
  if (ldx <= 0) ldx = c;
  if (ld_dy <= 0) ld_dy = c;
  if (ld_dx <= 0) ld_dx = c;
  if (ld_add <= 0) ld_add = c;
  SGNN_CHECK_ARG(ldx >= c && ld_dy >= c && ld_dx >= c && ld_add >= c);
  if (!ws || ws_bytes < sgnn_bn_ws_bytes(n, c)) {
    sgnn_set_error("sgnn_bn_bwd: workspace too small");
    return SGNN_ENOWS;
  }
  BnGeom g = bn_geom(c);
  if (g.vec == 4 && (ldx % 4 || ld_dy % 4 || ld_dx % 4 || ld_add % 4 || ((uintptr_t)x & 15) || ((uintptr_t)dy & 15) ||
                     ((uintptr_t)dx & 15) || ((uintptr_t)addend & 15)))
    g = bn_geom_scalar(c);
  const double *partial = pre_partial;
  int64_t nblk = pre_nblk;
  // the coefficient block always lives behind the largest partial table this library writes into ws
  float *coef = (float *)((char *)ws + (size_t)BN_MAX_BLOCKS * 2 * c * sizeof(double));
  if (!partial) {
    nblk = bn_blocks(n, g);
    const size_t shbytes = (size_t)g.rpb * g.cq * 2 * g.vec * sizeof(double);
    if (lin && g.vec == 4)
      SGNN_LAUNCH((k_bn_partial<4, 1, true>), dim3((unsigned)nblk), dim3(256), shbytes, s, x, ldx, dy, ld_dy, n, c, g.cq,
                         g.rpb, save_mean, save_invstd, gamma, beta, leak, (double *)ws, n_dev, *lin);
    else if (lin)
      SGNN_LAUNCH((k_bn_partial<1, 1, true>), dim3((unsigned)nblk), dim3(256), shbytes, s, x, ldx, dy, ld_dy, n, c, g.cq,
                         g.rpb, save_mean, save_invstd, gamma, beta, leak, (double *)ws, n_dev, *lin);
    else if (g.vec == 4)
      SGNN_LAUNCH((k_bn_partial<4, 1>), dim3((unsigned)nblk), dim3(256), shbytes, s, x, ldx, dy, ld_dy, n, c, g.cq,
                         g.rpb, save_mean, save_invstd, gamma, beta, leak, (double *)ws, n_dev, BnLin{});
    else
      SGNN_LAUNCH((k_bn_partial<1, 1>), dim3((unsigned)nblk), dim3(256), shbytes, s, x, ldx, dy, ld_dy, n, c, g.cq,
                         g.rpb, save_mean, save_invstd, gamma, beta, leak, (double *)ws, n_dev, BnLin{});
    partial = (const double *)ws;
  }
  BnFuse fuse{nullptr, 0, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (bn_fuse_ok(nblk, c, bn_apply_grid(n, g)))
    fuse = BnFuse{partial, (int)nblk, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, dgamma, dbeta};
  else
    SGNN_LAUNCH(k_bn_finalize_bwd, dim3(c), dim3(256), 0, s, partial, (int)nblk, n, c, dgamma, dbeta, coef, n_dev);
  const int grid = bn_apply_grid(n, g);
#define BN_BWD_APPLY(VECV, LINV, FUSEV)                                                                                       \
  SGNN_LAUNCH((k_bn_bwd_apply<VECV, LINV, FUSEV>), dim3(grid), dim3(256), 0, s, x, ldx, dy, ld_dy, n, c, g.cq, g.rpb, save_mean, \
              save_invstd, gamma, beta, leak, training, (const float *)coef, dx, ld_dx, fuse, addend, ld_add, n_dev,              \
              LINV ? *lin : BnLin{})
  const bool fz = fuse.partial != nullptr;
  if (lin && g.vec == 4) { if (fz) BN_BWD_APPLY(4, true, true); else BN_BWD_APPLY(4, true, false); }
  else if (lin) { if (fz) BN_BWD_APPLY(1, true, true); else BN_BWD_APPLY(1, true, false); }
  else if (g.vec == 4) { if (fz) BN_BWD_APPLY(4, false, true); else BN_BWD_APPLY(4, false, false); }
  else { if (fz) BN_BWD_APPLY(1, false, true); else BN_BWD_APPLY(1, false, false); }
#undef BN_BWD_APPLY
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}
