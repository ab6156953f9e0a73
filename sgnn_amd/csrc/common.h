// Shared device/host helpers for libsgnn_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sgnn_hip.h"

#define SGNN_EXPORT extern "C" __attribute__((visibility("default")))

extern "C" int sgnn_stamp(const char *label, sgnn_stream_t stream);   // prof.hip: nothing unless enabled
extern sgnn_tune g_tune;   // tune.hip: the library's one table of measurement switches (include/sgnn_hip.h documents every field)

void sgnn_set_error(const char *fmt, ...);
// prof.hip: optional HIP-event timing of conv launches (slot < 0 = not recording)
int sgnn_prof_begin_launch(int kind, int64_t n_out, int cin, int cout, int K, int flags, hipStream_t s);
void sgnn_prof_end_launch(int slot, hipStream_t s);

// Epilogue / layout options of the output-stationary kernel (sgnn_conv_fwd_epi; prog.hip fuses with them):
//  * ldx / ldy / ld_add: row strides (floats) of x, y and addend, so that a convolution can read from and write
//    into a column range of a wider buffer (JoinTable without a copy);
//  * addend: y = conv + addend (the residual AddTable, or gradient accumulation in place: addend may alias y);
//  * stats = 1: per-workgroup column sums of y and y*y (the statistics the next BatchNorm needs);
//    stats = 2: y is the gradient reaching a BatchNormReLU output; column sums of dz and dz*xhat with
//    dz = y * (bn_out > 0 ? 1 : leak), xhat from bn_x / mean / invstd (what BatchNorm backward reduces first).
//    partial[blk][2][COUT] doubles, blk = workgroup; summed later in fixed order (deterministic).
// the arithmetic of BatchNormReLU's apply pass (bn.hip)
__device__ __forceinline__ float sgnn_bn_act(float x, float mean, float invstd, float gamma, float beta, float leak) {
  const float xh = (x - mean) * invstd;
  const float t = fmaf(xh, gamma, beta);
  return t > 0.f ? t : t * leak;
}

struct ConvEpi {
  int64_t ldx, ldy, ld_add;
  const float *addend;
  int stats;
  double *partial;
  const float *bn_x;
  int64_t ld_bnx;
  const float *mean, *invstd, *gamma, *beta;
  float leak;
  // capacity mode: the output row count lives in device memory (*n_dev, clamped to the n_out the launch was sized
  // for); NULL = the host value is exact.  Lets a whole training step be captured in a HIP graph (HISTORY.md §2).
  const int64_t *n_dev;
};

// conv.hip internals used by prog.hip
int sgnn_conv_fwd_impl(const float *x, int64_t n_in, int cin, const float *w, int K, const int32_t *table, int64_t ld,
                       int64_t n_out, int cout, float *y, int flags, int in_shift, const int32_t *kmap,
                       const int32_t *kadd, int in_mul, int groups, int table_rows, const ConvEpi *epi,
                       sgnn_stream_t stream);
int64_t sgnn_conv_grid_blocks(int64_t n_out, int cin, int cout, int K);
// BatchNorm backward whose incoming gradient is the data gradient of a per-site linear head that nobody else reads (the
// surface head, model.py:433-447): dy[r][ch] = sum_o g[r*ldg + o] * w[o][ch] is formed in the two BatchNorm passes instead of
// being written (k_linear_bwd) and read back twice — the same fmaf chain, bit-identical values.
struct BnLin {
  const float *g;
  int64_t ldg;
  const float *w[2];
  int nout;
};
// bn.hip internals used by prog.hip (strided rows, statistics partials supplied by a convolution epilogue)
// n_dev (every internal entry point below, default NULL): device row count, clamped to the host value n, which then
// is the capacity the launch is sized for
int sgnn_bn_fwd_impl(const float *x, int64_t ldx, int64_t n, int c, const float *gamma, const float *beta,
                     float *running_mean, float *running_var, float eps, float momentum, int training, float leak,
                     float *save_mean, float *save_invstd, float *y, int64_t ldy, const double *pre_partial,
                     int64_t pre_nblk, void *ws, int64_t ws_bytes, sgnn_stream_t stream, const int64_t *n_dev = nullptr);
int sgnn_bn_bwd_impl(const float *x, int64_t ldx, const float *dy, int64_t ld_dy, int64_t n, int c, const float *gamma,
                     const float *beta, const float *save_mean, const float *save_invstd, int training, float leak,
                     const float *addend, int64_t ld_add, float *dx, int64_t ld_dx, float *dgamma, float *dbeta,
                     const double *pre_partial, int64_t pre_nblk, void *ws, int64_t ws_bytes, sgnn_stream_t stream,
                     const int64_t *n_dev = nullptr, const struct BnLin *lin = nullptr);
bool sgnn_conv_epi_supported(int cin, int cout);
bool dw_shape_ok(int cin, int cout);   // conv.hip: compiled weight-gradient shapes (strided rows need one)
// deferred weight-gradient reduces (conv.hip): partial[nblk][elems] -> dw[elems], many tensors in one launch
#define DW_BATCH_MAX 28
struct DwDesc {
  const float *partial;
  float *dw;
  int64_t nblk, elems;
  int blk0;
};
struct DwBatch {
  DwDesc d[DW_BATCH_MAX];
  int n;
};
extern DwBatch *sgnn_dw_batch;
int sgnn_dw_batch_flush(DwBatch *b, hipStream_t s);
// rows.hip: strided row movement (prog.hip: JoinTable inputs written in place)
int sgnn_gather_rows_ld(const float *src, int64_t ld_src, int c, const int32_t *idx, int64_t m, float *dst,
                        int64_t ld_dst, sgnn_stream_t stream, const int64_t *n_dev = nullptr);
int sgnn_gather_sum_ld(const float *src, int64_t ld_src, int c, const int32_t *table, int64_t ld, int K, int64_t n_out,
                       float *dst, int64_t ld_dst, sgnn_stream_t stream, const int64_t *n_dev = nullptr);
int sgnn_add_ld(const float *a, int64_t lda, const float *b, int64_t ldb, int64_t n, int c, float *y, int64_t ldy,
                sgnn_stream_t stream, const int64_t *n_dev = nullptr);
// fill / copy as plain kernels (rows.hip).  hipMemsetAsync / hipMemcpyAsync become memset / memcpy NODES in a captured
// graph, and a replay stalls around each of them (50-150 us bubbles in the rocprofv3 timeline of a replayed step,
// profiles/r03b_trace_summary.txt); kernel nodes replay back to back.  regions: up to 4 (pointer, 32-bit words) pairs.
int sgnn_fill32(void *p, uint32_t pattern, int64_t words, hipStream_t s);
int sgnn_fill32_multi(void *const *p, const int64_t *words, int nregions, uint32_t pattern, hipStream_t s);
int sgnn_copy_words(void *dst, const void *src, int64_t words, hipStream_t s);
int sgnn_sum_groups_dn(const float *src, int c, int64_t n, int rep, float *dst, sgnn_stream_t stream, const int64_t *n_dev);
int sgnn_concat_rows_dn(const float *a, int ca, const int32_t *ia, const float *b, int cb, const int32_t *ib, int64_t m,
                        float *dst, sgnn_stream_t stream, const int64_t *n_dev);
int sgnn_concat_rows_bwd_dn(const float *ddst, int ca, const int32_t *ia, int cb, const int32_t *ib, int64_t m, float *da,
                            int64_t na, float *db, int64_t nb, sgnn_stream_t stream, const int64_t *n_dev);
int sgnn_concat3_rows_dn(const float *a, int ca, const int32_t *ia, const float *b, int cb, const int32_t *ib,
                         const float *c, int cc, const int32_t *ic, int64_t m, float *dst, sgnn_stream_t stream,
                         const int64_t *n_dev);
int sgnn_concat3_rows_bwd_dn(const float *ddst, int ca, const int32_t *ia, int cb, const int32_t *ib, int cc,
                             const int32_t *ic, int64_t m, float *da, int64_t na, float *db, int64_t nb, float *dc,
                             int64_t nc, sgnn_stream_t stream, const int64_t *n_dev);
int sgnn_conv_bwd_weight_impl(const float *x, int64_t n_in, int cin, int64_t ldx, const float *dy, int cout, int64_t ld_dy,
                              const int32_t *table, int64_t ld, int K, int64_t n_out, float *dw, int in_shift,
                              const int32_t *kmap, const int32_t *kadd, int in_mul, int groups, int table_rows, void *ws,
                              int64_t ws_bytes, sgnn_stream_t stream, const int64_t *n_dev = nullptr);
// conv_bwd_fused.hip: data gradient + weight gradient of a 3x3x3 submanifold layer from one gather of dy
bool sgnn_conv_bwd_fused_ok(int64_t n, int cin, int cout, int K);
bool sgnn_conv_bwd_fused_usable(int64_t n, int cin, int cout, int K, const ConvEpi &epi, const float *dx, const float *x,
                                int64_t ldx);
int64_t sgnn_conv_bwd_fused_ws(int64_t n, int cin, int cout);
int sgnn_conv_bwd_fused_impl(const float *dy, int64_t n, int cout, const float *w, const int32_t *table, int64_t ld, int cin,
                             float *dx, const ConvEpi &epi, const float *x, int64_t ldx, float *dw, void *ws, int64_t ws_bytes,
                             sgnn_stream_t stream);
int sgnn_expand_maps(const int32_t **S, const int32_t **ST, const int32_t **PAR);
// linear.hip: heads whose weight rows / biases are separate tensors
int sgnn_linear_fwd_rows(const float *x, int64_t n, int cin, const float *const *w, const float *const *b, int cout,
                         float *y, sgnn_stream_t stream, const int64_t *n_dev = nullptr);
int sgnn_linear_bwd_rows(const float *x, const float *dy, int64_t n, int cin, const float *const *w, int cout,
                         float *dx, float *const *dw, float *const *db, void *ws, int64_t ws_bytes,
                         sgnn_stream_t stream, const int64_t *n_dev = nullptr, const float *addend = nullptr,
                         int64_t ld_add = 0);

// every kernel launch of the library goes through this macro: sgnn_launch_count() (bench.py: launches per step =
// kernel nodes this library contributes to the captured graph of one training step)
extern long long sgnn_launch_counter;
#define SGNN_LAUNCH(...)           \
  do {                             \
    ++sgnn_launch_counter;         \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

#define SGNN_CHECK_ARG(cond)                                                    \
  do {                                                                          \
    if (!(cond)) {                                                              \
      sgnn_set_error("%s: invalid argument: %s", __func__, #cond);              \
      return SGNN_EINVAL;                                                       \
    }                                                                           \
  } while (0)

#define SGNN_CHECK_LAUNCH()                                                     \
  do {                                                                          \
    hipError_t e_ = hipGetLastError();                                          \
    if (e_ != hipSuccess) {                                                     \
      sgnn_set_error("%s: HIP error: %s", __func__, hipGetErrorString(e_));     \
      return SGNN_EHIP;                                                         \
    }                                                                           \
  } while (0)

#define SGNN_HIP_TRY(expr)                                                      \
  do {                                                                          \
    hipError_t e_ = (expr);                                                     \
    if (e_ != hipSuccess) {                                                     \
      sgnn_set_error("%s: %s failed: %s", __func__, #expr, hipGetErrorString(e_)); \
      return SGNN_EHIP;                                                         \
    }                                                                           \
  } while (0)

// row count of a launch: the host value, or — capacity mode — the device value clamped to it
__device__ __forceinline__ int64_t sgnn_dyn_n(int64_t n_host, const int64_t *n_dev) {
  if (!n_dev) return n_host;
  const int64_t v = *n_dev;
  return v < n_host ? (v < 0 ? 0 : v) : n_host;
}

static inline int sgnn_grid_for(int64_t work, int block, int cap = 1 << 20) {
  int64_t g = (work + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// ---------------------------------------------------------------------------
// XCD-aware tile order.  The dispatcher places workgroup b on XCD b % 8 (observed, used for speed only);
// each XCD has a private 4 MiB L2.  Consecutive row tiles gather overlapping feature rows, so give every
// XCD one contiguous range of tiles: tile = (b % 8) * chunk + b / 8 (bijective for any grid size).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned sgnn_xcd_tile(unsigned b, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7u, xcd = b & 7u, i = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

// ---------------------------------------------------------------------------
// voxel keys / hash
// ---------------------------------------------------------------------------
#define SGNN_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ uint64_t sgnn_pack_key(int z, int y, int x, int b) {
  return ((uint64_t)(uint32_t)b << 48) | ((uint64_t)(uint32_t)z << 32) | ((uint64_t)(uint32_t)y << 16) |
         (uint64_t)(uint32_t)x;
}

__device__ __forceinline__ uint64_t sgnn_hash64(uint64_t k) {
  // murmur3 finaliser: spreads raster-adjacent keys over the table
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return k;
}

__device__ __forceinline__ int32_t sgnn_hash_find(const uint64_t *__restrict__ keys,
                                                  const int32_t *__restrict__ vals, uint64_t mask,
                                                  uint64_t key) {
  uint64_t slot = sgnn_hash64(key) & mask;
  while (true) {
    uint64_t k = keys[slot];
    if (k == key) return vals[slot];
    if (k == SGNN_EMPTY_KEY) return -1;
    slot = (slot + 1) & mask;
  }
}

// ---------------------------------------------------------------------------
// wave64 / block prefix sums over a 0/1 flag (ballot + popcount)
// ---------------------------------------------------------------------------
// Exclusive rank of `flag` among the 256 threads of the block and the block total.
// `lds4` = 4 ints of shared memory.  Contains two barriers.
__device__ __forceinline__ int sgnn_block_rank256(bool flag, int *lds4, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long bal = __ballot(flag);
  int prefix = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) lds4[wave] = __popcll(bal);
  __syncthreads();
  int w0 = lds4[0], w1 = lds4[1], w2 = lds4[2], w3 = lds4[3];
  __syncthreads();
  int base = (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
  total = w0 + w1 + w2 + w3;
  return base + prefix;
}
