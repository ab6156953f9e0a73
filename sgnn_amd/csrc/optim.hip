// Optimizer step of the training loop as ONE launch over flat parameter / gradient / moment buffers
// (torch/train.py:81 `optim.Adam`, :264 `optimizer.step()`; SURVEY.md §8 row a-H, the step around the hot path).
//
// torch's fused multi-tensor Adam needs 8 launches for SG-NN's 307 parameter tensors and a host-side list walk;
// here the 643 735 parameters live in one buffer (train.FlatAdam re-points every nn.Parameter at a view of it), so
// the update is a single HBM stream: 4 reads + 3 writes of 2.6 MB.  Two things make the launch graph-capturable
// with the rest of the step (HISTORY.md §2, capacity mode):
//   * segments — the flat buffer is cut into [begin, end) ranges (encoder, each generative stage); a segment is
//     updated only if it was reached this step (its stage had at least one input site: *cnt > 0, or the
//     all-reduced flag > 0 under data parallelism).  This is torch.optim.Adam's "skip parameters whose grad is None"
//     (the reference's empty-level early return, torch/model.py:211,260) decided on the device, without a read-back;
//   * the status word — if any kernel of the step flagged a capacity overflow (SGNN_STATUS_OVERFLOW) nothing is
//     updated: the step is all-or-nothing, the host re-runs the batch with larger capacities.
// lr, and the per-segment step counters, are device scalars so that neither a scheduler nor the step count bakes a
// constant into a captured graph.
#include "common.h"

#define ADAM_MAX_SEG 8

struct AdamSeg {
  int64_t begin, end;
  const int64_t *cnt;   // device row count that decides "reached" (NULL: always)
  const float *flag;    // or: device float, reached iff > 0 (takes precedence over cnt)
  float *step;          // device float: number of updates this segment has seen (torch's state['step'])
};
struct AdamArgs {
  AdamSeg seg[ADAM_MAX_SEG];
  int nseg;
};

__device__ __forceinline__ bool seg_active(const AdamSeg &s) {
  if (s.flag) return *s.flag > 0.f;
  if (s.cnt) return *s.cnt > 0;
  return true;
}

// one workgroup column per 1024 elements; block -> segment by scanning the (<= 8) ranges
__global__ __launch_bounds__(256) void k_adam_flat(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v, int64_t n,
                                                   AdamArgs a, const float *__restrict__ lr_dev, float beta1,
                                                   float beta2, float eps, float weight_decay, float grad_scale,
                                                   const int32_t *__restrict__ status) {
  if (status && (*status & SGNN_STATUS_OVERFLOW)) return;      // (uniform over the launch: no barrier is skipped by a part of it)
  // the bias corrections depend on the segment only: one thread per segment forms them for the workgroup (round 5: the two
  // double-precision pow() per ELEMENT were most of this kernel — 19 us for 2.6 MB of parameters, on the serial tail of the step)
  __shared__ float s_bc1[ADAM_MAX_SEG], s_bc2s[ADAM_MAX_SEG];
  __shared__ int s_on[ADAM_MAX_SEG];
  if (threadIdx.x < ADAM_MAX_SEG) {
    const int t = threadIdx.x;
    float b1 = 1.f, b2 = 1.f;
    int on = 0;
    if (t < a.nseg) {
      on = seg_active(a.seg[t]) ? 1 : 0;
      // the counter still holds the number of PREVIOUS updates (k_adam_steps bumps it after this kernel)
      const double step = (double)(*a.seg[t].step) + 1.0;
      b1 = (float)(1.0 - pow((double)beta1, step));
      b2 = sqrtf((float)(1.0 - pow((double)beta2, step)));
    }
    s_bc1[t] = b1;
    s_bc2s[t] = b2;
    s_on[t] = on;
  }
  __syncthreads();
  const float lr = *lr_dev;
  const int64_t stride = (int64_t)gridDim.x * 256 * 4;
  for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < n; i0 += stride) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t i = i0 + u;
      if (i >= n) break;
      int si = -1;
#pragma unroll
      for (int t = 0; t < ADAM_MAX_SEG; ++t)
        if (t < a.nseg && i >= a.seg[t].begin && i < a.seg[t].end) si = t;
      if (si < 0 || !s_on[si]) continue;
      const float bc1 = s_bc1[si];
      const float bc2s = s_bc2s[si];
      float grad = g[i] * grad_scale;
      float par = p[i];
      if (weight_decay != 0.f) grad += par * weight_decay;
      float ea = m[i], es = v[i];
      ea = ea + (grad - ea) * (1.f - beta1);
      es = beta2 * es + (1.f - beta2) * grad * grad;
      const float step_size = lr / bc1;
      const float denom = sqrtf(es) / bc2s + eps;
      par -= step_size * ea / denom;
      p[i] = par;
      m[i] = ea;
      v[i] = es;
    }
  }
}

__global__ void k_adam_steps(AdamArgs a, const int32_t *__restrict__ status) {
  if (status && (*status & SGNN_STATUS_OVERFLOW)) return;
  const int t = threadIdx.x;
  if (t < a.nseg && seg_active(a.seg[t])) *a.seg[t].step += 1.f;
}

// seg: HOST array of nseg x 5 int64 = {begin, end, cnt pointer, flag pointer, step pointer}
SGNN_EXPORT int sgnn_adam_flat(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                               const int64_t *seg, int nseg, const float *lr_dev, float beta1, float beta2, float eps,
                               float weight_decay, float grad_scale, const int32_t *status, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && n >= 0 && seg && nseg >= 1 && nseg <= ADAM_MAX_SEG && lr_dev);
  if (n == 0) return SGNN_OK;
  AdamArgs a{};
  a.nseg = nseg;
  for (int t = 0; t < nseg; ++t) {
    a.seg[t].begin = seg[5 * t];
    a.seg[t].end = seg[5 * t + 1];
    a.seg[t].cnt = (const int64_t *)(uintptr_t)seg[5 * t + 2];
    a.seg[t].flag = (const float *)(uintptr_t)seg[5 * t + 3];
    a.seg[t].step = (float *)(uintptr_t)seg[5 * t + 4];
    SGNN_CHECK_ARG(a.seg[t].begin >= 0 && a.seg[t].end >= a.seg[t].begin && a.seg[t].end <= n && a.seg[t].step);
  }
  hipStream_t s = (hipStream_t)stream;
  SGNN_LAUNCH(k_adam_flat, dim3(sgnn_grid_for((n + 3) / 4, 256, 1024)), dim3(256), 0, s, params, grads, exp_avg,
                     exp_avg_sq, n, a, lr_dev, beta1, beta2, eps, weight_decay, grad_scale, status);
  SGNN_LAUNCH(k_adam_steps, dim3(1), dim3(64), 0, s, a, status);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// flags[t] = 1 if *cnt[t] > 0 (cnt[t] NULL: 1) — what a data-parallel step appends to its flat gradient buffer so
// that the all-reduce also tells every rank which segments ANY rank reached (train.FlatGradAllReduce's flags)
struct SegCnt {
  const int64_t *cnt[ADAM_MAX_SEG];
};
__global__ void k_seg_flags(SegCnt c, int nseg, float *__restrict__ flags, const int32_t *__restrict__ status) {
  const int t = threadIdx.x;
  if (t < nseg) flags[t] = (!c.cnt[t] || *c.cnt[t] > 0) ? 1.f : 0.f;
  if (t == ADAM_MAX_SEG - 1 && status) flags[ADAM_MAX_SEG - 1] = (*status & SGNN_STATUS_OVERFLOW) ? 1.f : 0.f;
}

// status (may be NULL): flags[7] = 1 if this rank's step overflowed a capacity (nseg <= 7 then)
SGNN_EXPORT int sgnn_seg_flags(const int64_t *cnt_ptrs, int nseg, float *flags, const int32_t *status,
                               sgnn_stream_t stream) {
  SGNN_CHECK_ARG(cnt_ptrs && flags && nseg >= 1 && nseg <= ADAM_MAX_SEG && (!status || nseg < ADAM_MAX_SEG));
  SegCnt c{};
  for (int t = 0; t < nseg; ++t) c.cnt[t] = (const int64_t *)(uintptr_t)cnt_ptrs[t];
  SGNN_LAUNCH(k_seg_flags, dim3(1), dim3(64), 0, (hipStream_t)stream, c, nseg, flags, status);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// after the all-reduce of the flags: an overflow on ANY rank becomes this rank's overflow, so that every replica skips
// the same optimizer step and re-runs the same batch (data-parallel replicas must take identical decisions)
__global__ void k_status_merge(const float *__restrict__ flag, int32_t *__restrict__ status) {
  if (threadIdx.x == 0 && *flag > 0.f) atomicOr(status, SGNN_STATUS_OVERFLOW);
}

SGNN_EXPORT int sgnn_status_merge(const float *overflow_flag, int32_t *status, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(overflow_flag && status);
  SGNN_LAUNCH(k_status_merge, dim3(1), dim3(64), 0, (hipStream_t)stream, overflow_flag, status);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}
