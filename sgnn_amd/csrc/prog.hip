// Sparse-network program executor: runs a whole static sub-network (an encoder level stack, a
// FullyConvolutionalNet U, ...) forward or backward from ONE call, launching the same kernels in the
// same order as the per-layer path (bit-identical results) without a host round trip per layer.
//
// Counterpart of what upstream does with Python nn.Module containers (scn.Sequential / ConcatTable /
// AddTable / JoinTable composing <Op>_updateOutput calls): the reference's model.py builds
// those containers at torch/model.py:31-47, 178-188, 253-257; here the container tree is compiled once into
// a flat op list (sgnn_amd/scn/program.py) and interpreted natively.  Host-side code only: no kernels here.
#include <vector>
#include "common.h"

enum { OP_CONV_SUBM = 0, OP_CONV_DOWN = 1, OP_UNPOOL = 2, OP_BN = 3, OP_ADD = 4, OP_JOIN = 5 };

namespace {

struct View {
  const int32_t *ops;   // nops x 8: type, in0, in1, out, param, level, cin, cout
  const float *opf;     // nops x 4: eps, momentum, leak, unused
  int nops;
  const int32_t *bufs;  // nbuf x 2: level, channels
  int nbuf;
  const int64_t *lev_n, *lev_ld;
  void *const *lev_nbr, *const *lev_children, *const *lev_ptable, *const *lev_parent;
  int nlev;
};

inline int64_t round64(int64_t v) { return (v + 63) & ~int64_t(63); }

// arena layout: [buffers..., BN save areas (2*C floats per BN op)..., 2 scratch buffers (backward only)]
struct Layout {
  std::vector<int64_t> buf_off, buf_floats, aux_off;
  int64_t max_buf = 0, total = 0, scratch0 = 0, scratch1 = 0;
};

int make_layout(const View &v, Layout &L) {
  L.buf_off.resize(v.nbuf);
  L.buf_floats.resize(v.nbuf);
  L.aux_off.assign(v.nops, -1);
  int64_t off = 0;
  for (int b = 0; b < v.nbuf; ++b) {
    const int lev = v.bufs[2 * b], ch = v.bufs[2 * b + 1];
    if (lev < 0 || lev >= v.nlev || ch < 1) return -1;
    L.buf_floats[b] = v.lev_n[lev] * ch;
    L.buf_off[b] = off;
    off += round64(L.buf_floats[b]);
    if (L.buf_floats[b] > L.max_buf) L.max_buf = L.buf_floats[b];
  }
  for (int i = 0; i < v.nops; ++i)
    if (v.ops[8 * i] == OP_BN) {
      L.aux_off[i] = off;
      off += round64(2 * (int64_t)v.ops[8 * i + 6]);
    }
  L.scratch0 = off;
  off += round64(L.max_buf);
  L.scratch1 = off;
  off += round64(L.max_buf);
  L.total = off;
  return 0;
}

int64_t ws_main(const View &v) {
  int64_t need = 0;
  for (int i = 0; i < v.nops; ++i) {
    const int32_t *o = v.ops + 8 * i;
    int64_t w = 0;
    if (o[0] == OP_CONV_SUBM) w = sgnn_conv_bwd_weight_ws_bytes(v.lev_n[o[5]], 27, o[6], o[7]);
    if (o[0] == OP_CONV_DOWN) w = sgnn_conv_bwd_weight_ws_bytes(v.lev_n[o[5] + 1], 8, o[6], o[7]);
    if (o[0] == OP_BN) w = sgnn_bn_ws_bytes(v.lev_n[o[5]], o[6]);
    if (w > need) need = w;
  }
  return (need + 255) & ~int64_t(255);
}

// statistics partials a convolution epilogue hands to the neighbouring BatchNorm: [grid blocks][2][C] doubles,
// placed behind the main workspace (the BatchNorm kernels use the main part while they read these)
int64_t ws_stats(const View &v) {
  int64_t need = 0;
  for (int i = 0; i < v.nops; ++i) {
    const int32_t *o = v.ops + 8 * i;
    if (o[0] != OP_CONV_SUBM && o[0] != OP_CONV_DOWN) continue;
    const int64_t rows_f = v.lev_n[o[5]], rows_o = o[0] == OP_CONV_DOWN ? v.lev_n[o[5] + 1] : rows_f;
    const int64_t a = sgnn_conv_grid_blocks(rows_o > 0 ? rows_o : 1) * 2 * o[7] * (int64_t)sizeof(double);  // forward: out rows x cout
    const int64_t b = sgnn_conv_grid_blocks(rows_f > 0 ? rows_f : 1) * 2 * o[6] * (int64_t)sizeof(double);  // data gradient: in rows x cin
    if (a > need) need = a;
    if (b > need) need = b;
  }
  return (need + 255) & ~int64_t(255);
}

int64_t ws_need(const View &v) { return ws_main(v) + ws_stats(v); }

// number of ops that read buffer b
std::vector<int> count_readers(const View &v) {
  std::vector<int> r(v.nbuf, 0);
  for (int i = 0; i < v.nops; ++i) {
    const int32_t *o = v.ops + 8 * i;
    if (o[1] >= 0 && o[1] < v.nbuf) ++r[o[1]];
    if ((o[0] == OP_ADD || o[0] == OP_JOIN) && o[2] >= 0 && o[2] < v.nbuf) ++r[o[2]];
  }
  return r;
}

bool g_fuse = true;   // sgnn_prog_set_fusion: A/B switch for the epilogue fusions (tests, measurements)

// weight-gradient lane: sgnn_prog_backward can run every dW (+ its reduce) on a second stream with its own
// workspace, concurrently with the dX / BatchNorm chain that forms the critical path (both only READ dy)
struct SideLane {
  hipStream_t stream = nullptr;
  void *ws = nullptr;
  int64_t ws_bytes = 0;
  hipEvent_t fork = nullptr, join = nullptr;
} g_side;

int64_t dw_ws_need(const View &v) {
  int64_t need = 0;
  for (int i = 0; i < v.nops; ++i) {
    const int32_t *o = v.ops + 8 * i;
    int64_t w = 0;
    if (o[0] == OP_CONV_SUBM) w = sgnn_conv_bwd_weight_ws_bytes(v.lev_n[o[5]], 27, o[6], o[7]);
    if (o[0] == OP_CONV_DOWN) w = sgnn_conv_bwd_weight_ws_bytes(v.lev_n[o[5] + 1], 8, o[6], o[7]);
    if (w > need) need = w;
  }
  return need;
}

}  // namespace

// stream2 == NULL switches the lane off.  ws2 must not be used by anything else while a backward call is in flight.
SGNN_EXPORT int sgnn_prog_set_side_stream(sgnn_stream_t stream2, void *ws2, int64_t ws2_bytes) {
  if (stream2 && !g_side.fork) {
    SGNN_HIP_TRY(hipEventCreateWithFlags(&g_side.fork, hipEventDisableTiming));
    SGNN_HIP_TRY(hipEventCreateWithFlags(&g_side.join, hipEventDisableTiming));
  }
  g_side.stream = (hipStream_t)stream2;
  g_side.ws = stream2 ? ws2 : nullptr;
  g_side.ws_bytes = stream2 ? ws2_bytes : 0;
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_prog_set_fusion(int on) {
  const int prev = g_fuse ? 1 : 0;
  g_fuse = on != 0;
  return prev;
}

#define PROG_TRY(call)           \
  do {                           \
    const int rc_ = (call);      \
    if (rc_ != SGNN_OK) return rc_; \
  } while (0)

SGNN_EXPORT int64_t sgnn_prog_arena_floats(const int32_t *ops, int nops, const int32_t *bufs, int nbuf,
                                           const int64_t *lev_n, int nlev) {
  View v{ops, nullptr, nops, bufs, nbuf, lev_n, nullptr, nullptr, nullptr, nullptr, nullptr, nlev};
  Layout L;
  if (make_layout(v, L) != 0) return -1;
  return L.total;
}

SGNN_EXPORT int64_t sgnn_prog_ws_bytes(const int32_t *ops, int nops, const int64_t *lev_n, int nlev) {
  View v{ops, nullptr, nops, nullptr, 0, lev_n, nullptr, nullptr, nullptr, nullptr, nullptr, nlev};
  return ws_need(v);
}

// byte offset of buffer `b` inside an arena (so the host layer can hand out views)
SGNN_EXPORT int64_t sgnn_prog_buffer_offset(const int32_t *ops, int nops, const int32_t *bufs, int nbuf,
                                            const int64_t *lev_n, int nlev, int b) {
  View v{ops, nullptr, nops, bufs, nbuf, lev_n, nullptr, nullptr, nullptr, nullptr, nullptr, nlev};
  Layout L;
  if (make_layout(v, L) != 0 || b < 0 || b >= nbuf) return -1;
  return L.buf_off[b];
}

SGNN_EXPORT int sgnn_prog_forward(const int32_t *ops, const float *opf, int nops, const int32_t *bufs, int nbuf,
                                  const int64_t *lev_n, const int64_t *lev_ld, void *const *lev_nbr,
                                  void *const *lev_children, void *const *lev_ptable, void *const *lev_parent,
                                  int nlev, void *const *params, int nparams, const float *input, float *arena,
                                  int64_t arena_floats, const int32_t *keep, int training, void *ws, int64_t ws_bytes,
                                  sgnn_stream_t stream) {
  SGNN_CHECK_ARG(ops && opf && bufs && lev_n && lev_ld && params && arena && nops >= 0 && nbuf >= 1 && nlev >= 1);
  View v{ops, opf, nops, bufs, nbuf, lev_n, lev_ld, lev_nbr, lev_children, lev_ptable, lev_parent, nlev};
  Layout L;
  SGNN_CHECK_ARG(make_layout(v, L) == 0);
  if (arena_floats < L.total) {
    sgnn_set_error("sgnn_prog_forward: arena too small (%lld < %lld floats)", (long long)arena_floats,
                   (long long)L.total);
    return SGNN_ENOWS;
  }
  if (ws_bytes < ws_need(v)) {
    sgnn_set_error("sgnn_prog_forward: workspace too small");
    return SGNN_ENOWS;
  }
  // buffer 0 (the program input) may live outside the arena
  auto B = [&](int b) { return (b == 0 && input) ? const_cast<float *>(input) : arena + L.buf_off[b]; };
  auto P = [&](int p) { return (p >= 0 && p < nparams) ? (float *)params[p] : nullptr; };
  // Epilogue fusions (same arithmetic, fewer passes and launches):
  //  * conv -> AddTable: the convolution adds the other AddTable input while it stores (the sum buffer is written
  //    directly, the convolution's own output buffer stays untouched) when nothing else reads the convolution output;
  //  * conv [-> AddTable] -> BatchNorm (training): the convolution epilogue reduces the column sums the BatchNorm
  //    statistics pass would recompute from HBM.
  std::vector<int> readers = count_readers(v);
  if (keep)
    for (int b = 0; b < nbuf; ++b)
      if (keep[b]) ++readers[b];
  std::vector<char> skip(nops, 0);
  std::vector<const double *> pre(nops, nullptr);
  std::vector<int64_t> pre_nblk(nops, 0);
  double *stats_ws = (double *)((char *)ws + ws_main(v));
  for (int i = 0; i < nops; ++i) {
    const int32_t *o = ops + 8 * i;
    const int type = o[0], in0 = o[1], in1 = o[2], out = o[3], par = o[4], lev = o[5], cin = o[6], cout = o[7];
    SGNN_CHECK_ARG(in0 >= 0 && in0 < nbuf && out >= 0 && out < nbuf && lev >= 0 && lev < nlev);
    const int64_t n = lev_n[lev];
    if (skip[i]) continue;
    switch (type) {
      case OP_CONV_SUBM:
      case OP_CONV_DOWN: {
        const bool down = type == OP_CONV_DOWN;
        SGNN_CHECK_ARG(!down || lev + 1 < nlev);
        const int64_t n_out = down ? lev_n[lev + 1] : n;
        const int32_t *table = (const int32_t *)(down ? lev_children[lev] : lev_nbr[lev]);
        const int64_t ld = down ? lev_ld[lev + 1] : lev_ld[lev];
        ConvEpi epi{};
        float *dst = B(out);
        int dst_buf = out;
        if (g_fuse && sgnn_conv_epi_supported(cin, cout) && n_out > 0) {
          int j = i + 1;
          if (j < nops && ops[8 * j] == OP_ADD && readers[out] == 1 && (ops[8 * j + 1] == out || ops[8 * j + 2] == out) &&
              ops[8 * j + 1] != ops[8 * j + 2]) {
            const int other = ops[8 * j + 1] == out ? ops[8 * j + 2] : ops[8 * j + 1];
            epi.addend = B(other);
            dst_buf = ops[8 * j + 3];
            dst = B(dst_buf);
            skip[j] = 1;
            ++j;
          }
          if (training && j < nops && ops[8 * j] == OP_BN && ops[8 * j + 1] == dst_buf) {
            epi.stats = 1;
            epi.partial = stats_ws;
            pre[j] = stats_ws;
            pre_nblk[j] = sgnn_conv_grid_blocks(n_out);
          }
        }
        PROG_TRY(sgnn_conv_fwd_impl(B(in0), n, cin, P(par), down ? 8 : 27, table, ld, n_out, cout, dst, 0, 0, nullptr,
                                    nullptr, 1, 1, down ? 8 : 27, &epi, stream));
        break;
      }
      case OP_UNPOOL:  // in0 lives on level lev+1, out on level lev
        SGNN_CHECK_ARG(lev + 1 < nlev);
        PROG_TRY(sgnn_gather_rows(B(in0), cin, (const int32_t *)lev_parent[lev], n, B(out), stream));
        break;
      case OP_BN: {
        float *save = arena + L.aux_off[i];
        PROG_TRY(sgnn_bn_fwd_impl(B(in0), cin, n, cin, P(par), P(par + 1), P(par + 2), P(par + 3), opf[4 * i],
                                  opf[4 * i + 1], training, opf[4 * i + 2], save, save + cin, B(out), cin, pre[i],
                                  pre_nblk[i], ws, ws_bytes, stream));
        break;
      }
      case OP_ADD:
        SGNN_CHECK_ARG(in1 >= 0 && in1 < nbuf);
        PROG_TRY(sgnn_add(B(in0), B(in1), n * cin, B(out), stream));
        break;
      case OP_JOIN:  // cin = channels of in0, cout = channels of in1
        SGNN_CHECK_ARG(in1 >= 0 && in1 < nbuf);
        PROG_TRY(sgnn_concat_rows(B(in0), cin, nullptr, B(in1), cout, nullptr, n, B(out), stream));
        break;
      default:
        sgnn_set_error("sgnn_prog_forward: unknown op %d", type);
        return SGNN_EINVAL;
    }
  }
  return SGNN_OK;
}

// garena has the same layout as arena; ginit[b] != 0 marks gradient buffers the caller already filled
// (the program outputs).  need_input_grad: whether buffer 0's gradient is wanted.
SGNN_EXPORT int sgnn_prog_backward(const int32_t *ops, const float *opf, int nops, const int32_t *bufs, int nbuf,
                                   const int64_t *lev_n, const int64_t *lev_ld, void *const *lev_nbr,
                                   void *const *lev_children, void *const *lev_ptable, void *const *lev_parent,
                                   int nlev, void *const *params, void *const *pgrads, int nparams,
                                   const float *input, const float *arena, float *garena, int64_t arena_floats,
                                   const int32_t *ginit,
                                   int need_input_grad, int training, void *ws, int64_t ws_bytes,
                                   sgnn_stream_t stream) {
  SGNN_CHECK_ARG(ops && opf && bufs && lev_n && lev_ld && params && pgrads && arena && garena && ginit);
  View v{ops, opf, nops, bufs, nbuf, lev_n, lev_ld, lev_nbr, lev_children, lev_ptable, lev_parent, nlev};
  Layout L;
  SGNN_CHECK_ARG(make_layout(v, L) == 0);
  if (arena_floats < L.total || ws_bytes < ws_need(v)) {
    sgnn_set_error("sgnn_prog_backward: arena or workspace too small");
    return SGNN_ENOWS;
  }
  hipStream_t hs = (hipStream_t)stream;
  // gradient state of a buffer: 0 nothing yet, 1 G(b) holds it, 2 it EQUALS the gradient of buffer alias[b]
  // (an AddTable input whose only contribution so far is the sum's gradient: nothing is copied until something
  // has to be added to it, and a reader just follows the alias)
  std::vector<char> init(nbuf);
  std::vector<int> alias(nbuf, -1);
  for (int b = 0; b < nbuf; ++b) init[b] = ginit[b] != 0;
  auto X = [&](int b) { return (b == 0 && input) ? input : arena + L.buf_off[b]; };
  auto G = [&](int b) { return garena + L.buf_off[b]; };
  auto GR = [&](int b) -> const float * { return init[b] == 2 ? G(alias[b]) : G(b); };   // where b's gradient is read
  auto P = [&](int p) { return (p >= 0 && p < nparams) ? (float *)params[p] : nullptr; };
  auto PG = [&](int p) { return (p >= 0 && p < nparams) ? (float *)pgrads[p] : nullptr; };
  float *scratch[2] = {garena + L.scratch0, garena + L.scratch1};
  // where a kernel should write the gradient of buffer b: the buffer itself unless it already holds data
  auto target = [&](int b, int which) { return init[b] == 1 ? scratch[which] : G(b); };
  auto commit = [&](int b, float *wrote) -> int {  // fold a freshly written gradient into buffer b
    if (wrote == G(b)) {
      if (init[b] == 2) {                            // G(b) = fresh + the aliased gradient
        const int a = alias[b];
        alias[b] = -1;
        init[b] = 1;
        return sgnn_add(G(b), G(a), L.buf_floats[b], G(b), stream);
      }
      init[b] = 1;
      return SGNN_OK;
    }
    return sgnn_add(G(b), wrote, L.buf_floats[b], G(b), stream);
  };
  auto wants = [&](int b) { return b != 0 || need_input_grad; };
  // dW launches go to the side lane when one is configured and its workspace is big enough
  const bool side = g_side.stream && g_side.stream != hs && g_side.ws && g_side.ws_bytes >= dw_ws_need(v);
  bool forked = false;
  auto dw_lane = [&]() -> hipStream_t {
    if (!side) return hs;
    (void)hipEventRecord(g_side.fork, hs);                 // dy of this op is final here (all its consumers ran)
    (void)hipStreamWaitEvent(g_side.stream, g_side.fork, 0);
    forked = true;
    return g_side.stream;
  };
  void *dw_ws = side ? g_side.ws : ws;
  const int64_t dw_ws_bytes = side ? g_side.ws_bytes : ws_main(v);
  std::vector<const double *> pre(nops, nullptr);
  std::vector<int64_t> pre_nblk(nops, 0);
  double *stats_ws = (double *)((char *)ws + ws_main(v));

  for (int i = nops - 1; i >= 0; --i) {
    const int32_t *o = ops + 8 * i;
    const int type = o[0], in0 = o[1], in1 = o[2], out = o[3], par = o[4], lev = o[5], cin = o[6], cout = o[7];
    const int64_t n = lev_n[lev];
    if (!init[out]) {  // no gradient reached this output: its producers contribute nothing
      if (type == OP_CONV_SUBM || type == OP_CONV_DOWN)
        SGNN_HIP_TRY(hipMemsetAsync(PG(par), 0, (size_t)(type == OP_CONV_SUBM ? 27 : 8) * cin * cout * sizeof(float), hs));
      if (type == OP_BN) {
        if (PG(par)) SGNN_HIP_TRY(hipMemsetAsync(PG(par), 0, cin * sizeof(float), hs));
        if (PG(par + 1)) SGNN_HIP_TRY(hipMemsetAsync(PG(par + 1), 0, cin * sizeof(float), hs));
      }
      continue;
    }
    const float *dy = GR(out);
    switch (type) {
      case OP_CONV_SUBM:
      case OP_CONV_DOWN: {
        const bool down = type == OP_CONV_DOWN;
        const int K = down ? 8 : 27;
        const int64_t n_dy = down ? lev_n[lev + 1] : n;      // rows of dy (= rows of the forward output)
        const int32_t *tab_f = (const int32_t *)(down ? lev_children[lev] : lev_nbr[lev]);
        const int64_t ld_f = down ? lev_ld[lev + 1] : lev_ld[lev];
        const int32_t *tab_b = (const int32_t *)(down ? lev_ptable[lev] : lev_nbr[lev]);
        const int flags_b = down ? SGNN_CONV_TRANSPOSE_W : (SGNN_CONV_TRANSPOSE_W | SGNN_CONV_FLIP_K);
        const hipStream_t lane = dw_lane();
        if (wants(in0)) {
          if (g_fuse && sgnn_conv_epi_supported(cout, cin) && n > 0) {
            // the data gradient lands in G(in0) directly: what the buffer (or its alias) already holds is added in the
            // store (in place), and when in0 is the output of the BatchNormReLU right before this op and this is the
            // last contribution to its gradient, the epilogue also reduces sum dz / sum dz*xhat for that BatchNorm
            ConvEpi epi{};
            if (init[in0] == 1) epi.addend = G(in0);
            else if (init[in0] == 2) epi.addend = G(alias[in0]);
            if (i > 0 && ops[8 * (i - 1)] == OP_BN && ops[8 * (i - 1) + 3] == in0 && ops[8 * (i - 1) + 6] == cin) {
              const int32_t *bo = ops + 8 * (i - 1);
              const float *save = arena + L.aux_off[i - 1];
              epi.stats = 2;
              epi.partial = stats_ws;
              epi.bn_x = X(bo[1]);
              epi.mean = save;
              epi.invstd = save + cin;
              epi.gamma = P(bo[4]);
              epi.beta = P(bo[4] + 1);
              epi.leak = opf[4 * (i - 1) + 2];
              pre[i - 1] = stats_ws;
              pre_nblk[i - 1] = sgnn_conv_grid_blocks(n);
            }
            PROG_TRY(sgnn_conv_fwd_impl(dy, n_dy, cout, P(par), K, tab_b, lev_ld[lev], n, cin, G(in0), flags_b, 0, nullptr,
                                        nullptr, 1, 1, K, &epi, stream));
            init[in0] = 1;
            alias[in0] = -1;
          } else {
            float *t = target(in0, 0);
            PROG_TRY(sgnn_conv_fwd(dy, n_dy, cout, P(par), K, tab_b, lev_ld[lev], n, cin, t, flags_b, 0, stream));
            PROG_TRY(commit(in0, t));
          }
        }
        PROG_TRY(sgnn_conv_bwd_weight(X(in0), n, cin, dy, cout, tab_f, ld_f, K, n_dy, PG(par), 0, dw_ws, dw_ws_bytes,
                                      (sgnn_stream_t)lane));
        break;
      }
      case OP_UNPOOL:
        if (wants(in0)) {
          float *t = target(in0, 0);
          PROG_TRY(sgnn_gather_sum(dy, cin, (const int32_t *)lev_children[lev], lev_ld[lev + 1], 8, lev_n[lev + 1], t,
                                   stream));
          PROG_TRY(commit(in0, t));
        }
        break;
      case OP_BN: {
        const float *save = arena + L.aux_off[i];
        // the kernel adds what the buffer already holds (in place) or the aliased gradient: no scratch pass, no k_add
        const float *addend = !wants(in0) ? nullptr : (init[in0] == 1 ? G(in0) : (init[in0] == 2 ? G(alias[in0]) : nullptr));
        float *t = wants(in0) ? G(in0) : scratch[0];
        PROG_TRY(sgnn_bn_bwd_impl(X(in0), cin, dy, cin, n, cin, P(par), P(par + 1), save, save + cin, training,
                                  opf[4 * i + 2], addend, cin, t, cin, PG(par), PG(par + 1), pre[i], pre_nblk[i], ws, ws_bytes,
                                  stream));
        if (wants(in0)) {
          init[in0] = 1;
          alias[in0] = -1;
        }
        break;
      }
      case OP_ADD: {
        const int src = init[out] == 2 ? alias[out] : out;      // the buffer that physically holds dy
        for (int side = 0; side < 2; ++side) {
          const int b = side ? in1 : in0;
          if (!wants(b)) continue;
          if (init[b] == 1) {
            PROG_TRY(sgnn_add(G(b), dy, L.buf_floats[b], G(b), stream));
          } else if (init[b] == 2) {                            // two aliased contributions: materialise the sum
            PROG_TRY(sgnn_add(G(alias[b]), dy, L.buf_floats[b], G(b), stream));
            init[b] = 1;
            alias[b] = -1;
          } else {
            init[b] = 2;
            alias[b] = src;
          }
        }
        break;
      }
      case OP_JOIN: {
        float *ta = wants(in0) ? target(in0, 0) : nullptr;
        float *tb = wants(in1) ? target(in1, 1) : nullptr;
        PROG_TRY(sgnn_concat_rows_bwd(dy, cin, nullptr, cout, nullptr, n, ta, n, tb, n, stream));
        if (ta) PROG_TRY(commit(in0, ta));
        if (tb) PROG_TRY(commit(in1, tb));
        break;
      }
      default:
        sgnn_set_error("sgnn_prog_backward: unknown op %d", type);
        return SGNN_EINVAL;
    }
  }
  if (need_input_grad && init[0] == 2)                // the caller reads G(0): an alias has to become a copy
    SGNN_HIP_TRY(hipMemcpyAsync(G(0), G(alias[0]), (size_t)L.buf_floats[0] * sizeof(float), hipMemcpyDeviceToDevice, hs));
  if (forked) {                                       // parameter gradients are complete once the lane has drained
    SGNN_HIP_TRY(hipEventRecord(g_side.join, g_side.stream));
    SGNN_HIP_TRY(hipStreamWaitEvent(hs, g_side.join, 0));
  }
  return SGNN_OK;
}
