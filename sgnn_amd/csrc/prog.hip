// Sparse-network program executor: runs a whole static sub-network — an encoder level stack, or a complete
// generative stage (skip join -> SubmanifoldConvolution -> FullyConvolutionalNet U -> BatchNormReLU -> 8-child
// up-sampling convolution -> BatchNormReLU -> the two linear heads) — forward or backward from ONE call.
//
// Counterpart of what upstream does with Python nn.Module containers (scn.Sequential / ConcatTable / AddTable /
// JoinTable composing <Op>_updateOutput calls): the reference's model.py builds those containers at
// torch/model.py:31-47, 178-191, 253-258 and glues them with tensor ops at :209-247, 259-272, 338-355; here the tree
// is compiled once into a flat op list (sgnn_amd/scn/program.py) and interpreted natively.  A training step is
// host-bound (every microsecond of host time shows up in the step time, profiles/r02_host_bound.txt): one native call
// per stage and direction replaces ~12 Python autograd nodes and their tensor allocations.
// Host-side code only: no kernels here.
#include <mutex>
#include <vector>
#include "common.h"

enum {
  OP_CONV_SUBM = 0, OP_CONV_DOWN = 1, OP_UNPOOL = 2, OP_BN = 3, OP_ADD = 4, OP_JOIN = 5,
  OP_CONCAT_IN = 6,   // out = [in0[ia] | in1[ib] | in2[ic]]  (index arrays optional; inputs may be externals)
  OP_EXPAND = 7,      // SubmanifoldConvolution over the 8-child expansion, on the parent rulebook (sgnn_conv_fwd_ex)
  OP_LINEAR = 8       // cout (1 or 2) per-site heads; weight row o = param slot par + 2*o, its bias par + 2*o + 1
};
#define OPW 12   // ints per op: type, in0, in1, out, par, lev, cin, cout, in2, ia, ib, ic
#define EXPAND_DX_SPLIT 4

// conv.hip entry points without a prototype in the public header
extern "C" int sgnn_expand_weights(const float *w, int cin, int cout, float *wc, sgnn_stream_t stream);
extern "C" int sgnn_expand_weights_bwd(const float *dwc, int cin, int cout, float *dw, sgnn_stream_t stream);

namespace {

struct View {
  const int32_t *ops;   // nops x OPW
  const float *opf;     // nops x 4: eps, momentum, leak, unused
  int nops;
  const int32_t *bufs;  // nbuf x 2: rows class ("level"), channels
  int nbuf, n_ext;      // buffers [0, n_ext) are caller-owned inputs outside the arena
  const int64_t *lev_n, *lev_ld;
  void *const *lev_nbr, *const *lev_children, *const *lev_ptable, *const *lev_parent;
  int nlev;
};

inline int64_t round64(int64_t v) { return (v + 63) & ~int64_t(63); }

// number of ops that read buffer b
std::vector<int> count_readers(const View &v) {
  std::vector<int> r(v.nbuf, 0);
  auto hit = [&](int b) {
    if (b >= 0 && b < v.nbuf) ++r[b];
  };
  for (int i = 0; i < v.nops; ++i) {
    const int32_t *o = v.ops + OPW * i;
    hit(o[1]);
    if (o[0] == OP_ADD || o[0] == OP_JOIN || o[0] == OP_CONCAT_IN) hit(o[2]);
    if (o[0] == OP_CONCAT_IN) hit(o[8]);
  }
  return r;
}

// (the executor's switches live in the library's one table, sgnn_tune: prog_fusion, prog_lin_add, prog_lin_bn — tune.hip)
#define g_fuse (g_tune.prog_fusion != 0)
#define g_lin_add (g_tune.prog_lin_add != 0)
#define g_lin_bn (g_tune.prog_lin_bn != 0)
// What the executor decides once per call, identically in forward and backward:
//  * add_dst[i] >= 0: convolution i writes straight into the output of the AddTable right behind it (fused add);
//  * views: a JoinTable whose inputs can be produced in place gets no copy — its inputs LIVE in column ranges of the
//    join buffer (root / col / ld), every producer writes and every consumer reads through a row stride.
struct Plan {
  std::vector<int> add_dst;      // per op
  std::vector<char> skip;        // per op: forward launches nothing (fused AddTable, in-place JoinTable)
  std::vector<int> root, col;    // per buffer: storage owner and column offset inside it
  std::vector<int64_t> ld;       // per buffer: row stride in floats
  std::vector<char> join_view;   // per op: this JoinTable is in place
  std::vector<int> lin_bn;       // per op: a LINEAR head whose data gradient is formed inside the backward pass of BatchNorm lin_bn[i] (BnLin; -1: written)
};


void make_plan(const View &v, const int32_t *keep, Plan &P) {
  P.add_dst.assign(v.nops, -1);
  P.skip.assign(v.nops, 0);
  P.join_view.assign(v.nops, 0);
  P.lin_bn.assign(v.nops, -1);
  P.root.resize(v.nbuf);
  P.col.assign(v.nbuf, 0);
  P.ld.resize(v.nbuf);
  for (int b = 0; b < v.nbuf; ++b) {
    P.root[b] = b;
    P.ld[b] = v.bufs[2 * b + 1];
  }
  if (!g_fuse) return;
  std::vector<int> readers = count_readers(v);
  if (keep)
    for (int b = 0; b < v.nbuf; ++b)
      if (keep[b]) ++readers[b];
  // a per-site head that is the ONLY reader of a BatchNormReLU's output: its data gradient is never stored (BnLin)
  if (g_lin_bn)
    for (int i = 0; i < v.nops; ++i) {
      const int32_t *o = v.ops + OPW * i;
      if (o[0] != OP_LINEAR || o[7] > 2 || readers[o[1]] != 1) continue;
      for (int j = 0; j < i; ++j) {
        const int32_t *b = v.ops + OPW * j;
        if (b[0] == OP_BN && b[3] == o[1] && b[5] == o[5]) P.lin_bn[i] = j;
      }
    }
  // fused AddTable
  for (int i = 0; i + 1 < v.nops; ++i) {
    const int32_t *o = v.ops + OPW * i, *a = v.ops + OPW * (i + 1);
    if ((o[0] != OP_CONV_SUBM && o[0] != OP_CONV_DOWN) || a[0] != OP_ADD) continue;
    const int64_t n_out = o[0] == OP_CONV_DOWN ? v.lev_n[o[5] + 1] : v.lev_n[o[5]];
    if (!sgnn_conv_epi_supported(o[6], o[7]) || n_out <= 0 || readers[o[3]] != 1) continue;
    if ((a[1] != o[3] && a[2] != o[3]) || a[1] == a[2]) continue;
    P.add_dst[i] = a[3];
    P.skip[i + 1] = 1;
  }
  // producer op of every buffer (the fused convolution for a fused AddTable output) and its last reader
  std::vector<int> prod(v.nbuf, -1), last_reader(v.nbuf, -1);
  std::vector<char> strided_ok(v.nbuf, 1);   // every reader / writer of the buffer can work through a row stride
  for (int i = 0; i < v.nops; ++i) {
    const int32_t *o = v.ops + OPW * i;
    if (!P.skip[i]) prod[P.add_dst[i] >= 0 ? P.add_dst[i] : o[3]] = i;
    auto reads = [&](int b, bool ok) {
      if (b < 0) return;
      last_reader[b] = i;
      if (!ok) strided_ok[b] = 0;
    };
    switch (o[0]) {
      case OP_CONV_SUBM:
      case OP_CONV_DOWN: reads(o[1], sgnn_conv_epi_supported(o[6], o[7]) && sgnn_conv_epi_supported(o[7], o[6]) && dw_shape_ok(o[6], o[7])); break;
      case OP_BN: reads(o[1], true); break;
      case OP_UNPOOL: reads(o[1], true); break;
      case OP_JOIN:
        reads(o[1], last_reader[o[1]] < 0 || v.ops[OPW * last_reader[o[1]]] != OP_JOIN);   // two JoinTables reading it: no view
        reads(o[2], last_reader[o[2]] < 0 || v.ops[OPW * last_reader[o[2]]] != OP_JOIN);
        break;
      case OP_ADD: reads(o[1], P.skip[i] != 0); reads(o[2], P.skip[i] != 0); break;   // a fused AddTable reads through the conv epilogue
      case OP_CONCAT_IN: reads(o[1], false); reads(o[2], false); reads(o[8], false); break;
      default: reads(o[1], false); break;
    }
  }
  for (int i = 0; i < v.nops; ++i) {
    const int32_t *o = v.ops + OPW * i;
    if (o[0] != OP_JOIN || v.lev_n[o[5]] <= 0) continue;
    bool ok = true;
    for (int side = 0; side < 2 && ok; ++side) {
      const int q = o[1 + side];
      ok = q >= v.n_ext && !(keep && keep[q]) && P.root[q] == q && strided_ok[q] && last_reader[q] == i && prod[q] >= 0;
      if (!ok) break;
      const int32_t *p = v.ops + OPW * prod[q];
      const bool conv = (p[0] == OP_CONV_SUBM || p[0] == OP_CONV_DOWN);
      // writers that can store through a stride: conv epilogue (compiled shapes), BatchNorm apply, UnPooling gather;
      // in backward the producer reads the buffer's gradient through the same stride (dX conv, dW, BN, gather_sum)
      ok = (conv && sgnn_conv_epi_supported(p[6], p[7]) && sgnn_conv_epi_supported(p[7], p[6]) && dw_shape_ok(p[6], p[7])) ||
           p[0] == OP_BN || p[0] == OP_UNPOOL;
    }
    if (!ok || o[1] == o[2]) continue;
    P.join_view[i] = 1;
    P.skip[i] = 1;
    P.root[o[1]] = o[3];
    P.col[o[1]] = 0;
    P.root[o[2]] = o[3];
    P.col[o[2]] = o[6];        // cin = channels of in0
  }
  for (int b = 0; b < v.nbuf; ++b) {   // nested joins: resolve to the outermost storage
    int r = b, c = 0;
    while (P.root[r] != r) {
      c += P.col[r];
      r = P.root[r];
    }
    P.root[b] = r;
    P.col[b] = c;
    P.ld[b] = v.bufs[2 * r + 1];
  }
}

// arena layout: [buffers (in-place JoinTable inputs own no storage)..., per-op areas (BatchNorm: mean/invstd;
// up-sampling conv: its 64 pre-summed weight slices)..., backward scratch: 2 x largest buffer, up-sampling weight
// gradient + its split data-gradient rows]
struct Layout {
  std::vector<int64_t> buf_off, buf_floats, aux_off;
  int64_t max_buf = 0, total = 0, fwd_total = 0, scratch0 = 0, scratch1 = 0, bextra = 0;
};

// Inference layout (no backward pass will read the arena): a buffer's storage is handed to later buffers once its last
// reader has run.  Storage roots are allocated at the first op that writes into them (an in-place JoinTable's inputs
// write into the join buffer), released after the last op that touches them, outputs (`keep`) never; first-fit over an
// offset-ordered free list, so the arena is the high-water mark of the live set instead of the sum of all buffers —
// 3-4x smaller for a FullyConvolutionalNet stage (what bounds whole-scene inference, BASELINE configs[3]).
int make_layout_infer(const View &v, const Plan &P, const int32_t *keep, Layout &L) {
  const int never = v.nops + 1;
  std::vector<int> first(v.nbuf, never), last(v.nbuf, -1);
  auto touch = [&](int b, int i) {
    if (b < v.n_ext || b >= v.nbuf) return;
    const int r = P.root[b];
    if (i < first[r]) first[r] = i;
    if (i > last[r]) last[r] = i;
  };
  for (int i = 0; i < v.nops; ++i) {
    const int32_t *o = v.ops + OPW * i;
    touch(o[1], i);
    if (o[0] == OP_ADD || o[0] == OP_JOIN || o[0] == OP_CONCAT_IN) touch(o[2], i);
    if (o[0] == OP_CONCAT_IN) touch(o[8], i);
    if (P.add_dst[i] >= 0) touch(P.add_dst[i], i);   // fused AddTable: the convolution writes the sum buffer itself
    else touch(o[3], i);
  }
  for (int b = v.n_ext; b < v.nbuf; ++b)
    if (keep && keep[b]) last[P.root[b]] = never;
  int64_t off = 0;
  for (int i = 0; i < v.nops; ++i) {   // per-op areas first: small and alive for the whole call
    const int32_t *o = v.ops + OPW * i;
    if (o[0] == OP_BN) {
      L.aux_off[i] = off;
      off += round64(2 * (int64_t)o[6]);
    } else if (o[0] == OP_EXPAND) {
      L.aux_off[i] = off;
      off += round64(64 * (int64_t)o[6] * o[7]);
    }
  }
  std::vector<std::pair<int64_t, int64_t>> free_list;   // (offset, floats), ordered by offset, neighbours merged
  int64_t top = off;
  auto release = [&](int64_t o, int64_t n) {
    if (n <= 0) return;
    size_t k = 0;
    while (k < free_list.size() && free_list[k].first < o) ++k;
    free_list.insert(free_list.begin() + k, std::make_pair(o, n));
    if (k + 1 < free_list.size() && free_list[k].first + free_list[k].second == free_list[k + 1].first) {
      free_list[k].second += free_list[k + 1].second;
      free_list.erase(free_list.begin() + k + 1);
    }
    if (k > 0 && free_list[k - 1].first + free_list[k - 1].second == free_list[k].first) {
      free_list[k - 1].second += free_list[k].second;
      free_list.erase(free_list.begin() + k);
    }
  };
  auto take = [&](int64_t n) -> int64_t {
    if (n <= 0) return 0;
    size_t best = free_list.size();
    for (size_t k = 0; k < free_list.size(); ++k)
      if (free_list[k].second >= n && (best == free_list.size() || free_list[k].second < free_list[best].second)) best = k;
    if (best < free_list.size()) {
      const int64_t o = free_list[best].first;
      free_list[best].first += n;
      free_list[best].second -= n;
      if (free_list[best].second == 0) free_list.erase(free_list.begin() + best);
      return o;
    }
    if (!free_list.empty() && free_list.back().first + free_list.back().second == top) {   // grow the block at the top
      const int64_t o = free_list.back().first;
      free_list.pop_back();
      top = o + n;
      return o;
    }
    const int64_t o = top;
    top += n;
    return o;
  };
  for (int i = 0; i < v.nops; ++i) {
    for (int b = v.n_ext; b < v.nbuf; ++b)      // everything whose last toucher ran before this op
      if (P.root[b] == b && L.buf_off[b] >= 0 && last[b] == i - 1) release(L.buf_off[b], round64(L.buf_floats[b]));
    for (int b = v.n_ext; b < v.nbuf; ++b)
      if (P.root[b] == b && first[b] == i) L.buf_off[b] = take(round64(L.buf_floats[b]));
  }
  for (int b = v.n_ext; b < v.nbuf; ++b) {
    if (P.root[b] == b && L.buf_off[b] < 0) L.buf_off[b] = 0;   // never touched (a fused-away convolution output)
  }
  for (int b = v.n_ext; b < v.nbuf; ++b)
    if (P.root[b] != b) L.buf_off[b] = L.buf_off[P.root[b]] + P.col[b];
  L.fwd_total = L.total = top;
  L.scratch0 = L.scratch1 = L.bextra = top;
  return 0;
}

int make_layout(const View &v, const Plan &P, Layout &L, bool infer = false, const int32_t *keep = nullptr) {
  L.buf_off.assign(v.nbuf, -1);
  L.buf_floats.resize(v.nbuf);
  L.aux_off.assign(v.nops, -1);
  int64_t off = 0;
  if (infer) {
    for (int b = 0; b < v.nbuf; ++b) {
      const int lev = v.bufs[2 * b], ch = v.bufs[2 * b + 1];
      if (lev < 0 || lev >= v.nlev || ch < 1) return -1;
      L.buf_floats[b] = v.lev_n[lev] * (int64_t)(P.root[b] == b ? P.ld[b] : ch);
      if (L.buf_floats[b] > L.max_buf) L.max_buf = L.buf_floats[b];
    }
    return make_layout_infer(v, P, keep, L);
  }
  for (int b = 0; b < v.nbuf; ++b) {
    const int lev = v.bufs[2 * b], ch = v.bufs[2 * b + 1];
    if (lev < 0 || lev >= v.nlev || ch < 1) return -1;
    L.buf_floats[b] = v.lev_n[lev] * ch;
    if (L.buf_floats[b] > L.max_buf) L.max_buf = L.buf_floats[b];
    if (b < v.n_ext || P.root[b] != b) continue;
    L.buf_off[b] = off;
    off += round64(L.buf_floats[b]);
  }
  for (int b = v.n_ext; b < v.nbuf; ++b)
    if (P.root[b] != b) L.buf_off[b] = L.buf_off[P.root[b]] + P.col[b];
  int64_t bextra = 0;
  for (int i = 0; i < v.nops; ++i) {
    const int32_t *o = v.ops + OPW * i;
    if (o[0] == OP_BN) {
      L.aux_off[i] = off;
      off += round64(2 * (int64_t)o[6]);
    } else if (o[0] == OP_EXPAND) {
      L.aux_off[i] = off;
      off += round64(64 * (int64_t)o[6] * o[7]);
      const int64_t need = round64(64 * (int64_t)o[6] * o[7]) + round64(v.lev_n[o[5]] * EXPAND_DX_SPLIT * (int64_t)o[6]);
      if (need > bextra) bextra = need;
    }
  }
  L.fwd_total = off;       // what a forward pass touches: buffers + per-op areas
  L.scratch0 = off;
  off += round64(L.max_buf);
  L.scratch1 = off;
  off += round64(L.max_buf);
  L.bextra = off;
  off += bextra;
  L.total = off;
  return 0;
}

inline int64_t expand_dwc_bytes(int cin, int cout) { return (64 * (int64_t)cin * cout * 4 + 255) & ~int64_t(255); }

int64_t dw_slice(const View &v, int i) {   // workspace slice of op i's weight-gradient partials (256-byte multiple)
  const int32_t *o = v.ops + OPW * i;
  int64_t w = 0;
  if (o[0] == OP_CONV_SUBM) {
    w = sgnn_conv_bwd_weight_ws_bytes(v.lev_n[o[5]], 27, o[6], o[7]);
    // the fused backward kernel keeps one partial per resident workgroup (conv_bwd_fused.hip)
    if (sgnn_conv_bwd_fused_ok(v.lev_n[o[5]], o[6], o[7], 27)) {
      const int64_t wf = sgnn_conv_bwd_fused_ws(v.lev_n[o[5]], o[6], o[7]);
      if (wf > w) w = wf;
    }
  }
  if (o[0] == OP_CONV_DOWN) w = sgnn_conv_bwd_weight_ws_bytes(v.lev_n[o[5] + 1], 8, o[6], o[7]);
  if (o[0] == OP_EXPAND)   // partials + the 64 reduced slices themselves (they must not live in the shared gradient arena:
                           // with a deferred lane join the next program's backward pass would overwrite them)
    w = ((sgnn_conv_bwd_weight_ws_bytes(v.lev_n[o[5]], 64, o[6], o[7]) + 255) & ~int64_t(255)) + expand_dwc_bytes(o[6], o[7]);
  return (w + 255) & ~int64_t(255);
}

// every convolution keeps its weight-gradient partials in its own slice (their reduces run as one launch at the end)
int64_t dw_ws_need(const View &v) {
  int64_t need = 0;
  for (int i = 0; i < v.nops; ++i) need += dw_slice(v, i);
  return need;
}

int64_t ws_main(const View &v) {
  int64_t need = 0;
  for (int i = 0; i < v.nops; ++i) {
    const int32_t *o = v.ops + OPW * i;
    int64_t w = 0;
    if (o[0] == OP_BN) w = sgnn_bn_ws_bytes(v.lev_n[o[5]], o[6]);
    if (o[0] == OP_LINEAR) w = sgnn_linear_ws_bytes(v.lev_n[o[5]], o[6], o[7]);
    if (w > need) need = w;
  }
  const int64_t dw = dw_ws_need(v);     // without a side lane the partial slices live in the main workspace too
  if (dw > need) need = dw;
  return (need + 255) & ~int64_t(255);
}

// statistics partials a convolution epilogue hands to the neighbouring BatchNorm: [grid blocks][2][C] doubles,
// placed behind the main workspace (the BatchNorm kernels use the main part while they read these)
int64_t ws_stats(const View &v) {
  int64_t need = 0;
  for (int i = 0; i < v.nops; ++i) {
    const int32_t *o = v.ops + OPW * i;
    if (o[0] != OP_CONV_SUBM && o[0] != OP_CONV_DOWN) continue;
    const int64_t rows_f = v.lev_n[o[5]], rows_o = o[0] == OP_CONV_DOWN ? v.lev_n[o[5] + 1] : rows_f;
    // block counts of the finest-grained kernel that may run (the 16-row small kernel)
    const int64_t a = ((rows_o > 0 ? rows_o : 1) + 15) / 16 * 2 * o[7] * (int64_t)sizeof(double);   // forward: out rows x cout
    const int64_t b = ((rows_f > 0 ? rows_f : 1) + 15) / 16 * 2 * o[6] * (int64_t)sizeof(double);   // data gradient: in rows x cin
    if (a > need) need = a;
    if (b > need) need = b;
  }
  return (need + 255) & ~int64_t(255);
}

int64_t ws_need(const View &v) { return ws_main(v) + ws_stats(v); }

// weight-gradient lane: sgnn_prog_backward can run every dW (+ its reduce) on a second stream with its own
// workspace, concurrently with the dX / BatchNorm chain that forms the critical path (both only READ dy)
struct SideLane {
  hipStream_t stream = nullptr;
  void *ws = nullptr;
  int64_t ws_bytes = 0;
  hipEvent_t fork = nullptr, join = nullptr;
} g_side;
// The lane (its workspace and its two events) is one per process = one per GPU.  A backward call holds g_side_mu while
// it issues work; a second host thread that calls sgnn_prog_backward at the same time does not get the lane (its weight
// gradients run on its own stream and workspace — correct, just not overlapped), and sgnn_prog_set_side_stream waits
// for a call in flight before it swaps the lane.
std::mutex g_side_mu;
// sgnn_prog_defer_join(1): a backward call no longer makes its stream wait for the lane at its end — the CALLER joins the
// lane's stream before anything reads parameter gradients.  A program's last weight gradient (its first, widest
// convolution) otherwise stalls the dependent chain of the next program for as long as it runs.
bool g_defer_join = false;

}  // namespace

// stream2 == NULL switches the lane off.  ws2 must not be used by anything else while a backward call is in flight.
SGNN_EXPORT int sgnn_prog_set_side_stream(sgnn_stream_t stream2, void *ws2, int64_t ws2_bytes) {
  std::lock_guard<std::mutex> hold(g_side_mu);
  if (stream2 && !g_side.fork) {
    SGNN_HIP_TRY(hipEventCreateWithFlags(&g_side.fork, hipEventDisableTiming));
    SGNN_HIP_TRY(hipEventCreateWithFlags(&g_side.join, hipEventDisableTiming));
  }
  g_side.stream = (hipStream_t)stream2;
  g_side.ws = stream2 ? ws2 : nullptr;
  g_side.ws_bytes = stream2 ? ws2_bytes : 0;
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_prog_defer_join(int on) {
  const int prev = g_defer_join ? 1 : 0;
  g_defer_join = on != 0;
  return prev;
}


#define PROG_TRY(call)           \
  do {                           \
    const int rc_ = (call);      \
    if (rc_ != SGNN_OK) return rc_; \
  } while (0)

// mode 0: the gradient arena of sgnn_prog_backward (buffers + per-op areas + backward scratch); 1: the arena of
// sgnn_prog_forward (buffers + per-op areas); 2: the forward arena of an inference call (training = 2: liveness-packed)
SGNN_EXPORT int64_t sgnn_prog_arena_floats(const int32_t *ops, int nops, const int32_t *bufs, int nbuf, int n_ext,
                                           const int64_t *lev_n, int nlev, const int32_t *keep, int mode) {
  View v{ops, nullptr, nops, bufs, nbuf, n_ext, lev_n, nullptr, nullptr, nullptr, nullptr, nullptr, nlev};
  Plan P;
  make_plan(v, keep, P);
  Layout L;
  if (mode < 0 || mode > 2 || make_layout(v, P, L, mode == 2, keep) != 0) return -1;
  return mode == 0 ? L.total : L.fwd_total;
}

SGNN_EXPORT int64_t sgnn_prog_ws_bytes(const int32_t *ops, int nops, const int64_t *lev_n, int nlev) {
  View v{ops, nullptr, nops, nullptr, 0, 0, lev_n, nullptr, nullptr, nullptr, nullptr, nullptr, nlev};
  return ws_need(v);
}

// float offset of buffer `b` inside an arena (so the host layer can hand out views); -1 for externals
SGNN_EXPORT int64_t sgnn_prog_buffer_offset(const int32_t *ops, int nops, const int32_t *bufs, int nbuf, int n_ext,
                                            const int64_t *lev_n, int nlev, const int32_t *keep, int infer, int b) {
  View v{ops, nullptr, nops, bufs, nbuf, n_ext, lev_n, nullptr, nullptr, nullptr, nullptr, nullptr, nlev};
  Plan P;
  make_plan(v, keep, P);
  Layout L;
  if (make_layout(v, P, L, infer != 0, keep) != 0 || b < 0 || b >= nbuf) return -1;
  return P.root[b] == b ? L.buf_off[b] : -1;      // buffers the caller keeps are never views
}

SGNN_EXPORT int sgnn_prog_forward(const int32_t *ops, const float *opf, int nops, const int32_t *bufs, int nbuf, int n_ext,
                                  const int64_t *lev_n, const int64_t *lev_ld, void *const *lev_nbr,
                                  void *const *lev_children, void *const *lev_ptable, void *const *lev_parent,
                                  void *const *lev_cnt, int nlev, void *const *params, int nparams,
                                  void *const *ext, void *const *idx,
                                  int nidx, float *arena, int64_t arena_floats, const int32_t *keep, int training,
                                  void *wait_event, void *ws, int64_t ws_bytes, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(ops && opf && bufs && lev_n && lev_ld && params && arena && nops >= 0 && nbuf >= 1 && nlev >= 1 &&
                 n_ext >= 0 && n_ext <= nbuf && (n_ext == 0 || ext));
  View v{ops, opf, nops, bufs, nbuf, n_ext, lev_n, lev_ld, lev_nbr, lev_children, lev_ptable, lev_parent, nlev};
  Plan PL;
  make_plan(v, keep, PL);
  Layout L;
  const bool infer = (training & 2) != 0;     // inference layout: no backward call may follow
  training &= 1;
  SGNN_CHECK_ARG(make_layout(v, PL, L, infer, keep) == 0);
  if (arena_floats < L.fwd_total) {
    sgnn_set_error("sgnn_prog_forward: arena too small (%lld < %lld floats)", (long long)arena_floats,
                   (long long)L.fwd_total);
    return SGNN_ENOWS;
  }
  if (ws_bytes < ws_need(v)) {
    sgnn_set_error("sgnn_prog_forward: workspace too small");
    return SGNN_ENOWS;
  }
  auto B = [&](int b) -> float * { return b < 0 ? nullptr : (b < n_ext ? (float *)ext[b] : arena + L.buf_off[b]); };
  auto LD = [&](int b) -> int64_t { return PL.ld[b]; };       // row stride (floats): wider than the channels for a view
  auto CH = [&](int b) { return b < 0 ? 0 : bufs[2 * b + 1]; };
  auto ROWS = [&](int b) { return lev_n[bufs[2 * b]]; };
  auto P = [&](int p) { return (p >= 0 && p < nparams) ? (float *)params[p] : nullptr; };
  auto I = [&](int i) { return (i >= 0 && i < nidx && idx) ? (const int32_t *)idx[i] : nullptr; };
  // capacity mode: device row count of a rows class (NULL: lev_n is exact)
  auto CNT = [&](int cls) -> const int64_t * { return (lev_cnt && cls >= 0 && cls < nlev) ? (const int64_t *)lev_cnt[cls] : nullptr; };
  // Epilogue fusions (same arithmetic, fewer passes and launches; planned by make_plan):
  //  * conv -> AddTable: the convolution adds the other AddTable input while it stores (the sum buffer is written
  //    directly, the convolution's own output buffer stays untouched) when nothing else reads the convolution output;
  //  * conv [-> AddTable] -> BatchNorm (training): the convolution epilogue reduces the column sums the BatchNorm
  //    statistics pass would recompute from HBM;
  //  * JoinTable in place: its inputs are written straight into their column range of the join buffer.
  std::vector<const double *> pre(nops, nullptr);
  std::vector<int64_t> pre_nblk(nops, 0);
  double *stats_ws = (double *)((char *)ws + ws_main(v));
  for (int i = 0; i < nops; ++i) {
    const int32_t *o = ops + OPW * i;
    const int type = o[0], in0 = o[1], in1 = o[2], out = o[3], par = o[4], lev = o[5], cin = o[6], cout = o[7];
    SGNN_CHECK_ARG(out >= n_ext && out < nbuf && lev >= 0 && lev < nlev && in0 < nbuf && in1 < nbuf);
    SGNN_CHECK_ARG(type == OP_CONCAT_IN || in0 >= 0);
    const int64_t n = lev_n[lev];
    if (PL.skip[i]) continue;
    if (type == OP_CONV_DOWN && wait_event) {
      // the stride-2 tables and everything of the coarser levels (hash, 3x3x3 rulebooks, row counts) may still be in
      // flight on the caller's pyramid lane: the first Convolution(2,2) is the first operation that touches them
      sgnn_stamp("down-wait<", stream);
      SGNN_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)wait_event, 0));
      sgnn_stamp("down-wait>", stream);
      wait_event = nullptr;
    }
    switch (type) {
      case OP_CONV_SUBM:
      case OP_CONV_DOWN: {
        const bool down = type == OP_CONV_DOWN;
        SGNN_CHECK_ARG(!down || lev + 1 < nlev);
        const int64_t n_out = down ? lev_n[lev + 1] : n;
        const int32_t *table = (const int32_t *)(down ? lev_children[lev] : lev_nbr[lev]);
        const int64_t ld = down ? lev_ld[lev + 1] : lev_ld[lev];
        ConvEpi epi{};
        int dst_buf = out;
        if (PL.add_dst[i] >= 0) {
          const int32_t *a = ops + OPW * (i + 1);
          const int other = a[1] == out ? a[2] : a[1];
          epi.addend = B(other);
          epi.ld_add = LD(other);
          dst_buf = PL.add_dst[i];
        }
        int j = i + 1 + (PL.add_dst[i] >= 0 ? 1 : 0);
        if (g_fuse && training && n_out > 0 && j < nops && ops[OPW * j] == OP_BN && ops[OPW * j + 1] == dst_buf &&
            sgnn_conv_epi_supported(cin, cout)) {
          epi.stats = 1;
          epi.partial = stats_ws;
          pre[j] = stats_ws;
          pre_nblk[j] = sgnn_conv_grid_blocks(n_out, cin, cout, down ? 8 : 27);
        }
        epi.ldx = LD(in0);
        epi.ldy = LD(dst_buf);
        epi.n_dev = CNT(down ? lev + 1 : lev);
        PROG_TRY(sgnn_conv_fwd_impl(B(in0), n, cin, P(par), down ? 8 : 27, table, ld, n_out, cout, B(dst_buf), 0, 0, nullptr,
                                    nullptr, 1, 1, down ? 8 : 27, &epi, stream));
        break;
      }
      case OP_UNPOOL:  // in0 lives on level lev+1, out on level lev
        SGNN_CHECK_ARG(lev + 1 < nlev);
        PROG_TRY(sgnn_gather_rows_ld(B(in0), LD(in0), cin, (const int32_t *)lev_parent[lev], n, B(out), LD(out), stream, CNT(lev)));
        break;
      case OP_BN: {
        float *save = arena + L.aux_off[i];
        PROG_TRY(sgnn_bn_fwd_impl(B(in0), LD(in0), n, cin, P(par), P(par + 1), P(par + 2), P(par + 3), opf[4 * i],
                                  opf[4 * i + 1], training, opf[4 * i + 2], save, save + cin, B(out), LD(out), pre[i],
                                  pre_nblk[i], ws, ws_bytes, stream, CNT(lev)));
        break;
      }
      case OP_ADD:
        SGNN_CHECK_ARG(in1 >= 0);
        PROG_TRY(sgnn_add_ld(B(in0), LD(in0), B(in1), LD(in1), n, cin, B(out), LD(out), stream, CNT(lev)));
        break;
      case OP_JOIN:  // cin = channels of in0, cout = channels of in1
        SGNN_CHECK_ARG(in1 >= 0 && LD(in0) == cin && LD(in1) == cout && LD(out) == cin + cout);
        PROG_TRY(sgnn_concat_rows_dn(B(in0), cin, nullptr, B(in1), cout, nullptr, n, B(out), stream, CNT(lev)));
        break;
      case OP_CONCAT_IN: {
        const int in2 = o[8];
        SGNN_CHECK_ARG(in2 < nbuf && CH(in0) + CH(in1) + CH(in2) == CH(out) && LD(out) == CH(out));
        PROG_TRY(sgnn_concat3_rows_dn(B(in0), CH(in0), I(o[9]), B(in1), CH(in1), I(o[10]), B(in2), CH(in2), I(o[11]), n,
                                      B(out), stream, CNT(lev)));
        break;
      }
      case OP_EXPAND: {   // out rows = 8 * n (child row 8p + parity), features of the parents never replicated
        SGNN_CHECK_ARG(ROWS(out) == 8 * n && LD(in0) == cin && LD(out) == cout);
        const int32_t *S, *ST, *PAR;
        PROG_TRY(sgnn_expand_maps(&S, &ST, &PAR));
        float *wc = arena + L.aux_off[i];
        PROG_TRY(sgnn_expand_weights(P(par), cin, cout, wc, stream));
        ConvEpi xepi{};
        xepi.n_dev = CNT(lev);
        PROG_TRY(sgnn_conv_fwd_impl(B(in0), n, cin, wc, 8, (const int32_t *)lev_nbr[lev], lev_ld[lev], n, cout, B(out), 0,
                                    0, S, nullptr, 1, 8, 27, &xepi, stream));
        break;
      }
      case OP_LINEAR: {
        SGNN_CHECK_ARG(cout >= 1 && cout <= 4 && LD(in0) == cin);
        const float *w[4] = {}, *b[4] = {};
        for (int q = 0; q < cout; ++q) {
          w[q] = P(par + 2 * q);
          b[q] = P(par + 2 * q + 1);
        }
        PROG_TRY(sgnn_linear_fwd_rows(B(in0), n, cin, w, b, cout, B(out), stream, CNT(lev)));
        break;
      }
      default:
        sgnn_set_error("sgnn_prog_forward: unknown op %d", type);
        return SGNN_EINVAL;
    }
  }
  return SGNN_OK;
}

// garena has the same layout as arena.  gout[b] != NULL: the caller's gradient of buffer b (a program output); it
// is copied into the arena first (the executor accumulates into its own memory only).  gext[e] != NULL: where the
// gradient of external input e is wanted (fully overwritten).
SGNN_EXPORT int sgnn_prog_backward(const int32_t *ops, const float *opf, int nops, const int32_t *bufs, int nbuf, int n_ext,
                                   const int64_t *lev_n, const int64_t *lev_ld, void *const *lev_nbr,
                                   void *const *lev_children, void *const *lev_ptable, void *const *lev_parent,
                                   void *const *lev_cnt, int nlev, void *const *params,
                                   void *const *pgrads, int nparams,
                                   void *const *ext, void *const *gext, void *const *idx, int nidx,
                                   const float *arena, float *garena, int64_t arena_floats, void *const *gout,
                                   const int32_t *keep, int training, void *ws, int64_t ws_bytes,
                                   sgnn_stream_t stream) {
  SGNN_CHECK_ARG(ops && opf && bufs && lev_n && lev_ld && params && pgrads && arena && garena && gout &&
                 n_ext >= 0 && n_ext <= nbuf && (n_ext == 0 || (ext && gext)));
  View v{ops, opf, nops, bufs, nbuf, n_ext, lev_n, lev_ld, lev_nbr, lev_children, lev_ptable, lev_parent, nlev};
  Plan PL;
  make_plan(v, keep, PL);               // the same decisions the forward call took (same inputs)
  Layout L;
  SGNN_CHECK_ARG(make_layout(v, PL, L) == 0);
  if (arena_floats < L.total || ws_bytes < ws_need(v)) {
    sgnn_set_error("sgnn_prog_backward: arena or workspace too small");
    return SGNN_ENOWS;
  }
  hipStream_t hs = (hipStream_t)stream;
  // gradient state of a buffer: 0 nothing yet, 1 G(b) holds it, 2 it EQUALS the gradient of buffer alias[b]
  // (an AddTable input whose only contribution so far is the sum's gradient: nothing is copied until something
  // has to be added to it, and a reader just follows the alias), 3 it is the caller's tensor gout[b], read in place
  // (contiguous rows; folded into G(b) by the first kernel that has to add to it — no up-front copy of the outputs'
  // gradients into the arena)
  std::vector<char> init(nbuf, 0);
  std::vector<int> alias(nbuf, -1);
  std::vector<int> lazy_lin(nbuf, -1);   // buffer -> LINEAR op whose data gradient the buffer's BatchNorm forms itself (BnLin)
  std::vector<char> viewed(nbuf, 0);      // storage shared through an in-place JoinTable: keeps the copying path
  for (int b = 0; b < nbuf; ++b)
    if (PL.root[b] != b) viewed[b] = viewed[PL.root[b]] = 1;
  auto X = [&](int b) -> const float * { return b < 0 ? nullptr : (b < n_ext ? (const float *)ext[b] : arena + L.buf_off[b]); };
  auto G = [&](int b) -> float * { return b < n_ext ? (float *)gext[b] : garena + L.buf_off[b]; };
  auto LD = [&](int b) -> int64_t { return PL.ld[b]; };                                  // row stride of X(b) and G(b)
  auto CH = [&](int b) { return b < 0 ? 0 : bufs[2 * b + 1]; };
  // where the gradient a buffer HOLDS lives (its own arena slot, or the caller's tensor) and that storage's row stride
  auto GH = [&](int b) -> const float * { return init[b] == 3 ? (const float *)gout[b] : G(b); };
  auto GHLD = [&](int b) -> int64_t { return init[b] == 3 ? (int64_t)CH(b) : PL.ld[b]; };
  auto GR = [&](int b) -> const float * { return init[b] == 2 ? GH(alias[b]) : GH(b); };   // where b's gradient is read
  auto GRLD = [&](int b) -> int64_t { return init[b] == 2 ? GHLD(alias[b]) : GHLD(b); };
  auto P = [&](int p) { return (p >= 0 && p < nparams) ? (float *)params[p] : nullptr; };
  auto PG = [&](int p) { return (p >= 0 && p < nparams) ? (float *)pgrads[p] : nullptr; };
  auto ROWS = [&](int b) { return lev_n[bufs[2 * b]]; };
  auto I = [&](int i) { return (i >= 0 && i < nidx && idx) ? (const int32_t *)idx[i] : nullptr; };
  auto CNT = [&](int cls) -> const int64_t * { return (lev_cnt && cls >= 0 && cls < nlev) ? (const int64_t *)lev_cnt[cls] : nullptr; };
  auto BCNT = [&](int b) -> const int64_t * { return CNT(bufs[2 * b]); };     // device row count of buffer b's rows class
  for (int b = n_ext; b < nbuf; ++b)
    if (gout[b]) {
      if (viewed[b] || !g_fuse) {
        if (L.buf_floats[b] > 0) PROG_TRY(sgnn_copy_words(G(b), gout[b], L.buf_floats[b], hs));
        init[b] = 1;
      } else {
        init[b] = 3;
      }
    }
  float *scratch[2] = {garena + L.scratch0, garena + L.scratch1};
  // where a kernel should write the gradient of buffer b: the buffer itself unless it already holds data
  auto target = [&](int b, int which) { return init[b] == 1 ? scratch[which] : G(b); };
  auto TLD = [&](int b, const float *t) -> int64_t { return t == G(b) ? LD(b) : CH(b); };   // scratch rows are contiguous
  auto commit = [&](int b, float *wrote) -> int {  // fold a freshly written gradient into buffer b
    if (wrote == G(b)) {
      if (init[b] == 2 || init[b] == 3) {            // G(b) = fresh + the aliased / the caller's gradient
        const float *other = GR(b);
        const int64_t ldo = GRLD(b);
        alias[b] = -1;
        init[b] = 1;
        return sgnn_add_ld(G(b), LD(b), other, ldo, ROWS(b), CH(b), G(b), LD(b), stream, BCNT(b));
      }
      init[b] = 1;
      return SGNN_OK;
    }
    return sgnn_add_ld(G(b), LD(b), wrote, CH(b), ROWS(b), CH(b), G(b), LD(b), stream, BCNT(b));
  };
  auto wants = [&](int b) { return b >= n_ext || gext[b] != nullptr; };
  // dW launches go to the side lane when one is configured and its workspace is big enough
  std::unique_lock<std::mutex> lane_lock(g_side_mu, std::try_to_lock);
  const bool side = lane_lock.owns_lock() && g_side.stream && g_side.stream != hs && g_side.ws &&
                    g_side.ws_bytes >= dw_ws_need(v);
  bool forked = false;
  auto dw_lane = [&]() -> hipStream_t {
    if (!side) return hs;
    (void)hipEventRecord(g_side.fork, hs);                 // dy of this op is final here (all its consumers ran)
    (void)hipStreamWaitEvent(g_side.stream, g_side.fork, 0);
    forked = true;
    return g_side.stream;
  };
  char *dw_base = (char *)(side ? g_side.ws : ws);
  int64_t dw_off = 0;
  DwBatch batch{};
  struct BatchGuard {            // the deferral is on only while this call runs, whatever path it leaves by
    explicit BatchGuard(DwBatch *b) { sgnn_dw_batch = b; }
    ~BatchGuard() { sgnn_dw_batch = nullptr; }
  } guard(side ? &batch : nullptr);   // without the lane the slices share `ws` with the BatchNorm kernels: reduce at once
  struct PendingExpand { const float *dwc; int cin, cout; float *dw; };
  std::vector<PendingExpand> pending_expand;
  std::vector<const double *> pre(nops, nullptr);
  std::vector<int64_t> pre_nblk(nops, 0);
  double *stats_ws = (double *)((char *)ws + ws_main(v));

  for (int i = nops - 1; i >= 0; --i) {
    const int32_t *o = ops + OPW * i;
    const int type = o[0], in0 = o[1], in1 = o[2], out = o[3], par = o[4], lev = o[5], cin = o[6], cout = o[7];
    const int64_t n = lev_n[lev];
    if (!init[out] && !(type == OP_BN && lazy_lin[out] >= 0)) {  // no gradient reached this output: its producers contribute nothing
      if (type == OP_CONV_SUBM || type == OP_CONV_DOWN || type == OP_EXPAND)
        PROG_TRY(sgnn_fill32(PG(par), 0u, (int64_t)(type == OP_CONV_DOWN ? 8 : 27) * cin * cout, hs));
      if (type == OP_BN) {
        {
          void *zp[2] = {PG(par), PG(par + 1)};
          const int64_t zw[2] = {cin, cin};
          PROG_TRY(sgnn_fill32_multi(zp, zw, 2, 0u, hs));
        }
      }
      if (type == OP_LINEAR)
        for (int q = 0; q < cout; ++q) {
          void *zp[2] = {PG(par + 2 * q), PG(par + 2 * q + 1)};
          const int64_t zw[2] = {cin, 1};
          PROG_TRY(sgnn_fill32_multi(zp, zw, 2, 0u, hs));
        }
      continue;
    }
    const float *dy = GR(out);
    const int64_t ld_dy = GRLD(out);
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
        // dX and dW from ONE kernel on the training stream where the shape allows and sgnn_tune.conv_bwd_fused asks for it
        // (conv_bwd_fused.hip; its reduce runs on the training stream too): no lane fork.  Otherwise the lane forks HERE, in
        // front of the data-gradient launch: the weight gradient runs beside the dX kernel of its layer.
        // (Never make the lane wait for a LATER kernel of the training stream: one such edge per program — no kernel moved —
        //  costs a replayed step +0.75 ms, profiles/r06l_ab_endfork.txt; every variant of round 3-6 that re-timed the forks lost
        //  0.8-0.9 ms the same way.)
        bool fused_bwd = false;
        const bool fused_try = !down && wants(in0) && g_fuse && n > 0 &&
                               sgnn_conv_epi_supported(cout, cin) && sgnn_conv_bwd_fused_ok(n, cin, cout, K);
        hipStream_t lane = fused_try ? hs : dw_lane();
        if (wants(in0)) {
          if (g_fuse && sgnn_conv_epi_supported(cout, cin) && n > 0) {
            // the data gradient lands in G(in0) directly: what the buffer (or its alias) already holds is added in the
            // store (in place), and when in0 is the output of the BatchNormReLU right before this op and this is the
            // last contribution to its gradient, the epilogue also reduces sum dz / sum dz*xhat for that BatchNorm
            ConvEpi epi{};
            if (init[in0]) {                    // what the buffer (or its alias / the caller's tensor) already holds
              epi.addend = GR(in0);
              epi.ld_add = GRLD(in0);
            }
            if (i > 0 && ops[OPW * (i - 1)] == OP_BN && ops[OPW * (i - 1) + 3] == in0 && ops[OPW * (i - 1) + 6] == cin) {
              const int32_t *bo = ops + OPW * (i - 1);
              const float *save = arena + L.aux_off[i - 1];
              epi.stats = 2;
              epi.partial = stats_ws;
              epi.bn_x = X(bo[1]);
              epi.ld_bnx = LD(bo[1]);
              epi.mean = save;
              epi.invstd = save + cin;
              epi.gamma = P(bo[4]);
              epi.beta = P(bo[4] + 1);
              epi.leak = opf[4 * (i - 1) + 2];
              pre[i - 1] = stats_ws;
              pre_nblk[i - 1] = sgnn_conv_grid_blocks(n, cout, cin, K);
            }
            epi.ldx = ld_dy;
            epi.ldy = LD(in0);
            epi.n_dev = CNT(lev);
            fused_bwd = fused_try && sgnn_conv_bwd_fused_usable(n, cin, cout, K, epi, G(in0), X(in0), LD(in0));
            if (fused_bwd) {
              PROG_TRY(sgnn_conv_bwd_fused_impl(dy, n, cout, P(par), tab_b, lev_ld[lev], cin, G(in0), epi, X(in0), LD(in0),
                                                PG(par), dw_base + dw_off, dw_slice(v, i), stream));
            } else {
              PROG_TRY(sgnn_conv_fwd_impl(dy, n_dy, cout, P(par), K, tab_b, lev_ld[lev], n, cin, G(in0), flags_b, 0, nullptr,
                                          nullptr, 1, 1, K, &epi, stream));
            }
            init[in0] = 1;
            alias[in0] = -1;
          } else {
            SGNN_CHECK_ARG(ld_dy == cout);               // views are only planned around compiled shapes
            float *t = target(in0, 0);
            ConvEpi pepi{};
            pepi.n_dev = CNT(lev);
            PROG_TRY(sgnn_conv_fwd_impl(dy, n_dy, cout, P(par), K, tab_b, lev_ld[lev], n, cin, t, flags_b, 0, nullptr, nullptr,
                                        1, 1, K, &pepi, stream));
            PROG_TRY(commit(in0, t));
          }
        }
        if (!fused_bwd) {
          if (fused_try) lane = dw_lane();      // (strides the fused kernel does not take: fork late)
          PROG_TRY(sgnn_conv_bwd_weight_impl(X(in0), n, cin, LD(in0), dy, cout, ld_dy, tab_f, ld_f, K, n_dy, PG(par), 0,
                                             nullptr, nullptr, 1, 1, K, dw_base + dw_off, dw_slice(v, i),
                                             (sgnn_stream_t)lane, CNT(down ? lev + 1 : lev)));
          if (lane != hs) sgnn_stamp("dw>", (sgnn_stream_t)lane);      // (nothing unless stamps are on: scripts/lane_stamps.py)
        }
        dw_off += dw_slice(v, i);
        break;
      }
      case OP_UNPOOL:
        if (wants(in0)) {
          float *t = target(in0, 0);
          PROG_TRY(sgnn_gather_sum_ld(dy, ld_dy, cin, (const int32_t *)lev_children[lev], lev_ld[lev + 1], 8, lev_n[lev + 1],
                                      t, TLD(in0, t), stream, CNT(lev + 1)));
          PROG_TRY(commit(in0, t));
        }
        break;
      case OP_BN: {
        const float *save = arena + L.aux_off[i];
        // the kernel adds what the buffer already holds (in place) or the aliased gradient: no scratch pass, no k_add
        const float *addend = nullptr;
        int64_t ld_add = cin;
        if (wants(in0) && init[in0]) {
          addend = GR(in0);
          ld_add = GRLD(in0);
        }
        float *t = wants(in0) ? G(in0) : scratch[0];
        BnLin bl{};
        const bool lazy = lazy_lin[out] >= 0;
        if (lazy) {                       // dy = (gradient of the head's output) x (the head's weights), never stored
          const int32_t *lo = ops + OPW * lazy_lin[out];
          bl.g = GR(lo[3]);
          bl.ldg = GRLD(lo[3]);
          bl.nout = lo[7];
          for (int q = 0; q < lo[7]; ++q) bl.w[q] = P(lo[4] + 2 * q);
        }
        PROG_TRY(sgnn_bn_bwd_impl(X(in0), LD(in0), lazy ? nullptr : dy, ld_dy, n, cin, P(par), P(par + 1), save, save + cin, training,
                                  opf[4 * i + 2], addend, ld_add, t, wants(in0) ? LD(in0) : cin, PG(par), PG(par + 1), pre[i],
                                  pre_nblk[i], ws, ws_bytes, stream, CNT(lev), lazy ? &bl : nullptr));
        if (wants(in0)) {
          init[in0] = 1;
          alias[in0] = -1;
        }
        break;
      }
      case OP_ADD: {
        const int src = init[out] == 2 ? alias[out] : out;      // the buffer that physically holds dy
        for (int side_ = 0; side_ < 2; ++side_) {
          const int b = side_ ? in1 : in0;
          if (!wants(b)) continue;
          if (init[b] == 1) {
            PROG_TRY(sgnn_add_ld(G(b), LD(b), dy, ld_dy, n, cin, G(b), LD(b), stream, CNT(lev)));
          } else if (init[b] == 2 || init[b] == 3) {            // two contributions held elsewhere: materialise the sum
            PROG_TRY(sgnn_add_ld(GR(b), GRLD(b), dy, ld_dy, n, cin, G(b), LD(b), stream, CNT(lev)));
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
        if (PL.join_view[i]) {                                   // in place: the inputs' gradients ARE column ranges of dy
          if (init[out] != 1 || init[in0] || init[in1]) {
            sgnn_set_error("sgnn_prog_backward: in-place JoinTable met an unexpected gradient state");
            return SGNN_EINVAL;
          }
          init[in0] = init[in1] = 1;
          break;
        }
        SGNN_CHECK_ARG(ld_dy == cin + cout && LD(in0) == cin && LD(in1) == cout);   // make_plan: a copying JoinTable never reads views
        float *ta = wants(in0) ? target(in0, 0) : nullptr;
        float *tb = wants(in1) ? target(in1, 1) : nullptr;
        PROG_TRY(sgnn_concat_rows_bwd_dn(dy, cin, nullptr, cout, nullptr, n, ta, n, tb, n, stream, CNT(lev)));
        if (ta) PROG_TRY(commit(in0, ta));
        if (tb) PROG_TRY(commit(in1, tb));
        break;
      }
      case OP_CONCAT_IN: {
        const int src[3] = {in0, in1, o[8]};
        float *d[3] = {nullptr, nullptr, nullptr};
        for (int q = 0; q < 3; ++q)
          if (src[q] >= 0 && wants(src[q])) {
            if (init[src[q]]) {
              sgnn_set_error("sgnn_prog_backward: a CONCAT_IN source already carries a gradient (unsupported)");
              return SGNN_EINVAL;
            }
            d[q] = G(src[q]);
            init[src[q]] = 1;
          }
        SGNN_CHECK_ARG(ld_dy == CH(out));
        PROG_TRY(sgnn_concat3_rows_bwd_dn(dy, CH(in0), I(o[9]), CH(in1), I(o[10]), CH(o[8]), I(o[11]), n, d[0],
                                          in0 >= 0 ? ROWS(in0) : 0, d[1], in1 >= 0 ? ROWS(in1) : 0, d[2],
                                          o[8] >= 0 ? ROWS(o[8]) : 0, stream, CNT(lev)));
        break;
      }
      case OP_EXPAND: {
        const int32_t *S, *ST, *PAR;
        PROG_TRY(sgnn_expand_maps(&S, &ST, &PAR));
        const float *wc = arena + L.aux_off[i];
        const int32_t *nbr = (const int32_t *)lev_nbr[lev];
        // reduced 64-slice weight gradient: at the tail of this op's weight-gradient workspace slice (lane-owned memory)
        // (without the lane the slices share `ws` with the BatchNorm kernels: the gradient arena then, as before)
        float *dwc = side ? (float *)(dw_base + dw_off + dw_slice(v, i) - expand_dwc_bytes(cin, cout)) : garena + L.bextra;
        float *part = garena + L.bextra + round64(64 * (int64_t)cin * cout);
        const hipStream_t lane = dw_lane();
        SGNN_CHECK_ARG(ld_dy == cout && LD(in0) == cin);
        if (wants(in0) && n > 0) {
          // 64 offsets per parent row, cut into G slices that run as conv groups; the slices are then added
          const int Gs = EXPAND_DX_SPLIT;
          ConvEpi xepi{};
          xepi.n_dev = CNT(lev);
          PROG_TRY(sgnn_conv_fwd_impl(dy, 8 * n, cout, wc, 64 / Gs, nbr, lev_ld[lev], n, cin, part, SGNN_CONV_TRANSPOSE_W,
                                      0, ST, PAR, 8, Gs, 27, &xepi, stream));
          float *t = target(in0, 0);
          PROG_TRY(sgnn_sum_groups_dn(part, cin, n, Gs, t, stream, CNT(lev)));
          PROG_TRY(commit(in0, t));
        }
        {
          PROG_TRY(sgnn_conv_bwd_weight_impl(X(in0), n, cin, cin, dy, cout, cout, nbr, lev_ld[lev], 8, n, dwc, 0, S, nullptr, 1,
                                             8, 27, dw_base + dw_off, dw_slice(v, i), (sgnn_stream_t)lane, CNT(lev)));
        }
        dw_off += dw_slice(v, i);
        pending_expand.push_back(PendingExpand{dwc, cin, cout, PG(par)});   // dwc is final after the batched reduce
        break;
      }
      case OP_LINEAR: {
        const float *w[4] = {};
        float *dw[4] = {}, *db[4] = {};
        for (int q = 0; q < cout; ++q) {
          w[q] = P(par + 2 * q);
          dw[q] = PG(par + 2 * q);
          db[q] = PG(par + 2 * q + 1);
        }
        SGNN_CHECK_ARG(ld_dy == cout && LD(in0) == cin);
        if (PL.lin_bn[i] >= 0 && wants(in0) && !init[in0] && n > 0 && !pre[PL.lin_bn[i]]) {
          // weight / bias gradients only; the BatchNorm before the head forms dx = dy w itself in both of its passes
          PROG_TRY(sgnn_linear_bwd_rows(X(in0), dy, n, cin, w, cout, nullptr, dw, db, ws, ws_bytes, stream, CNT(lev)));
          lazy_lin[in0] = i;
          break;
        }
        // the input rows already carry a gradient (the caller's, for the rows the next level reads; an alias; or G itself):
        // the head adds it in its own pass — dx = dy w + that — instead of an add launch over the level (round 5)
        const float *have = (g_fuse && g_lin_add && wants(in0) && init[in0] && n > 0) ? GR(in0) : nullptr;
        const int64_t have_ld = have ? GRLD(in0) : 0;
        if (have && have_ld % 4 == 0 && ((uintptr_t)have & 15) == 0 && !viewed[in0]) {
          PROG_TRY(sgnn_linear_bwd_rows(X(in0), dy, n, cin, w, cout, G(in0), dw, db, ws, ws_bytes, stream, CNT(lev), have,
                                        have_ld));
          alias[in0] = -1;
          init[in0] = 1;
          break;
        }
        float *t = wants(in0) ? target(in0, 0) : nullptr;
        PROG_TRY(sgnn_linear_bwd_rows(X(in0), dy, n, cin, w, cout, t, dw, db, ws, ws_bytes, stream, CNT(lev)));
        if (t) PROG_TRY(commit(in0, t));
        break;
      }
      default:
        sgnn_set_error("sgnn_prog_backward: unknown op %d", type);
        return SGNN_EINVAL;
    }
  }
  for (int b = 0; b < n_ext; ++b) {                   // the caller reads gext[b]: an alias has to become a copy,
    if (!gext[b] || L.buf_floats[b] == 0) continue;   // an input nothing reached gets zeros
    if (init[b] == 2)
      PROG_TRY(sgnn_copy_words(G(b), GR(b), L.buf_floats[b], hs));
    else if (init[b] == 0)
      PROG_TRY(sgnn_fill32(G(b), 0u, L.buf_floats[b], hs));
  }
  {
    const hipStream_t lane = side ? g_side.stream : hs;
    PROG_TRY(sgnn_dw_batch_flush(&batch, lane));      // all deferred weight-gradient reduces: one launch
    for (const PendingExpand &pe : pending_expand)
      PROG_TRY(sgnn_expand_weights_bwd(pe.dwc, pe.cin, pe.cout, pe.dw, (sgnn_stream_t)lane));
  }
  if (forked && !g_defer_join) {                      // parameter gradients are complete once the lane has drained
    SGNN_HIP_TRY(hipEventRecord(g_side.join, g_side.stream));
    SGNN_HIP_TRY(hipStreamWaitEvent(hs, g_side.join, 0));
  }
  return SGNN_OK;
}
