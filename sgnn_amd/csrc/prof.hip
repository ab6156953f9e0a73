// Optional per-launch timing of the convolution kernels with HIP events recorded on the caller's
// stream (bench.py's live roofline leg).  Disabled by default: zero overhead unless enabled.
#include <cstdlib>
#include <cstring>
#include <vector>
#include "common.h"

struct ProfRec {
  int kind, cin, cout, K, flags;
  int64_t n_out;
  hipEvent_t e0, e1;
};

static std::vector<ProfRec> g_recs;
static int g_used = 0;
static int g_dropped = 0;
static bool g_on = false;

int sgnn_prof_begin_launch(int kind, int64_t n_out, int cin, int cout, int K, int flags, hipStream_t s) {
  if (!g_on) return -1;
  if (g_used >= (int)g_recs.size()) {
    ++g_dropped;
    return -1;
  }
  ProfRec &r = g_recs[g_used];
  r.kind = kind; r.cin = cin; r.cout = cout; r.K = K; r.flags = flags; r.n_out = n_out;
  if (hipEventRecord(r.e0, s) != hipSuccess) return -1;
  return g_used++;
}

void sgnn_prof_end_launch(int slot, hipStream_t s) {
  if (slot >= 0) (void)hipEventRecord(g_recs[slot].e1, s);
}

SGNN_EXPORT int sgnn_prof_enable(int max_records) {
  SGNN_CHECK_ARG(max_records > 0 && max_records <= (1 << 20));
  if ((int)g_recs.size() < max_records) {
    const size_t old = g_recs.size();
    g_recs.resize(max_records);
    for (size_t i = old; i < g_recs.size(); ++i) {
      SGNN_HIP_TRY(hipEventCreate(&g_recs[i].e0));
      SGNN_HIP_TRY(hipEventCreate(&g_recs[i].e1));
    }
  }
  g_used = 0;
  g_dropped = 0;
  g_on = true;
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_prof_disable(void) {
  g_on = false;
  return SGNN_OK;
}

// continue recording into the existing buffers (after sgnn_prof_enable ... sgnn_prof_disable)
SGNN_EXPORT int sgnn_prof_resume(void) {
  SGNN_CHECK_ARG(!g_recs.empty());
  g_on = true;
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_prof_count(void) { return g_used; }
SGNN_EXPORT int sgnn_prof_dropped(void) { return g_dropped; }

// host-side read-back; the caller must have synchronised the stream first
SGNN_EXPORT int sgnn_prof_get(int i, int *kind, int64_t *n_out, int *cin, int *cout, int *K, int *flags,
                              float *ms) {
  SGNN_CHECK_ARG(i >= 0 && i < g_used && kind && n_out && cin && cout && K && flags && ms);
  const ProfRec &r = g_recs[i];
  *kind = r.kind; *n_out = r.n_out; *cin = r.cin; *cout = r.cout; *K = r.K; *flags = r.flags;
  SGNN_HIP_TRY(hipEventElapsedTime(ms, r.e0, r.e1));
  return SGNN_OK;
}


// number of kernel launches this library has issued in this process (all streams; see SGNN_LAUNCH in common.h)
long long sgnn_launch_counter = 0;
SGNN_EXPORT int64_t sgnn_launch_count(void) { return (int64_t)sgnn_launch_counter; }


// ---------------------------------------------------------------------------
// Device time stamps inside a captured graph (scripts/lane_stamps.py): events cannot be recorded in a replayed graph and a
// kernel trace perturbs exactly what is being asked (which branch of the graph waits for which), so a one-thread kernel
// writes the constant 100 MHz clock (s_memrealtime) into a caller-owned slot.  Labels are kept host-side in capture
// order; every replay rewrites the same slots.  Disabled (no launch at all) unless sgnn_stamp_enable was given a buffer.
// ---------------------------------------------------------------------------
#define STAMP_LABEL 64
static int64_t *g_stamp_buf = nullptr;
static int g_stamp_max = 0, g_stamp_used = 0;
static std::vector<char> g_stamp_labels;

__global__ void k_stamp(int64_t *slot) { *slot = (int64_t)wall_clock64(); }

SGNN_EXPORT int sgnn_stamp_enable(int64_t *buf, int max_stamps) {
  SGNN_CHECK_ARG(max_stamps >= 0 && (buf || max_stamps == 0));
  g_stamp_buf = max_stamps ? buf : nullptr;
  g_stamp_max = max_stamps;
  g_stamp_used = 0;
  g_stamp_labels.assign((size_t)max_stamps * STAMP_LABEL, 0);
  return SGNN_OK;
}
SGNN_EXPORT int sgnn_stamp_reset(void) {
  g_stamp_used = 0;
  return SGNN_OK;
}
SGNN_EXPORT int sgnn_stamp_count(void) { return g_stamp_used; }
SGNN_EXPORT const char *sgnn_stamp_label(int i) {
  return (i >= 0 && i < g_stamp_used) ? &g_stamp_labels[(size_t)i * STAMP_LABEL] : "";
}
// returns the slot written, or -1 when stamps are off / the buffer is full
SGNN_EXPORT int sgnn_stamp(const char *label, sgnn_stream_t stream) {
  if (!g_stamp_buf || g_stamp_used >= g_stamp_max) return -1;
  if (const char *only = getenv("SGNN_STAMP_ONLY")) {       // comma-separated labels: which stamps disturb the replay?
    const size_t len = strlen(label ? label : "");
    bool hit = false;
    for (const char *p = only; *p && !hit;) {
      const char *e = strchr(p, ',');
      const size_t l = e ? (size_t)(e - p) : strlen(p);
      hit = l == len && strncmp(p, label, len) == 0;
      p += l + (e ? 1 : 0);
    }
    if (!hit) return -1;
  }
  const int slot = g_stamp_used++;
  snprintf(&g_stamp_labels[(size_t)slot * STAMP_LABEL], STAMP_LABEL, "%s", label ? label : "");
  hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, (hipStream_t)stream, g_stamp_buf + slot);
  return slot;
}
