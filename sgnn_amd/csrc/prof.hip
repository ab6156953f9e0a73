// Optional per-launch timing of the convolution kernels with HIP events recorded on the caller's
// stream (bench.py's live roofline leg).  Disabled by default: zero overhead unless enabled.
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
