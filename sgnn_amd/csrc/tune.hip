// The library's measurement switches: ONE table.  Rounds 2-5 grew a process-global g_* switch with its
// own exported setter next to every kernel variant that was ever A/B-measured; they are fields of `sgnn_tune` now (documented
// in include/sgnn_hip.h), read where they were read, set by name.  None of them changes results beyond fp32 / fp64 summation
// order; defaults are the measured winners.  Host code only.
#include <string.h>
#include "common.h"

sgnn_tune g_tune = {
    /* conv_small          */ 1,
    /* conv_small_rows     */ 160 * 256,
    /* conv_unrolled       */ 1,
    /* conv_one_round      */ 1,
    /* conv_wide_epi       */ 1,
    /* conv_dw_blocks      */ 256,
    /* conv_dw_c1          */ 1,
    /* conv_bwd_fused      */ 0,
    /* conv_bwd_fused_rows */ 40960,
    /* rulebook_lds        */ 0,
    /* rulebook_multi      */ 1,
    /* scan_inline         */ 1,
    /* chain_merged        */ 1,
    /* prog_fusion         */ 1,
    /* prog_lin_bn         */ 1,
    /* prog_lin_add        */ 1,
};

namespace {
struct Field {
  const char *name;
  int64_t sgnn_tune::*ptr;
  int64_t lo, hi;     // accepted range (booleans: any non-zero value stores 1)
  bool boolean;
};
const Field kFields[] = {
    {"conv_small", &sgnn_tune::conv_small, 0, 1, true},
    {"conv_small_rows", &sgnn_tune::conv_small_rows, 0, (int64_t)1 << 36, false},
    {"conv_unrolled", &sgnn_tune::conv_unrolled, 0, 1, true},
    {"conv_one_round", &sgnn_tune::conv_one_round, 0, 1, true},
    {"conv_wide_epi", &sgnn_tune::conv_wide_epi, 0, 1, true},
    {"conv_dw_blocks", &sgnn_tune::conv_dw_blocks, 1, 4096, false},
    {"conv_dw_c1", &sgnn_tune::conv_dw_c1, 0, 1, true},
    {"conv_bwd_fused", &sgnn_tune::conv_bwd_fused, 0, 1, true},
    {"conv_bwd_fused_rows", &sgnn_tune::conv_bwd_fused_rows, 256, (int64_t)1 << 36, false},
    {"rulebook_lds", &sgnn_tune::rulebook_lds, 0, 1, true},
    {"rulebook_multi", &sgnn_tune::rulebook_multi, 0, 1, true},
    {"scan_inline", &sgnn_tune::scan_inline, 0, 1, true},
    {"chain_merged", &sgnn_tune::chain_merged, 0, 1, true},
    {"prog_fusion", &sgnn_tune::prog_fusion, 0, 1, true},
    {"prog_lin_bn", &sgnn_tune::prog_lin_bn, 0, 1, true},
    {"prog_lin_add", &sgnn_tune::prog_lin_add, 0, 1, true},
};
const Field *find(const char *name) {
  if (!name) return nullptr;
  for (const Field &f : kFields)
    if (strcmp(f.name, name) == 0) return &f;
  return nullptr;
}
}  // namespace

SGNN_EXPORT int64_t sgnn_tune_get(const char *name) {
  const Field *f = find(name);
  if (!f) {
    sgnn_set_error("sgnn_tune_get: no switch named '%s' (sgnn_tune_names lists them)", name ? name : "(null)");
    return SGNN_TUNE_UNKNOWN;
  }
  return g_tune.*(f->ptr);
}

SGNN_EXPORT int64_t sgnn_tune_set(const char *name, int64_t value) {
  const Field *f = find(name);
  if (!f) {
    sgnn_set_error("sgnn_tune_set: no switch named '%s' (sgnn_tune_names lists them)", name ? name : "(null)");
    return SGNN_TUNE_UNKNOWN;
  }
  const int64_t prev = g_tune.*(f->ptr);
  if (f->boolean) value = value != 0;
  if (value < f->lo || value > f->hi) {
    sgnn_set_error("sgnn_tune_set: %s = %lld is outside [%lld, %lld]", name, (long long)value, (long long)f->lo, (long long)f->hi);
    return SGNN_TUNE_UNKNOWN;
  }
  g_tune.*(f->ptr) = value;
  return prev;
}

SGNN_EXPORT const char *sgnn_tune_names(void) {
  static char buf[512];
  if (!buf[0]) {
    size_t n = 0;
    for (const Field &f : kFields) n += (size_t)snprintf(buf + n, sizeof(buf) - n, "%s%s", n ? "," : "", f.name);
  }
  return buf;
}

SGNN_EXPORT const sgnn_tune *sgnn_tune_current(void) { return &g_tune; }
