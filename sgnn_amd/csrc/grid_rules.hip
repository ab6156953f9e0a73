// Voxel hash grid, rulebook construction and stable compaction for gfx950.
//
// Replaces the host-side hash maps / rulebook builders of the sparse-op library the
// reference imports (torch/model.py:7).
// Everything here is integer work bounded by HBM / L2 latency: one thread per site,
// coalesced offset-major table writes, 64-bit keys in an open-addressing table that
// stays L2/MALL resident, wave64 ballot + popcount prefix sums for the compactions.
#include <stdarg.h>
#include "common.h"

// ---------------------------------------------------------------------------
// error plumbing (shared by all translation units)
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void sgnn_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

SGNN_EXPORT const char *sgnn_last_error(void) { return g_err; }
SGNN_EXPORT int sgnn_version(void) { return 100; }
SGNN_EXPORT const char *sgnn_arch(void) { return "gfx950"; }

SGNN_EXPORT int64_t sgnn_hash_capacity(int64_t n) {
  // load factor <= 0.5; <= 0.25 below 64 k sites: there a launch is latency-bound and what a probing kernel costs is the
  // LENGTH of the longest probe sequence in a wave — every step of it one more dependent memory round trip (1.5-2 us) — not
  // the table's cache footprint.  k_rulebook_subm3 on 2.5 k / 9 k / 33 k rows: 19.0 / 18.6 / 21.8 -> 13.5 / 11.1 / 12.5 us
  // (round 5; rocprofv3, isolated launches).  What remains is the chain coords -> keys -> values -> walk -> stores itself.
  // Monotone in n (a table sized from an upper bound must never be smaller than one sized from a live count):
  // target = max(2 n, min(4 n, 262 144)) — 4 n below 64 k sites as measured, flat at 256 k slots up to 128 k, 2 n beyond.
  const int64_t want = 4 * n < 262144 ? 4 * n : (2 * n > 262144 ? 2 * n : 262144);
  int64_t cap = 1024;
  while (cap < want) cap <<= 1;
  return cap;
}

// ---------------------------------------------------------------------------
// coordinate conversion
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_coords_from_i64(const int64_t *__restrict__ locs, int64_t n,
                                                        int4 *__restrict__ coords, int32_t *status,
                                                        const int64_t *n_dev) {
  n = sgnn_dyn_n(n, n_dev);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  bool bad = false;
  for (; i < n; i += stride) {
    const longlong2 a = reinterpret_cast<const longlong2 *>(locs)[2 * i];
    const longlong2 b = reinterpret_cast<const longlong2 *>(locs)[2 * i + 1];
    bad |= (a.x < 0 || a.x > 65535 || a.y < 0 || a.y > 65535 || b.x < 0 || b.x > 65535 || b.y < 0 ||
            b.y > 32767);
    coords[i] = make_int4((int)a.x, (int)a.y, (int)b.x, (int)b.y);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(status, SGNN_STATUS_COORD_RANGE);
}

__global__ __launch_bounds__(256) void k_coords_to_i64(const int4 *__restrict__ coords, int64_t n,
                                                      int64_t *__restrict__ locs, const int64_t *n_dev) {
  n = sgnn_dyn_n(n, n_dev);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    const int4 c = coords[i];
    reinterpret_cast<longlong2 *>(locs)[2 * i] = make_longlong2(c.x, c.y);
    reinterpret_cast<longlong2 *>(locs)[2 * i + 1] = make_longlong2(c.z, c.w);
  }
}

SGNN_EXPORT int sgnn_coords_from_i64(const int64_t *locs, int64_t n, int32_t *coords, int32_t *status,
                                     const int64_t *n_dev, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && status != nullptr);
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(locs && coords);
  SGNN_LAUNCH(k_coords_from_i64, dim3(sgnn_grid_for(n, 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, locs, n, (int4 *)coords, status, n_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_coords_to_i64(const int32_t *coords, int64_t n, int64_t *locs, const int64_t *n_dev,
                                   sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0);
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(locs && coords);
  SGNN_LAUNCH(k_coords_to_i64, dim3(sgnn_grid_for(n, 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, (const int4 *)coords, n, locs, n_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// ---------------------------------------------------------------------------
// hash build / lookup
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hash_build(const int4 *__restrict__ coords, int64_t n,
                                                   unsigned long long *__restrict__ keys,
                                                   int32_t *__restrict__ vals, uint64_t mask,
                                                   int32_t *status, const int64_t *n_dev) {
  n = sgnn_dyn_n(n, n_dev);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    const int4 c = coords[i];
    const uint64_t key = sgnn_pack_key(c.x, c.y, c.z, c.w);
    uint64_t slot = sgnn_hash64(key) & mask;
    while (true) {
      const unsigned long long prev = atomicCAS(&keys[slot], SGNN_EMPTY_KEY, (unsigned long long)key);
      if (prev == SGNN_EMPTY_KEY) {
        vals[slot] = (int32_t)i;
        break;
      }
      if (prev == key) {
        atomicOr(status, SGNN_STATUS_DUPLICATE);
        break;
      }
      slot = (slot + 1) & mask;
    }
  }
}

SGNN_EXPORT int sgnn_hash_build(const int32_t *coords, int64_t n, uint64_t *keys, int32_t *vals,
                                int64_t cap, int32_t *status, const int64_t *n_dev, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && keys && vals && status);
  SGNN_CHECK_ARG(cap >= 2 * n && cap >= 2 && (cap & (cap - 1)) == 0);
  if (n >= (1ll << 31)) {
    sgnn_set_error("sgnn_hash_build: %lld sites exceed the 31-bit row index", (long long)n);
    return SGNN_EOVERFLOW;
  }
  if (sgnn_fill32(keys, 0xFFFFFFFFu, cap * 2, (hipStream_t)stream) != SGNN_OK) return SGNN_EHIP;
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(coords);
  SGNN_LAUNCH(k_hash_build, dim3(sgnn_grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const int4 *)coords, n, (unsigned long long *)keys, vals, (uint64_t)(cap - 1),
                     status, n_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

__global__ __launch_bounds__(256) void k_hash_lookup(const uint64_t *__restrict__ keys,
                                                    const int32_t *__restrict__ vals, uint64_t mask,
                                                    const int4 *__restrict__ query, int64_t m,
                                                    int32_t *__restrict__ rows, const int64_t *m_dev) {
  m = sgnn_dyn_n(m, m_dev);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < m; i += stride) {
    const int4 c = query[i];
    const bool ok = ((unsigned)c.x <= 65535u) && ((unsigned)c.y <= 65535u) && ((unsigned)c.z <= 65535u) &&
                    ((unsigned)c.w <= 32767u);
    rows[i] = ok ? sgnn_hash_find(keys, vals, mask, sgnn_pack_key(c.x, c.y, c.z, c.w)) : -1;
  }
}

SGNN_EXPORT int sgnn_hash_lookup(const uint64_t *keys, const int32_t *vals, int64_t cap,
                                 const int32_t *query, int64_t m, int32_t *rows, const int64_t *m_dev,
                                 sgnn_stream_t stream) {
  SGNN_CHECK_ARG(m >= 0 && keys && vals && cap >= 2 && (cap & (cap - 1)) == 0);
  if (m == 0) return SGNN_OK;
  SGNN_CHECK_ARG(query && rows);
  SGNN_LAUNCH(k_hash_lookup, dim3(sgnn_grid_for(m, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     keys, vals, (uint64_t)(cap - 1), (const int4 *)query, m, rows, m_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// Capacity mode: tables are sized for the level's capacity, but only the entries of rows [0, roundup256(live rows)) are
// ever read (the convolution kernels exit per 256-row tile past the live count), so padding / pre-fill writes stop
// there instead of costing capacity-proportional traffic.
__device__ __forceinline__ int64_t pad_end(int64_t n, int64_t ld) {
  const int64_t e = (n + 255) & ~int64_t(255);
  return e < ld ? e : ld;
}

// rows x [0, pad_end(n)) entries of an offset-major int32 table := -1
__global__ __launch_bounds__(256) void k_fill_rows_dyn(int32_t *__restrict__ t, int rows, int64_t ld, int64_t n,
                                                      const int64_t *n_dev) {
  const int64_t end = pad_end(sgnn_dyn_n(n, n_dev), ld);
  const int64_t total = end * rows, stride = (int64_t)gridDim.x * 256;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) t[(g / end) * ld + (g % end)] = -1;
}

// ---------------------------------------------------------------------------
// 3x3x3 submanifold rulebook: nbr[k][j].  One thread per site.  The rulebook is symmetric —
// nbr[k][j] = i  <=>  nbr[26-k][i] = j — so only the 13 "lower" offsets are probed in the hash; each hit
// also writes its mirror entry (unique writer per entry, no atomics).  Rows 14..26 are pre-filled with -1.
// Probes are the cost (random L2 lines); the direct per-offset stores are coalesced 256-B runs.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void rulebook_subm3_site(const uint64_t *__restrict__ keys, const int32_t *__restrict__ vals,
                                                    uint64_t mask, const int4 *__restrict__ coords, int64_t n,
                                                    int32_t *__restrict__ nbr, int64_t ld, int64_t j) {
  if (j >= pad_end(n, ld)) return;
  if (j >= n) {  // padding entries: the conv kernels rely on them being -1 (rows 14..26 come from the memset)
#pragma unroll
    for (int k = 0; k <= 13; ++k) nbr[(int64_t)k * ld + j] = -1;
    return;
  }
  const int4 c = coords[j];
  // the 13 probes are independent random L2 lines: issue all first-slot key loads, then all value loads of the hits,
  // and only walk on for the (rare, load factor <= 0.5) collisions — 2 dependent round trips instead of up to 26
  uint64_t key[13], slot[13], got[13];
#pragma unroll
  for (int k = 0; k < 13; ++k) {
    const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
    const int z = c.x + dz, y = c.y + dy, x = c.z + dx;
    const bool ok = ((unsigned)z <= 65535u) && ((unsigned)y <= 65535u) && ((unsigned)x <= 65535u);
    key[k] = ok ? sgnn_pack_key(z, y, x, c.w) : SGNN_EMPTY_KEY;     // the empty key never matches a stored one
    slot[k] = sgnn_hash64(key[k]) & mask;
  }
#pragma unroll
  for (int k = 0; k < 13; ++k) got[k] = key[k] == SGNN_EMPTY_KEY ? SGNN_EMPTY_KEY : keys[slot[k]];
  int32_t r[13];
#pragma unroll
  for (int k = 0; k < 13; ++k) r[k] = (got[k] == key[k] && key[k] != SGNN_EMPTY_KEY) ? vals[slot[k]] : -1;
#pragma unroll
  for (int k = 0; k < 13; ++k) {
    if (got[k] != key[k] && got[k] != SGNN_EMPTY_KEY) {              // first slot held another key: keep probing
      uint64_t sl = (slot[k] + 1) & mask;
      while (true) {
        const uint64_t kk = keys[sl];
        if (kk == key[k]) {
          r[k] = vals[sl];
          break;
        }
        if (kk == SGNN_EMPTY_KEY) break;
        sl = (sl + 1) & mask;
      }
    }
    nbr[(int64_t)k * ld + j] = r[k];
    if (r[k] >= 0) nbr[(int64_t)(26 - k) * ld + r[k]] = (int32_t)j;
  }
  nbr[(int64_t)13 * ld + j] = (int32_t)j;
}

__global__ __launch_bounds__(256) void k_rulebook_subm3(const uint64_t *__restrict__ keys,
                                                       const int32_t *__restrict__ vals, uint64_t mask,
                                                       const int4 *__restrict__ coords, int64_t n,
                                                       int32_t *__restrict__ nbr, int64_t ld, const int64_t *n_dev) {
  rulebook_subm3_site(keys, vals, mask, coords, sgnn_dyn_n(n, n_dev), nbr, ld, (int64_t)blockIdx.x * 256 + threadIdx.x);
}

// Several levels in one launch (the coarse levels of one hierarchy: 2-3 rulebooks of 0.4 k - 40 k rows each, whose launches
// are mostly ramp): the descriptor table travels by value in the kernel arguments, workgroup -> level by the block prefix.
// The site body is the single-level kernel's, so the tables are the same entries.
struct RbMulti {
  const uint64_t *keys[SGNN_RULEBOOK_MULTI_MAX];
  const int32_t *vals[SGNN_RULEBOOK_MULTI_MAX];
  const int4 *coords[SGNN_RULEBOOK_MULTI_MAX];
  int32_t *nbr[SGNN_RULEBOOK_MULTI_MAX];
  const int64_t *n_dev[SGNN_RULEBOOK_MULTI_MAX];
  uint64_t mask[SGNN_RULEBOOK_MULTI_MAX];
  int64_t n[SGNN_RULEBOOK_MULTI_MAX], ld[SGNN_RULEBOOK_MULTI_MAX];
  unsigned blk0[SGNN_RULEBOOK_MULTI_MAX + 1];        // first workgroup of level i; blk0[count] = grid
  int count;
};

__device__ __forceinline__ int rb_multi_level(const RbMulti &b, unsigned blk) {
  int i = 0;
#pragma unroll
  for (int q = 1; q < SGNN_RULEBOOK_MULTI_MAX; ++q) i += (q < b.count && blk >= b.blk0[q]) ? 1 : 0;
  return i;
}

// rows 14..26 of every level := -1 over the live (padded) range — the mirror writes of the builder land there
__global__ __launch_bounds__(256) void k_fill_rows_dyn_multi(RbMulti b) {
  const int i = rb_multi_level(b, blockIdx.x);
  const int64_t ld = b.ld[i];
  const int64_t end = pad_end(sgnn_dyn_n(b.n[i], b.n_dev[i]), ld);
  int32_t *t = b.nbr[i] + 14 * ld;
  const int64_t total = end * 13, stride = (int64_t)(b.blk0[i + 1] - b.blk0[i]) * 256;
  for (int64_t g = (int64_t)(blockIdx.x - b.blk0[i]) * 256 + threadIdx.x; g < total; g += stride)
    t[(g / end) * ld + (g % end)] = -1;
}

__global__ __launch_bounds__(256) void k_rulebook_subm3_multi(RbMulti b) {
  const int i = rb_multi_level(b, blockIdx.x);
  rulebook_subm3_site(b.keys[i], b.vals[i], b.mask[i], b.coords[i], sgnn_dyn_n(b.n[i], b.n_dev[i]), b.nbr[i], b.ld[i],
                      (int64_t)(blockIdx.x - b.blk0[i]) * 256 + threadIdx.x);
}

// (Round 5: a 26-probe variant — every entry written by the site's own thread, no mirror scatter, no pre-fill launch of
//  rows 14..26 — was built for the levels below 32 k rows, produced identical tables, and ran 2-4x as long as this kernel plus
//  its pre-fill (27 x {key, slot, probe} live per thread; 44 vs 16 us at 12.6 k rows, 27 vs 14 us at 2.7 k rows under
//  rocprofv3, profiles/r05w_step_launches.csv) for a launch that costs < 2 us in a replayed graph: deleted.)
// ---------------------------------------------------------------------------
// The same rulebook with the voxel index of a row WINDOW held in LDS (north_star: "hash-table voxel indexing in LDS").
// A workgroup owns 256 consecutive rows; in every site order this pipeline produces — batch-major raster order of the
// input blocks (scene_dataloader.py:13-36), 8-children-per-parent order of the generated levels (model.py:195-207) —
// most of a row's 26 neighbours are rows close by.  The workgroup hashes the coordinates of the rows [row0 - 256,
// row0 + 512) into a 2048-slot LDS table, every site probes its 26 neighbours THERE (a hit is final), and only the
// misses — neighbours that do not exist, or live outside the window — go to the global table.  All 27 table rows are
// written by the site's own thread (coalesced): no mirror scatter, no pre-fill memset.
// ---------------------------------------------------------------------------
#define RB_WIN_BEFORE 256
#define RB_WIN_ROWS 768
#define RB_LDS_SLOTS 2048

__global__ __launch_bounds__(256) void k_rulebook_subm3_lds(const uint64_t *__restrict__ keys,
                                                           const int32_t *__restrict__ vals, uint64_t mask,
                                                           const int4 *__restrict__ coords, int64_t n,
                                                           int32_t *__restrict__ nbr, int64_t ld,
                                                           const int64_t *n_dev) {
  n = sgnn_dyn_n(n, n_dev);
  __shared__ unsigned long long lk[RB_LDS_SLOTS];
  __shared__ int32_t lv[RB_LDS_SLOTS];
  const int tid = threadIdx.x;
  const int64_t row0 = (int64_t)blockIdx.x * 256;
  const int64_t wlo = row0 > RB_WIN_BEFORE ? row0 - RB_WIN_BEFORE : 0;
  const int64_t whi = (wlo + RB_WIN_ROWS) < n ? (wlo + RB_WIN_ROWS) : n;
  for (int e = tid; e < RB_LDS_SLOTS; e += 256) lk[e] = SGNN_EMPTY_KEY;
  __syncthreads();
  for (int64_t i = wlo + tid; i < whi; i += 256) {
    const int4 c = coords[i];
    const unsigned long long key = sgnn_pack_key(c.x, c.y, c.z, c.w);
    unsigned slot = (unsigned)sgnn_hash64(key) & (RB_LDS_SLOTS - 1);
    while (true) {
      const unsigned long long prev = atomicCAS(&lk[slot], SGNN_EMPTY_KEY, key);
      if (prev == SGNN_EMPTY_KEY) {
        lv[slot] = (int32_t)i;
        break;
      }
      if (prev == key) break;                 // duplicate site: the caller error is reported by sgnn_hash_build
      slot = (slot + 1) & (RB_LDS_SLOTS - 1);
    }
  }
  __syncthreads();
  const int64_t j = row0 + tid;
  if (j >= ld) return;
  if (j >= n) {
#pragma unroll
    for (int k = 0; k < 27; ++k) nbr[(int64_t)k * ld + j] = -1;
    return;
  }
  const int4 c = coords[j];
  int32_t r[27];
  unsigned missing = 0;                       // bit k: neighbour k was not in the LDS window
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    if (k == 13) {
      r[k] = (int32_t)j;
      continue;
    }
    const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
    const int z = c.x + dz, y = c.y + dy, x = c.z + dx;
    r[k] = -1;
    if (((unsigned)z > 65535u) || ((unsigned)y > 65535u) || ((unsigned)x > 65535u)) continue;
    const unsigned long long key = sgnn_pack_key(z, y, x, c.w);
    unsigned slot = (unsigned)sgnn_hash64(key) & (RB_LDS_SLOTS - 1);
    bool hit = false;
    while (true) {
      const unsigned long long kk = lk[slot];
      if (kk == key) {
        hit = true;
        break;
      }
      if (kk == SGNN_EMPTY_KEY) break;
      slot = (slot + 1) & (RB_LDS_SLOTS - 1);
    }
    if (hit) r[k] = lv[slot];
    else missing |= 1u << k;
  }
  // the misses: first-slot key loads of all of them in flight together, then the (rare) longer probe walks
  uint64_t gk[27];
  uint64_t gs[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    if (!((missing >> k) & 1u)) continue;
    const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
    const uint64_t key = sgnn_pack_key(c.x + dz, c.y + dy, c.z + dx, c.w);
    gs[k] = sgnn_hash64(key) & mask;
    gk[k] = keys[gs[k]];
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    if (!((missing >> k) & 1u)) continue;
    const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
    const uint64_t key = sgnn_pack_key(c.x + dz, c.y + dy, c.z + dx, c.w);
    uint64_t sl = gs[k], kk = gk[k];
    while (kk != key && kk != SGNN_EMPTY_KEY) {
      sl = (sl + 1) & mask;
      kk = keys[sl];
    }
    if (kk == key) r[k] = vals[sl];
  }
#pragma unroll
  for (int k = 0; k < 27; ++k) nbr[(int64_t)k * ld + j] = r[k];
}

// Measured (scripts/bench_conv.py, MI355X): N = 366 085 raster-ordered sites 123.9 us vs 96.7 us for the global-probe kernel
// (incl. its memset), N = 2 017 264 children-ordered sites 661 vs 498 us: the 26 dependent LDS probe walks + 768 LDS CAS
// inserts per workgroup cost more than the 13 L2-resident global probes they replace, so the global kernel is the default.
// ---------------------------------------------------------------------------
// The same rulebook through a dense index volume: vol[((b*Z + z)*Y + y)*X + x] = row of the site, -1 elsewhere.  A
// neighbour look-up is then ONE 4-byte read next to the reads of the neighbouring threads (sites arrive in raster or
// children order) instead of a hash probe into a random 8-byte key plus the value read, and all 27 entries of a site
// are written by its own thread (coalesced; no mirror scatter, no pre-fill).  The volume is a persistent workspace that
// is all -1 between calls: k_vol_mark writes the rows of this level, the table kernel reads, k_vol_mark clears them
// again (n scattered 4-byte writes each, instead of a memset of the whole volume).  Positions the volume does not
// cover — batch index >= bcap = entries / (Z*Y*X), or a coordinate outside [0, dims) — are looked up in the hash grid,
// and such sites are not marked: the result is the hash rulebook's for every input.
// ---------------------------------------------------------------------------
struct VolDims {
  int Z, Y, X, bcap;
};

__device__ __forceinline__ bool vol_covers(const VolDims d, int z, int y, int x, int b) {
  return (unsigned)z < (unsigned)d.Z && (unsigned)y < (unsigned)d.Y && (unsigned)x < (unsigned)d.X &&
         (unsigned)b < (unsigned)d.bcap;
}

// status != NULL (volume-only rulebooks, sgnn_rulebook_subm3_volume): a site the volume does not cover has no other index
// to be found in — SGNN_STATUS_COORD_RANGE is raised instead of building a table with holes
__global__ __launch_bounds__(256) void k_vol_mark(const int4 *__restrict__ coords, int64_t n, int32_t *__restrict__ vol,
                                                 VolDims d, int clear, const int64_t *n_dev, int32_t *status) {
  n = sgnn_dyn_n(n, n_dev);
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int4 c = coords[j];
  if (vol_covers(d, c.x, c.y, c.z, c.w)) vol[(((int64_t)c.w * d.Z + c.x) * d.Y + c.y) * d.X + c.z] = clear ? -1 : (int32_t)j;
  else if (status) atomicOr(status, SGNN_STATUS_COORD_RANGE);
}

__global__ __launch_bounds__(256) void k_rulebook_subm3_vol(const uint64_t *__restrict__ keys,
                                                           const int32_t *__restrict__ vals, uint64_t mask,
                                                           const int4 *__restrict__ coords, int64_t n,
                                                           const int32_t *__restrict__ vol, VolDims d,
                                                           int32_t *__restrict__ nbr, int64_t ld,
                                                           const int64_t *n_dev) {
  n = sgnn_dyn_n(n, n_dev);
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= pad_end(n, ld)) return;
  if (j >= n) {                                 // padding entries: the conv kernels rely on them being -1
#pragma unroll
    for (int k = 0; k < 27; ++k) nbr[(int64_t)k * ld + j] = -1;
    return;
  }
  const int4 c = coords[j];
  // interior site of a covered block: 26 plain reads, all in flight together
  const bool inner = c.x >= 1 && c.x + 1 < d.Z && c.y >= 1 && c.y + 1 < d.Y && c.z >= 1 && c.z + 1 < d.X &&
                     (unsigned)c.w < (unsigned)d.bcap;
  if (inner) {
    const int32_t *p = vol + (((int64_t)c.w * d.Z + c.x) * d.Y + c.y) * d.X + c.z;
    int32_t r[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
      const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
      r[k] = k == 13 ? (int32_t)j : p[((int64_t)dz * d.Y + dy) * d.X + dx];
    }
#pragma unroll
    for (int k = 0; k < 27; ++k) nbr[(int64_t)k * ld + j] = r[k];
  } else {
#pragma unroll 1
    for (int k = 0; k < 27; ++k) {
      const int dz = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
      const int z = c.x + dz, y = c.y + dy, x = c.z + dx;
      int32_t v = -1;
      if (k == 13) {
        v = (int32_t)j;
      } else if (vol_covers(d, z, y, x, c.w)) {
        v = vol[(((int64_t)c.w * d.Z + z) * d.Y + y) * d.X + x];
      } else if (keys && ((unsigned)z <= 65535u) && ((unsigned)y <= 65535u) && ((unsigned)x <= 65535u)) {   // (keys == NULL: volume only)
        const uint64_t key = sgnn_pack_key(z, y, x, c.w);
        uint64_t sl = sgnn_hash64(key) & mask;
        while (true) {
          const uint64_t kk = keys[sl];
          if (kk == key) {
            v = vals[sl];
            break;
          }
          if (kk == SGNN_EMPTY_KEY) break;
          sl = (sl + 1) & mask;
        }
      }
      nbr[(int64_t)k * ld + j] = v;
    }
  }
}

SGNN_EXPORT int sgnn_rulebook_subm3_dense(const uint64_t *keys, const int32_t *vals, int64_t cap, const int32_t *coords,
                                          int64_t n, int dim_z, int dim_y, int dim_x, int32_t *volume,
                                          int64_t volume_entries, int32_t *nbr, int64_t ld, const int64_t *n_dev,
                                          sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && ld >= n && keys && vals && cap >= 2 && (cap & (cap - 1)) == 0);
  SGNN_CHECK_ARG(dim_z >= 1 && dim_y >= 1 && dim_x >= 1 && dim_z <= 65536 && dim_y <= 65536 && dim_x <= 65536 &&
                 volume_entries >= 0 && (volume || volume_entries == 0));
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(coords && nbr);
  const int64_t slab = (int64_t)dim_z * dim_y * dim_x;
  const int64_t bcap = volume_entries / slab;
  const VolDims d{dim_z, dim_y, dim_x, (int)(bcap > 65536 ? 65536 : bcap)};
  hipStream_t s = (hipStream_t)stream;
  const unsigned gn = (unsigned)((n + 255) / 256);
  if (d.bcap > 0) SGNN_LAUNCH(k_vol_mark, dim3(gn), dim3(256), 0, s, (const int4 *)coords, n, volume, d, 0, n_dev, (int32_t *)nullptr);
  SGNN_LAUNCH(k_rulebook_subm3_vol, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, s, keys, vals,
                     (uint64_t)(cap - 1), (const int4 *)coords, n, volume, d, nbr, ld, n_dev);
  if (d.bcap > 0) SGNN_LAUNCH(k_vol_mark, dim3(gn), dim3(256), 0, s, (const int4 *)coords, n, volume, d, 1, n_dev, (int32_t *)nullptr);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// The same table for a level whose sites are KNOWN to lie inside the volume (generated levels: children of a dense coarse
// volume, torch/model.py:192-207, bounded by the model's own sizes) — no hash grid of the level exists or is built: neighbours
// outside the volume are absent by construction.  A site the volume does not cover breaks that promise and raises
// SGNN_STATUS_COORD_RANGE in *status (the caller's deferred-error word) instead of producing a table with holes.
SGNN_EXPORT int sgnn_rulebook_subm3_volume(const int32_t *coords, int64_t n, int dim_z, int dim_y, int dim_x, int32_t *volume,
                                           int64_t volume_entries, int32_t *nbr, int64_t ld, const int64_t *n_dev,
                                           int32_t *status, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && ld >= n && status && volume);
  SGNN_CHECK_ARG(dim_z >= 1 && dim_y >= 1 && dim_x >= 1 && dim_z <= 65536 && dim_y <= 65536 && dim_x <= 65536);
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(coords && nbr);
  const int64_t slab = (int64_t)dim_z * dim_y * dim_x;
  const int64_t bcap = volume_entries / slab;
  SGNN_CHECK_ARG(bcap >= 1);
  const VolDims d{dim_z, dim_y, dim_x, (int)(bcap > 65536 ? 65536 : bcap)};
  hipStream_t s = (hipStream_t)stream;
  const unsigned gn = (unsigned)((n + 255) / 256);
  SGNN_LAUNCH(k_vol_mark, dim3(gn), dim3(256), 0, s, (const int4 *)coords, n, volume, d, 0, n_dev, status);
  SGNN_LAUNCH(k_rulebook_subm3_vol, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, s, (const uint64_t *)nullptr,
                     (const int32_t *)nullptr, (uint64_t)0, (const int4 *)coords, n, volume, d, nbr, ld, n_dev);
  SGNN_LAUNCH(k_vol_mark, dim3(gn), dim3(256), 0, s, (const int4 *)coords, n, volume, d, 1, n_dev, (int32_t *)nullptr);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}


SGNN_EXPORT int sgnn_rulebook_subm3(const uint64_t *keys, const int32_t *vals, int64_t cap,
                                    const int32_t *coords, int64_t n, int32_t *nbr, int64_t ld,
                                    const int64_t *n_dev, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && ld >= n && keys && vals && cap >= 2 && (cap & (cap - 1)) == 0);
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(coords && nbr);
  if (g_tune.rulebook_lds) {
    SGNN_LAUNCH(k_rulebook_subm3_lds, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, (hipStream_t)stream, keys,
                       vals, (uint64_t)(cap - 1), (const int4 *)coords, n, nbr, ld, n_dev);
    SGNN_CHECK_LAUNCH();
    return SGNN_OK;
  }
  if (n_dev)      // capacity mode: pre-fill only what the live rows can reach
    SGNN_LAUNCH(k_fill_rows_dyn, dim3(sgnn_grid_for(13 * n, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       nbr + 14 * ld, 13, ld, n, n_dev);
  else
    if (sgnn_fill32(nbr + 14 * ld, 0xFFFFFFFFu, 13 * ld, (hipStream_t)stream) != SGNN_OK) return SGNN_EHIP;
  SGNN_LAUNCH(k_rulebook_subm3, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     keys, vals, (uint64_t)(cap - 1), (const int4 *)coords, n, nbr, ld, n_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_rulebook_subm3_multi(int count, const uint64_t *const *keys, const int32_t *const *vals,
                                          const int64_t *caps, const int32_t *const *coords, const int64_t *ns,
                                          int32_t *const *nbrs, const int64_t *lds, const int64_t *const *n_devs,
                                          sgnn_stream_t stream) {
  SGNN_CHECK_ARG(count >= 0 && count <= SGNN_RULEBOOK_MULTI_MAX);
  if (count == 0) return SGNN_OK;
  SGNN_CHECK_ARG(keys && vals && caps && coords && ns && nbrs && lds && n_devs);
  if (!g_tune.rulebook_multi || g_tune.rulebook_lds) {
    for (int i = 0; i < count; ++i) {
      SGNN_CHECK_ARG(ns[i] == 0 || n_devs[i]);
      const int rc = sgnn_rulebook_subm3(keys[i], vals[i], caps[i], coords[i], ns[i], nbrs[i], lds[i], n_devs[i], stream);
      if (rc != SGNN_OK) return rc;
    }
    return SGNN_OK;
  }
  RbMulti fill{}, rb{};
  unsigned fblk = 0, rblk = 0;
  int m = 0;
  for (int i = 0; i < count; ++i) {
    const int64_t n = ns[i], ld = lds[i], cap = caps[i];
    SGNN_CHECK_ARG(n >= 0 && ld >= n && keys[i] && vals[i] && cap >= 2 && (cap & (cap - 1)) == 0);
    if (n == 0) continue;
    SGNN_CHECK_ARG(coords[i] && nbrs[i] && n_devs[i]);       // capacity-mode entry: every level carries its device count
    fill.nbr[m] = rb.nbr[m] = nbrs[i];
    fill.n[m] = rb.n[m] = n;
    fill.ld[m] = rb.ld[m] = ld;
    fill.n_dev[m] = rb.n_dev[m] = n_devs[i];
    rb.keys[m] = keys[i];
    rb.vals[m] = vals[i];
    rb.mask[m] = (uint64_t)(cap - 1);
    rb.coords[m] = (const int4 *)coords[i];
    fill.blk0[m] = fblk;
    rb.blk0[m] = rblk;
    fblk += (unsigned)sgnn_grid_for(13 * n, 256, 4096);
    rblk += (unsigned)((ld + 255) / 256);
    ++m;
  }
  if (m == 0) return SGNN_OK;
  fill.blk0[m] = fblk;
  rb.blk0[m] = rblk;
  fill.count = rb.count = m;
  SGNN_LAUNCH(k_fill_rows_dyn_multi, dim3(fblk), dim3(256), 0, (hipStream_t)stream, fill);
  SGNN_LAUNCH(k_rulebook_subm3_multi, dim3(rblk), dim3(256), 0, (hipStream_t)stream, rb);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// ---------------------------------------------------------------------------
// stable compaction machinery: count -> scan of block sums -> write
// ---------------------------------------------------------------------------
#define SCAN_ITEMS 8
#define SCAN_BLOCK (256 * SCAN_ITEMS)

struct FlagSigmoid {
  const float *logits;
  int64_t stride;
  __device__ __forceinline__ bool operator()(int64_t i) const {
    // identical predicate to the reference's `nn.Sigmoid()(out) > 0.5` (torch/model.py:233,322)
    const float x = logits[i * stride];
    return (1.0f / (1.0f + expf(-x))) > 0.5f;
  }
};
struct FlagMask {
  const uint8_t *mask;
  __device__ __forceinline__ bool operator()(int64_t i) const { return mask[i] != 0; }
};
struct FlagDense {  // site i is kept iff a dense (B,1,d0,d1,d2) volume is > 0.5 at its coordinates (teacher forcing)
  const int4 *coords;
  const float *vol;
  int batch, d0, d1, d2;
  __device__ __forceinline__ bool operator()(int64_t i) const {
    const int4 c = coords[i];
    if ((unsigned)c.x >= (unsigned)d0 || (unsigned)c.y >= (unsigned)d1 || (unsigned)c.z >= (unsigned)d2 ||
        (unsigned)c.w >= (unsigned)batch)
      return false;
    return vol[(((int64_t)c.w * d0 + c.x) * d1 + c.y) * d2 + c.z] > 0.5f;
  }
};
struct FlagOwner {  // fine site i owns its parent iff it is the smallest row that touched it
  const int32_t *slot_of;
  const int32_t *cvals;
  __device__ __forceinline__ bool operator()(int64_t i) const { return cvals[slot_of[i]] == (int32_t)i; }
};

template <class F>
__global__ __launch_bounds__(256) void k_scan_count(F flag, int64_t n, int32_t *__restrict__ block_sums,
                                                   const int64_t *n_dev) {
  n = sgnn_dyn_n(n, n_dev);
  __shared__ int lds[4];
  const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK;
  int cnt = 0;
#pragma unroll
  for (int it = 0; it < SCAN_ITEMS; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    const bool f = (i < n) && flag(i);
    cnt += __popcll(__ballot(f));
  }
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// single workgroup: exclusive scan of block_sums in place, total -> *count.  Capacity mode (lim.cap >= 0): the
// count is clamped to the capacity of the buffers that will hold the selected rows (SGNN_STATUS_OVERFLOW is raised
// when it had to be), and lim.mul > 0 also publishes count * mul (rows of the 8-child expansion) at lim.count_mul.
struct ScanLimit {
  int64_t cap;
  int64_t *count_mul;
  int mul;
  int32_t *status;
};
static const ScanLimit kNoLimit{-1, nullptr, 0, nullptr};

// Round 5: the write kernels derive their offsets from the RAW block sums themselves where the table is short
// (nblk <= SCAN_INLINE_MAX: at most one 16 KB, L2-resident read per workgroup), so the single-workgroup scan launch between
// the count and the write kernel disappears (15 launches of a configs[1] training step: 11 stride-2 levels + 4 mask
// compactions).  Integer sums: the offsets and counts are the scan kernel's, bit for bit.  ScanInline.nblk == 0 keeps the
// three-launch form (block_sums already hold exclusive offsets).
#define SCAN_INLINE_MAX 4096
static inline bool scan_inline_ok(int64_t nblk) { return g_tune.scan_inline && nblk >= 1 && nblk <= SCAN_INLINE_MAX; }

struct ScanInline {
  int64_t nblk;      // > 0: block_sums are raw counts, this many of them
  int64_t *count;    // where workgroup 0 publishes the (clamped) total
  ScanLimit lim;
};
static const ScanInline kNoInline{0, nullptr, {-1, nullptr, 0, nullptr}};

// before = sum of sums[0 .. b), total = sum of all; lds8: 8 ints of LDS (free again on return).  Call from all threads.
__device__ __forceinline__ void scan_offsets_inline(const int32_t *sums, int64_t nblk, int64_t b, int *lds8, int &before,
                                                    int64_t &total) {
  int sb = 0, sa = 0;
  for (int64_t i = threadIdx.x; i < nblk; i += 256) {
    const int v = sums[i];
    sa += v;
    sb += (i < b) ? v : 0;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    sa += __shfl_xor(sa, d);
    sb += __shfl_xor(sb, d);
  }
  if ((threadIdx.x & 63) == 0) {
    lds8[threadIdx.x >> 6] = sb;
    lds8[4 + (threadIdx.x >> 6)] = sa;
  }
  __syncthreads();
  before = lds8[0] + lds8[1] + lds8[2] + lds8[3];
  total = (int64_t)lds8[4] + lds8[5] + lds8[6] + lds8[7];
  __syncthreads();
}

// what k_scan_block_sums' last thread does with the total: clamp, flag, publish.  Returns the clamped count; only the
// publishing thread (workgroup 0, thread 0) writes.
__device__ __forceinline__ int64_t scan_publish(int64_t total, int64_t *count, const ScanLimit &lim, bool writer) {
  int64_t c = total;
  const bool over = lim.cap >= 0 && c > lim.cap;
  if (over) c = lim.cap;
  if (writer) {
    if (over && lim.status) atomicOr(lim.status, SGNN_STATUS_OVERFLOW);
    *count = c;
    if (lim.count_mul) *lim.count_mul = c * lim.mul;
  }
  return c;
}

__global__ __launch_bounds__(1024) void k_scan_block_sums(int32_t *__restrict__ block_sums, int64_t nblk,
                                                         int64_t *__restrict__ count, ScanLimit lim) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < nblk; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const int v = (i < nblk) ? block_sums[i] : 0;
    // wave inclusive scan
    int s = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(s, d);
      if (lane >= d) s += t;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int tot = 0;
    for (int w = 0; w < 16; ++w) tot += wsum[w];
    const int carry = carry_s;
    if (i < nblk) block_sums[i] = carry + woff + s - v;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int64_t c = (int64_t)carry_s;
    if (lim.cap >= 0 && c > lim.cap) {
      c = lim.cap;
      if (lim.status) atomicOr(lim.status, SGNN_STATUS_OVERFLOW);
    }
    *count = c;
    if (lim.count_mul) *lim.count_mul = c * lim.mul;
  }
}

template <class F, class Emit>
__global__ __launch_bounds__(256) void k_scan_emit(F flag, Emit emit, int64_t n,
                                                  const int32_t *block_offsets, const int64_t *n_dev, ScanInline si) {
  n = sgnn_dyn_n(n, n_dev);
  __shared__ int lds[8];
  const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK;
  int running;
  if (si.nblk > 0) {     // raw block sums: this workgroup's offset and the total, no scan launch (uniform over the launch)
    int64_t total;
    scan_offsets_inline(block_offsets, si.nblk, blockIdx.x, lds, running, total);
    scan_publish(total, si.count, si.lim, blockIdx.x == 0 && threadIdx.x == 0);
  } else {
    running = block_offsets[blockIdx.x];
  }
#pragma unroll 1
  for (int it = 0; it < SCAN_ITEMS; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    const bool f = (i < n) && flag(i);
    int total;
    const int r = sgnn_block_rank256(f, lds, total);
    if (f) emit(i, running + r);
    running += total;
  }
}

struct EmitSel {
  int32_t *sel;
  __device__ __forceinline__ void operator()(int64_t i, int rank) const { sel[rank] = (int32_t)i; }
};

struct EmitSelLocs {   // + the kept sites' coordinates, as sgnn_gather_rows_dn(coords, sel) would copy them (rows past the
  int32_t *sel;        //   capacity of `locs` are dropped: such a step is flagged SGNN_STATUS_OVERFLOW and discarded)
  const int4 *coords;
  int4 *locs;
  int64_t cap;
  __device__ __forceinline__ void operator()(int64_t i, int rank) const {
    sel[rank] = (int32_t)i;
    if (rank < cap) locs[rank] = coords[i];
  }
};

SGNN_EXPORT int64_t sgnn_compact_ws_bytes(int64_t n) {
  const int64_t nblk = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
  return (nblk + 1) * (int64_t)sizeof(int32_t) + 64;
}

template <class F, class Emit>
static int compact_impl(F flag, Emit emit, int64_t n, int64_t *count, void *ws, int64_t ws_bytes,
                        hipStream_t s, const int64_t *n_dev = nullptr, const ScanLimit &lim = kNoLimit) {
  if (n == 0) {
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int64_t), s);
    if (e != hipSuccess) {
      sgnn_set_error("compact: memset failed: %s", hipGetErrorString(e));
      return SGNN_EHIP;
    }
    return SGNN_OK;
  }
  if (ws_bytes < sgnn_compact_ws_bytes(n) || !ws) {
    sgnn_set_error("compact: workspace too small (%lld < %lld)", (long long)ws_bytes,
                   (long long)sgnn_compact_ws_bytes(n));
    return SGNN_ENOWS;
  }
  const int64_t nblk = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
  int32_t *block_sums = (int32_t *)ws;
  SGNN_LAUNCH((k_scan_count<F>), dim3((unsigned)nblk), dim3(256), 0, s, flag, n, block_sums, n_dev);
  const bool inl = scan_inline_ok(nblk);
  if (!inl) SGNN_LAUNCH(k_scan_block_sums, dim3(1), dim3(1024), 0, s, block_sums, nblk, count, lim);
  SGNN_LAUNCH((k_scan_emit<F, Emit>), dim3((unsigned)nblk), dim3(256), 0, s, flag, emit, n,
                     (const int32_t *)block_sums, n_dev, inl ? ScanInline{nblk, count, lim} : kNoInline);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    sgnn_set_error("compact: HIP error: %s", hipGetErrorString(e));
    return SGNN_EHIP;
  }
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_compact_sigmoid(const float *logits, int64_t stride, int64_t n, int32_t *sel,
                                     int64_t *count, void *ws, int64_t ws_bytes, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && count && stride >= 1 && n < (1ll << 31));
  SGNN_CHECK_ARG(n == 0 || (logits && sel));
  return compact_impl(FlagSigmoid{logits, stride}, EmitSel{sel}, n, count, ws, ws_bytes, (hipStream_t)stream);
}

// Capacity mode of the two generative mask compactions: the candidate count comes from device memory (*n_dev, NULL =
// n), the kept count is clamped to keep_cap (SGNN_STATUS_OVERFLOW in *status otherwise) and count[1] = 8 * count[0] is
// published for the 8-child expansion of the kept sites — everything a following stage needs without a host round trip.
SGNN_EXPORT int sgnn_compact_sigmoid_cap(const float *logits, int64_t stride, int64_t n, const int64_t *n_dev,
                                         int32_t *sel, int64_t *count2, int64_t keep_cap, int32_t *status, void *ws,
                                         int64_t ws_bytes, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && count2 && stride >= 1 && n < (1ll << 31) && keep_cap >= 0 && status);
  SGNN_CHECK_ARG(n == 0 || (logits && sel));
  if (n == 0) SGNN_HIP_TRY(hipMemsetAsync(count2, 0, 2 * sizeof(int64_t), (hipStream_t)stream));
  return compact_impl(FlagSigmoid{logits, stride}, EmitSel{sel}, n, count2, ws, ws_bytes, (hipStream_t)stream, n_dev,
                      ScanLimit{keep_cap, count2 + 1, 8, status});
}

SGNN_EXPORT int sgnn_compact_dense_cap(const int32_t *coords, int64_t n, const int64_t *n_dev, const float *vol, int batch,
                                       int d0, int d1, int d2, int32_t *sel, int64_t *count2, int64_t keep_cap,
                                       int32_t *status, void *ws, int64_t ws_bytes, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && count2 && n < (1ll << 31) && batch >= 0 && d0 >= 0 && d1 >= 0 && d2 >= 0 && keep_cap >= 0 &&
                 status);
  SGNN_CHECK_ARG(n == 0 || (coords && vol && sel));
  if (n == 0) SGNN_HIP_TRY(hipMemsetAsync(count2, 0, 2 * sizeof(int64_t), (hipStream_t)stream));
  return compact_impl(FlagDense{(const int4 *)coords, vol, batch, d0, d1, d2}, EmitSel{sel}, n, count2, ws, ws_bytes,
                      (hipStream_t)stream, n_dev, ScanLimit{keep_cap, count2 + 1, 8, status});
}

// The same two compactions, also writing locs[r] = coords[sel[r]] for the kept rows r < keep_cap — what the caller otherwise
// gathers with a launch of its own (sgnn_gather_rows_dn) on the critical path between two generative stages.
SGNN_EXPORT int sgnn_compact_sigmoid_cap_locs(const float *logits, int64_t stride, int64_t n, const int64_t *n_dev,
                                              const int32_t *coords, int32_t *sel, int32_t *locs, int64_t *count2,
                                              int64_t keep_cap, int32_t *status, void *ws, int64_t ws_bytes,
                                              sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && count2 && stride >= 1 && n < (1ll << 31) && keep_cap >= 0 && status);
  SGNN_CHECK_ARG(n == 0 || (logits && sel && coords && (locs || keep_cap == 0)));
  if (n == 0) SGNN_HIP_TRY(hipMemsetAsync(count2, 0, 2 * sizeof(int64_t), (hipStream_t)stream));
  return compact_impl(FlagSigmoid{logits, stride}, EmitSelLocs{sel, (const int4 *)coords, (int4 *)locs, keep_cap}, n, count2,
                      ws, ws_bytes, (hipStream_t)stream, n_dev, ScanLimit{keep_cap, count2 + 1, 8, status});
}

SGNN_EXPORT int sgnn_compact_dense_cap_locs(const int32_t *coords, int64_t n, const int64_t *n_dev, const float *vol,
                                            int batch, int d0, int d1, int d2, int32_t *sel, int32_t *locs, int64_t *count2,
                                            int64_t keep_cap, int32_t *status, void *ws, int64_t ws_bytes,
                                            sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && count2 && n < (1ll << 31) && batch >= 0 && d0 >= 0 && d1 >= 0 && d2 >= 0 && keep_cap >= 0 &&
                 status);
  SGNN_CHECK_ARG(n == 0 || (coords && vol && sel && (locs || keep_cap == 0)));
  if (n == 0) SGNN_HIP_TRY(hipMemsetAsync(count2, 0, 2 * sizeof(int64_t), (hipStream_t)stream));
  return compact_impl(FlagDense{(const int4 *)coords, vol, batch, d0, d1, d2},
                      EmitSelLocs{sel, (const int4 *)coords, (int4 *)locs, keep_cap}, n, count2, ws, ws_bytes,
                      (hipStream_t)stream, n_dev, ScanLimit{keep_cap, count2 + 1, 8, status});
}

// keep site i iff vol[b, z, y, x] > 0.5 at coords[i] = {z, y, x, b}: the generative masks taken from the TARGET
// occupancy pyramid instead of the predicted one (teacher forcing; bench.py uses it so that per-level row counts do
// not depend on the random initial weights)
SGNN_EXPORT int sgnn_compact_dense(const int32_t *coords, int64_t n, const float *vol, int batch, int d0, int d1, int d2,
                                   int32_t *sel, int64_t *count, void *ws, int64_t ws_bytes, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && count && n < (1ll << 31) && batch >= 0 && d0 >= 0 && d1 >= 0 && d2 >= 0);
  SGNN_CHECK_ARG(n == 0 || (coords && vol && sel));
  return compact_impl(FlagDense{(const int4 *)coords, vol, batch, d0, d1, d2}, EmitSel{sel}, n, count, ws, ws_bytes,
                      (hipStream_t)stream);
}

SGNN_EXPORT int sgnn_compact_mask(const uint8_t *mask, int64_t n, int32_t *sel, int64_t *count, void *ws,
                                  int64_t ws_bytes, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && count && n < (1ll << 31));
  SGNN_CHECK_ARG(n == 0 || (mask && sel));
  return compact_impl(FlagMask{mask}, EmitSel{sel}, n, count, ws, ws_bytes, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------
// stride-2 rulebook
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_down2_insert(const int4 *__restrict__ fine, int64_t nf,
                                                     unsigned long long *__restrict__ ckeys,
                                                     int32_t *__restrict__ cvals, uint64_t mask,
                                                     int32_t *__restrict__ slot_of) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < nf; i += stride) {
    const int4 c = fine[i];
    const uint64_t key = sgnn_pack_key(c.x >> 1, c.y >> 1, c.z >> 1, c.w);
    uint64_t slot = sgnn_hash64(key) & mask;
    while (true) {
      const unsigned long long prev = atomicCAS(&ckeys[slot], SGNN_EMPTY_KEY, (unsigned long long)key);
      if (prev == SGNN_EMPTY_KEY || prev == key) break;
      slot = (slot + 1) & mask;
    }
    atomicMin(&cvals[slot], (int32_t)i);  // first-touch owner = smallest fine row
    slot_of[i] = (int32_t)slot;
  }
}

struct EmitOwner {  // owner i gets coarse row `rank`
  const int4 *fine;
  int4 *coarse;
  int32_t *rank_at;
  __device__ __forceinline__ void operator()(int64_t i, int rank) const {
    const int4 c = fine[i];
    coarse[rank] = make_int4(c.x >> 1, c.y >> 1, c.z >> 1, c.w);
    rank_at[i] = rank;
  }
};

__global__ __launch_bounds__(256) void k_down2_parent(int64_t nf, const int32_t *__restrict__ cvals,
                                                     const int32_t *__restrict__ rank_at,
                                                     const int32_t *__restrict__ slot_of,
                                                     int32_t *__restrict__ parent) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < nf; i += stride) parent[i] = rank_at[cvals[slot_of[i]]];
}

// after every parent[] is known: turn the coarse table's values from "owner fine row"
// into "coarse row" so it is an ordinary grid hash for the next level
__global__ __launch_bounds__(256) void k_down2_fix_vals(int64_t nf, const int32_t *__restrict__ slot_of,
                                                       const int32_t *__restrict__ parent,
                                                       const int32_t *__restrict__ rank_at,
                                                       int32_t *__restrict__ cvals) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < nf; i += stride) {
    // exactly one fine row per coarse site is its owner: cvals[slot] == i before this kernel.
    // Owners are identified through rank_at (written only for owners, -1 elsewhere).
    if (rank_at[i] >= 0) cvals[slot_of[i]] = parent[i];
  }
}

SGNN_EXPORT int64_t sgnn_down2_ws_bytes(int64_t nf) {
  const int64_t nblk = (nf + SCAN_BLOCK - 1) / SCAN_BLOCK;
  // slot_of[nf] + rank_at[nf] + block sums
  return 2 * nf * (int64_t)sizeof(int32_t) + (nblk + 1) * (int64_t)sizeof(int32_t) + 256;
}

// one launch initialises the coarse hash (keys = empty, values = INT_MAX) and the rank scratch (-1)
__global__ __launch_bounds__(256) void k_down2_init(unsigned long long *__restrict__ ckeys, int32_t *__restrict__ cvals,
                                                   int64_t ccap, int32_t *__restrict__ rank_at, int64_t nf) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t top = ccap > nf ? ccap : nf;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < top; i += stride) {
    if (i < ccap) {
      ckeys[i] = ~0ull;
      cvals[i] = 0x7FFFFFFF;
    }
    if (i < nf) rank_at[i] = -1;
  }
}

SGNN_EXPORT int sgnn_rulebook_down2(const int32_t *fine_coords, int64_t nf, uint64_t *ckeys, int32_t *cvals,
                                    int64_t ccap, int32_t *parent, int32_t *coarse_coords,
                                    int64_t *n_coarse, void *ws, int64_t ws_bytes, sgnn_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  SGNN_CHECK_ARG(nf >= 0 && n_coarse && ckeys && cvals);
  SGNN_CHECK_ARG(ccap >= 2 * nf && ccap >= 2 && (ccap & (ccap - 1)) == 0 && ccap < (1ll << 31));
  if (nf == 0) {
    SGNN_HIP_TRY(hipMemsetAsync(ckeys, 0xFF, (size_t)ccap * sizeof(uint64_t), s));
    SGNN_HIP_TRY(hipMemsetAsync(n_coarse, 0, sizeof(int64_t), s));
    return SGNN_OK;
  }
  SGNN_CHECK_ARG(fine_coords && parent && coarse_coords);
  if (!ws || ws_bytes < sgnn_down2_ws_bytes(nf)) {
    sgnn_set_error("sgnn_rulebook_down2: workspace too small");
    return SGNN_ENOWS;
  }
  const int64_t nblk = (nf + SCAN_BLOCK - 1) / SCAN_BLOCK;
  int32_t *slot_of = (int32_t *)ws;
  int32_t *rank_at = slot_of + nf;
  int32_t *block_sums = rank_at + nf;
  SGNN_LAUNCH(k_down2_init, dim3(sgnn_grid_for(ccap, 256, 4096)), dim3(256), 0, s, (unsigned long long *)ckeys,
                     cvals, ccap, rank_at, nf);
  const int g = sgnn_grid_for(nf, 256, 8192);
  SGNN_LAUNCH(k_down2_insert, dim3(g), dim3(256), 0, s, (const int4 *)fine_coords, nf,
                     (unsigned long long *)ckeys, cvals, (uint64_t)(ccap - 1), slot_of);
  FlagOwner flag{slot_of, cvals};
  SGNN_LAUNCH((k_scan_count<FlagOwner>), dim3((unsigned)nblk), dim3(256), 0, s, flag, nf, block_sums,
                     (const int64_t *)nullptr);
  const bool inl = scan_inline_ok(nblk);
  if (!inl) SGNN_LAUNCH(k_scan_block_sums, dim3(1), dim3(1024), 0, s, block_sums, nblk, n_coarse, kNoLimit);
  SGNN_LAUNCH((k_scan_emit<FlagOwner, EmitOwner>), dim3((unsigned)nblk), dim3(256), 0, s, flag,
                     EmitOwner{(const int4 *)fine_coords, (int4 *)coarse_coords, rank_at}, nf,
                     (const int32_t *)block_sums, (const int64_t *)nullptr,
                     inl ? ScanInline{nblk, n_coarse, kNoLimit} : kNoInline);
  SGNN_LAUNCH(k_down2_parent, dim3(g), dim3(256), 0, s, nf, (const int32_t *)cvals,
                     (const int32_t *)rank_at, (const int32_t *)slot_of, parent);
  SGNN_LAUNCH(k_down2_fix_vals, dim3(g), dim3(256), 0, s, nf, (const int32_t *)slot_of,
                     (const int32_t *)parent, (const int32_t *)rank_at, cvals);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// ---------------------------------------------------------------------------
// stride-2 rulebooks of several successive levels WITHOUT a host round trip in between: every kernel takes its row
// count from device memory (the count the previous level / a mask compaction just wrote) and is launched for the
// host-known upper bound `cap`.  One read-back then returns all counts (15 -> 5 host syncs per training step).
// ---------------------------------------------------------------------------
__device__ __forceinline__ int64_t dev_n(const int64_t *n_dev, int64_t n_host) { return sgnn_dyn_n(n_host, n_dev); }

__global__ __launch_bounds__(256) void k_chain_init(unsigned long long *__restrict__ ckeys, int32_t *__restrict__ owner,
                                                   int64_t ccap, int32_t *__restrict__ rank_at, int64_t cap) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t top = ccap > cap ? ccap : cap;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < top; i += stride) {
    if (i < ccap) {
      ckeys[i] = ~0ull;
      owner[i] = 0x7FFFFFFF;
    }
    if (i < cap) rank_at[i] = -1;
  }
}

__device__ __forceinline__ void chain_insert_rows(const int4 *__restrict__ fine, const int64_t *n_dev, int64_t n_host,
                                                  unsigned long long *__restrict__ ckeys,
                                                  int32_t *__restrict__ owner, uint64_t mask,
                                                  int32_t *__restrict__ slot_of) {
  const int64_t nf = dev_n(n_dev, n_host);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < nf; i += stride) {
    const int4 c = fine[i];
    const uint64_t key = sgnn_pack_key(c.x >> 1, c.y >> 1, c.z >> 1, c.w);
    uint64_t slot = sgnn_hash64(key) & mask;
    while (true) {
      const unsigned long long prev = atomicCAS(&ckeys[slot], SGNN_EMPTY_KEY, (unsigned long long)key);
      if (prev == SGNN_EMPTY_KEY || prev == key) break;
      slot = (slot + 1) & mask;
    }
    atomicMin(&owner[slot], (int32_t)i);  // first-touch owner = smallest fine row
    slot_of[i] = (int32_t)slot;
  }
}

__global__ __launch_bounds__(256) void k_chain_insert(const int4 *__restrict__ fine, const int64_t *n_dev, int64_t n_host,
                                                     unsigned long long *__restrict__ ckeys,
                                                     int32_t *__restrict__ owner, uint64_t mask,
                                                     int32_t *__restrict__ slot_of) {
  chain_insert_rows(fine, n_dev, n_host, ckeys, owner, mask, slot_of);
}

__global__ __launch_bounds__(256) void k_chain_count(const int32_t *__restrict__ slot_of, const int32_t *__restrict__ owner,
                                                    const int64_t *n_dev, int64_t n_host,
                                                    int32_t *__restrict__ block_sums) {
  __shared__ int lds[4];
  const int64_t n = dev_n(n_dev, n_host);
  const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK;
  int cnt = 0;
#pragma unroll
  for (int it = 0; it < SCAN_ITEMS; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    const bool f = (i < n) && owner[slot_of[i]] == (int32_t)i;
    cnt += __popcll(__ballot(f));
  }
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}

// owners, in fine-row order, become the coarse rows: coordinates and rank
__global__ __launch_bounds__(256) void k_chain_emit(const int4 *__restrict__ fine, const int32_t *__restrict__ slot_of,
                                                   const int32_t *__restrict__ owner, const int64_t *n_dev,
                                                   int64_t n_host, const int32_t *block_offsets,
                                                   int4 *__restrict__ coarse, int32_t *__restrict__ rank_at, ScanInline si) {
  __shared__ int lds[8];
  const int64_t n = dev_n(n_dev, n_host);
  const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK;
  int running;
  if (si.nblk > 0) {
    int64_t total;
    scan_offsets_inline(block_offsets, si.nblk, blockIdx.x, lds, running, total);
    scan_publish(total, si.count, si.lim, blockIdx.x == 0 && threadIdx.x == 0);
  } else {
    running = block_offsets[blockIdx.x];
  }
#pragma unroll 1
  for (int it = 0; it < SCAN_ITEMS; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    const bool f = (i < n) && owner[slot_of[i]] == (int32_t)i;
    int total;
    const int r = sgnn_block_rank256(f, lds, total);
    if (f) {
      const int4 c = fine[i];
      coarse[running + r] = make_int4(c.x >> 1, c.y >> 1, c.z >> 1, c.w);
      rank_at[i] = running + r;
    }
    running += total;
  }
}

// parent row of every fine site; the coarse hash's values become coarse rows (written by the owners into cvals, a
// different array than the owner table the other threads are still reading)
__global__ __launch_bounds__(256) void k_chain_parent(const int64_t *n_dev, int64_t n_host,
                                                     const int32_t *__restrict__ owner,
                                                     const int32_t *__restrict__ rank_at,
                                                     const int32_t *__restrict__ slot_of,
                                                     int32_t *__restrict__ parent, int32_t *__restrict__ cvals) {
  const int64_t nf = dev_n(n_dev, n_host);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < nf; i += stride) {
    const int32_t sl = slot_of[i];
    const int32_t p = rank_at[owner[sl]];
    parent[i] = p;
    if (rank_at[i] >= 0) cvals[sl] = p;
  }
}

SGNN_EXPORT int64_t sgnn_down2_chain_ws_bytes(int64_t cap) {
  const int64_t nblk = (cap + SCAN_BLOCK - 1) / SCAN_BLOCK;
  // slot_of[cap] + rank_at[cap] + owner[ccap] + block sums
  return (2 * cap + sgnn_hash_capacity(cap)) * (int64_t)sizeof(int32_t) + (nblk + 1) * (int64_t)sizeof(int32_t) + 256;
}

SGNN_EXPORT int sgnn_down2_chain(const int32_t *fine_coords, int64_t n0, const int64_t *n0_dev, int64_t cap, int depth,
                                 void *const *ckeys, void *const *cvals, int64_t ccap, void *const *parent,
                                 void *const *coarse_coords, int64_t *counts_dev, const int64_t *level_caps,
                                 int32_t *status, void *ws, int64_t ws_bytes, sgnn_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  SGNN_CHECK_ARG(!level_caps || status);
  SGNN_CHECK_ARG(depth >= 1 && depth <= 8 && cap >= 0 && n0 >= 0 && n0 <= cap && counts_dev && ckeys && cvals && parent &&
                 coarse_coords);
  SGNN_CHECK_ARG(ccap >= 2 * cap && ccap >= 2 && (ccap & (ccap - 1)) == 0 && ccap < (1ll << 31));
  if (cap == 0) {
    SGNN_HIP_TRY(hipMemsetAsync(counts_dev, 0, depth * sizeof(int64_t), s));
    for (int l = 0; l < depth; ++l) SGNN_HIP_TRY(hipMemsetAsync(ckeys[l], 0xFF, (size_t)ccap * sizeof(uint64_t), s));
    return SGNN_OK;
  }
  SGNN_CHECK_ARG(fine_coords);
  if (!ws || ws_bytes < sgnn_down2_chain_ws_bytes(cap)) {
    sgnn_set_error("sgnn_down2_chain: workspace too small");
    return SGNN_ENOWS;
  }
  const int64_t nblk = (cap + SCAN_BLOCK - 1) / SCAN_BLOCK;
  int32_t *slot_of = (int32_t *)ws;
  int32_t *rank_at = slot_of + cap;
  int32_t *owner = rank_at + cap;
  int32_t *block_sums = owner + ccap;
  const int4 *fine = (const int4 *)fine_coords;
  const int64_t *n_dev = n0_dev;
  int64_t n_host = n0_dev ? cap : n0;
  const int g = sgnn_grid_for(cap, 256, 8192);
  for (int l = 0; l < depth; ++l) {
    SGNN_CHECK_ARG(ckeys[l] && cvals[l] && parent[l] && coarse_coords[l]);
    SGNN_LAUNCH(k_chain_init, dim3(sgnn_grid_for(ccap, 256, 4096)), dim3(256), 0, s,
                       (unsigned long long *)ckeys[l], owner, ccap, rank_at, cap);
    SGNN_LAUNCH(k_chain_insert, dim3(g), dim3(256), 0, s, fine, n_dev, n_host, (unsigned long long *)ckeys[l],
                       owner, (uint64_t)(ccap - 1), slot_of);
    SGNN_LAUNCH(k_chain_count, dim3((unsigned)nblk), dim3(256), 0, s, (const int32_t *)slot_of,
                       (const int32_t *)owner, n_dev, n_host, block_sums);
    const ScanLimit lim = level_caps ? ScanLimit{level_caps[l] < cap ? level_caps[l] : cap, nullptr, 0, status} : kNoLimit;
    const bool inl = scan_inline_ok(nblk);
    if (!inl) SGNN_LAUNCH(k_scan_block_sums, dim3(1), dim3(1024), 0, s, block_sums, nblk, counts_dev + l, lim);
    SGNN_LAUNCH(k_chain_emit, dim3((unsigned)nblk), dim3(256), 0, s, fine, (const int32_t *)slot_of,
                       (const int32_t *)owner, n_dev, n_host, (const int32_t *)block_sums, (int4 *)coarse_coords[l],
                       rank_at, inl ? ScanInline{nblk, counts_dev + l, lim} : kNoInline);
    SGNN_LAUNCH(k_chain_parent, dim3(g), dim3(256), 0, s, n_dev, n_host, (const int32_t *)owner,
                       (const int32_t *)rank_at, (const int32_t *)slot_of, (int32_t *)parent[l], (int32_t *)cvals[l]);
    fine = (const int4 *)coarse_coords[l];
    n_dev = counts_dev + l;
    n_host = cap;
  }
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// ---------------------------------------------------------------------------
// Capacity mode: the whole stride-2 pyramid INCLUDING its children / ptable tables in one submission with 5 launches
// per level + 1 per chain (the step-by-step form above needs 8 per level: init, insert, count, scan, emit, parent,
// children pre-fill, tables) — 3 per level + 2 since round 5: the write kernel sums the block counts itself (no scan
// launch, scan_offsets_inline) and the tables pass of level l shares a launch with the insertion of level l + 1:
//   k_chain_init_all   every level's hash / owner / rank scratch (each level has its own scratch slice)
//   per level: k_chain_insert, k_chain_count, k_scan_block_sums (clamps to the level capacity),
//              k_chain_emit2 (also pre-fills the children table up to the live coarse rows),
//              k_chain_parent_tables (parent[], coarse hash values, children[], ptable[])
// Same first-touch order, same tables as sgnn_down2_chain + sgnn_down2_tables (tests/test_gpu_capacity.py).
// ---------------------------------------------------------------------------
#define CHAIN_MAX_DEPTH 8
struct ChainInit {
  unsigned long long *ckeys[CHAIN_MAX_DEPTH];
  int32_t *owner[CHAIN_MAX_DEPTH];
  int32_t *rank_at[CHAIN_MAX_DEPTH];
  int64_t ccap, cap;
  int depth;
};

__global__ __launch_bounds__(256) void k_chain_init_all(ChainInit a) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t top = a.ccap > a.cap ? a.ccap : a.cap;
  for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < top * a.depth; g += stride) {
    const int l = (int)(g / top);
    const int64_t i = g - (int64_t)l * top;
    if (i < a.ccap) {
      a.ckeys[l][i] = ~0ull;
      a.owner[l][i] = 0x7FFFFFFF;
    }
    if (i < a.cap) a.rank_at[l][i] = -1;
  }
}

__global__ __launch_bounds__(256) void k_chain_emit2(const int4 *__restrict__ fine, const int32_t *__restrict__ slot_of,
                                                    const int32_t *__restrict__ owner, const int64_t *n_dev,
                                                    int64_t n_host, const int32_t *block_offsets,
                                                    int4 *__restrict__ coarse, int32_t *__restrict__ rank_at,
                                                    const int64_t *nc_dev, int32_t *__restrict__ children, int64_t ldc,
                                                    ScanInline si) {
  __shared__ int lds[8];
  const int64_t n = dev_n(n_dev, n_host);
  int running = 0;
  int64_t nc_live;
  if (si.nblk > 0) {     // raw block sums: offset, total and the clamped live count of the coarse level from this workgroup's own sum
    int64_t total;
    scan_offsets_inline(block_offsets, si.nblk, blockIdx.x, lds, running, total);
    nc_live = scan_publish(total, si.count, si.lim, blockIdx.x == 0 && threadIdx.x == 0);
  } else {
    nc_live = sgnn_dyn_n(ldc, nc_dev);      // final: the scan kernel ran before this one
  }
  // children[8][0 .. roundup256(live coarse rows)) := -1
  {
    const int64_t end = pad_end(nc_live, ldc);
    const int64_t total = end * 8, stride = (int64_t)gridDim.x * 256;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < total; g += stride) children[(g / end) * ldc + (g % end)] = -1;
  }
  const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK;
  if (base >= n) return;
  if (si.nblk <= 0) running = block_offsets[blockIdx.x];
#pragma unroll 1
  for (int it = 0; it < SCAN_ITEMS; ++it) {
    const int64_t i = base + it * 256 + threadIdx.x;
    const bool f = (i < n) && owner[slot_of[i]] == (int32_t)i;
    int total;
    const int r = sgnn_block_rank256(f, lds, total);
    if (f) {
      const int4 c = fine[i];
      coarse[running + r] = make_int4(c.x >> 1, c.y >> 1, c.z >> 1, c.w);
      rank_at[i] = running + r;
    }
    running += total;
  }
}

struct ChainTables {   // arguments of the parent / children / ptable pass of one level
  const int4 *fine;
  const int64_t *n_dev;
  int64_t n_host;
  const int32_t *owner, *rank_at, *slot_of;
  int32_t *parent, *cvals;
  const int64_t *nc_dev;
  int64_t nc_cap;
  int32_t *children;
  int64_t ldc;
  int32_t *ptable;
  int64_t ldf;
};

__device__ __forceinline__ void chain_parent_tables_rows(const int4 *__restrict__ fine, const int64_t *n_dev,
                                                         int64_t n_host, const int32_t *__restrict__ owner,
                                                         const int32_t *__restrict__ rank_at,
                                                         const int32_t *__restrict__ slot_of,
                                                         int32_t *__restrict__ parent, int32_t *__restrict__ cvals,
                                                         const int64_t *nc_dev, int64_t nc_cap,
                                                         int32_t *__restrict__ children, int64_t ldc,
                                                         int32_t *__restrict__ ptable, int64_t ldf) {
  const int64_t nf = dev_n(n_dev, n_host);
  const int64_t nc = sgnn_dyn_n(nc_cap, nc_dev);
  const int64_t iend = pad_end(nf, ldf);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; i < iend; i += stride) {
    if (i >= nf) {  // padding rows of the data-gradient table
#pragma unroll
      for (int k = 0; k < 8; ++k) ptable[(int64_t)k * ldf + i] = -1;
      continue;
    }
    const int32_t sl = slot_of[i];
    int32_t p = rank_at[owner[sl]];
    if (rank_at[i] >= 0) cvals[sl] = p;      // the coarse hash keeps the true row even past a clamped capacity
    if (p >= nc) p = -1;                      // only after a capacity overflow (the step is flagged and discarded)
    parent[i] = p;
    const int4 c = fine[i];
    const int off = ((c.x & 1) << 2) | ((c.y & 1) << 1) | (c.z & 1);
    if (p >= 0) children[(int64_t)off * ldc + p] = (int32_t)i;
#pragma unroll
    for (int k = 0; k < 8; ++k) ptable[(int64_t)k * ldf + i] = (k == off) ? p : -1;
  }
}

__global__ __launch_bounds__(256) void k_chain_parent_tables(ChainTables t) {
  chain_parent_tables_rows(t.fine, t.n_dev, t.n_host, t.owner, t.rank_at, t.slot_of, t.parent, t.cvals, t.nc_dev, t.nc_cap,
                           t.children, t.ldc, t.ptable, t.ldf);
}

// Round 5: the tables pass of level l and the hash insertion of level l + 1 both wait for level l's write kernel only
// (coarse coordinates + count, ranks) and touch disjoint arrays — one launch runs both loops (a launch less per inner level).
__global__ __launch_bounds__(256) void k_chain_tables_insert(ChainTables t, unsigned long long *__restrict__ ckeys_next,
                                                            int32_t *__restrict__ owner_next, uint64_t mask,
                                                            int32_t *__restrict__ slot_of_next, const int4 *coarse,
                                                            int64_t n_host_next) {
  chain_parent_tables_rows(t.fine, t.n_dev, t.n_host, t.owner, t.rank_at, t.slot_of, t.parent, t.cvals, t.nc_dev, t.nc_cap,
                           t.children, t.ldc, t.ptable, t.ldf);
  chain_insert_rows(coarse, t.nc_dev, n_host_next, ckeys_next, owner_next, mask, slot_of_next);
}


SGNN_EXPORT int64_t sgnn_down2_chain_tables_ws_bytes(int64_t cap, int depth) {
  const int64_t nblk = (cap + SCAN_BLOCK - 1) / SCAN_BLOCK;
  return depth * (2 * cap + sgnn_hash_capacity(cap)) * (int64_t)sizeof(int32_t) + (nblk + 1) * (int64_t)sizeof(int32_t) + 256;
}

// level l: fine rows = level_ld[l] stride tables; children[l] is (8 x ldc[l]) with ldc[l] = roundup256(level_caps[l]),
// ptable[l] is (8 x ldf[l]) with ldf[0] = roundup256(cap), ldf[l] = ldc[l-1].  All pointer arrays are HOST arrays.
SGNN_EXPORT int sgnn_down2_chain_tables(const int32_t *fine_coords, const int64_t *n0_dev, int64_t cap, int depth,
                                        void *const *ckeys, void *const *cvals, int64_t ccap, void *const *parent,
                                        void *const *coarse_coords, int64_t *counts_dev, const int64_t *level_caps,
                                        void *const *children, void *const *ptable, int32_t *status, void *ws,
                                        int64_t ws_bytes, sgnn_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  SGNN_CHECK_ARG(depth >= 1 && depth <= CHAIN_MAX_DEPTH && cap >= 1 && n0_dev && counts_dev && level_caps && status &&
                 ckeys && cvals && parent && coarse_coords && children && ptable && fine_coords);
  SGNN_CHECK_ARG(ccap >= 2 * cap && ccap >= 2 && (ccap & (ccap - 1)) == 0 && ccap < (1ll << 31));
  if (!ws || ws_bytes < sgnn_down2_chain_tables_ws_bytes(cap, depth)) {
    sgnn_set_error("sgnn_down2_chain_tables: workspace too small");
    return SGNN_ENOWS;
  }
  const int64_t nblk = (cap + SCAN_BLOCK - 1) / SCAN_BLOCK;
  int32_t *base = (int32_t *)ws;
  int32_t *slot_of[CHAIN_MAX_DEPTH], *rank_at[CHAIN_MAX_DEPTH], *owner[CHAIN_MAX_DEPTH];
  ChainInit ini{};
  for (int l = 0; l < depth; ++l) {
    SGNN_CHECK_ARG(ckeys[l] && cvals[l] && parent[l] && coarse_coords[l] && children[l] && ptable[l] && level_caps[l] >= 1);
    slot_of[l] = base;
    rank_at[l] = base + cap;
    owner[l] = base + 2 * cap;
    base += 2 * cap + ccap;
    ini.ckeys[l] = (unsigned long long *)ckeys[l];
    ini.owner[l] = owner[l];
    ini.rank_at[l] = rank_at[l];
  }
  int32_t *block_sums = base;
  ini.ccap = ccap;
  ini.cap = cap;
  ini.depth = depth;
  const int64_t top = ccap > cap ? ccap : cap;
  SGNN_LAUNCH(k_chain_init_all, dim3(sgnn_grid_for(top * depth, 256, 8192)), dim3(256), 0, s, ini);
  const int4 *fine = (const int4 *)fine_coords;
  const int64_t *n_dev = n0_dev;
  int64_t fine_cap = cap;
  const int g = sgnn_grid_for(cap, 256, 8192);
  const bool inl = scan_inline_ok(nblk), merged = g_tune.chain_merged != 0;
  for (int l = 0; l < depth; ++l) {
    const int64_t ccap_l = level_caps[l] < cap ? level_caps[l] : cap;
    const int64_t ldc = ((ccap_l + 255) / 256) * 256, ldf = ((fine_cap + 255) / 256) * 256;
    if (l == 0 || !merged)     // (merged: level l's insertion ran in the tables launch of level l - 1)
      SGNN_LAUNCH(k_chain_insert, dim3(g), dim3(256), 0, s, fine, n_dev, cap, (unsigned long long *)ckeys[l], owner[l],
                         (uint64_t)(ccap - 1), slot_of[l]);
    SGNN_LAUNCH(k_chain_count, dim3((unsigned)nblk), dim3(256), 0, s, (const int32_t *)slot_of[l],
                       (const int32_t *)owner[l], n_dev, cap, block_sums);
    const ScanLimit lim{ccap_l, nullptr, 0, status};
    if (!inl) SGNN_LAUNCH(k_scan_block_sums, dim3(1), dim3(1024), 0, s, block_sums, nblk, counts_dev + l, lim);
    SGNN_LAUNCH(k_chain_emit2, dim3((unsigned)nblk), dim3(256), 0, s, fine, (const int32_t *)slot_of[l],
                       (const int32_t *)owner[l], n_dev, cap, (const int32_t *)block_sums, (int4 *)coarse_coords[l],
                       rank_at[l], (const int64_t *)(counts_dev + l), (int32_t *)children[l], ldc,
                       inl ? ScanInline{nblk, counts_dev + l, lim} : kNoInline);
    const ChainTables t{fine, n_dev, cap, owner[l], rank_at[l], slot_of[l], (int32_t *)parent[l], (int32_t *)cvals[l],
                        counts_dev + l, ccap_l, (int32_t *)children[l], ldc, (int32_t *)ptable[l], ldf};
    if (merged && l + 1 < depth)
      SGNN_LAUNCH(k_chain_tables_insert, dim3(g), dim3(256), 0, s, t, (unsigned long long *)ckeys[l + 1], owner[l + 1],
                         (uint64_t)(ccap - 1), slot_of[l + 1], (const int4 *)coarse_coords[l], cap);
    else
      SGNN_LAUNCH(k_chain_parent_tables, dim3(g), dim3(256), 0, s, t);
    fine = (const int4 *)coarse_coords[l];
    n_dev = counts_dev + l;
    fine_cap = ccap_l;
  }
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

__global__ __launch_bounds__(256) void k_down2_tables(const int4 *__restrict__ fine,
                                                     const int32_t *__restrict__ parent, int64_t nf,
                                                     int32_t *__restrict__ children, int64_t ldc,
                                                     int32_t *__restrict__ ptable, int64_t ldf, int64_t nc,
                                                     const int64_t *nf_dev, const int64_t *nc_dev) {
  nf = sgnn_dyn_n(nf, nf_dev);
  nc = sgnn_dyn_n(nc, nc_dev);
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t iend = pad_end(nf, ldf);
  for (; i < iend; i += stride) {
    if (i >= nf) {  // padding
#pragma unroll
      for (int k = 0; k < 8; ++k) ptable[(int64_t)k * ldf + i] = -1;
      continue;
    }
    const int4 c = fine[i];
    const int off = ((c.x & 1) << 2) | ((c.y & 1) << 1) | (c.z & 1);
    int32_t p = parent[i];
    if (p >= nc) p = -1;       // only after a capacity overflow (the step is flagged and discarded): stay in bounds
    if (p >= 0) children[(int64_t)off * ldc + p] = (int32_t)i;
#pragma unroll
    for (int k = 0; k < 8; ++k) ptable[(int64_t)k * ldf + i] = (k == off) ? p : -1;
  }
}

SGNN_EXPORT int sgnn_down2_tables(const int32_t *fine_coords, const int32_t *parent, int64_t nf,
                                  int32_t *children, int64_t ldc, int64_t nc, int32_t *ptable, int64_t ldf,
                                  const int64_t *nf_dev, const int64_t *nc_dev, sgnn_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  SGNN_CHECK_ARG(nf >= 0 && nc >= 0 && ldc >= nc && ldf >= nf);
  if (nc > 0) {
    SGNN_CHECK_ARG(children);
    if (nc_dev)
      SGNN_LAUNCH(k_fill_rows_dyn, dim3(sgnn_grid_for(8 * nc, 256, 4096)), dim3(256), 0, s, children, 8, ldc, nc, nc_dev);
    else
      if (sgnn_fill32(children, 0xFFFFFFFFu, 8 * ldc, s) != SGNN_OK) return SGNN_EHIP;
  }
  if (nf == 0) return SGNN_OK;
  SGNN_CHECK_ARG(fine_coords && parent && ptable && children);
  SGNN_LAUNCH(k_down2_tables, dim3(sgnn_grid_for(ldf, 256, 8192)), dim3(256), 0, s,
                     (const int4 *)fine_coords, parent, nf, children, ldc, ptable, ldf, nc, nf_dev, nc_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// ---------------------------------------------------------------------------
// generative glue: 8-child expansion, dense coordinate table
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_expand8(const int4 *__restrict__ coords, int64_t n,
                                                int4 *__restrict__ out, const int64_t *n_dev) {
  n = sgnn_dyn_n(n, n_dev);
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per child
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; t < 8 * n; t += stride) {
    const int4 c = coords[t >> 3];
    const int j = (int)(t & 7);
    out[t] = make_int4(2 * c.x + (j >> 2), 2 * c.y + ((j >> 1) & 1), 2 * c.z + (j & 1), c.w);
  }
}

SGNN_EXPORT int sgnn_expand8_coords(const int32_t *coords, int64_t n, int32_t *out, const int64_t *n_dev,
                                    sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0);
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(coords && out);
  SGNN_LAUNCH(k_expand8, dim3(sgnn_grid_for(8 * n, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const int4 *)coords, n, (int4 *)out, n_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// the children AND their int64 (z, y, x, b) rows — the per-level `locs` output of the model (torch/model.py:207,243) — in one
// pass instead of an expansion launch followed by a conversion launch on the critical path between two stages
__global__ __launch_bounds__(256) void k_expand8_i64(const int4 *__restrict__ coords, int64_t n, int4 *__restrict__ out,
                                                    int64_t *__restrict__ locs, const int64_t *n_dev) {
  n = sgnn_dyn_n(n, n_dev);
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per child
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; t < 8 * n; t += stride) {
    const int4 c = coords[t >> 3];
    const int j = (int)(t & 7);
    const int4 o = make_int4(2 * c.x + (j >> 2), 2 * c.y + ((j >> 1) & 1), 2 * c.z + (j & 1), c.w);
    out[t] = o;
    reinterpret_cast<longlong2 *>(locs)[2 * t] = make_longlong2(o.x, o.y);
    reinterpret_cast<longlong2 *>(locs)[2 * t + 1] = make_longlong2(o.z, o.w);
  }
}

SGNN_EXPORT int sgnn_expand8_coords_i64(const int32_t *coords, int64_t n, int32_t *out, int64_t *locs, const int64_t *n_dev,
                                        sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0);
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(coords && out && locs);
  SGNN_LAUNCH(k_expand8_i64, dim3(sgnn_grid_for(8 * n, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const int4 *)coords, n, (int4 *)out, locs, n_dev);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

__global__ __launch_bounds__(256) void k_dense_coords(int batch, int d0, int d1, int d2,
                                                     int4 *__restrict__ out) {
  const int64_t vol = (int64_t)d0 * d1 * d2, total = vol * batch;
  int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (; t < total; t += stride) {
    const int b = (int)(t / vol);
    int64_t v = t - (int64_t)b * vol;
    const int x = (int)(v % d2);
    v /= d2;
    const int y = (int)(v % d1);
    const int z = (int)(v / d1);
    out[t] = make_int4(z, y, x, b);
  }
}

SGNN_EXPORT int sgnn_dense_coords(int batch, int d0, int d1, int d2, int32_t *out, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(batch >= 0 && d0 >= 0 && d1 >= 0 && d2 >= 0);
  const int64_t total = (int64_t)batch * d0 * d1 * d2;
  if (total == 0) return SGNN_OK;
  SGNN_CHECK_ARG(out);
  SGNN_LAUNCH(k_dense_coords, dim3(sgnn_grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream,
                     batch, d0, d1, d2, (int4 *)out);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}
