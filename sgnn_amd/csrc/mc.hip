// Marching cubes over a dense TSDF volume + mesh clean-up, on the device (the step after
// the path in test_scene.py:100 / data_util.py:270-284).
//
// Reference being replaced (torch/marching_cubes/marching_cubes.cpp):
//   :66-92    get_voxel          validity: in bounds, != -inf, |d| < truncation
//   :107-131  trilerp            corner value = the 8 surrounding voxels, weights 0.5^3, summed in a fixed order
//   :133-157  vertexInterp       snaps within 1e-5, otherwise p1 + mu (p2 - p1)
//   :159-262  extract_isosurface_at_position   cube index, jump thresholds, Bourke's tables, triangle list
//   :359-456  merge_close_vertices (approx)    sequential greedy welding on a 1e-5 grid with a 27-cell lookup
//   :266-297, :298-321  remove_duplicate_faces, remove_degenerate_faces
//   :458-476  run_marching_cubes_internal      single-threaded z,y,x triple loop over the whole volume
//
// Results are IDENTICAL to the reference (vertex order, indices, bits): this file is compiled with
// -ffp-contract=off and uses correctly rounded divisions, so every float operation is the one g++ emits for the
// reference on x86-64; triangles are emitted in the reference's voxel order through a block scan; the greedy
// welding, which the reference defines by its sequential insertion order, is reproduced as the unique fixed point of
// "a grid cell is created iff no earlier-created cell exists among its 26 neighbours" (a greedy maximal independent
// set ordered by each cell's first vertex), solved by parallel sweeps.
//
// All passes are HBM/latency-bound streaming or hash-probe passes; nothing here is GEMM-shaped.
#include <mutex>
#include "common.h"
#include "mc_table.h"

#define MC_BLOCK 256
#define MC_NEG_INF (-__builtin_huge_valf())

__constant__ uint64_t c_mc_tri[256];

static int mc_upload_table() {
  static std::once_flag once;
  static hipError_t err = hipSuccess;
  std::call_once(once, [] { err = hipMemcpyToSymbol(HIP_SYMBOL(c_mc_tri), kMcTriangles, sizeof(kMcTriangles)); });
  return err == hipSuccess ? 0 : -1;
}

struct McGeom {
  int d0, d1, d2;   // z, y, x extents
  float iso, trunc, thresh;
};

__device__ __forceinline__ bool mc_valid(float d, float trunc) { return d != MC_NEG_INF && fabsf(d) < trunc; }

// Cube configuration of voxel (x,y,z) or -1.  dist[] in the reference's distArray order 000,100,010,001,110,011,101,111
// (digits = x,y,z side of the corner).
__device__ int mc_cube(const float *__restrict__ tsdf, const McGeom &g, int x, int y, int z, float (&dist)[8]) {
  if (x < 1 || y < 1 || z < 1 || x + 1 >= g.d2 || y + 1 >= g.d1 || z + 1 >= g.d0) return -1;  // a corner would leave the volume
  float v[3][3][3];   // [z][y][x] offsets -1..1; every one of the 27 feeds at least one corner
  bool ok = true;
#pragma unroll
  for (int dz = 0; dz < 3; ++dz)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const float d = tsdf[((int64_t)(z - 1 + dz) * g.d1 + (y - 1 + dy)) * g.d2 + (x - 1 + dx)];
        v[dz][dy][dx] = d;
        ok = ok && mc_valid(d, g.trunc);
      }
  if (!ok) return -1;
  // corner (sx,sy,sz): trilerp at pos + (s - 0.5): all weights are exactly 0.5, accumulation order of :118-125
  auto corner = [&](int sx, int sy, int sz) {
    float d = 0.0f;
    d += 0.125f * v[sz][sy][sx];
    d += 0.125f * v[sz][sy][sx + 1];
    d += 0.125f * v[sz][sy + 1][sx];
    d += 0.125f * v[sz + 1][sy][sx];
    d += 0.125f * v[sz][sy + 1][sx + 1];
    d += 0.125f * v[sz + 1][sy + 1][sx];
    d += 0.125f * v[sz + 1][sy][sx + 1];
    d += 0.125f * v[sz + 1][sy + 1][sx + 1];
    return d;
  };
  dist[0] = corner(0, 0, 0);
  dist[1] = corner(1, 0, 0);
  dist[2] = corner(0, 1, 0);
  dist[3] = corner(0, 0, 1);
  dist[4] = corner(1, 1, 0);
  dist[5] = corner(0, 1, 1);
  dist[6] = corner(1, 0, 1);
  dist[7] = corner(1, 1, 1);
  int ci = 0;
  if (dist[2] < g.iso) ci += 1;     // 010
  if (dist[4] < g.iso) ci += 2;     // 110
  if (dist[1] < g.iso) ci += 4;     // 100
  if (dist[0] < g.iso) ci += 8;     // 000
  if (dist[5] < g.iso) ci += 16;    // 011
  if (dist[7] < g.iso) ci += 32;    // 111
  if (dist[6] < g.iso) ci += 64;    // 101
  if (dist[3] < g.iso) ci += 128;   // 001
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      if (dist[k] * dist[l] < 0.0f) {
        if (fabsf(dist[k]) + fabsf(dist[l]) > g.thresh) return -1;
      } else {
        if (fabsf(dist[k] - dist[l]) > g.thresh) return -1;
      }
    }
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (fabsf(dist[k]) > g.thresh) return -1;
  return ci;
}

__device__ __forceinline__ int mc_triangles_of(uint64_t word, int &edge_mask) {
  int n = 0;
  edge_mask = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = (int)((word >> (4 * i)) & 0xF);
    if (e != 0xF) {
      edge_mask |= 1 << e;
      ++n;
    }
  }
  return n / 3;
}

// pass 1: triangles per voxel (u8) and per block of 256 consecutive voxels (z,y,x raster = the reference's loop order)
__global__ __launch_bounds__(MC_BLOCK) void k_mc_count(const float *__restrict__ tsdf, McGeom g, int64_t vol,
                                                      uint8_t *__restrict__ cnt, int32_t *__restrict__ blocksum) {
  __shared__ int sh[MC_BLOCK];
  const int64_t i = (int64_t)blockIdx.x * MC_BLOCK + threadIdx.x;
  int n = 0;
  if (i < vol) {
    const float own = tsdf[i];
    if (mc_valid(own, g.trunc)) {
      const int x = (int)(i % g.d2), y = (int)((i / g.d2) % g.d1), z = (int)(i / ((int64_t)g.d2 * g.d1));
      float dist[8];
      const int ci = mc_cube(tsdf, g, x, y, z, dist);
      if (ci >= 0) {
        int mask;
        n = mc_triangles_of(c_mc_tri[ci], mask);
        if (mask == 0 || mask == 255) n = 0;   // :215 (the 255 case is the reference author's addition)
      }
    }
    cnt[i] = (uint8_t)n;
  }
  sh[threadIdx.x] = n;
  __syncthreads();
  for (int d = MC_BLOCK / 2; d > 0; d >>= 1) {
    if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
    __syncthreads();
  }
  if (threadIdx.x == 0) blocksum[blockIdx.x] = sh[0];
}

// exclusive scan of the block sums (one workgroup; nblk <= a few hundred thousand)
__global__ __launch_bounds__(1024) void k_mc_scan(const int32_t *__restrict__ blocksum, int64_t nblk,
                                                 int64_t *__restrict__ blockoff, int64_t *__restrict__ total) {
  __shared__ int64_t sh[1024];
  __shared__ int64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < nblk; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const int64_t v = i < nblk ? blocksum[i] : 0;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
      const int64_t t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblk) blockoff[i] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += sh[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__device__ __forceinline__ void mc_interp(float iso, const float (&p1)[3], const float (&p2)[3], float d1, float d2,
                                          float (&out)[3]) {
  const float *r = nullptr;
  if (fabsf(iso - d1) < 0.00001f) r = p1;            // :140-142, in this order
  else if (fabsf(iso - d2) < 0.00001f) r = p2;
  else if (fabsf(d1 - d2) < 0.00001f) r = p1;
  if (r) {
    out[0] = r[0];
    out[1] = r[1];
    out[2] = r[2];
    return;
  }
  const float mu = __fdiv_rn(iso - d1, d2 - d1);
  out[0] = p1[0] + mu * (p2[0] - p1[0]);
  out[1] = p1[1] + mu * (p2[1] - p1[1]);
  out[2] = p1[2] + mu * (p2[2] - p1[2]);
}

// pass 2: emit the triangle soup in voxel order: verts (3*ntri, 3) x,y,z; vcols (3*ntri, 3)
__global__ __launch_bounds__(MC_BLOCK) void k_mc_emit(const float *__restrict__ tsdf, const uint8_t *__restrict__ colors,
                                                     McGeom g, int64_t vol, const uint8_t *__restrict__ cnt,
                                                     const int64_t *__restrict__ blockoff, float *__restrict__ verts,
                                                     uint8_t *__restrict__ vcols) {
  __shared__ int sh[MC_BLOCK];
  const int64_t i = (int64_t)blockIdx.x * MC_BLOCK + threadIdx.x;
  const int n = i < vol ? cnt[i] : 0;
  sh[threadIdx.x] = n;
  __syncthreads();
  for (int d = 1; d < MC_BLOCK; d <<= 1) {     // inclusive scan
    const int t = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  if (n == 0) return;
  int64_t tri = blockoff[blockIdx.x] + sh[threadIdx.x] - n;
  const int x = (int)(i % g.d2), y = (int)((i / g.d2) % g.d1), z = (int)(i / ((int64_t)g.d2 * g.d1));
  float dist[8];
  const int ci = mc_cube(tsdf, g, x, y, z, dist);
  // corner c (distArray order) sits at pos + (+-0.5): bit 0/1/2 of CB[c] = +x/+y/+z side
  const float fx = (float)x, fy = (float)y, fz = (float)z;
  const int CB[8] = {0, 1, 2, 4, 3, 6, 5, 7};
  auto corner_pos = [&](int c, float (&p)[3]) {
    const int bits = CB[c];
    p[0] = (bits & 1) ? fx + 0.5f : fx - 0.5f;
    p[1] = (bits & 2) ? fy + 0.5f : fy - 0.5f;
    p[2] = (bits & 4) ? fz + 0.5f : fz - 0.5f;
  };
  // cube edge e joins corners (a,b) in the reference's vertexInterp argument order (:226-237)
  const int EA[12] = {2, 4, 1, 0, 5, 7, 6, 3, 2, 4, 1, 0};
  const int EB[12] = {4, 1, 0, 2, 7, 6, 3, 5, 5, 7, 6, 3};
  const uint64_t word = c_mc_tri[ci];
  uint8_t c0 = 220, c1 = 220, c2 = 220;
  if (colors) {
    c0 = colors[3 * i];
    c1 = colors[3 * i + 1];
    c2 = colors[3 * i + 2];
  }
  for (int t = 0; t < n; ++t, ++tri) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int e = (int)((word >> (4 * (3 * t + k))) & 0xF);
      float out[3], pa[3], pb[3];
      corner_pos(EA[e], pa);
      corner_pos(EB[e], pb);
      mc_interp(g.iso, pa, pb, dist[EA[e]], dist[EB[e]], out);
      float *vp = verts + (3 * tri + k) * 3;
      vp[0] = out[0];
      vp[1] = out[1];
      vp[2] = out[2];
      uint8_t *cp = vcols + (3 * tri + k) * 3;
      cp[0] = c0;
      cp[1] = c1;
      cp[2] = c2;
    }
  }
}

static int64_t mc_blocks(int64_t vol) { return (vol + MC_BLOCK - 1) / MC_BLOCK; }

// workspace: cnt u8[vol] | blocksum i32[nblk] | blockoff i64[nblk]
SGNN_EXPORT int64_t sgnn_mc_ws_bytes(int d0, int d1, int d2) {
  const int64_t vol = (int64_t)d0 * d1 * d2, nblk = mc_blocks(vol);
  return ((vol + 255) & ~int64_t(255)) + ((nblk * 4 + 255) & ~int64_t(255)) + nblk * 8 + 256;
}

namespace {
struct McWs {
  uint8_t *cnt;
  int32_t *blocksum;
  int64_t *blockoff;
};
McWs mc_ws(void *ws, int64_t vol) {
  const int64_t nblk = mc_blocks(vol);
  char *p = (char *)ws;
  McWs w;
  w.cnt = (uint8_t *)p;
  p += (vol + 255) & ~int64_t(255);
  w.blocksum = (int32_t *)p;
  p += (nblk * 4 + 255) & ~int64_t(255);
  w.blockoff = (int64_t *)p;
  return w;
}
}  // namespace

SGNN_EXPORT int sgnn_mc_count(const float *tsdf, int d0, int d1, int d2, float isovalue, float truncation,
                              float thresh, void *ws, int64_t ws_bytes, int64_t *ntri, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(tsdf && ntri && d0 >= 1 && d1 >= 1 && d2 >= 1);
  if (!ws || ws_bytes < sgnn_mc_ws_bytes(d0, d1, d2)) {
    sgnn_set_error("sgnn_mc_count: workspace too small");
    return SGNN_ENOWS;
  }
  if (mc_upload_table() != 0) {
    sgnn_set_error("sgnn_mc_count: could not upload the triangle table");
    return SGNN_EHIP;
  }
  const int64_t vol = (int64_t)d0 * d1 * d2, nblk = mc_blocks(vol);
  SGNN_CHECK_ARG(nblk < (int64_t)1 << 31);
  const McGeom g{d0, d1, d2, isovalue, truncation, thresh};
  const McWs w = mc_ws(ws, vol);
  hipStream_t s = (hipStream_t)stream;
  SGNN_LAUNCH(k_mc_count, dim3((unsigned)nblk), dim3(MC_BLOCK), 0, s, tsdf, g, vol, w.cnt, w.blocksum);
  SGNN_LAUNCH(k_mc_scan, dim3(1), dim3(1024), 0, s, (const int32_t *)w.blocksum, nblk, w.blockoff, ntri);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_mc_emit(const float *tsdf, const uint8_t *colors, int d0, int d1, int d2, float isovalue,
                             float truncation, float thresh, void *ws, int64_t ws_bytes, float *verts,
                             uint8_t *vcols, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(tsdf && verts && vcols && d0 >= 1 && d1 >= 1 && d2 >= 1);
  if (!ws || ws_bytes < sgnn_mc_ws_bytes(d0, d1, d2)) {
    sgnn_set_error("sgnn_mc_emit: workspace too small");
    return SGNN_ENOWS;
  }
  const int64_t vol = (int64_t)d0 * d1 * d2, nblk = mc_blocks(vol);
  const McGeom g{d0, d1, d2, isovalue, truncation, thresh};
  const McWs w = mc_ws(ws, vol);
  SGNN_LAUNCH(k_mc_emit, dim3((unsigned)nblk), dim3(MC_BLOCK), 0, (hipStream_t)stream, tsdf, colors, g, vol,
                     (const uint8_t *)w.cnt, (const int64_t *)w.blockoff, verts, vcols);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// welding (merge_close_vertices with approx = true, :398-415)
// ---------------------------------------------------------------------------------------------------------
#define WELD_EMPTY (-1)
enum { W_UNDECIDED = 0, W_IN = 1, W_OUT = 2 };

struct Cell {
  int x, y, z;
};
__device__ __forceinline__ bool operator==(const Cell &a, const Cell &b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

__device__ __forceinline__ uint64_t weld_hash(const Cell &c) {
  uint64_t h = (uint64_t)(uint32_t)c.x * 0x9E3779B97F4A7C15ull;
  h ^= (uint64_t)(uint32_t)c.y * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
  h ^= (uint64_t)(uint32_t)c.z * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
  h ^= h >> 29;
  return h;
}

__device__ __forceinline__ int weld_sgn(float v) { return (0.0f < v) - (v < 0.0f); }

// cells[i] = (int)(v / thresh + 0.5f * sgn(v)) per axis (:402)
__global__ __launch_bounds__(256) void k_weld_cells(const float *__restrict__ verts, int64_t nv, float thresh,
                                                   Cell *__restrict__ cells) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nv) return;
  Cell c;
  const float vx = verts[3 * i], vy = verts[3 * i + 1], vz = verts[3 * i + 2];
  c.x = (int)(__fdiv_rn(vx, thresh) + 0.5f * (float)weld_sgn(vx));
  c.y = (int)(__fdiv_rn(vy, thresh) + 0.5f * (float)weld_sgn(vy));
  c.z = (int)(__fdiv_rn(vz, thresh) + 0.5f * (float)weld_sgn(vz));
  cells[i] = c;
}

// open addressing keyed by the cell, but a slot stores a VERTEX: rep[slot] = some vertex of the cell (claimed by
// CAS; its cell, written by the previous kernel, is the slot's key), first[slot] = smallest vertex index of the cell
__global__ __launch_bounds__(256) void k_weld_insert(const Cell *__restrict__ cells, int64_t nv,
                                                    int32_t *__restrict__ rep, int32_t *__restrict__ first,
                                                    int64_t cap) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nv) return;
  const Cell c = cells[i];
  int64_t h = (int64_t)(weld_hash(c) % (uint64_t)cap);
  for (;;) {
    int32_t cur = __hip_atomic_load(&rep[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == WELD_EMPTY) {
      const int32_t old = atomicCAS(&rep[h], WELD_EMPTY, (int32_t)i);
      cur = old == WELD_EMPTY ? (int32_t)i : old;
    }
    if (cells[cur] == c) {
      atomicMin(&first[h], (int32_t)i);
      return;
    }
    h = h + 1 == cap ? 0 : h + 1;
  }
}

__device__ __forceinline__ int64_t weld_find(const Cell &c, const Cell *__restrict__ cells,
                                             const int32_t *__restrict__ rep, int64_t cap) {
  int64_t h = (int64_t)(weld_hash(c) % (uint64_t)cap);
  for (;;) {
    const int32_t cur = rep[h];
    if (cur == WELD_EMPTY) return -1;
    if (cells[cur] == c) return h;
    h = h + 1 == cap ? 0 : h + 1;
  }
}

// one sweep of the greedy independent-set fixed point over the occupied slots
__global__ __launch_bounds__(256) void k_weld_sweep(const Cell *__restrict__ cells, const int32_t *__restrict__ rep,
                                                   const int32_t *__restrict__ first, uint8_t *state, int64_t cap,
                                                   unsigned long long *__restrict__ undecided) {
  const int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (h >= cap || rep[h] == WELD_EMPTY) return;
  if (__hip_atomic_load(&state[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != W_UNDECIDED) return;
  const Cell c = cells[rep[h]];
  const int32_t f = first[h];
  bool any_in = false, all_out = true;
  for (int dx = -1; dx <= 1 && !any_in; ++dx)
    for (int dy = -1; dy <= 1 && !any_in; ++dy)
      for (int dz = -1; dz <= 1; ++dz) {
        if (!dx && !dy && !dz) continue;
        const int64_t s = weld_find(Cell{c.x + dx, c.y + dy, c.z + dz}, cells, rep, cap);
        if (s < 0 || first[s] > f) continue;            // later cells cannot block this one
        const uint8_t st = __hip_atomic_load(&state[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (st == W_IN) {
          any_in = true;
          break;
        }
        if (st == W_UNDECIDED) all_out = false;
      }
  if (any_in) __hip_atomic_store(&state[h], (uint8_t)W_OUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (all_out) __hip_atomic_store(&state[h], (uint8_t)W_IN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else atomicAdd(undecided, 1ull);
}

// creator_of[i] = vertex whose grid cell vertex i resolves to (hasNearestNeighborApprox's i,j,k scan order, :342-356);
// is_creator[i] = 1 when that is i itself
__global__ __launch_bounds__(256) void k_weld_lookup(const Cell *__restrict__ cells, int64_t nv,
                                                    const int32_t *__restrict__ rep, const int32_t *__restrict__ first,
                                                    const uint8_t *__restrict__ state, int64_t cap,
                                                    int32_t *__restrict__ creator_of, uint8_t *__restrict__ is_creator) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nv) return;
  const Cell c = cells[i];
  int32_t found = -1;
  for (int dx = -1; dx <= 1 && found < 0; ++dx)
    for (int dy = -1; dy <= 1 && found < 0; ++dy)
      for (int dz = -1; dz <= 1; ++dz) {
        const int64_t s = weld_find(Cell{c.x + dx, c.y + dy, c.z + dz}, cells, rep, cap);
        if (s >= 0 && state[s] == W_IN && first[s] < (int32_t)i) {   // the cell existed when vertex i was processed
          found = first[s];
          break;
        }
      }
  const bool creator = found < 0;
  creator_of[i] = creator ? (int32_t)i : found;
  is_creator[i] = creator;
}

SGNN_EXPORT int64_t sgnn_weld_slots(int64_t n) { return n < 8 ? 16 : 2 * n + 1; }

SGNN_EXPORT int sgnn_weld_build(const float *verts, int64_t nv, float thresh, int32_t *cells, int32_t *rep,
                                int32_t *first, uint8_t *state, int64_t cap, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(nv >= 0 && nv < ((int64_t)1 << 31) && thresh > 0.f && cap >= sgnn_weld_slots(nv) && rep && first && state);
  hipStream_t s = (hipStream_t)stream;
  SGNN_HIP_TRY(hipMemsetAsync(rep, 0xFF, (size_t)cap * sizeof(int32_t), s));
  SGNN_HIP_TRY(hipMemsetAsync(first, 0x7F, (size_t)cap * sizeof(int32_t), s));
  SGNN_HIP_TRY(hipMemsetAsync(state, 0, (size_t)cap, s));
  if (nv == 0) return SGNN_OK;
  SGNN_CHECK_ARG(verts && cells);
  const dim3 grid((unsigned)((nv + 255) / 256));
  SGNN_LAUNCH(k_weld_cells, grid, dim3(256), 0, s, verts, nv, thresh, (Cell *)cells);
  SGNN_LAUNCH(k_weld_insert, grid, dim3(256), 0, s, (const Cell *)cells, nv, rep, first, cap);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_weld_sweep(const int32_t *cells, const int32_t *rep, const int32_t *first, uint8_t *state,
                                int64_t cap, int64_t *undecided, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(cells && rep && first && state && undecided && cap >= 1);
  hipStream_t s = (hipStream_t)stream;
  SGNN_HIP_TRY(hipMemsetAsync(undecided, 0, sizeof(int64_t), s));
  SGNN_LAUNCH(k_weld_sweep, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, s, (const Cell *)cells, rep, first,
                     state, cap, (unsigned long long *)undecided);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_weld_lookup(const int32_t *cells, int64_t nv, const int32_t *rep, const int32_t *first,
                                 const uint8_t *state, int64_t cap, int32_t *creator_of, uint8_t *is_creator,
                                 sgnn_stream_t stream) {
  SGNN_CHECK_ARG(nv >= 0 && cap >= 1);
  if (nv == 0) return SGNN_OK;
  SGNN_CHECK_ARG(cells && rep && first && state && creator_of && is_creator);
  SGNN_LAUNCH(k_weld_lookup, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const Cell *)cells, nv, rep, first, state, cap, creator_of, is_creator);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// ---------------------------------------------------------------------------------------------------------
// faces: remap through the weld, drop degenerate (:298-321) and duplicate (:266-297) triangles
// ---------------------------------------------------------------------------------------------------------
struct Tri {
  int a, b, c;
};
__device__ __forceinline__ Tri tri_sorted(const int32_t *__restrict__ f) {
  int a = f[0], b = f[1], c = f[2], t;
  if (a > b) { t = a; a = b; b = t; }
  if (b > c) { t = b; b = c; c = t; }
  if (a > b) { t = a; a = b; b = t; }
  return Tri{a, b, c};
}

// newid[sel[p]] = p : new index of every creator vertex (cnt order of :405-410 = soup order)
__global__ __launch_bounds__(256) void k_weld_number(const int32_t *__restrict__ sel, int64_t n, int32_t *__restrict__ newid) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p < n) newid[sel[p]] = (int32_t)p;
}

__global__ __launch_bounds__(256) void k_faces_remap(const int32_t *__restrict__ creator_of,
                                                    const int32_t *__restrict__ newid, int64_t ntri,
                                                    int32_t *__restrict__ faces) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e < 3 * ntri) faces[e] = newid[creator_of[e]];
}

__device__ __forceinline__ bool tri_degenerate(const int32_t *__restrict__ f) {
  return f[0] == f[1] || f[0] == f[2] || f[1] == f[2];
}
__device__ __forceinline__ bool tri_same(const Tri &a, const Tri &b) { return a.a == b.a && a.b == b.b && a.c == b.c; }

// set of unordered vertex triples, same slot-stores-an-index scheme; degenerate faces never enter it (:440 runs first)
__global__ __launch_bounds__(256) void k_faces_insert(const int32_t *__restrict__ faces, int64_t ntri,
                                                     int32_t *__restrict__ frep, int32_t *__restrict__ ffirst,
                                                     int64_t cap) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= ntri || tri_degenerate(faces + 3 * t)) return;
  const Tri key = tri_sorted(faces + 3 * t);
  int64_t h = (int64_t)(weld_hash(Cell{key.a, key.b, key.c}) % (uint64_t)cap);
  for (;;) {
    int32_t cur = __hip_atomic_load(&frep[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == WELD_EMPTY) {
      const int32_t old = atomicCAS(&frep[h], WELD_EMPTY, (int32_t)t);
      cur = old == WELD_EMPTY ? (int32_t)t : old;
    }
    if (tri_same(tri_sorted(faces + 3 * (int64_t)cur), key)) {
      atomicMin(&ffirst[h], (int32_t)t);
      return;
    }
    h = h + 1 == cap ? 0 : h + 1;
  }
}

// keep[t] = non-degenerate and the first face of its (unordered) vertex triple
__global__ __launch_bounds__(256) void k_faces_keep(const int32_t *__restrict__ faces, int64_t ntri,
                                                   const int32_t *__restrict__ frep, const int32_t *__restrict__ ffirst,
                                                   int64_t cap, uint8_t *__restrict__ keep) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= ntri) return;
  if (tri_degenerate(faces + 3 * t)) {
    keep[t] = 0;
    return;
  }
  const Tri key = tri_sorted(faces + 3 * t);
  int64_t h = (int64_t)(weld_hash(Cell{key.a, key.b, key.c}) % (uint64_t)cap);
  while (!tri_same(tri_sorted(faces + 3 * (int64_t)frep[h]), key)) h = h + 1 == cap ? 0 : h + 1;
  keep[t] = ffirst[h] == (int32_t)t;
}

SGNN_EXPORT int sgnn_weld_number(const int32_t *sel, int64_t n, int32_t *newid, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0);
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(sel && newid);
  SGNN_LAUNCH(k_weld_number, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sel, n, newid);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_mesh_faces(const int32_t *creator_of, const int32_t *newid, int64_t ntri, int32_t *faces,
                                int32_t *frep, int32_t *ffirst, int64_t cap, uint8_t *keep, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(ntri >= 0 && cap >= sgnn_weld_slots(ntri) && frep && ffirst);
  hipStream_t s = (hipStream_t)stream;
  SGNN_HIP_TRY(hipMemsetAsync(frep, 0xFF, (size_t)cap * sizeof(int32_t), s));
  SGNN_HIP_TRY(hipMemsetAsync(ffirst, 0x7F, (size_t)cap * sizeof(int32_t), s));
  if (ntri == 0) return SGNN_OK;
  SGNN_CHECK_ARG(creator_of && newid && faces && keep);
  const dim3 grid((unsigned)((ntri + 255) / 256));
  SGNN_LAUNCH(k_faces_remap, dim3((unsigned)((3 * ntri + 255) / 256)), dim3(256), 0, s, creator_of, newid, ntri,
                     faces);
  SGNN_LAUNCH(k_faces_insert, grid, dim3(256), 0, s, (const int32_t *)faces, ntri, frep, ffirst, cap);
  SGNN_LAUNCH(k_faces_keep, grid, dim3(256), 0, s, (const int32_t *)faces, ntri, (const int32_t *)frep,
                     (const int32_t *)ffirst, cap, keep);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

// out rows p < n: rows sel[p] of a (., 3) float / uint8 / int32 array
template <typename T>
__global__ __launch_bounds__(256) void k_take3(const T *__restrict__ src, const int32_t *__restrict__ sel, int64_t n,
                                              T *__restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= 3 * n) return;
  dst[e] = src[3 * (int64_t)sel[e / 3] + e % 3];
}

SGNN_EXPORT int sgnn_take_rows3(const void *src, int elem_bytes, const int32_t *sel, int64_t n, void *dst,
                                sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && (elem_bytes == 1 || elem_bytes == 4));
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(src && sel && dst);
  const dim3 grid((unsigned)((3 * n + 255) / 256));
  if (elem_bytes == 4)
    SGNN_LAUNCH((k_take3<uint32_t>), grid, dim3(256), 0, (hipStream_t)stream, (const uint32_t *)src, sel, n,
                       (uint32_t *)dst);
  else
    SGNN_LAUNCH((k_take3<uint8_t>), grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t *)src, sel, n,
                       (uint8_t *)dst);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}
