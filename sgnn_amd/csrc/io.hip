// On-disk formats of the data that feeds the path (.sdfs training chunks, .sdf scenes, .knw known masks) and
// their decode into the collated batch layout — SURVEY.md §8 row f2, the step before the path.
//
// Reference being replaced: torch/data_util.py:63-117 (load_train_file: struct.unpack of every scalar),
// :121-155 (load_scene, load_scene_known), torch/scene_dataloader.py:101-105 (|sdf| < truncation mask) and
// :13-36 (collate: batch index appended to the coordinates, samples concatenated / stacked).
//
// Split of work:
//   host   sgnn_io_layout        bounds-checked section table of one file image (no copies, no GPU)
//   device k_io_flag/k_io_emit   packed (x,y,z u32 | value f32) entries of a whole batch -> kept rows in
//                                [z,y,x,b] int64 + value/voxelsize, order preserved (stable compaction
//                                through sgnn_compact_mask between the two kernels)
//          k_io_scatter          sparse entries -> dense (B,1,d0,d1,d2) volumes (targets, hierarchy levels)
// so only the sparse entries and the u8 known volume cross PCIe; the dense fp32 target volumes (33 MB per
// 32-chunk batch, plus hierarchy) are produced in HBM.  All kernels are HBM-bound streaming passes.
#include <cstring>
#include "common.h"

#define IO_HEADER_BYTES 92   // 3 x u64 dims, f32 voxelsize, 16 x f32 world2grid

namespace {

struct Cursor {
  const unsigned char *p;
  int64_t n, at;
  bool ok;
  uint64_t u64() {
    uint64_t v = 0;
    if (at + 8 > n) {
      ok = false;
      return 0;
    }
    std::memcpy(&v, p + at, 8);
    at += 8;
    return v;
  }
  int64_t skip(int64_t bytes) {  // returns the offset of the skipped section
    const int64_t off = at;
    if (bytes < 0 || at + bytes > n) {
      ok = false;
      return -1;
    }
    at += bytes;
    return off;
  }
};

// one "count, count x (x,y,z) u32, count x f32" block (data_util.py:73-79)
bool sparse_block(Cursor &c, int64_t *count, int64_t *off_locs, int64_t *off_vals) {
  const uint64_t n = c.u64();
  if (!c.ok || n > (uint64_t)1 << 40) return false;
  *count = (int64_t)n;
  *off_locs = c.skip((int64_t)n * 12);
  *off_vals = c.skip((int64_t)n * 4);
  return c.ok;
}

}  // namespace

// out[24]: see include/sgnn_hip.h (SGNN_IO_* indices)
SGNN_EXPORT int sgnn_io_layout(const void *bytes, int64_t nbytes, int kind, int64_t *out) {
  SGNN_CHECK_ARG(bytes && out && nbytes >= 0 && kind >= 0 && kind <= 2);
  for (int i = 0; i < 24; ++i) out[i] = -1;
  Cursor c{(const unsigned char *)bytes, nbytes, 0, true};
  if (nbytes < IO_HEADER_BYTES) {
    sgnn_set_error("sgnn_io_layout: file shorter than its %d-byte header (%lld bytes)", IO_HEADER_BYTES,
                   (long long)nbytes);
    return SGNN_EINVAL;
  }
  const uint64_t dx = c.u64(), dy = c.u64(), dz = c.u64();
  if (dx == 0 || dy == 0 || dz == 0 || dx > 65535 || dy > 65535 || dz > 65535) {
    sgnn_set_error("sgnn_io_layout: implausible dimensions %llu x %llu x %llu", (unsigned long long)dx,
                   (unsigned long long)dy, (unsigned long long)dz);
    return SGNN_EINVAL;
  }
  out[0] = (int64_t)dx;
  out[1] = (int64_t)dy;
  out[2] = (int64_t)dz;
  uint32_t vs_bits = 0;
  std::memcpy(&vs_bits, c.p + 24, 4);
  out[3] = (int64_t)vs_bits;
  out[4] = 28;
  c.at = IO_HEADER_BYTES;
  const int64_t vol = (int64_t)(dx * dy * dz);
  bool ok = true;
  if (kind == 0) {  // .sdfs chunk: input, target, known (with count), 3 hierarchy levels
    ok = ok && sparse_block(c, &out[5], &out[6], &out[7]);
    ok = ok && sparse_block(c, &out[8], &out[9], &out[10]);
    if (ok) {
      const uint64_t nk = c.u64();
      if (c.ok && (int64_t)nk != vol) {  // data_util.py:93 assert
        sgnn_set_error("sgnn_io_layout: known-mask count %llu != %lld voxels", (unsigned long long)nk, (long long)vol);
        return SGNN_EINVAL;
      }
      out[11] = c.skip(vol);
      ok = c.ok;
    }
    for (int h = 0; h < 3 && ok; ++h) ok = sparse_block(c, &out[12 + 3 * h], &out[13 + 3 * h], &out[14 + 3 * h]);
  } else if (kind == 1) {  // .sdf scene: one sparse block
    ok = sparse_block(c, &out[5], &out[6], &out[7]);
  } else {  // .knw: raw u8 volume straight after the header (data_util.py:150)
    out[11] = c.skip(vol);
    ok = c.ok;
  }
  if (!ok) {
    sgnn_set_error("sgnn_io_layout: truncated or corrupt file (kind %d, %lld bytes, stopped at %lld)", kind,
                   (long long)nbytes, (long long)c.at);
    return SGNN_EINVAL;
  }
  out[21] = c.at;
  return SGNN_OK;
}

// sample of packed entry i: largest s with seg[s] <= i (nb <= a few hundred: binary search in registers)
__device__ __forceinline__ int io_sample_of(const int64_t *__restrict__ seg, int nb, int64_t i) {
  int lo = 0, hi = nb - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (seg[mid] <= i) lo = mid;
    else hi = mid - 1;
  }
  return lo;
}

// numpy divides float32 by float32 (data_util.py:79 `input_sdfs /= voxelsize`): one correctly rounded division
__device__ __forceinline__ float io_value(float raw, float voxelsize) { return __fdiv_rn(raw, voxelsize); }

// mask[i] = |value| < truncation (scene_dataloader.py:101) and z < max_z (scene_dataloader.py:83-86)
__global__ __launch_bounds__(256) void k_io_flag(const uint32_t *__restrict__ locs, const float *__restrict__ vals,
                                                const float *__restrict__ voxelsize,
                                                const int64_t *__restrict__ seg, int nb, int64_t n,
                                                float truncation, uint32_t max_z, uint8_t *__restrict__ mask) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int s = io_sample_of(seg, nb, i);
    const float v = io_value(vals[i], voxelsize[s]);
    mask[i] = (fabsf(v) < truncation) && (locs[3 * i + 2] < max_z);
  }
}

__global__ __launch_bounds__(256) void k_io_emit(const uint32_t *__restrict__ locs, const float *__restrict__ vals,
                                                const float *__restrict__ voxelsize,
                                                const int64_t *__restrict__ seg, int nb,
                                                const int32_t *__restrict__ sel, const int64_t *__restrict__ count,
                                                int64_t *__restrict__ out_locs, float *__restrict__ out_feats) {
  const int64_t m = *count;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += stride) {
    const int64_t i = sel[j];
    const int s = io_sample_of(seg, nb, i);
    const uint32_t x = locs[3 * i], y = locs[3 * i + 1], z = locs[3 * i + 2];
    longlong2 *o = reinterpret_cast<longlong2 *>(out_locs + 4 * j);      // file order is x,y,z; rows are z,y,x,b
    o[0] = make_longlong2((long long)z, (long long)y);
    o[1] = make_longlong2((long long)x, (long long)s);
    out_feats[j] = io_value(vals[i], voxelsize[s]);
  }
}

// dense[s][z][y][x] = value / voxelsize for entries inside (d0,d1,d2) and below max_z; the volume is pre-filled
__global__ __launch_bounds__(256) void k_io_scatter(const uint32_t *__restrict__ locs, const float *__restrict__ vals,
                                                   const float *__restrict__ voxelsize,
                                                   const int64_t *__restrict__ seg, int nb, int64_t n, int d0,
                                                   int d1, int d2, uint32_t max_z, float *__restrict__ dense) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const int s = io_sample_of(seg, nb, i);
    const uint32_t x = locs[3 * i], y = locs[3 * i + 1], z = locs[3 * i + 2];
    if (z >= (uint32_t)d0 || y >= (uint32_t)d1 || x >= (uint32_t)d2 || z >= max_z) continue;
    dense[(((int64_t)s * d0 + z) * d1 + y) * d2 + x] = io_value(vals[i], voxelsize[s]);
  }
}

SGNN_EXPORT int sgnn_io_flag_entries(const uint32_t *locs_xyz, const float *vals, const float *voxelsize,
                                     const int64_t *seg, int nb, int64_t n, float truncation, int64_t max_z,
                                     uint8_t *mask, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && nb >= 1 && max_z >= 0);
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(locs_xyz && vals && voxelsize && seg && mask);
  const uint32_t mz = max_z > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)max_z;
  SGNN_LAUNCH(k_io_flag, dim3(sgnn_grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, locs_xyz, vals,
                     voxelsize, seg, nb, n, truncation, mz, mask);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_io_emit_entries(const uint32_t *locs_xyz, const float *vals, const float *voxelsize,
                                     const int64_t *seg, int nb, const int32_t *sel, const int64_t *count,
                                     int64_t n_max, int64_t *out_locs, float *out_feats, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n_max >= 0 && nb >= 1);
  if (n_max == 0) return SGNN_OK;
  SGNN_CHECK_ARG(locs_xyz && vals && voxelsize && seg && sel && count && out_locs && out_feats);
  SGNN_LAUNCH(k_io_emit, dim3(sgnn_grid_for(n_max, 256, 4096)), dim3(256), 0, (hipStream_t)stream, locs_xyz,
                     vals, voxelsize, seg, nb, sel, count, out_locs, out_feats);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}

SGNN_EXPORT int sgnn_io_scatter_dense(const uint32_t *locs_xyz, const float *vals, const float *voxelsize,
                                      const int64_t *seg, int nb, int64_t n, int d0, int d1, int d2, int64_t max_z,
                                      float *dense, sgnn_stream_t stream) {
  SGNN_CHECK_ARG(n >= 0 && nb >= 1 && d0 >= 1 && d1 >= 1 && d2 >= 1 && max_z >= 0);
  if (n == 0) return SGNN_OK;
  SGNN_CHECK_ARG(locs_xyz && vals && voxelsize && seg && dense);
  const uint32_t mz = max_z > 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)max_z;
  SGNN_LAUNCH(k_io_scatter, dim3(sgnn_grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, locs_xyz, vals,
                     voxelsize, seg, nb, n, d0, d1, d2, mz, dense);
  SGNN_CHECK_LAUNCH();
  return SGNN_OK;
}
