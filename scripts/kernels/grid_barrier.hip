// Measurement kernel (not part of libsgnn_hip.so): what does a grid-wide barrier cost on MI355X, next to the 1.7 us a
// dependent kernel node costs inside a replayed graph (scripts/bench_graph_node.py)?  VERDICT r5 item 2 proposes one
// cooperative kernel per residual block on small levels (conv -> statistics -> barrier -> apply -> conv ...): its phases are
// separated by exactly this primitive.  Every phase each workgroup reads 1 KiB another workgroup wrote in the phase before
// (agent-scope release / acquire: the eight XCDs' L2s are not coherent with each other), adds, writes, and meets the others
// at an atomic counter.  Spins are bounded: a barrier that is never met sets *err and the kernel ends.
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target, int *err) {
  __syncthreads();
  __shared__ int ok;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    ok = 1;
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) {
        *err = 1;
        ok = 0;
        break;
      }
    }
  }
  __syncthreads();
  return ok != 0;
}

__global__ __launch_bounds__(256) void k_phases(float *buf, unsigned *counter, int phases, int *err) {
  const unsigned G = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
  float acc = 0.f;
  for (int p = 0; p < phases; ++p) {
    const float *src = buf + (size_t)((p + 1) & 1) * G * 256;
    float *dst = buf + (size_t)(p & 1) * G * 256;
    acc += src[((wg + 1) % G) * 256 + tid];       // a row another workgroup wrote one phase ago
    dst[wg * 256 + tid] = acc + 1.f;
    if (!grid_barrier(counter, (unsigned)(p + 1) * G, err)) return;
  }
  if (acc == 12345.678f) buf[0] = acc;
}

extern "C" __attribute__((visibility("default"))) int grid_barrier_run(float *buf, unsigned *counter, int blocks, int phases,
                                                                       int *err, void *stream) {
  hipLaunchKernelGGL(k_phases, dim3(blocks), dim3(256), 0, (hipStream_t)stream, buf, counter, phases, err);
  return (int)hipGetLastError();
}
