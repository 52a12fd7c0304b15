// Device side of tools/microbench/aql_dispatch.cpp: a kernel shaped like the engine's launches as far as the host can
// tell -- a ~3 KB argument block, a grid of a few hundred workgroups, and a completion stamp stored write-through to
// pinned host memory by the last workgroup to finish counting.   hipcc --offload-arch=gfx950 --genco -O2 ... -o aql_stamp_kernel.hsaco
#include <hip/hip_runtime.h>

struct StampArgs {
  unsigned long long* host_slot;
  unsigned int* counter;  // device memory, zero between launches (reset by the last workgroup)
  unsigned long long seq;
  double payload[400];
};

extern "C" __global__ void stamp_kernel(const StampArgs a) {
  __shared__ double s;
  if (threadIdx.x == 0) s = a.payload[blockIdx.x % 400];
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int done = atomicAdd(a.counter, 1u) + 1u;
    if (done == gridDim.x) {
      *a.counter = 0;
      __hip_atomic_store(a.host_slot, a.seq + (s == 12345.5 ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
