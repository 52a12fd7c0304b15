// Diagnostic: how long a wave waits for its kernel arguments, with and without kernarg preloading
// (-mllvm -amdgpu-kernarg-preload-count=N puts the first N dwords of explicit arguments into SGPRs at
// wave launch).  Build twice:  hipcc --offload-arch=gfx950 -O2 kernarg_preload.hip -o kp0
//                               hipcc ... -mllvm -amdgpu-kernarg-preload-count=8 ... -o kp8
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

__global__ void k(unsigned long long* stamps, const double* a, long long n, double* out) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  // first use of an argument: the pointer `a` (a scalar load unless preloaded), then one vector load
  const double v = a[threadIdx.x];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    stamps[blockIdx.x * 4 + 0] = t0;
    stamps[blockIdx.x * 4 + 1] = t1;
    stamps[blockIdx.x * 4 + 2] = t2;
  }
  if (v == 12345.0 && n == 7) out[0] = v;
}

int main() {
  const int blocks = 256;
  unsigned long long* stamps;
  double *a, *out;
  hipMalloc(&stamps, sizeof(unsigned long long) * blocks * 4);
  hipMalloc(&a, 4096);
  hipMalloc(&out, 64);
  hipMemset(a, 0, 4096);
  std::vector<unsigned long long> h(blocks * 4);
  std::vector<double> args_wait, load_wait;
  for (int rep = 0; rep < 50; ++rep) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, stamps, a, 3LL, out);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), stamps, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
    if (rep < 5) continue;
    for (int b = 0; b < blocks; ++b) {
      args_wait.push_back(10.0 * (double)(h[b * 4 + 1] - h[b * 4 + 0]));  // 100 MHz counter -> ns
      load_wait.push_back(10.0 * (double)(h[b * 4 + 2] - h[b * 4 + 1]));
    }
  }
  std::sort(args_wait.begin(), args_wait.end());
  std::sort(load_wait.begin(), load_wait.end());
  std::printf("wave entry -> arguments available: median %.0f ns (p90 %.0f); then first vector load: median %.0f ns\n", args_wait[args_wait.size() / 2],
              args_wait[args_wait.size() * 9 / 10], load_wait[load_wait.size() / 2]);
  return 0;
}
