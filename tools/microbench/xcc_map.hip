// Diagnostic: which XCD (HW_REG_XCC_ID) each workgroup of a dispatch runs on -- is it blockIdx.x mod 8, launch after launch, for
// any grid size, with other work in flight?   hipcc --offload-arch=gfx950 -O2 xcc_map.hip -o xcc_map && ./xcc_map
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void k(int* out, int spin) {
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(xcc & 0xf);
  for (int i = 0; i < spin; ++i) asm volatile("s_sleep 10");
}

int main() {
  int* d;
  hipMalloc(&d, sizeof(int) * 65536);
  hipStream_t s1, s2;
  hipStreamCreate(&s1);
  hipStreamCreate(&s2);
  int bad_total = 0;
  for (int grid : {8, 64, 395, 788, 1000, 2048, 7999}) {
    for (int rep = 0; rep < 50; ++rep) {
      // other work in flight on a second stream every other repetition
      if (rep & 1) hipLaunchKernelGGL(k, dim3(777), dim3(256), 0, s2, d + 32768, 20);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, s1, d, rep % 3 == 0 ? 5 : 0);
      hipStreamSynchronize(s1);
      std::vector<int> h(grid);
      hipMemcpy(h.data(), d, sizeof(int) * grid, hipMemcpyDeviceToHost);
      int bad = 0, shift = (h[0] - 0 + 8) % 8;
      for (int i = 0; i < grid; ++i) bad += (h[i] != (i + shift) % 8);
      if (bad || shift) printf("grid %d rep %d: start xcd %d, %d of %d workgroups off the round-robin pattern (first ids: %d %d %d %d %d %d %d %d %d)\n", grid, rep, shift, bad, grid, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8 % grid]);
      bad_total += bad + (shift != 0);
    }
    hipDeviceSynchronize();
  }
  printf("total deviations from xcd = blockIdx.x mod 8: %d\n", bad_total);
  return 0;
}
