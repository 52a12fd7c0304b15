// Diagnostic: read-only HBM sweep variants (loads in flight per lane, workgroups per CU, non-temporal loads) -- the tuning run
// behind bw_read_kernel / gwi_hbm_bandwidth.   hipcc --offload-arch=gfx950 -O3 hbm_read.hip -o hbm_read && ./hbm_read
#include <hip/hip_runtime.h>

#include <cstdio>

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void rd(const double2* __restrict__ a, long long n2, double* out, int n_blocks) {
  double s = 0.0;
  const long long stride = (long long)n_blocks * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n2; i += UNROLL * stride) {
    double2 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (NT) {
        v[u].x = __builtin_nontemporal_load(&a[i + u * stride].x);
        v[u].y = __builtin_nontemporal_load(&a[i + u * stride].y);
      } else {
        v[u] = a[i + u * stride];
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y;
  }
  for (; i < n2; i += stride) s += a[i].x + a[i].y;
  if (s == 12345.678) out[0] = s;
}

template <int UNROLL, bool NT>
void run(const double2* a, long long n2, double* out, int wg_per_cu) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  float best = 1e30f;
  for (int it = 0; it < 8; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((rd<UNROLL, NT>), dim3(grid), dim3(256), 0, 0, a, n2, out, grid);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (it >= 2 && ms < best) best = ms;
  }
  printf("unroll %d nt %d wg/cu %2d: %.0f GB/s\n", UNROLL, (int)NT, wg_per_cu, 16.0 * n2 / (best * 1e-3) / 1e9);
}

int main() {
  const long long n2 = 1LL << 27;  // 2 GiB
  double2* a;
  double* out;
  hipMalloc(&a, sizeof(double2) * n2);
  hipMalloc(&out, 64);
  hipMemset(a, 0, sizeof(double2) * n2);
  for (int w : {4, 8, 16, 32}) {
    run<4, false>(a, n2, out, w);
    run<8, false>(a, n2, out, w);
    run<4, true>(a, n2, out, w);
    run<8, true>(a, n2, out, w);
  }
  return 0;
}
