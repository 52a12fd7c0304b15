// mfma_f64_layout.hip -- which lane / register holds which element of v_mfma_f64_16x16x4_f64 on gfx950, and what the
// instruction costs back to back.  Run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o mfma_f64_layout mfma_f64_layout.hip && ./mfma_f64_layout
// The batched spline-gradient GEMM of gwi_device.h (scan_mfma_kernel) relies on:
//   A (16 x 4):  lane l holds A[i = l % 16][k = l / 16]
//   B (4 x 16):  lane l holds B[k = l / 16][j = l % 16]
//   D (16 x 16): lane l, register r holds D[i = l / 16 + 4 * r][j = l % 16]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void probe(const double* A /*[16][4]*/, const double* B /*[4][16]*/, double* D /*[64][4]*/) {
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + l / 16];
  const double b = B[(l / 16) * 16 + l % 16];
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

__global__ void rate(double* out, int n) {
  const int l = threadIdx.x;
  v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const double a = 1.0 + l, b = 2.0 - l;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + l] = c0[0] + c1[1] + c2[2] + c3[3];
  if (l == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0);
}

int main() {
  std::vector<double> A(64), B(64), D(256);
  // D[i][j] = 100 (j + 1) + (i + 1): every element identifies itself
  for (int i = 0; i < 16; ++i)
    for (int k = 0; k < 4; ++k) A[i * 4 + k] = k == 0 ? 1.0 : (k == 1 ? 1.0 + i : 0.0);
  for (int k = 0; k < 4; ++k)
    for (int j = 0; j < 16; ++j) B[k * 16 + j] = k == 0 ? 100.0 * (j + 1) : (k == 1 ? 1.0 : 0.0);
  double *dA, *dB, *dD;
  hipMalloc(&dA, 64 * 8), hipMalloc(&dB, 64 * 8), hipMalloc(&dD, 256 * 8);
  hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice), hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int i = l / 16 + 4 * r, j = l % 16;
      double want = 0;
      for (int k = 0; k < 4; ++k) want += A[i * 4 + k] * B[k * 16 + j];
      if (D[l * 4 + r] != want) ++bad;
    }
  std::printf("layout A[i = l %% 16][k = l / 16], B[k = l / 16][j = l %% 16], D[i = l / 16 + 4 r][j = l %% 16]: %s (%d mismatches)\n", bad ? "NO" : "confirmed", bad);
  for (int l = 0; l < 64; l += 5)
    for (int r = 0; r < 4; ++r) {
      const int v = (int)D[l * 4 + r];
      std::printf("lane %2d reg %d holds D[i = %2d][j = %2d]\n", l, r, v % 100 - 1, v / 100 - 1);
    }
  // issue rate: 4 independent accumulators per wave, 1 and 4 waves per SIMD
  double* out;
  hipMalloc(&out, ((1 << 20) + 8) * 8);
  for (int waves : {1, 4}) {
    const int n = 20000;
    hipLaunchKernelGGL(rate, dim3(256 * waves), dim3(64 * 4), 0, 0, out, n);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate, dim3(256 * waves), dim3(64 * 4), 0, 0, out, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * waves * 4 * (double)n * 4 * 2048;  // workgroups x waves x iterations x 4 mfma x 16*16*4*2
    std::printf("%d wave(s)/SIMD: %.3f ms, %.1f TFLOP/s fp64 matrix, %.1f cycles per mfma per SIMD at 2.4 GHz\n", waves, ms, flops / ms / 1e9,
                ms * 1e-3 * 2.4e9 / ((double)n * 4 * waves));
  }
  return bad != 0;
}
