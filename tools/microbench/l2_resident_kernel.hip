// Device side of l2_resident.cpp: the scan's access pattern (struct-of-array fp64 columns, contiguous tiles, 256 lanes, one-trip-ahead
// register prefetch) with W stand-in FMAs per sample; no gridDim / blockDim use, so the kernel takes no implicit arguments.
//   hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 -mllvm -amdgpu-kernarg-preload-count=16 l2_resident_kernel.hip -o l2_resident_kernel.hsaco
// (the preload option only affects stream_kernel_preload: by-value structs are not preloaded)
#include <hip/hip_runtime.h>

constexpr int kMaxCols = 9;
struct Args {
  const double* col[kMaxCols];
  double* out;
  long long n;
  int n_cols, tile, work, pad;
};

extern "C" __global__ __launch_bounds__(256) void stream_kernel(const Args a) {
  const long long t0 = (long long)blockIdx.x * a.tile;
  const long long t1 = t0 + a.tile < a.n ? t0 + a.tile : a.n;
  double acc = 0.0;
  double cur[kMaxCols], nxt[kMaxCols];
  auto load = [&](double* dst, long long i) {
#pragma unroll
    for (int c = 0; c < kMaxCols; ++c) {
      if (c >= a.n_cols) break;
      dst[c] = a.col[c][i];
    }
  };
  long long i = t0 + threadIdx.x;
  if (i < t1) load(cur, i);
  for (; i < t1; i += 256) {
    if (i + 256 < t1) load(nxt, i + 256);
    double x = 0.0;
#pragma unroll
    for (int c = 0; c < kMaxCols; ++c)
      if (c < a.n_cols) x += cur[c];
    double y = x;
    for (int w = 0; w < a.work; ++w) y = fma(y, 0.999999, x);
    acc += y;
#pragma unroll
    for (int c = 0; c < kMaxCols; ++c) cur[c] = nxt[c];
  }
  if (acc == 12345.678) a.out[0] = acc;
}

// The same kernel with its first arguments as SCALARS, which the command processor can place in scalar registers before the wave
// starts (kernel-argument preload, 16 dwords: four column pointers, the tile size and the sample count here): the column loads
// need no scalar round trip to the argument block first.  Columns beyond the fourth come from the struct as before.
extern "C" __global__ __launch_bounds__(256) void stream_kernel_preload(const double* c0, const double* c1, const double* c2, const double* c3, int tile, int n_cols, long long n, int work, int pad,
                                                                        const Args a) {
  const long long t0 = (long long)blockIdx.x * tile;
  const long long t1 = t0 + tile < n ? t0 + tile : n;
  double acc = 0.0;
  double cur[kMaxCols], nxt[kMaxCols];
  auto load = [&](double* dst, long long i) {
    dst[0] = c0[i];
    dst[1] = c1[i];
    dst[2] = c2[i];
    dst[3] = c3[i];
#pragma unroll
    for (int c = 4; c < kMaxCols; ++c) {
      if (c >= n_cols) break;
      dst[c] = a.col[c][i];
    }
  };
  long long i = t0 + threadIdx.x;
  if (i < t1) load(cur, i);
  for (; i < t1; i += 256) {
    if (i + 256 < t1) load(nxt, i + 256);
    double x = 0.0;
#pragma unroll
    for (int c = 0; c < kMaxCols; ++c)
      if (c < n_cols) x += cur[c];
    double y = x;
    for (int w = 0; w < work; ++w) y = fma(y, 0.999999, x);
    acc += y;
#pragma unroll
    for (int c = 0; c < kMaxCols; ++c) cur[c] = nxt[c];
  }
  if (acc == 12345.678) a.out[0] = acc;
}

// The same stand-in with workgroups of 1 024 lanes (16 waves: four per SIMD of one CU) -- the shape a single-launch config 2
// would need (EXPERIMENTS section 15: ~250 fat workgroups instead of 788) -- to see what the kernel side of that design costs.
extern "C" __global__ __launch_bounds__(1024) void stream_kernel_fat(const double* c0, const double* c1, const double* c2, const double* c3, int tile, int n_cols, long long n, int work, int pad,
                                                                     const Args a) {
  const long long t0 = (long long)blockIdx.x * tile;
  const long long t1 = t0 + tile < n ? t0 + tile : n;
  double acc = 0.0;
  for (long long i = t0 + threadIdx.x; i < t1; i += 1024) {
    const double x = c0[i] + c1[i] + c2[i] + c3[i];
    double y = x;
    for (int w = 0; w < work; ++w) y = fma(y, 0.999999, x);
    acc += y;
  }
  if (acc == 12345.678) a.out[0] = acc;
}
