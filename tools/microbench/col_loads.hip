// Diagnostic (VERDICT r3, weak point 10): does the scan's streaming phase gain from 16-byte column loads (two ADJACENT samples
// per lane, global_load_dwordx4) over the 8-byte loads it issues now (one sample per lane and column, global_load_dwordx2)?
//   hipcc --offload-arch=gfx950 -O3 col_loads.hip -o col_loads && ./col_loads
// The scan's access pattern without its arithmetic: C struct-of-array columns of fp64, workgroups of 256 lanes own contiguous
// tiles, every lane reads its sample(s) from each column, register-prefetched one trip ahead, then W dependent-free fp64 FMAs
// per sample stand in for the evaluation (W = 0: pure streaming).  Same bytes, same tiles, same resident workgroups per CU.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int kMaxCols = 9;
struct Args {
  const double* col[kMaxCols];
  double* out;
  long long n;
  int n_cols, tile, work;
};

template <int WIDE>  // WIDE = 1: 8 B per lane and column (sample i); 2: 16 B (samples 2 i, 2 i + 1)
__global__ __launch_bounds__(256) void stream(const Args a) {
  const long long t0 = (long long)blockIdx.x * a.tile;
  const long long t1 = t0 + a.tile < a.n ? t0 + a.tile : a.n;
  double acc = 0.0;
  double cur[kMaxCols][WIDE], nxt[kMaxCols][WIDE];
  auto load = [&](double (*dst)[WIDE], long long i) {
#pragma unroll
    for (int c = 0; c < kMaxCols; ++c) {
      if (c >= a.n_cols) break;
      if (WIDE == 1) {
        dst[c][0] = a.col[c][i];
      } else {
        const double2 v = *reinterpret_cast<const double2*>(a.col[c] + i);
        dst[c][0] = v.x;
        dst[c][WIDE - 1] = v.y;
      }
    }
  };
  const long long step = 256LL * WIDE;
  long long i = t0 + (long long)threadIdx.x * WIDE;
  if (i + WIDE <= t1) load(cur, i);
  for (; i + WIDE <= t1; i += step) {
    if (i + step + WIDE <= t1) load(nxt, i + step);
#pragma unroll
    for (int u = 0; u < WIDE; ++u) {
      double x = 0.0;
#pragma unroll
      for (int c = 0; c < kMaxCols; ++c)
        if (c < a.n_cols) x += cur[c][u];
      double y = x;
      for (int w = 0; w < a.work; ++w) y = fma(y, 0.999999, x);  // W fp64 FMAs per sample
      acc += y;
    }
#pragma unroll
    for (int c = 0; c < kMaxCols; ++c)
#pragma unroll
      for (int u = 0; u < WIDE; ++u) cur[c][u] = nxt[c][u];
  }
  if (acc == 12345.678) a.out[0] = acc;
}

int main() {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double* out;
  hipMalloc(&out, 64);
  struct Case {
    const char* name;
    long long n;
    int cols, tile;
  } cases[] = {{"config 2 (4 columns x 395 k samples, 12.6 MB)", 395000, 4, 512},
               {"config 3 (8 columns x 445 k samples, 28.5 MB)", 445000, 8, 455},
               {"config 5 (9 columns x 2.5 M samples, 180 MB)", 2500000, 9, 1280},
               {"config 5 x 10 (9 columns x 25 M samples, 1.8 GB)", 25000000, 9, 1280}};
  for (const Case& cs : cases) {
    Args a{};
    a.n = cs.n;
    a.n_cols = cs.cols;
    a.out = out;
    std::vector<double*> bufs;
    for (int c = 0; c < cs.cols; ++c) {
      double* p;
      hipMalloc(&p, sizeof(double) * (cs.n + 2));
      hipMemset(p, 0, sizeof(double) * (cs.n + 2));
      a.col[c] = p;
      bufs.push_back(p);
    }
    for (int work : {0, 64, 256}) {
      for (int wide : {1, 2}) {
        a.work = work;
        a.tile = cs.tile % 2 ? cs.tile + 1 : cs.tile;  // even tiles: the 16-byte loads stay aligned
        const int grid = (int)((cs.n + a.tile - 1) / a.tile);
        float best = 1e30f, sum = 0.0f;
        int cnt = 0;
        for (int it = 0; it < 40; ++it) {
          hipEventRecord(e0);
          if (wide == 1)
            hipLaunchKernelGGL(stream<1>, dim3(grid), dim3(256), 0, 0, a);
          else
            hipLaunchKernelGGL(stream<2>, dim3(grid), dim3(256), 0, 0, a);
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          float ms;
          hipEventElapsedTime(&ms, e0, e1);
          if (it >= 8) {
            best = ms < best ? ms : best;
            sum += ms;
            ++cnt;
          }
        }
        const double bytes = 8.0 * cs.cols * cs.n;
        printf("%-52s W = %3d FMAs  %2d B per lane and column: mean %8.2f us  best %8.2f us  (%6.0f GB/s at the mean)\n", cs.name, work, 8 * wide, 1e3 * sum / cnt, 1e3 * best,
               bytes / (sum / cnt * 1e-3) / 1e9);
      }
    }
    for (double* p : bufs) hipFree(p);
  }
  return 0;
}
