// Diagnostic: would config 2's step be shorter with ONE launch -- every scan workgroup publishing its
// 9-double tile record straight to pinned host memory as two self-stamped 64-byte lines, the HOST merging the ten tiles of an
// event (power-of-two normalised records: ldexp, no exp) -- than with the scan -> combine pair it runs now (788 tile records
// in device memory -> 76 group rows of two lines each to the host)?
//   hipcc --offload-arch=gfx950 -O2 host_merge.hip -o host_merge && ./host_merge [producers] [work_us]
// Producers: P workgroups busy for `work_us`, then a 9-value record (binade exponent, S1, S2, six gradient numerators).
// Host clock: from just before the (first) launch to the merged per-group results (sum of log S1, sum of S2 / S1^2, gradient).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kVals = 9, kPerGroup = 10, kLines = 2;  // 9 values = 2 lines of 7 + stamp

struct Args {
  double* rec;        // device: [P][16]
  double* host_rows;  // pinned: mode 0: [G][kLines][8]; mode 1: [P][kLines][8]
  unsigned long long seq;
  int P, G;
  long long work_ticks;
};

__device__ inline void busy(long long ticks) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}
__device__ inline double value(int b, int v, unsigned long long seq) {
  if (v == 0) return (double)((b * 7 + (int)(seq % 5)) % 11 - 5);  // binade exponent of the tile
  return 1.0 + 0.37 * v + 1e-3 * (b % 97);                        // S1 in [1, 2)-ish, others of that order
}
// lines of seven values + the sequence number: every line validates itself
__device__ inline void publish_lines(double* dst, const double* vals, int n_vals, unsigned long long seq, int tid) {
  if (tid < 8 * kLines) {
    const int line = tid >> 3, j = tid & 7, i = line * 7 + j;
    const unsigned long long bits = j == 7 ? seq : (i < n_vals ? (unsigned long long)__double_as_longlong(vals[i]) : 0ull);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst) + tid, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(256) void producer_kernel(const Args a) {  // record to device memory (the scan of today)
  busy(a.work_ticks);
  if (threadIdx.x < kVals) a.rec[(long long)blockIdx.x * 16 + threadIdx.x] = value(blockIdx.x, threadIdx.x, a.seq);
}
__global__ __launch_bounds__(64) void combine_kernel(const Args a) {  // one wave per group: merge ten records, publish two lines
  const int g = blockIdx.x, lane = threadIdx.x;
  __shared__ double row[16];
  const double* r = a.rec + (long long)g * kPerGroup * 16;
  const bool has = lane < kPerGroup;
  const double m = has ? __hip_atomic_load(r + lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : -1e300;
  double mx = m;
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
  double mts[kPerGroup];  // every lane takes part in the shuffles: lane 9 holds a tile's exponent too (inside the branch below it is inactive)
  for (int t = 0; t < kPerGroup; ++t) mts[t] = __shfl(m, t);
  if (lane < kVals) {
    double s = 0.0;
    for (int t = 0; t < kPerGroup; ++t) {
      const double mt = mts[t];
      const double v = __hip_atomic_load(r + t * 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      s += lane == 0 ? 0.0 : ldexp(v, (int)(mt - mx) * (lane == 2 ? 2 : 1));
    }
    row[lane] = lane == 0 ? mx : s;
  }
  __syncthreads();
  publish_lines(a.host_rows + (long long)g * kLines * 8, row, kVals, a.seq, lane);
}
__global__ __launch_bounds__(256) void producer_host_kernel(const Args a) {  // record straight to the host
  busy(a.work_ticks);
  __shared__ double row[16];
  if (threadIdx.x < kVals) row[threadIdx.x] = value(blockIdx.x, threadIdx.x, a.seq);
  __syncthreads();
  publish_lines(a.host_rows + (long long)blockIdx.x * kLines * 8, row, kVals, a.seq, threadIdx.x);
}

struct Merged {
  double sum_log_s1, sum_var, g[6];
};
static inline void unpack(const double* lines, double* v) {  // two lines -> 9 values
  for (int i = 0; i < 7; ++i) v[i] = lines[i];
  v[7] = lines[8];
  v[8] = lines[9];
}

int main(int argc, char** argv) {
  const int P = argc > 1 ? std::atoi(argv[1]) : 780, G = P / kPerGroup;
  const double work_us = argc > 2 ? std::atof(argv[2]) : 5.0;
  Args a{};
  a.P = P;
  a.G = G;
  a.work_ticks = (long long)(work_us * 100.0);
  hipMalloc(&a.rec, sizeof(double) * P * 16);
  double* rows = nullptr;
  hipHostMalloc((void**)&rows, sizeof(double) * (size_t)P * kLines * 8, hipHostMallocMapped);
  a.host_rows = rows;
  volatile unsigned long long* stamps = reinterpret_cast<volatile unsigned long long*>(rows);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int n = 3000;
  unsigned long long seq = 0;
  double check[2] = {0, 0};
  for (int mode = 0; mode < 2; ++mode) {
    double total = 0.0, t_poll = 0.0;
    for (int it = 0; it < n + 200; ++it) {
      a.seq = ++seq;
      const auto t0 = std::chrono::steady_clock::now();
      Merged mg{};
      if (mode == 0) {
        hipLaunchKernelGGL(producer_kernel, dim3(P), dim3(256), 0, s, a);
        hipLaunchKernelGGL(combine_kernel, dim3(G), dim3(64), 0, s, a);
        for (int l = 0; l < G * kLines; ++l)
          while (stamps[l * 8 + 7] != seq) {
          }
        const auto tp = std::chrono::steady_clock::now();
        t_poll += std::chrono::duration<double>(tp - t0).count();
        for (int g = 0; g < G; ++g) {
          double v[9];
          unpack(rows + (size_t)g * kLines * 8, v);
          mg.sum_log_s1 += std::log(v[1]) + v[0] * 0.6931471805599453;
          mg.sum_var += v[2] / (v[1] * v[1]);
          for (int p = 0; p < 6; ++p) mg.g[p] += v[3 + p] / v[1];
        }
      } else {
        hipLaunchKernelGGL(producer_host_kernel, dim3(P), dim3(256), 0, s, a);
        for (int l = 0; l < P * kLines; ++l)
          while (stamps[l * 8 + 7] != seq) {
          }
        const auto tp = std::chrono::steady_clock::now();
        t_poll += std::chrono::duration<double>(tp - t0).count();
        double prod = 1.0;  // sum of log S1 as the log of a product of [1, 2) mantissas: one log per evaluation, not per group
        int e_sum = 0;
        for (int g = 0; g < G; ++g) {
          double v[kPerGroup][9], mx = -1e300;
          for (int t = 0; t < kPerGroup; ++t) {
            unpack(rows + ((size_t)g * kPerGroup + t) * kLines * 8, v[t]);
            mx = std::fmax(mx, v[t][0]);
          }
          double acc[9] = {0};
          for (int t = 0; t < kPerGroup; ++t) {
            const int d = (int)(v[t][0] - mx);
            acc[1] += std::ldexp(v[t][1], d);
            acc[2] += std::ldexp(v[t][2], 2 * d);
            for (int p = 3; p < 9; ++p) acc[p] += std::ldexp(v[t][p], d);
          }
          int e;
          prod *= std::frexp(acc[1], &e);
          e_sum += e + (int)mx;
          mg.sum_var += acc[2] / (acc[1] * acc[1]);
          const double inv = 1.0 / acc[1];
          for (int p = 0; p < 6; ++p) mg.g[p] += acc[3 + p] * inv;
        }
        mg.sum_log_s1 = std::log(prod) + e_sum * 0.6931471805599453;
      }
      const auto t1 = std::chrono::steady_clock::now();
      if (it >= 200) total += std::chrono::duration<double>(t1 - t0).count();
      if (it < 200) t_poll = 0.0;
      check[mode] = mg.sum_log_s1 + mg.sum_var + mg.g[0] + mg.g[5];
      hipStreamSynchronize(s);
    }
    std::printf("%s: %d producers busy %.1f us, %d groups: launch -> merged result on the host %.2f us (all lines seen after %.2f us; %d lines of 64 B over PCIe)\n",
                mode == 0 ? "scan + combine launches, host sums group rows" : "one launch, host merges tile records        ", P, work_us, G, 1e6 * total / n, 1e6 * t_poll / n,
                (mode == 0 ? G : P) * kLines);
  }
  std::printf("checksums %.6f %.6f\n", check[0], check[1]);
  return 0;
}
