// Diagnostic: is a reduction that waits INSIDE the producing launch (consumer workgroups at the end of the grid, spinning on
// arrival counters) visible on the host sooner than the same reduction as a second, dependent launch?  The question behind the
// scan -> combine pair of a small catalog (config 2: 788 tile records -> 76 group rows; the second launch and its dependency
// cost ~4 us of a 15 us step; round 2 tried the LAST-ARRIVING producer as the reducer and lost 1.3 us).
//   hipcc --offload-arch=gfx950 -O2 fused_consumer.hip -o fused_consumer && ./fused_consumer
// Producers: P workgroups busy for `work_us`, then a 16-double record with agent-scope stores, a release fence and one agent-scope
// atomic add on their group's counter.  Consumers: one workgroup per group of 10 producers; sum the group's records, write one
// self-stamped 64-byte line to pinned host memory.  Host clock: from just before the (first) launch to the last line's stamp.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kRec = 16, kPerGroup = 10;

struct Args {
  double* rec;                  // [P][kRec]
  unsigned* counter;            // [G]
  double* host_rows;            // pinned: [G][8], slot 7 = stamp
  unsigned long long seq;
  int P, G;
  long long work_ticks;         // s_memrealtime ticks (100 MHz) every producer stays busy
};

__device__ inline void busy(long long ticks) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}

__device__ inline void produce(const Args& a, int b) {
  busy(a.work_ticks);
  if (threadIdx.x < kRec) __hip_atomic_store(a.rec + (long long)b * kRec + threadIdx.x, (double)(b + 1) + 0.001 * threadIdx.x + (double)a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline void consume(const Args& a, int g, bool fresh) {
  // 10 records x 16 values: thread t < 16 sums value t over the group's records
  double s = 0.0;
  if (threadIdx.x < kRec)
    for (int j = 0; j < kPerGroup; ++j) {
      const double* p = a.rec + ((long long)g * kPerGroup + j) * kRec + threadIdx.x;
      s += fresh ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
    }
  __shared__ double sh[kRec];
  if (threadIdx.x < kRec) sh[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < 8) {
    const double v = threadIdx.x < 7 ? sh[threadIdx.x] + sh[threadIdx.x + 8] : __longlong_as_double((long long)a.seq);
    __hip_atomic_store(a.host_rows + (long long)g * 8 + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// one launch: producers first, consumers behind them in the grid
__global__ __launch_bounds__(256) void fused_kernel(const Args a) {
  const int b = blockIdx.x;
  if (b < a.P) {
    produce(a, b);
    __threadfence();  // agent scope: the record before the arrival
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(a.counter + b / kPerGroup, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const int g = b - a.P;
  if (threadIdx.x == 0) {
    const unsigned want = (unsigned)(a.seq * kPerGroup);  // counters are never reset: the n-th evaluation waits for n x 10
    while (__hip_atomic_load(a.counter + g, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  consume(a, g, true);
}
__global__ __launch_bounds__(256) void producer_kernel(const Args a) { produce(a, blockIdx.x); }
__global__ __launch_bounds__(256) void consumer_kernel(const Args a) { consume(a, blockIdx.x, false); }

int main(int argc, char** argv) {
  const int P = argc > 1 ? std::atoi(argv[1]) : 780, G = P / kPerGroup;
  const double work_us = argc > 2 ? std::atof(argv[2]) : 5.0;
  Args a{};
  a.P = P;
  a.G = G;
  a.work_ticks = (long long)(work_us * 100.0);
  hipMalloc(&a.rec, sizeof(double) * P * kRec);
  hipMalloc(&a.counter, sizeof(unsigned) * G);
  hipMemset(a.counter, 0, sizeof(unsigned) * G);
  hipHostMalloc((void**)&a.host_rows, sizeof(double) * G * 8, hipHostMallocMapped);
  volatile unsigned long long* stamps = reinterpret_cast<volatile unsigned long long*>(a.host_rows);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  auto wait_rows = [&](unsigned long long seq) {
    for (int g = 0; g < G; ++g)
      while (stamps[g * 8 + 7] != seq) {
      }
  };
  const int n = 3000;
  unsigned long long seq = 0;
  for (int mode = 0; mode < 2; ++mode) {
    double total = 0.0;
    for (int it = 0; it < n + 200; ++it) {
      a.seq = ++seq;
      const auto t0 = std::chrono::steady_clock::now();
      if (mode == 0) {
        hipLaunchKernelGGL(producer_kernel, dim3(P), dim3(256), 0, s, a);
        hipLaunchKernelGGL(consumer_kernel, dim3(G), dim3(256), 0, s, a);
      } else {
        hipLaunchKernelGGL(fused_kernel, dim3(P + G), dim3(256), 0, s, a);
      }
      wait_rows(seq);
      const auto t1 = std::chrono::steady_clock::now();
      if (it >= 200) total += std::chrono::duration<double>(t1 - t0).count();
      hipStreamSynchronize(s);
    }
    std::printf("%s: %d producers busy %.1f us, %d groups: launch -> last row on the host %.2f us\n", mode == 0 ? "two launches   " : "fused consumers", P, work_us, G,
                1e6 * total / n);
    if (mode == 0) {  // the fused kernel waits for seq x 10 arrivals: start its counters at the current sequence number
      std::vector<unsigned> c(G, (unsigned)(seq * kPerGroup));
      hipMemcpy(a.counter, c.data(), sizeof(unsigned) * G, hipMemcpyHostToDevice);
    }
  }
  return 0;
}
