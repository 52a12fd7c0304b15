// Diagnostic: can the LAST-ARRIVING tile of a group merge the group's records inside the producing launch -- no second launch --
// if all tiles of a group run on ONE XCD and talk through that XCD's L2?  Round 2 tried "last arriver merges" with agent-scope
// atomics and write-through records: the eight XCDs have separate L2s, so every workgroup paid a round trip to memory (~5 us).
// Here block b runs on XCD b mod 8 (xcc_map.hip: no deviation in any condition tried), group g's ten tiles are the blocks
// 8 (10 (g / 8) + t) + g mod 8, records are plain stores (the vector L1 is write-through: they are in the XCD's L2 once vmcnt
// says so), the arrival ticket is an L2 atomic WITHOUT sc1 (workgroup scope in the compiler's terms: executed at the L2, which
// all CUs of an XCD share), and the last arriver reads the records with L1-bypassing loads (sc0).  Values change with every
// launch, so a stale read shows up as a mismatch; every merged row also carries the XCC ids its tiles ran on.
//   hipcc --offload-arch=gfx950 -O2 xcd_merge.hip -o xcd_merge && ./xcd_merge [groups] [work_us]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kVals = 9, kPerGroup = 10, kLines = 2;

struct Args {
  double* rec;          // device: [blocks][16]
  unsigned* tickets;    // device: [G]
  double* host_rows;    // pinned: [G][kLines][8]
  unsigned long long seq;
  int G;
  long long work_ticks;
};

__device__ inline void busy(long long ticks) {
  const long long t0 = (long long)__builtin_amdgcn_s_memrealtime();
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}
__host__ __device__ inline double value(int g, int t, int v, unsigned long long seq) {
  if (v == 0) return (double)((g * 7 + t * 3 + (int)(seq % 5)) % 11 - 5);         // binade exponent of the tile
  return 1.0 + 0.37 * v + 1e-3 * ((g * kPerGroup + t) % 97) + 1e-6 * (double)(seq % 1000);  // changes with every launch
}
__device__ inline void publish_lines(double* dst, const double* vals, int n_vals, unsigned long long seq, int tid) {
  if (tid < 8 * kLines) {
    const int line = tid >> 3, j = tid & 7, i = line * 7 + j;
    const unsigned long long bits = j == 7 ? seq : (i < n_vals ? (unsigned long long)__double_as_longlong(vals[i]) : 0ull);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst) + tid, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// grid = 8 * ceil(G / 8) * kPerGroup blocks of 64 lanes
__global__ __launch_bounds__(64) void fused_kernel(const Args a) {
  const int b = blockIdx.x, x = b & 7, j = b >> 3;
  const int g = (j / kPerGroup) * 8 + x, t = j % kPerGroup;
  if (g >= a.G) return;
  const int lane = threadIdx.x;
  busy(a.work_ticks);
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  double* mine = a.rec + ((long long)g * kPerGroup + t) * 16;
  if (lane < kVals) mine[lane] = value(g, t, lane, a.seq);
  if (lane == kVals) mine[lane] = (double)(xcc & 0xf);
  // the record is in the L2 before the ticket is taken: the stores have been acknowledged (write-through L1), then the L2 atomic
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __shared__ unsigned s_ticket;
  if (lane == 0) s_ticket = __hip_atomic_fetch_add(a.tickets + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __syncthreads();
  if ((s_ticket + 1) % kPerGroup != 0) return;  // not the last tile of its group in this launch
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  // ---- the last arriver: merge the ten records out of the L2 (loads that bypass this CU's L1), publish two lines
  __shared__ double row[16];
  const double* r = a.rec + (long long)g * kPerGroup * 16;
  const bool has = lane < kPerGroup;
  const double m = has ? __hip_atomic_load(r + lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : -1e300;
  const double xc = has ? __hip_atomic_load(r + lane * 16 + kVals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : (double)(xcc & 0xf);
  double mx = m;
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
  const unsigned long long off_xcd = __ballot(xc != (double)(xcc & 0xf));  // tiles that ran on another XCD than the merger
  double mts[kPerGroup];  // (every lane takes part in the shuffles: lane 9 holds a tile's exponent too)
  for (int tt = 0; tt < kPerGroup; ++tt) mts[tt] = __shfl(m, tt);
  if (lane < kVals) {
    double s = 0.0;
    for (int tt = 0; tt < kPerGroup; ++tt) {
      const double mt = mts[tt];
      const double v = __hip_atomic_load(r + tt * 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      s += lane == 0 ? 0.0 : ldexp(v, (int)(mt - mx) * (lane == 2 ? 2 : 1));
    }
    row[lane] = lane == 0 ? mx : s;
  }
  if (lane == kVals) row[lane] = off_xcd ? -1.0 : 1.0;  // tenth value of the row: all tiles on the merger's XCD?
  __syncthreads();
  publish_lines(a.host_rows + (long long)g * kLines * 8, row, kVals + 1, a.seq, lane);
}

// Variant without a ticket: the group's LAST tile by index (dispatched after the others, so they are resident or done) is the
// merger; every record carries the launch's sequence number in its sixteenth slot, stored by the same instruction as the values
// (one 128-byte line); the merger re-reads the ten records through the L2 until all ten carry this launch's number.
// MEASURED: it never sees them -- 2 000 polls (600 us) with `buffer_inv sc0` before every round of sc0 loads still return what the
// first poll fetched: a workgroup-scope load is served from the CU's L1, and nothing short of an agent-scope invalidate empties it.
__global__ __launch_bounds__(64) void polling_kernel(const Args a) {
  const int b = blockIdx.x, x = b & 7, j = b >> 3;
  const int g = (j / kPerGroup) * 8 + x, t = j % kPerGroup;
  if (g >= a.G) return;
  const int lane = threadIdx.x;
  busy(a.work_ticks);
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  double* mine = a.rec + ((long long)g * kPerGroup + t) * 16;
  if (lane < 16) {
    const double v = lane < kVals ? value(g, t, lane, a.seq) : (lane == kVals ? (double)(xcc & 0xf) : (lane == 15 ? __longlong_as_double((long long)a.seq) : 0.0));
    mine[lane] = v;
  }
  if (t != kPerGroup - 1) return;
  __shared__ double row[16];
  const double* r = a.rec + (long long)g * kPerGroup * 16;
  // lanes 0..9: tile (lane)'s header + stamp; all lanes: values; repeat until the ten stamps are this launch's
  double m = -1e300, xc = (double)(xcc & 0xf), vals[kPerGroup];
  for (int spin = 0; spin < 2000; ++spin) {
    asm volatile("buffer_inv sc0" ::: "memory");  // this CU's L1: a workgroup-scope load may be served from it, and it holds what the poll before fetched
    const bool has = lane < kPerGroup;
    const unsigned long long st = has ? (unsigned long long)__double_as_longlong(__hip_atomic_load(r + lane * 16 + 15, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) : a.seq;
    m = has ? __hip_atomic_load(r + lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : -1e300;
    xc = has ? __hip_atomic_load(r + lane * 16 + kVals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : (double)(xcc & 0xf);
#pragma unroll
    for (int tt = 0; tt < kPerGroup; ++tt) vals[tt] = lane < kVals ? __hip_atomic_load(r + tt * 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0.0;
    if (__ballot(st != a.seq) == 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
  double mx = m;
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
  const unsigned long long off_xcd = __ballot(xc != (double)(xcc & 0xf));
  double mts2[kPerGroup];
  for (int tt = 0; tt < kPerGroup; ++tt) mts2[tt] = __shfl(m, tt);
  if (lane < kVals) {
    double s = 0.0;
#pragma unroll
    for (int tt = 0; tt < kPerGroup; ++tt) s += lane == 0 ? 0.0 : ldexp(vals[tt], (int)(mts2[tt] - mx) * (lane == 2 ? 2 : 1));
    row[lane] = lane == 0 ? mx : s;
  }
  if (lane == kVals) row[lane] = off_xcd ? -1.0 : 1.0;
  __syncthreads();
  publish_lines(a.host_rows + (long long)g * kLines * 8, row, kVals + 1, a.seq, lane);
}

// the two-launch reference: producers write records, a second launch merges them (system-scope loads, as the engine's combine)
__global__ __launch_bounds__(64) void producer_kernel(const Args a) {
  const int b = blockIdx.x, x = b & 7, j = b >> 3;
  const int g = (j / kPerGroup) * 8 + x, t = j % kPerGroup;
  if (g >= a.G) return;
  busy(a.work_ticks);
  double* mine = a.rec + ((long long)g * kPerGroup + t) * 16;
  if (threadIdx.x < kVals) mine[threadIdx.x] = value(g, t, threadIdx.x, a.seq);
}
__global__ __launch_bounds__(64) void combine_kernel(const Args a) {
  const int g = blockIdx.x, lane = threadIdx.x;
  __shared__ double row[16];
  const double* r = a.rec + (long long)g * kPerGroup * 16;
  const bool has = lane < kPerGroup;
  const double m = has ? __hip_atomic_load(r + lane * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : -1e300;
  double mx = m;
  for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
  double mts[kPerGroup];  // (every lane takes part in the shuffles: lane 9 holds a tile's exponent too)
  for (int tt = 0; tt < kPerGroup; ++tt) mts[tt] = __shfl(m, tt);
  if (lane < kVals) {
    double s = 0.0;
    for (int tt = 0; tt < kPerGroup; ++tt) {
      const double mt = mts[tt];
      const double v = __hip_atomic_load(r + tt * 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      s += lane == 0 ? 0.0 : ldexp(v, (int)(mt - mx) * (lane == 2 ? 2 : 1));
    }
    row[lane] = lane == 0 ? mx : s;
  }
  if (lane == kVals) row[lane] = 1.0;
  __syncthreads();
  publish_lines(a.host_rows + (long long)g * kLines * 8, row, kVals + 1, a.seq, lane);
}

int main(int argc, char** argv) {
  const int G = argc > 1 ? std::atoi(argv[1]) : 78;
  const double work_us = argc > 2 ? std::atof(argv[2]) : 5.0;
  const int blocks = 8 * ((G + 7) / 8) * kPerGroup;
  Args a{};
  a.G = G;
  a.work_ticks = (long long)(work_us * 100.0);
  hipMalloc(&a.rec, sizeof(double) * (size_t)((G + 7) / 8 * 8) * kPerGroup * 16);
  hipMalloc(&a.tickets, sizeof(unsigned) * G);
  hipMemset(a.tickets, 0, sizeof(unsigned) * G);
  double* rows = nullptr;
  hipHostMalloc((void**)&rows, sizeof(double) * (size_t)G * kLines * 8, hipHostMallocMapped);
  a.host_rows = rows;
  volatile unsigned long long* stamps = reinterpret_cast<volatile unsigned long long*>(rows);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int n = 5000;
  unsigned long long seq = 0;
  const int n_modes = argc > 3 ? 3 : 2;  // a third argument also runs the polling variant (which never sees the records: see below)
  for (int mode = 0; mode < n_modes; ++mode) {
    double total = 0.0;
    long long wrong = 0, off = 0;
    for (int it = 0; it < n + 200; ++it) {
      a.seq = ++seq;
      const auto t0 = std::chrono::steady_clock::now();
      if (mode == 0) {
        hipLaunchKernelGGL(producer_kernel, dim3(blocks), dim3(64), 0, s, a);
        hipLaunchKernelGGL(combine_kernel, dim3(G), dim3(64), 0, s, a);
      } else if (mode == 1) {
        hipLaunchKernelGGL(fused_kernel, dim3(blocks), dim3(64), 0, s, a);
      } else {
        hipLaunchKernelGGL(polling_kernel, dim3(blocks), dim3(64), 0, s, a);
      }
      for (int l = 0; l < G * kLines; ++l)
        while (stamps[l * 8 + 7] != seq) {
        }
      const auto t1 = std::chrono::steady_clock::now();
      asm volatile("" ::: "memory");  // the rows are read through a plain pointer: not before the stamps have been seen
      if (it >= 200) total += std::chrono::duration<double>(t1 - t0).count();
      // check every merged row against the same merge done here
      for (int g = 0; g < G; ++g) {
        const double* L = rows + (size_t)g * kLines * 8;
        double got[10];
        for (int i = 0; i < 7; ++i) got[i] = L[i];
        got[7] = L[8], got[8] = L[9], got[9] = L[10];
        double mx = -1e300;
        for (int t = 0; t < kPerGroup; ++t) mx = std::fmax(mx, value(g, t, 0, seq));
        bool bad = got[0] != mx;
        auto differs = [](double x, double y) { return std::fabs(x - y) > 1e-9 * (1.0 + std::fabs(y)); };
        for (int v = 1; v < kVals && !bad; ++v) {
          double sref = 0.0;
          for (int t = 0; t < kPerGroup; ++t) sref += std::ldexp(value(g, t, v, seq), (int)(value(g, t, 0, seq) - mx) * (v == 2 ? 2 : 1));
          bad = differs(got[v], sref);
        }
        if (bad && wrong < 3) {
          std::printf("  row %d of launch %llu: got", g, seq);
          for (int v = 0; v < 10; ++v) std::printf(" %.12g", got[v]);
          std::printf("; expected exponent %.0f\n", mx);
        }
        wrong += bad;
        off += got[9] < 0.0;
      }
      hipStreamSynchronize(s);
    }
    std::printf("%s: %d groups of %d tiles busy %.1f us: launch -> all %d lines on the host %.2f us; rows that differ from the host's merge: %lld of %lld; rows with a tile on another XCD: %lld\n",
                mode == 0 ? "two launches (records -> combine, system-scope loads)" : (mode == 1 ? "ONE launch, last arriver (L2 ticket) merges             " : "ONE launch, the group's last tile polls the L2 and merges"), G, kPerGroup, work_us, G * kLines,
                1e6 * total / n, wrong, (long long)(n + 200) * G, off);
  }
  return 0;
}
