// Diagnostic: issue cost (cycles per wave-instruction per SIMD) of the VALU / LDS instructions the scan
// kernel is built from, measured on the GPU it runs on.  Each kernel runs a long unrolled stream of
// INDEPENDENT instances of one instruction (8 register chains per lane) with 8 waves per SIMD resident,
// so the figure is throughput, not latency.   hipcc --offload-arch=gfx950 -O2 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

constexpr int kIters = 512;
constexpr int kChains = 8;

#define KERNEL_D(NAME, ASM)                                                                   \
  __global__ __launch_bounds__(256) void NAME(double* out, double seed) {                     \
    double a[kChains];                                                                        \
    for (int c = 0; c < kChains; ++c) a[c] = seed + c + threadIdx.x * 1e-3;                   \
    double b = seed * 0.999, d = seed * 1e-3;                                                 \
    int ii = threadIdx.x;                                                                     \
    (void)ii;                                                                                 \
    for (int it = 0; it < kIters; ++it) {                                                     \
      _Pragma("unroll") for (int c = 0; c < kChains; ++c) { ASM; }                            \
    }                                                                                         \
    double s = 0;                                                                             \
    for (int c = 0; c < kChains; ++c) s += a[c];                                              \
    if (s == 12345.678) out[0] = s;                                                           \
  }

KERNEL_D(k_fma_f64, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "v"(d)))
KERNEL_D(k_mul_f64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b)))
KERNEL_D(k_add_f64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(d)))
KERNEL_D(k_max_f64, asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[c]) : "v"(d)))
KERNEL_D(k_floor_f64, asm volatile("v_floor_f64 %0, %0" : "+v"(a[c])))
KERNEL_D(k_rndne_f64, asm volatile("v_rndne_f64 %0, %0" : "+v"(a[c])))
KERNEL_D(k_fract_f64, asm volatile("v_fract_f64 %0, %0" : "+v"(a[c])))
KERNEL_D(k_rcp_f64, asm volatile("v_rcp_f64 %0, %0" : "+v"(a[c])))
KERNEL_D(k_rsq_f64, asm volatile("v_rsq_f64 %0, %0" : "+v"(a[c])))
KERNEL_D(k_ldexp_f64, asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a[c]) : "v"(ii)))
KERNEL_D(k_cvt_i32_f64, {
  int t_;
  asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t_) : "v"(a[c]));
  asm volatile("" ::"v"(t_));
})
KERNEL_D(k_cvt_f64_i32, asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[c]) : "v"(ii)))
KERNEL_D(k_cmp_f64, asm volatile("v_cmp_lt_f64 vcc, %0, %1" ::"v"(a[c]), "v"(b) : "vcc"))
KERNEL_D(k_cmp_class_f64, asm volatile("v_cmp_class_f64 vcc, %0, %1" ::"v"(a[c]), "v"(ii) : "vcc"))
KERNEL_D(k_frexp_mant_f64, asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(a[c])))
KERNEL_D(k_div_fixup_f64, asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "v"(d)))
KERNEL_D(k_mov_dpp, {
  int* p_ = reinterpret_cast<int*>(&a[c]);
  asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(p_[0]));
})

#define KERNEL_I(NAME, ASM)                                                                   \
  __global__ __launch_bounds__(256) void NAME(double* out, double seed) {                     \
    int a[kChains];                                                                           \
    for (int c = 0; c < kChains; ++c) a[c] = (int)seed + c + threadIdx.x;                     \
    int b = (int)seed + 3, d = threadIdx.x & 1;                                               \
    float fb = (float)seed;                                                                   \
    (void)fb;                                                                                 \
    (void)d;                                                                                  \
    for (int it = 0; it < kIters; ++it) {                                                     \
      _Pragma("unroll") for (int c = 0; c < kChains; ++c) { ASM; }                            \
    }                                                                                         \
    int s = 0;                                                                                \
    for (int c = 0; c < kChains; ++c) s += a[c];                                              \
    if (s == 123456789) out[0] = s;                                                           \
  }

KERNEL_I(k_add_u32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b)))
KERNEL_I(k_cndmask_b32, asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "s"(__builtin_amdgcn_ballot_w64(d != 0))))
KERNEL_I(k_fma_f32, asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(fb)))
KERNEL_I(k_exp_f32, asm volatile("v_exp_f32 %0, %0" : "+v"(a[c])))
KERNEL_I(k_mul_lo_u32, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b)))
KERNEL_I(k_lshl_add_u32, asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[c]) : "v"(b)))
KERNEL_I(k_med3_i32, asm volatile("v_med3_i32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b)))
KERNEL_I(k_pk_fma_f32, {
  double* p_ = reinterpret_cast<double*>(&a[c & ~1]);
  asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*p_));
})

// LDS: reads and f64 atomic adds with a chosen number of lanes per address
template <int LANES_PER_ADDR, bool ATOMIC>
__global__ __launch_bounds__(256) void k_lds(double* out, double seed) {
  __shared__ double s[4][kChains * 64 + 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = lane; i < kChains * 64 + 64; i += 64) s[wave][i] = 0.0;
  __syncthreads();
  double acc = 0.0;
  double* base = &s[wave][(lane / LANES_PER_ADDR)];
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
      if (ATOMIC)
        unsafeAtomicAdd(base + c * 64, seed);
      else
        acc += *(volatile double*)(base + c * 64);
    }
  }
  if (acc == 12345.678) out[0] = acc;
}

// LDS reads through explicit ds instructions: MODE 0 ds_read_b64, 1 ds_read2_b64, 2 ds_read_b128;
// LANES_PER_ADDR = 1: every lane its own address (consecutive), 8: eight lanes share one
template <int MODE, int LANES_PER_ADDR>
__global__ __launch_bounds__(256) void k_ldsr(double* out, double seed) {
  __shared__ __attribute__((aligned(16))) double s[4][kChains * 64 * 2 + 128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = lane; i < kChains * 64 * 2 + 128; i += 64) s[wave][i] = seed;
  __syncthreads();
  const unsigned addr = (unsigned)(unsigned long long)(&s[wave][(lane / LANES_PER_ADDR) * (MODE == 0 ? 1 : 2)]);
  double acc = 0.0;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int c = 0; c < kChains; ++c) {
      if (MODE == 0) {
        double v;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(c * 1024));
        asm volatile("" ::"v"(v));
      } else if (MODE == 1) {
        double __attribute__((ext_vector_type(2))) v;
        asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(c * 16), "n"(c * 16 + 1));
        asm volatile("" ::"v"(v));
      } else {
        double __attribute__((ext_vector_type(2))) v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(c * 1024));
        asm volatile("" ::"v"(v));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
  }
  if (acc == 12345.678) out[0] = acc;
}

struct Entry {
  const char* name;
  void (*fn)(double*, double);
};

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  const double mhz = prop.clockRate / 1000.0;
  double* out;
  CHECK(hipMalloc(&out, 64));
  const int waves_per_simd = 8;
  const int blocks = n_cu * waves_per_simd;  // 4 waves per block -> 8 waves on each of the CU's 4 SIMDs
  std::vector<Entry> es = {
      {"v_fma_f64", k_fma_f64},       {"v_mul_f64", k_mul_f64},         {"v_add_f64", k_add_f64},       {"v_max_f64", k_max_f64},
      {"v_floor_f64", k_floor_f64},   {"v_rndne_f64", k_rndne_f64},     {"v_fract_f64", k_fract_f64},   {"v_rcp_f64", k_rcp_f64},
      {"v_rsq_f64", k_rsq_f64},       {"v_ldexp_f64", k_ldexp_f64},     {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_cvt_f64_i32", k_cvt_f64_i32},
      {"v_cmp_lt_f64", k_cmp_f64},    {"v_cmp_class_f64", k_cmp_class_f64}, {"v_frexp_mant_f64", k_frexp_mant_f64},
      {"v_div_fixup_f64", k_div_fixup_f64}, {"v_mov_b32_dpp", k_mov_dpp}, {"v_add_u32", k_add_u32},     {"v_cndmask_b32", k_cndmask_b32},
      {"v_fma_f32", k_fma_f32},       {"v_exp_f32", k_exp_f32},         {"v_mul_lo_u32", k_mul_lo_u32}, {"v_lshl_add_u32", k_lshl_add_u32},
      {"v_med3_i32", k_med3_i32},     {"v_pk_fma_f32", k_pk_fma_f32},
      {"ds_read_b64 (1 lane/addr)", k_ldsr<0, 1>},  {"ds_read_b64 (8 lanes/addr)", k_ldsr<0, 8>},
      {"ds_read2_b64 (1 lane/addr)", k_ldsr<1, 1>}, {"ds_read2_b64 (8 lanes/addr)", k_ldsr<1, 8>},
      {"ds_read_b128 (1 lane/addr)", k_ldsr<2, 1>}, {"ds_read_b128 (8 lanes/addr)", k_ldsr<2, 8>},
      {"ds_add_f64 (1 lane/addr)", k_lds<1, true>},    {"ds_add_f64 (2 lanes/addr)", k_lds<2, true>},
      {"ds_add_f64 (4 lanes/addr)", k_lds<4, true>},   {"ds_add_f64 (8 lanes/addr)", k_lds<8, true>},
      {"ds_add_f64 (64 lanes/addr)", k_lds<64, true>},
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  std::printf("device %s: %d CUs, clockRate %.0f MHz; %d waves/SIMD, %d instr per wave\n", prop.name, n_cu, mhz, waves_per_simd, kIters * kChains);
  for (const Entry& e : es) {
    for (int w = 0; w < 2; ++w) e.fn<<<blocks, 256>>>(out, 1.5);
    CHECK(hipDeviceSynchronize());
    const int reps = 5;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) e.fn<<<blocks, 256>>>(out, 1.5);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps;
    const double instr_per_simd = (double)waves_per_simd * kIters * kChains;
    std::printf("%-30s %9.2f us  -> %6.2f cycles per wave-instruction per SIMD (at %.0f MHz)\n", e.name, us, us * mhz / instr_per_simd, mhz);
  }
  return 0;
}
