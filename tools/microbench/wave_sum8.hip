// Diagnostic (GPU box): eight fp64 sums over a wavefront at once -- a halving butterfly on v_permlane32_swap / v_permlane16_swap
// (gfx950) and DPP, against eight row-shift DPP reductions (gwi_device.h wave_sum).  Checks where each total lands and times both.
//   hipcc --offload-arch=gfx950 -O3 wave_sum8.hip -o wave_sum8 && ./wave_sum8
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_take<0x111, 0xf>(v);
  v += dpp_take<0x112, 0xf>(v);
  v += dpp_take<0x114, 0xf>(v);
  v += dpp_take<0x118, 0xf>(v);
  v += dpp_take<0x142, 0xa>(v);
  v += dpp_take<0x143, 0xc>(v);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// a <- [a.lanes 0-31 | b.lanes 0-31], b <- [a.lanes 32-63 | b.lanes 32-63]  (v_permlane32_swap: lanes 32-63 of vdst <-> lanes 0-31 of src)
__device__ __forceinline__ void swap32(double& a, double& b) {
  auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)r1[0], (int)r0[0]);
  b = __hiloint2double((int)r1[1], (int)r0[1]);
}
__device__ __forceinline__ void swap16(double& a, double& b) {
  auto r0 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  auto r1 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)r1[0], (int)r0[0]);
  b = __hiloint2double((int)r1[1], (int)r0[1]);
}
// Totals of v[0..7] over the 64 lanes: every lane of the group 8k..8k+7 returns the total of v[k].
__device__ __forceinline__ double wave_sum8(const double (&v)[8]) {
  double x[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double a = v[j], b = v[j + 4];
    swap32(a, b);
    x[j] = a + b;  // lanes 0-31: v[j] over lanes l, l + 32; lanes 32-63: v[j + 4]
  }
  double y[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    double a = x[j], b = x[j + 2];
    swap16(a, b);
    y[j] = a + b;  // rows 0 / 1 / 2 / 3 of 16 lanes: v[j], v[j + 2], v[j + 4], v[j + 6] over four lanes each
  }
  // lanes with bit 3 clear keep y[0] and take the partner's (lane ^ 8) y[0]; the others y[1]
  const bool up = (__lane_id() & 8) != 0;
  const double keep = up ? y[1] : y[0], send = up ? y[0] : y[1];
  double z = keep + dpp_take<0x128, 0xf>(send);  // row_ror:8
  z += dpp_take<0x141, 0xf>(z);                  // row_half_mirror: lane i <-> 7 - i of its group of eight
  z += dpp_take<0xB1, 0xf>(z);                   // quad_perm [1,0,3,2]
  z += dpp_take<0x4E, 0xf>(z);                   // quad_perm [2,3,0,1]
  return z;
}

__global__ void check(const double* in, double* out8, double* ref8, long long* ticks) {
  const int lane = threadIdx.x;
  double v[8];
  for (int k = 0; k < 8; ++k) v[k] = in[k * 64 + lane];
  long long t0 = __builtin_amdgcn_s_memtime();
  const double z = wave_sum8(v);
  asm volatile("" ::"v"(z));
  long long t1 = __builtin_amdgcn_s_memtime();
  double r[8];
  for (int k = 0; k < 8; ++k) r[k] = wave_sum(v[k]);
  asm volatile("" ::"v"(r[0]), "v"(r[7]));
  long long t2 = __builtin_amdgcn_s_memtime();
  out8[lane] = z;
  if (lane == 0) {
    for (int k = 0; k < 8; ++k) ref8[k] = r[k];
    ticks[0] = t1 - t0;
    ticks[1] = t2 - t1;
  }
}

int main() {
  std::vector<double> h(512);
  for (int k = 0; k < 8; ++k)
    for (int l = 0; l < 64; ++l) h[k * 64 + l] = std::sin(0.37 * l + 1.7 * k) * std::exp(0.1 * k);
  double *in, *out8, *ref8;
  long long* ticks;
  hipMalloc(&in, 512 * 8);
  hipMalloc(&out8, 64 * 8);
  hipMalloc(&ref8, 8 * 8);
  hipMalloc(&ticks, 16);
  hipMemcpy(in, h.data(), 512 * 8, hipMemcpyHostToDevice);
  for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, in, out8, ref8, ticks);
  hipDeviceSynchronize();
  double o[64], r[8];
  long long t[2];
  hipMemcpy(o, out8, sizeof(o), hipMemcpyDeviceToHost);
  hipMemcpy(r, ref8, sizeof(r), hipMemcpyDeviceToHost);
  hipMemcpy(t, ticks, sizeof(t), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    double exact = 0.0;
    for (int j = 0; j < 64; ++j) exact += h[(l / 8) * 64 + j];
    const bool ok = std::fabs(o[l] - exact) <= 1e-13 * (1.0 + std::fabs(exact)) && std::fabs(r[l / 8] - exact) <= 1e-13 * (1.0 + std::fabs(exact));
    if (!ok) {
      ++bad;
      printf("lane %2d: butterfly %.15g, row-shift total of v[%d] %.15g, exact %.15g\n", l, o[l], l / 8, r[l / 8], exact);
    }
  }
  printf("lanes 8k..8k+7 hold the total of v[k]: %s (%d lanes off)\n", bad ? "NO" : "yes", bad);
  printf("s_memtime ticks (100 MHz): butterfly of eight %lld, eight row-shift sums %lld\n", t[0], t[1]);
  return bad != 0;
}
