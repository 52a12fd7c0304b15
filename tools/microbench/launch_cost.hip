// Diagnostic: host cost of hipLaunchKernelGGL and launch-to-completion latency as a function of the
// kernel-argument size.   hipcc --offload-arch=gfx950 -O2 launch_cost.hip -o launch_cost
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

template <int N>
struct Args {
  double v[N];
};
template <int N>
__global__ void k(Args<N> a, double* out) {
  if (a.v[0] == 12345.0) out[0] = a.v[N - 1];
}

template <int N>
void run(double* out, hipStream_t s) {
  Args<N> a;
  for (int i = 0; i < N; ++i) a.v[i] = i;
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, s, a, out);
  hipStreamSynchronize(s);
  const int n = 2000;
  double t_launch = 0, t_total = 0;
  for (int i = 0; i < n; ++i) {
    const auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, s, a, out);
    const auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    const auto t2 = std::chrono::steady_clock::now();
    t_launch += std::chrono::duration<double>(t1 - t0).count();
    t_total += std::chrono::duration<double>(t2 - t0).count();
  }
  // back-to-back launches without synchronising: steady-state submission cost
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, s, a, out);
  const auto t1 = std::chrono::steady_clock::now();
  hipStreamSynchronize(s);
  std::printf("kernarg %5zu B: launch call %.2f us, launch+sync %.2f us, back-to-back submit %.2f us/launch\n", sizeof(Args<N>) + 8, 1e6 * t_launch / n, 1e6 * t_total / n,
              1e6 * std::chrono::duration<double>(t1 - t0).count() / n);
}

// the same through hipModuleLaunchKernel with a pre-packed argument buffer (no per-launch symbol lookup)
template <int N>
void run_module(double* out, hipStream_t s) {
  struct Packed {
    Args<N> a;
    double* out;
  } p;
  for (int i = 0; i < N; ++i) p.a.v[i] = i;
  p.out = out;
  hipFunction_t f;
  if (hipGetFuncBySymbol(&f, reinterpret_cast<const void*>(&k<N>)) != hipSuccess) {
    std::printf("hipGetFuncBySymbol failed\n");
    return;
  }
  size_t size = sizeof(p);
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  for (int i = 0; i < 200; ++i) hipModuleLaunchKernel(f, 1, 1, 1, 64, 1, 1, 0, s, nullptr, extra);
  hipStreamSynchronize(s);
  const int n = 2000;
  double t_launch = 0;
  for (int i = 0; i < n; ++i) {
    const auto t0 = std::chrono::steady_clock::now();
    hipModuleLaunchKernel(f, 1, 1, 1, 64, 1, 1, 0, s, nullptr, extra);
    const auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    t_launch += std::chrono::duration<double>(t1 - t0).count();
  }
  std::printf("kernarg %5zu B: hipModuleLaunchKernel call %.2f us\n", sizeof(p), 1e6 * t_launch / n);
}

int main() {
  double* out;
  hipMalloc(&out, 64);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  run<4>(out, s);
  run<32>(out, s);
  run<128>(out, s);
  run<256>(out, s);
  run<420>(out, s);
  run<500>(out, s);
  run_module<4>(out, s);
  run_module<420>(out, s);
  return 0;
}
