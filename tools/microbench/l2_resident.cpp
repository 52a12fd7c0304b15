// Diagnostic (GPU box): does a catalog that fits the eight 4 MB L2s stay there from one launch to the next when the dispatch
// packet carries NO acquire fence?  A HIP-stream launch (and the engine's AQL packets so far) starts every kernel with an
// agent-scope acquire, which invalidates the L2s: config 2's 12.6 MB are fetched again per evaluation (FETCH_SIZE = the whole
// catalog, profiles/round4) although workgroup b always runs on XCD b mod 8 (xcc_map.hip) and so always reads the same 1.6 MB
// through the same L2.  The same streaming kernel dispatched with acquire scope none / agent / system, kernel time from the
// queue's dispatch timestamps and host time from the doorbell to the completion signal.
//   hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 l2_resident_kernel.hip -o l2_resident_kernel.hsaco
//   hipcc -O2 l2_resident.cpp -o l2_resident -lhsa-runtime64 && ./l2_resident l2_resident_kernel.hsaco
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                             \
  do {                                                    \
    hsa_status_t s_ = (x);                                \
    if (s_ != HSA_STATUS_SUCCESS) {                       \
      const char* m_ = nullptr;                           \
      hsa_status_string(s_, &m_);                         \
      std::printf("%s failed: %s\n", #x, m_ ? m_ : "?");  \
      return 1;                                           \
    }                                                     \
  } while (0)
#define HK(x)                                                      \
  do {                                                             \
    hipError_t e_ = (x);                                           \
    if (e_ != hipSuccess) {                                        \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));   \
      return 1;                                                    \
    }                                                              \
  } while (0)

constexpr int kMaxCols = 9;
struct Args {
  const double* col[kMaxCols];
  double* out;
  long long n;
  int n_cols, tile, work, pad;
};

static hsa_agent_t g_gpu, g_cpu;
static bool g_have_gpu = false, g_have_cpu = false;
static hsa_status_t pick_agents(hsa_agent_t a, void*) {
  hsa_device_type_t t;
  hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) g_gpu = a, g_have_gpu = true;
  if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) g_cpu = a, g_have_cpu = true;
  return HSA_STATUS_SUCCESS;
}
static hsa_amd_memory_pool_t g_dev_pool;
static bool g_have_pool = false;
static hsa_status_t pick_pool(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  uint32_t flags = 0;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  bool alloc = false;
  hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_pool) g_dev_pool = p, g_have_pool = true;
  return HSA_STATUS_SUCCESS;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  if (argc < 2) return std::printf("usage: l2_resident <hsaco>\n"), 1;
  HK(hipSetDevice(0));
  CK(hsa_init());
  CK(hsa_iterate_agents(pick_agents, nullptr));
  if (!g_have_gpu || !g_have_cpu) return std::printf("no agents\n"), 1;
  hsa_queue_t* q = nullptr;
  CK(hsa_queue_create(g_gpu, 1024, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
  CK(hsa_amd_profiling_set_profiler_enabled(q, 1));
  uint64_t tick_hz = 0;
  CK(hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &tick_hz));
  std::vector<char> blob;
  {
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) return std::printf("cannot open %s\n", argv[1]), 1;
    std::fseek(f, 0, SEEK_END);
    blob.resize(std::ftell(f));
    std::fseek(f, 0, SEEK_SET);
    if (std::fread(blob.data(), 1, blob.size(), f) != blob.size()) return 1;
    std::fclose(f);
  }
  hsa_code_object_reader_t reader;
  CK(hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &reader));
  hsa_executable_t exe;
  CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
  CK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
  CK(hsa_executable_freeze(exe, nullptr));
  hsa_executable_symbol_t sym;
  CK(hsa_executable_get_symbol_by_name(exe, "stream_kernel.kd", &g_gpu, &sym));
  uint64_t kobj = 0;
  uint32_t karg_size = 0, group = 0, priv = 0;
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &karg_size));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &group));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &priv));
  std::printf("kernarg %u B (explicit %zu), LDS %u B, scratch %u B\n", karg_size, sizeof(Args), group, priv);
  hsa_executable_symbol_t sym_p;
  CK(hsa_executable_get_symbol_by_name(exe, "stream_kernel_preload.kd", &g_gpu, &sym_p));
  uint64_t kobj_p = 0;
  uint32_t karg_size_p = 0;
  CK(hsa_executable_symbol_get_info(sym_p, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj_p));
  CK(hsa_executable_symbol_get_info(sym_p, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &karg_size_p));
  struct ArgsP {
    const double* c[4];
    int tile, n_cols;
    long long n;
    int work, pad;
    Args a;
  };
  std::printf("preload variant: kernarg %u B (explicit %zu)\n", karg_size_p, sizeof(ArgsP));
  if (karg_size > 1024 || karg_size_p > 1024) return std::printf("unexpected kernarg size\n"), 1;
  CK(hsa_amd_agent_iterate_memory_pools(g_gpu, pick_pool, nullptr));
  if (!g_have_pool) return std::printf("no device pool\n"), 1;
  char* karg = nullptr;
  CK(hsa_amd_memory_pool_allocate(g_dev_pool, 4096, 0, (void**)&karg));
  if (hsa_amd_agents_allow_access(1, &g_cpu, nullptr, karg) != HSA_STATUS_SUCCESS) return std::printf("device kernarg not host-accessible\n"), 1;
  hsa_signal_t done;
  CK(hsa_signal_create(1, 0, nullptr, &done));
  double* out;
  HK(hipMalloc(&out, 64));

  struct Case {
    const char* name;
    long long n;
    int cols, tile;
  } cases[] = {{"config 2 (4 x 395 k, 12.6 MB)", 395000, 4, 512},
               {"config 3 (8 x 445 k, 28.5 MB)", 445000, 8, 456},
               {"half of config 3 (8 x 222 k, 14.2 MB)", 222000, 8, 456},
               {"config 5 (9 x 2.5 M, 180 MB)", 2500000, 9, 1280}};
  auto* base = static_cast<hsa_kernel_dispatch_packet_t*>(q->base_address);
  const uint32_t mask = q->size - 1;
  for (const Case& cs : cases) {
    Args a;
    std::memset(&a, 0, sizeof(a));
    a.n = cs.n, a.n_cols = cs.cols, a.tile = cs.tile, a.out = out;
    std::vector<double*> bufs;
    for (int c = 0; c < cs.cols; ++c) {
      double* p;
      HK(hipMalloc(&p, sizeof(double) * (cs.n + 2)));
      HK(hipMemset(p, 0, sizeof(double) * (cs.n + 2)));
      a.col[c] = p;
      bufs.push_back(p);
    }
    HK(hipDeviceSynchronize());
    const int grid = (int)((cs.n + a.tile - 1) / a.tile);
    for (int work : {0, 48}) {
      a.work = work;
      std::memset(karg, 0, 1024);
      std::memcpy(karg, &a, sizeof(a));
      __sync_synchronize();
      volatile char sink = karg[sizeof(a) - 1];  // read back through the BAR: the posted writes have landed
      (void)sink;
      ArgsP ap;
      std::memset(&ap, 0, sizeof(ap));
      for (int c = 0; c < 4; ++c) ap.c[c] = a.col[c];
      ap.tile = a.tile, ap.n_cols = a.n_cols, ap.n = a.n, ap.work = a.work, ap.a = a;
      std::memcpy(karg + 2048, &ap, sizeof(ap));
      __sync_synchronize();
      sink = karg[2048 + sizeof(ap) - 1];
      for (int scope : {(int)HSA_FENCE_SCOPE_AGENT, (int)HSA_FENCE_SCOPE_NONE, (int)HSA_FENCE_SCOPE_SYSTEM, (int)HSA_FENCE_SCOPE_NONE, -1, (int)HSA_FENCE_SCOPE_AGENT, -1, -2, (int)HSA_FENCE_SCOPE_AGENT, -2}) {
        const bool preload = scope == -1;
        const bool no_release = scope == -2;  // agent-scope acquire, NO release fence at the end of the kernel
        if (scope < 0) scope = HSA_FENCE_SCOPE_AGENT;
        const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                (scope << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | ((no_release ? HSA_FENCE_SCOPE_NONE : HSA_FENCE_SCOPE_AGENT) << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        std::vector<double> kern, host;
        for (int it = 0; it < 60; ++it) {
          hsa_signal_store_relaxed(done, 1);
          const auto t0 = std::chrono::steady_clock::now();
          const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
          hsa_kernel_dispatch_packet_t* p = base + (idx & mask);
          p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
          p->workgroup_size_x = 256, p->workgroup_size_y = 1, p->workgroup_size_z = 1;
          p->grid_size_x = (uint32_t)grid * 256, p->grid_size_y = 1, p->grid_size_z = 1;
          p->private_segment_size = priv, p->group_segment_size = group;
          p->kernel_object = preload ? kobj_p : kobj;
          p->kernarg_address = preload ? karg + 2048 : karg;
          p->completion_signal = done;
          __atomic_store_n(reinterpret_cast<uint32_t*>(p), (uint32_t)header | ((uint32_t)p->setup << 16), __ATOMIC_RELEASE);
          hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
          if (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) != 0) return std::printf("dispatch did not complete\n"), 3;
          const auto t1 = std::chrono::steady_clock::now();
          hsa_amd_profiling_dispatch_time_t dt;
          CK(hsa_amd_profiling_get_dispatch_time(g_gpu, done, &dt));
          if (it >= 10) {
            kern.push_back(1e6 * (double)(dt.end - dt.start) / (double)tick_hz);
            host.push_back(1e6 * std::chrono::duration<double>(t1 - t0).count());
          }
        }
        std::sort(kern.begin(), kern.end());
        std::sort(host.begin(), host.end());
        const char* sname = no_release ? "agent, no release fence" : preload ? "agent, scalar arguments preloaded" : (scope == HSA_FENCE_SCOPE_NONE ? "none  " : (scope == HSA_FENCE_SCOPE_AGENT ? "agent " : "system"));
        std::printf("%-40s W = %2d  acquire %s: kernel median %7.2f us (min %7.2f)   doorbell -> signal median %7.2f us\n", cs.name, work, sname, kern[kern.size() / 2], kern[0],
                    host[host.size() / 2]);
      }
    }
    if (cs.cols == 4) {  // config 2's shape: the same bytes through ~250 workgroups of 1 024 lanes (tiles of 1 580 samples)
      hsa_executable_symbol_t sym_f;
      CK(hsa_executable_get_symbol_by_name(exe, "stream_kernel_fat.kd", &g_gpu, &sym_f));
      uint64_t kobj_f = 0;
      CK(hsa_executable_symbol_get_info(sym_f, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj_f));
      for (int work : {48, 110, 170}) {
        for (int fat : {0, 1}) {
          ArgsP ap;
          std::memset(&ap, 0, sizeof(ap));
          for (int c = 0; c < 4; ++c) ap.c[c] = a.col[c];
          ap.tile = fat ? 1580 : a.tile, ap.n_cols = 4, ap.n = a.n, ap.work = work, ap.a = a;
          std::memcpy(karg + 2048, &ap, sizeof(ap));
          __sync_synchronize();
          volatile char sink2 = karg[2048 + sizeof(ap) - 1];
          (void)sink2;
          const int g = (int)((cs.n + ap.tile - 1) / ap.tile);
          const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                  (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
          std::vector<double> kern;
          for (int it = 0; it < 60; ++it) {
            hsa_signal_store_relaxed(done, 1);
            const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
            hsa_kernel_dispatch_packet_t* p = base + (idx & mask);
            p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
            p->workgroup_size_x = fat ? 1024 : 256, p->workgroup_size_y = 1, p->workgroup_size_z = 1;
            p->grid_size_x = (uint32_t)g * (fat ? 1024 : 256), p->grid_size_y = 1, p->grid_size_z = 1;
            p->private_segment_size = priv, p->group_segment_size = group;
            p->kernel_object = fat ? kobj_f : kobj_p;
            p->kernarg_address = karg + 2048;
            p->completion_signal = done;
            __atomic_store_n(reinterpret_cast<uint32_t*>(p), (uint32_t)header | ((uint32_t)p->setup << 16), __ATOMIC_RELEASE);
            hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
            if (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) != 0) return std::printf("dispatch did not complete\n"), 3;
            hsa_amd_profiling_dispatch_time_t dt;
            CK(hsa_amd_profiling_get_dispatch_time(g_gpu, done, &dt));
            if (it >= 10) kern.push_back(1e6 * (double)(dt.end - dt.start) / (double)tick_hz);
          }
          std::sort(kern.begin(), kern.end());
          std::printf("%-40s W = %3d  %s: kernel median %7.2f us (min %7.2f)\n", cs.name, work, fat ? "250 workgroups of 1 024 lanes, preloaded" : "772 workgroups of 256 lanes, preloaded  ", kern[kern.size() / 2], kern[0]);
        }
      }
    }
    for (double* p : bufs) HK(hipFree(p));
  }
  hsa_queue_destroy(q);
  return 0;
}
