// Diagnostic (GPU box): launch-to-stamp latency of ONE kernel, through hipModuleLaunchKernel on a HIP stream and through
// an AQL dispatch packet written straight into an HSA user-mode queue of our own (kernel arguments in device memory,
// written by the host through the PCIe BAR; no completion signal: the kernel's own write-through stamp in pinned host
// memory is what the host polls, as the engine does).  Decides whether bypassing the HIP launch path is worth having.
//   hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O2 aql_stamp_kernel.hip -o aql_stamp_kernel_raw.hsaco   (a raw ELF: HSA does not unbundle)
//   hipcc -O2 aql_dispatch.cpp -o aql_dispatch -lhsa-runtime64 && ./aql_dispatch aql_stamp_kernel.hsaco
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                       \
  do {                                                              \
    hsa_status_t s_ = (x);                                          \
    if (s_ != HSA_STATUS_SUCCESS) {                                 \
      const char* m_ = nullptr;                                     \
      hsa_status_string(s_, &m_);                                   \
      std::printf("%s failed: %s\n", #x, m_ ? m_ : "?");            \
      return 1;                                                     \
    }                                                               \
  } while (0)
#define HK(x)                                                             \
  do {                                                                    \
    hipError_t e_ = (x);                                                  \
    if (e_ != hipSuccess) {                                               \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));          \
      return 1;                                                           \
    }                                                                     \
  } while (0)

struct StampArgs {
  unsigned long long* host_slot;
  unsigned int* counter;
  unsigned long long seq;
  double payload[400];
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
  if (argc < 2) return std::printf("usage: aql_dispatch <hsaco>\n"), 1;
  const int blocks = argc > 2 ? std::atoi(argv[2]) : 400;
  HK(hipSetDevice(0));
  unsigned long long* slot;
  HK(hipHostMalloc((void**)&slot, 64, hipHostMallocMapped));
  unsigned long long* slot_dev;
  HK(hipHostGetDevicePointer((void**)&slot_dev, slot, 0));
  unsigned int* counter;
  HK(hipMalloc(&counter, 4));
  HK(hipMemset(counter, 0, 4));
  hipStream_t stream;
  HK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

  // ---- HIP path: module launch with a packed argument buffer
  hipModule_t mod;
  HK(hipModuleLoad(&mod, argv[1]));
  hipFunction_t fn;
  HK(hipModuleGetFunction(&fn, mod, "stamp_kernel"));
  StampArgs args;
  std::memset(&args, 0, sizeof(args));
  args.host_slot = slot_dev;
  args.counter = counter;
  const int n = 3000;
  auto poll = [&](unsigned long long want) {
    const auto t_start = std::chrono::steady_clock::now();
    unsigned long long spins = 0;
    while (*(volatile unsigned long long*)slot != want) {
      __builtin_ia32_pause();
      if ((++spins & 0xfffff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > 5.0) {
        std::printf("stamp %llu never arrived (slot holds %llu)\n", want, *(volatile unsigned long long*)slot);
        std::exit(3);
      }
    }
  };
  unsigned long long seq = 0;
  double t_hip_call = 0, t_hip_total = 0;
  for (int i = 0; i < n + 200; ++i) {
    args.seq = ++seq;
    size_t size = sizeof(args);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    const auto t0 = std::chrono::steady_clock::now();
    HK(hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, stream, nullptr, extra));
    const auto t1 = std::chrono::steady_clock::now();
    poll(seq);
    const auto t2 = std::chrono::steady_clock::now();
    if (i >= 200) {
      t_hip_call += std::chrono::duration<double>(t1 - t0).count();
      t_hip_total += std::chrono::duration<double>(t2 - t0).count();
    }
  }
  std::printf("HIP  launch (%d workgroups, %zu B of arguments): call %.2f us, launch -> stamp seen %.2f us\n", blocks, sizeof(args), 1e6 * t_hip_call / n, 1e6 * t_hip_total / n);

  // ---- AQL path
  CK(hsa_init());
  CK(hsa_iterate_agents(pick_agents, nullptr));
  if (!g_have_gpu || !g_have_cpu) return std::printf("no agents\n"), 1;
  hsa_queue_t* q = nullptr;
  CK(hsa_queue_create(g_gpu, 1024, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
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
  CK(hsa_executable_get_symbol_by_name(exe, "stamp_kernel.kd", &g_gpu, &sym));
  uint64_t kobj = 0;
  uint32_t karg_size = 0, group = 0, priv = 0;
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kobj));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &karg_size));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &group));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &priv));
  std::printf("kernel object %llx, kernarg %u B, LDS %u B, scratch %u B\n", (unsigned long long)kobj, karg_size, group, priv);
  // kernel arguments in device memory, mapped for the host (two slots, alternated)
  CK(hsa_amd_agent_iterate_memory_pools(g_gpu, pick_pool, nullptr));
  if (!g_have_pool) return std::printf("no device pool\n"), 1;
  char* karg = nullptr;
  const size_t slot_bytes = (sizeof(StampArgs) + 256 + 255) & ~(size_t)255;
  CK(hsa_amd_memory_pool_allocate(g_dev_pool, 8 * slot_bytes, 0, (void**)&karg));
  hsa_status_t acc = hsa_amd_agents_allow_access(1, &g_cpu, nullptr, karg);
  const bool dev_kernarg = acc == HSA_STATUS_SUCCESS;
  if (!dev_kernarg) {  // no host window into device memory: fall back to pinned host memory
    std::printf("device kernarg not host-accessible; using pinned host memory\n");
    HK(hipHostMalloc((void**)&karg, 8 * slot_bytes, hipHostMallocMapped));
  }
  // code-object-v5 implicit arguments behind the explicit ones (llvm-readelf --notes: hidden_block_count_[xyz] u32,
  // hidden_group_size_[xyz] u16, hidden_remainder_[xyz] u16, 16 B gap, hidden_global_offset_[xyz] u64, hidden_grid_dims u16):
  // gridDim / blockDim come from HERE, not from the dispatch packet
  struct Hidden {
    uint32_t block_count[3];
    uint16_t group_size[3], remainder[3];
    uint8_t gap[16];
    uint64_t global_offset[3];
    uint16_t grid_dims;
    uint8_t rest[256 - 66];
  } hidden;
  static_assert(sizeof(Hidden) == 256, "implicit argument block");
  std::memset(&hidden, 0, sizeof(hidden));
  hidden.block_count[0] = (uint32_t)blocks, hidden.block_count[1] = hidden.block_count[2] = 1;
  hidden.group_size[0] = 256, hidden.group_size[1] = hidden.group_size[2] = 1;
  hidden.grid_dims = 1;
  if (karg_size > sizeof(StampArgs) + sizeof(Hidden)) return std::printf("unexpected kernarg size\n"), 1;
  auto* base = static_cast<hsa_kernel_dispatch_packet_t*>(q->base_address);
  const uint32_t mask = q->size - 1;
  for (int fence = 0; fence < 2; ++fence) {
    const uint16_t scope = fence ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT;
    const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) | (scope << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                            (scope << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    double t_call = 0, t_total = 0;
    for (int i = 0; i < n + 200; ++i) {
      args.seq = ++seq;
      const auto t0 = std::chrono::steady_clock::now();
      char* ka = karg + (size_t)(i & 7) * slot_bytes;
      std::memcpy(ka, &args, sizeof(args));
      std::memcpy(ka + sizeof(args), &hidden, sizeof(hidden));
      const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
      hsa_kernel_dispatch_packet_t* p = base + (idx & mask);
      p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
      p->workgroup_size_x = 256, p->workgroup_size_y = 1, p->workgroup_size_z = 1;
      p->grid_size_x = (uint32_t)blocks * 256, p->grid_size_y = 1, p->grid_size_z = 1;
      p->private_segment_size = priv, p->group_segment_size = group;
      p->kernel_object = kobj;
      p->kernarg_address = ka;
      p->completion_signal.handle = 0;
      __atomic_store_n(reinterpret_cast<uint32_t*>(p), (uint32_t)header | ((uint32_t)p->setup << 16), __ATOMIC_RELEASE);
      hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
      const auto t1 = std::chrono::steady_clock::now();
      poll(seq);
      const auto t2 = std::chrono::steady_clock::now();
      if (i >= 200) {
        t_call += std::chrono::duration<double>(t1 - t0).count();
        t_total += std::chrono::duration<double>(t2 - t0).count();
      }
    }
    std::printf("AQL  dispatch, %s-scope fences, kernargs in %s memory: submit %.2f us, dispatch -> stamp seen %.2f us\n", fence ? "system" : "agent", dev_kernarg ? "device" : "host",
                1e6 * t_call / n, 1e6 * t_total / n);
  }
  hsa_queue_destroy(q);
  std::printf("done\n");
  return 0;
}
