// gwi_aql.h -- the engine's own AQL dispatch path (host side).
//
// One likelihood evaluation is 2-3 kernel launches whose combined host cost (hipLaunchKernelGGL: 3.2-3.8 us per call on the
// MI355X boxes, and 6.9 us from the call to the first result of a one-workgroup kernel) is a third of the step.  The
// same kernels dispatched as AQL packets written straight into a user-mode HSA queue that belongs to the engine cost
// 0.4 us of host time and 4.3 us to the first result (tools/microbench/aql_dispatch.cpp).  So the plain evaluation path
// -- scan, combine[, final] of ONE hyper-parameter point, nothing else ordered behind it on the HIP stream -- goes this
// way (timed or not: timing mode reads the queue's dispatch timestamps); everything else (batched launches, the sharded
// path with its RCCL exchange on the HIP stream, gwi_log_weights) keeps the HIP stream.  An evaluation is fully drained before its entry point
// returns, so the two queues never hold work of the same engine at the same time.
//
//  * code: the device code compiled a second time into a raw code object (gwi_kernels.hsaco next to the library; the
//    HIP fat binary inside the .so is a bundle the HSA loader does not read), loaded into an HSA executable of ours;
//    kernel symbols are found by the names HIP reports for the host-side function pointers.
//  * kernel arguments: a ring of 4 KiB slots in DEVICE memory that the host writes through the PCIe BAR (arguments in
//    host memory cost the scan 5-7 us of scalar loads over PCIe); an mfence + a read-back of the last byte hand them
//    over before the packet header is published (see dispatch()).  The engine's kernels take no implicit arguments (llvm-readelf --notes: by_value only).
//  * queues: a small process-wide pool (four by default) shared by all engines on the device, see pool_size().
//  * ordering: every packet carries the barrier bit and agent-scope acquire/release fences, i.e. what a HIP stream gives
//    consecutive kernels.  Results reach the host through the kernels' own write-through stores and stamps, as before.
//  * the HSA runtime is bound with dlopen to the instance HIP already loaded (no link-time dependency); any failure while
//    setting up (no host window into device memory, symbol not found, ...) leaves the engine on the HIP path.
//    GWI_AQL=0 disables it.
#ifndef GWI_AQL_H
#define GWI_AQL_H

#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <immintrin.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>
#include <vector>

namespace gwi {
namespace aql {

struct Api {
  decltype(&hsa_init) init = nullptr;
  decltype(&hsa_status_string) status_string = nullptr;
  decltype(&hsa_iterate_agents) iterate_agents = nullptr;
  decltype(&hsa_agent_get_info) agent_get_info = nullptr;
  decltype(&hsa_queue_create) queue_create = nullptr;
  decltype(&hsa_queue_destroy) queue_destroy = nullptr;
  decltype(&hsa_queue_add_write_index_relaxed) add_write_index = nullptr;
  decltype(&hsa_queue_load_read_index_scacquire) load_read_index = nullptr;
  decltype(&hsa_signal_store_screlease) signal_store = nullptr;
  decltype(&hsa_code_object_reader_create_from_memory) reader_create = nullptr;
  decltype(&hsa_executable_create_alt) executable_create = nullptr;
  decltype(&hsa_executable_load_agent_code_object) load_code_object = nullptr;
  decltype(&hsa_executable_freeze) freeze = nullptr;
  decltype(&hsa_executable_get_symbol_by_name) get_symbol = nullptr;
  decltype(&hsa_executable_symbol_get_info) symbol_info = nullptr;
  decltype(&hsa_amd_agent_iterate_memory_pools) iterate_pools = nullptr;
  decltype(&hsa_amd_memory_pool_get_info) pool_info = nullptr;
  decltype(&hsa_amd_memory_pool_allocate) pool_allocate = nullptr;
  decltype(&hsa_amd_memory_pool_free) pool_free = nullptr;
  decltype(&hsa_amd_agents_allow_access) allow_access = nullptr;
  decltype(&hsa_amd_agent_memory_pool_get_info) agent_pool_info = nullptr;
  // kernel begin/end timestamps of a dispatch (timing mode only)
  decltype(&hsa_signal_create) signal_create = nullptr;
  decltype(&hsa_signal_destroy) signal_destroy = nullptr;
  decltype(&hsa_signal_store_relaxed) signal_set = nullptr;
  decltype(&hsa_signal_wait_scacquire) signal_wait = nullptr;
  decltype(&hsa_amd_profiling_set_profiler_enabled) profiling_enable = nullptr;
  decltype(&hsa_amd_profiling_get_dispatch_time) dispatch_time = nullptr;
  decltype(&hsa_system_get_info) system_info = nullptr;
};

struct Kernel {
  uint64_t object = 0;
  uint32_t kernarg_bytes = 0, group_bytes = 0, private_bytes = 0;
};

// A user-mode queue shared by the engines assigned to it (multi-producer; the barrier bit orders ALL its packets, so
// engines on one queue take turns -- exactly what HIP streams that share a hardware queue do).
struct SharedQueue {
  hsa_queue_t* q = nullptr;
  int users = 0;  // engines currently assigned (under Device::mu)
  volatile bool failed = false;
  bool profiling = false;  // dispatch timestamps are on (set when the queue is created)
  std::string why;
};

// one per (process, device): agents, the device-memory pool, the loaded executable, the queue pool
struct Device {
  bool ok = false;
  std::string why;
  hsa_agent_t gpu{}, cpu{};
  hsa_amd_memory_pool_t pool{};
  hsa_executable_t exe{};
  std::vector<char> blob;  // the code object must outlive the executable
  std::mutex mu;
  std::vector<SharedQueue*> queues;  // created together on first use, never destroyed (process lifetime)
  unsigned next_engine = 0;
  volatile uint32_t* hdp_mem_flush = nullptr;  // HSA_AMD_AGENT_INFO_HDP_FLUSH: the device's host-data-path flush register (user-mode mapping), if exposed
};

enum class Handoff { kReadback, kHdp, kNone };  // see settle()
inline Handoff handoff_mode_from_env();

// one per engine: its queue (shared) and its own ring of kernel-argument slots
struct Queue {
  Device* dev = nullptr;
  Handoff handoff = Handoff::kReadback;
  SharedQueue* sq = nullptr;
  char* kernarg = nullptr;  // ring of kSlots x kSlotBytes in device memory, host-writable; followed by kExtraBytes of host-writable device memory for the engine (theta blocks of batched launches)
  unsigned next_slot = 0;
  std::vector<char> tail_shadow[2];  // dispatch_tail: the fixed head last written into each of its two slots
  hsa_signal_t done[3] = {{0}, {0}, {0}};  // completion signals of the scan / combine / final packets of a TIMED evaluation
  bool have_signals = false;
  bool failed() const { return sq && sq->failed; }
  const std::string& why() const { return sq->why; }
};
constexpr unsigned kSlots = 16, kSlotBytes = 4352;  // the scan's segment: 56 bytes of preloaded scalars + the 4 KiB argument block (a multiple of 256)
constexpr size_t kExtraBytes = 256 * 1024;
inline char* extra_area(const Queue& q) { return q.kernarg ? q.kernarg + (size_t)kSlots * kSlotBytes : nullptr; }

inline Api& api() {
  static Api a;
  return a;
}

inline bool bind_api(std::string& why) {
  static std::once_flag once;
  static bool ok = false;
  static std::string err;
  std::call_once(once, [] {
    void* lib = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_NOLOAD);  // the instance HIP runs on
    if (!lib) lib = dlopen("libhsa-runtime64.so.1", RTLD_NOW);
    if (!lib) {
      err = "libhsa-runtime64.so.1 not loadable";
      return;
    }
    Api& a = api();
    bool all = true;
#define GWI_AQL_SYM(member, name)                                   \
  a.member = reinterpret_cast<decltype(a.member)>(dlsym(lib, name)); \
  all = all && a.member != nullptr;
    GWI_AQL_SYM(init, "hsa_init")
    GWI_AQL_SYM(status_string, "hsa_status_string")
    GWI_AQL_SYM(iterate_agents, "hsa_iterate_agents")
    GWI_AQL_SYM(agent_get_info, "hsa_agent_get_info")
    GWI_AQL_SYM(queue_create, "hsa_queue_create")
    GWI_AQL_SYM(queue_destroy, "hsa_queue_destroy")
    GWI_AQL_SYM(add_write_index, "hsa_queue_add_write_index_relaxed")
    GWI_AQL_SYM(load_read_index, "hsa_queue_load_read_index_scacquire")
    GWI_AQL_SYM(signal_store, "hsa_signal_store_screlease")
    GWI_AQL_SYM(reader_create, "hsa_code_object_reader_create_from_memory")
    GWI_AQL_SYM(executable_create, "hsa_executable_create_alt")
    GWI_AQL_SYM(load_code_object, "hsa_executable_load_agent_code_object")
    GWI_AQL_SYM(freeze, "hsa_executable_freeze")
    GWI_AQL_SYM(get_symbol, "hsa_executable_get_symbol_by_name")
    GWI_AQL_SYM(symbol_info, "hsa_executable_symbol_get_info")
    GWI_AQL_SYM(iterate_pools, "hsa_amd_agent_iterate_memory_pools")
    GWI_AQL_SYM(pool_info, "hsa_amd_memory_pool_get_info")
    GWI_AQL_SYM(pool_allocate, "hsa_amd_memory_pool_allocate")
    GWI_AQL_SYM(pool_free, "hsa_amd_memory_pool_free")
    GWI_AQL_SYM(allow_access, "hsa_amd_agents_allow_access")
    GWI_AQL_SYM(agent_pool_info, "hsa_amd_agent_memory_pool_get_info")
    GWI_AQL_SYM(signal_create, "hsa_signal_create")
    GWI_AQL_SYM(signal_destroy, "hsa_signal_destroy")
    GWI_AQL_SYM(signal_set, "hsa_signal_store_relaxed")
    GWI_AQL_SYM(signal_wait, "hsa_signal_wait_scacquire")
    GWI_AQL_SYM(profiling_enable, "hsa_amd_profiling_set_profiler_enabled")
    GWI_AQL_SYM(dispatch_time, "hsa_amd_profiling_get_dispatch_time")
    GWI_AQL_SYM(system_info, "hsa_system_get_info")
#undef GWI_AQL_SYM
    if (!all) {
      err = "HSA runtime lacks an entry point";
      return;
    }
    if (a.init() != HSA_STATUS_SUCCESS) {
      err = "hsa_init failed";
      return;
    }
    ok = true;
  });
  if (!ok) why = err;
  return ok;
}

inline std::string status_text(hsa_status_t s) {
  const char* m = nullptr;
  if (api().status_string && api().status_string(s, &m) == HSA_STATUS_SUCCESS && m) return m;
  return "HSA status " + std::to_string((int)s);
}

struct AgentSearch {
  uint32_t want_bdf, want_domain;
  hsa_agent_t gpu{}, cpu{};
  bool have_gpu = false, have_cpu = false;
};
inline hsa_status_t visit_agent(hsa_agent_t a, void* data) {
  auto* s = static_cast<AgentSearch*>(data);
  hsa_device_type_t t;
  if (api().agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
  if (t == HSA_DEVICE_TYPE_CPU && !s->have_cpu) s->cpu = a, s->have_cpu = true;
  if (t == HSA_DEVICE_TYPE_GPU && !s->have_gpu) {
    uint32_t bdf = 0, domain = 0;
    api().agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
    api().agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
    if (bdf == s->want_bdf && domain == s->want_domain) s->gpu = a, s->have_gpu = true;
  }
  return HSA_STATUS_SUCCESS;
}
struct PoolSearch {
  hsa_amd_memory_pool_t pool{};
  bool have = false;
};
inline hsa_status_t visit_pool(hsa_amd_memory_pool_t p, void* data) {
  auto* s = static_cast<PoolSearch*>(data);
  hsa_amd_segment_t seg;
  uint32_t flags = 0;
  bool alloc = false;
  api().pool_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  api().pool_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
  api().pool_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (!s->have && seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED)) s->pool = p, s->have = true;
  return HSA_STATUS_SUCCESS;
}

// Agents and executable for the GPU at PCI (domain, bus, device, function); `code_path` = the raw code object.
inline Device* open_device(uint32_t domain, uint32_t bus, uint32_t device, uint32_t function, const std::string& code_path) {
  static std::mutex mu;
  static std::vector<std::pair<uint64_t, Device*>> cache;
  std::lock_guard<std::mutex> lock(mu);
  const uint32_t bdf = (bus << 8) | (device << 3) | function;
  const uint64_t key = ((uint64_t)domain << 32) | bdf;
  for (auto& kv : cache)
    if (kv.first == key) return kv.second;
  Device* d = new Device;
  cache.emplace_back(key, d);
  if (!bind_api(d->why)) return d;
  Api& a = api();
  AgentSearch as{bdf, domain};
  a.iterate_agents(visit_agent, &as);
  if (!as.have_gpu || !as.have_cpu) {
    d->why = "no HSA agent at the PCI address of the HIP device";
    return d;
  }
  d->gpu = as.gpu, d->cpu = as.cpu;
  {
    hsa_amd_hdp_flush_t hdp{nullptr, nullptr};
    if (a.agent_get_info(d->gpu, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_HDP_FLUSH, &hdp) == HSA_STATUS_SUCCESS) d->hdp_mem_flush = hdp.HDP_MEM_FLUSH_CNTL;
  }
  PoolSearch ps;
  a.iterate_pools(d->gpu, visit_pool, &ps);
  if (!ps.have) {
    d->why = "no coarse-grained device memory pool";
    return d;
  }
  d->pool = ps.pool;
  // the host must be able to map that pool (a PCIe BAR covering device memory): ask before ever touching such a pointer
  hsa_amd_memory_pool_access_t access = HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;
  if (a.agent_pool_info(d->cpu, d->pool, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &access) != HSA_STATUS_SUCCESS || access == HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED) {
    d->why = "device memory is not host-accessible on this system (no large BAR): kernel arguments cannot live there";
    return d;
  }
  FILE* f = std::fopen(code_path.c_str(), "rb");
  if (!f) {
    d->why = code_path + " not found (built by __graft_entry__.build())";
    return d;
  }
  std::fseek(f, 0, SEEK_END);
  d->blob.resize((size_t)std::ftell(f));
  std::fseek(f, 0, SEEK_SET);
  const bool read_ok = std::fread(d->blob.data(), 1, d->blob.size(), f) == d->blob.size();
  std::fclose(f);
  if (!read_ok || d->blob.size() < 64 || std::memcmp(d->blob.data(), "\177ELF", 4) != 0) {
    d->why = code_path + " is not a raw code object";
    return d;
  }
  hsa_code_object_reader_t reader;
  hsa_status_t st = a.reader_create(d->blob.data(), d->blob.size(), &reader);
  if (st == HSA_STATUS_SUCCESS) st = a.executable_create(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &d->exe);
  if (st == HSA_STATUS_SUCCESS) st = a.load_code_object(d->exe, d->gpu, reader, nullptr, nullptr);
  if (st == HSA_STATUS_SUCCESS) st = a.freeze(d->exe, nullptr);
  if (st != HSA_STATUS_SUCCESS) {
    d->why = "loading " + code_path + ": " + status_text(st);
    return d;
  }
  d->ok = true;
  return d;
}

// A second code object on the same device: a scan chain compiled at gwi_create (gwi_jit.h).  The blob must outlive the
// executable (the caller keeps it: jit::Chain::code); executables live as long as the process.
inline bool load_code(Device* d, const void* blob, size_t bytes, hsa_executable_t& exe, std::string& why) {
  Api& a = api();
  hsa_code_object_reader_t reader;
  hsa_status_t st = a.reader_create(blob, bytes, &reader);
  if (st == HSA_STATUS_SUCCESS) st = a.executable_create(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe);
  if (st == HSA_STATUS_SUCCESS) st = a.load_code_object(exe, d->gpu, reader, nullptr, nullptr);
  if (st == HSA_STATUS_SUCCESS) st = a.freeze(exe, nullptr);
  if (st != HSA_STATUS_SUCCESS) {
    why = "loading a run-time compiled chain into the HSA executable: " + status_text(st);
    return false;
  }
  return true;
}

inline bool find_kernel(Device* d, const char* mangled_name, Kernel& out, std::string& why, const hsa_executable_t* in_exe = nullptr) {
  if (!mangled_name) {
    why = "HIP did not report a kernel name";
    return false;
  }
  Api& a = api();
  const std::string sym_name = std::string(mangled_name) + ".kd";
  hsa_executable_symbol_t sym;
  hsa_status_t st = a.get_symbol(in_exe ? *in_exe : d->exe, sym_name.c_str(), &d->gpu, &sym);
  if (st == HSA_STATUS_SUCCESS) st = a.symbol_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &out.object);
  if (st == HSA_STATUS_SUCCESS) st = a.symbol_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &out.kernarg_bytes);
  if (st == HSA_STATUS_SUCCESS) st = a.symbol_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &out.group_bytes);
  if (st == HSA_STATUS_SUCCESS) st = a.symbol_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &out.private_bytes);
  if (st != HSA_STATUS_SUCCESS) {
    why = sym_name + ": " + status_text(st);
    return false;
  }
  if (out.private_bytes != 0) {  // the engine's kernels use no scratch; a build that does stays on the HIP path
    why = sym_name + " needs scratch memory";
    return false;
  }
  return true;
}

inline void queue_error(hsa_status_t status, hsa_queue_t*, void* data) {
  auto* q = static_cast<SharedQueue*>(data);
  q->why = "HSA queue error: " + status_text(status);
  q->failed = true;
}

// Queue pool.  One queue per engine loses badly as soon as three or more are ACTIVE at once (config 2, blocking chains in
// host threads: 1 chain 63 k evals/s, 2 chains 120 k, 3 chains 52 k, 8 chains 64 k): queues created at different times
// end up sharing hardware pipes and keep evicting each other.  So the engines of a process share a small pool created in
// one go -- as HIP shares its (default four) hardware queues among streams.  Measured with the pool (1 / 2 / 3 / 4 / 6 / 8
// chains, k evals/s): pool of 1: 62 81 77 75 76 76; 2: 61 108 112 105 146 146; 3: 55 116 153 134 207 189;
// 4: 61 103 154 138 188 203; HIP streams: 49 80 118 122 185 173.  GWI_AQL_QUEUES overrides the default of 4.
inline unsigned pool_size() {
  unsigned n = 4;
  if (const char* env = std::getenv("GWI_AQL_QUEUES")) n = (unsigned)std::atoi(env);
  return n < 1 ? 1 : (n > 8 ? 8 : n);
}

inline bool open_queue(Device* d, Queue& out, std::string& why) {
  Api& a = api();
  out.dev = d;
  out.handoff = handoff_mode_from_env();
  {
    std::lock_guard<std::mutex> lock(d->mu);
    if (d->queues.empty()) {
      const unsigned n = pool_size();
      for (unsigned i = 0; i < n; ++i) {
        auto* sq = new SharedQueue;
        const hsa_status_t st = a.queue_create(d->gpu, 1024, HSA_QUEUE_TYPE_MULTI, queue_error, sq, UINT32_MAX, UINT32_MAX, &sq->q);
        if (st != HSA_STATUS_SUCCESS) {
          why = "hsa_queue_create: " + status_text(st);
          delete sq;
          break;
        }
        // Dispatch timestamps on from the start: the packet processor picks the property up when the queue is first
        // mapped; switched on later it was honoured in some processes and not in others (timestamps never written).
        // It costs nothing for packets without a completion signal, which is all of them outside timing mode.
        const char* ts = std::getenv("GWI_AQL_TIMESTAMPS");  // 0: leave them off (timed evaluations then use HIP events on the HIP stream)
        sq->profiling = !(ts && std::atoi(ts) == 0) && a.profiling_enable(sq->q, 1) == HSA_STATUS_SUCCESS;
        d->queues.push_back(sq);
      }
    }
    if (d->queues.empty()) return false;
    // the queue with the fewest engines on it (engines come and go: plain round robin would double up two live engines
    // on one queue -- whose barrier bits serialise them -- while another queue sits empty)
    out.sq = d->queues[0];
    for (SharedQueue* sq : d->queues)
      if (sq->users < out.sq->users) out.sq = sq;
    ++out.sq->users;
  }
  hsa_status_t st = a.pool_allocate(d->pool, (size_t)kSlots * kSlotBytes + kExtraBytes, 0, reinterpret_cast<void**>(&out.kernarg));
  if (st == HSA_STATUS_SUCCESS) st = a.allow_access(1, &d->cpu, nullptr, out.kernarg);  // needs a host window into device memory (large BAR)
  if (st != HSA_STATUS_SUCCESS) {
    why = "kernel-argument ring in device memory: " + status_text(st);
    if (out.kernarg) a.pool_free(out.kernarg);
    out.kernarg = nullptr;
    {
      std::lock_guard<std::mutex> lock(d->mu);
      --out.sq->users;
    }
    out.sq = nullptr;
    return false;
  }
  std::memset(out.kernarg, 0, (size_t)kSlots * kSlotBytes + kExtraBytes);
  out.have_signals = true;
  for (auto& sg : out.done)
    if (a.signal_create(1, 0, nullptr, &sg) != HSA_STATUS_SUCCESS) out.have_signals = false;
  return true;
}

// Timing mode: kernel begin / end of the three dispatches as the packet processor stamps them (the quantity rocprofv3's
// kernel trace reports), in milliseconds.  `n` = 2 or 3 packets carried signals.
inline bool timed_prepare(Queue& q) {
  if (!q.have_signals || !q.sq) return false;
  Api& a = api();
  if (!q.sq->profiling) return false;
  for (auto& sg : q.done) a.signal_set(sg, 1);
  return true;
}
inline bool timed_collect(Queue& q, int n, float* ms) {
  Api& a = api();
  uint64_t hz = 0;
  if (a.system_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &hz) != HSA_STATUS_SUCCESS || hz == 0) return false;
  for (int i = 0; i < n; ++i) {
    if (a.signal_wait(q.done[i], HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) >= 1) return false;
    hsa_amd_profiling_dispatch_time_t t{};
    if (a.dispatch_time(q.dev->gpu, q.done[i], &t) != HSA_STATUS_SUCCESS) return false;
    if (t.end <= t.start) return false;  // never stamped
    ms[i] = (float)(1e3 * (double)(t.end - t.start) / (double)hz);
  }
  return true;
}

// An engine whose evaluation timed out with work possibly in flight gives its place on the shared queue back without
// freeing anything a kernel may still touch (the kernel-argument allocation stays).
inline void abandon_queue(Queue& q) {
  if (q.sq && q.dev) {
    std::lock_guard<std::mutex> lock(q.dev->mu);
    --q.sq->users;
  }
  q.sq = nullptr;
}

inline void close_queue(Queue& q) {
  if (q.have_signals)
    for (auto& sg : q.done)
      if (sg.handle) api().signal_destroy(sg);
  q.have_signals = false;
  if (q.kernarg) api().pool_free(q.kernarg);
  q.kernarg = nullptr;
  if (q.sq && q.dev) {
    std::lock_guard<std::mutex> lock(q.dev->mu);
    --q.sq->users;
  }
  q.sq = nullptr;  // the shared queues live as long as the process
}

// One kernel dispatch: arguments -> next ring slot, packet -> queue, doorbell.  Returns false (nothing submitted) when the
// queue has reported an error.
//
// Order (the queue is multi-producer: once this packet's header turns valid, ANOTHER engine's doorbell may already cover
// its index, so everything the packet refers to must have landed before the header is published):
//   1. arguments into the ring slot (write-combined stores through the PCIe BAR);
//   2. mfence (drain the write-combining buffers; mfence, not sfence: the read that follows is a load), then READ BACK the last argument byte through the BAR: a non-posted
//      read cannot pass the posted writes ahead of it, and it forces the device's host data path to retire them to
//      memory -- the same hand-off HIP's own device-kernarg path performs before it rings (read-back or HDP flush);
//   3. packet body, then the header with release order;
//   4. doorbell.
// GWI_AQL_READBACK=0 drops the read of step 2 (A/B timing only: the hand-off then rests on the fence + in-order posted writes).
inline bool readback_enabled() {
  static const bool on = [] {
    const char* e = std::getenv("GWI_AQL_READBACK");
    return !(e && std::atoi(e) == 0);
  }();
  return on;
}
// How bytes written through the BAR are handed over before a packet that reads them is published (GWI_AQL_HANDOFF):
//   (all modes start with an mfence: the write-combining buffers are drained before anything below executes)
//   readback (default): read the last byte written back through the BAR -- a non-posted read cannot pass the posted writes
//                       ahead of it, and its completion means the device's host data path has retired them;
//   hdp:                write the device's HDP flush register and read it back (what ROCclr's device-kernarg path does);
//   none:               the fence only (A/B timing; GWI_AQL_READBACK=0 is the older spelling).
inline Handoff handoff_mode_from_env() {  // read when an engine opens its queue
  if (!readback_enabled()) return Handoff::kNone;
  const char* e = std::getenv("GWI_AQL_HANDOFF");
  if (e && std::strcmp(e, "hdp") == 0) return Handoff::kHdp;
  if (e && std::strcmp(e, "none") == 0) return Handoff::kNone;
  return Handoff::kReadback;
}
// after the fence: true unless the read-back saw a byte other than `expect`
inline bool settle(const Device* dev, Handoff m, const char* last_byte_written, unsigned char expect) {
  if (m == Handoff::kHdp && dev && dev->hdp_mem_flush) {
    *dev->hdp_mem_flush = 1u;
    (void)*dev->hdp_mem_flush;
    return true;
  }
  if (m == Handoff::kNone) return true;
  return *reinterpret_cast<const volatile unsigned char*>(last_byte_written) == expect;
}

// steps 1-2: copy an argument block into ring slot `slot` and hand it over to the device; nullptr on failure
inline char* stage_args(Queue& q, unsigned slot, const void* args, size_t arg_bytes) {
  if (!q.sq || q.sq->failed || arg_bytes > kSlotBytes || arg_bytes == 0 || slot >= kSlots) return nullptr;
  char* ka = q.kernarg + (size_t)slot * kSlotBytes;
  std::memcpy(ka, args, arg_bytes);
  _mm_mfence();  // not sfence: the read-back below is a LOAD, which sfence does not order behind the write-combined stores
  if (!settle(q.dev, q.handoff, ka + arg_bytes - 1, static_cast<const unsigned char*>(args)[arg_bytes - 1])) {  // cannot happen on a coherent BAR mapping; refuse to launch on stale arguments
    q.sq->why = "kernel-argument read-back through the BAR returned a stale byte";
    q.sq->failed = true;
    return nullptr;
  }
  return ka;
}

// hand over bytes the caller wrote itself into host-writable device memory (the extra area): drain + read back the last byte
inline void handoff(const Queue& q, const char* last_byte_written) {
  _mm_mfence();  // not sfence: the read-back below is a LOAD, which sfence does not order behind the write-combined stores
  (void)settle(q.dev, q.handoff, last_byte_written, 0);
}

// steps 3-4 for an argument block that is already in place (stage_args)
// `acquire`: whether the packet carries an agent-scope acquire fence (cache invalidate before the kernel starts).  A kernel
// that reads everything the launch before it wrote with cache-bypassing loads (combine_group) goes without.
inline bool dispatch_staged(Queue& q, const Kernel& k, char* ka, uint32_t grid_x_blocks, uint32_t grid_y_blocks, uint32_t block_threads, uint32_t dynamic_lds,
                            hsa_signal_t completion = hsa_signal_t{0}, bool acquire = true) {
  if (!q.sq || q.sq->failed || !ka) return false;
  Api& a = api();
  hsa_queue_t* hq = q.sq->q;
  const uint64_t idx = a.add_write_index(hq, 1);  // atomic: several engines (host threads) may produce into one queue
  while (idx - a.load_read_index(hq) >= hq->size) _mm_pause();
  auto* p = static_cast<hsa_kernel_dispatch_packet_t*>(hq->base_address) + (idx & (hq->size - 1));
  const uint16_t setup = (uint16_t)((grid_y_blocks > 1 ? 2 : 1) << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS);
  p->workgroup_size_x = (uint16_t)block_threads, p->workgroup_size_y = 1, p->workgroup_size_z = 1;
  p->reserved0 = 0;
  p->grid_size_x = grid_x_blocks * block_threads, p->grid_size_y = grid_y_blocks, p->grid_size_z = 1;
  p->private_segment_size = 0;
  p->group_segment_size = k.group_bytes + dynamic_lds;
  p->kernel_object = k.object;
  p->kernarg_address = ka;
  p->reserved2 = 0;
  p->completion_signal = completion;
  const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                          ((acquire ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE) << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                          (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
  __atomic_store_n(reinterpret_cast<uint32_t*>(p), (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
  a.signal_store(hq->doorbell_signal, (hsa_signal_value_t)idx);
  return true;
}

// the rotating slots hold per-evaluation argument blocks; the last kPersistentSlots are written once (gwi_create) for
// launches whose arguments never change -- those dispatches need no per-evaluation hand-off at all
constexpr unsigned kPersistentSlots = 4;
constexpr unsigned kTailSlots = 2;  // dispatch_tail's two slots sit below the persistent ones
constexpr unsigned kRingSlots = kSlots - kPersistentSlots - kTailSlots;
inline bool dispatch(Queue& q, const Kernel& k, const void* args, size_t arg_bytes, uint32_t grid_x_blocks, uint32_t grid_y_blocks, uint32_t block_threads, uint32_t dynamic_lds,
                     hsa_signal_t completion = hsa_signal_t{0}) {
  if (arg_bytes > k.kernarg_bytes + 0u) return false;
  char* ka = stage_args(q, q.next_slot++ % kRingSlots, args, arg_bytes);
  return dispatch_staged(q, k, ka, grid_x_blocks, grid_y_blocks, block_threads, dynamic_lds, completion);
}

// An argument block whose first `head_bytes` never change between launches (pointers, sizes, descriptors) and whose tail
// does (the hyper-parameters): the block keeps its place in one of two slots (`parity` alternates, so the block of the
// launch before is never the one being rewritten) and only the `n_ranges` byte ranges of the tail travel through the BAR --
// a few hundred bytes instead of 3.5 KB per evaluation (the BAR write of the whole block was 1.2 us of host time on the
// critical path).  The head is compared with what the slot holds (a host-side copy) and rewritten whole when it differs.
// Same hand-off as stage_args: mfence, read-back of the last byte written.
inline bool dispatch_tail(Queue& q, const Kernel& k, unsigned parity, const void* args, size_t head_bytes, size_t arg_bytes, const size_t (*ranges)[2], int n_ranges,
                          uint32_t grid_x_blocks, uint32_t grid_y_blocks, uint32_t block_threads, uint32_t dynamic_lds, hsa_signal_t completion = hsa_signal_t{0}) {
  if (!q.sq || q.sq->failed || arg_bytes > kSlotBytes || arg_bytes > k.kernarg_bytes + 0u || head_bytes > arg_bytes || n_ranges < 1) return false;
  parity &= 1u;
  char* ka = q.kernarg + (size_t)(kRingSlots + parity) * kSlotBytes;
  const char* src = static_cast<const char*>(args);
  std::vector<char>& shadow = q.tail_shadow[parity];
  size_t last;
  if (shadow.size() != head_bytes || std::memcmp(shadow.data(), src, head_bytes) != 0) {
    std::memcpy(ka, src, arg_bytes);
    shadow.assign(src, src + head_bytes);
    last = arg_bytes - 1;
  } else {
    last = 0;
    for (int r = 0; r < n_ranges; ++r) {
      if (ranges[r][1] == 0) continue;
      if (ranges[r][0] < head_bytes || ranges[r][0] + ranges[r][1] > arg_bytes) return false;
      std::memcpy(ka + ranges[r][0], src + ranges[r][0], ranges[r][1]);
      if (ranges[r][0] + ranges[r][1] - 1 > last) last = ranges[r][0] + ranges[r][1] - 1;
    }
  }
  _mm_mfence();  // not sfence: the read-back below is a LOAD, which sfence does not order behind the write-combined stores
  if (!settle(q.dev, q.handoff, ka + last, static_cast<const unsigned char*>(args)[last])) {
    q.sq->why = "kernel-argument read-back through the BAR returned a stale byte";
    q.sq->failed = true;
    return false;
  }
  return dispatch_staged(q, k, ka, grid_x_blocks, grid_y_blocks, block_threads, dynamic_lds, completion);
}

// All packets of this process's queues retired?  (gwi_destroy after a timed-out evaluation must not free buffers a
// kernel may still write.)  Bounded wait; returns whether the queue drained.
inline bool drain(Queue& q, double seconds) {
  if (!q.sq || !q.sq->q) return true;
  Api& a = api();
  hsa_queue_t* hq = q.sq->q;
  const uint64_t target = a.add_write_index(hq, 0);
  for (uint64_t spin = 0; spin < (uint64_t)(seconds * 2.0e7); ++spin) {
    if (a.load_read_index(hq) >= target) return true;
    _mm_pause();
  }
  return false;
}

}  // namespace aql
}  // namespace gwi
#endif  // GWI_AQL_H
