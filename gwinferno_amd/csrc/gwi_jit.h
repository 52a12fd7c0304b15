// gwi_jit.h -- a compiled scan chain for ANY product of densities, built when the engine first meets it.
//
// The reference is a toolkit: a user's model function multiplies whatever densities they pick
// (tests/inference_test.py:256-260, gwinferno/models/bsplines/separable.py:295-778, examples/simple_bspline_example.py:58-71).
// The scan kernel is a compile-time chain of terms; the library ships instantiations for the BASELINE configurations and the
// reference's own models (kVariants, gwi_engine.hip), and every other sequence used to run the generic kernel (run-time term
// loop, several times slower).  Here such a sequence gets its own instantiation of the SAME template at gwi_create:
//
//   * source: three lines -- `#include "gwi_device.h"` + one name expression per kernel role -- compiled by hipRTC for gfx950
//     with the flags of the ahead-of-time build (the kernel-argument preload included).  The two headers are embedded in the
//     library when it is built (.incbin, gwi_engine.hip): the chain is this very build's template, whatever has happened to
//     the source tree since.  The COMPILER is whichever hipRTC the process has: a Python process that imported PyTorch first
//     (gwinferno_amd/_native.py does, so that the engine and torch.distributed share one HIP runtime) gets the libhiprtc /
//     libamd_comgr bundled with PyTorch's ROCm, not the one hipcc of the build belongs to -- same template and flags, a
//     neighbouring compiler version: config 2's chain comes out with 99 vector registers instead of 83 and the same scan
//     time (profiles/round5); the cache key carries the library's path and version;
//   * cache: the raw code object + the kernels' lowered names in one file under $GWI_JIT_CACHE (default
//     $XDG_CACHE_HOME/gwinferno_amd or ~/.cache/gwinferno_amd, else /tmp/gwinferno_amd-<uid>), keyed by the kind sequence, the
//     samples per lane and a hash of (headers, flags, hipRTC version); written to a temporary name and renamed;
//   * use: hipModuleLoadData for launches on the HIP stream, a second HSA executable for the engine's AQL queue (gwi_aql.h);
//   * hipRTC is bound with dlopen (no link-time dependency).  Where it is missing or the compilation fails the engine says
//     why once and runs the generic kernel; GWI_JIT=0 switches the whole mechanism off.
//
// The batched matrix-core kernel of a spline model (gwi_mfma.h: scan_mfma_kernel, keyed by kinds AND 16-basis tile counts) is
// compiled the same way where no ahead-of-time instantiation fits (get_chain with U = 0; gwi_engine.hip: try_jit_mfma).
//
// Compilation needs no GPU (hipRTC cross-compiles like hipcc): gwi_jit_compile() is part of the CPU test-suite.
#ifndef GWI_JIT_H
#define GWI_JIT_H

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <cctype>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "gwi_engine.h"

namespace gwi {
constexpr int pbatch_u(int chain_u);  // gwi_device.h
namespace jit {

// kernel roles of one term sequence (the template arguments <WRITE_LOGW, BATCH, SAFE> of scan_kernel, and scan_pbatch_kernel)
enum Role { kScan = 0, kLogw = 1, kBatch = 2, kSafe = 3, kPbatch = 4, kRoles = 5 };

struct Chain {
  std::string name;  // "jit:2,3,6,8/u2"
  int n = 0;
  int kinds[GWI_MAX_TERMS] = {0};
  int samples_per_lane = 2;
  bool spline = false;           // the sequence has a spline term: a SAFE instantiation exists, a pbatch one does not
  std::vector<char> code;        // raw gfx950 code object
  std::string lowered[kRoles];   // mangled kernel names ("" where the role has no instantiation)
  std::string path;              // the cache file ("" when the cache directory is not writable)
  double compile_seconds = 0.0;  // hipRTC time of THIS process (0 when the code object came from the disk cache)
  bool from_cache = false;
  // per device: the loaded module (HIP-stream launches) and the HSA executable of the engine's AQL queue (handle kept as a
  // plain integer: this header does not depend on gwi_aql.h)
  std::mutex mu;
  std::vector<std::pair<int, hipModule_t>> modules;
  std::vector<std::pair<const void*, unsigned long long>> hsa_executables;
};

struct Rtc {
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
  decltype(&hiprtcAddNameExpression) add_name = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetLoweredName) lowered = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcVersion) version = nullptr;
  bool ok = false;
  std::string why;
};

inline Rtc& rtc() {
  static Rtc r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"};
    void* lib = nullptr;
    if (const char* env = std::getenv("GWI_HIPRTC_LIB")) lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    for (const char* nm : names)
      if (!lib) lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (!lib) {
      r.why = std::string("libhiprtc.so is not loadable (") + (dlerror() ? dlerror() : "dlopen failed") + ")";
      return;
    }
    bool all = true;
#define GWI_RTC_SYM(member, sym)                                     \
  r.member = reinterpret_cast<decltype(r.member)>(dlsym(lib, sym)); \
  all = all && r.member != nullptr;
    GWI_RTC_SYM(create, "hiprtcCreateProgram")
    GWI_RTC_SYM(destroy, "hiprtcDestroyProgram")
    GWI_RTC_SYM(add_name, "hiprtcAddNameExpression")
    GWI_RTC_SYM(compile, "hiprtcCompileProgram")
    GWI_RTC_SYM(log_size, "hiprtcGetProgramLogSize")
    GWI_RTC_SYM(log, "hiprtcGetProgramLog")
    GWI_RTC_SYM(lowered, "hiprtcGetLoweredName")
    GWI_RTC_SYM(code_size, "hiprtcGetCodeSize")
    GWI_RTC_SYM(code, "hiprtcGetCode")
    GWI_RTC_SYM(version, "hiprtcVersion")
#undef GWI_RTC_SYM
    if (!all) {
      r.why = "libhiprtc.so lacks an entry point";
      return;
    }
    r.ok = true;
  });
  return r;
}

inline bool is_spline_kind(int k) { return k == GWI_TERM_EXP_SPLINE || k == GWI_TERM_LINEAR_SPLINE || k == GWI_TERM_EXP_SPLINE_LERP; }

// The flags of the ahead-of-time build (__graft_entry__.build): a chain compiled here is the chain hipcc would have built.
inline const std::vector<const char*>& flags() {
  // (the two inliner switches are what the hipcc driver's device pipeline does by default and hipRTC's does not: with them the
  // instruction streams agree; without, config 2's chain came out 2 % longer and measured 8 % slower)
  static const std::vector<const char*> f = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-mllvm", "-amdgpu-kernarg-preload-count=16",
                                             "-mllvm", "-amdgpu-early-inline-all=true", "-mllvm", "-amdgpu-function-calls=false"};
  return f;
}

inline unsigned long long fnv1a(const void* data, size_t n, unsigned long long h = 1469598103934665603ull) {
  const unsigned char* p = static_cast<const unsigned char*>(data);
  for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
  return h;
}

// A cache directory is trusted only if it is OURS: a real directory (not a link), owned by this user, writable by nobody else.
// What it holds is loaded onto the GPU inside this process, so a directory someone else could have prepared -- the predictable
// /tmp fallback above all -- is skipped, and the caller goes on to the next candidate or compiles without a cache.
inline bool trusted_dir(const std::string& path, std::string* why = nullptr) {
  struct stat st;
  if (lstat(path.c_str(), &st) != 0) {
    if (why) *why = path + ": " + std::strerror(errno);
    return false;
  }
  if (!S_ISDIR(st.st_mode)) {
    if (why) *why = path + " is not a directory (a symbolic link is not followed)";
    return false;
  }
  if (st.st_uid != geteuid()) {
    if (why) *why = path + " belongs to another user";
    return false;
  }
  if (st.st_mode & (S_IWGRP | S_IWOTH)) {
    if (why) *why = path + " is writable by group or others";
    return false;
  }
  return access(path.c_str(), W_OK | X_OK) == 0;
}

inline bool make_dirs(const std::string& path, std::string* why = nullptr) {
  for (size_t i = 1; i <= path.size(); ++i)
    if (i == path.size() || path[i] == '/') {
      const std::string sub = path.substr(0, i);
      if (mkdir(sub.c_str(), 0700) != 0 && errno != EEXIST) {
        if (why) *why = sub + ": " + std::strerror(errno);
        return false;
      }
    }
  return trusted_dir(path, why);
}

// where compiled chains are kept; "" when no candidate is both writable and trusted (every process then compiles for itself,
// and says so once)
inline std::string cache_dir() {
  std::vector<std::string> cands;
  if (const char* e = std::getenv("GWI_JIT_CACHE")) {
    if (!*e || std::strcmp(e, "0") == 0 || std::strcmp(e, "off") == 0) return "";
    cands.push_back(e);
  } else {
    if (const char* x = std::getenv("XDG_CACHE_HOME"))
      if (*x) cands.push_back(std::string(x) + "/gwinferno_amd");
    if (const char* hm = std::getenv("HOME"))
      if (*hm) cands.push_back(std::string(hm) + "/.cache/gwinferno_amd");
    cands.push_back("/tmp/gwinferno_amd-" + std::to_string((long)getuid()));
  }
  std::string refused;
  for (const std::string& c : cands) {
    std::string why;
    if (make_dirs(c, &why)) return c;
    refused += (refused.empty() ? "" : "; ") + why;
  }
  static std::once_flag said;
  std::call_once(said, [&] {
    if (!std::getenv("GWI_QUIET"))
      std::fprintf(stderr, "gwinferno_amd: no trusted cache directory for run-time compiled kernels (%s): compiling in this process only\n", refused.c_str());
  });
  return "";
}

// two independent 64-bit FNV-1a passes over everything a cache file holds besides the digest itself
inline void digest_of(const Chain& c, unsigned long long out[2]) {
  const unsigned long long seeds[2] = {1469598103934665603ull, 0x9e3779b97f4a7c15ull};
  for (int k = 0; k < 2; ++k) {
    unsigned long long h = fnv1a(&c.n, sizeof(c.n), seeds[k]);
    h = fnv1a(c.kinds, sizeof(int) * (size_t)c.n, h);
    h = fnv1a(&c.samples_per_lane, sizeof(c.samples_per_lane), h);
    for (int r = 0; r < kRoles; ++r) h = fnv1a(c.lowered[r].c_str(), c.lowered[r].size() + 1, h);
    out[k] = fnv1a(c.code.data(), c.code.size(), h);
  }
}
inline bool plausible_symbol(const std::string& s) {
  for (const char ch : s)
    if (!(std::isalnum((unsigned char)ch) || ch == '_' || ch == '$' || ch == '.')) return false;
  return s.size() < 4096;
}

// cache file: "GWIJIT2\n", the five lowered names (one per line, empty lines for absent roles), a line with the digest of
// (kinds, samples per lane, names, code object) in hex, then the code object.  A file is used only if it is a regular file of
// this user (opened without following links) whose digest matches -- a damaged or foreign file is recompiled over, never loaded.
inline bool read_cache(const std::string& path, Chain& c) {
  const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
  if (fd < 0) return false;
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_uid != geteuid() || (st.st_mode & (S_IWGRP | S_IWOTH))) {
    close(fd);
    return false;
  }
  FILE* f = fdopen(fd, "rb");
  if (!f) {
    close(fd);
    return false;
  }
  std::vector<char> all;
  char buf[65536];
  size_t got;
  while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) all.insert(all.end(), buf, buf + got);
  std::fclose(f);
  size_t pos = 0;
  auto line = [&](std::string& out) {
    const size_t start = pos;
    while (pos < all.size() && all[pos] != '\n') ++pos;
    if (pos >= all.size()) return false;
    out.assign(all.data() + start, pos - start);
    ++pos;
    return true;
  };
  auto reject = [&] {
    for (int r = 0; r < kRoles; ++r) c.lowered[r].clear();
    c.code.clear();
    return false;
  };
  std::string magic, hex;
  if (!line(magic) || magic != "GWIJIT2") return false;
  for (int r = 0; r < kRoles; ++r)
    if (!line(c.lowered[r]) || !plausible_symbol(c.lowered[r])) return reject();
  if (!line(hex) || hex.size() != 32) return reject();
  if (all.size() - pos < 64 || std::memcmp(all.data() + pos, "\177ELF", 4) != 0) return reject();
  c.code.assign(all.begin() + (long)pos, all.end());
  unsigned long long want[2];
  digest_of(c, want);
  char have[40];
  std::snprintf(have, sizeof(have), "%016llx%016llx", want[0], want[1]);
  if (hex != have) return reject();
  return true;
}
inline bool write_cache(const std::string& path, const Chain& c) {
  const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
  const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
  FILE* f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
  if (!f) {
    if (fd >= 0) close(fd);
    return false;
  }
  unsigned long long dg[2];
  digest_of(c, dg);
  bool ok = std::fputs("GWIJIT2\n", f) >= 0;
  for (int r = 0; r < kRoles; ++r) ok = ok && std::fprintf(f, "%s\n", c.lowered[r].c_str()) >= 0;
  ok = ok && std::fprintf(f, "%016llx%016llx\n", dg[0], dg[1]) >= 0;
  ok = ok && std::fwrite(c.code.data(), 1, c.code.size(), f) == c.code.size();
  ok = (std::fclose(f) == 0) && ok;
  if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) {
    std::remove(tmp.c_str());
    return false;
  }
  return true;
}

// the name expression of one role's kernel
inline std::string name_expression(const Chain& c, int role) {
  std::string ks;
  for (int t = 0; t < c.n; ++t) ks += ", " + std::to_string(c.kinds[t]);
  const std::string u = std::to_string(c.samples_per_lane);
  switch (role) {
    case kScan: return "&gwi::scan_kernel<false, false, false, " + u + ks + ">";
    case kLogw: return "&gwi::scan_kernel<true, false, false, " + u + ks + ">";
    case kBatch: return "&gwi::scan_kernel<false, true, false, " + u + ks + ">";
    case kSafe: return c.spline ? "&gwi::scan_kernel<false, false, true, " + u + ks + ">" : "";
    case kPbatch: return c.spline ? "" : "&gwi::scan_pbatch_kernel<" + std::to_string(pbatch_u(c.samples_per_lane)) + ks + ">";
    default: return "";
  }
}

// hipRTC has no system headers and keeps the fixed-width integer types in a namespace of its own: the preamble supplies what
// include/gwi_engine.h and gwi_device.h take from <stdint.h> / <cstddef> in the ahead-of-time build
inline const char* preamble() {
  return "typedef signed int int32_t;\ntypedef long int64_t;\ntypedef unsigned int uint32_t;\ntypedef unsigned long uint64_t;\n"
         "typedef unsigned short uint16_t;\ntypedef unsigned char uint8_t;\n#define offsetof(t, m) __builtin_offsetof(t, m)\n"
         "#include \"gwi_device.h\"\n";
}
// ... and for the batched matrix-core kernel of a spline model (gwi_mfma.h includes gwi_device.h itself)
inline const char* preamble_mfma() {
  return "typedef signed int int32_t;\ntypedef long int64_t;\ntypedef unsigned int uint32_t;\ntypedef unsigned long uint64_t;\n"
         "typedef unsigned short uint16_t;\ntypedef unsigned char uint8_t;\n#define offsetof(t, m) __builtin_offsetof(t, m)\n"
         "#include \"gwi_mfma.h\"\n";
}

// Compile (or fetch from the disk cache) the chain of `kinds` with U samples per lane.  `device_h` / `engine_h`: the texts of
// gwi_device.h and include/gwi_engine.h this library was built from.  Process-wide cache; returns nullptr and says why.
inline std::mutex& chains_mutex() {
  static std::mutex mu;
  return mu;
}
inline std::vector<Chain*>& chains_list() {
  static std::vector<Chain*> chains;
  return chains;
}
// A cached code object the runtime refuses to load (a file damaged on disk): forget the chain and its file, so that the next
// get_chain() compiles afresh.  The Chain object itself is leaked (another engine may still hold it).
inline void discard_chain(Chain* c) {
  std::lock_guard<std::mutex> lock(chains_mutex());
  auto& v = chains_list();
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i] == c) {
      v.erase(v.begin() + (long)i);
      break;
    }
  if (!c->path.empty()) std::remove(c->path.c_str());
}

// U = 0 (with `mfma_h`, the text of gwi_mfma.h): the batched matrix-core kernel scan_mfma_kernel of a spline model instead of the
// scan roles -- `kinds` then hold kind + 100 x (16-basis gradient tiles of the term), as gwi_mfma.h's chain takes them; one
// kernel (role kScan of the returned object).
inline Chain* get_chain(const int* kinds, int n, int U, const char* device_h, const char* engine_h, std::string& why, const char* mfma_h = nullptr) {
  std::mutex& mu = chains_mutex();
  std::vector<Chain*>& chains = chains_list();
  const bool mfma = U == 0 && mfma_h != nullptr;
  if (n < 1 || n > GWI_MAX_TERMS || (U != 1 && U != 2 && !mfma)) {
    why = "jit: 1 to 12 term kinds, one or two samples per lane";
    return nullptr;
  }
  for (int t = 0; t < n; ++t) {
    const int k = mfma ? kinds[t] % 100 : kinds[t], k_prev = t > 0 ? (mfma ? kinds[t - 1] % 100 : kinds[t - 1]) : 0;
    if (k < 1 || k > GWI_TERM_EXP_SPLINE_LERP || k < k_prev || (mfma && (kinds[t] / 100 < 0 || kinds[t] / 100 > 8))) {
      why = "jit: term kinds are the GWI_TERM_* numbers in ascending order";
      return nullptr;
    }
  }
  std::lock_guard<std::mutex> lock(mu);
  for (Chain* c : chains) {
    bool same = c->n == n && c->samples_per_lane == U;
    for (int t = 0; t < n && same; ++t) same = c->kinds[t] == kinds[t];
    if (same) return c;
  }
  Chain* c = new Chain;
  c->n = n;
  c->samples_per_lane = U;
  std::string ks;
  for (int t = 0; t < n; ++t) {
    c->kinds[t] = kinds[t];
    c->spline = c->spline || is_spline_kind(kinds[t]);
    ks += (t ? "," : "") + std::to_string(kinds[t]);
  }
  c->name = mfma ? "jit-mfma:" + ks : "jit:" + ks + "/u" + std::to_string(U);
  const char* const pre = mfma ? preamble_mfma() : preamble();
  // key of the build: headers + flags (+ the hipRTC version, read below when the library is there)
  unsigned long long hsh = fnv1a(device_h, std::strlen(device_h));
  hsh = fnv1a(engine_h, std::strlen(engine_h), hsh);
  if (mfma) hsh = fnv1a(mfma_h, std::strlen(mfma_h), hsh);
  hsh = fnv1a(pre, std::strlen(pre), hsh);
  for (const char* f : flags()) hsh = fnv1a(f, std::strlen(f) + 1, hsh);
  Rtc& r = rtc();
  if (r.ok) {  // which compiler: version and the file it was loaded from (PyTorch's bundled ROCm or the system's)
    int v[2] = {0, 0};
    r.version(&v[0], &v[1]);
    hsh = fnv1a(v, sizeof(v), hsh);
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(r.compile), &info) && info.dli_fname) hsh = fnv1a(info.dli_fname, std::strlen(info.dli_fname), hsh);
  }
  char hx[20];
  std::snprintf(hx, sizeof(hx), "%016llx", hsh);
  const std::string dir = cache_dir();
  std::string file_ks = ks;
  for (char& ch : file_ks)
    if (ch == ',') ch = '-';
  if (!dir.empty()) c->path = dir + (mfma ? "/mfma_" : "/chain_") + file_ks + "_u" + std::to_string(U) + "_" + hx + ".gwijit";
  if (!c->path.empty() && read_cache(c->path, *c)) {
    c->from_cache = true;
    chains.push_back(c);
    return c;
  }
  if (!r.ok) {
    why = "jit: " + r.why;
    delete c;
    return nullptr;
  }
  const auto t0 = std::chrono::steady_clock::now();
  const char* header_texts[3] = {device_h, engine_h, mfma_h};
  const char* header_names[3] = {"gwi_device.h", "gwi_engine.h", "gwi_mfma.h"};
  hiprtcProgram prog = nullptr;
  if (r.create(&prog, pre, "gwi_jit_chain.hip", mfma ? 3 : 2, header_texts, header_names) != HIPRTC_SUCCESS) {
    why = "jit: hiprtcCreateProgram failed";
    delete c;
    return nullptr;
  }
  std::string exprs[kRoles];
  for (int role = 0; role < kRoles; ++role) {
    exprs[role] = mfma ? (role == kScan ? "&gwi::scan_mfma_kernel<1, " + ks + ">" : std::string()) : name_expression(*c, role);
    if (!exprs[role].empty()) r.add_name(prog, exprs[role].c_str());
  }
  const hiprtcResult rc = r.compile(prog, (int)flags().size(), const_cast<const char**>(flags().data()));
  if (rc != HIPRTC_SUCCESS) {
    size_t ls = 0;
    r.log_size(prog, &ls);
    std::string log(ls > 1 ? ls : 1, '\0');
    if (ls > 1) r.log(prog, &log[0]);
    if (log.size() > 1500) log.resize(1500);
    why = "jit: hipRTC could not compile chain " + ks + ": " + log;
    r.destroy(&prog);
    delete c;
    return nullptr;
  }
  bool ok = true;
  for (int role = 0; role < kRoles && ok; ++role) {
    if (exprs[role].empty()) continue;
    const char* low = nullptr;
    ok = r.lowered(prog, exprs[role].c_str(), &low) == HIPRTC_SUCCESS && low;
    if (ok) c->lowered[role] = low;
  }
  size_t cs = 0;
  ok = ok && r.code_size(prog, &cs) == HIPRTC_SUCCESS && cs > 64;
  if (ok) {
    c->code.resize(cs);
    ok = r.code(prog, c->code.data()) == HIPRTC_SUCCESS && std::memcmp(c->code.data(), "\177ELF", 4) == 0;
  }
  r.destroy(&prog);
  if (!ok) {
    why = "jit: hipRTC returned no code object for chain " + ks;
    delete c;
    return nullptr;
  }
  c->compile_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (!c->path.empty() && !write_cache(c->path, *c)) c->path.clear();
  chains.push_back(c);
  return c;
}

// the chain's module on `device` (the current device of the calling thread), loaded on first use
inline hipModule_t module_on(Chain* c, int device, std::string& why) {
  std::lock_guard<std::mutex> lock(c->mu);
  for (auto& kv : c->modules)
    if (kv.first == device) return kv.second;
  hipModule_t m = nullptr;
  const hipError_t e = hipModuleLoadData(&m, c->code.data());
  if (e != hipSuccess) {
    why = std::string("jit: hipModuleLoadData: ") + hipGetErrorString(e);
    (void)hipGetLastError();  // the runtime keeps the failure as its "last error": the next launch's check must not find it
    return nullptr;
  }
  c->modules.emplace_back(device, m);
  return m;
}

}  // namespace jit
}  // namespace gwi
#endif  // GWI_JIT_H
