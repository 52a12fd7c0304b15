// gwi_engine.hip -- host side + C ABI of the population-likelihood engine (see include/gwi_engine.h).
// gfx950 only; no CPU fallback: every entry point that computes needs a live HIP device.
#include "gwi_device.h"
#include "gwi_mfma.h"
#include "gwi_aql.h"
#include "gwi_ingest.h"
#include "gwi_jit.h"

#include <hip/hip_ext.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cerrno>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

using namespace gwi;

// The two headers a scan chain is compiled from, embedded as text: what hipRTC gets when a model's term sequence has no
// ahead-of-time instantiation (gwi_jit.h).  (.incbin searches the -I directories of the build; host pass only.)
#ifndef __HIP_DEVICE_COMPILE__
__asm__(
    ".pushsection .rodata\n"
    ".global gwi_embedded_device_h\n"
    "gwi_embedded_device_h:\n"
    ".incbin \"gwi_device.h\"\n"
    ".byte 0\n"
    ".global gwi_embedded_engine_h\n"
    "gwi_embedded_engine_h:\n"
    ".incbin \"gwi_engine.h\"\n"
    ".byte 0\n"
    ".global gwi_embedded_mfma_h\n"
    "gwi_embedded_mfma_h:\n"
    ".incbin \"gwi_mfma.h\"\n"
    ".byte 0\n"
    ".popsection\n");
#endif
extern "C" const char gwi_embedded_device_h[];
extern "C" const char gwi_embedded_engine_h[];
extern "C" const char gwi_embedded_mfma_h[];

namespace {

// ------------------------------------------------------------------------------------------------
// compiled term sequences: each entry below is one explicit instantiation of scan_kernel.
// ------------------------------------------------------------------------------------------------
// the scan takes its tile geometry and first column pointers as fourteen scalar dwords ahead of the argument block (ScanHead,
// gwi_device.h): the command processor preloads those into scalar registers
using ScanFn = void (*)(const double*, const double*, const double*, const double*, const double*, unsigned, unsigned, unsigned, unsigned, const KArgs);
using MfmaFn = void (*)(const KArgs);
// the scan's kernel-argument segment as the host stages it
struct ScanBlock {
  ScanHead head;
  KArgs k;
};
static_assert(offsetof(ScanBlock, k) == sizeof(ScanHead) && sizeof(ScanHead) % alignof(KArgs) == 0, "the struct follows the scalars without padding, as in the kernel's argument list");

// Kernel roles of a term sequence (jit::Role): scan (single evaluation), logw (per-sample log-weights), batch (one grid row
// per point), safe (spline models: two-pass / replay instantiation, single and batched launches), pbatch (parametric models:
// batched launches that load every sample once, scan_pbatch_kernel).
struct Variant {
  const char* name;
  int n;
  int kinds[GWI_MAX_TERMS];
  int samples_per_lane;
  ScanFn fn[jit::kRoles];  // nullptr where the role has no instantiation -- and for every role of a chain compiled at run time:
  jit::Chain* jit;         // ... whose kernels come out of this chain's code object (gwi_jit.h)
  bool has(int role) const { return jit ? !jit->lowered[role].empty() : fn[role] != nullptr; }
};

// the SAFE instantiation exists for spline term sequences only, the pbatch one for the others
template <int U, int... Ks>
constexpr ScanFn safe_scan() {
  if constexpr (Chain<U, Ks...>::kSpline)
    return &scan_kernel<false, false, true, U, Ks...>;
  else
    return nullptr;
}
template <int U, int... Ks>
constexpr ScanFn pbatch_scan() {
  if constexpr (Chain<U, Ks...>::kSpline)
    return nullptr;
  else
    return &scan_pbatch_kernel<pbatch_u(U), Ks...>;
}

#define K_PL GWI_TERM_POWERLAW
#define K_PP GWI_TERM_PLPEAK
#define K_PQ GWI_TERM_POWERLAW_RATIO
#define K_BE GWI_TERM_BETA
#define K_TI GWI_TERM_TILT_MIXTURE
#define K_PZ GWI_TERM_POWERLAW_REDSHIFT
#define K_SP GWI_TERM_EXP_SPLINE
#define K_TN GWI_TERM_TRUNCNORM
#define K_LS GWI_TERM_LINEAR_SPLINE
#define K_TJ GWI_TERM_TILT_JOINT
#define K_SM GWI_TERM_SMOOTH
#define K_PS GWI_TERM_PLPEAK_SMOOTH
#define K_PB GWI_TERM_POWERLAW_BOUNDS
#define K_SL GWI_TERM_EXP_SPLINE_LERP

// U = samples per lane per trip (2 for the register-light parametric models, 1 or 2 for spline models)
#define GWI_VARIANT_U(NAME, U, ...) \
  { NAME, (int)(sizeof((int[]){__VA_ARGS__}) / sizeof(int)), {__VA_ARGS__}, U, {&scan_kernel<false, false, false, U, __VA_ARGS__>, \
    &scan_kernel<true, false, false, U, __VA_ARGS__>, &scan_kernel<false, true, false, U, __VA_ARGS__>,                           \
    safe_scan<U, __VA_ARGS__>(), pbatch_scan<U, __VA_ARGS__>()}, nullptr }
#define GWI_VARIANT(NAME, ...) GWI_VARIANT_U(NAME, 2, __VA_ARGS__)

// Term sequences are canonical: the host sorts a model's terms by kind id (stable).
const Variant kVariants[] = {
#ifdef GWI_AB_FEW_VARIANTS  // quick experiment builds (tools/build_ablations.sh): the chains of the BASELINE configurations only
    GWI_VARIANT("plpeak+plq+plz", K_PP, K_PQ, K_PZ),
    GWI_VARIANT("plpeak+plq+beta2+tilt2+plz", K_PP, K_PQ, K_BE, K_BE, K_TI, K_TI, K_PZ),
    GWI_VARIANT("plq+plz+spline5", K_PQ, K_PZ, K_SP, K_SP, K_SP, K_SP, K_SP),
    GWI_VARIANT_U("plq+plz+spline5/u1", 1, K_PQ, K_PZ, K_SP, K_SP, K_SP, K_SP, K_SP),
    GWI_VARIANT_U("plz+spline7", 1, K_PZ, K_SP, K_SP, K_SP, K_SP, K_SP, K_SP, K_SP),
#else
    // tests/inference_test.py:162-197 -- powerlaw_primary_ratio_pdf x PowerlawRedshiftModel
    GWI_VARIANT("pl+plq+plz", K_PL, K_PQ, K_PZ),
    // BASELINE config 2 -- PL+Peak m1 x PL q [x PL z]
    GWI_VARIANT("plpeak+plq", K_PP, K_PQ),
    GWI_VARIANT("plpeak+plq+plz", K_PP, K_PQ, K_PZ),
    GWI_VARIANT_U("plpeak+plq+plz/u1", 1, K_PP, K_PQ, K_PZ),  // 128 VGPRs -> 4 waves/SIMD
    // BASELINE config 1 -- + independent Beta magnitudes + independent tilt mixtures
    GWI_VARIANT("plpeak+plq+beta2+tilt2+plz", K_PP, K_PQ, K_BE, K_BE, K_TI, K_TI, K_PZ),
    // tests/inference_test.py:244-285 -- PL z x {BSpline m1, BSpline q, spline(log z)}
    GWI_VARIANT("plz+spline3", K_PZ, K_SP, K_SP, K_SP),
    GWI_VARIANT_U("plz+spline3/u1", 1, K_PZ, K_SP, K_SP, K_SP),
    // BASELINE config 3/4 -- PL q x PL z x {BSpline m1, IID spin magnitudes [, IID tilts]}
    GWI_VARIANT("plq+plz+spline3", K_PQ, K_PZ, K_SP, K_SP, K_SP),
    GWI_VARIANT("plq+plz+spline5", K_PQ, K_PZ, K_SP, K_SP, K_SP, K_SP, K_SP),
    GWI_VARIANT_U("plq+plz+spline5/u1", 1, K_PQ, K_PZ, K_SP, K_SP, K_SP, K_SP, K_SP),
    // BASELINE config 5 -- PL z x {BSpline m1, q, a1, a2, ct1, ct2, spline(log z)}
    // (one sample per lane: 160 VGPRs -> 3 waves/SIMD; measured 83 vs 93 us per scan on config 5)
    GWI_VARIANT_U("plz+spline7", 1, K_PZ, K_SP, K_SP, K_SP, K_SP, K_SP, K_SP, K_SP),
    GWI_VARIANT_U("plz+spline7/u2", 2, K_PZ, K_SP, K_SP, K_SP, K_SP, K_SP, K_SP, K_SP),
    // PLPeakPrimaryBSplineRatio (separable.py:368-443) x PL z
    GWI_VARIANT("plpeak+plz+spline", K_PP, K_PZ, K_SP),
    // plpeak_primary_ratio_pdf (parametric.py:39-46) x B-spline spin magnitudes and tilts (IID or independent:
    // separable.py:17-292) x PL z -- parametric masses with non-parametric spins
    GWI_VARIANT("plpeak+plq+plz+spline4", K_PP, K_PQ, K_PZ, K_SP, K_SP, K_SP, K_SP),
    GWI_VARIANT("plpeak+plq+plz+spline2", K_PP, K_PQ, K_PZ, K_SP, K_SP),                      // ... magnitudes only (or tilts only)
    GWI_VARIANT("plpeak+plq+plz+spline5", K_PP, K_PQ, K_PZ, K_SP, K_SP, K_SP, K_SP, K_SP),    // ... and the redshift spline
    // other products of the separable B-spline models (separable.py) the reference's factories allow (pipeline/utils.py:104-155):
    // B-spline masses with spin magnitudes / tilts and a power-law or spline redshift model
    GWI_VARIANT("plz+spline4", K_PZ, K_SP, K_SP, K_SP, K_SP),
    GWI_VARIANT("plz+spline5", K_PZ, K_SP, K_SP, K_SP, K_SP, K_SP),
    GWI_VARIANT_U("plz+spline6", 1, K_PZ, K_SP, K_SP, K_SP, K_SP, K_SP, K_SP),
    GWI_VARIANT("plq+plz+spline2", K_PQ, K_PZ, K_SP, K_SP),
    GWI_VARIANT("plq+plz+spline4", K_PQ, K_PZ, K_SP, K_SP, K_SP, K_SP),
    GWI_VARIANT_U("plq+plz+spline6", 1, K_PQ, K_PZ, K_SP, K_SP, K_SP, K_SP, K_SP, K_SP),
    // mass-only B-spline models: BSplinePrimaryBSplineRatio / BSplinePrimaryPowerlawRatio x PL z
    GWI_VARIANT("plz+spline2", K_PZ, K_SP, K_SP),
    GWI_VARIANT("plq+plz+spline", K_PQ, K_PZ, K_SP),
    // BSplinePrimaryBSplineRatio x BSplineEffectiveSpinDims (chi_eff, chi_p: separable.py:706-778) x PL z
    GWI_VARIANT_U("plz+spline2+lspline2", 1, K_PZ, K_SP, K_SP, K_LS, K_LS),
    // BSplineIID/IndependentComponentMasses (separable.py:533-703): (m2/m1)^beta x p(m1) p(m2) x PL z
    GWI_VARIANT("pl+plz+spline2", K_PL, K_PZ, K_SP, K_SP),
    // PL+Peak x PL q x default_spin_tilt (parametric.py:97-102) x PL z
    GWI_VARIANT("plpeak+plq+plz+tiltjoint", K_PP, K_PQ, K_PZ, K_TJ),
    // BSplineRedshift (single.py:398-492) in place of the power-law redshift factor: with the parametric
    // mass pair, and with BSplinePrimaryBSplineRatio
    GWI_VARIANT("pl+plq+spline", K_PL, K_PQ, K_SP),
    GWI_VARIANT("spline3", K_SP, K_SP, K_SP),
    // BSplinePrimaryBSplineRatio alone: the (m1, q) mesh of the posterior-predictive curves (postprocess/calculations.py:20-60)
    GWI_VARIANT("spline2", K_SP, K_SP),
    // log-normal m1 peak x BSplineRatio (postprocess/calculations.py:94-130)
    GWI_VARIANT("spline+truncnorm", K_SP, K_TN),
    // merger rate of redshift, (1 + z)^lamb exp(spline(log z)) (postprocess/calculations.py:261-276)
    GWI_VARIANT("pl+spline", K_PL, K_SP),
    // plpeak_primary_ratio_pdf with the low-mass taper `delta` (parametric.py:39-53) [x PL z]
    GWI_VARIANT("plq+plz+smooth+plpeaksmooth", K_PQ, K_PZ, K_SM, K_PS),
    GWI_VARIANT("plq+smooth+plpeaksmooth", K_PQ, K_SM, K_PS),
    GWI_VARIANT("plz+plpeaksmooth", K_PZ, K_PS),
    // PLPeakPrimaryBSplineRatio (separable.py:368-443) x BSplineSymmetricChiEffective (single.py:233-284) x PL z
    GWI_VARIANT("plpeak+plz+spline+lspline", K_PP, K_PZ, K_SP, K_LS),
    // construct_hierarchical_model (analysis.py:359-424) on the reference's own distributions
    // (numpyro_distributions.py): Powerlaw m1 and q with sampled bounds x PowerlawRedshift
    // (examples/config_files/config.yml), and BSplineDistribution m1, q x PowerlawRedshift
    GWI_VARIANT("plz+plb2", K_PZ, K_PB, K_PB),
    GWI_VARIANT("plz+lerp2", K_PZ, K_SL, K_SL),
    GWI_VARIANT("plb", K_PB),
    GWI_VARIANT("lerp", K_SL),
    // single-term sequences (term-level parity tests)
    GWI_VARIANT("lspline", K_LS),
    GWI_VARIANT("tiltjoint", K_TJ),
    GWI_VARIANT("pl", K_PL),
    GWI_VARIANT("plpeak", K_PP),
    GWI_VARIANT("plq", K_PQ),
    GWI_VARIANT("beta", K_BE),
    GWI_VARIANT("tilt", K_TI),
    GWI_VARIANT("plz", K_PZ),
    GWI_VARIANT("spline", K_SP),
    GWI_VARIANT("truncnorm", K_TN),
    GWI_VARIANT("smooth", K_SM),
    GWI_VARIANT("plpeaksmooth", K_PS),
#endif
};
constexpr int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));

// The fallback for every other product of terms: the generic chain (gwi_device.h, kGenericChain), whose term kinds are read
// from the argument block at run time.  One kernel plays every role (single / batched / two-pass / replay: the SAFE
// instantiation takes those as run-time options) plus the log-weight variant.
const Variant kGenericVariant = {"generic (run-time term loop)", 0, {0}, 1, {&scan_kernel<false, false, true, 1, kGenericChain>, &scan_kernel<true, false, false, 1, kGenericChain>,
                                 &scan_kernel<false, false, true, 1, kGenericChain>, &scan_kernel<false, false, true, 1, kGenericChain>, nullptr}, nullptr};

// ---- batched launches of spline models on the matrix cores (gwi_mfma.h): term sequences with the number of 16-basis
// gradient tiles of every spline term fixed at compile time.  A model qualifies when its kinds match and every spline
// term has n_basis <= 16 * tiles; the first qualifying entry is used (entries with fewer tiles first).
struct MfmaVariant {
  const char* name;
  int n;
  int kinds[GWI_MAX_TERMS];
  int tiles[GWI_MAX_TERMS];
  MfmaFn fn;        // gradient tiles on the matrix cores (scan_mfma_kernel)
  MfmaFn rows_fn;   // gradient rows in LDS (scan_rows_kernel)
  int row_doubles;  // doubles a staged sample occupies (gwi_mfma.h: MChain::kRowDoubles)
};
#define T1(K) (100 + (K))
#define T2(K) (200 + (K))
#define T4(K) (400 + (K))
#define GWI_MFMA(NAME, U, ...) \
  { NAME, (int)(sizeof((int[]){__VA_ARGS__}) / sizeof(int)), {__VA_ARGS__}, {0}, &scan_mfma_kernel<U, __VA_ARGS__>, &scan_rows_kernel<U, __VA_ARGS__>, MChain<false, __VA_ARGS__>::kRowDoubles }
MfmaVariant kMfmaVariants[] = {
    // tests/inference_test.py:244-285 model and the mass-only models
    GWI_MFMA("plz+spline3 (16,16,16)", 1, K_PZ, T1(K_SP), T1(K_SP), T1(K_SP)),
    GWI_MFMA("plz+spline2 (32,16)", 1, K_PZ, T2(K_SP), T1(K_SP)),
    // BASELINE config 3/4: PL q x PL z x {m1 (30), IID spin magnitudes (16, 16), IID tilts (16, 16)}
    GWI_MFMA("plq+plz+spline5 (32,16,16,16,16)", 1, K_PQ, K_PZ, T2(K_SP), T1(K_SP), T1(K_SP), T1(K_SP), T1(K_SP)),
    GWI_MFMA("plq+plz+spline3 (32,16,16)", 1, K_PQ, K_PZ, T2(K_SP), T1(K_SP), T1(K_SP)),
    // BASELINE config 5: PL z x {m1 (30), q (14), a1, a2, ct1, ct2 (12 each), z (12)}
    GWI_MFMA("plz+spline7 (32,16,16,16,16,16,16)", 1, K_PZ, T2(K_SP), T1(K_SP), T1(K_SP), T1(K_SP), T1(K_SP), T1(K_SP), T1(K_SP)),
    // the reference's default spline counts (pipeline/utils.py:29-33): m1 50, q 30, spins 16, z 20
    GWI_MFMA("plz+spline7 (64,32,16,16,16,16,32)", 1, K_PZ, T4(K_SP), T2(K_SP), T1(K_SP), T1(K_SP), T1(K_SP), T1(K_SP), T2(K_SP)),
    // linear (chi_eff / chi_p) splines
    GWI_MFMA("plz+spline2+lspline2 (16,16,16,16)", 1, K_PZ, T1(K_SP), T1(K_SP), T1(K_LS), T1(K_LS)),
    // parametric masses with B-spline spins
    GWI_MFMA("plpeak+plq+plz+spline4 (16,16,16,16)", 1, K_PP, K_PQ, K_PZ, T1(K_SP), T1(K_SP), T1(K_SP), T1(K_SP)),
};
constexpr int kNumMfmaVariants = (int)(sizeof(kMfmaVariants) / sizeof(kMfmaVariants[0]));
struct MfmaTableInit {
  MfmaTableInit() {
    for (auto& v : kMfmaVariants)
      for (int t = 0; t < v.n; ++t) {
        v.tiles[t] = v.kinds[t] / 100;
        v.kinds[t] %= 100;
      }
  }
} g_mfma_table_init;

const MfmaVariant* find_mfma_variant(const gwi_spec& s) {
  for (int v = 0; v < kNumMfmaVariants; ++v) {
    const MfmaVariant& m = kMfmaVariants[v];
    if (m.n != s.n_terms) continue;
    bool ok = true;
    for (int t = 0; t < s.n_terms && ok; ++t) {
      ok = m.kinds[t] == s.terms[t].kind;
      if (ok && m.tiles[t] > 0) ok = s.terms[t].n_basis <= 16 * m.tiles[t];
    }
    if (ok) return &m;
  }
  return nullptr;
}

// First entry whose kind sequence matches; GWI_SAMPLES_PER_LANE=1|2 prefers that unroll where compiled.
const Variant* find_variant(const gwi_spec& s) {
  int prefer = 0;
  if (const char* env = std::getenv("GWI_SAMPLES_PER_LANE")) prefer = std::atoi(env);
  const Variant* first = nullptr;
  for (int v = 0; v < kNumVariants; ++v) {
    if (kVariants[v].n != s.n_terms) continue;
    bool same = true;
    for (int t = 0; t < s.n_terms; ++t) same = same && kVariants[v].kinds[t] == s.terms[t].kind;
    if (!same) continue;
    if (!first) first = &kVariants[v];
    if (prefer && kVariants[v].samples_per_lane == prefer) return &kVariants[v];
  }
  return first;
}

// record published by final_kernel (doubles):
//   [0] completion stamp  [1] sum_i logsumexp_i  [2] sum_i variance_i  [3] min_i nan_to_num(log n_eff_i)
//   [4] inj M  [5] inj S1  [6] inj S2  [7] n_ev (local)  [8..) norms, grad_pe[n_theta], grad_inj[n_theta]
constexpr int kRecNormOff = 8;

// ------------------------------------------------------------------------------------------------
// hyper-parameter-only scalars ("prelude"), computed on the host in double precision
// ------------------------------------------------------------------------------------------------
// log of the power-law normaliser (1+a)/(hi^(1+a) - lo^(1+a)) and its alpha-derivative
// (distributions.py:112-116), evaluated in the log domain so large |alpha| cannot overflow.
void powerlaw_lognorm(double alpha, double lo, double hi, double* logA, double* dlogA) {
  const double a1 = 1.0 + alpha, llo = std::log(lo), lhi = std::log(hi);
  if (a1 == 0.0) {
    *logA = -std::log(lhi - llo);
    *dlogA = -0.5 * (lhi + llo);
    return;
  }
  if (a1 > 0) {
    const double rho = std::exp(a1 * (llo - lhi));  // (lo/hi)^a1
    *logA = std::log(a1) - (a1 * lhi + std::log1p(-rho));
    *dlogA = 1.0 / a1 - (lhi - rho * llo) / (1.0 - rho);
  } else {
    const double rho = std::exp(a1 * (lhi - llo));  // (hi/lo)^a1
    *logA = std::log(-a1) - (a1 * llo + std::log1p(-rho));
    *dlogA = 1.0 / a1 - (llo - rho * lhi) / (1.0 - rho);
  }
}

// log of the truncated-normal normaliser 1/(sig sqrt(2pi) (Phi(b)-Phi(a))) and its derivatives
// (distributions.py:136-142)
void truncnorm_lognorm(double mu, double sg, double lo, double hi, double* logC, double* dmu, double* dsg) {
  const double r2 = std::sqrt(2.0);
  const double a = (lo - mu) / sg, b = (hi - mu) / sg;
  const double dphi = 0.5 * (1.0 + std::erf(b / r2)) - 0.5 * (1.0 + std::erf(a / r2));
  const double inv_s2pi = 1.0 / std::sqrt(2.0 * M_PI);
  const double pa = std::exp(-0.5 * a * a) * inv_s2pi, pb = std::exp(-0.5 * b * b) * inv_s2pi;
  *logC = -std::log(sg) - 0.5 * std::log(2.0 * M_PI) - std::log(dphi);
  *dmu = (pb - pa) / (sg * dphi);
  *dsg = -1.0 / sg + (b * pb - a * pa) / (sg * dphi);
}

}  // namespace

// ---- RCCL, bound at run time (no link-time dependency; the same librccl the host framework uses) ----
namespace {
struct NcclId {  // ncclUniqueId (rccl.h:43), passed by value
  char b[128];
};
struct NcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
NcclApi g_nccl;
constexpr int kNcclDouble = 8;  // ncclFloat64 (rccl.h:467)

bool load_nccl(const char* path, std::string* err) {
  if (g_nccl.lib) return true;
  const char* p = (path && *path) ? path : "librccl.so.1";
  void* lib = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
  if (!lib) {
    *err = std::string("dlopen(") + p + ") failed: " + dlerror();
    return false;
  }
  g_nccl.GetUniqueId = reinterpret_cast<decltype(g_nccl.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  g_nccl.CommInitRank = reinterpret_cast<decltype(g_nccl.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  g_nccl.AllGather = reinterpret_cast<decltype(g_nccl.AllGather)>(dlsym(lib, "ncclAllGather"));
  g_nccl.CommDestroy = reinterpret_cast<decltype(g_nccl.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  g_nccl.GetErrorString = reinterpret_cast<decltype(g_nccl.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  if (!g_nccl.GetUniqueId || !g_nccl.CommInitRank || !g_nccl.AllGather || !g_nccl.CommDestroy) {
    *err = std::string("librccl at ") + p + " lacks the expected symbols";
    return false;
  }
  g_nccl.lib = lib;
  return true;
}
}  // namespace

struct gwi_engine {
  gwi_spec spec;
  const Variant* variant = nullptr;
  Variant* jit_variant = nullptr;     // owned: the variant record of a chain compiled at gwi_create (variant points at it)
  hipFunction_t jit_fn[jit::kRoles] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // ... its kernels in the chain's module on this device
  std::string jit_note;               // why the generic kernel runs although a chain could have been compiled (gwi_jit_info)
  long long aql_rerings = 0;          // times a wait on the AQL queue rang the doorbell a second time (aql_wait_slow)
  bool pbatch = false;                // parametric model: batched launches load every sample once (scan_pbatch_kernel)
  int pbatch_pts = 0;                 // ... GWI_PBATCH_PTS: points per grid row in rows mode (0: chosen per launch from the batch size and the grid)
  bool pbatch_balanced = true;        // ... (tile, point) units dealt out evenly to one round of resident workgroups (GWI_PBATCH_BALANCED=0: rows mode)
  int pbatch_wgs_per_cu = 4;          // ... resident workgroups of scan_pbatch_kernel per CU (occupancy query at gwi_create)
  int scan_role = jit::kScan;         // role of the scan launch being issued
  const MfmaVariant* mfma = nullptr;  // batched launches with 16 points per wavefront (gwi_mfma.h), when the model qualifies
  bool batch_rows = false;            // ... with the gradient in LDS rows (scan_rows_kernel) instead of MFMA tiles
  int rows_rep = 4;
  int mfma_min_batch = 9;             // ... from this many points per launch (a wave carries 16)
  // which of the two batched kernels a spline model runs follows a STATIC rule (matrix cores from 9 points on when the model has
  // <= 8 gradient tiles): the two kernels sum in different orders, so the choice must not depend on a race of wall times -- same
  // model + same catalog shape = same kernel = same bits, on every handle and every box.  GWI_BATCH_AUTOTUNE=1 opts into the
  // measurement (calibrate_batch_path: three launches of each on the caller's own points on the first batched launch, the faster
  // one stays; per handle, costs 8 extra evaluation sets once, and results then depend on which kernel won).
  // a spline model without an ahead-of-time matrix-core instantiation gets one compiled (gwi_jit.h) on its first batched launch
  // of >= 9 points, or at gwi_create when GWI_BATCH_MFMA=1 asks for that path
  MfmaVariant* jit_mfma = nullptr;    // owned record of that instantiation (mfma points at it once it is loaded)
  hipFunction_t jit_mfma_fn = nullptr;
  bool mfma_jit_pending = false;      // worth trying, not tried yet
  std::string mfma_jit_note;
  bool autotune_wanted = false;       // GWI_BATCH_AUTOTUNE=1
  bool batch_autotune = false;        // a choice is still to be made (only ever true when autotune_wanted)
  bool batch_measured = false;        // ... and has been
  double batch_us[2] = {0.0, 0.0};    // best wall time of one batched evaluation set: [matrix-core kernel, 4-tap kernel]
  size_t mfma_lds_bytes = 0;
  bool batch_used_mfma = false;       // path of the most recent batched launch
  bool batch_events = true;           // gwi_eval_batch: the caller wants the per-event sites
  int combine_threads = kBlock;       // workgroup size of the combine launch
  int device = 0;
  int n_cus = 256;
  hipStream_t stream = nullptr;
  long long n_ev = 0, n_pe = 0, n_inj = 0;
  // device memory
  std::vector<double*> d_cols_pe, d_cols_inj;
  NormD* d_norms = nullptr;
  std::vector<double*> d_norm_arrays;
  double *d_partials = nullptr, *d_ev_out = nullptr, *d_ev_grad = nullptr, *d_inj_out = nullptr, *d_inj_grad = nullptr;
  std::vector<double> sq_records;  // records of the squared-weight pass (marginalize_selection gradient)
  double *d_logw_pe = nullptr, *d_logw_inj = nullptr;
  // pinned, device-visible host memory
  double *h_record = nullptr, *h_record_dev = nullptr;
  // device-final mode: the final launch's G workgroups publish one partial record each here; the host merges them into h_record
  double *h_fin = nullptr, *h_fin_dev = nullptr;
  int final_groups = 1;
  double *h_ev = nullptr, *h_ev_dev = nullptr;
  // host-final mode: per-group result rows + normaliser values in pinned host memory
  bool host_final = false;
  // launch geometry of batched launches (K >= 4, device-final) where it differs from the single evaluation's: two trips
  // per workgroup instead of one (gwi_create)
  struct BatchGeometry {
    bool distinct = false;
    int chunk_pe = 0, chunk_inj = 0, tiles_per_event = 0, n_inj_tiles = 0, n_scan_blocks = 0, tiles_per_inj_group = 0, n_inj_groups = 0;
  } bgeo;
  bool use_bgeo = false;  // the pipeline being issued runs on bgeo
  double *h_rows = nullptr, *h_rows_dev = nullptr;
  double *h_norm = nullptr, *h_norm_dev = nullptr;                    // pinned: Z_j
  unsigned long long *h_norm_stamp = nullptr, *h_norm_stamp_dev = nullptr;  // pinned: per-normaliser stamps
  // launch geometry
  int tiles_per_event = 1, chunk_pe = 256, n_inj_tiles = 1, chunk_inj = 256, rec_stride = 0, n_scan_blocks = 0;
  int n_inj_groups = 1, tiles_per_inj_group = 1;

  size_t scan_lds_bytes = 0;
  int gacc_rep = 1;
  bool deterministic = false;   // GWI_DETERMINISTIC=1: replay mode of the shared gradient rows (scan_kernel)
  unsigned long long* d_seq = nullptr;                              // device words: [0] sequence number of the evaluation in flight, [1] redo request
  int* d_tile_nref = nullptr;   // spline models: every tile's reference exponent = its exact maximum at the previous evaluation (KArgs::tile_nref)
  unsigned long long *h_redo = nullptr, *h_redo_dev = nullptr;      // pinned: a scan workgroup asks for the two-pass repeat
  long redo_count = 0;          // evaluations repeated in two-pass mode so far
  unsigned long long seq = 0;
  // results of the last prelude (one per hyper-parameter point of the last launch)
  std::vector<double> host_consts = std::vector<double>(1, 0.0);
  // batched evaluation: up to max_batch hyper-parameter points per launch (blockIdx.y)
  int max_batch = 16;
  ThetaBlock *d_tblocks = nullptr, *h_tblocks = nullptr, *h_tblocks_dev = nullptr;
  bool stage_kernel = true;  // GWI_STAGE_KERNEL=0: upload theta blocks with hipMemcpyAsync instead
  // timing
  bool timing = false;
  bool spin_wait = true;
  bool host_only = false;
  // in-engine RCCL communicator (optional)
  void* nccl_comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  double *d_send = nullptr, *d_recv = nullptr;
  double *h_gather = nullptr, *h_gather_dev = nullptr;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // start/stop pairs: scan, combine, final
  float last_ms[3] = {0, 0, 0};
  bool timed_final = false;
  bool last_host_rows = false;   // how the most recent run_pipeline publishes (what its waiter must poll)
  bool pending = false;          // gwi_eval_begin issued, gwi_eval_end not yet called
  bool pending_sq = false;
  bool pending_batch = false;          // ... the pending evaluation is a batch (gwi_eval_batch_begin)
  int pending_k = 0;
  std::vector<double> pending_thetas;
  gwi_options pending_opt{};
  std::string err;
  // the engine's own AQL queue (gwi_aql.h): plain single-point evaluations are dispatched through it
  aql::Queue aq;
  aql::Kernel aq_scan, aq_scan_safe, aq_scan_batch, aq_scan_pbatch, aq_combine, aq_final;
  bool aq_have_pbatch = false;
  bool aql_batch = false;          // batched launches (4-tap kernel) can go through the AQL queue too
  char* aq_tail_batch[2] = {nullptr, nullptr};  // persistent TailArgs of batched launches: [publish_events]
  bool scan_is_batch = false;
  int aql_tail_variant = 0;        // 0: single evaluation, 1 / 2: batched without / with the per-event sites
  bool scan_is_safe = false;  // the scan launch being issued is the SAFE instantiation
  bool generic = false;       // no compiled chain for this model: the generic scan kernel (kGenericVariant)
  bool small_geometry = false;  // small catalog of a spline model: the one-sample-per-lane sibling on many small workgroups (gwi_create)
  bool combine_acquire = false;  // the combine packet carries an acquire fence after all (A/B only)
  bool aql_tail_only = true;  // scan launches rewrite only the per-evaluation tail of their argument block (GWI_AQL_TAIL=0: the whole block)
  unsigned tail_parity = 0; // which of the two persistent scan-argument slots the next launch rewrites (aql::dispatch_tail)
  bool aql_active = false;  // queue, argument ring and the three kernels are ready
  char* aq_tail_args = nullptr;  // persistent kernel-argument slot holding this engine's (constant) TailArgs
  bool aql_now = false;     // the pipeline being issued / awaited went through the AQL queue
  bool force_hip_stream = false;  // gwi_set_timing(h, 2): time with HIP events on the HIP stream (A/B against the AQL path)
  std::string aql_note;     // why not, when not
  bool poisoned = false;    // an evaluation timed out or the queue failed with work possibly in flight: no further evaluations
  // single-node record exchange through a POSIX shared-memory segment (gwi_shm_comm_init)
  char* shm_base = nullptr;
  size_t shm_bytes = 0, shm_slot_bytes = 0;
  int shm_rank = 0, shm_world = 0;
  unsigned long long shm_seq = 0;
  std::vector<double> shm_gather;
  ScanBlock sblock;          // what a scan launch takes: preloaded head + argument block
  KArgs& kargs = sblock.k;
  long long inj_off = 0;     // element offset of the injections inside every column allocation (inj_offset)
};

namespace {

#define GWI_HIP(call)                                                                               \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_);                                   \
      return GWI_ERR_HIP;                                                                           \
    }                                                                                               \
  } while (0)

gwi_status fail(gwi_handle h, gwi_status s, const std::string& msg) {
  if (h) h->err = msg;
  return s;
}

int record_len(const gwi_engine* h) { return kRecNormOff + h->spec.n_norms + 2 * h->spec.n_theta; }

gwi_status validate_spec(gwi_handle h, const gwi_spec* s) {
  if (s->abi_version != GWI_ABI_VERSION) return fail(h, GWI_ERR_INVALID, "abi_version mismatch");
  if (s->n_cols < 1 || s->n_cols > GWI_MAX_COLS) return fail(h, GWI_ERR_INVALID, "n_cols out of range");
  if (s->n_terms < 1 || s->n_terms > GWI_MAX_TERMS) return fail(h, GWI_ERR_INVALID, "n_terms out of range");
  if (s->n_theta < 1 || s->n_theta > GWI_MAX_THETA) return fail(h, GWI_ERR_INVALID, "n_theta out of range");
  if (s->n_norms < 0 || s->n_norms > GWI_MAX_NORMS) return fail(h, GWI_ERR_INVALID, "n_norms out of range");
  if (s->kappa_col < 0 || s->kappa_col >= s->n_cols) return fail(h, GWI_ERR_INVALID, "kappa_col out of range");
  if (s->vt_norm >= s->n_norms) return fail(h, GWI_ERR_INVALID, "vt_norm out of range");
  auto theta_ok = [&](int i) { return i >= 0 && i < s->n_theta; };
  auto col_ok = [&](int i) { return i >= 0 && i < s->n_cols; };
  for (int t = 0; t < s->n_terms; ++t) {
    const gwi_term& tm = s->terms[t];
    int n_cols = 1, n_th = 1;
    switch (tm.kind) {
      case GWI_TERM_POWERLAW: break;
      case GWI_TERM_PLPEAK: n_th = 4; break;  // one column: log x
      case GWI_TERM_POWERLAW_RATIO: n_cols = 2; break;
      case GWI_TERM_BETA: n_cols = 2; n_th = 2; break;
      case GWI_TERM_TILT_MIXTURE: n_th = 2; break;
      case GWI_TERM_POWERLAW_REDSHIFT: break;
      case GWI_TERM_TRUNCNORM: n_th = 2; break;
      case GWI_TERM_TILT_JOINT: n_cols = 2; n_th = 2; break;
      case GWI_TERM_SMOOTH: break;
      case GWI_TERM_PLPEAK_SMOOTH:
        n_cols = 2;
        n_th = 4;
        if (!theta_ok(tm.coef_off)) return fail(h, GWI_ERR_INVALID, "PLPEAK_SMOOTH: coef_off must be the theta index of delta");
        break;
      case GWI_TERM_POWERLAW_BOUNDS: n_cols = 2; n_th = 3; break;
      case GWI_TERM_EXP_SPLINE_LERP:
        if (tm.norm < 0 || tm.norm >= s->n_norms || !s->norms[tm.norm].us || s->norms[tm.norm].n_pts < 2)
          return fail(h, GWI_ERR_INVALID, "EXP_SPLINE_LERP: norm must name the grid normaliser carrying the grid's spline coordinates (us)");
        [[fallthrough]];
      case GWI_TERM_LINEAR_SPLINE:
      case GWI_TERM_EXP_SPLINE:
        n_th = 0;
        if (tm.n_basis < 4 || !theta_ok(tm.coef_off) || !theta_ok(tm.coef_off + tm.n_basis - 1)) return fail(h, GWI_ERR_INVALID, "spline coefficient range invalid");
        if (!(tm.p[1] > tm.p[0])) return fail(h, GWI_ERR_INVALID, "spline domain invalid");
        break;
      default: return fail(h, GWI_ERR_INVALID, "unknown term kind");
    }
    for (int c = 0; c < n_cols; ++c)
      if (!col_ok(tm.cols[c])) return fail(h, GWI_ERR_INVALID, "term column index out of range");
    for (int k = 0; k < n_th; ++k)
      if (!theta_ok(tm.theta[k])) return fail(h, GWI_ERR_INVALID, "term theta index out of range");
    if (tm.norm >= s->n_norms) return fail(h, GWI_ERR_INVALID, "term norm index out of range");
  }
  for (int j = 0; j < s->n_norms; ++j) {
    const gwi_norm& nm = s->norms[j];
    if (nm.n_pts < 2 || !nm.tw) return fail(h, GWI_ERR_INVALID, "normaliser grid invalid");
    if (nm.expo_theta >= 0 && (!theta_ok(nm.expo_theta) || !nm.l1)) return fail(h, GWI_ERR_INVALID, "normaliser exponent invalid");
    if (nm.n_basis > 0 && (nm.n_basis < 4 || !nm.us || !theta_ok(nm.coef_off) || !theta_ok(nm.coef_off + nm.n_basis - 1) || !(nm.hi > nm.lo)))
      return fail(h, GWI_ERR_INVALID, "normaliser spline invalid");
  }
  return GWI_OK;
}

// theta -> derived scalars of every term + the sample-independent log-normaliser total
void prelude(gwi_engine* h, const double* theta, double* theta_out, double (*derived_out)[kMaxDerived], double* host_const) {
  double c = 0.0;
  for (int t = 0; t < h->spec.n_terms; ++t) {
    const gwi_term& tm = h->spec.terms[t];
    double* d = derived_out[t];
    for (int i = 0; i < kMaxDerived; ++i) d[i] = 0.0;
    switch (tm.kind) {
      case GWI_TERM_POWERLAW: {
        if (tm.flags & GWI_POWERLAW_UNNORMALISED) break;  // bare x^alpha pairing factor
        double la, dla;
        powerlaw_lognorm(theta[tm.theta[0]], tm.p[0], tm.p[1], &la, &dla);
        c += la;
        break;
      }
      case GWI_TERM_POWERLAW_BOUNDS: {
        const double lo = theta[tm.theta[1]], hi = theta[tm.theta[2]];
        double la, dla;
        powerlaw_lognorm(theta[tm.theta[0]], lo, hi, &la, &dla);
        // at alpha == -1 exactly the reference's Powerlaw.log_prob subtracts log(max/min), not log(log(max/min))
        // (numpyro_distributions.py:130); reproduced, since the per-event sites show it (it cancels in log_l)
        if (theta[tm.theta[0]] == -1.0) la = -std::log(hi / lo);
        c += la;
        break;
      }
      case GWI_TERM_TILT_JOINT: {
        double dmu;
        truncnorm_lognorm(1.0, theta[tm.theta[1]], -1.0, 1.0, &d[0], &dmu, &d[1]);
        const double sg = theta[tm.theta[1]];
        d[2] = 1.0 / (sg * sg);
        d[3] = d[2] / sg;
        break;
      }
      case GWI_TERM_PLPEAK_SMOOTH:
      case GWI_TERM_PLPEAK: {
        powerlaw_lognorm(theta[tm.theta[0]], tm.p[0], tm.p[1], &d[0], &d[1]);
        truncnorm_lognorm(theta[tm.theta[1]], theta[tm.theta[2]], tm.p[0], tm.p[1], &d[2], &d[3], &d[4]);
        const double sg = theta[tm.theta[2]];
        d[5] = 1.0 / (sg * sg);
        d[6] = d[5] / sg;
        break;
      }
      case GWI_TERM_POWERLAW_RATIO: {
        const double b1 = 1.0 + theta[tm.theta[0]];
        d[0] = b1 != 0.0 ? 1.0 / b1 : 0.0;
        break;
      }
      case GWI_TERM_BETA: {
        const double a = theta[tm.theta[0]], b = theta[tm.theta[1]];
        c -= std::lgamma(a) + std::lgamma(b) - std::lgamma(a + b);  // betaln (distributions.py:161)
        break;
      }
      case GWI_TERM_TILT_MIXTURE: {
        double dmu;
        truncnorm_lognorm(1.0, theta[tm.theta[1]], -1.0, 1.0, &d[0], &dmu, &d[1]);
        const double sg = theta[tm.theta[1]];
        d[2] = 1.0 / (sg * sg);
        d[3] = d[2] / sg;
        break;
      }
      case GWI_TERM_TRUNCNORM: {
        double lc, dmu, dsg;
        truncnorm_lognorm(theta[tm.theta[0]], theta[tm.theta[1]], tm.p[0], tm.p[1], &lc, &dmu, &dsg);
        c += lc;
        const double sg = theta[tm.theta[1]];
        d[0] = 1.0 / (sg * sg);
        d[1] = d[0] / sg;
        break;
      }
      default: break;
    }
  }
  // A non-finite hyper-parameter makes every weight NaN in the reference, i.e. log_l = nan_to_num(-inf)
  // and a zero gradient (analysis.py:287-289).  The device exp clamps its argument (NaN -> 0), so the
  // case is decided here: a NaN constant routes assemble() down exactly that branch.
  bool theta_finite = true;
  for (int p = 0; p < h->spec.n_theta; ++p) theta_finite = theta_finite && std::isfinite(theta[p]);
  *host_const = theta_finite ? c : NAN;
  std::memcpy(theta_out, theta, sizeof(double) * h->spec.n_theta);
}

// Timing mode attaches a start/stop event pair to the launch itself (hipExtLaunchKernelGGL): the pair
// brackets the kernel's own begin/end as the dispatch reports it -- the quantity rocprofv3's kernel
// trace shows -- instead of stream positions around it, which add 1-3 us of launch gap per bracket.
template <typename F, typename A>
void launch_timed(gwi_handle h, int slot, F fn, dim3 grid, dim3 block, size_t lds, const A& args, size_t used_bytes = sizeof(A)) {
  if (h->aql_now) {  // slot 0 / 1 / 2 = scan / combine / final of the plain evaluation path
    const aql::Kernel& k = slot == 0 ? (h->scan_role == jit::kSafe ? h->aq_scan_safe : (h->scan_role == jit::kPbatch ? h->aq_scan_pbatch : (h->scan_role == jit::kBatch ? h->aq_scan_batch : h->aq_scan)))
                                     : (slot == 1 ? h->aq_combine : h->aq_final);
    const hsa_signal_t done = h->timing ? h->aq.done[slot] : hsa_signal_t{0};
    if (slot > 0 && h->aq_tail_args) {  // constant arguments, staged once at gwi_create
      char* staged = h->aql_tail_variant == 0 ? h->aq_tail_args : h->aq_tail_batch[h->aql_tail_variant - 1];
      // the combine launch reads the scan's records with cache-bypassing loads: no acquire fence (GWI_AQL_COMBINE_ACQUIRE=1: with)
      (void)aql::dispatch_staged(h->aq, k, staged, grid.x, grid.y, block.x, (uint32_t)lds, done, /*acquire=*/slot != 1 || h->combine_acquire);
      return;
    }
    if constexpr (std::is_same<A, ScanBlock>::value) {
      constexpr size_t kOff = offsetof(ScanBlock, k);  // the argument block sits behind the preloaded scalars
      if (!h->aql_tail_only) {  // GWI_AQL_TAIL=0: the whole block into a ring slot per launch (A/B only)
        (void)aql::dispatch(h->aq, k, &args, kOff + used_bytes, grid.x, grid.y, block.x, (uint32_t)lds, done);
        return;
      }
      // the scan's block: fixed head in place, only the per-evaluation tail through the BAR (aql::dispatch_tail)
      const size_t off_theta = kOff + offsetof(KArgs, theta);
      const size_t ranges[3][2] = {{kOff + offsetof(KArgs, norm_seq), offsetof(KArgs, derived) - offsetof(KArgs, norm_seq)},
                                   {kOff + offsetof(KArgs, derived), sizeof(double) * kMaxDerived * (size_t)h->spec.n_terms},
                                   {off_theta, kOff + used_bytes > off_theta ? kOff + used_bytes - off_theta : 0}};
      (void)aql::dispatch_tail(h->aq, k, h->tail_parity++, &args, kOff + offsetof(KArgs, norm_seq), kOff + used_bytes, ranges, 3, grid.x, grid.y, block.x, (uint32_t)lds, done);
      return;  // on a queue error nothing was submitted; the waiters surface it
    } else {
      if (aql::dispatch(h->aq, k, &args, used_bytes, grid.x, grid.y, block.x, (uint32_t)lds, done)) return;
      // the queue reported an error: nothing was submitted; the waiters surface it
      return;
    }
  }
  if constexpr (std::is_same<A, ScanBlock>::value) {
    const ScanHead& hd = args.head;
    if (h->variant->jit) {
      // a chain compiled at gwi_create: its kernel lives in a module, and the staged block IS the kernel-argument segment
      // (preloaded scalars + argument block: static_assert on ScanBlock above)
      size_t bytes = sizeof(ScanBlock);
      void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<ScanBlock*>(&args), HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END};
      hipFunction_t f = h->jit_fn[h->scan_role];
      if (h->timing)
        (void)hipExtModuleLaunchKernel(f, grid.x * block.x, grid.y, 1, block.x, 1, 1, lds, h->stream, nullptr, extra, h->ev[2 * slot], h->ev[2 * slot + 1], 0);
      else
        (void)hipModuleLaunchKernel(f, grid.x, grid.y, 1, block.x, 1, 1, (unsigned)lds, h->stream, nullptr, extra);
      return;
    }
    if (h->timing)
      hipExtLaunchKernelGGL(fn, grid, block, (unsigned)lds, h->stream, h->ev[2 * slot], h->ev[2 * slot + 1], 0, hd.col[0], hd.col[1], hd.col[2], hd.col[3], hd.col[4], hd.geom, hd.chunks, hd.n_pe, hd.n_inj, args.k);
    else
      hipLaunchKernelGGL(fn, grid, block, lds, h->stream, hd.col[0], hd.col[1], hd.col[2], hd.col[3], hd.col[4], hd.geom, hd.chunks, hd.n_pe, hd.n_inj, args.k);
  } else {
    if (h->timing)
      hipExtLaunchKernelGGL(fn, grid, block, (unsigned)lds, h->stream, h->ev[2 * slot], h->ev[2 * slot + 1], 0, args);
    else
      hipLaunchKernelGGL(fn, grid, block, lds, h->stream, args);
  }
}

// scan_pbatch_kernel takes single-trip tiles: it applies when the geometry of the launch being issued has them
bool pbatch_applies(const gwi_engine* h) {
  if (!h->pbatch) return false;
  const long long gran = (long long)pbatch_u(h->variant->samples_per_lane) * kBlock;
  return h->kargs.chunk_pe <= gran && h->kargs.chunk_inj <= gran;
}
// Points per grid row of a pbatch launch of K points.  One row (every sample loaded once for all K points) is the least work,
// but a tile x K points is a long workgroup: the rows are split until the launch has about eight workgroups per CU to balance
// (config 2, K = 16: 788 tiles on 256 CUs -- one row leaves a quarter of the chip idle behind the CUs that drew four tiles).
int pbatch_points(const gwi_engine* h, int K, int on_bgeo = -1) {
  int pts = h->pbatch_pts;
  if (pts <= 0) {
    const long long blocks = (on_bgeo < 0 ? h->use_bgeo : on_bgeo != 0) ? h->bgeo.n_scan_blocks : h->n_scan_blocks;
    int rows = 1;
    while (rows < K && blocks * rows < 8LL * h->n_cus) rows *= 2;
    pts = (K + rows - 1) / rows;
  }
  return std::max(1, std::min(pts, kPbatchMaxPts));
}

constexpr int kPbatchBalancedFrom = 8;  // points per launch from which the balanced mode is used
// Balanced mode: how many workgroups a pbatch launch of K points gets.  One round of resident workgroups, or the next smaller
// count with which every workgroup draws the same number of (tile, point) units: ceil(N / ceil(N / capacity)).
int pbatch_workgroups(const gwi_engine* h, int K, int on_bgeo = -1) {
  const long long blocks = (on_bgeo < 0 ? h->use_bgeo : on_bgeo != 0) ? h->bgeo.n_scan_blocks : h->n_scan_blocks;
  const long long n_units = blocks * K, capacity = (long long)h->n_cus * h->pbatch_wgs_per_cu;
  const long long per_wg = (n_units + capacity - 1) / capacity;
  return (int)((n_units + per_wg - 1) / per_wg);
}

gwi_status launch_scan(gwi_handle h, bool logw, int K = 1, bool batch = false) {
  if (logw) h->aql_now = false;  // the log-weight variant is another kernel and always goes through the HIP stream
  const int grid = (h->use_bgeo && !logw ? h->bgeo.n_scan_blocks : h->n_scan_blocks) + (logw ? 0 : h->spec.n_norms);  // the first n_norms workgroups integrate the normaliser grids
  // two-pass repeats and the replay mode run the SAFE instantiation (spline models; it takes single and batched launches)
  // ... and so does any replica count other than the 16 the regular kernels are built for (GWI_GACC_REP)
  const bool safe = !logw && h->variant->has(jit::kSafe) && (h->generic || h->kargs.two_pass || h->kargs.deterministic || h->gacc_rep != (1 << kRegularRepShift));
  // parametric models: a batched launch on single-trip tiles loads every sample once for all its points (scan_pbatch_kernel)
  // ... where a grid row holds more than one point: with one point per row (small catalogs: the rows are split until the launch
  // fills the chip) it has nothing to share and the one-row-per-point kernel is the same thing without the staging
  // (balanced mode from eight points on: below that the shared loads no longer pay for the staging -- config 2, K = 4: 18.5 us
  // against 17.4 with one grid row per point, K = 2: 12.8 against 11.3; profiles/round6/balanced_ab.txt)
  const bool balanced = h->pbatch_balanced && K >= kPbatchBalancedFrom;
  const bool pb = batch && !safe && !logw && pbatch_applies(h) && (balanced || pbatch_points(h, K) > 1);
  h->scan_role = logw ? jit::kLogw : (safe ? jit::kSafe : (pb ? jit::kPbatch : (batch ? jit::kBatch : jit::kScan)));
  ScanFn fn = h->variant->fn[h->scan_role];
  h->scan_is_safe = safe;
  h->kargs.k_batch = batch ? K : 1;
  if (pb) {
    const size_t used = offsetof(KArgs, theta);  // the points' hyper-parameters travel in their ThetaBlocks
    if (balanced) {
      const int n_wg = pbatch_workgroups(h, K);
      h->kargs.pbatch_pts = -n_wg;
      launch_timed(h, 0, fn, dim3(h->spec.n_norms * K + n_wg, 1), dim3(kBlock), 0, h->sblock, used);
      GWI_HIP(hipGetLastError());
      return GWI_OK;
    }
    const int pts = pbatch_points(h, K);
    h->kargs.pbatch_pts = pts;
    launch_timed(h, 0, fn, dim3(grid - h->spec.n_norms + h->spec.n_norms * K, (K + pts - 1) / pts), dim3(kBlock), 0, h->sblock, used);
    GWI_HIP(hipGetLastError());
    return GWI_OK;
  }
  if (batch && !safe && !logw) {
    h->batch_used_mfma = h->mfma && K >= h->mfma_min_batch;
    if (h->batch_used_mfma && h->mfma == h->jit_mfma && h->jit_mfma_fn) {  // ... the instantiation compiled at run time: a module launch, the argument block as the buffer
      size_t bytes = sizeof(KArgs);
      void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &h->kargs, HIP_LAUNCH_PARAM_BUFFER_SIZE, &bytes, HIP_LAUNCH_PARAM_END};
      const unsigned gy_m = (unsigned)((K + kPts - 1) / kPts);
      h->aql_now = false;
      if (h->timing)
        (void)hipExtModuleLaunchKernel(h->jit_mfma_fn, (unsigned)grid * kBlock, gy_m, 1, kBlock, 1, 1, h->mfma_lds_bytes, h->stream, nullptr, extra, h->ev[0], h->ev[1], 0);
      else
        (void)hipModuleLaunchKernel(h->jit_mfma_fn, (unsigned)grid, gy_m, 1, kBlock, 1, 1, (unsigned)h->mfma_lds_bytes, h->stream, nullptr, extra);
      GWI_HIP(hipGetLastError());
      return GWI_OK;
    }
    if (h->batch_used_mfma) {  // 16 points per wavefront: the grid's second dimension counts groups of 16
      launch_timed(h, 0, h->batch_rows ? h->mfma->rows_fn : h->mfma->fn, dim3(grid, (K + kPts - 1) / kPts), dim3(kBlock), h->mfma_lds_bytes, h->kargs, offsetof(KArgs, theta));
      GWI_HIP(hipGetLastError());
      return GWI_OK;
    }
  }
  // theta is the LAST member of the argument block: only the hyper-parameters in use travel through the BAR
  const size_t used = offsetof(KArgs, theta) + sizeof(double) * (size_t)h->spec.n_theta;
  launch_timed(h, 0, fn, dim3(grid, batch ? K : 1), dim3(kBlock), h->scan_lds_bytes, h->sblock, used);
  GWI_HIP(hipGetLastError());
  return GWI_OK;
}

gwi_status wait_for_stamp(gwi_handle h, double* host_buf, int K = 1);
void merge_final_records(gwi_handle h, int K);
gwi_status wait_for_rows(gwi_handle h, int K = 1);
gwi_status wait_for_norms(gwi_handle h, double* record, int K = 1);

// launches scan -> combine -> final; `record_dev` is where final_kernel publishes (pinned host record
// or, for the sharded path, the device send buffer); `wait` polls the pinned completion stamp.
// (Folding the two tail stages into the scan launch -- the workgroup that completes a group combines
// it -- was measured and lost: every workgroup then has to drain agent-scope write-through stores and
// wait for an atomic round trip across the XCDs' separate L2s, ~5 us of a resident slot each, more than
// the two launch boundaries cost.  See DESIGN.md.)
#ifdef GWI_HOST_PHASES
// diagnostic build only: accumulated host time per phase of run_pipeline [prelude, scan launch, combine
// launch, norm launch, wait] in seconds, and the call count
double g_phase[5] = {0, 0, 0, 0, 0};
long g_phase_calls = 0;
#define GWI_PHASE(i)                                                                       \
  do {                                                                                     \
    const auto now_ = std::chrono::steady_clock::now();                                    \
    g_phase[i] += std::chrono::duration<double>(now_ - phase_t_).count();                  \
    phase_t_ = now_;                                                                       \
  } while (0)
#else
#define GWI_PHASE(i) \
  do {               \
  } while (0)
#endif

// Arguments of the combine / final launches.  Nothing in them changes from one evaluation to the next (the completion
// stamp travels through a device word the scan writes): on the AQL path they sit in two persistent kernel-argument slots.
TailArgs tail_args(const gwi_engine* h, double* record_dev, bool batch_geometry = false) {
  TailArgs ta;
  std::memset(&ta, 0, sizeof(ta));
  ta.partials = h->d_partials;
  ta.ev_out = h->d_ev_out;
  ta.ev_grad = h->d_ev_grad;
  ta.inj_out = h->d_inj_out;
  ta.inj_grad = h->d_inj_grad;
  ta.ev_host = h->h_ev_dev;
  ta.host_rows = h->host_final && !record_dev ? h->h_rows_dev : nullptr;
  ta.record = record_dev ? record_dev : h->h_fin_dev;
  ta.final_groups = record_dev ? 1 : h->final_groups;  // the sharded path's single record feeds the all-gather
  ta.seq_ptr = h->d_seq;
  ta.redo_ptr = h->d_seq + 1;
  ta.n_ev = (int)h->n_ev;
  ta.tiles_per_event = batch_geometry ? h->bgeo.tiles_per_event : h->tiles_per_event;
  ta.n_inj_tiles = batch_geometry ? h->bgeo.n_inj_tiles : h->n_inj_tiles;
  ta.n_inj_groups = batch_geometry ? h->bgeo.n_inj_groups : h->n_inj_groups;
  ta.tiles_per_inj_group = batch_geometry ? h->bgeo.tiles_per_inj_group : h->tiles_per_inj_group;
  ta.n_theta = h->spec.n_theta;
  ta.rec_stride = h->rec_stride;
  ta.n_scan_blocks = batch_geometry ? h->bgeo.n_scan_blocks : h->n_scan_blocks;
  ta.n_norms = h->spec.n_norms;
  ta.record_len = record_len(h);
  ta.n_pe = (double)h->n_pe;
  ta.publish_events = 1;
  ta.combine_threads = h->combine_threads;
  ta.row_lines = (3 + h->spec.n_theta + 6) / 7;
  return ta;
}

// which launch geometry the scan's argument block describes: the batched one (bgeo) or the single evaluation's
void set_geometry(gwi_engine* h, bool batched) {
  h->use_bgeo = batched;
  h->kargs.tiles_per_event = batched ? h->bgeo.tiles_per_event : h->tiles_per_event;
  h->kargs.chunk_pe = batched ? h->bgeo.chunk_pe : h->chunk_pe;
  h->kargs.n_inj_tiles = batched ? h->bgeo.n_inj_tiles : h->n_inj_tiles;
  h->kargs.chunk_inj = batched ? h->bgeo.chunk_inj : h->chunk_inj;
  ScanHead& hd = h->sblock.head;
  hd.geom = (unsigned)h->kargs.n_ev | ((unsigned)h->kargs.tiles_per_event << kGeomEventBits) | ((unsigned)h->kargs.n_norms << (kGeomEventBits + kGeomTilesBits));
  hd.chunks = pack_chunk(h->kargs.chunk_pe) | (pack_chunk(h->kargs.chunk_inj) << 16);
}

// columns a term kind reads (= the doubles of its Term<K>::In): the order of the scan's preloaded column slots
int term_cols(int kind) {
  switch (kind) {
#define GWI_X(K) \
  case K: return (int)(sizeof(typename Term<K>::In) / sizeof(double));
    GWI_FOR_EACH_KIND(GWI_X)
#undef GWI_X
    default: return 0;
  }
}

gwi_status run_pipeline_once(gwi_handle h, const double* theta, double* record_dev, bool wait, int K, bool batch, bool square) {
#ifdef GWI_HOST_PHASES
  auto phase_t_ = std::chrono::steady_clock::now();
  ++g_phase_calls;
#endif
  const int n_theta = h->spec.n_theta;
  h->kargs.square = square ? 1 : 0;
  // tile references: row 0 belongs to single evaluations, rows 1..max_batch to the points of a batch on the single
  // evaluation's tiling, rows 1 + max_batch.. to the points of a batch on the batched launches' own tiling (bgeo: other tile
  // boundaries, so another tile's maximum).  Row k of a block keeps meaning "point k of the batch": a vectorised caller
  // should keep chain k in slot k from one call to the next (a reference left by another chain is still only a range
  // question -- a miss costs one repeat, never a wrong bit).
  h->kargs.nref_row0 = batch ? ((K >= 4 && h->bgeo.distinct) ? 1 + h->max_batch : 1) : 0;
  // plain evaluations go through the engine's AQL queue; whatever must be ordered with other work on the HIP stream
  // (batched theta uploads, the sharded path's exchange behind record_dev) stays on the stream, and so does everything
  // after gwi_set_timing(h, 2)
  // batched launches of the 4-tap kernel follow once their theta blocks can be written through the BAR (K >= 4: the
  // device-final form whose tail arguments are staged)
  const bool batch_on_aql = batch && K >= 4 && h->aql_batch && !h->kargs.two_pass && !h->kargs.deterministic && !(h->mfma && K >= h->mfma_min_batch);
  h->aql_now = h->aql_active && !h->aq.failed() && !h->force_hip_stream && record_dev == nullptr && ((!batch && K == 1) || batch_on_aql);
  h->scan_is_batch = batch;
  h->aql_tail_variant = batch_on_aql ? (h->batch_events ? 2 : 1) : 0;
  if (h->aql_now && h->timing && !aql::timed_prepare(h->aq)) h->aql_now = false;  // no dispatch timestamps: time this one with HIP events
  if ((int)h->host_consts.size() < K) h->host_consts.resize(K);
  if (!batch) {
    prelude(h, theta, h->kargs.theta, h->kargs.derived, &h->host_consts[0]);
    h->kargs.tblocks = nullptr;
  } else {
    for (int k = 0; k < K; ++k) prelude(h, theta + (size_t)k * n_theta, h->h_tblocks[k].theta, h->h_tblocks[k].derived, &h->host_consts[k]);
    if (h->aql_now) {
      // theta blocks straight into device memory through the BAR (only the parts in use: a few hundred bytes per point),
      // one hand-off for all of them; the scan packet that reads them is published afterwards
      ThetaBlock* dev = reinterpret_cast<ThetaBlock*>(aql::extra_area(h->aq));
      const size_t th_bytes = sizeof(double) * (size_t)n_theta, der_bytes = sizeof(double) * kMaxDerived * (size_t)h->spec.n_terms;
      for (int k = 0; k < K; ++k) {
        std::memcpy(dev[k].theta, h->h_tblocks[k].theta, th_bytes);
        std::memcpy(dev[k].derived, h->h_tblocks[k].derived, der_bytes);
      }
      // no hand-off of their own: the scan's argument block is written next and handed over with ONE sfence + read-back
      // (aql::dispatch_tail / stage_args), which retires these posted writes as well -- a read cannot pass any posted write
      // ahead of it, whatever its address (a second read-back here cost every batch 1.4 us)
      h->kargs.tblocks = dev;
    } else if (h->stage_kernel && K >= 10) {  // below ~20 KiB the runtime's small-copy path is quicker than a launch
      hipLaunchKernelGGL(stage_theta_kernel, dim3(K), dim3(kBlock), 0, h->stream, (const ThetaBlock*)h->h_tblocks_dev, h->d_tblocks);
      GWI_HIP(hipGetLastError());
    } else {
      GWI_HIP(hipMemcpyAsync(h->d_tblocks, h->h_tblocks, sizeof(ThetaBlock) * K, hipMemcpyHostToDevice, h->stream));
    }
    if (!h->aql_now) h->kargs.tblocks = h->d_tblocks;
  }
  const unsigned gy = batch ? (unsigned)K : 1u;
  // batched device-final launches run on their own geometry where gwi_create found one (two trips per workgroup)
  set_geometry(h, batch && K >= 4 && h->bgeo.distinct);
  TailArgs ta = tail_args(h, record_dev, h->use_bgeo);
  if (batch && K >= 4) {
    // Many points per launch: sum over groups on the DEVICE whatever the problem size.  Host-final mode publishes one row
    // per (group, point) -- K x (N_ev + injection groups) rows of two small posted PCIe writes each, which is what a
    // 16-point batch of config 2 spent a quarter of its time on (1216 rows) -- against one record per point here; the
    // per-event sites travel (three more small writes per event and point) only when the caller asked for them.
    ta.host_rows = nullptr;
    ta.publish_events = h->batch_events ? 1 : 0;
  }
  h->last_host_rows = ta.host_rows != nullptr;
  h->kargs.norm_seq = h->seq + 1;  // the normaliser workgroups of this launch stamp their results with it
  GWI_PHASE(0);
  gwi_status st = launch_scan(h, false, K, batch);
  if (st != GWI_OK) return st;
  GWI_PHASE(1);
  launch_timed(h, 1, combine_kernel, dim3((unsigned)(h->n_ev + ta.n_inj_groups), gy), dim3((unsigned)ta.combine_threads), 0, ta);
  GWI_HIP(hipGetLastError());
  ++h->seq;
  h->timed_final = false;
  GWI_PHASE(2);
  if (ta.host_rows) {
    GWI_PHASE(3);
    const gwi_status sw = wait ? wait_for_rows(h, K) : GWI_OK;
    GWI_PHASE(4);
    return sw;
  }
  launch_timed(h, 2, final_kernel, dim3((unsigned)ta.final_groups, gy), dim3(kFinalThreads), 0, ta);
  GWI_HIP(hipGetLastError());
  h->timed_final = true;
  if (!wait) return GWI_OK;
  gwi_status st_ = wait_for_stamp(h, h->h_fin, K * h->final_groups);
  if (st_ == GWI_OK) merge_final_records(h, K);
  if (st_ != GWI_OK) return st_;
  return wait_for_norms(h, h->h_record, K);
}

// A scan workgroup whose reference exponent turned out too far from its tile's true maximum (models with spline terms:
// scan_kernel, shared mode) has stored this evaluation's sequence number in the pinned redo word.
bool redo_requested(const gwi_engine* h) { return *reinterpret_cast<volatile unsigned long long*>(h->h_redo) == h->seq; }

// launches scan -> combine [-> final]; with `wait`, repeats the evaluation when a workgroup asked for it.  The failed attempt
// has left every tile's exact maximum in tile_nref, so the repeat is the same (fast) kernel with exact references and cannot
// miss; should it ever ask again (it never has) the two-pass instantiation finds the maxima in a sweep of its own.
// Callers that pass wait = false check redo_requested() themselves once their results are in (repeat_after_redo).
gwi_status repeat_after_redo(gwi_handle h, const double* theta, double* record_dev, int K, bool batch, bool square) {
  ++h->redo_count;
  gwi_status st = run_pipeline_once(h, theta, record_dev, true, K, batch, square);
  if (st == GWI_OK && redo_requested(h) && h->variant->has(jit::kSafe)) {
    h->kargs.two_pass = 1;
    st = run_pipeline_once(h, theta, record_dev, true, K, batch, square);
    h->kargs.two_pass = 0;
  }
  return st;
}
// The matrix-core batched kernel for a spline model that has no ahead-of-time instantiation of it: scan_mfma_kernel
// instantiated for this model's kinds and tile counts by hipRTC (gwi_jit.h), loaded as a module.  Returns whether h->mfma is set.
bool try_jit_mfma(gwi_handle h) {
  h->mfma_jit_pending = false;
  const gwi_spec& s = h->spec;
  int kts[GWI_MAX_TERMS], tiles_total = 0, row_doubles = 0;
  bool any_spline = false;
  for (int t = 0; t < s.n_terms; ++t) {
    const int kind = s.terms[t].kind;
    if (kind == GWI_TERM_EXP_SPLINE_LERP) {
      h->mfma_jit_note = "the interpolated-grid spline term has no matrix-core form";
      return false;
    }
    const bool spline = kind == GWI_TERM_EXP_SPLINE || kind == GWI_TERM_LINEAR_SPLINE;
    const int tiles = spline ? (s.terms[t].n_basis + 15) / 16 : 0;
    any_spline = any_spline || spline;
    kts[t] = kind + 100 * tiles;
    tiles_total += tiles;
    row_doubles += spline ? kSplineStage : term_cols(kind);
  }
  if (!any_spline || tiles_total > 8) {  // more than eight gradient tiles: the 4-tap kernel wins by arithmetic (DESIGN section 0 of round 4, row 4)
    h->mfma_jit_note = any_spline ? "more than eight 16-basis gradient tiles" : "no spline term";
    return false;
  }
  std::string why;
  jit::Chain* c = jit::get_chain(kts, s.n_terms, 0, gwi_embedded_device_h, gwi_embedded_engine_h, why, gwi_embedded_mfma_h);
  hipModule_t mod = c ? jit::module_on(c, h->device, why) : nullptr;
  if (c && !mod && c->from_cache) {
    jit::discard_chain(c);
    c = jit::get_chain(kts, s.n_terms, 0, gwi_embedded_device_h, gwi_embedded_engine_h, why, gwi_embedded_mfma_h);
    mod = c ? jit::module_on(c, h->device, why) : nullptr;
  }
  hipFunction_t fn = nullptr;
  if (mod && hipModuleGetFunction(&fn, mod, c->lowered[jit::kScan].c_str()) != hipSuccess) {
    why = "jit: hipModuleGetFunction(" + c->lowered[jit::kScan] + ") failed";
    (void)hipGetLastError();
    fn = nullptr;
  }
  int scratch = 0;
  if (fn && hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, fn) == hipSuccess && scratch > 0) {
    why = "the instantiation spills registers to scratch (" + std::to_string(scratch) + " B per lane)";
    fn = nullptr;
  }
  if (!fn) {
    h->mfma_jit_note = why;
    return false;
  }
  MfmaVariant* v = new MfmaVariant();
  std::memset(v, 0, sizeof(*v));
  v->name = c->name.c_str();
  v->n = s.n_terms;
  for (int t = 0; t < s.n_terms; ++t) {
    v->kinds[t] = kts[t] % 100;
    v->tiles[t] = kts[t] / 100;
  }
  v->row_doubles = row_doubles;
  h->jit_mfma = v;
  h->jit_mfma_fn = fn;
  h->mfma = v;
  h->batch_rows = false;
  h->mfma_lds_bytes = sizeof(double) * mfma_lds_doubles(s.n_theta, s.n_terms, row_doubles, 0);
  h->mfma_jit_note = "compiled " + c->name + (c->from_cache ? " (disk cache)" : "");
  return true;
}

// First batched launch of a spline model that has both batched kernels: three timed evaluation sets of each (after one untimed
// set each) on the caller's points, host theta -> host results as a sampler pays them; the faster kernel stays.
gwi_status run_pipeline(gwi_handle h, const double* theta, double* record_dev, bool wait, int K, bool batch, bool square);
gwi_status calibrate_batch_path(gwi_handle h, const double* theta, int K) {
  h->batch_autotune = false;
  const MfmaVariant* const both[2] = {h->mfma, nullptr};
  double best[2] = {1e30, 1e30};
  gwi_status st = GWI_OK;
  for (int rep = 0; rep < 4 && st == GWI_OK; ++rep)
    for (int which = 0; which < 2 && st == GWI_OK; ++which) {
      h->mfma = both[which];
      const auto t0 = std::chrono::steady_clock::now();
      st = run_pipeline(h, theta, nullptr, true, K, true, false);
      const double us = 1e6 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (rep > 0 && us < best[which]) best[which] = us;
    }
  h->batch_us[0] = best[0];
  h->batch_us[1] = best[1];
  h->batch_measured = st == GWI_OK;
  h->mfma = (st != GWI_OK || best[0] <= best[1]) ? both[0] : nullptr;
  return st;
}

gwi_status run_pipeline(gwi_handle h, const double* theta, double* record_dev = nullptr, bool wait = true, int K = 1, bool batch = false, bool square = false) {
  if (batch && h->mfma_jit_pending && K >= h->mfma_min_batch && wait && !record_dev) h->batch_autotune = try_jit_mfma(h) && h->autotune_wanted;
  if (batch && h->batch_autotune && h->mfma && K >= h->mfma_min_batch && wait && !record_dev) {
    const gwi_status sc = calibrate_batch_path(h, theta, K);
    if (sc != GWI_OK) return sc;
  }
  gwi_status st = run_pipeline_once(h, theta, record_dev, wait, K, batch, square);
  if (st == GWI_OK && wait && redo_requested(h)) st = repeat_after_redo(h, theta, record_dev, K, batch, square);
  return st;
}

// The AQL path has no HIP stream behind it: when the quick poll gives up, keep polling (with the queue's error flag in
// view) for up to 10 s, then report.  `ready` re-evaluates the completion condition.
template <typename Ready>
gwi_status aql_wait_slow(gwi_handle h, Ready ready, const char* what) {
  const auto t0 = std::chrono::steady_clock::now();
  bool rung_again = false;
  for (unsigned long long spin = 1;; ++spin) {
    if (ready()) {
      // (said once per process: with several processes time-slicing one GPU an evaluation can simply take longer than 50 ms)
      static std::atomic<bool> said{false};
      if (rung_again && !std::getenv("GWI_QUIET") && !said.exchange(true))
        std::fprintf(stderr, "gwi: waited more than 50 ms for %s and rang the queue's doorbell again as a precaution; they arrived (a busy or shared GPU does this too)\n", what);
      return GWI_OK;
    }
    // packets of this evaluation may still be in flight: the handle takes no further evaluations, and gwi_destroy
    // waits for the queue to drain (or leaks the buffers) instead of freeing memory a kernel may still write
    if (h->aq.failed()) {
      h->poisoned = true;
      return fail(h, GWI_ERR_HIP, h->aq.why());
    }
    __builtin_ia32_pause();
    if ((spin & 0xffff) == 0) {
      const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      hsa_queue_t* hq = h->aq.sq ? h->aq.sq->q : nullptr;
      if (waited > 0.05 && !rung_again && hq) {
        // 50 ms without a result is ~1000 evaluations' worth: if the command processor has not taken the packets (read index
        // behind the write index), ring the doorbell once more with the last index written -- a doorbell that did not register
        // would otherwise cost the whole evaluation; ringing twice with the same index is harmless
        aql::Api& api = aql::api();
        const uint64_t w = api.add_write_index(hq, 0), r = api.load_read_index(hq);
        if (r < w) api.signal_store(hq->doorbell_signal, (hsa_signal_value_t)(w - 1));
        rung_again = true;
        h->aql_rerings++;
      }
      if (waited > 10.0) {
        h->poisoned = true;
        std::string state;
        if (hq) {
          aql::Api& api = aql::api();
          state = " (queue write index " + std::to_string((unsigned long long)api.add_write_index(hq, 0)) + ", read index " + std::to_string((unsigned long long)api.load_read_index(hq)) +
                  ", evaluation " + std::to_string((unsigned long long)h->seq) + ", doorbell rung again: " + (rung_again ? "yes" : "no") + "; normaliser stamps";
          for (int j = 0; j < h->spec.n_norms && j < 16; ++j) state += " " + std::to_string((unsigned long long)h->h_norm_stamp[j]) + ":" + std::to_string(h->h_norm[j]);
          state += "; redo word " + std::to_string((unsigned long long)*reinterpret_cast<volatile unsigned long long*>(h->h_redo)) + "; kernel " + (h->variant ? h->variant->name : "?") +
                   ", role " + std::to_string(h->scan_role) + ", blocks " + std::to_string(h->n_scan_blocks) + ")";
        }
        return fail(h, GWI_ERR_TIMEOUT, std::string(what) + " did not arrive from the AQL queue within 10 s" + state);
      }
    }
  }
}

// the normaliser launch publishes Z_j + a stamp per normaliser; copy them into rank 0's record slots
gwi_status wait_for_norms(gwi_handle h, double* record, int K) {
  const int n = h->spec.n_norms;
  if (n == 0) return GWI_OK;
  const int total = n * K;
  bool done = false;
  for (long spin = 0; spin < 400000 && !done; ++spin) {
    done = true;
    for (int j = 0; j < total; ++j) done = done && *reinterpret_cast<volatile unsigned long long*>(h->h_norm_stamp + j) == h->seq;
    if (!done) __builtin_ia32_pause();
  }
  if (!done && h->aql_now) {
    const gwi_status sw = aql_wait_slow(h, [&] {
      for (int j = 0; j < total; ++j)
        if (*reinterpret_cast<volatile unsigned long long*>(h->h_norm_stamp + j) != h->seq) return false;
      return true;
    }, "normaliser stamps");
    if (sw != GWI_OK) return sw;
  } else if (!done) {
    GWI_HIP(hipStreamSynchronize(h->stream));
    for (int j = 0; j < total; ++j)
      if (h->h_norm_stamp[j] != h->seq) return fail(h, GWI_ERR_HIP, "normaliser stamp mismatch after stream synchronise");
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  const int len = record_len(h);
  for (int k = 0; k < K; ++k)
    for (int j = 0; j < n; ++j) record[(size_t)k * len + kRecNormOff + j] = h->h_norm[k * n + j];
  return GWI_OK;
}

// Device-final mode: the G partial records of point k (h_fin) -> one record (h_record), merged exactly as assemble() merges the
// records of ranks: sums, the minimum, the injection triples brought to their common exponent, gradients in workgroup order.
void merge_final_records(gwi_handle h, int K) {
  const int G = h->final_groups, n_theta = h->spec.n_theta, n_norms = h->spec.n_norms;
  const size_t len = (size_t)record_len(h);
  for (int k = 0; k < K; ++k) {
    double* out = h->h_record + (size_t)k * len;
    const double* in = h->h_fin + (size_t)k * G * len;
    double sum = 0.0, var = 0.0, mn = INFINITY, n_ev = 0.0, M = -INFINITY;
    bool redo = false;
    for (int g = 0; g < G; ++g) {
      const double* r = in + (size_t)g * len;
      sum += r[1];
      var += r[2];
      mn = std::fmin(mn, r[3]);
      M = std::fmax(M, r[4]);
      redo = redo || r[7] < 0.0;
      n_ev += r[7] < 0.0 ? -r[7] - 1.0 : r[7];
    }
    double S1 = 0.0, S2 = 0.0;
    double* gpe = out + kRecNormOff + n_norms;
    double* ginj = gpe + n_theta;
    for (int p = 0; p < n_theta; ++p) gpe[p] = ginj[p] = 0.0;
    for (int g = 0; g < G; ++g) {
      const double* r = in + (size_t)g * len;
      const double f = (r[4] == -INFINITY) ? 0.0 : std::exp(r[4] - M);
      S1 += f * r[5];
      S2 += f * f * r[6];
      const double* rp = r + kRecNormOff + n_norms;
      const double* ri = rp + n_theta;
      for (int p = 0; p < n_theta; ++p) {
        gpe[p] += rp[p];
        ginj[p] += f * ri[p];
      }
    }
    out[1] = sum;
    out[2] = var;
    out[3] = mn;
    out[4] = M;
    out[5] = S1;
    out[6] = S2;
    out[7] = redo ? -(n_ev + 1.0) : n_ev;
  }
}

gwi_status wait_for_stamp(gwi_handle h, double* host_buf, int K) {
  // Completion: final_kernel stores the sequence stamp into pinned host memory LAST (system-scope
  // release after __threadfence_system), so the host can poll it instead of paying a stream
  // synchronise; after ~2 ms of polling fall back to the blocking call (and surface any error).
  const size_t len = (size_t)record_len(h);
  auto stamp_of = [&](int k) { return *reinterpret_cast<volatile unsigned long long*>(host_buf + (size_t)k * len); };
  bool done = false;
  if (!h->timing && h->spin_wait) {
    int k = 0;
    for (long spin = 0; spin < 400000 && !done; ++spin) {
      while (k < K && stamp_of(k) == h->seq) ++k;
      done = k == K;
      if (!done) __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  if (!done && h->aql_now) {
    const gwi_status sw = aql_wait_slow(h, [&] {
      for (int k = 0; k < K; ++k)
        if (stamp_of(k) != h->seq) return false;
      return true;
    }, "the completion stamp");
    if (sw != GWI_OK) return sw;
    std::atomic_thread_fence(std::memory_order_acquire);
  } else if (!done) {
    GWI_HIP(hipStreamSynchronize(h->stream));
    for (int k = 0; k < K; ++k)
      if (stamp_of(k) != h->seq) return fail(h, GWI_ERR_HIP, "completion stamp mismatch after stream synchronise");
  }
  if (h->timing && h->aql_now) {
    h->last_ms[2] = 0.0f;
    if (!aql::timed_collect(h->aq, h->timed_final ? 3 : 2, h->last_ms)) return fail(h, GWI_ERR_HIP, "dispatch timestamps of the AQL queue are not available");
  } else if (h->timing) {
    GWI_HIP(hipStreamSynchronize(h->stream));
    h->last_ms[2] = 0.0f;  // host-final mode has no third launch
    for (int i = 0; i < (h->timed_final ? 3 : 2); ++i) GWI_HIP(hipEventElapsedTime(&h->last_ms[i], h->ev[2 * i], h->ev[2 * i + 1]));
  }
  return GWI_OK;
}

// Assemble the sites of analysis.py:259-319 from gathered per-rank records.
// `records_sq` (same layout, from a pass run with KArgs::square): its injection slots hold the sums
// weighted by w^2, which the gradient of the marginalised selection term needs; nullptr otherwise.
void assemble(const gwi_engine* h, const double* records, int n_ranks, const gwi_options* opt, gwi_summary* out, double* grad, double* norms,
              double host_const, const double* records_sq = nullptr) {
  const int n_theta = h->spec.n_theta, n_norms = h->spec.n_norms;
  const int len = record_len(h);
  const double NEG_BIG = -1.7976931348623157e308;  // jnp.nan_to_num(-inf)
  double sum_lse = 0.0, sum_var = 0.0, min_lneff = INFINITY, n_ev_total = 0.0, M = -INFINITY;
  for (int r = 0; r < n_ranks; ++r) {
    const double* rec = records + (size_t)r * len;
    sum_lse += rec[1];
    sum_var += rec[2];
    min_lneff = std::fmin(min_lneff, rec[3]);
    M = std::fmax(M, rec[4]);
    n_ev_total += rec[7];
  }
  double S1 = 0.0, S2 = 0.0;
  std::vector<double> g_pe(n_theta, 0.0), g_inj(n_theta, 0.0);
  for (int r = 0; r < n_ranks; ++r) {
    const double* rec = records + (size_t)r * len;
    const double f = (rec[4] == -INFINITY) ? 0.0 : std::exp(rec[4] - M);
    S1 += f * rec[5];
    S2 += f * f * rec[6];
    const double* gp = rec + kRecNormOff + n_norms;
    const double* gi = gp + n_theta;
    for (int p = 0; p < n_theta; ++p) {
      g_pe[p] += gp[p];
      g_inj[p] += f * gi[p];
    }
  }
  const double* nrm = records + kRecNormOff;  // every rank integrates the same grids
  double log_const = host_const;
  for (int t = 0; t < h->spec.n_terms; ++t)
    if (h->spec.terms[t].norm >= 0) log_const -= std::log(nrm[h->spec.terms[t].norm]);
  if (norms)
    for (int j = 0; j < n_norms; ++j) norms[j] = nrm[j];

  const double n_obs = opt->n_obs, n_tot = opt->total_inj, n_pe = (double)h->n_pe;
  gwi_summary s;
  std::memset(&s, 0, sizeof(s));
  s.log_norm_const = log_const;
  s.sum_logBFs = sum_lse + n_ev_total * (log_const - std::log(n_pe));
  // detection_efficiency (analysis.py:124-136), scale-free forms of var and n_eff
  const double log_mu = std::log(S1) + M - std::log(n_tot) + log_const;
  const double log_neff_inj = 2.0 * std::log(S1) - std::log(S2 - S1 * S1 / n_tot);
  const double var_mu = 1.0 / std::exp(log_neff_inj) - 1.0 / n_tot;
  s.log_det_eff = log_mu;
  s.log_nEff_inj = log_neff_inj;
  s.variance_log_detection_efficiency = var_mu;
  s.min_log_nEff = min_lneff;
  s.surveyed_hypervolume_norm = h->spec.vt_norm >= 0 ? nrm[h->spec.vt_norm] : NAN;
  double lde = log_mu;
  if (opt->marginalize_selection) lde = lde - (3.0 + n_obs) / (2.0 * std::exp(log_neff_inj));  // :271
  bool cut = false;
  if (opt->min_neff_cut && !(log_neff_inj >= std::log(4.0 * n_obs))) lde = INFINITY;  // :273-277
  s.selection_factor = std::isinf(lde) ? NEG_BIG : -n_obs * lde;                       // :278-281
  if (std::isinf(lde)) cut = true;
  double log_l = s.selection_factor + s.sum_logBFs;
  if (std::isnan(log_l)) {
    log_l = NEG_BIG;  // :287-288
    cut = true;
  } else if (std::isinf(log_l)) {
    log_l = log_l > 0 ? 1.7976931348623157e308 : NEG_BIG;  // nan_to_num :289
    cut = true;
  }
  s.log_l = log_l;
  if (opt->min_neff_cut) {
    const double min_neff = std::exp(min_lneff);  // :295
    if (min_neff <= n_obs) {                      // :296-303
      log_l = NEG_BIG;
      cut = true;
    }
  }
  s.variance_log_likelihood = n_obs * n_obs * var_mu + sum_var;  // :305-308
  if (opt->max_variance_cut && !(s.variance_log_likelihood <= 1.0)) {  // :309-317
    log_l = NEG_BIG;
    cut = true;
  }
  s.log_likelihood = log_l;
  if (out) *out = s;
  if (grad) {
    // d log_l / d theta = sum_i sum_j s_ij dl_ij - N_obs sum_j s_j dl_j  (SURVEY.md appendix A);
    // a cut replaces log_l by a constant, whose gradient is zero.
    for (int p = 0; p < n_theta; ++p) grad[p] = cut ? 0.0 : g_pe[p] - n_obs * (S1 > 0 ? g_inj[p] / S1 : 0.0);
    if (opt->marginalize_selection && records_sq && !cut && S1 > 0) {
      // lde = log mu - c / n_eff, c = (3 + N_obs)/2 (analysis.py:271);  log n_eff = 2 log S1 - log V,
      // V = S2 - S1^2/N_tot;  dS1 = G (sum w dl), dS2 = 2 H (H = sum w^2 dl, from the squared pass)
      double M2 = -INFINITY;
      for (int r = 0; r < n_ranks; ++r) M2 = std::fmax(M2, records_sq[(size_t)r * len + 4]);
      std::vector<double> H(n_theta, 0.0);
      for (int r = 0; r < n_ranks; ++r) {
        const double* rec = records_sq + (size_t)r * len;
        const double f2 = (rec[4] == -INFINITY) ? 0.0 : std::exp(rec[4] - M2);
        const double* gi = rec + kRecNormOff + n_norms + n_theta;
        for (int p = 0; p < n_theta; ++p) H[p] += f2 * gi[p];
      }
      // S1, S2 and g_inj are in units of e^M (e^2M for S2), H in units of e^M2: each pass reports the exponent its own records
      // were brought to, and nothing makes M2 equal 2 M (the two passes normalise their tile records independently)
      const double h_scale = (M2 == -INFINITY || M == -INFINITY) ? 0.0 : std::exp(M2 - 2.0 * M);
      const double V = S2 - S1 * S1 / n_tot;
      const double c_over_neff = (3.0 + n_obs) / (2.0 * std::exp(log_neff_inj));
      for (int p = 0; p < n_theta; ++p) {
        const double dlog_neff = 2.0 * g_inj[p] / S1 - (2.0 * h_scale * H[p] - 2.0 * S1 * g_inj[p] / n_tot) / V;
        grad[p] -= n_obs * c_over_neff * dlog_neff;
      }
    }
  }
}

// Host-final mode: poll every group's stamp, then do what final_kernel does (fixed summation order)
// into h_record so that assemble() is shared with the device-final and sharded paths.
gwi_status wait_for_rows(gwi_handle h, int K) {
  const int n_groups = (int)h->n_ev + h->n_inj_groups;
  const int n_theta = h->spec.n_theta, n_lines = (3 + n_theta + 6) / 7, stride = 8 * n_lines;
  const size_t total_lines = (size_t)n_groups * K * n_lines;
  // every 64-byte line carries the evaluation's sequence number in its eighth slot
  auto line_ok = [&](size_t ln) { return *reinterpret_cast<volatile unsigned long long*>(h->h_rows + ln * 8 + 7) == h->seq; };
  bool done = false;
  if (!h->timing && h->spin_wait) {
    size_t g = 0;
    for (long spin = 0; spin < 400000 && !done; ++spin) {
      while (g < total_lines && line_ok(g)) ++g;
      done = g == total_lines;
      if (!done) __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  auto all_ok = [&] {
    for (size_t g = 0; g < total_lines; ++g)
      if (!line_ok(g)) return false;
    return true;
  };
  if (!done && h->aql_now) {
    const gwi_status sw = aql_wait_slow(h, all_ok, "group result rows");
    if (sw != GWI_OK) return sw;
    std::atomic_thread_fence(std::memory_order_acquire);
  } else if (!done) {
    GWI_HIP(hipStreamSynchronize(h->stream));
    // the lines are posted writes behind the stream's completion: give the last ones a moment to land
    for (long spin = 0; spin < 4000000 && !all_ok(); ++spin) __builtin_ia32_pause();
    if (!all_ok()) return fail(h, GWI_ERR_HIP, "group row stamp mismatch after stream synchronise");
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  if (h->timing && h->aql_now) {
    h->last_ms[2] = 0.0f;
    if (!aql::timed_collect(h->aq, h->timed_final ? 3 : 2, h->last_ms)) return fail(h, GWI_ERR_HIP, "dispatch timestamps of the AQL queue are not available");
  } else if (h->timing) {
    GWI_HIP(hipStreamSynchronize(h->stream));
    h->last_ms[2] = 0.0f;  // host-final mode has no third launch
    for (int i = 0; i < (h->timed_final ? 3 : 2); ++i) GWI_HIP(hipEventElapsedTime(&h->last_ms[i], h->ev[2 * i], h->ev[2 * i + 1]));
  }
  const int n_ev = (int)h->n_ev, n_norms = h->spec.n_norms, len = record_len(h);
  for (int k = 0; k < K; ++k) {
    double* r = h->h_record + (size_t)k * len;
    const double* rows = h->h_rows + (size_t)k * n_groups * stride;
    // a group's row unpacked from its 64-byte lines (seven values + the sequence number each) into row[0 .. 3 + n_theta):
    // the sums below then run over contiguous values (index arithmetic per value -- i / 7, i % 7 -- cost config 3, 79 rows
    // of 56 values, 1.4 us per evaluation)
    double row[3 + GWI_MAX_THETA + 7];
    auto unpack = [&](int g) {
      const double* src = rows + (size_t)g * stride;
      for (int l = 0; l < n_lines; ++l) std::memcpy(row + 7 * l, src + 8 * l, 7 * sizeof(double));
    };
    double* ev = h->h_ev + (size_t)k * 3 * n_ev;
    double sum = 0.0, var = 0.0, mn = INFINITY;
    double* gpe = r + kRecNormOff + n_norms;
    double* ginj = gpe + n_theta;
    for (int p = 0; p < n_theta; ++p) gpe[p] = ginj[p] = 0.0;
    for (int e = 0; e < n_ev; ++e) {
      unpack(e);
      const double lse = row[0], lneff = row[1], v = row[2];
      sum += lse;
      var += v;
      double le = lneff;  // jnp.min(jnp.nan_to_num(logn_effs)) (analysis.py:295)
      if (le != le) le = 0.0;
      le = std::fmin(std::fmax(le, -1.7976931348623157e308), 1.7976931348623157e308);
      mn = std::fmin(mn, le);
      ev[e] = lse;
      ev[n_ev + e] = lneff;
      ev[2 * n_ev + e] = v;
      for (int p = 0; p < n_theta; ++p) gpe[p] += row[3 + p];
    }
    double M = -INFINITY;
    for (int j = 0; j < h->n_inj_groups; ++j) M = std::fmax(M, rows[(size_t)(n_ev + j) * stride]);
    double S1 = 0.0, S2 = 0.0;
    for (int j = 0; j < h->n_inj_groups; ++j) {
      unpack(n_ev + j);
      const double mj = row[0];
      const double f = (mj == -INFINITY) ? 0.0 : std::exp(mj - M);
      S1 += f * row[1];
      S2 += f * f * row[2];
      for (int p = 0; p < n_theta; ++p) ginj[p] += f * row[3 + p];
    }
    r[1] = sum;
    r[2] = var;
    r[3] = mn;
    r[4] = M;
    r[5] = S1;
    r[6] = S2;
    r[7] = (double)n_ev;
  }
  return wait_for_norms(h, h->h_record, K);
}

void destroy_impl(gwi_engine* h) {
  if (!h) return;
  struct VariantGuard {  // the variant record of a run-time compiled chain goes with the handle (the chain itself is process-wide)
    Variant* v;
    ~VariantGuard() { delete v; }
  } guard{h->jit_variant};
  h->jit_variant = nullptr;
  struct MfmaGuard {
    MfmaVariant* v;
    ~MfmaGuard() { delete v; }
  } mguard{h->jit_mfma};
  h->jit_mfma = nullptr;
  if (h->host_only) {
    if (h->shm_base) munmap(h->shm_base, h->shm_bytes);
    delete h;
    return;
  }
  (void)hipSetDevice(h->device);
  if (h->shm_base) munmap(h->shm_base, h->shm_bytes);
  if (h->poisoned && !aql::drain(h->aq, 2.0)) {
    // a kernel of the timed-out evaluation may still be running: leak the device buffers rather than free them under it
    aql::abandon_queue(h->aq);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return;
  }
  if (h->poisoned) (void)hipDeviceSynchronize();
  aql::close_queue(h->aq);
  for (double* p : h->d_cols_pe) (void)hipFree(p);  // each holds both sample sets (alloc_pair); d_cols_inj are interior pointers
  for (double* p : h->d_norm_arrays) (void)hipFree(p);
  (void)hipFree(h->d_norms);
  (void)hipFree(h->d_partials);
  (void)hipFree(h->d_ev_out);
  (void)hipFree(h->d_ev_grad);
  (void)hipFree(h->d_inj_out);
  (void)hipFree(h->d_inj_grad);
  (void)hipFree(h->d_logw_pe);
  (void)hipFree(h->d_logw_inj);
  if (h->nccl_comm && g_nccl.CommDestroy) (void)g_nccl.CommDestroy(h->nccl_comm);
  (void)hipFree(h->d_send);
  (void)hipFree(h->d_recv);
  if (h->h_gather) (void)hipHostFree(h->h_gather);
  if (h->h_record) (void)hipHostFree(h->h_record);
  if (h->h_fin) (void)hipHostFree(h->h_fin);
  if (h->h_ev) (void)hipHostFree(h->h_ev);
  if (h->h_rows) (void)hipHostFree(h->h_rows);
  if (h->h_norm) (void)hipHostFree(h->h_norm);
  if (h->h_norm_stamp) (void)hipHostFree(h->h_norm_stamp);
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  (void)hipFree(h->d_tblocks);
  (void)hipFree(h->d_seq);
  (void)hipFree(h->d_tile_nref);
  if (h->h_redo) (void)hipHostFree(h->h_redo);
  if (h->h_tblocks) (void)hipHostFree(h->h_tblocks);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

gwi_status upload(gwi_handle h, const double* src, size_t n, double** dst, std::vector<double*>* keep) {
  double* d = nullptr;
  GWI_HIP(hipMalloc(&d, sizeof(double) * (n ? n : 1)));
  keep->push_back(d);
  if (n) GWI_HIP(hipMemcpy(d, src, sizeof(double) * n, hipMemcpyHostToDevice));
  *dst = d;
  return GWI_OK;
}

// One allocation per column for BOTH sample sets: the posterior samples first, the injections inj_offset() elements behind
// (a 256-byte boundary).  The scan addresses either set through the same pointer (ScanHead, gwi_device.h).  d_cols_pe owns
// the allocations; d_cols_inj holds the interior pointers.
gwi_status alloc_pair(gwi_handle h, double** dpe, double** dinj) {
  double* d = nullptr;
  const size_t n = (size_t)h->inj_off + (size_t)(h->n_inj ? h->n_inj : 1);
  GWI_HIP(hipMalloc(&d, sizeof(double) * n));
  h->d_cols_pe.push_back(d);
  h->d_cols_inj.push_back(d + h->inj_off);
  *dpe = d;
  *dinj = d + h->inj_off;
  return GWI_OK;
}

// Every evaluating entry point: the handle must not hold an uncollected gwi_eval_begin (its kernel arguments, sequence
// stamp and dispatch path would be overwritten under the evaluation in flight) and must not be poisoned by a time-out.
gwi_status busy_guard(gwi_handle h, const char* who) {
  if (h->poisoned) return fail(h, GWI_ERR_INVALID, std::string(who) + ": an earlier evaluation of this handle timed out or its queue failed; destroy the handle");
  if (h->pending) return fail(h, GWI_ERR_INVALID, std::string(who) + ": an evaluation begun with gwi_eval_begin has not been collected (gwi_eval_end)");
  return GWI_OK;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int32_t gwi_abi_version(void) { return GWI_ABI_VERSION; }
int32_t gwi_kernel_variants(void) { return kNumVariants; }
const char* gwi_kernel_variant_name(int32_t i) { return (i >= 0 && i < kNumVariants) ? kVariants[i].name : nullptr; }
const char* gwi_scan_kernel_name(gwi_handle h) { return (h && h->variant) ? h->variant->name : "none"; }

const char* gwi_last_error(gwi_handle h) {
  static const char* none = "";
  static const char* null_handle = "null handle (gwi_create failed before an engine existed: no HIP device?)";
  if (!h) return null_handle;
  return h->err.empty() ? none : h->err.c_str();
}

void gwi_destroy(gwi_handle h) { destroy_impl(h); }

// The engine's AQL queue (gwi_aql.h).  Never fatal: on any failure the note says why and the HIP stream is used.
static void setup_aql(gwi_engine* h, const hipDeviceProp_t& prop) {
  h->aql_active = false;
  if (const char* env = std::getenv("GWI_AQL"))
    if (std::atoi(env) == 0) {
      h->aql_note = "disabled by GWI_AQL=0";
      return;
    }
  std::string code = "gwi_kernels.hsaco";
  if (const char* env = std::getenv("GWI_AQL_CODE")) {
    code = env;
  } else {
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&gwi_abi_version), &info) && info.dli_fname) {
      const std::string so = info.dli_fname;
      const size_t cut = so.find_last_of('/');
      code = (cut == std::string::npos ? std::string(".") : so.substr(0, cut)) + "/gwi_kernels.hsaco";
    }
  }
  aql::Device* dev = aql::open_device((uint32_t)prop.pciDomainID, (uint32_t)prop.pciBusID, (uint32_t)prop.pciDeviceID, 0u, code);
  if (!dev->ok) {
    h->aql_note = dev->why;
    return;
  }
  // the scan kernels of a chain compiled at gwi_create come out of its own code object, loaded into a second executable
  hsa_executable_t jit_exe{};
  jit::Chain* jc = h->variant->jit;
  if (jc) {
    std::lock_guard<std::mutex> lock(jc->mu);
    bool have = false;
    for (auto& kv : jc->hsa_executables)
      if (kv.first == dev) jit_exe.handle = kv.second, have = true;
    if (!have) {
      if (!aql::load_code(dev, jc->code.data(), jc->code.size(), jit_exe, h->aql_note)) return;
      jc->hsa_executables.emplace_back(dev, jit_exe.handle);
    }
  }
  auto find_scan = [&](int role, aql::Kernel& out) {
    if (!h->variant->has(role)) return false;
    if (jc) return aql::find_kernel(dev, jc->lowered[role].c_str(), out, h->aql_note, &jit_exe);
    return aql::find_kernel(dev, hipKernelNameRefByPtr(reinterpret_cast<const void*>(h->variant->fn[role]), h->stream), out, h->aql_note);
  };
  if (!find_scan(jit::kScan, h->aq_scan)) return;
  if (!aql::find_kernel(dev, hipKernelNameRefByPtr(reinterpret_cast<const void*>(&combine_kernel), h->stream), h->aq_combine, h->aql_note)) return;
  if (!aql::find_kernel(dev, hipKernelNameRefByPtr(reinterpret_cast<const void*>(&final_kernel), h->stream), h->aq_final, h->aql_note)) return;
  if (h->variant->has(jit::kSafe) && !find_scan(jit::kSafe, h->aq_scan_safe)) return;
  if (h->aq_scan.kernarg_bytes != sizeof(ScanBlock) || h->aq_combine.kernarg_bytes != sizeof(TailArgs) || h->aq_final.kernarg_bytes != sizeof(TailArgs)) {
    h->aql_note = "kernel argument sizes of the code object differ from this build (stale gwi_kernels.hsaco?)";
    return;
  }
  bool have_batch_kernel = find_scan(jit::kBatch, h->aq_scan_batch);
  // the batched launches of a parametric model are pbatch launches wherever their tiles are single trips: both kernels or neither
  h->aq_have_pbatch = h->variant->has(jit::kPbatch) && find_scan(jit::kPbatch, h->aq_scan_pbatch);
  if (h->pbatch && !h->aq_have_pbatch) have_batch_kernel = false;
  if (!aql::open_queue(dev, h->aq, h->aql_note)) return;
  if (have_batch_kernel && sizeof(ThetaBlock) * (size_t)h->max_batch <= aql::kExtraBytes) {
    for (int ev = 0; ev < 2; ++ev) {
      TailArgs tb = tail_args(h, nullptr, h->bgeo.distinct);
      tb.host_rows = nullptr;
      tb.publish_events = ev;
      h->aq_tail_batch[ev] = aql::stage_args(h->aq, aql::kSlots - 2 - ev, &tb, sizeof(tb));
    }
    h->aql_batch = h->aq_tail_batch[0] && h->aq_tail_batch[1];
    if (const char* env = std::getenv("GWI_AQL_BATCH")) h->aql_batch = h->aql_batch && std::atoi(env) != 0;
  }
  {
    const TailArgs ta = tail_args(h, nullptr);  // the AQL path never publishes to a device record (that is the RCCL exchange, on the HIP stream)
    h->aq_tail_args = aql::stage_args(h->aq, aql::kSlots - 1, &ta, sizeof(ta));
    if (!h->aq_tail_args) {
      h->aql_note = "staging the tail kernels' arguments failed";
      return;
    }
  }
  if (const char* env = std::getenv("GWI_AQL_TAIL")) h->aql_tail_only = std::atoi(env) != 0;
  if (const char* env = std::getenv("GWI_AQL_COMBINE_ACQUIRE")) h->combine_acquire = std::atoi(env) != 0;
  h->aql_active = true;
  h->aql_note = "active";
}

// gwi_create / gwi_create_ingest: the columns either come from the host ready-made (pe_cols / inj_cols) or are computed
// on the device from raw sources by the two setup programs (gwi_ingest.h)
static gwi_status create_impl(const gwi_spec* spec, const double* const* pe_cols, int64_t n_ev, int64_t n_pe, const double* const* inj_cols, int64_t n_inj,
                              int32_t device, gwi_handle* out, const gwi_ingest_program* ing_pe, const gwi_ingest_program* ing_inj) {
  if (!out) return GWI_ERR_INVALID;
  *out = nullptr;
  const bool ingest = ing_pe != nullptr;
  if (!spec || n_ev < 0 || n_pe < 1 || n_inj < 0) return GWI_ERR_INVALID;
  if (ingest ? !ing_inj : (!pe_cols || !inj_cols)) return GWI_ERR_INVALID;
  int n_dev = 0;
  if (device != GWI_DEVICE_HOST_ONLY && (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1)) return GWI_ERR_NO_DEVICE;
  gwi_engine* h = new (std::nothrow) gwi_engine();
  if (!h) return GWI_ERR_INVALID;
  *out = h;  // returned even on failure so gwi_last_error() can explain; caller must gwi_destroy()
  gwi_status st = validate_spec(h, spec);
  if (st != GWI_OK) return st;
  h->spec = *spec;
  if (device == GWI_DEVICE_HOST_ONLY) {
    h->host_only = true;
    h->n_ev = n_ev;
    h->n_pe = n_pe;
    h->n_inj = n_inj;
    for (int j = 0; j < spec->n_norms; ++j) h->spec.norms[j].tw = h->spec.norms[j].lb = h->spec.norms[j].l1 = h->spec.norms[j].us = nullptr;
    std::memset(&h->kargs, 0, sizeof(h->kargs));
    return GWI_OK;
  }
  if (device < 0) {
    GWI_HIP(hipGetDevice(&h->device));
  } else {
    if (device >= n_dev) return fail(h, GWI_ERR_NO_DEVICE, "device index out of range");
    h->device = device;
  }
  GWI_HIP(hipSetDevice(h->device));
  hipDeviceProp_t prop;
  GWI_HIP(hipGetDeviceProperties(&prop, h->device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(h, GWI_ERR_NO_DEVICE, std::string("engine is built for gfx950 only; device reports ") + prop.gcnArchName);
  h->n_cus = prop.multiProcessorCount;
  // ---- the scan kernel of this product of terms: an ahead-of-time chain (kVariants), else a chain compiled now for exactly
  // this sequence (gwi_jit.h: hipRTC, cached on disk), else -- hipRTC missing, GWI_JIT=0 -- the generic kernel
  auto env_on = [](const char* name) {
    const char* e = std::getenv(name);
    return e && std::atoi(e) != 0;
  };
  const bool force_generic = env_on("GWI_FORCE_GENERIC");  // tests / measurements: the generic kernel for a model that has a compiled chain
  const bool force_jit = env_on("GWI_FORCE_JIT");          // ... a run-time compiled chain for a model that has an ahead-of-time one
  h->variant = (force_generic || force_jit) ? nullptr : find_variant(*spec);
  bool jit_allowed = !force_generic;
  if (const char* env = std::getenv("GWI_JIT")) jit_allowed = jit_allowed && std::atoi(env) != 0;
  if (!h->variant && jit_allowed) {
    int kinds[GWI_MAX_TERMS], n_spline = 0;
    for (int t = 0; t < spec->n_terms; ++t) {
      kinds[t] = spec->terms[t].kind;
      n_spline += jit::is_spline_kind(kinds[t]) ? 1 : 0;
    }
    // samples per lane: 2, or 1 from six spline terms on (the register budget of the config-5 chain) and for small catalogs
    // of spline models (the rule below: one round of small workgroups)
    int U = n_spline >= 6 ? 1 : 2;
    const long long total = n_ev * n_pe + n_inj;
    const bool geometry_knobs = std::getenv("GWI_SAMPLES_PER_BLOCK") || std::getenv("GWI_PE_CHUNK") || std::getenv("GWI_INJ_CHUNK") ||
                                (std::getenv("GWI_SMALL_GEOMETRY") && std::atoi(std::getenv("GWI_SMALL_GEOMETRY")) == 0);
    bool small = false;
    if (U == 2 && n_spline > 0 && !geometry_knobs && total < 2816LL * prop.multiProcessorCount && total >= 64LL * prop.multiProcessorCount) U = 1, small = true;
    if (const char* env = std::getenv("GWI_SAMPLES_PER_LANE"))
      if (std::atoi(env) == 1 || std::atoi(env) == 2) U = std::atoi(env), small = false;
    std::string why;
    jit::Chain* jc = jit::get_chain(kinds, spec->n_terms, U, gwi_embedded_device_h, gwi_embedded_engine_h, why);
    hipModule_t mod = jc ? jit::module_on(jc, h->device, why) : nullptr;
    if (jc && !mod && jc->from_cache) {  // a cache file the runtime does not accept: drop it and compile once more
      jit::discard_chain(jc);
      jc = jit::get_chain(kinds, spec->n_terms, U, gwi_embedded_device_h, gwi_embedded_engine_h, why);
      mod = jc ? jit::module_on(jc, h->device, why) : nullptr;
    }
    bool ok = mod != nullptr;
    for (int role = 0; role < jit::kRoles && ok; ++role) {
      if (jc->lowered[role].empty()) continue;
      const hipError_t e = hipModuleGetFunction(&h->jit_fn[role], mod, jc->lowered[role].c_str());
      if (e != hipSuccess) {
        why = "jit: hipModuleGetFunction(" + jc->lowered[role] + "): " + hipGetErrorString(e);
        ok = false;
      }
    }
    if (ok) {
      Variant* v = new Variant();
      std::memset(v, 0, sizeof(*v));
      v->name = jc->name.c_str();
      v->n = jc->n;
      for (int t = 0; t < jc->n; ++t) v->kinds[t] = jc->kinds[t];
      v->samples_per_lane = jc->samples_per_lane;
      v->jit = jc;
      h->jit_variant = v;
      h->variant = v;
      h->small_geometry = small;
    } else {
      h->jit_note = why;
    }
  } else if (!h->variant) {
    h->jit_note = force_generic ? "GWI_FORCE_GENERIC" : "switched off (GWI_JIT=0)";
  }
  if (!h->variant) {
    // no compiled chain for this product of terms: the generic scan kernel evaluates it (term kinds read at run time)
    h->variant = &kGenericVariant;
    h->generic = true;
    static std::atomic<bool> warned{false};
    if (!force_generic && !warned.exchange(true) && !std::getenv("GWI_QUIET")) {
      std::string cmd;
      for (int t = 0; t < spec->n_terms; ++t) cmd += (t ? " " : "") + std::to_string(spec->terms[t].kind);
      std::fprintf(stderr,
                   "gwi: term-kind sequence [%s] has no ahead-of-time scan kernel and none could be compiled now (%s): using the generic one (run-time term loop, "
                   "2-2.7 x the scan time).  `python -m gwinferno_amd.precompile %s` on a machine with libhiprtc fills a cache directory this one can be pointed at (GWI_JIT_CACHE).\n",
                   cmd.c_str(), h->jit_note.c_str(), cmd.c_str());
    }
  }
  // Small catalogs of spline models (fewer than ~11 trips of 256 samples per CU: BASELINE config 3) are a chain of latencies, not
  // a throughput problem: more and smaller workgroups of the one-sample-per-lane sibling -- four per CU, equal tiles inside an
  // event -- measured 12.5-12.9 us for the config-3 scan against 13.4-14.3 for 443 workgroups of two samples per lane and two trips
  // (tools/geometry_sweep.py; profiles/round3/EXPERIMENTS.md).  Explicit geometry knobs switch the rule off.
  if (!h->variant->jit && h->variant->samples_per_lane == 2 && h->variant->has(jit::kSafe) && !h->generic && !std::getenv("GWI_SAMPLES_PER_LANE") && !std::getenv("GWI_SAMPLES_PER_BLOCK") &&
      !std::getenv("GWI_PE_CHUNK") && !std::getenv("GWI_INJ_CHUNK") && !(std::getenv("GWI_SMALL_GEOMETRY") && std::atoi(std::getenv("GWI_SMALL_GEOMETRY")) == 0)) {
    const Variant* sib = nullptr;
    for (int v = 0; v < kNumVariants; ++v) {
      const Variant& c = kVariants[v];
      if (c.samples_per_lane != 1 || c.n != h->variant->n) continue;
      bool same = true;
      for (int t = 0; t < c.n; ++t) same = same && c.kinds[t] == h->variant->kinds[t];
      if (same) sib = &c;
    }
    const long long total = n_ev * n_pe + n_inj;
    if (sib && total < 2816LL * prop.multiProcessorCount && total >= 64LL * prop.multiProcessorCount) {
      h->variant = sib;
      h->small_geometry = true;
    }
  }
  GWI_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  if (const char* env = std::getenv("GWI_SPIN_WAIT")) h->spin_wait = std::atoi(env) != 0;
  for (auto& e : h->ev) GWI_HIP(hipEventCreate(&e));
  if (const char* env = std::getenv("GWI_MAX_BATCH")) h->max_batch = std::atoi(env);
  if (h->max_batch < 1) h->max_batch = 1;
  if (h->max_batch > 64) h->max_batch = 64;
  h->n_ev = n_ev;
  h->n_pe = n_pe;
  h->n_inj = n_inj;
  h->inj_off = inj_offset(n_ev, n_pe);

  // ---- columns -> HBM (struct-of-arrays: one contiguous fp64 array per column and sample set)
  std::vector<const double*> tab_pe(spec->n_cols), tab_inj(spec->n_cols);
  if (ingest) {
    // setup on the device: raw catalog columns up, one kernel per sample set writes the engine's columns (gwi_ingest.h)
    for (int c = 0; c < spec->n_cols; ++c) {
      double *dpe = nullptr, *dinj = nullptr;
      if ((st = alloc_pair(h, &dpe, &dinj)) != GWI_OK) return st;
      tab_pe[c] = dpe;
      tab_inj[c] = dinj;
    }
    if ((st = ingest_run(h->err, ing_pe, n_ev * n_pe, spec->n_cols, h->d_cols_pe.data(), h->stream)) != GWI_OK) return st;
    if ((st = ingest_run(h->err, ing_inj, n_inj, spec->n_cols, h->d_cols_inj.data(), h->stream)) != GWI_OK) return st;
  } else {
    for (int c = 0; c < spec->n_cols; ++c) {
      double *dpe = nullptr, *dinj = nullptr;
      if ((st = alloc_pair(h, &dpe, &dinj)) != GWI_OK) return st;
      if (n_ev * n_pe) GWI_HIP(hipMemcpy(dpe, pe_cols[c], sizeof(double) * (size_t)(n_ev * n_pe), hipMemcpyHostToDevice));
      if (n_inj) GWI_HIP(hipMemcpy(dinj, inj_cols[c], sizeof(double) * (size_t)n_inj, hipMemcpyHostToDevice));
      tab_pe[c] = dpe;
      tab_inj[c] = dinj;
    }
  }

  // ---- spline terms read the KNOT coordinate of their column (gwi_device.h: spline_locate_knot): convert x -> u once, here.
  // A column that several spline terms read with different knots, or that another kind of term (or kappa) reads too, is
  // copied for each distinct use; otherwise it is converted in place.
  std::vector<const double*> over_pe(spec->n_terms, nullptr), over_inj(spec->n_terms, nullptr), over_pe1, over_inj1;
  {
    struct Use {
      double lo, inv_dx, top;
      const double *pe, *inj;
    };
    std::vector<std::vector<Use>> uses(spec->n_cols);
    std::vector<char> plain(spec->n_cols, 0);  // read as it is by some term or as kappa
    auto is_knot_term = [](const gwi_term& tm) { return tm.kind == GWI_TERM_EXP_SPLINE || tm.kind == GWI_TERM_LINEAR_SPLINE; };
    plain[spec->kappa_col] = 1;
    // a mass-ratio power law that takes log m1 from the m1 spline's column (GWI_RATIO_LOGM_FROM_SPLINE) reads the knot
    // coordinate too: it follows whatever conversion the spline term of the same knots asked for
    auto follows_knots = [](const gwi_term& tm, int j) { return tm.kind == GWI_TERM_POWERLAW_RATIO && (tm.flags & GWI_RATIO_LOGM_FROM_SPLINE) && j == 1; };
    for (int t = 0; t < spec->n_terms; ++t) {
      const gwi_term& tm = spec->terms[t];
      for (int j = 0; j < 2; ++j) {
        const int c = tm.cols[j];
        if (c < 0 || c >= spec->n_cols) continue;
        if (!(is_knot_term(tm) && j == 0) && !follows_knots(tm, j)) plain[c] = 1;
      }
    }
    for (int t = 0; t < spec->n_terms; ++t) {
      const gwi_term& tm = spec->terms[t];
      if (!is_knot_term(tm)) continue;
      const int c = tm.cols[0];
      const int n_int = tm.n_basis - 3;
      const double inv_dx = (double)n_int / (tm.p[1] - tm.p[0]);  // 1/dx of the uniform knots (interpolation.py:100-101)
      const bool clamp = tm.kind == GWI_TERM_EXP_SPLINE && !(tm.flags & GWI_SPLINE_OUTSIDE_ZERO_EXPONENT);
      const double top = clamp ? std::nextafter((double)n_int, 0.0) : -1.0;
      const Use* hit = nullptr;
      for (const Use& u : uses[c])
        if (u.lo == tm.p[0] && u.inv_dx == inv_dx && u.top == top) hit = &u;
      if (!hit) {
        Use u{tm.p[0], inv_dx, top, nullptr, nullptr};
        // in place when every knot term of this column wants the same conversion and nobody reads the column as it is;
        // else every distinct use gets a copy and the column itself stays what the caller handed over
        bool all_agree = !plain[c];
        for (int t2 = 0; t2 < spec->n_terms && all_agree; ++t2) {
          const gwi_term& o = spec->terms[t2];
          if (!is_knot_term(o) || o.cols[0] != c) continue;
          const bool oclamp = o.kind == GWI_TERM_EXP_SPLINE && !(o.flags & GWI_SPLINE_OUTSIDE_ZERO_EXPONENT);
          const double otop = oclamp ? std::nextafter((double)(o.n_basis - 3), 0.0) : -1.0;
          all_agree = o.p[0] == u.lo && (double)(o.n_basis - 3) / (o.p[1] - o.p[0]) == u.inv_dx && otop == u.top;
        }
        double *dpe = h->d_cols_pe[c], *dinj = h->d_cols_inj[c];
        if (!all_agree) {  // a private copy for this use
          if ((st = alloc_pair(h, &dpe, &dinj)) != GWI_OK) return st;
        }
        GWI_HIP(spline_knot_run(tab_pe[c], dpe, n_ev * n_pe, u.lo, u.inv_dx, u.top, h->stream));
        GWI_HIP(spline_knot_run(tab_inj[c], dinj, n_inj, u.lo, u.inv_dx, u.top, h->stream));
        u.pe = dpe;
        u.inj = dinj;
        uses[c].push_back(u);
        hit = &uses[c].back();
      }
      over_pe[t] = hit->pe;
      over_inj[t] = hit->inj;
    }
    over_pe1.assign(spec->n_terms, nullptr);
    over_inj1.assign(spec->n_terms, nullptr);
    for (int t = 0; t < spec->n_terms; ++t) {
      const gwi_term& tm = spec->terms[t];
      if (!follows_knots(tm, 1)) continue;
      const int c = tm.cols[1];
      const Use* hit = nullptr;
      if (c >= 0 && c < spec->n_cols)
        for (const Use& u : uses[c])
          if (u.lo == tm.p[1] && u.inv_dx == tm.p[2]) hit = &u;
      if (!hit) return fail(h, GWI_ERR_INVALID, "GWI_RATIO_LOGM_FROM_SPLINE: cols[1] is not the coordinate column of a spline term with the knots given in p[1], p[2]");
      over_pe1[t] = hit->pe;
      over_inj1[t] = hit->inj;
    }
    GWI_HIP(hipStreamSynchronize(h->stream));
  }

  // ---- normaliser grids
  std::vector<NormD> nd(spec->n_norms ? spec->n_norms : 1);
  for (int j = 0; j < spec->n_norms; ++j) {
    const gwi_norm& nm = spec->norms[j];
    NormD& d = nd[j];
    std::memset(&d, 0, sizeof(d));
    d.n_pts = nm.n_pts;
    d.expo_theta = nm.expo_theta;
    d.n_basis = nm.n_basis;
    d.coef_off = nm.coef_off;
    d.flags = nm.spline_flags;
    d.expo_add = nm.expo_add;
    d.lo = nm.lo;
    d.hi = nm.hi;
    double* p;
    if ((st = upload(h, nm.tw, nm.n_pts, &p, &h->d_norm_arrays)) != GWI_OK) return st;
    d.tw = p;
    if (nm.lb) {
      if ((st = upload(h, nm.lb, nm.n_pts, &p, &h->d_norm_arrays)) != GWI_OK) return st;
      d.lb = p;
    }
    if (nm.expo_theta >= 0) {
      if ((st = upload(h, nm.l1, nm.n_pts, &p, &h->d_norm_arrays)) != GWI_OK) return st;
      d.l1 = p;
    }
    if (nm.n_basis > 0) {
      if ((st = upload(h, nm.us, nm.n_pts, &p, &h->d_norm_arrays)) != GWI_OK) return st;
      d.us = p;
    }
    // the spec's host pointers are not retained
    h->spec.norms[j].tw = h->spec.norms[j].lb = h->spec.norms[j].l1 = h->spec.norms[j].us = nullptr;
  }
  GWI_HIP(hipMalloc(&h->d_norms, sizeof(NormD) * nd.size()));
  GWI_HIP(hipMemcpy(h->d_norms, nd.data(), sizeof(NormD) * nd.size(), hipMemcpyHostToDevice));

  // dynamic LDS of the scan kernel: the workgroup's spline-gradient rows, [n_theta][rep] doubles (spline_scatter in
  // gwi_device.h).  rep = 64 would give every lane its own replica; 16 (four lanes per replica, bank = replica) measured
  // the same or better on the BASELINE catalogs (config 5 scan: rep 8 / 16 / 32 / 64 = 61.5 / 51.2 / 51.7 / 70.0 us, config 3:
  // 15.5 / 14.6 / 15.3 / 16.1) because the rows must also fit next to the kernel's static LDS as many times as the
  // register budget allows workgroups on a CU, and are zeroed and summed once per workgroup.
  bool has_spline = false;
  for (int t = 0; t < spec->n_terms; ++t)
    has_spline = has_spline || spec->terms[t].kind == GWI_TERM_EXP_SPLINE || spec->terms[t].kind == GWI_TERM_LINEAR_SPLINE || spec->terms[t].kind == GWI_TERM_EXP_SPLINE_LERP;
  has_spline = has_spline || h->generic;  // the generic chain keeps every gradient sum in the LDS rows
  if (const char* env = std::getenv("GWI_DETERMINISTIC")) h->deterministic = std::atoi(env) != 0;
  size_t scan_lds = 0;
  int rep = 1;
  if (has_spline) {
    size_t lds_per_cu = 160 * 1024, static_lds = 14 * 1024;  // gfx950: 160 KiB per CU; static: s_theta + s_out + s_part + s_wrec (what the kernel reports replaces this guess)
    if (h->variant->jit) {
      int v = 0;
      if (hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_SHARED_SIZE_BYTES, h->jit_fn[jit::kScan]) == hipSuccess && v > 0) static_lds = (size_t)v;
    } else {
      hipFuncAttributes fa;
      if (hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(h->variant->fn[jit::kScan])) == hipSuccess && fa.sharedSizeBytes > 0) static_lds = fa.sharedSizeBytes;
    }
    // 16 replicas: what the regular scan kernels are compiled for (immediate row offsets).  Where rows that wide cost a
    // resident workgroup (n_theta beyond ~100), that is the cheaper loss: 8 replicas measured 20 % slower at config 5.
    // Any other count (GWI_GACC_REP, the replay mode's 64) runs the SAFE instantiation, which takes it at run time.
    rep = 1 << kRegularRepShift;
    if (const char* env = std::getenv("GWI_GACC_REP")) rep = std::atoi(env);
    if (h->deterministic) rep = 64;  // one replica per lane: a wave instruction never meets itself on an address
    if (rep < 1) rep = 1;
    if (rep > 64) rep = 64;
    while (rep & (rep - 1)) rep &= rep - 1;  // power of two
    while (rep > 1 && sizeof(double) * (size_t)spec->n_theta * rep + static_lds > lds_per_cu) rep >>= 1;
    const size_t poly_lds = 4 * sizeof(double) * (size_t)kPolyStride;  // the power-basis table of the spline values (gwi_device.h: spline_poly), behind the rows
    while (rep > 1 && sizeof(double) * (size_t)spec->n_theta * rep + poly_lds + static_lds > lds_per_cu) rep >>= 1;
    scan_lds = sizeof(double) * (size_t)spec->n_theta * rep + poly_lds;
    if (scan_lds > 48 * 1024 && !h->variant->jit) {  // beyond the default dynamic-LDS limit of a HIP launch (the AQL packets carry any size; so do module launches)
      for (int role = 0; role < jit::kRoles; ++role)
        if (h->variant->fn[role]) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(h->variant->fn[role]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)scan_lds);
    }
  }
  h->gacc_rep = rep;
  if (has_spline) {
    // Batched launches of >= 9 points take the 16-points-per-wavefront kernel (gwi_mfma.h) where the model has an
    // instantiation: at K = 16 it measures 15 % ahead of the 4-tap kernel on the BASELINE catalogs (config 5: 34.9 vs 41.0 us
    // per evaluation, config 3: 8.7 vs 10.5) and its gradient is bit-reproducible.  GWI_BATCH_MFMA=0 keeps the 4-tap kernel,
    // =2 uses the matrix-core kernel for every batch size; GWI_BATCH_ROWS=1 selects the LDS-row variant of the same kernel.
    h->mfma = find_mfma_variant(*spec);
    if (h->mfma && !std::getenv("GWI_BATCH_MFMA") && !std::getenv("GWI_BATCH_ROWS")) {
      // more than 8 gradient tiles: the 4-tap kernel is faster (the reference's default spline counts, 11 tiles / 165
      // hyper-parameters, on the config-3 catalog: 11.7 us per evaluation against 14.4 on the matrix cores and 14.5 with LDS rows)
      int tiles = 0;
      for (int t = 0; t < h->mfma->n; ++t) tiles += h->mfma->tiles[t];
      if (tiles > 8) h->mfma = nullptr;
    }
    if (const char* env = std::getenv("GWI_BATCH_MFMA")) {
      if (std::atoi(env) == 0) h->mfma = nullptr;
      if (std::atoi(env) >= 2) h->mfma_min_batch = 1;
    }
    if (const char* env = std::getenv("GWI_BATCH_ROWS")) {
      if (std::atoi(env) >= 1) {
        h->mfma = find_mfma_variant(*spec);
        h->batch_rows = h->mfma != nullptr;
      }
      if (std::atoi(env) >= 2) h->mfma_min_batch = 1;
    }
    // no path named by the environment: the static rule above stands (config 3 is a tie between the kernels that flipped from box
    // to box when it was measured by default, config 5 prefers the matrix cores by 15 %); GWI_BATCH_AUTOTUNE=1 measures instead
    if (const char* env = std::getenv("GWI_BATCH_AUTOTUNE")) h->autotune_wanted = std::atoi(env) != 0;
    h->batch_autotune = h->autotune_wanted && h->mfma && !std::getenv("GWI_BATCH_MFMA") && !std::getenv("GWI_BATCH_ROWS") && !h->deterministic;
    // a spline model without an ahead-of-time matrix-core instantiation: compiled on its first batched launch of >= 9 points and used
    // from then on (GWI_BATCH_MFMA=1: compiled now; GWI_JIT=0 or GWI_BATCH_MFMA=0: not at all; GWI_BATCH_AUTOTUNE=1: measured against
    // the 4-tap kernel there)
    {
      bool jit_ok = !h->generic && !h->deterministic && !find_mfma_variant(*spec) && !std::getenv("GWI_BATCH_ROWS");
      if (const char* env = std::getenv("GWI_JIT")) jit_ok = jit_ok && std::atoi(env) != 0;
      const char* want = std::getenv("GWI_BATCH_MFMA");
      if (jit_ok && want && std::atoi(want) >= 1) {
        if (try_jit_mfma(h) && std::atoi(want) >= 2) h->mfma_min_batch = 1;
      } else if (jit_ok && !want) {
        h->mfma_jit_pending = true;
      }
    }
    if (h->mfma && h->batch_rows) {
      // sample-slot replicas of the gradient rows: as many (4, 2, 1) as leave two workgroups per CU their LDS
      h->rows_rep = 4;
      while (h->rows_rep > 1 && sizeof(double) * mfma_lds_doubles(spec->n_theta, spec->n_terms, h->mfma->row_doubles, h->rows_rep) + 4608 > 80 * 1024) h->rows_rep >>= 1;
      if (const char* env = std::getenv("GWI_ROWS_REP")) h->rows_rep = std::max(1, std::min(4, std::atoi(env)));
      h->mfma_lds_bytes = sizeof(double) * mfma_lds_doubles(spec->n_theta, spec->n_terms, h->mfma->row_doubles, h->rows_rep);
      if (h->mfma_lds_bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(h->mfma->rows_fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->mfma_lds_bytes);
    } else if (h->mfma) {
      h->mfma_lds_bytes = sizeof(double) * mfma_lds_doubles(spec->n_theta, spec->n_terms, h->mfma->row_doubles, 0);
      if (h->mfma_lds_bytes > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(h->mfma->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->mfma_lds_bytes);
    }
  }
  // ---- launch geometry.  Default: ~2048 scan workgroups (8 per CU).  A step lasts only ~10 us, so a
  // partial second dispatch round (a few workgroups that can only start when the first finishers
  // retire) costs a large fraction of it: when one round of resident workgroups can hold the whole
  // catalog with <= 4 trips each, size the workgroups for exactly one round instead.
  const long long gran = (long long)h->variant->samples_per_lane * kBlock;  // every lane carries U samples per trip
  long long spb = 0, spb_batch = 0;  // spb_batch != 0: batched launches use another tile size
  if (const char* env = std::getenv("GWI_SAMPLES_PER_BLOCK")) spb = std::atoll(env);
  if (spb <= 0) {
    const long long total = n_ev * n_pe + n_inj;
    spb = (total + 2047) / 2048;
    int occ = 0;
    bool single_round = true;
    if (const char* env = std::getenv("GWI_SINGLE_ROUND")) single_round = std::atoi(env) != 0;
    const hipError_t occ_rc = h->variant->jit ? hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, h->jit_fn[jit::kScan], kBlock, scan_lds)
                                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, h->variant->fn[jit::kScan], kBlock, scan_lds);
    if (single_round && occ_rc == hipSuccess && occ > 0) {
      const long long capacity = (long long)prop.multiProcessorCount * occ;
      for (long long cand = gran; cand <= 4 * gran; cand += gran) {
        const long long pad = ((n_pe + gran - 1) / gran) * gran;
        const long long cpe = cand < pad ? cand : pad;
        const long long blocks = n_ev * ((n_pe + cpe - 1) / cpe) + (n_inj + cand - 1) / cand;
        if (blocks <= capacity) {
          if (cand > spb) spb = cand;
          break;
        }
      }
      // One trip per workgroup where two would still give every CU a workgroup: BATCHED launches take two (bgeo below).
      // Prologue and record reduction are a quarter of a one-trip workgroup's instructions (config 2, K = 16: scan 52.1 ->
      // 45.3 us, 205 k -> 236 k evals/s); a single evaluation gains nothing from it (16.9 vs 16.8 us) and four concurrent
      // chains lose ~10 %, so the single-evaluation geometry stays at one trip.
      if (spb == gran) {
        const long long c2 = 2 * gran, pad = ((n_pe + gran - 1) / gran) * gran;
        const long long cpe = c2 < pad ? c2 : pad;
        const long long blocks2 = n_ev * ((n_pe + cpe - 1) / cpe) + (n_inj + c2 - 1) / c2;
        if (blocks2 >= (long long)prop.multiProcessorCount) spb_batch = c2;
      }
    }
  }
  spb = ((spb + gran - 1) / gran) * gran;
  if (spb < gran) spb = gran;
  const long long n_pe_pad = ((n_pe + gran - 1) / gran) * gran;
  h->chunk_pe = (int)(spb < n_pe_pad ? spb : n_pe_pad);
  h->chunk_inj = (int)spb;
  if (h->small_geometry) {
    const long long total = n_ev * n_pe + n_inj;
    double per_cu = 4.0;  // one round of resident workgroups at four waves per SIMD; 2.0 / 2.75 / 3.4 / 4.0 / 5.5 / 7.0 measured 15.4 / 13.8 / 13.7 / 13.1 / 16.2 / 14.7 us on one box
    if (const char* env = std::getenv("GWI_SMALL_WGS_PER_CU")) per_cu = std::max(0.5, std::atof(env));
    const long long target = std::max<long long>(64, (long long)((double)total / (per_cu * prop.multiProcessorCount) + 0.5));
    const long long tiles_pe = std::max<long long>(1, (n_pe + target / 2) / target);
    h->chunk_pe = (int)((n_pe + tiles_pe - 1) / tiles_pe);  // equal tiles inside an event
    h->chunk_inj = (int)target;
    spb_batch = 0;
  }
  // The combine launch requests the records of up to 16 tiles of an event in its first memory round trip (kEarly in
  // combine_group) and needs another dependent round per 16 more: a catalog of few events with many posterior samples each --
  // one rank's share of config 5 on 8 GPUs: 25 events x 10 000 -- got 40 tiles of 256 per event and a 9.5 us combine behind a
  // 10.7 us scan.  At most 16 tiles per event where that still leaves every CU a workgroup: 768-sample tiles there, scan
  // 11.7 us, combine 4.1 us, 27.4 -> 21.3 us per local evaluation (tools/shard_time.py).
  if (!std::getenv("GWI_SAMPLES_PER_BLOCK") && !(std::getenv("GWI_TILE_CAP") && std::atoi(std::getenv("GWI_TILE_CAP")) == 0)) {
    const long long cap_chunk = (((n_pe + 15) / 16 + gran - 1) / gran) * gran;
    if (h->chunk_pe < cap_chunk) {
      const long long inj_chunk = h->chunk_inj < cap_chunk ? cap_chunk : h->chunk_inj;
      const long long blocks = n_ev * ((n_pe + cap_chunk - 1) / cap_chunk) + (n_inj + inj_chunk - 1) / inj_chunk;
      if (blocks >= (long long)prop.multiProcessorCount) {
        h->chunk_pe = (int)cap_chunk;
        h->chunk_inj = (int)inj_chunk;
      }
    }
  }
  // experiment knobs: exact tile sizes (the kernel takes any size; a trip covers samples_per_lane * 256 samples)
  if (const char* env = std::getenv("GWI_PE_CHUNK")) h->chunk_pe = std::max(1, std::atoi(env));
  if (const char* env = std::getenv("GWI_INJ_CHUNK")) h->chunk_inj = std::max(1, std::atoi(env));
  // the tail kernels map the tile records of one group to the lanes of ONE wave: an event may have at most 64 tiles, the
  // injections at most 64 groups x 64 tiles.  Few events with very many posterior samples (3 events x 1 M) or a very long
  // injection set exceed that with the default tile size: grow the tiles (whole trips) until they fit.
  {
    auto round_up = [&](long long v) { return ((v + gran - 1) / gran) * gran; };
    const long long min_pe = round_up((n_pe + 63) / 64), min_inj = round_up((n_inj + 64 * 64 - 1) / (64 * 64));
    if (h->chunk_pe < min_pe) h->chunk_pe = (int)min_pe;
    if (h->chunk_inj < min_inj) h->chunk_inj = (int)min_inj;
  }
  // the scan receives its tile sizes in 16 bits each (ScanHead::chunks): exact below 32 768 samples, multiples of 256 above
  auto packable = [](int c) { return c >= 32768 && c % 256 ? (c / 256 + 1) * 256 : c; };
  h->chunk_pe = packable(h->chunk_pe);
  h->chunk_inj = packable(h->chunk_inj);
  h->tiles_per_event = (int)((n_pe + h->chunk_pe - 1) / h->chunk_pe);
  h->n_inj_tiles = (int)((n_inj + h->chunk_inj - 1) / h->chunk_inj);
  h->n_scan_blocks = (int)(n_ev * h->tiles_per_event + h->n_inj_tiles);
  h->rec_stride = kRecHeader + spec->n_theta;
  // injection tiles are combined in groups of <= 16 records (one workgroup each): a group's tile values are then
  // all requested in the combine kernel's first memory round trip (kEarly there)
  h->tiles_per_inj_group = 16;  // <= 64: one tile per lane in combine_kernel
  if (const char* env = std::getenv("GWI_TILES_PER_INJ_GROUP")) h->tiles_per_inj_group = std::max(1, std::min(64, std::atoi(env)));
  h->n_inj_groups = (h->n_inj_tiles + h->tiles_per_inj_group - 1) / h->tiles_per_inj_group;
  if (h->n_inj_groups < 1) h->n_inj_groups = 1;
  if (h->n_inj_groups > 64) {  // final_kernel maps groups to the lanes of one wave
    h->tiles_per_inj_group = (h->n_inj_tiles + 63) / 64;
    h->n_inj_groups = (h->n_inj_tiles + h->tiles_per_inj_group - 1) / h->tiles_per_inj_group;
  }
  if (h->tiles_per_event > 64 || h->tiles_per_inj_group > 64 || h->n_inj_groups > 64)
    return fail(h, GWI_ERR_INVALID, "launch geometry: more than 64 tile records per group (internal error: the tile sizes above should have prevented this)");
  h->scan_lds_bytes = scan_lds;
  if (const char* env = std::getenv("GWI_BATCH_GEOMETRY")) {  // 0: batched launches on the single evaluation's geometry
    if (std::atoi(env) == 0) spb_batch = 0;
  }
  // Parametric models have two batched kernels: one grid row per point (scan_kernel BATCH: the catalog streams K times through
  // L2 / the Infinity Cache) and scan_pbatch_kernel (every sample loaded once for the points a workgroup draws; single-trip
  // tiles: the single evaluation's where those are single trips already, else a batch geometry of one trip per workgroup).
  // Which one is faster turned out to depend on the BOX: round 5's boxes ran pbatch 3-8 % ahead (42.3 against 43.7 us at config 2,
  // K = 16), every box of round 6 ran it 5-14 % behind at every catalog size from 1 to 8 x config 2 (49.4 against 54.8 us; 287
  // against 327 us at 8 x; blocking 242-251 k against 215-227 k evals/s: profiles/round6/EXPERIMENTS.md section 5) although it issues
  // 30 % fewer instructions and moves a seventh of the bytes -- the denser fp64 kernel is the one whose time varies from box to
  // box.  The choice must be static (the two sum in different orders): the row-per-point kernel is the default since round 6,
  // GWI_PBATCH=1 (or a row size, GWI_PBATCH_PTS) selects the one-load-per-sample kernel.
  h->pbatch = false;
  if (const char* env = std::getenv("GWI_PBATCH_PTS")) h->pbatch_pts = std::max(0, std::atoi(env));
  if (const char* env = std::getenv("GWI_PBATCH")) h->pbatch = std::atoi(env) != 0;
  else h->pbatch = h->pbatch_pts > 0;
  h->pbatch = h->pbatch && h->variant->has(jit::kPbatch) && !h->generic;
  h->pbatch_balanced = h->pbatch_pts == 0;  // (naming a row size asks for the rows mode)
  if (const char* env = std::getenv("GWI_PBATCH_BALANCED")) h->pbatch_balanced = std::atoi(env) != 0;
  if (h->pbatch) {
    int occ = 0;
    const hipError_t oe = h->variant->jit ? hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, h->jit_fn[jit::kPbatch], kBlock, 0)
                                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, h->variant->fn[jit::kPbatch], kBlock, 0);
    if (oe != hipSuccess) (void)hipGetLastError();
    h->pbatch_wgs_per_cu = (oe == hipSuccess && occ > 0) ? std::min(occ, 8) : 4;
    if (const char* env = std::getenv("GWI_PBATCH_WGS_PER_CU")) h->pbatch_wgs_per_cu = std::max(1, std::min(16, std::atoi(env)));
  }
  if (h->pbatch && !std::getenv("GWI_BATCH_GEOMETRY")) spb_batch = (long long)pbatch_u(h->variant->samples_per_lane) * kBlock;  // (not distinct below where that is the single geometry)
  if (spb_batch > 0 && !std::getenv("GWI_PE_CHUNK") && !std::getenv("GWI_INJ_CHUNK")) {
    auto& b = h->bgeo;
    b.chunk_pe = packable((int)(spb_batch < n_pe_pad ? spb_batch : n_pe_pad));
    b.chunk_inj = packable((int)spb_batch);
    b.tiles_per_event = (int)((n_pe + b.chunk_pe - 1) / b.chunk_pe);
    b.n_inj_tiles = (int)((n_inj + b.chunk_inj - 1) / b.chunk_inj);
    b.n_scan_blocks = (int)(n_ev * b.tiles_per_event + b.n_inj_tiles);
    b.tiles_per_inj_group = 16;
    b.n_inj_groups = std::max(1, (b.n_inj_tiles + b.tiles_per_inj_group - 1) / b.tiles_per_inj_group);
    if (b.n_inj_groups > 64) {  // final_kernel maps groups to the lanes of one wave
      b.tiles_per_inj_group = (b.n_inj_tiles + 63) / 64;
      b.n_inj_groups = (b.n_inj_tiles + b.tiles_per_inj_group - 1) / b.tiles_per_inj_group;
    }
    // within what the tail kernels take (64 tile records per group, 64 groups); the buffers below hold either geometry
    b.distinct = b.tiles_per_event <= 64 && b.tiles_per_event < (1 << kGeomTilesBits) && b.tiles_per_inj_group <= 64 && b.n_inj_groups <= 64 &&
                 (b.chunk_pe != h->chunk_pe || b.chunk_inj != h->chunk_inj);
  }
  const int max_scan_blocks = std::max(h->n_scan_blocks, h->bgeo.distinct ? h->bgeo.n_scan_blocks : 0);
  const int max_inj_groups = std::max(h->n_inj_groups, h->bgeo.distinct ? h->bgeo.n_inj_groups : 0);

  const size_t KB = (size_t)h->max_batch;  // every per-evaluation buffer holds max_batch hyper-parameter points
  GWI_HIP(hipMalloc(&h->d_partials, sizeof(double) * KB * (size_t)(max_scan_blocks ? max_scan_blocks : 1) * h->rec_stride));
  GWI_HIP(hipMalloc(&h->d_ev_out, sizeof(double) * KB * 4 * (size_t)(n_ev ? n_ev : 1)));
  GWI_HIP(hipMalloc(&h->d_ev_grad, sizeof(double) * KB * (size_t)(n_ev ? n_ev : 1) * spec->n_theta));
  GWI_HIP(hipMalloc(&h->d_inj_out, sizeof(double) * KB * 4 * (size_t)max_inj_groups));
  GWI_HIP(hipMalloc(&h->d_inj_grad, sizeof(double) * KB * (size_t)max_inj_groups * spec->n_theta));
  GWI_HIP(hipMalloc(&h->d_tblocks, sizeof(ThetaBlock) * KB));
  GWI_HIP(hipHostMalloc((void**)&h->h_tblocks, sizeof(ThetaBlock) * KB, hipHostMallocMapped));
  GWI_HIP(hipHostGetDevicePointer((void**)&h->h_tblocks_dev, h->h_tblocks, 0));
  if (const char* env = std::getenv("GWI_STAGE_KERNEL")) h->stage_kernel = std::atoi(env) != 0;
  GWI_HIP(hipHostMalloc((void**)&h->h_record, sizeof(double) * KB * record_len(h), hipHostMallocMapped));
  GWI_HIP(hipHostGetDevicePointer((void**)&h->h_record_dev, h->h_record, 0));
  GWI_HIP(hipHostMalloc((void**)&h->h_ev, sizeof(double) * KB * 3 * (size_t)(n_ev ? n_ev : 1), hipHostMallocMapped));
  GWI_HIP(hipHostGetDevicePointer((void**)&h->h_ev_dev, h->h_ev, 0));
  std::memset(h->h_record, 0, sizeof(double) * KB * record_len(h));
  // the final launch: one workgroup per ~48 events (at most 8), each publishing a partial record (config 5, 200 events:
  // 1 / 4 / 8 / 16 workgroups = 7.6 / 5.8 / 5.9 / 6.7 us -- more records are more small PCIe writes)
  h->final_groups = (int)std::max<long long>(1, std::min<long long>(8, n_ev / 48));
  if (const char* env = std::getenv("GWI_FINAL_GROUPS")) h->final_groups = std::max(1, std::min(64, std::atoi(env)));
  GWI_HIP(hipHostMalloc((void**)&h->h_fin, sizeof(double) * KB * h->final_groups * record_len(h), hipHostMallocMapped));
  GWI_HIP(hipHostGetDevicePointer((void**)&h->h_fin_dev, h->h_fin, 0));
  std::memset(h->h_fin, 0, sizeof(double) * KB * h->final_groups * record_len(h));
  {  // tile references of spline models (scan_kernel, shared mode): none yet
    const size_t n = (size_t)(1 + 2 * h->max_batch) * (size_t)(max_scan_blocks ? max_scan_blocks : 1);  // rows: see nref_row0
    std::vector<int> none(n, kNoRef);
    GWI_HIP(hipMalloc(&h->d_tile_nref, sizeof(int) * n));
    GWI_HIP(hipMemcpy(h->d_tile_nref, none.data(), sizeof(int) * n, hipMemcpyHostToDevice));
  }
  GWI_HIP(hipMalloc(&h->d_seq, 2 * sizeof(unsigned long long)));
  GWI_HIP(hipMemset(h->d_seq, 0, 2 * sizeof(unsigned long long)));
  GWI_HIP(hipHostMalloc((void**)&h->h_redo, sizeof(unsigned long long), hipHostMallocMapped));
  GWI_HIP(hipHostGetDevicePointer((void**)&h->h_redo_dev, h->h_redo, 0));
  *h->h_redo = 0;
  // host-final mode for small problems: the per-group rows fit a few KiB, so the host sums them and
  // the third launch (final_kernel: ~1.5 us boundary + ~6-9 us of latency chain) disappears
  {
    const size_t n_groups = (size_t)n_ev + h->n_inj_groups;
    const size_t row_bytes = 64 * n_groups * (size_t)((3 + spec->n_theta + 6) / 7);  // self-validating 64-byte lines: 7 values + the sequence number
    // measured per evaluation: config 3 (48 KB of rows) gains 2 us from host-final, the reference's default spline counts on 69 events
    // (165 hyper-parameters: 114 KB) 0.85 us (27.35 against 28.21 us; profiles/round6/hostfinal_def50k.txt), config 5 (195 KB) loses 5-8
    size_t host_final_limit = 120 * 1024;
    if (const char* env = std::getenv("GWI_HOST_FINAL_BYTES")) host_final_limit = (size_t)std::atoll(env);
    h->host_final = row_bytes <= host_final_limit;
    if (const char* env = std::getenv("GWI_HOST_FINAL")) h->host_final = h->host_final && std::atoi(env) != 0;
    GWI_HIP(hipHostMalloc((void**)&h->h_rows, KB * row_bytes, hipHostMallocMapped));
    GWI_HIP(hipHostGetDevicePointer((void**)&h->h_rows_dev, h->h_rows, 0));
    std::memset(h->h_rows, 0, KB * row_bytes);
    const size_t nn = KB * (size_t)(spec->n_norms ? spec->n_norms : 1);
    GWI_HIP(hipHostMalloc((void**)&h->h_norm, sizeof(double) * nn, hipHostMallocMapped));
    GWI_HIP(hipHostGetDevicePointer((void**)&h->h_norm_dev, h->h_norm, 0));
    GWI_HIP(hipHostMalloc((void**)&h->h_norm_stamp, sizeof(unsigned long long) * nn, hipHostMallocMapped));
    GWI_HIP(hipHostGetDevicePointer((void**)&h->h_norm_stamp_dev, h->h_norm_stamp, 0));
    std::memset(h->h_norm_stamp, 0, sizeof(unsigned long long) * nn);
  }

  // ---- constant part of the kernel-argument block
  KArgs& k = h->kargs;
  std::memset(&k, 0, sizeof(k));
  for (int t = 0; t < spec->n_terms; ++t)
    for (int j = 0; j < 2; ++j) {
      const int c = spec->terms[t].cols[j] >= 0 && spec->terms[t].cols[j] < spec->n_cols ? spec->terms[t].cols[j] : spec->terms[t].cols[0];
      k.pe_tcols[t][j] = (j == 0 && over_pe[t]) ? over_pe[t] : (j == 1 && over_pe1[t]) ? over_pe1[t] : tab_pe[c];
      k.inj_tcols[t][j] = (j == 0 && over_inj[t]) ? over_inj[t] : (j == 1 && over_inj1[t]) ? over_inj1[t] : tab_inj[c];
    }
  k.kappa_pe = tab_pe[spec->kappa_col];
  k.kappa_inj = tab_inj[spec->kappa_col];
  k.norms = h->d_norms;
  k.norm_out_host = h->h_norm_dev;
  k.norm_stamps_host = h->h_norm_stamp_dev;
  k.partials = h->d_partials;
  k.n_pe = n_pe;
  k.n_inj = n_inj;
  k.n_ev = (int)n_ev;
  k.tiles_per_event = h->tiles_per_event;
  k.chunk_pe = h->chunk_pe;
  k.n_inj_tiles = h->n_inj_tiles;
  k.chunk_inj = h->chunk_inj;
  k.n_norms = spec->n_norms;
  k.n_terms = spec->n_terms;
  k.n_theta = spec->n_theta;
  k.kappa_col = spec->kappa_col;
  k.rec_stride = h->rec_stride;
#ifdef GWI_STAMPS
  GWI_HIP(hipMalloc(&k.stamps, sizeof(unsigned long long) * (size_t)(h->n_scan_blocks + spec->n_norms + 1) * kWaves * 8));
  GWI_HIP(hipMemset(k.stamps, 0, sizeof(unsigned long long) * (size_t)(h->n_scan_blocks + spec->n_norms + 1) * kWaves * 8));
#endif
  k.gacc_rep = rep;
  k.gacc_shift = __builtin_ctz((unsigned)rep);
  k.seq_dev = h->d_seq;
  k.tile_nref = h->d_tile_nref;
  k.nref_stride = max_scan_blocks ? max_scan_blocks : 1;
  k.rows_rep = h->rows_rep;
  k.redo_host = h->h_redo_dev;
  k.redo_dev = h->d_seq + 1;
  k.two_pass = 0;
  k.deterministic = h->deterministic ? 1 : 0;
  {  // the scan's preloaded arguments (ScanHead): kappa, then the terms' columns in term order; geometry by set_geometry
    ScanHead& hd = h->sblock.head;
    std::memset(&hd, 0, sizeof(hd));
    int slot = 0;
    bool joint = true;
    auto put = [&](const double* pe, const double* inj) {
      joint = joint && inj == pe + h->inj_off;
      if (slot < kHeadCols) hd.col[slot] = pe;
      ++slot;
    };
    put(k.kappa_pe, k.kappa_inj);
    for (int t = 0; t < spec->n_terms; ++t)
      for (int j = 0; j < term_cols(spec->terms[t].kind) && j < 2; ++j) put(k.pe_tcols[t][j], k.inj_tcols[t][j]);
    for (; slot < kHeadCols; ++slot) hd.col[slot] = k.kappa_pe;  // unused slots: any valid address
    if (!joint) return fail(h, GWI_ERR_INVALID, "internal error: a column's injection part does not sit inj_offset() behind its posterior-sample part");
    if (n_ev >= (1LL << kGeomEventBits) || n_pe >= (1LL << 32) || n_inj >= (1LL << 32) || spec->n_norms >= 16 || h->tiles_per_event >= (1 << kGeomTilesBits) ||
        !chunk_packs(h->chunk_pe) || !chunk_packs(h->chunk_inj) || (h->bgeo.distinct && (!chunk_packs(h->bgeo.chunk_pe) || !chunk_packs(h->bgeo.chunk_inj))))
      return fail(h, GWI_ERR_INVALID, "catalog shape outside the scan's packed geometry (events < 2^20, samples per event and injections < 2^32, tiles < 8.4 M samples)");
    hd.n_pe = (unsigned)n_pe;
    hd.n_inj = (unsigned)n_inj;
  }
  for (int t = 0; t < spec->n_terms; ++t) {
    const gwi_term& tm = spec->terms[t];
    TermD& d = k.terms[t];
    d.kind = tm.kind;
    d.n_basis = tm.n_basis;
    d.th0 = tm.theta[0];
    d.th1 = tm.theta[1];
    d.th2 = tm.theta[2];
    d.th3 = tm.theta[3];
    d.flags = tm.flags;
    d.p0 = tm.p[0];
    d.p1 = tm.p[1];
    d.p2 = tm.p[2];
    d.p3 = tm.p[3];
    d.th4 = tm.kind == GWI_TERM_PLPEAK_SMOOTH ? tm.coef_off : 0;
    if (tm.kind == GWI_TERM_EXP_SPLINE_LERP) d.th1 = tm.norm;  // the grid's spline coordinates live in that normaliser's `us`
    if (tm.kind == GWI_TERM_EXP_SPLINE || tm.kind == GWI_TERM_LINEAR_SPLINE || tm.kind == GWI_TERM_EXP_SPLINE_LERP) {
      d.th0 = tm.coef_off;
      d.p2 = (double)(tm.n_basis - 3) / (tm.p[1] - tm.p[0]);  // 1/dx of the uniform knots (interpolation.py:100-101)
      d.p3 = (double)(tm.n_basis - 3);                          // the closed domain in knot coordinates: [0, p3]
    }
  }
  h->combine_threads = spec->n_theta + 4 <= 64 ? 64 : kBlock;
  if (const char* env = std::getenv("GWI_COMBINE_THREADS")) h->combine_threads = std::atoi(env) == 64 ? 64 : kBlock;
  setup_aql(h, prop);
  return GWI_OK;
}

gwi_status gwi_create(const gwi_spec* spec, const double* const* pe_cols, int64_t n_ev, int64_t n_pe, const double* const* inj_cols, int64_t n_inj,
                      int32_t device, gwi_handle* out) {
  return create_impl(spec, pe_cols, n_ev, n_pe, inj_cols, n_inj, device, out, nullptr, nullptr);
}

gwi_status gwi_create_ingest(const gwi_spec* spec, const gwi_ingest_program* pe, int64_t n_ev, int64_t n_pe, const gwi_ingest_program* inj, int64_t n_inj,
                             int32_t device, gwi_handle* out) {
  if (!pe || !inj) return GWI_ERR_INVALID;
  if (device == GWI_DEVICE_HOST_ONLY) return GWI_ERR_INVALID;  // a host-only handle owns no columns: use gwi_create
  return create_impl(spec, nullptr, n_ev, n_pe, nullptr, n_inj, device, out, pe, inj);
}

gwi_status gwi_ingest_columns(const gwi_ingest_program* prog, int64_t n, int32_t n_cols, double* const* cols, int32_t device) {
  if (!prog || !cols || n < 0 || n_cols < 1 || n_cols > GWI_MAX_COLS) return GWI_ERR_INVALID;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return GWI_ERR_NO_DEVICE;
  if (device >= n_dev) return GWI_ERR_NO_DEVICE;
  if (device >= 0 && hipSetDevice(device) != hipSuccess) return GWI_ERR_HIP;
  std::string err;
  std::vector<double*> d(n_cols, nullptr);
  gwi_status st = GWI_OK;
  for (int c = 0; c < n_cols && st == GWI_OK; ++c)
    if (hipMalloc(&d[c], sizeof(double) * (size_t)(n ? n : 1)) != hipSuccess) st = GWI_ERR_HIP;
  if (st == GWI_OK) st = ingest_run(err, prog, n, n_cols, d.data(), nullptr);
  for (int c = 0; c < n_cols && st == GWI_OK; ++c)
    if (n && hipMemcpy(cols[c], d[c], sizeof(double) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) st = GWI_ERR_HIP;
  for (double* q : d) (void)hipFree(q);
  if (st != GWI_OK && !err.empty()) std::fprintf(stderr, "gwi_ingest_columns: %s\n", err.c_str());
  return st;
}

gwi_status gwi_read_column(gwi_handle h, int32_t pe_side, int32_t col, double* out) {
  if (!h || !out || h->host_only) return GWI_ERR_INVALID;
  const std::vector<double*>& cols = pe_side ? h->d_cols_pe : h->d_cols_inj;
  if (col < 0 || col >= (int)cols.size()) return fail(h, GWI_ERR_INVALID, "gwi_read_column: column out of range");
  const size_t n = pe_side ? (size_t)(h->n_ev * h->n_pe) : (size_t)h->n_inj;
  GWI_HIP(hipSetDevice(h->device));
  if (n) GWI_HIP(hipMemcpy(out, cols[col], sizeof(double) * n, hipMemcpyDeviceToHost));
  return GWI_OK;
}

// Pin the calling thread to the CPUs next to a GPU (sysfs local_cpulist of its PCI function), within what the thread is
// allowed already.  An evaluation is a handful of PCIe round trips (arguments out through the BAR, results and stamps
// polled in pinned memory): from the far socket each costs more (config 2: 18.8 vs 17.5 us per evaluation).
static gwi_status pin_thread_to_device(int device, std::string& why) {
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) {
    why = "hipDeviceGetPCIBusId failed";
    return GWI_ERR_NO_DEVICE;
  }
  for (char* c = bus; *c; ++c) *c = (char)std::tolower((unsigned char)*c);
  const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/local_cpulist";
  FILE* f = std::fopen(path.c_str(), "r");
  char line[4096] = {0};
  const bool got = f && std::fgets(line, sizeof(line), f) != nullptr;
  if (f) std::fclose(f);
  if (!got) {
    why = path + " is not readable";
    return GWI_ERR_UNSUPPORTED;
  }
  cpu_set_t allowed, want;
  CPU_ZERO(&want);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) {
    why = "sched_getaffinity failed";
    return GWI_ERR_UNSUPPORTED;
  }
  int n_set = 0;
  for (char* tok = std::strtok(line, ",\n"); tok; tok = std::strtok(nullptr, ",\n")) {  // "0-63,128-191"
    int a = 0, b = 0;
    const int fields = std::sscanf(tok, "%d-%d", &a, &b);
    if (fields < 1) continue;
    if (fields == 1) b = a;
    for (int c = a; c <= b && c < CPU_SETSIZE; ++c)
      if (CPU_ISSET(c, &allowed)) {
        CPU_SET(c, &want);
        ++n_set;
      }
  }
  if (n_set == 0 || sched_setaffinity(0, sizeof(want), &want) != 0) {
    why = "none of the GPU's local CPUs is available to this thread";
    return GWI_ERR_UNSUPPORTED;
  }
  return GWI_OK;
}

gwi_status gwi_pin_thread_to_device(int32_t device) {
  std::string why;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return GWI_ERR_NO_DEVICE;
  if (device < 0) (void)hipGetDevice(&device);
  return pin_thread_to_device(device, why);
}

gwi_status gwi_pin_thread_to_engine(gwi_handle h) {
  if (!h || h->host_only) return GWI_ERR_INVALID;
  std::string why;
  const gwi_status st = pin_thread_to_device(h->device, why);
  if (st != GWI_OK) h->err = why;
  return st;
}

gwi_status gwi_hbm_bandwidth(int32_t device, int64_t n_doubles, int32_t iters, double* read_gbs, double* triad_gbs) {
  if (n_doubles < 1024 || iters < 1 || !read_gbs || !triad_gbs) return GWI_ERR_INVALID;
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return GWI_ERR_NO_DEVICE;
  if (device == GWI_DEVICE_CURRENT && hipGetDevice(&device) != hipSuccess) return GWI_ERR_HIP;
  if (device < 0 || device >= n_dev || hipSetDevice(device) != hipSuccess) return GWI_ERR_INVALID;
  const long long n2 = n_doubles / 2;
  double2 *a = nullptr, *b = nullptr, *c = nullptr;
  double* out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t st = nullptr;
  gwi_status rc = GWI_ERR_HIP;
  do {
    if (hipMalloc(&a, sizeof(double2) * n2) != hipSuccess || hipMalloc(&b, sizeof(double2) * n2) != hipSuccess || hipMalloc(&c, sizeof(double2) * n2) != hipSuccess ||
        hipMalloc(&out, 64 * sizeof(double)) != hipSuccess)
      break;
    if (hipStreamCreate(&st) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) break;
    if (hipMemsetAsync(a, 0, sizeof(double2) * n2, st) != hipSuccess || hipMemsetAsync(b, 0, sizeof(double2) * n2, st) != hipSuccess ||
        hipMemsetAsync(c, 0, sizeof(double2) * n2, st) != hipSuccess || hipMemsetAsync(out, 0, 64 * sizeof(double), st) != hipSuccess)
      break;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) break;
    const unsigned grid = (unsigned)prop.multiProcessorCount * 32u;
    float best_r = 1e30f, best_t = 1e30f;
    bool ok = true;
    for (int it = 0; it < iters + 2 && ok; ++it) {  // two untimed warm-up rounds
      float ms = 0.0f;
      ok = ok && hipEventRecord(e0, st) == hipSuccess;
      hipLaunchKernelGGL(bw_read_kernel, dim3(grid), dim3(kBlock), 0, st, (const double2*)b, n2, out, (int)grid);
      ok = ok && hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
      if (it >= 2 && ms < best_r) best_r = ms;
      ok = ok && hipEventRecord(e0, st) == hipSuccess;
      hipLaunchKernelGGL(bw_triad_kernel, dim3(grid), dim3(kBlock), 0, st, a, (const double2*)b, (const double2*)c, 3.0, n2, (int)grid);
      ok = ok && hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
      if (it >= 2 && ms < best_t) best_t = ms;
    }
    if (!ok || hipGetLastError() != hipSuccess) break;
    *read_gbs = 16.0 * (double)n2 / ((double)best_r * 1e-3) / 1e9;
    *triad_gbs = 48.0 * (double)n2 / ((double)best_t * 1e-3) / 1e9;
    rc = GWI_OK;
  } while (false);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  (void)hipFree(a);
  (void)hipFree(b);
  (void)hipFree(c);
  (void)hipFree(out);
  return rc;
}

const char* gwi_dispatch_info(gwi_handle h) {
  if (!h) return "no engine";
  return h->aql_active ? (h->aq.failed() ? h->aq.why().c_str() : "aql: active") : h->aql_note.c_str();
}

gwi_status gwi_launch_geometry(gwi_handle h, int32_t out[6]) {
  if (!h || !out || h->host_only) return GWI_ERR_INVALID;
  out[0] = h->chunk_pe;
  out[1] = h->chunk_inj;
  out[2] = h->tiles_per_event;
  out[3] = h->n_inj_tiles;
  out[4] = h->n_scan_blocks;
  out[5] = h->n_inj_groups;
  return GWI_OK;
}

gwi_status gwi_set_timing(gwi_handle h, int32_t enabled) {
  if (!h) return GWI_ERR_INVALID;
  h->timing = enabled != 0;
  h->force_hip_stream = enabled == 2;
  return GWI_OK;
}

gwi_status gwi_last_kernel_ms(gwi_handle h, float ms[3]) {
  if (!h || !ms) return GWI_ERR_INVALID;
  for (int i = 0; i < 3; ++i) ms[i] = h->last_ms[i];
  return GWI_OK;
}

int64_t gwi_partial_len(gwi_handle h) { return h ? record_len(h) : 0; }
int64_t gwi_two_pass_repeats(gwi_handle h) { return h ? h->redo_count : 0; }
const char* gwi_batch_path(gwi_handle h, int32_t k_batch) {
  if (!h || h->host_only) return "none";
  const bool safe = h->variant && h->variant->has(jit::kSafe) && h->kargs.deterministic;
  if (h->variant && h->variant->has(jit::kPbatch) && !h->generic) {  // parametric model: one load per sample where the tiles of a launch of k_batch points are single trips
    const long long gran = (long long)pbatch_u(h->variant->samples_per_lane) * kBlock;
    const bool bg = k_batch >= 4 && h->bgeo.distinct;
    if (h->pbatch && (bg ? h->bgeo.chunk_pe : h->chunk_pe) <= gran && (bg ? h->bgeo.chunk_inj : h->chunk_inj) <= gran &&
        ((h->pbatch_balanced && k_batch >= kPbatchBalancedFrom) || pbatch_points(h, k_batch, bg ? 1 : 0) > 1))
      return "pbatch";
    return "rows-per-point";
  }
  return (h->mfma && !safe && k_batch >= h->mfma_min_batch) ? (h->batch_rows ? "rows" : "mfma") : "taps";
}

const char* gwi_batch_kernel_note(gwi_handle h) { return h ? h->mfma_jit_note.c_str() : ""; }

gwi_status gwi_batch_calibration(gwi_handle h, int32_t* measured, double* mfma_us, double* taps_us) {
  if (!h) return GWI_ERR_INVALID;
  if (measured) *measured = h->batch_measured ? 1 : 0;
  if (mfma_us) *mfma_us = h->batch_us[0];
  if (taps_us) *taps_us = h->batch_us[1];
  return GWI_OK;
}

gwi_status gwi_jit_compile(const int32_t* kinds, int32_t n_kinds, int32_t samples_per_lane, char* path_out, int64_t path_cap, double* compile_seconds, int32_t* from_cache) {
  if (!kinds || n_kinds < 1 || n_kinds > GWI_MAX_TERMS) return GWI_ERR_INVALID;
  int ks[GWI_MAX_TERMS];
  for (int t = 0; t < n_kinds; ++t) ks[t] = kinds[t];
  std::string why;
  jit::Chain* c = jit::get_chain(ks, n_kinds, samples_per_lane, gwi_embedded_device_h, gwi_embedded_engine_h, why, samples_per_lane == 0 ? gwi_embedded_mfma_h : nullptr);
  if (!c) {
    std::fprintf(stderr, "gwi_jit_compile: %s\n", why.c_str());
    if (path_out && path_cap > 0) std::snprintf(path_out, (size_t)path_cap, "%s", why.c_str());
    return why.find("kinds") != std::string::npos ? GWI_ERR_INVALID : GWI_ERR_UNSUPPORTED;
  }
  if (path_out && path_cap > 0) std::snprintf(path_out, (size_t)path_cap, "%s", c->path.c_str());
  if (compile_seconds) *compile_seconds = c->compile_seconds;
  if (from_cache) *from_cache = c->from_cache ? 1 : 0;
  return GWI_OK;
}

gwi_status gwi_jit_info(gwi_handle h, int32_t* compiled_at_run_time, double* compile_seconds, int32_t* from_cache, const char** note) {
  if (!h) return GWI_ERR_INVALID;
  const jit::Chain* c = h->variant ? h->variant->jit : nullptr;
  if (compiled_at_run_time) *compiled_at_run_time = c ? 1 : 0;
  if (compile_seconds) *compile_seconds = c ? c->compile_seconds : 0.0;
  if (from_cache) *from_cache = (c && c->from_cache) ? 1 : 0;
  if (note) *note = h->jit_note.c_str();
  return GWI_OK;
}

gwi_status gwi_prepare_combine(gwi_handle h, const double* theta) {
  if (!h || !theta) return GWI_ERR_INVALID;
  prelude(h, theta, h->kargs.theta, h->kargs.derived, &h->host_consts[0]);
  return GWI_OK;
}

gwi_status gwi_eval_partial(gwi_handle h, const double* theta, double* record_host, double* log_bfs, double* log_neffs, double* variances) {
  if (!h || !theta || !h->variant) return GWI_ERR_INVALID;
  if (h->host_only) return fail(h, GWI_ERR_NO_DEVICE, "host-only handle: no device to evaluate on");
  gwi_status st = busy_guard(h, "gwi_eval_partial");
  if (st != GWI_OK) return st;
  GWI_HIP(hipSetDevice(h->device));
  st = run_pipeline(h, theta);
  if (st != GWI_OK) return st;
  if (record_host) std::memcpy(record_host, h->h_record, sizeof(double) * record_len(h));
  const size_t n = (size_t)h->n_ev;
  // per-event sites without the global constant (added by the caller after gwi_combine)
  if (log_bfs) std::memcpy(log_bfs, h->h_ev, sizeof(double) * n);
  if (log_neffs) std::memcpy(log_neffs, h->h_ev + n, sizeof(double) * n);
  if (variances) std::memcpy(variances, h->h_ev + 2 * n, sizeof(double) * n);
  return GWI_OK;
}

gwi_status gwi_combine(gwi_handle h, const double* records, int32_t n_ranks, const gwi_options* opt, gwi_summary* summary, double* grad, double* norms) {
  if (!h || !records || n_ranks < 1 || !opt) return GWI_ERR_INVALID;
  if (opt->max_variance_cut && (opt->marginalize_selection || opt->min_neff_cut))
    return fail(h, GWI_ERR_INVALID, "max_variance_cut requires marginalize_selection and min_neff_cut to be off (analysis.py:237-243)");
  if (opt->marginalize_selection && grad)
    return fail(h, GWI_ERR_UNSUPPORTED, "gradient with marginalize_selection=True needs the squared-weight records, which the caller-exchanged path (gwi_eval_partial / gwi_combine) does not carry: use gwi_eval or gwi_eval_sharded");
  assemble(h, records, n_ranks, opt, summary, grad, norms, h->host_consts[0]);
  return GWI_OK;
}

gwi_status gwi_eval_begin(gwi_handle h, const double* theta, const gwi_options* opt, int32_t want_grad) {
  if (!h || !theta || !opt || !h->variant) return GWI_ERR_INVALID;
  if (h->pending) return fail(h, GWI_ERR_INVALID, "gwi_eval_begin: the previous evaluation of this handle has not been collected (gwi_eval_end)");
  if (h->poisoned) return busy_guard(h, "gwi_eval_begin");
  if (opt->max_variance_cut && (opt->marginalize_selection || opt->min_neff_cut))
    return fail(h, GWI_ERR_INVALID, "max_variance_cut requires marginalize_selection and min_neff_cut to be off (analysis.py:237-243)");
  if (h->host_only) return fail(h, GWI_ERR_NO_DEVICE, "host-only handle: no device to evaluate on");
  GWI_HIP(hipSetDevice(h->device));
  gwi_status st;
  h->pending_opt = *opt;
  h->pending_sq = opt->marginalize_selection && want_grad;
  if (h->pending_sq) {  // squared-weight pass first: the regular pass then leaves its per-event arrays in place
    st = run_pipeline(h, theta, nullptr, true, 1, false, /*square=*/true);
    if (st != GWI_OK) return st;
    h->sq_records.assign(h->h_record, h->h_record + record_len(h));
  }
  st = run_pipeline(h, theta, nullptr, /*wait=*/false);
  if (st != GWI_OK) return st;
  h->pending = true;
  return GWI_OK;
}

gwi_status gwi_eval_end(gwi_handle h, gwi_summary* summary, double* grad, double* log_bfs, double* log_neffs, double* variances, double* norms) {
  if (!h) return GWI_ERR_INVALID;
  if (!h->pending || h->pending_batch) return fail(h, GWI_ERR_INVALID, "gwi_eval_end without gwi_eval_begin");
  h->pending = false;
  GWI_HIP(hipSetDevice(h->device));
  gwi_status st;
  if (h->last_host_rows) {
    st = wait_for_rows(h, 1);
  } else {
    st = wait_for_stamp(h, h->h_fin, h->final_groups);
    if (st == GWI_OK) merge_final_records(h, 1);
    if (st == GWI_OK) st = wait_for_norms(h, h->h_record, 1);
  }
  if (st != GWI_OK) return st;
  if (redo_requested(h)) {  // repeat, blocking (the squared-weight pass, if any, ran through run_pipeline already)
    const std::vector<double> th(h->kargs.theta, h->kargs.theta + h->spec.n_theta);
    st = repeat_after_redo(h, th.data(), nullptr, 1, false, false);
    if (st != GWI_OK) return st;
  }
  gwi_summary s;
  assemble(h, h->h_record, 1, &h->pending_opt, &s, grad, norms, h->host_consts[0], (h->pending_sq && grad) ? h->sq_records.data() : nullptr);
  if (summary) *summary = s;
  const size_t n = (size_t)h->n_ev;
  const double shift = s.log_norm_const - std::log((double)h->n_pe);
  if (log_bfs)
    for (size_t i = 0; i < n; ++i) log_bfs[i] = h->h_ev[i] + shift;  // logBF_i = logsumexp_i - log N_pe (analysis.py:80)
  if (log_neffs) std::memcpy(log_neffs, h->h_ev + n, sizeof(double) * n);
  if (variances) std::memcpy(variances, h->h_ev + 2 * n, sizeof(double) * n);
  return GWI_OK;
}

gwi_status gwi_eval(gwi_handle h, const double* theta, const gwi_options* opt, gwi_summary* summary, double* grad, double* log_bfs, double* log_neffs,
                    double* variances, double* norms) {
  const gwi_status st = gwi_eval_begin(h, theta, opt, grad != nullptr);
  if (st != GWI_OK) return st;
  return gwi_eval_end(h, summary, grad, log_bfs, log_neffs, variances, norms);
}

gwi_status gwi_eval_batch_begin(gwi_handle h, const double* thetas, int32_t k_batch, const gwi_options* opt, int32_t want_grad, int32_t want_events) {
  if (!h || !thetas || !opt || !h->variant || k_batch < 1) return GWI_ERR_INVALID;
  if (k_batch > h->max_batch) return fail(h, GWI_ERR_INVALID, "k_batch exceeds the engine's max_batch (GWI_MAX_BATCH, default 16)");
  if (opt->max_variance_cut && (opt->marginalize_selection || opt->min_neff_cut))
    return fail(h, GWI_ERR_INVALID, "max_variance_cut requires marginalize_selection and min_neff_cut to be off (analysis.py:237-243)");
  if (h->host_only) return fail(h, GWI_ERR_NO_DEVICE, "host-only handle: no device to evaluate on");
  gwi_status st = busy_guard(h, "gwi_eval_batch_begin");
  if (st != GWI_OK) return st;
  GWI_HIP(hipSetDevice(h->device));
  const size_t len = (size_t)record_len(h);
  h->pending_opt = *opt;
  h->pending_sq = opt->marginalize_selection && want_grad;
  h->pending_k = k_batch;
  h->pending_thetas.assign(thetas, thetas + (size_t)k_batch * h->spec.n_theta);  // a repeat (reference exponent outrun) needs the points again
  h->batch_events = want_events != 0;
  // a matrix-core kernel still to be compiled is compiled on the first batched launch, and (GWI_BATCH_AUTOTUNE=1 only) measured
  // against the 4-tap kernel: blocking, here
  if (h->mfma_jit_pending && k_batch >= h->mfma_min_batch) h->batch_autotune = try_jit_mfma(h) && h->autotune_wanted;
  if (h->batch_autotune && h->mfma && k_batch >= h->mfma_min_batch) {
    st = calibrate_batch_path(h, thetas, k_batch);
    if (st != GWI_OK) return st;
  }
  if (h->pending_sq) {  // squared-weight pass first, blocking: the regular pass then leaves its per-event arrays in place
    st = run_pipeline(h, thetas, nullptr, true, k_batch, true, /*square=*/true);
    if (st != GWI_OK) return st;
    h->sq_records.assign(h->h_record, h->h_record + len * k_batch);
  }
  st = run_pipeline(h, thetas, nullptr, /*wait=*/false, k_batch, true);
  if (st != GWI_OK) return st;
  h->pending = true;
  h->pending_batch = true;
  return GWI_OK;
}

gwi_status gwi_eval_batch_end(gwi_handle h, gwi_summary* summaries, double* grads, double* log_bfs, double* log_neffs, double* variances, double* norms) {
  if (!h) return GWI_ERR_INVALID;
  if (!h->pending || !h->pending_batch) return fail(h, GWI_ERR_INVALID, "gwi_eval_batch_end without gwi_eval_batch_begin");
  h->pending = false;
  h->pending_batch = false;
  GWI_HIP(hipSetDevice(h->device));
  const int K = h->pending_k;
  gwi_status st;
  if (h->last_host_rows) {
    st = wait_for_rows(h, K);
  } else {
    st = wait_for_stamp(h, h->h_fin, K * h->final_groups);
    if (st == GWI_OK) merge_final_records(h, K);
    if (st == GWI_OK) st = wait_for_norms(h, h->h_record, K);
  }
  if (st != GWI_OK) return st;
  if (redo_requested(h)) {  // repeat, blocking (the squared-weight pass, if any, went through run_pipeline already)
    st = repeat_after_redo(h, h->pending_thetas.data(), nullptr, K, true, false);
    if (st != GWI_OK) return st;
  }
  const size_t n = (size_t)h->n_ev, len = (size_t)record_len(h);
  const gwi_options* opt = &h->pending_opt;
  const bool need_sq = h->pending_sq && grads;
  const int n_theta = h->spec.n_theta, n_norms = h->spec.n_norms;
  for (int k = 0; k < K; ++k) {
    gwi_summary s;
    assemble(h, h->h_record + k * len, 1, opt, &s, grads ? grads + (size_t)k * n_theta : nullptr, norms ? norms + (size_t)k * n_norms : nullptr, h->host_consts[k],
             need_sq ? h->sq_records.data() + k * len : nullptr);
    if (summaries) summaries[k] = s;
    const double* ev = h->h_ev + (size_t)k * 3 * n;
    const double shift = s.log_norm_const - std::log((double)h->n_pe);
    if (log_bfs)
      for (size_t i = 0; i < n; ++i) log_bfs[k * n + i] = ev[i] + shift;
    if (log_neffs) std::memcpy(log_neffs + k * n, ev + n, sizeof(double) * n);
    if (variances) std::memcpy(variances + k * n, ev + 2 * n, sizeof(double) * n);
  }
  return GWI_OK;
}

gwi_status gwi_eval_batch(gwi_handle h, const double* thetas, int32_t k_batch, const gwi_options* opt, gwi_summary* summaries, double* grads,
                          double* log_bfs, double* log_neffs, double* variances, double* norms) {
  const gwi_status st = gwi_eval_batch_begin(h, thetas, k_batch, opt, grads != nullptr, (log_bfs || log_neffs || variances) ? 1 : 0);
  if (st != GWI_OK) return st;
  return gwi_eval_batch_end(h, summaries, grads, log_bfs, log_neffs, variances, norms);
}

gwi_status gwi_comm_unique_id(const char* rccl_path, void* id128) {
  if (!id128) return GWI_ERR_INVALID;
  std::string err;
  if (!load_nccl(rccl_path, &err)) return GWI_ERR_HIP;
  return g_nccl.GetUniqueId(id128) == 0 ? GWI_OK : GWI_ERR_HIP;
}

gwi_status gwi_comm_init(gwi_handle h, const char* rccl_path, const void* id128, int32_t rank, int32_t world) {
  if (!h || !id128 || world < 1 || rank < 0 || rank >= world) return GWI_ERR_INVALID;
  if (h->host_only) return fail(h, GWI_ERR_NO_DEVICE, "host-only handle: no device to communicate from");
  if (!load_nccl(rccl_path, &h->err)) return GWI_ERR_HIP;
  GWI_HIP(hipSetDevice(h->device));
  NcclId id;
  std::memcpy(id.b, id128, 128);
  void* comm = nullptr;
  const int rc = g_nccl.CommInitRank(&comm, world, id, rank);
  if (rc != 0) return fail(h, GWI_ERR_HIP, std::string("ncclCommInitRank: ") + (g_nccl.GetErrorString ? g_nccl.GetErrorString(rc) : "error"));
  h->nccl_comm = comm;
  h->comm_rank = rank;
  h->comm_world = world;
  const size_t len = (size_t)record_len(h);
  GWI_HIP(hipMalloc(&h->d_send, sizeof(double) * len));
  GWI_HIP(hipMalloc(&h->d_recv, sizeof(double) * len * world));
  GWI_HIP(hipHostMalloc((void**)&h->h_gather, sizeof(double) * len * world, hipHostMallocMapped));
  GWI_HIP(hipHostGetDevicePointer((void**)&h->h_gather_dev, h->h_gather, 0));
  std::memset(h->h_gather, 0, sizeof(double) * len * world);
  return GWI_OK;
}


// ---- single-node record exchange through POSIX shared memory ------------------------------------------------------
// Segment: [2 parities][world ranks] slots of { u64 stamp; double record[len] }, each padded to a multiple of 128 B.
// Exchange s (s = 1, 2, ...): write the record into slot [s & 1][rank], release-store the stamp s, then acquire-poll
// the stamps of all ranks.  Two parities suffice: a rank can only reach exchange s + 2 after every rank has published
// s + 1, i.e. after every rank has finished READING exchange s.
gwi_status gwi_shm_comm_unlink(const char* name) {
  if (!name || !*name) return GWI_ERR_INVALID;
  return shm_unlink(name) == 0 ? GWI_OK : GWI_ERR_INVALID;
}

gwi_status gwi_shm_comm_init(gwi_handle h, const char* name, int32_t rank, int32_t world) {
  if (!h || !name || !*name || world < 1 || rank < 0 || rank >= world) return GWI_ERR_INVALID;
  if (h->shm_base) return fail(h, GWI_ERR_INVALID, "gwi_shm_comm_init: already attached");
  const size_t len = (size_t)record_len(h);
  const size_t slot = ((sizeof(unsigned long long) + sizeof(double) * len + 127) / 128) * 128;
  const size_t bytes = slot * 2 * (size_t)world;
  const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return fail(h, GWI_ERR_INVALID, std::string("shm_open(") + name + "): " + std::strerror(errno));
  struct stat sb;
  // every rank sizes the (zero-filled) segment to the same length; whoever comes later finds it sized already
  if (fstat(fd, &sb) != 0 || ((size_t)sb.st_size != bytes && ftruncate(fd, (off_t)bytes) != 0)) {
    const std::string why = std::strerror(errno);
    close(fd);
    return fail(h, GWI_ERR_INVALID, std::string("sizing shared-memory segment ") + name + ": " + why);
  }
  void* base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (base == MAP_FAILED) return fail(h, GWI_ERR_INVALID, std::string("mmap of shared-memory segment ") + name + ": " + std::strerror(errno));
  h->shm_base = static_cast<char*>(base);
  h->shm_bytes = bytes;
  h->shm_slot_bytes = slot;
  h->shm_rank = rank;
  h->shm_world = world;
  h->shm_seq = 0;
  h->shm_gather.assign(len * (size_t)world, 0.0);
  h->comm_rank = rank;
  h->comm_world = world;
  return GWI_OK;
}

gwi_status gwi_shm_exchange(gwi_handle h, const double* record, double* gathered) {
  if (!h || !record || !gathered) return GWI_ERR_INVALID;
  if (!h->shm_base) return fail(h, GWI_ERR_INVALID, "gwi_shm_comm_init has not been called");
  const size_t len = (size_t)record_len(h);
  const unsigned long long s = ++h->shm_seq;
  char* const bank = h->shm_base + (size_t)(s & 1) * h->shm_slot_bytes * (size_t)h->shm_world;
  char* mine = bank + (size_t)h->shm_rank * h->shm_slot_bytes;
  std::memcpy(mine + sizeof(unsigned long long), record, sizeof(double) * len);
  __atomic_store_n(reinterpret_cast<unsigned long long*>(mine), s, __ATOMIC_RELEASE);
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < h->shm_world; ++r) {
    const char* theirs = bank + (size_t)r * h->shm_slot_bytes;
    const unsigned long long* stamp = reinterpret_cast<const unsigned long long*>(theirs);
    for (unsigned long long spin = 1; __atomic_load_n(stamp, __ATOMIC_ACQUIRE) != s; ++spin) {
      __builtin_ia32_pause();
      if ((spin & 0xfffff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0)
        return fail(h, GWI_ERR_TIMEOUT, "shared-memory exchange: rank " + std::to_string(r) + " did not publish exchange " + std::to_string(s) + " within 60 s");
    }
    std::memcpy(gathered + (size_t)r * len, theirs + sizeof(unsigned long long), sizeof(double) * len);
  }
  return GWI_OK;
}

gwi_status gwi_eval_sharded(gwi_handle h, const double* theta, const gwi_options* opt, gwi_summary* summary, double* grad, double* log_bfs,
                            double* log_neffs, double* variances, double* norms) {
  if (!h || !theta || !opt || !h->variant) return GWI_ERR_INVALID;
  if (!h->nccl_comm && !h->shm_base) return fail(h, GWI_ERR_INVALID, "neither gwi_shm_comm_init nor gwi_comm_init has been called");
  if (opt->max_variance_cut && (opt->marginalize_selection || opt->min_neff_cut))
    return fail(h, GWI_ERR_INVALID, "max_variance_cut requires marginalize_selection and min_neff_cut to be off (analysis.py:237-243)");
  if (h->host_only) return fail(h, GWI_ERR_NO_DEVICE, "host-only handle: no device to evaluate on");
  gwi_status st = busy_guard(h, "gwi_eval_sharded");
  if (st != GWI_OK) return st;
  GWI_HIP(hipSetDevice(h->device));
  const size_t len = (size_t)record_len(h);
  const double* const gathered = h->shm_base ? h->shm_gather.data() : h->h_gather;
  auto run = [&](bool square) -> gwi_status {
    if (h->shm_base) {
      // this rank's shard through the regular fast path (AQL dispatch, host-final where it applies): its record ends up
      // in host memory anyway, so the exchange is a publish + poll between host cores of the node -- no collective launch
      gwi_status st_ = run_pipeline(h, theta, nullptr, /*wait=*/true, 1, false, square);
      if (st_ != GWI_OK) return st_;
      return gwi_shm_exchange(h, h->h_record, h->shm_gather.data());
    }
    // scan -> combine -> final (record stays on the device) -> all-gather -> publish, all on one stream
    gwi_status st_ = run_pipeline(h, theta, h->d_send, /*wait=*/false, 1, false, square);
    if (st_ != GWI_OK) return st_;
    const int rc = g_nccl.AllGather(h->d_send, h->d_recv, len, kNcclDouble, h->nccl_comm, h->stream);
    if (rc != 0) return fail(h, GWI_ERR_HIP, std::string("ncclAllGather: ") + (g_nccl.GetErrorString ? g_nccl.GetErrorString(rc) : "error"));
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(kBlock), 0, h->stream, h->d_recv, h->h_gather_dev, (int)(len * h->comm_world), h->seq);
    GWI_HIP(hipGetLastError());
    st_ = wait_for_stamp(h, h->h_gather);
    if (st_ != GWI_OK) return st_;
    return wait_for_norms(h, h->h_gather);  // every rank integrates the same grids; rank-0 slots are what assemble() reads
  };
  // in-engine RCCL path: a rank whose scan asked for the two-pass repeat marked its record (negative event count), so
  // every rank sees the request in the gathered records and all repeat the exchange together
  auto run_checked = [&](bool square) -> gwi_status {
    gwi_status st_ = run(square);
    if (st_ != GWI_OK || h->shm_base) return st_;
    bool redo = false;
    for (int r = 0; r < h->comm_world; ++r) redo = redo || h->h_gather[(size_t)r * len + 7] < 0.0;
    if (!redo) return GWI_OK;
    // every rank repeats the exchange (the ranks whose tiles were fine find their references exact as well)
    ++h->redo_count;
    st_ = run(square);
    if (st_ != GWI_OK) return st_;
    redo = false;
    for (int r = 0; r < h->comm_world; ++r) redo = redo || h->h_gather[(size_t)r * len + 7] < 0.0;
    if (!redo || !h->variant->has(jit::kSafe)) return GWI_OK;
    h->kargs.two_pass = 1;
    st_ = run(square);
    h->kargs.two_pass = 0;
    return st_;
  };
  const bool need_sq = opt->marginalize_selection && grad;
  if (need_sq) {  // a second exchange carries the squared-weight numerators
    st = run_checked(true);
    if (st != GWI_OK) return st;
    h->sq_records.assign(gathered, gathered + len * h->comm_world);
  }
  st = run_checked(false);
  if (st != GWI_OK) return st;
  gwi_summary s;
  assemble(h, gathered, h->comm_world, opt, &s, grad, norms, h->host_consts[0], need_sq ? h->sq_records.data() : nullptr);
  if (summary) *summary = s;
  const size_t n = (size_t)h->n_ev;
  const double shift = s.log_norm_const - std::log((double)h->n_pe);
  if (log_bfs)
    for (size_t i = 0; i < n; ++i) log_bfs[i] = h->h_ev[i] + shift;
  if (log_neffs) std::memcpy(log_neffs, h->h_ev + n, sizeof(double) * n);
  if (variances) std::memcpy(variances, h->h_ev + 2 * n, sizeof(double) * n);
  return GWI_OK;
}

gwi_status gwi_selftime(gwi_handle h, const double* theta, const gwi_options* opt, int32_t n_iter, double* seconds_per_eval) {
  if (!h || !theta || !opt || n_iter < 1 || !seconds_per_eval) return GWI_ERR_INVALID;
  std::vector<double> grad(h->spec.n_theta);
  gwi_summary s;
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n_iter; ++i) {
    gwi_status st = gwi_eval(h, theta, opt, &s, grad.data(), nullptr, nullptr, nullptr, nullptr);
    if (st != GWI_OK) return st;
  }
  *seconds_per_eval = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / n_iter;
  return GWI_OK;
}

// A given trajectory of hyper-parameter points, one blocking evaluation after the other (what a sampler's
// leapfrog loop does between two of its own few flops), without a host-language binding in the loop.
gwi_status gwi_eval_sequence(gwi_handle h, const double* thetas, int32_t n, const gwi_options* opt, double* log_likelihoods, double* grads, int32_t timing_every,
                             float* kernel_ms) {
  if (!h || !thetas || !opt || n < 1 || !log_likelihoods) return GWI_ERR_INVALID;
  const int nt = h->spec.n_theta;
  std::vector<double> scratch(grads ? 0 : nt);
  const bool was_timing = h->timing;
  gwi_summary s;
  for (int i = 0; i < n; ++i) {
    const bool timed = kernel_ms && timing_every > 0 && (i % timing_every == 0);
    h->timing = timed;
    double* g = grads ? grads + (size_t)i * nt : scratch.data();
    const double* th = thetas + (size_t)i * nt;
    const gwi_status st = (h->nccl_comm || h->shm_base) ? gwi_eval_sharded(h, th, opt, &s, g, nullptr, nullptr, nullptr, nullptr) : gwi_eval(h, th, opt, &s, g, nullptr, nullptr, nullptr, nullptr);
    if (st != GWI_OK) {
      h->timing = was_timing;
      return st;
    }
    log_likelihoods[i] = s.log_likelihood;
    if (kernel_ms)
      for (int k = 0; k < 3; ++k) kernel_ms[(size_t)i * 3 + k] = timed ? h->last_ms[k] : -1.0f;
  }
  h->timing = was_timing;
  return GWI_OK;
}

gwi_status gwi_eval_latencies(gwi_handle h, const double* thetas, int32_t n, const gwi_options* opt, double* seconds) {
  if (!h || !thetas || !opt || n < 1 || !seconds) return GWI_ERR_INVALID;
  const int nt = h->spec.n_theta;
  std::vector<double> g(nt);
  gwi_summary s;
  const bool sharded = h->nccl_comm || h->shm_base;
  for (int i = 0; i < n; ++i) {
    const double* th = thetas + (size_t)i * nt;
    const auto t0 = std::chrono::steady_clock::now();
    const gwi_status st = sharded ? gwi_eval_sharded(h, th, opt, &s, g.data(), nullptr, nullptr, nullptr, nullptr) : gwi_eval(h, th, opt, &s, g.data(), nullptr, nullptr, nullptr, nullptr);
    seconds[i] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (st != GWI_OK) return st;
  }
  return GWI_OK;
}

#ifdef GWI_HOST_PHASES
extern "C" void gwi_debug_host_phases(double* out6) {
  for (int i = 0; i < 5; ++i) out6[i] = g_phase[i];
  out6[5] = (double)g_phase_calls;
  for (int i = 0; i < 5; ++i) g_phase[i] = 0;
  g_phase_calls = 0;
}
#endif

#ifdef GWI_STAMPS
// diagnostic build only (not part of the ABI): fetch the per-wave phase stamps of the last scan launch
gwi_status gwi_debug_stamps(gwi_handle h, unsigned long long* out, int64_t n_words) {
  if (!h || !out) return GWI_ERR_INVALID;
  const int64_t have = (int64_t)h->n_scan_blocks * kWaves * 8;
  GWI_HIP(hipMemcpy(out, h->kargs.stamps, sizeof(unsigned long long) * (size_t)(n_words < have ? n_words : have), hipMemcpyDeviceToHost));
  return GWI_OK;
}
#endif

gwi_status gwi_log_weights(gwi_handle h, const double* theta, double* pe_logw, double* inj_logw) {
  if (!h || !theta || !h->variant) return GWI_ERR_INVALID;
  if (h->host_only) return fail(h, GWI_ERR_NO_DEVICE, "host-only handle: no device to evaluate on");
  gwi_status st = busy_guard(h, "gwi_log_weights");
  if (st != GWI_OK) return st;
  GWI_HIP(hipSetDevice(h->device));
  const size_t n_pe_tot = (size_t)(h->n_ev * h->n_pe), n_inj = (size_t)h->n_inj;
  if (!h->d_logw_pe) GWI_HIP(hipMalloc(&h->d_logw_pe, sizeof(double) * (n_pe_tot ? n_pe_tot : 1)));
  if (!h->d_logw_inj) GWI_HIP(hipMalloc(&h->d_logw_inj, sizeof(double) * (n_inj ? n_inj : 1)));
  // normaliser values come from a regular evaluation
  st = run_pipeline(h, theta);
  if (st != GWI_OK) return st;
  double log_const = h->host_consts[0];
  const double* nrm = h->h_record + kRecNormOff;
  for (int t = 0; t < h->spec.n_terms; ++t)
    if (h->spec.terms[t].norm >= 0) log_const -= std::log(nrm[h->spec.terms[t].norm]);
  h->kargs.logw_pe = h->d_logw_pe;
  h->kargs.logw_inj = h->d_logw_inj;
  set_geometry(h, false);
  st = launch_scan(h, true);
  if (st != GWI_OK) return st;
  GWI_HIP(hipStreamSynchronize(h->stream));
  if (pe_logw) {
    GWI_HIP(hipMemcpy(pe_logw, h->d_logw_pe, sizeof(double) * n_pe_tot, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n_pe_tot; ++i) pe_logw[i] += log_const;
  }
  if (inj_logw) {
    GWI_HIP(hipMemcpy(inj_logw, h->d_logw_inj, sizeof(double) * n_inj, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n_inj; ++i) inj_logw[i] += log_const;
  }
  return GWI_OK;
}

}  // extern "C"
