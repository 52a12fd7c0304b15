// gwi_device.h -- device side of the population-likelihood engine (gfx950 / CDNA4 only).
//
// One fused "scan" launch streams the catalog columns once and produces, per workgroup, a partial
// record (running max m, S1 = sum e^{l-m}, S2 = sum e^{2(l-m)}, G[p] = sum e^{l-m} dl/dtheta_p).
// A small launch then combines records per event (and, for large problems, a third sums over
// events); the grid normalisers are integrated by the first workgroups of the scan launch.  Everything is fp64.
//
// Reference arithmetic being replaced (paths relative to the reference root):
//   per-sample densities      gwinferno/distributions.py:100-162, models/parametric/parametric.py:27-145
//   B-spline projection       gwinferno/interpolation.py:293-304, 381-394 (dense GEMV in the reference;
//                             here 4 taps per sample recomputed in registers, uniform knots :98-106)
//   masked scatter            models/bsplines/single.py:77-109 (here: kappa = -inf)
//   reductions                pipeline/analysis.py:50-136
#pragma once
#ifndef __HIPCC_RTC__  // hipRTC (gwi_jit.h: chains compiled at gwi_create) brings the device runtime with it and has no system headers
#include <hip/hip_runtime.h>
#endif

#include "gwi_engine.h"

namespace gwi {

constexpr int kBlock = 256;            // 4 wavefronts of 64
constexpr int kWaves = kBlock / 64;
constexpr int kMaxDerived = 7;          // PLPEAK uses d0..d6; 7 keeps KArgs with 256 hyper-parameters inside 4 KiB
constexpr int kRecHeader = 3;          // m, S1, S2 precede the gradient numerators in a record

struct TermD {
  int kind, n_basis;
  int th0, th1, th2, th3;  // EXP_SPLINE: th0 = coef_off
  int flags, th4;     // th4: fifth hyper-parameter (PLPEAK_SMOOTH: delta)
  double p0, p1, p2;  // spline kinds: lo, hi, 1/dx of the spline coordinate
  double p3;          // spline kinds: number of knot intervals, n_basis - 3 (the closed domain in knot coordinates is [0, p3])
};
static_assert(sizeof(TermD) == 64, "twelve of these sit in the 4 KiB kernel-argument block");

struct NormD {
  int n_pts, expo_theta, n_basis, coef_off, flags, pad;
  double expo_add, lo, hi;
  const double* tw;
  const double* lb;
  const double* l1;
  const double* us;
};

// hyper-parameter point + its host-precomputed theta-only scalars; a batched launch reads one per
// blockIdx.y from device memory, a single evaluation reads the copy embedded in the kernel arguments
struct ThetaBlock {
  double theta[GWI_MAX_THETA];
  double derived[GWI_MAX_TERMS][kMaxDerived];
};

// Arguments of the two launches after the scan: combine_kernel (tile records of a group -- an event,
// or a run of injection tiles -- to one result per group) and final_kernel (sum over groups).
struct TailArgs {
  const double* partials;
  double* ev_out;     // [n_ev][4]: logsumexp (= log sum_j w_ij, no -log N_pe), log n_eff, variance, S1
  double* ev_grad;    // [n_ev][n_theta]: G_p / S1
  double* inj_out;    // [n_inj_groups][4]: M, S1, S2
  double* inj_grad;   // [n_inj_groups][n_theta]: G_p relative to that group's M
  double* ev_host;    // pinned host [3][n_ev]: logsumexp, log n_eff, variance
  // host-final mode (small problems): every group publishes its whole result row (a, b, c, grad[n_theta]) to pinned host
  // memory and the HOST sums over groups; a = logsumexp | M, b = log n_eff | S1, c = variance | S2.  The row travels as
  // 64-byte LINES of seven values + the evaluation's sequence number in the eighth slot: every line validates itself, so a
  // row is one store instruction per 8 lines with nothing to drain and no separate stamp write behind it
  double* host_rows;  // nullptr: device-final mode
  double* record;     // device-final mode: pinned host record (or the device send buffer when sharded)
  // completion stamp of this evaluation: read from a device word the scan launch of the same evaluation wrote
  // (KArgs::seq_dev), so that this argument block is CONSTANT across evaluations -- on the AQL path it is written
  // through the PCIe BAR once, at gwi_create, and needs no per-evaluation hand-off
  const unsigned long long* seq_ptr;
  const unsigned long long* redo_ptr;  // KArgs::redo_dev
  int n_ev, tiles_per_event, n_inj_tiles, n_inj_groups, tiles_per_inj_group, n_theta, rec_stride;
  int n_scan_blocks;  // records per hyper-parameter point (batched launches: blockIdx.y = point)
  int n_norms, record_len;
  int final_groups;          // workgroups of the final launch: each publishes a partial record
  int row_lines;             // host-final mode: 64-byte lines per result row
  int combine_threads;       // workgroup size of the combine launch (64 or kBlock): passed here, not read from blockDim, which would pull in implicit kernel arguments the AQL packets do not carry
  int publish_events;        // device-final mode: also store the per-event sites to pinned host memory (3 small PCIe writes per event)
  double n_pe;
};

struct KArgs {
  // column base pointers, already resolved per term by the host ([term][0|1]): ONE scalar load from the
  // kernel-argument block per pointer, not term descriptor -> column index -> pointer table
  const double* pe_tcols[GWI_MAX_TERMS][2];
  const double* inj_tcols[GWI_MAX_TERMS][2];
  const double* kappa_pe;
  const double* kappa_inj;
  const NormD* norms;
  double* partials;   // [n_scan_blocks][rec_stride]
  double* logw_pe;    // only for the log-weight variant
  double* logw_inj;
#ifdef GWI_STAMPS
  unsigned long long* stamps;  // diagnostic build only: [n_blocks][kWaves][8] s_memrealtime stamps
#endif
  long long n_pe;     // samples per event
  long long n_inj;
  int n_ev, tiles_per_event, chunk_pe, n_inj_tiles, chunk_inj, n_norms;
  int n_terms, n_theta, kappa_col, rec_stride;
  int gacc_rep, gacc_shift;  // spline-gradient LDS rows: replicas per coefficient (power of two <= 64) and log2 of it
  double* norm_out_host;               // pinned host: Z_j of hyper-parameter point k at [k * n_norms + j]
  unsigned long long* norm_stamps_host;  // pinned host: completion stamp per (k, j)
  unsigned long long* seq_dev;            // device word: the scan publishes norm_seq here for the tail launches
  unsigned long long* redo_host;          // pinned host word: a workgroup whose fixed reference exponent turned out too low stores norm_seq here
  unsigned long long* redo_dev;           // ... and here (device word, read by final_kernel: the sharded path's record carries the request to every rank)
  // spline models: the reference exponent (in binades: powers of two) every tile weighs its samples against = the tile's
  // exact maximum at the PREVIOUS evaluation of this handle, [1 + 2 max_batch][nref_stride] (row 0: single evaluations,
  // rows 1..: the points of a batched launch, one block of rows per launch geometry); kNoRef where there is none yet
  int* tile_nref;
  int nref_stride;
  int rows_rep;  // scan_rows_kernel (gwi_mfma.h): sample-slot replicas of its gradient rows (4, 2 or 1)
  TermD terms[GWI_MAX_TERMS];
  // ---- everything ABOVE is fixed at gwi_create; what follows changes from one evaluation to the next.  On the AQL path the
  //      block lives in a persistent kernel-argument slot in device memory and only this tail is rewritten through the PCIe
  //      BAR per evaluation (aql::dispatch_tail): a few hundred bytes instead of 3.5 KB
  unsigned long long norm_seq;            // = the evaluation's sequence number
  int two_pass, deterministic;            // two_pass: find each tile's exact maximum first; deterministic: waves take turns at the shared rows
  int square, k_batch;     // k_batch: hyper-parameter points of a batched launch (scan_mfma_kernel: 16 per grid row); square != 0: accumulate with w^2 instead of w (the sum_j w_j^2 dl_j/dtheta numerators
                           // the gradient of marginalize_selection needs); records then carry 2M as exponent
  int nref_row0;           // first row of tile_nref this launch reads and writes (0: single evaluation, 1: batched launch)
  int pbatch_pts;          // scan_pbatch_kernel: > 0: rows mode, hyper-parameter points per grid row (<= kPbatchMaxPts); < 0: balanced mode, minus the number of scan workgroups
  const ThetaBlock* tblocks;  // batched launches only: [gridDim.y]
  double derived[GWI_MAX_TERMS][kMaxDerived];
  double theta[GWI_MAX_THETA];
};
static_assert(sizeof(KArgs) <= 4096, "kernel argument block must fit the 4 KiB kernarg segment");

#define GWI_NEG_INF (-__builtin_huge_val())
#define GWI_POS_INF (__builtin_huge_val())

// ---- wave-level reductions (64 lanes) on the DPP path ----------------------------------------------
// Row-shift scan inside each row of 16 lanes, then row_bcast:15 / row_bcast:31 fold the four rows;
// the total lands in lane 63 and is broadcast through an SGPR pair (v_readlane), so the result is
// wave-uniform.  Only lane 63 is ever read: lanes whose DPP source is out of range receive 0
// (bound_ctrl) and rows masked off by row_mask keep stale register contents, but the scan
// structure never routes those lanes into lane 63, so no identity preload is needed.
// No LDS traffic and no ds_bpermute latency chain (12 DPP moves + 6 VALU per fp64 value).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_take(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane63(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
// a value every lane holds identically -> scalar registers (v_readfirstlane): comparisons on it become scalar branches
__device__ __forceinline__ double uniform(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_take<0x111, 0xf>(v);  // row_shr:1
  v += dpp_take<0x112, 0xf>(v);  // row_shr:2
  v += dpp_take<0x114, 0xf>(v);  // row_shr:4
  v += dpp_take<0x118, 0xf>(v);  // row_shr:8
  v += dpp_take<0x142, 0xa>(v);  // row_bcast:15 into rows 1,3
  v += dpp_take<0x143, 0xc>(v);  // row_bcast:31 into rows 2,3
  return lane63(v);
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_take<0x111, 0xf>(v));
  v = fmax(v, dpp_take<0x112, 0xf>(v));
  v = fmax(v, dpp_take<0x114, 0xf>(v));
  v = fmax(v, dpp_take<0x118, 0xf>(v));
  v = fmax(v, dpp_take<0x142, 0xa>(v));
  v = fmax(v, dpp_take<0x143, 0xc>(v));
  return lane63(v);
}

// The wave's maximum to single precision -- for a REFERENCE exponent, which only has to lie within a few units of the true
// maximum (weights are exp(l - m); m off by 6e-8 |m| moves nothing but the split between the record's exponent and its sums).
// A double goes through the DPP network as two 32-bit moves + a canonicalising v_max_f64 + the v_max_f64 itself per step (26
// vector instructions for the six steps of wave_max); here the value becomes an ordered 32-bit key (the float's bits, the
// lower 31 flipped for negative values: signed-integer order = float order, -inf the smallest) and every step is ONE
// v_max_i32 with a DPP operand: 11 vector instructions, the key's way back on the scalar unit.  NaN never gets here.
__device__ __forceinline__ double wave_max_coarse(double v) {
  const int bits = __float_as_int((float)v);
  int k = bits ^ ((bits >> 31) & 0x7fffffff);
  // v_max_i32 with the DPP operand in place (the compiler makes a copy + a DPP move + the maximum of every step).  A lane without
  // a source, or in a row outside the mask, is not written and keeps its own key.  The s_nop are the two wait states a DPP read
  // needs after a vector write of the same register (the hazard recogniser does not look inside an asm block).
  asm volatile(
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(k));
  const int top = __builtin_amdgcn_readlane(k, 63);
  return (double)__int_as_float(top ^ ((top >> 31) & 0x7fffffff));
}

// Eight sums over the wavefront at once: a halving butterfly.  v_permlane32_swap / v_permlane16_swap (gfx950) exchange the
// halves / the odd and even rows of TWO registers in one instruction, so a step that adds lanes l and l + 32 (l and l + 16)
// of eight (four) values leaves four (two) values per lane and needs no select; one step on lane bit 3 (row_ror:8 + select)
// leaves one, and three DPP steps finish inside the groups of eight lanes.  Every lane of the group 8k .. 8k + 7 returns
// the total of v[k]: 54 vector instructions and six dependent steps for eight values, where eight row-shift reductions
// (wave_sum) are 160 and measured three times as long (tools/microbench/wave_sum8.hip: 544 against 1 620 cycles).  The
// order of the additions is fixed: results are bit-reproducible.
__device__ __forceinline__ void swap_halves(double& a, double& b) {  // a <- [a.lanes 0-31 | b.lanes 0-31], b <- [a.lanes 32-63 | b.lanes 32-63]
  auto r0 = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)r1[0], (int)r0[0]);
  b = __hiloint2double((int)r1[1], (int)r0[1]);
}
__device__ __forceinline__ void swap_rows(double& a, double& b) {  // the same for rows of 16 lanes inside each half
  auto r0 = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  auto r1 = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)r1[0], (int)r0[0]);
  b = __hiloint2double((int)r1[1], (int)r0[1]);
}
__device__ __forceinline__ double wave_sum8(const double* v) {
  double x[4], y[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double p = v[j], q = v[j + 4];
    swap_halves(p, q);
    x[j] = p + q;  // lanes 0-31: v[j] over lanes l, l + 32; lanes 32-63: v[j + 4]
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    double p = x[j], q = x[j + 2];
    swap_rows(p, q);
    y[j] = p + q;  // rows 0 / 1 / 2 / 3: v[j], v[j + 2], v[j + 4], v[j + 6] over four lanes each
  }
  const bool up = (__lane_id() & 8) != 0;  // lanes with bit 3 clear keep y[0] and take y[0] of lane ^ 8; the others y[1]
  const double keep = up ? y[1] : y[0], send = up ? y[0] : y[1];
  double z = keep + dpp_take<0x128, 0xf>(send);  // row_ror:8
  z += dpp_take<0x141, 0xf>(z);                  // row_half_mirror: lane i <-> 7 - i of its group of eight
  z += dpp_take<0xB1, 0xf>(z);                   // quad_perm [1,0,3,2]
  z += dpp_take<0x4E, 0xf>(z);                   // quad_perm [2,3,0,1]
  return z;
}

// Below this a mixture density counts as zero for its gradient states: fast_rcp returns NaN for subnormal arguments
// (seed = inf, Newton step inf - inf), and a mixture term that has absorbed the closing exponential (Absorbs<K>) sees
// densities times e^{l - m}, i.e. down to the bottom of the range for samples ~700 e-folds under the tile's best one
// (a narrow peak: the 2000-point fuzz sweep found sigma = 0.33).  Such a sample's weight is < 1e-290 of the best one's.
constexpr double kRcpFloor = 1e-290;
// 1/x to ~1 ulp for normal x: hardware seed + two Newton steps (no IEEE special-case handling:
// callers only need it where the weight is non-zero)
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = fma(-x, r, 1.0);
  r = fma(r, e, r);
  e = fma(-x, r, 1.0);
  return fma(r, e, r);
}

// ---- exp / expm1 for the scan loop ---------------------------------------------------------
// exp(x) = 2^n (1 + r q(r)),  n = rint(x log2 e),  r = x - n ln2 (two-piece ln2),  |r| <= 0.3466,
// q = Taylor series of (e^r - 1)/r to r^10 (truncation 0.3466^12/12! = 6e-15 relative at the ends of the range, 1e-16 in
// its middle half: five orders below what the parity bars of this path can see; two Horner steps fewer).  Every Horner
// step is ONE v_fma_f64 whose addend is a scalar-register constant: left to itself the compiler parks
// the constants in vector registers and spends a v_mov_b64 + v_fmac_f64 per step (two-address form).
// The argument is clamped to [-1000, 710]: exp(-inf) = 0 and exp(>709.8) = +inf fall out of ldexp.
// v_max/v_min return the non-NaN operand, so a NaN argument gives 0: NaN can only enter through a
// non-finite hyper-parameter, which the host rejects before launching (theta_finite in the engine);
// data NaNs sit in samples already excluded by kappa = -inf.  expm1 shares the core: 2^n (1 + rq) - 1 = fma(2^n, rq, 2^n - 1), exact for n = 0.
// minimax coefficients of (e^r - 1)/r on |r| <= 0.3466, degree 8 (tools/exp_poly.py: weighted for the relative error of e^r)
#define GWI_EXP_C0 9.99999999999913181e-01
#define GWI_EXP_C1 4.99999999996100564e-01
#define GWI_EXP_C2 1.66666666677465963e-01
#define GWI_EXP_C3 4.16666669853561281e-02
#define GWI_EXP_C4 8.33333300756820827e-03
#define GWI_EXP_C5 1.38888081598322684e-03
#define GWI_EXP_C6 1.98416032733609286e-04
#define GWI_EXP_C7 2.48820124271676425e-05
#define GWI_EXP_C8 2.74767496787804738e-06
__device__ __forceinline__ double fma_sc(double a, double b, double c_uniform) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_uniform));
  return r;
}
// Round 6 (GWI_EXP_TAYLOR=0, the default): three vector instructions fewer per exponential, where the parity budget has room --
//   * q = an eighth-degree MINIMAX polynomial for (e^r - 1)/r on |r| <= 0.3466 instead of the tenth-degree Taylor series: the
//     relative error of e^r = 1 + r q(r) is <= 1.6e-14 over the whole interval (tools/exp_poly.py; the Taylor series: 6e-15 at its
//     ends), eight FMAs instead of ten;
//   * r = x - n ln2 in ONE fused multiply-add with ln2 rounded to double (off by 2.3e-17): r is then off by 2.3e-17 |n|, i.e. the
//     result by 2e-14 relative at |x| = 700 and 1e-13 at the |l| of a few thousand that spline models under the reference's priors
//     reach -- against a bar of 1e-10 on per-sample log-weights and 1e-9 on log_l (tests: every golden and oracle comparison at
//     its old tolerance).  The shift of fast_exp_shift stays exact: it is applied to the exponent field.
#ifndef GWI_EXP_TAYLOR
#define GWI_EXP_TAYLOR 0
#endif
struct ExpParts {
  double rq;  // e^r - 1
  int n;
};
__device__ __forceinline__ double exp_reduce(double x, double nf) {
#if GWI_EXP_TAYLOR
  const double r = fma(nf, -6.93147180369123816490e-01, x);
  return fma(nf, -1.90821492927058770002e-10, r);
#else
  return fma(nf, -6.93147180559945286227e-01, x);
#endif
}
// (e^r - 1)/r, |r| <= 0.3466
__device__ __forceinline__ double exp_q(double r) {
#if GWI_EXP_TAYLOR
  double q = 1.0 / 39916800.0;            // 1/11!
  q = fma_sc(q, r, 1.0 / 3628800.0);
  q = fma_sc(q, r, 1.0 / 362880.0);
  q = fma_sc(q, r, 1.0 / 40320.0);
  q = fma_sc(q, r, 1.0 / 5040.0);
  q = fma_sc(q, r, 1.0 / 720.0);
  q = fma_sc(q, r, 1.0 / 120.0);
  q = fma_sc(q, r, 1.0 / 24.0);
  q = fma_sc(q, r, 1.0 / 6.0);
  q = fma(q, r, 0.5);
  return fma(q, r, 1.0);
#else
  double q = GWI_EXP_C8;
  q = fma_sc(q, r, GWI_EXP_C7);
  q = fma_sc(q, r, GWI_EXP_C6);
  q = fma_sc(q, r, GWI_EXP_C5);
  q = fma_sc(q, r, GWI_EXP_C4);
  q = fma_sc(q, r, GWI_EXP_C3);
  q = fma_sc(q, r, GWI_EXP_C2);
  q = fma_sc(q, r, GWI_EXP_C1);
  return fma_sc(q, r, GWI_EXP_C0);
#endif
}
__device__ __forceinline__ ExpParts exp_parts(double x) {
  x = fmin(fmax(x, -1000.0), 710.0);
  const double nf = __builtin_rint(x * 1.4426950408889634);
  const double r = exp_reduce(x, nf);
  ExpParts p;
  p.rq = exp_q(r) * r;
  p.n = (int)nf;
  return p;
}
__device__ __forceinline__ double fast_exp(double x) {
  const ExpParts p = exp_parts(x);
  return ldexp(p.rq + 1.0, p.n);
}
// exp(x) 2^-shift with the shift applied to the EXPONENT FIELD (v_ldexp): the result is exp(x) scaled by an exact power of
// two, so sums of such weights scale exactly with the shift and every quantity derived from them is, to the bit,
// independent of which shift was used (as long as nothing leaves the normal range -- the scan checks that).  That is what
// lets spline models weigh a tile against a reference exponent remembered from the previous evaluation (scan_kernel)
// without the results depending on the evaluation history.  The argument range is the full range of log-weights:
// |x| <= 7e5 keeps n ln2_hi exact (ln2_hi has 21 trailing zero bits, |n| < 2^20).
constexpr double kLog2e = 1.4426950408889634;
constexpr double kLn2 = 0.6931471805599453;
constexpr int kNoRef = -2147483647 - 1;  // tile_nref: no reference yet / tile without a live sample
__device__ __forceinline__ double fast_exp_shift(double x, int shift) {
  x = fmin(fmax(x, -7.0e5), 7.0e5);
  const double nf = __builtin_rint(x * kLog2e);
  const double r = exp_reduce(x, nf);
  return ldexp(fma(exp_q(r), r, 1.0), (int)nf - shift);
}
__device__ __forceinline__ double fast_expm1(double x) {
  const ExpParts p = exp_parts(x);
  const double s = ldexp(1.0, p.n);
  return fma(s, p.rq, s - 1.0);
}

// ---- uniform cubic B-spline: 4 taps from the fractional knot coordinate --------------------
// Knots are uniform (interpolation.py:98-106), so with u = (x - lo) / dx the only non-zero bases
// at x are B_k..B_{k+3}, k = floor(u), and their values depend on t = u - k alone.  x == hi is
// assigned to the last interval with t = 1, which reproduces the (1/6, 2/3, 1/6) taps the
// reference's half-open order-1 pieces give there.
struct Taps {
  double b0, b1, b2, b3;
};
// b0 = v^3/6, b3 = t^3/6, b1 = 2/3 - t^2 + t^3/2, b2 = b1(1 - t)  (v = 1 - t): 6 multiplies, 4 FMAs, 1 add
__device__ __forceinline__ Taps cubic_taps(double t) {
  const double v = 1.0 - t;
  const double t2 = t * t, v2 = v * v;
  Taps r;
  r.b0 = v2 * (v * (1.0 / 6.0));
  r.b1 = fma(t2, fma(t, 0.5, -1.0), 2.0 / 3.0);
  r.b2 = fma(v2, fma(v, 0.5, -1.0), 2.0 / 3.0);
  r.b3 = t2 * (t * (1.0 / 6.0));
  return r;
}
// the same scaled by a weight w, for the gradient numerators (w folded into the 1/6 factors).  The fraction is laundered
// through an empty asm so that the compiler cannot share sub-expressions with the cubic_taps() of the evaluation: shared,
// they stay live across the sample's exponential -- 8-10 VGPRs per spline term, config 5: 108 -> 159, a resident wave per
// SIMD -- for 7 of ~50 vector instructions per term (GWI_KEEP_TAPS=1 builds the other choice, for A/B timing)
__device__ __forceinline__ Taps cubic_taps_weighted(double t, double w) {
#ifndef GWI_KEEP_TAPS
  asm("" : "+v"(t));
#endif
  const double v = 1.0 - t;
  const double t2 = t * t, v2 = v * v;
  const double w6 = w * (1.0 / 6.0);
  Taps r;
  r.b0 = v2 * (v * w6);
  r.b1 = w * fma(t2, fma(t, 0.5, -1.0), 2.0 / 3.0);
  r.b2 = w * fma(v2, fma(v, 0.5, -1.0), 2.0 / 3.0);
  r.b3 = t2 * (t * w6);
  return r;
}
// The same with w/6 and 2w/3 handed in -- they are the SAME for every spline term of a sample, but each term's block sits
// behind its own exec-mask branch, across which the compiler does not share them -- and the inner taps formed from the
// outer ones, b1 = 2/3 - t^2 + t^3/2 = 2/3 - t^2 + 3 b3:  w b1 = fma(3, w b3, fma(-w, t^2, 2w/3)): 11 vector instructions per
// term instead of 14 + the w/6 multiply.  (Cancellation is benign: the three addends are O(w), the result >= w/6.)
struct Weight {
  double w, w6, w23;
};
__device__ __forceinline__ Weight make_weight(double w) {
  Weight r;
  r.w = w;
  r.w6 = w * (1.0 / 6.0);
  r.w23 = w * (2.0 / 3.0);
  return r;
}
__device__ __forceinline__ Taps cubic_taps_weighted(double t, const Weight& ww) {
#ifdef GWI_AB_OLD_TAPS
  return cubic_taps_weighted(t, ww.w);
#else
#ifndef GWI_KEEP_TAPS
  asm("" : "+v"(t));
#endif
  const double v = 1.0 - t;
  const double t2 = t * t, v2 = v * v;
  Taps r;
  r.b3 = (ww.w6 * t) * t2;
  r.b0 = (ww.w6 * v) * v2;
  r.b1 = fma(3.0, r.b3, fma(-ww.w, t2, ww.w23));
  r.b2 = fma(3.0, r.b0, fma(-ww.w, v2, ww.w23));
  return r;
#endif
}
// The spline VALUE in the local power basis: on knot interval k the uniform cubic B-spline sum is one cubic in t,
//   sum_i c_{k+i} b_i(t) = A + B t + C t^2 + D t^3,   A = (c0 + 4 c1 + c2)/6, B = (c2 - c0)/2, C = (c0 - 2 c1 + c2)/2, D = (c3 - c0)/6 + (c1 - c2)/2,
// tabulated per workgroup when theta is staged: four arrays A[], B[], C[], D[] indexed like the coefficients (position
// th0 + k), kPolyStride doubles apart -- each array is read exactly as the coefficient array was (neighbouring lanes on the
// same or neighbouring 8-byte words: broadcast, no bank conflicts; interleaved 32-byte rows per interval measured 7 % SLOWER
// than the taps, profiles/round3/EXPERIMENTS.md section 8), and the constant stride lets one ds_read2st64_b64 fetch two of them.
// A sample's value is then 3 FMAs instead of the four taps + dot product (15 vector instructions); the taps are only formed
// once per sample, weighted, for the gradient rows.
// The stride is deliberately NOT a multiple of 64 doubles: with 256 the compiler fuses the four reads of a sample into two
// ds_read2st64_b64, which the LDS serves as two 4 x 16-lane accesses each (8 LDS cycles per instruction, 16 per term);
// 257 keeps them four ds_read_b64 at immediate offsets (2 x 32 lanes, ~2.3 cycles each: 9 per term, conflict-free for any
// mix of knot intervals -- neighbouring intervals are neighbouring 8-byte words).  The LDS pipe is what config 5 waits for
// (SQ_WAIT_INST_LDS 24 % of its wave cycles in round 3): profiles/round4/EXPERIMENTS.md section 2.
#ifdef GWI_AB_POLY_STRIDE_256
constexpr int kPolyStride = GWI_MAX_THETA;
#else
constexpr int kPolyStride = GWI_MAX_THETA + 1;
#endif
__device__ __forceinline__ void spline_poly(double c0, double c1, double c2, double c3, double* out) {
  out[0] = (c0 + 4.0 * c1 + c2) * (1.0 / 6.0);
  out[kPolyStride] = (c2 - c0) * 0.5;
  out[2 * kPolyStride] = (c0 - 2.0 * c1 + c2) * 0.5;
  out[3 * kPolyStride] = (c3 - c0) * (1.0 / 6.0) + (c1 - c2) * 0.5;
}
// The columns of EXP_SPLINE / LINEAR_SPLINE terms hold the KNOT COORDINATE u = (x - lo) / dx of the sample, not x: the engine
// converts them once when the catalog becomes resident (gwi_engine.hip: spline_knot_kernel), so the scan never spends the
// subtract and multiply -- nor the scalar registers of lo and 1/dx -- per sample and term.
//   * exponentiated splines without the zero-outside flag (the LogY bases of every mass / ratio / spin model): the
//     conversion also clamps u into [0, n_int): interval and fraction are two instructions, (int)u and fract(u).  (x = hi
//     lands one ulp below n_int: t = 1 - 2^-52 n_int instead of 1, taps within 1e-15 of the reference's (1/6, 2/3, 1/6).)
//   * zero-outside and linear splines keep the unclamped coordinate (samples outside the domain are alive there): clamp here.
__device__ __forceinline__ void spline_locate_knot(double u, int& k, double& t) {
  k = (int)u;
  t = __builtin_amdgcn_fract(u);
}
__device__ __forceinline__ void spline_locate_term(double u, const TermD& td, int& k, double& t) {
  const int last = td.n_basis - 4;
  int kk = (int)u;  // truncation == floor wherever the clamp below does not decide anyway (u < 0 -> 0; overflow saturates)
  kk = max(0, min(kk, last));
  k = kk;
  t = u - (double)kk;
}
__device__ __forceinline__ bool spline_outside(double u, const TermD& td) { return !((u >= 0.0) && (u <= td.p3)); }
// the same from a spline coordinate x (grid nodes of the LERP term, computed in the kernel)
__device__ __forceinline__ void spline_locate_x(double x, const TermD& td, int& k, double& t) {
  spline_locate_term((x - td.p0) * td.p2, td, k, t);
}
__device__ __forceinline__ void spline_locate(double x, double lo, double inv_dx, int n_basis, int& k, double& t) {
  const double u = (x - lo) * inv_dx;
  const int last = n_basis - 4;  // index of the last interval
  int kk = (int)u;  // truncation == floor wherever the clamp below does not decide anyway (u < 0 -> 0; NaN -> 0; overflow saturates)
  kk = max(0, min(kk, last));
  k = kk;
  t = u - (double)kk;
}

// Column loads: the column pointers come out of a pointer table, so the compiler cannot infer their
// address space and would emit flat_load; they are HBM (global) pointers.
typedef const double __attribute__((address_space(1))) * gptr_t;
__device__ __forceinline__ double gload(const double* p, long long idx) { return ((gptr_t)p)[idx]; }
// A sample inside a workgroup's tile: `origin` (elements) is workgroup-uniform, `boff` the lane's BYTE offset from it.  The
// load then takes its base from a scalar register pair and a 32-bit vector offset (global_load ... v_off, s[base:base+1]):
// no 64-bit address arithmetic per column and lane (it was 22 v_lshl_add_u64 per trip at config 2).
struct SIdx {
  long long origin;
  unsigned boff;
};
typedef const char __attribute__((address_space(1))) * gbytes_t;
__device__ __forceinline__ double gload(const double* p, SIdx i) { return *(gptr_t)((gbytes_t)(p + i.origin) + i.boff); }

// ---- evaluation context --------------------------------------------------------------------
struct Ctx {
  const KArgs* a;               // kernel arguments (term descriptors, sizes, column pointers)
  const double* theta;          // scalar hyper-parameters: uniform index -> scalar loads, never LDS
  const double (*derived)[kMaxDerived];  // host-precomputed theta-only scalars per term
  const double* coefs;          // LDS copy of theta for the lane-varying spline coefficient reads
  const double* poly;           // LDS: the cubic of every knot interval in the local power basis, four arrays kPolyStride apart (spline_poly)
  double* gacc;                 // LDS gradient numerators [n_theta][rep] + this lane's replica: coefficient p lives at gacc[p << rep_shift]
  int rep_shift;
  const double* const (*tcols)[2];  // per-term column pointers of the sample set this workgroup scans
  mutable Weight wt;                // the weight of the sample being accumulated with w/6 and 2w/3 (set by the scan loop: shared by its spline terms)
#ifdef GWI_ABL_SCATTER_TO_REG
  mutable double sink = 0.0;  // timing-only ablation: the weighted taps end up here instead of in the LDS rows
#endif
};

__device__ __forceinline__ double spline_value(const Ctx& c, int first, double t) {
#ifdef GWI_SPLINE_TAPS_VALUE  // A/B: the four-tap form
  const double* cf = c.coefs + first;
  const Taps b = cubic_taps(t);
  return cf[0] * b.b0 + cf[1] * b.b1 + cf[2] * b.b2 + cf[3] * b.b3;
#else
  const double* q = c.poly + first;
  return fma(fma(fma(q[3 * kPolyStride], t, q[2 * kPolyStride]), t, q[kPolyStride]), t, q[0]);
#endif
}

// ---- spline-coefficient gradient numerators ------------------------------------------------------
// G_p += w B_p(x) for the four non-zero bases of a sample: ds_add_f64 into the workgroup's rows in LDS.  Layout
// [coefficient][replica], replica = lane & (rep - 1): the LDS bank of an address then depends on the LANE alone
// (8-byte words: bank pair = replica mod 32), never on the knot interval the sample falls in.  With rep = 64 every lane
// owns its replica: no two lanes of a wave instruction share an address or -- beyond the two passes a 64 x 8-byte access
// takes anyway -- a bank, whatever the data looks like (posterior samples of one event cluster in a few knot intervals:
// per-wave rows [replica][coefficient] with 8 replicas serialised up to 8 lanes per address and collided on banks at
// random; config 5 spent 30 % of its scan there).  The four waves of the workgroup share the rows (the adds are atomic),
// which is what makes 64 replicas fit: n_theta x 512 B per WORKGROUP.
__device__ __forceinline__ void spline_scatter(const Ctx& c, int first, const Taps& b) {
#ifdef GWI_ABL_NO_SCATTER  // timing-only ablation build (tools/build_ablations.sh): results are wrong by construction
  return;
#endif
#ifdef GWI_ABL_SCATTER_TO_REG  // ... the same with the weighted taps still computed (isolates the cost of the LDS atomics)
  c.sink += (b.b0 + b.b1) + (b.b2 + b.b3) + (double)first;
  return;
#endif
#ifdef GWI_ABL_ALL_REP64  // timing-only: every lane an address of its own (64 "replicas" that overlap the neighbouring rows): no two lanes of a wave instruction ever meet
  double* g = c.gacc - (__lane_id() & 15) + __lane_id() + (first << c.rep_shift);
#else
  double* g = c.gacc + (first << c.rep_shift);
#endif
  const int step = 1 << c.rep_shift;
  unsafeAtomicAdd(g, b.b0);
  unsafeAtomicAdd(g + step, b.b1);
  unsafeAtomicAdd(g + 2 * step, b.b2);
  unsafeAtomicAdd(g + 3 * step, b.b3);
}
#ifdef GWI_ABL_Z_REP64  // timing-only: the same for zero-outside terms alone (the redshift spline of configs 5 / bspline_test)
__device__ __forceinline__ void spline_scatter_rep64(const Ctx& c, int first, const Taps& b) {
  double* g = c.gacc - (__lane_id() & 15) + __lane_id() + (first << c.rep_shift);
  const int step = 1 << c.rep_shift;
  unsafeAtomicAdd(g, b.b0);
  unsafeAtomicAdd(g + step, b.b1);
  unsafeAtomicAdd(g + 2 * step, b.b2);
  unsafeAtomicAdd(g + 3 * step, b.b3);
}
#endif
// ---- term library --------------------------------------------------------------------------
// A sample's weight is  w = L * exp(l - m):  each term either adds to the log part l (power laws,
// splines: already exponents) or multiplies the linear part L (mixtures, the ratio normaliser:
// sums of exponentials whose log would cost a transcendental for nothing) -- the reference
// multiplies densities in the linear domain throughout.  Sample-independent normalisers are left
// out (added on the host: they cancel in log_l and in its gradient).  State caches dl/dtheta; once
// w is known accumulate() adds w * dl/dtheta into registers (scalars) or the wave's LDS row
// (spline coefficients).
template <int K>
struct Term;
// Terms whose linear factor is a SUM of exponentials (the PL+Peak mixtures) can take the sample's closing factor
// exp(l - m) into their own exponents: (1-lam) e^{a1 + E} + lam e^{a2 + E} instead of ((1-lam) e^{a1} + lam e^{a2}) e^{E},
// one exp per sample less (a quarter of config 2's).  The chain evaluates the first such term LAST (Chain::finish), once
// E = l - m is known; its gradient states are ratios of the shifted parts, which the common factor leaves alone.
template <int K>
struct Absorbs {
  static constexpr bool value = false;
};

#define GWI_ACC1(member)                                                                    \
  __device__ static void init(Acc& a) { a.member = 0; }                                     \
  __device__ static void rescale(Acc& a, double sc) { a.member *= sc; }

// x^alpha on fixed [lo,hi] (distributions.py:100-119)
template <>
struct Term<GWI_TERM_POWERLAW> {
  static constexpr bool kSpline = false;
  struct In {
    double x0;
  };
  __device__ static void load(const double* const* tc, SIdx idx, In& in) { in.x0 = gload(tc[0], idx); }
  struct State {
    double lx;
  };
  struct Acc {
    double g0;
  };
  __device__ static double eval(const TermD& t, const double*, const Ctx& c, const In& in, State& s, double&) {
    s.lx = in.x0;
    return c.theta[t.th0] * s.lx;
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) { a.g0 += w * s.lx; }
  GWI_ACC1(g0)
  static constexpr int kNumAcc = 1;
  __device__ static void collect(const TermD& t, const Acc& a, double* vals, int* th) {
    vals[0] = a.g0;
    th[0] = t.th0;
  }
};

// (1-lam) A x^alpha + lam Cn exp(-(x-mu)^2/(2 sig^2))  (parametric.py:49-53)
// derived: d0=log A, d1=dlogA/dalpha, d2=log Cn, d3=dlogCn/dmu, d4=dlogCn/dsig, d5=1/sig^2, d6=1/sig^3
template <>
struct Term<GWI_TERM_PLPEAK> {
  static constexpr bool kSpline = false;
  struct In {
    double x1;  // log x: the one column of the term; x for the Gaussian component is exp(log x), formed in eval_shifted()
  };
  __device__ static void load(const double* const* tc, SIdx idx, In& in) { in.x1 = gload(tc[0], idx); }
  struct State {
    double da, dmu, dsg, dlam;
  };
  struct Acc {
    double g[4];
  };
  __device__ static double eval(const TermD& t, const double* d, const Ctx& c, const In& in, State& s, double& lin) {
    return eval_shifted<false>(t, d, c, in, s, lin, 0.0, 0);
  }
  // SHIFT: the sample's closing factor folded in: both exponents take E, and the result is scaled by 2^-nshift (fast_exp_shift)
  template <bool SHIFT>
  __device__ static double eval_shifted(const TermD& t, const double* d, const Ctx& c, const In& in, State& s, double& lin, double E, int nshift) {
    // x = exp(log x): round 2 streamed x as a second column (config 2: 40 B per sample against 32 algorithmic, FETCH_SIZE
    // 1.29 x the algorithmic bytes); the exponential costs the config-2 scan nothing measurable (7.88-8.05 vs 7.87-8.01 us
    // over four interleaved rounds) and moves the peak's exponent by <= 1e-12
    const double lx = in.x1;
    const double x = fast_exp(lx);
    const double alpha = c.theta[t.th0], mu = c.theta[t.th1], lam = c.theta[t.th3];
    const double dx = x - mu;
    const double dx2 = dx * dx;
    const double e_pl = SHIFT ? fast_exp_shift(fma(alpha, lx, d[0] + E), nshift) : fast_exp(alpha * lx + d[0]);
    const double e_tn = SHIFT ? fast_exp_shift(fma(-0.5 * dx2, d[5], d[2] + E), nshift) : fast_exp(-0.5 * dx2 * d[5] + d[2]);
    const double P = (1.0 - lam) * e_pl, T = lam * e_tn;
    const double p = P + T;
    const double ip = (p > kRcpFloor) ? fast_rcp(p) : 0.0;  // p == 0 (or next to it, see kRcpFloor): a dead sample must not carry NaN into the sums
    s.da = P * (lx + d[1]) * ip;
    s.dmu = T * (dx * d[5] + d[3]) * ip;
    s.dsg = T * (dx2 * d[6] + d[4]) * ip;
    s.dlam = (e_tn - e_pl) * ip;
    lin *= p;
    return 0.0;
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) {
    a.g[0] += w * s.da;
    a.g[1] += w * s.dmu;
    a.g[2] += w * s.dsg;
    a.g[3] += w * s.dlam;
  }
  // The chain's absorbing term, evaluated last (Chain::finish): L = the product of the sample's other linear factors (0 for a
  // dead sample), E = its closing exponent l - m.  Returns the sample's WEIGHT w = L p e^E 2^-nshift and leaves the states
  // as w dl/dtheta already: w (P/p)(...) = (L P)(...) -- the division by the mixture density p cancels against the weight, so
  // this form needs no reciprocal at all (the ratio form above: v_rcp_f64 + two Newton steps + the guard against p = 0, and a
  // multiplication by 1/p per state; 13 of config 2's ~170 vector instructions per sample).  Dead samples arrive with L = 0
  // and E = -inf: both exponentials are 0, every state an exact 0 (columns are finite for every sample: the binder parks a
  // finite value where the data are not, engine.py bind()).
  __device__ static double finish(const TermD& t, const double* d, const Ctx& c, const In& in, State& s, double L, double E, int nshift) {
    const double lx = in.x1;
    const double x = fast_exp(lx);
    const double alpha = c.theta[t.th0], mu = c.theta[t.th1], lam = c.theta[t.th3];
    const double dx = x - mu;
    const double dx2 = dx * dx;
    const double e_pl = fast_exp_shift(fma(alpha, lx, d[0] + E), nshift);
    const double e_tn = fast_exp_shift(fma(-0.5 * dx2, d[5], d[2] + E), nshift);
    const double LP = (L * (1.0 - lam)) * e_pl, LT = (L * lam) * e_tn;
    s.da = LP * (lx + d[1]);
    s.dmu = LT * fma(dx, d[5], d[3]);
    s.dsg = LT * fma(dx2, d[6], d[4]);
    s.dlam = L * (e_tn - e_pl);
    return LP + LT;
  }
  // ... and their accumulation: wfac = 1, or the weight itself in a squared-weight pass (w^2 dl = w x (w dl))
  __device__ static void accumulate_weighted(double wfac, const State& s, Acc& a) {
    a.g[0] = fma(wfac, s.da, a.g[0]);
    a.g[1] = fma(wfac, s.dmu, a.g[1]);
    a.g[2] = fma(wfac, s.dsg, a.g[2]);
    a.g[3] = fma(wfac, s.dlam, a.g[3]);
  }
  __device__ static void init(Acc& a) { a.g[0] = a.g[1] = a.g[2] = a.g[3] = 0; }
  __device__ static void rescale(Acc& a, double sc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a.g[j] *= sc;
  }
  static constexpr int kNumAcc = 4;
  __device__ static void collect(const TermD& t, const Acc& a, double* vals, int* th) {
#pragma unroll
    for (int j = 0; j < 4; ++j) vals[j] = a.g[j];
    th[0] = t.th0;
    th[1] = t.th1;
    th[2] = t.th2;
    th[3] = t.th3;
  }
};

// q^beta (1+beta)/(1 - r^(1+beta)), r = mmin/m1  (distributions.py:111-116 with low = mmin/m1)
// derived: d0 = 1/(1+beta)
template <>
struct Term<GWI_TERM_POWERLAW_RATIO> {
  static constexpr bool kSpline = false;
  struct In {
    double x0, x1;
  };
  __device__ static void load(const double* const* tc, SIdx idx, In& in) {
    in.x0 = gload(tc[0], idx);
    in.x1 = gload(tc[1], idx);
  }
  struct State {
    double db;
  };
  struct Acc {
    double g0;
  };
  __device__ static double eval(const TermD& t, const double* d, const Ctx& c, const In& in, State& s, double& lin) {
    const double lq = in.x0;
    // log m1: its own column, or the knot coordinate u of the model's m1 spline (log m1 = lo + u dx: one FMA instead of an
    // 8-byte column per sample; include/gwi_engine.h GWI_RATIO_LOGM_FROM_SPLINE).  The flag is wave-uniform.
    const double lm = (t.flags & GWI_RATIO_LOGM_FROM_SPLINE) ? fma(in.x1, t.p3, t.p1) : in.x1;
    const double lr = t.p0 - lm;  // log(mmin/m1) <= 0 for every non-excluded sample
    const double beta = c.theta[t.th0];
    const double b1 = 1.0 + beta;
    if (b1 == 0.0) {  // alpha == -1 branch of the reference: 1/log(high/low)
      s.db = lq - 0.5 * lr;
      lin *= -fast_rcp(lr);
      return -lq;
    }
    const double em1 = fast_expm1(b1 * lr);   // r^(1+beta) - 1
    // m1 == mmin exactly (lr = 0): the q interval is empty, the reference's weight is NaN -> 0; keep the
    // gradient state finite so that the dead sample adds 0, not NaN
    const double inv_den = (em1 != 0.0) ? -fast_rcp(em1) : 0.0;  // 1/(1 - r^(1+beta))
    s.db = lq + d[0] + (em1 + 1.0) * lr * inv_den;
    lin *= b1 * inv_den;
    return beta * lq;
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) { a.g0 += w * s.db; }
  GWI_ACC1(g0)
  static constexpr int kNumAcc = 1;
  __device__ static void collect(const TermD& t, const Acc& a, double* vals, int* th) {
    vals[0] = a.g0;
    th[0] = t.th0;
  }
};

// Beta(a; alpha, beta): (alpha-1) log a + (beta-1) log(1-a) - betaln  (distributions.py:160-161)
template <>
struct Term<GWI_TERM_BETA> {
  static constexpr bool kSpline = false;
  struct In {
    double x0, x1;
  };
  __device__ static void load(const double* const* tc, SIdx idx, In& in) {
    in.x0 = gload(tc[0], idx);
    in.x1 = gload(tc[1], idx);
  }
  struct State {
    double la, l1;
  };
  struct Acc {
    double g[2];
  };
  __device__ static double eval(const TermD& t, const double*, const Ctx& c, const In& in, State& s, double&) {
    s.la = in.x0;
    s.l1 = in.x1;
    return (c.theta[t.th0] - 1.0) * s.la + (c.theta[t.th1] - 1.0) * s.l1;
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) {
    a.g[0] += w * s.la;
    a.g[1] += w * s.l1;
  }
  __device__ static void init(Acc& a) { a.g[0] = a.g[1] = 0; }
  __device__ static void rescale(Acc& a, double sc) {
    a.g[0] *= sc;
    a.g[1] *= sc;
  }
  static constexpr int kNumAcc = 2;
  __device__ static void collect(const TermD& t, const Acc& a, double* vals, int* th) {
    vals[0] = a.g[0];
    vals[1] = a.g[1];
    th[0] = t.th0;
    th[1] = t.th1;
  }
};

// (1-xi)/2 + xi Cn exp(-(ct-1)^2/(2 sig^2))  (parametric.py:84-86)
// derived: d0=log Cn, d1=dlogCn/dsig, d2=1/sig^2, d3=1/sig^3
template <>
struct Term<GWI_TERM_TILT_MIXTURE> {
  static constexpr bool kSpline = false;
  struct In {
    double x0;
  };
  __device__ static void load(const double* const* tc, SIdx idx, In& in) { in.x0 = gload(tc[0], idx); }
  struct State {
    double dxi, dsg;
  };
  struct Acc {
    double g[2];
  };
  __device__ static double eval(const TermD& t, const double* d, const Ctx& c, const In& in, State& s, double& lin) {
    const double ct = in.x0;
    const double xi = c.theta[t.th0];
    const double dx = ct - 1.0;
    const double dx2 = dx * dx;
    const double e_tn = fast_exp(-0.5 * dx2 * d[2] + d[0]);
    const double p = 0.5 * (1.0 - xi) + xi * e_tn;
    const double ip = (p > kRcpFloor) ? fast_rcp(p) : 0.0;  // p == 0 (or next to it, see kRcpFloor): a dead sample must not carry NaN into the sums
    s.dxi = (e_tn - 0.5) * ip;
    s.dsg = xi * e_tn * (dx2 * d[3] + d[1]) * ip;
    lin *= p;
    return 0.0;
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) {
    a.g[0] += w * s.dxi;
    a.g[1] += w * s.dsg;
  }
  __device__ static void init(Acc& a) { a.g[0] = a.g[1] = 0; }
  __device__ static void rescale(Acc& a, double sc) {
    a.g[0] *= sc;
    a.g[1] *= sc;
  }
  static constexpr int kNumAcc = 2;
  __device__ static void collect(const TermD& t, const Acc& a, double* vals, int* th) {
    vals[0] = a.g[0];
    vals[1] = a.g[1];
    th[0] = t.th0;
    th[1] = t.th1;
  }
};

// TN(x; mu, sig, lo, hi) alone (distributions.py:136-143); derived: d0=1/sig^2, d1=1/sig^3
template <>
struct Term<GWI_TERM_TRUNCNORM> {
  static constexpr bool kSpline = false;
  struct In {
    double x0;
  };
  __device__ static void load(const double* const* tc, SIdx idx, In& in) { in.x0 = gload(tc[0], idx); }
  struct State {
    double dmu, dsg;
  };
  struct Acc {
    double g[2];
  };
  __device__ static double eval(const TermD& t, const double* d, const Ctx& c, const In& in, State& s, double&) {
    const double x = in.x0;
    const double dx = x - c.theta[t.th0];
    const double dx2 = dx * dx;
    s.dmu = dx * d[0];
    s.dsg = dx2 * d[1];
    return -0.5 * dx2 * d[0];
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) {
    a.g[0] += w * s.dmu;
    a.g[1] += w * s.dsg;
  }
  __device__ static void init(Acc& a) { a.g[0] = a.g[1] = 0; }
  __device__ static void rescale(Acc& a, double sc) {
    a.g[0] *= sc;
    a.g[1] *= sc;
  }
  static constexpr int kNumAcc = 2;
  __device__ static void collect(const TermD& t, const Acc& a, double* vals, int* th) {
    vals[0] = a.g[0];
    vals[1] = a.g[1];
    th[0] = t.th0;
    th[1] = t.th1;
  }
};

// (1+z)^(lamb-1) (parametric.py:126-127); dVc/dz is in kappa, the grid normaliser on the host side.
template <>
struct Term<GWI_TERM_POWERLAW_REDSHIFT> {
  static constexpr bool kSpline = false;
  struct In {
    double x0;
  };
  __device__ static void load(const double* const* tc, SIdx idx, In& in) { in.x0 = gload(tc[0], idx); }
  struct State {
    double l1pz;
  };
  struct Acc {
    double g0;
  };
  __device__ static double eval(const TermD& t, const double*, const Ctx& c, const In& in, State& s, double&) {
    s.l1pz = in.x0;
    return (c.theta[t.th0] - 1.0) * s.l1pz;
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) { a.g0 += w * s.l1pz; }
  GWI_ACC1(g0)
  static constexpr int kNumAcc = 1;
  __device__ static void collect(const TermD& t, const Acc& a, double* vals, int* th) {
    vals[0] = a.g0;
    th[0] = t.th0;
  }
};

// exp(sum_k c_k B_k(x))  (interpolation.py:381-394 / :293-304 + exp, spline_perturbation.py:352)
// p0 = lo, p1 = hi, p2 = 1/dx of the spline coordinate; th0 = theta offset of c_0.
template <>
struct Term<GWI_TERM_EXP_SPLINE> {
  static constexpr bool kSpline = true;
  struct In {
    double x0;
  };
  __device__ static void load(const double* const* tc, SIdx idx, In& in) { in.x0 = gload(tc[0], idx); }
  struct State {
    double t;
    int k;  // -1: outside the domain of a zero-outside basis (factor 1, no gradient)
  };
  struct Acc {};
  __device__ static double eval(const TermD& t, const double*, const Ctx& c, const In& in, State& s, double&) {
    const double u = in.x0;  // knot coordinate (see spline_locate_knot)
    int k;
    double tt, v;
    // BSpline / LogXBSpline bases are 0 outside the closed domain (interpolation.py:175); LogY bases exclude the sample
    // (kappa = -inf, decided when the catalog was bound) and their coordinate was clamped into the domain.  The flag is
    // wave-uniform and the two forms sit behind a real scalar branch (the empty asm cannot be speculated, so the compiler does
    // not turn it into selects): terms without the flag -- six of config 5's seven, all five of config 3's -- pay neither
    // the clamps nor the two compares and three selects of the domain check
    if (t.flags & GWI_SPLINE_OUTSIDE_ZERO_EXPONENT) {
      asm volatile("");
      spline_locate_term(u, t, k, tt);
      v = spline_value(c, t.th0 + k, tt);
      const bool outside = spline_outside(u, t);
      k = outside ? -1 : k;
      v = outside ? 0.0 : v;
    } else {
      spline_locate_knot(u, k, tt);
      v = spline_value(c, t.th0 + k, tt);
    }
    s.t = tt;
    s.k = k;
    return v;
  }
  __device__ static void accumulate(const TermD& t, const Ctx& c, double w, const State& s, Acc&) {
    if (s.k >= 0 && w != 0.0) {  // c.wt.w == w
#ifdef GWI_ABL_Z_REP64
      if (t.flags & GWI_SPLINE_OUTSIDE_ZERO_EXPONENT)
        spline_scatter_rep64(c, t.th0 + s.k, cubic_taps_weighted(s.t, c.wt));
      else
#endif
        spline_scatter(c, t.th0 + s.k, cubic_taps_weighted(s.t, c.wt));
    }
  }
  __device__ static void init(Acc&) {}
  __device__ static void rescale(Acc&, double) {}
  static constexpr int kNumAcc = 0;
  __device__ static void collect(const TermD&, const Acc&, double*, int*) {}
};

// sum_k c_k B_k(x): linear-Y B-spline density (interpolation.py:293-304; chi_eff / chi_p models,
// single.py:199-318).  Linear factor; dl/dc_k = B_k / f.  p0 = lo, p1 = hi, p2 = 1/dx.
template <>
struct Term<GWI_TERM_LINEAR_SPLINE> {
  static constexpr bool kSpline = true;
  struct In {
    double x0;
  };
  __device__ static void load(const double* const* tc, SIdx idx, In& in) { in.x0 = gload(tc[0], idx); }
  struct State {
    double t, inv_f;
    int k;
  };
  struct Acc {};
  __device__ static double eval(const TermD& t, const double*, const Ctx& c, const In& in, State& s, double& lin) {
    const double u = in.x0;  // knot coordinate, unclamped (see spline_locate_knot)
    int k;
    double tt;
    spline_locate_term(u, t, k, tt);
    double f = spline_value(c, t.th0 + k, tt);
    if (spline_outside(u, t)) f = 0.0;  // bases are 0 outside the closed domain (:175)
    s.t = tt;
    s.k = k;
    s.inv_f = f > 0.0 ? fast_rcp(f) : 0.0;
    lin *= f;  // f <= 0 makes the sample dead (lin > 0 test in the scan loop)
    return 0.0;
  }
  __device__ static void accumulate(const TermD& t, const Ctx& c, double w, const State& s, Acc&) {
    if (w != 0.0) spline_scatter(c, t.th0 + s.k, cubic_taps_weighted(s.t, make_weight(w * s.inv_f)));
  }
  __device__ static void init(Acc&) {}
  __device__ static void rescale(Acc&, double) {}
  static constexpr int kNumAcc = 0;
  __device__ static void collect(const TermD&, const Acc&, double*, int*) {}
};

// (1-xi)/4 + xi Cn^2 exp(-((ct1-1)^2 + (ct2-1)^2)/(2 sig^2))  (parametric.py:97-102)
// derived: d0 = log Cn, d1 = dlogCn/dsig, d2 = 1/sig^2, d3 = 1/sig^3
template <>
struct Term<GWI_TERM_TILT_JOINT> {
  static constexpr bool kSpline = false;
  struct In {
    double x0, x1;
  };
  __device__ static void load(const double* const* tc, SIdx idx, In& in) {
    in.x0 = gload(tc[0], idx);
    in.x1 = gload(tc[1], idx);
  }
  struct State {
    double dxi, dsg;
  };
  struct Acc {
    double g[2];
  };
  __device__ static double eval(const TermD& t, const double* d, const Ctx& c, const In& in, State& s, double& lin) {
    const double xi = c.theta[t.th0];
    const double d1 = in.x0 - 1.0, d2 = in.x1 - 1.0;
    const double r2 = d1 * d1 + d2 * d2;
    const double A = fast_exp(-0.5 * r2 * d[2] + 2.0 * d[0]);
    const double p = 0.25 * (1.0 - xi) + xi * A;
    const double ip = (p > kRcpFloor) ? fast_rcp(p) : 0.0;  // p == 0 (or next to it, see kRcpFloor): a dead sample must not carry NaN into the sums
    s.dxi = (A - 0.25) * ip;
    s.dsg = xi * A * (r2 * d[3] + 2.0 * d[1]) * ip;
    lin *= p;
    return 0.0;
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) {
    a.g[0] += w * s.dxi;
    a.g[1] += w * s.dsg;
  }
  __device__ static void init(Acc& a) { a.g[0] = a.g[1] = 0; }
  __device__ static void rescale(Acc& a, double sc) {
    a.g[0] *= sc;
    a.g[1] *= sc;
  }
  static constexpr int kNumAcc = 2;
  __device__ static void collect(const TermD& t, const Acc& a, double* vals, int* th) {
    vals[0] = a.g[0];
    vals[1] = a.g[1];
    th[0] = t.th0;
    th[1] = t.th1;
  }
};

// reference `smooth` (distributions.py:16-21), which is 1/(1 + exp(d/y + d/(y-d))) for every y = x - xmin.
// Returns S and d log S / d delta = -(1 - S) (1/y + y/(y-d)^2).
// fast_rcp has no IEEE special cases: 1/(+-0) and 1/inf are patched here, because an overflowing exponent
// (x just above xmin + delta) must give S = 0 exactly -- in the PL+Peak mixture the Gaussian part survives it.
__device__ __forceinline__ double taper(double y, double dl, double& dlog_ddelta) {
  const double yd = y - dl;
  const double iy = (y == 0.0) ? __builtin_copysign(GWI_POS_INF, y) : fast_rcp(y);
  const double iyd = (yd == 0.0) ? __builtin_copysign(GWI_POS_INF, yd) : fast_rcp(yd);
  const double E = fast_exp(dl * iy + dl * iyd);
  const double S = (E < GWI_POS_INF) ? fast_rcp(1.0 + E) : 0.0;
  dlog_ddelta = (S > 0.0) ? -(1.0 - S) * (iy + y * iyd * iyd) : 0.0;  // S == 0: factor (and its derivative) vanish
  return S;
}

template <>
struct Term<GWI_TERM_SMOOTH> {
  static constexpr bool kSpline = false;
  struct In {
    double x0;
  };
  __device__ static void load(const double* const* tc, SIdx idx, In& in) { in.x0 = gload(tc[0], idx); }
  struct State {
    double dd;
  };
  struct Acc {
    double g0;
  };
  __device__ static double eval(const TermD& t, const double*, const Ctx& c, const In& in, State& s, double& lin) {
    lin *= taper(in.x0, c.theta[t.th0], s.dd);
    return 0.0;
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) { a.g0 += w * s.dd; }
  GWI_ACC1(g0)
  static constexpr int kNumAcc = 1;
  __device__ static void collect(const TermD& t, const Acc& a, double* vals, int* th) {
    vals[0] = a.g0;
    th[0] = t.th0;
  }
};

// (1-lam) A x^alpha S(x - lo; delta) + lam Cn exp(-(x-mu)^2/(2 sig^2))  (parametric.py:49-53 with delta)
// derived as PLPEAK; th4 = delta
template <>
struct Term<GWI_TERM_PLPEAK_SMOOTH> {
  static constexpr bool kSpline = false;
  struct In {
    double x0, x1;
  };
  __device__ static void load(const double* const* tc, SIdx idx, In& in) {
    in.x0 = gload(tc[0], idx);
    in.x1 = gload(tc[1], idx);
  }
  struct State {
    double da, dmu, dsg, dlam, ddel;
  };
  struct Acc {
    double g[5];
  };
  __device__ static double eval(const TermD& t, const double* d, const Ctx& c, const In& in, State& s, double& lin) {
    return eval_shifted<false>(t, d, c, in, s, lin, 0.0, 0);
  }
  template <bool SHIFT>
  __device__ static double eval_shifted(const TermD& t, const double* d, const Ctx& c, const In& in, State& s, double& lin, double E, int nshift) {
    const double x = in.x0;
    const double lx = in.x1;
    const double alpha = c.theta[t.th0], mu = c.theta[t.th1], lam = c.theta[t.th3];
    const double dx = x - mu;
    const double dx2 = dx * dx;
    double dlogS;
    const double S = taper(x - t.p0, c.theta[t.th4], dlogS);
    const double e_pl = (SHIFT ? fast_exp_shift(fma(alpha, lx, d[0] + E), nshift) : fast_exp(alpha * lx + d[0])) * S;
    const double e_tn = SHIFT ? fast_exp_shift(fma(-0.5 * dx2, d[5], d[2] + E), nshift) : fast_exp(-0.5 * dx2 * d[5] + d[2]);
    const double P = (1.0 - lam) * e_pl, T = lam * e_tn;
    const double p = P + T;
    const double ip = (p > kRcpFloor) ? fast_rcp(p) : 0.0;  // p == 0 (or next to it, see kRcpFloor): a dead sample must not carry NaN into the sums
    s.da = P * (lx + d[1]) * ip;
    s.dmu = T * (dx * d[5] + d[3]) * ip;
    s.dsg = T * (dx2 * d[6] + d[4]) * ip;
    s.dlam = (e_tn - e_pl) * ip;
    s.ddel = (P > 0.0) ? P * dlogS * ip : 0.0;  // S = 0: the power-law part and its delta-derivative vanish
    lin *= p;
    return 0.0;
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) {
    a.g[0] += w * s.da;
    a.g[1] += w * s.dmu;
    a.g[2] += w * s.dsg;
    a.g[3] += w * s.dlam;
    a.g[4] += w * s.ddel;
  }
  // the absorbing form (see Term<GWI_TERM_PLPEAK>::finish): returns the weight, states are w dl/dtheta
  __device__ static double finish(const TermD& t, const double* d, const Ctx& c, const In& in, State& s, double L, double E, int nshift) {
    const double x = in.x0;
    const double lx = in.x1;
    const double alpha = c.theta[t.th0], mu = c.theta[t.th1], lam = c.theta[t.th3];
    const double dx = x - mu;
    const double dx2 = dx * dx;
    double dlogS;
    const double S = taper(x - t.p0, c.theta[t.th4], dlogS);
    const double e_pl = fast_exp_shift(fma(alpha, lx, d[0] + E), nshift) * S;
    const double e_tn = fast_exp_shift(fma(-0.5 * dx2, d[5], d[2] + E), nshift);
    const double LP = (L * (1.0 - lam)) * e_pl, LT = (L * lam) * e_tn;
    s.da = LP * (lx + d[1]);
    s.dmu = LT * fma(dx, d[5], d[3]);
    s.dsg = LT * fma(dx2, d[6], d[4]);
    s.dlam = L * (e_tn - e_pl);
    s.ddel = (LP > 0.0) ? LP * dlogS : 0.0;  // S = 0: the power-law part and its delta-derivative vanish
    return LP + LT;
  }
  __device__ static void accumulate_weighted(double wfac, const State& s, Acc& a) {
    a.g[0] = fma(wfac, s.da, a.g[0]);
    a.g[1] = fma(wfac, s.dmu, a.g[1]);
    a.g[2] = fma(wfac, s.dsg, a.g[2]);
    a.g[3] = fma(wfac, s.dlam, a.g[3]);
    a.g[4] = fma(wfac, s.ddel, a.g[4]);
  }
  __device__ static void init(Acc& a) {
#pragma unroll
    for (int j = 0; j < 5; ++j) a.g[j] = 0;
  }
  __device__ static void rescale(Acc& a, double sc) {
#pragma unroll
    for (int j = 0; j < 5; ++j) a.g[j] *= sc;
  }
  static constexpr int kNumAcc = 5;
  __device__ static void collect(const TermD& t, const Acc& a, double* vals, int* th) {
#pragma unroll
    for (int j = 0; j < 5; ++j) vals[j] = a.g[j];
    th[0] = t.th0;
    th[1] = t.th1;
    th[2] = t.th2;
    th[3] = t.th3;
    th[4] = t.th4;
  }
};

template <>
struct Absorbs<GWI_TERM_PLPEAK> {
  static constexpr bool value = true;
};
template <>
struct Absorbs<GWI_TERM_PLPEAK_SMOOTH> {
  static constexpr bool value = true;
};

// x^alpha on [lo, hi] with the bounds themselves hyper-parameters (numpyro_distributions.py:101-136: Powerlaw with
// sampled minimum / maximum, examples/config_files/config.yml:8-25).  The truncation is a theta-dependent mask, so
// it cannot live in kappa: the term reads x next to log x and applies the reference's own test, x < lo | x > hi
// excluded (a sample exactly on a bound is inside).  The normaliser is sample-independent (host);
// d log_l / d lo = d log_l / d hi = 0 wherever log_l is differentiable (the normaliser cancels).
template <>
struct Term<GWI_TERM_POWERLAW_BOUNDS> {
  static constexpr bool kSpline = false;
  struct In {
    double x0, x1;
  };
  __device__ static void load(const double* const* tc, SIdx idx, In& in) {
    in.x0 = gload(tc[0], idx);
    in.x1 = gload(tc[1], idx);
  }
  struct State {
    double lx;
  };
  struct Acc {
    double g0;
  };
  __device__ static double eval(const TermD& t, const double*, const Ctx& c, const In& in, State& s, double& lin) {
    s.lx = in.x0;
    if ((in.x1 < c.theta[t.th1]) || (in.x1 > c.theta[t.th2])) {
      lin = 0.0;   // dead sample
      s.lx = 0.0;  // keep the gradient state finite whatever log x is out there
    }
    return c.theta[t.th0] * s.lx;
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) { a.g0 += w * s.lx; }
  GWI_ACC1(g0)
  static constexpr int kNumAcc = 1;
  __device__ static void collect(const TermD& t, const Acc& a, double* vals, int* th) {
    vals[0] = a.g0;
    th[0] = t.th0;
  }
};

// exp of a log-density tabulated on a grid and linearly interpolated between grid points
// (BSplineDistribution, numpyro_distributions.py:266-293: lpdfs = cs . grid_dmat, log_prob = interp(value, grid, lpdfs)).
// cols[0] = fractional grid index u = j + f of the sample (static; outside the grid np.interp holds the end value,
// so u is clamped by the caller); th1 = index of the grid normaliser, whose `us` table holds the spline coordinate
// of every grid point.  l = (1-f) L_j + f L_{j+1}, L_g = sum_k c_k B_k(us_g); d l / d c_k is the same blend of taps.
template <>
struct Term<GWI_TERM_EXP_SPLINE_LERP> {
  static constexpr bool kSpline = true;
  struct In {
    double x0;
  };
  __device__ static void load(const double* const* tc, SIdx idx, In& in) { in.x0 = gload(tc[0], idx); }
  struct State {
    double t0, t1, f;
    int k0, k1;  // -1: node contributes nothing (outside a zero-outside basis, or zero blend weight)
  };
  struct Acc {};
  __device__ static double node(const TermD& t, const Ctx& c, double sx, double wt, int& k, double& tt, double& lin) {
    spline_locate_x(sx, t, k, tt);
    double v = spline_value(c, t.th0 + k, tt);
    if (!((sx >= t.p0) && (sx <= t.p1))) {
      k = -1;
      v = 0.0;
      // a log-Y basis is -inf out there (interpolation.py:407, :449): the interpolated log-density is -inf
      if (!(t.flags & GWI_SPLINE_OUTSIDE_ZERO_EXPONENT) && wt != 0.0) lin = 0.0;
    }
    if (wt == 0.0) k = -1;
    return v;
  }
  __device__ static double eval(const TermD& t, const double*, const Ctx& c, const In& in, State& s, double& lin) {
    const NormD& nd = c.a->norms[t.th1];
    const double u = in.x0;
    int j = (int)u;
    j = max(0, min(j, nd.n_pts - 2));
    const double f = u - (double)j;
    const double s0 = gload(nd.us, j), s1 = gload(nd.us, j + 1);
    const double l0 = node(t, c, s0, 1.0 - f, s.k0, s.t0, lin);
    const double l1 = node(t, c, s1, f, s.k1, s.t1, lin);
    s.f = f;
    return fma(f, l1 - l0, l0);
  }
  __device__ static void accumulate(const TermD& t, const Ctx& c, double w, const State& s, Acc&) {
    if (w != 0.0) {
      if (s.k0 >= 0) spline_scatter(c, t.th0 + s.k0, cubic_taps_weighted(s.t0, make_weight(w * (1.0 - s.f))));
      if (s.k1 >= 0) spline_scatter(c, t.th0 + s.k1, cubic_taps_weighted(s.t1, make_weight(w * s.f)));
    }
  }
  __device__ static void init(Acc&) {}
  __device__ static void rescale(Acc&, double) {}
  static constexpr int kNumAcc = 0;
  __device__ static void collect(const TermD&, const Acc&, double*, int*) {}
};

// ---- compile-time chain of terms.  U = samples per lane per trip; inputs are double-buffered in
//      registers (in[0] = current trip, in[1] = next trip) so the column loads of trip k+1 are in
//      flight while trip k is being evaluated. ----------------------------------------------------------
//      TAKEN: a term further up the chain already absorbs the closing exponent (Absorbs<K>); the first absorbing term is
//      left out of eval() and evaluated by finish(), with the exponent l - m folded into its own.
template <int U, bool TAKEN, int... Ks>
struct ChainImpl;
template <int U, bool TAKEN>
struct ChainImpl<U, TAKEN> {
  static constexpr bool kGeneric = false;
  static constexpr bool kSpline = false;
  static constexpr bool kAbsorb = false;
  static constexpr int kNumAcc = 0;
  __device__ void init() {}
  __device__ void load(int, int, int, const Ctx&, SIdx) {}
  __device__ void advance() {}
  __device__ double eval(int, int, const Ctx&, double&) { return 0.0; }
  __device__ double finish(int, int, const Ctx&, double, double, int) { return 0.0; }
  __device__ void accumulate(int, int, const Ctx&, double, double) {}
  __device__ void rescale(double) {}
  __device__ void collect(int, const Ctx&, double*, int*) {}
};
template <int U, bool TAKEN, int K, int... Rest>
struct ChainImpl<U, TAKEN, K, Rest...> {
  static constexpr bool kDefer = !TAKEN && Absorbs<K>::value;
  using RestT = ChainImpl<U, TAKEN || kDefer, Rest...>;
  static constexpr bool kGeneric = false;
  static constexpr bool kSpline = Term<K>::kSpline || RestT::kSpline;
  static constexpr bool kAbsorb = kDefer || RestT::kAbsorb;
  static constexpr int kNumAcc = Term<K>::kNumAcc + RestT::kNumAcc;
  typename Term<K>::In in[2][U];
  typename Term<K>::State st[U];
  typename Term<K>::Acc acc;
  RestT rest;
  __device__ void init() {
    Term<K>::init(acc);
    rest.init();
  }
  __device__ void load(int buf, int u, int ti, const Ctx& c, SIdx idx) {
    Term<K>::load(c.tcols[ti], idx, in[buf][u]);
    rest.load(buf, u, ti + 1, c, idx);
  }
  __device__ void advance() {
#pragma unroll
    for (int u = 0; u < U; ++u) in[0][u] = in[1][u];
    rest.advance();
  }
  __device__ double eval(int u, int ti, const Ctx& c, double& lin) {
    if constexpr (kDefer) {
      return rest.eval(u, ti + 1, c, lin);
    } else {
      const double l = Term<K>::eval(c.a->terms[ti], c.derived[ti], c, in[0][u], st[u], lin);
      return l + rest.eval(u, ti + 1, c, lin);
    }
  }
  // the deferred term: returns the sample's weight L x (its density) x exp(E) 2^-nshift, its states pre-weighted (Term<K>::finish)
  __device__ double finish(int u, int ti, const Ctx& c, double L, double E, int nshift) {
    if constexpr (kDefer)
      return Term<K>::finish(c.a->terms[ti], c.derived[ti], c, in[0][u], st[u], L, E, nshift);
    else
      return rest.finish(u, ti + 1, c, L, E, nshift);
  }
  // w: the sample's weight (squared in a squared-weight pass); wfac: what the deferred term's pre-weighted states take (1, or
  // the unsquared weight in a squared-weight pass)
  __device__ void accumulate(int u, int ti, const Ctx& c, double w, double wfac) {
    if constexpr (kDefer)
      Term<K>::accumulate_weighted(wfac, st[u], acc);
    else
      Term<K>::accumulate(c.a->terms[ti], c, w, st[u], acc);
    rest.accumulate(u, ti + 1, c, w, wfac);
  }
  __device__ void rescale(double sc) {
    Term<K>::rescale(acc, sc);
    rest.rescale(sc);
  }
  __device__ void collect(int ti, const Ctx& c, double* vals, int* th) {
    Term<K>::collect(c.a->terms[ti], acc, vals, th);
    rest.collect(ti + 1, c, vals + Term<K>::kNumAcc, th + Term<K>::kNumAcc);
  }
};

// ---- the generic chain (kind sequence {0}): term kinds read from the argument block at RUN time, for products of
//      densities that have no compiled chain (the reference's user model multiplies any densities:
//      tests/inference_test.py:256-260, examples/simple_bspline_example.py:58-71).  Nothing is held per term: a sample's
//      terms are evaluated once for the value (columns loaded on the spot, no prefetch) and once more for the gradient,
//      whose every entry -- scalar parameters included -- goes through the workgroup's LDS rows like the spline
//      coefficients of the compiled chains (kSpline = true selects that mode of scan_kernel).  No register state indexed by
//      a run-time term number, hence no scratch; 2-2.7 x the scan time of a compiled chain, which gwi_create obtains from
//      hipRTC where it can (gwi_jit.h); this kernel is what runs where it cannot.
constexpr int kGenericChain = 0;
#define GWI_FOR_EACH_KIND(X)                                                                                             \
  X(GWI_TERM_POWERLAW) X(GWI_TERM_PLPEAK) X(GWI_TERM_POWERLAW_RATIO) X(GWI_TERM_BETA) X(GWI_TERM_TILT_MIXTURE)           \
  X(GWI_TERM_POWERLAW_REDSHIFT) X(GWI_TERM_EXP_SPLINE) X(GWI_TERM_TRUNCNORM) X(GWI_TERM_LINEAR_SPLINE) X(GWI_TERM_TILT_JOINT) \
  X(GWI_TERM_SMOOTH) X(GWI_TERM_PLPEAK_SMOOTH) X(GWI_TERM_POWERLAW_BOUNDS) X(GWI_TERM_EXP_SPLINE_LERP)
template <int K>
__device__ __forceinline__ double generic_value(const TermD& t, const double* d, const Ctx& c, const double* const* tc, SIdx idx, double& lin) {
  typename Term<K>::In in;
  typename Term<K>::State st;
  Term<K>::load(tc, idx, in);
  return Term<K>::eval(t, d, c, in, st, lin);
}
template <int K>
__device__ __forceinline__ void generic_gradient(const TermD& t, const double* d, const Ctx& c, const double* const* tc, SIdx idx, double w) {
  typename Term<K>::In in;
  typename Term<K>::State st;
  typename Term<K>::Acc acc;
  Term<K>::load(tc, idx, in);
  double lin = 1.0;
  Term<K>::eval(t, d, c, in, st, lin);
  Term<K>::init(acc);
  Term<K>::accumulate(t, c, w, st, acc);  // spline kinds scatter into the rows themselves; the others leave w dl/dtheta in acc
  constexpr int kN = Term<K>::kNumAcc;
  if constexpr (kN > 0) {
    double vals[kN];
    int th[kN];
    Term<K>::collect(t, acc, vals, th);
#pragma unroll
    for (int j = 0; j < kN; ++j)
      if (vals[j] != 0.0) unsafeAtomicAdd(c.gacc + (th[j] << c.rep_shift), vals[j]);
  }
}
template <int U>
struct ChainImpl<U, false, kGenericChain> {
  static constexpr bool kGeneric = true;
  static constexpr bool kSpline = true;
  static constexpr bool kAbsorb = false;
  static constexpr int kNumAcc = 0;
  SIdx idx[2][U];
  __device__ void init() {}
  __device__ void load(int buf, int u, int, const Ctx&, SIdx i) { idx[buf][u] = i; }
  __device__ void advance() {
#pragma unroll
    for (int u = 0; u < U; ++u) idx[0][u] = idx[1][u];
  }
  __device__ double eval(int u, int, const Ctx& c, double& lin) {
    double l = 0.0;
    for (int t = 0; t < c.a->n_terms; ++t) {
      const TermD& td = c.a->terms[t];
      switch (td.kind) {
#define GWI_X(K) \
  case K: l += generic_value<K>(td, c.derived[t], c, c.tcols[t], idx[0][u], lin); break;
        GWI_FOR_EACH_KIND(GWI_X)
#undef GWI_X
        default: lin = 0.0;
      }
    }
    return l;
  }
  __device__ double finish(int, int, const Ctx&, double, double, int) { return 0.0; }
  __device__ void accumulate(int u, int, const Ctx& c, double w, double) {
    if (w == 0.0) return;
    c.wt = make_weight(w);
    for (int t = 0; t < c.a->n_terms; ++t) {
      const TermD& td = c.a->terms[t];
      switch (td.kind) {
#define GWI_X(K) \
  case K: generic_gradient<K>(td, c.derived[t], c, c.tcols[t], idx[0][u], w); break;
        GWI_FOR_EACH_KIND(GWI_X)
#undef GWI_X
        default: break;
      }
    }
  }
  __device__ void rescale(double) {}
  __device__ void collect(int, const Ctx&, double*, int*) {}
};

template <int U, int... Ks>
using Chain = ChainImpl<U, false, Ks...>;

// ---- grid normalisers (interpolation.py:280-291, parametric.py:123-124, spline_perturbation.py:323-336):
//      Z_j = sum_g tw_g exp(lb_g + (theta+add) l1_g + spline(us_g)), one workgroup per normaliser.
//      Only the HOST consumes Z (it rescales sites; Z cancels in log_l and its gradient).  The integration is
//      norm_block() below, run by the FIRST n_norms workgroups of the scan launch (they finish long before the
//      scan does, and a launch of their own cost the host 4 us per evaluation); results go straight to pinned
//      host memory with a stamp per normaliser.

// ---- publishing to pinned host memory without a system-scope fence ---------------------------------
// __threadfence_system() = write back the whole L2 + invalidate (several us).  Results bound for the
// host are instead stored write-through at system scope (global_store ... sc0 sc1: they bypass the
// caches), every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup meets at a
// barrier, and only then one lane stores the completion stamp, also write-through
// (MI355X_MICROARCH.md: "sc1 payload -> vmcnt(0) -> sc1 flag" hand-off form).
__device__ __forceinline__ void store_sys(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void publish_stamp(double* slot, unsigned long long seq, int tid) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(slot), seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double lane_bcast(double v, int src) {  // src is wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// ---- stage 2: combine the tile records of one GROUP with a common reference exponent.  Groups
//      0..n_ev-1 are the events (all tiles of one event); groups n_ev.. split the injection tiles
//      into runs of <= tiles_per_inj_group.  Latency is everything here (a few KB per group), so there
//      is NO workgroup barrier: tiles run across the lanes of a wave, record values across the
//      waves; every wave derives the group's exponent and S1 itself (DPP reductions, fixed order).
//      One workgroup per group (e = group, kb = hyper-parameter point): 64 threads (one wave) when the gradient has at most
//      60 slots, else kBlock -- a wave per group is all the parametric models need, and a batched launch runs K x groups of them.
// Every value the scan launch wrote is read with a load that bypasses the caches (system scope: sc0 sc1) -- the same trip
// to memory a miss would make right after the launch's cache invalidate, so it costs nothing; in exchange the combine
// PACKET needs no acquire fence at all (gwi_aql.h: dispatch_staged, acquire = false): its constant argument block stays in
// the scalar cache / L2 from one evaluation to the next instead of being fetched from memory first, and the packet
// processor skips the invalidate.
__device__ __forceinline__ double load_fresh(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void combine_group(const TailArgs& a, const int e, const int kb, const int tid) {
  const int lane = tid & 63;
  const bool is_inj = e >= a.n_ev;
  int n_tiles;
  long long first;
  if (is_inj) {
    const int j = e - a.n_ev;
    const int t0 = j * a.tiles_per_inj_group;
    n_tiles = a.n_inj_tiles - t0 < a.tiles_per_inj_group ? a.n_inj_tiles - t0 : a.tiles_per_inj_group;
    first = (long long)a.n_ev * a.tiles_per_event + t0;
  } else {
    n_tiles = a.tiles_per_event;
    first = (long long)e * a.tiles_per_event;
  }
  const int n_groups = a.n_ev + a.n_inj_groups;
  const double* rec = a.partials + ((long long)kb * a.n_scan_blocks + first) * a.rec_stride;
  double* const ev_out = a.ev_out + (long long)kb * a.n_ev * 4;
  double* const ev_grad = a.ev_grad + (long long)kb * a.n_ev * a.n_theta;
  double* const inj_out = a.inj_out + (long long)kb * a.n_inj_groups * 4;
  double* const inj_grad = a.inj_grad + (long long)kb * a.n_inj_groups * a.n_theta;
  double* const ev_host = a.ev_host + (long long)kb * 3 * a.n_ev;
  const int row_doubles = 8 * a.row_lines;  // row_lines = ceil((3 + n_theta) / 7)
  double* const host_rows = a.host_rows ? a.host_rows + (long long)kb * n_groups * row_doubles : nullptr;
  // host-final mode publishes one row per group: the row is staged in LDS and leaves in one contiguous sweep (2
  // PCIe writes per row instead of 5).  Small posted writes are what the host waits for: config 2, 76 rows:
  // 26.0 -> 21.1 us per evaluation on the same box; 16-point batches 111 -> 93 us
  __shared__ double s_row[4 + GWI_MAX_THETA];

  // phase 1, every wave redundantly: lanes <- tiles.  Common exponent M, per-tile factor f_t, S1, S2.
  // (host guarantees n_tiles <= 64 per group)
  const bool has = lane < n_tiles;
  const double* mine = rec + (long long)(has ? lane : 0) * a.rec_stride;
  const double m_t = has ? load_fresh(mine) : GWI_NEG_INF;
  const double r1 = has ? load_fresh(mine + 1) : 0.0, r2 = has ? load_fresh(mine + 2) : 0.0;
  // the first gradient slot of this thread: its tile values are requested NOW, together with the
  // headers, so that one memory round trip (not two) precedes the arithmetic
  constexpr int kEarly = 16;
  double early[kEarly];
  const bool early_on = tid < a.n_theta;
  {
    const double* col0 = rec + kRecHeader + (early_on ? tid : 0);
#pragma unroll
    for (int t = 0; t < kEarly; ++t) early[t] = (early_on && t < n_tiles) ? load_fresh(col0 + (long long)t * a.rec_stride) : 0.0;
  }
  const double M = wave_max(m_t);
  const double f = (m_t == GWI_NEG_INF) ? 0.0 : exp(m_t - M);
  const double S1 = wave_sum(f * r1);
  const double inv_s1 = S1 > 0.0 ? 1.0 / S1 : 0.0;
  // phase 2: threads <- gradient slots (coalesced across p), tiles in order with f_t broadcast from
  // its lane: no cross-lane reduction, no barrier, fixed summation order
  for (int p = tid; p < a.n_theta; p += a.combine_threads) {
    double acc = 0.0;
    const double* col = rec + kRecHeader + p;
    int t = 0;
    if (p == tid) {
#pragma unroll
      for (; t < kEarly; ++t)
        if (t < n_tiles) acc += lane_bcast(f, t) * early[t];
      t = n_tiles < kEarly ? n_tiles : kEarly;
    }
#pragma unroll 4
    for (; t < n_tiles; ++t) acc += lane_bcast(f, t) * load_fresh(col + (long long)t * a.rec_stride);
    if (host_rows)
      s_row[3 + p] = is_inj ? acc : acc * inv_s1;
    else if (is_inj)
      inj_grad[(long long)(e - a.n_ev) * a.n_theta + p] = acc;
    else
      ev_grad[(long long)e * a.n_theta + p] = acc * inv_s1;
  }
  if (tid < 64) {
    const double S2 = wave_sum(f * f * r2);
    if (lane == 0 && host_rows) {
      if (is_inj) {
        s_row[0] = M;
        s_row[1] = S1;
        s_row[2] = S2;
      } else {
        const double log_s1 = log(S1);
        const double log_neff = 2.0 * log_s1 - log(S2);  // analysis.py:79
        s_row[0] = log_s1 + M;
        s_row[1] = log_neff;
        s_row[2] = 1.0 / exp(log_neff) - 1.0 / a.n_pe;  // :87
      }
    } else if (lane == 0) {
      if (is_inj) {
        double* o = inj_out + (long long)(e - a.n_ev) * 4;
        o[0] = M;
        o[1] = S1;
        o[2] = S2;
      } else {
        // analysis.py:78-87: logBF = logsumexp - log N_pe (constant added on the host),
        // log n_eff = 2 logsumexp(l) - logsumexp(2 l), variance = 1/n_eff - 1/N_pe
        const double log_s1 = log(S1);
        const double log_neff = 2.0 * log_s1 - log(S2);
        const double var = 1.0 / exp(log_neff) - 1.0 / a.n_pe;
        double* o = ev_out + (long long)e * 4;
        o[0] = log_s1 + M;
        o[1] = log_neff;
        o[2] = var;
        o[3] = S1;
        if (a.publish_events) {
          store_sys(ev_host + e, log_s1 + M);
          store_sys(ev_host + a.n_ev + e, log_neff);
          store_sys(ev_host + 2 * a.n_ev + e, var);
        }
      }
    }
  }
  if (host_rows) {  // the whole row leaves as self-validating 64-byte lines (seven values + the sequence number)
    __syncthreads();
    unsigned long long* o = reinterpret_cast<unsigned long long*>(host_rows + (long long)e * row_doubles);
    const unsigned long long seq = __hip_atomic_load(a.seq_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int n_vals = 3 + a.n_theta;
    for (int slot = tid; slot < row_doubles; slot += a.combine_threads) {
      const int line = slot >> 3, j = slot & 7, i = line * 7 + j;
      const unsigned long long bits = j == 7 ? seq : (i < n_vals ? (unsigned long long)__double_as_longlong(s_row[i]) : 0ull);
      __hip_atomic_store(o + slot, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// ---- stage 3: reduce over events, merge the injection groups, and publish this device's record to
//      pinned host memory.  One workgroup of THREADS; events (or injection groups) run across the
//      lanes, output values across the waves; one barrier, before the completion stamp.
//      record layout (doubles): see kRecNormOff in gwi_engine.hip -------------------------------------
__device__ __forceinline__ int pow2_at_least(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// Workgroup g of G takes the events [g n_ev / G, (g + 1) n_ev / G) and the injection groups [g n_ig / G, (g + 1) n_ig / G)
// and publishes a partial record of its own (same layout); the host merges the G records as it merges the records of
// ranks.  One workgroup over everything (G = 1: the sharded path, whose record feeds the all-gather) spent 7.4 us on
// config 5 -- 200 event rows x 105 slots through one CU, two dependent rounds of loads -- against 5.8 for G = 4.
template <int THREADS>
__device__ __forceinline__ void final_reduce(const TailArgs& a, const int kb, const int g, const int tid, double* s_tile /* [THREADS] */) {
  const int lane = tid & 63, wave = tid >> 6;
  const int G = a.final_groups;
  const int e_lo = (int)((long long)g * a.n_ev / G), e_hi = (int)((long long)(g + 1) * a.n_ev / G);
  const int j_lo = (int)((long long)g * a.n_inj_groups / G), j_hi = (int)((long long)(g + 1) * a.n_inj_groups / G);
  const int n_ig = j_hi - j_lo;
  double* r = a.record + ((long long)kb * G + g) * a.record_len;
  const double* const ev_out = a.ev_out + (long long)kb * a.n_ev * 4;
  const double* const ev_grad = a.ev_grad + (long long)kb * a.n_ev * a.n_theta;
  const double* const inj_out = a.inj_out + ((long long)kb * a.n_inj_groups + j_lo) * 4;
  const double* const inj_grad = a.inj_grad + ((long long)kb * a.n_inj_groups + j_lo) * a.n_theta;
  const int off_norm = 8, off_gpe = off_norm + a.n_norms, off_ginj = off_gpe + a.n_theta;

  // ---- gradient sums over events: threads <- (event row, slot p); p fast => coalesced; one barrier
  const int vp = pow2_at_least(a.n_theta < 8 ? 8 : a.n_theta);  // <= 256 = GWI_MAX_THETA
  const int rows = THREADS / vp;
  const int row = tid / vp, col = tid - row * vp;
  double acc = 0.0;
  if (col < a.n_theta) {
    // chunks of 16 events: the 16 loads go out together (one memory round trip), then the ordered sum
    for (int e0 = e_lo + row; e0 < e_hi; e0 += 16 * rows) {
      double v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int e = e0 + q * rows;
        v[q] = e < e_hi ? ev_grad[(long long)e * a.n_theta + col] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) acc += v[q];
    }
  }
  s_tile[tid] = acc;

  // ---- meanwhile wave 0: scalar sums over events; wave 1: injection groups
  if (wave == 0) {
    double sum = 0.0, var = 0.0, mn = GWI_POS_INF;
    for (int e0 = e_lo + lane; e0 < e_hi; e0 += 4 * 64) {  // four events per lane per round trip
      double o0[4], o1[4], o2[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = e0 + 64 * q;
        const double* o = ev_out + (long long)(e < e_hi ? e : e_lo) * 4;
        o0[q] = o[0];
        o1[q] = o[1];
        o2[q] = o[2];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (e0 + 64 * q >= e_hi) continue;
        sum += o0[q];
        var += o2[q];
        double le = o1[q];  // jnp.min(jnp.nan_to_num(logn_effs)) (analysis.py:295)
        if (le != le) le = 0.0;
        le = fmin(fmax(le, -1.7976931348623157e308), 1.7976931348623157e308);
        mn = fmin(mn, le);
      }
    }
    sum = wave_sum(sum);
    var = wave_sum(var);
    mn = -wave_max(-mn);
    if (lane == 0) {
      store_sys(r + 1, sum);
      store_sys(r + 2, var);
      store_sys(r + 3, mn);
      // a negative event count asks whoever assembles the gathered records to repeat the evaluation in two-pass mode
      store_sys(r + 7, (*a.redo_ptr == *a.seq_ptr) ? -(double)(e_hi - e_lo + 1) : (double)(e_hi - e_lo));
    }
  }
  // injection groups (host guarantees n_inj_groups <= 64): lanes <- groups.  Everything this thread
  // will need from memory is requested before the first dependent instruction (one round trip).
  const bool hasg = lane < n_ig;
  const double m_j = hasg ? inj_out[lane * 4] : GWI_NEG_INF;
  const double s1_j = hasg ? inj_out[lane * 4 + 1] : 0.0, s2_j = hasg ? inj_out[lane * 4 + 2] : 0.0;
  constexpr int kEarly = 32;
  double early[kEarly];
  const bool early_on = tid < a.n_theta;
#pragma unroll
  for (int j = 0; j < kEarly; ++j) early[j] = (early_on && j < n_ig) ? inj_grad[(long long)j * a.n_theta + tid] : 0.0;
  const double Minj = wave_max(m_j);
  const double fj = (m_j == GWI_NEG_INF) ? 0.0 : exp(m_j - Minj);
  if (wave == 1) {
    const double S1 = wave_sum(fj * s1_j);
    const double S2 = wave_sum(fj * fj * s2_j);
    if (lane == 0) {
      store_sys(r + 4, Minj);
      store_sys(r + 5, S1);
      store_sys(r + 6, S2);
    }
  }
  // injection gradient numerators: threads <- slots, groups in order with f_j broadcast
  for (int p = tid; p < a.n_theta; p += THREADS) {
    double gsum = 0.0;
    int j = 0;
    if (p == tid) {
#pragma unroll
      for (; j < kEarly; ++j)
        if (j < n_ig) gsum += lane_bcast(fj, j) * early[j];
      j = n_ig < kEarly ? n_ig : kEarly;
    }
    for (; j < n_ig; j += 8) {  // eight independent loads per round trip, then the ordered sum
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = (j + q < n_ig) ? inj_grad[(long long)(j + q) * a.n_theta + p] : 0.0;
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (j + q < n_ig) gsum += lane_bcast(fj, j + q) * v[q];
    }
    store_sys(r + off_ginj + p, gsum);
  }
  __syncthreads();
  if (row == 0 && col < a.n_theta) {
    double gsum = 0.0;
    for (int q = 0; q < rows; ++q) gsum += s_tile[q * vp + col];  // fixed order
    store_sys(r + off_gpe + col, gsum);
  }
  publish_stamp(r, *a.seq_ptr, tid);
}

// ---- one grid normaliser Z_j(theta), integrated by one workgroup (see the comment above NormD's users):
//      Z = sum_g tw_g exp(lb_g + (theta + add) l1_g + spline(us_g)), published to pinned host memory with a stamp
__device__ __forceinline__ void norm_block(const NormD* norms, const double* theta_src, int n_theta, int j, double* out_slot, unsigned long long* stamp_slot,
                                           unsigned long long seq, double* s_theta, double* s_part /* [kWaves] */) {
  const int tid = threadIdx.x;
  for (int p = tid; p < n_theta; p += kBlock) s_theta[p] = theta_src[p];
  __syncthreads();
  const NormD nd = norms[j];
  double acc = 0.0;
  const double expo = nd.expo_theta >= 0 ? s_theta[nd.expo_theta] + nd.expo_add : 0.0;
  const double inv_dx = nd.n_basis > 0 ? (double)(nd.n_basis - 3) / (nd.hi - nd.lo) : 0.0;
  for (int g = tid; g < nd.n_pts; g += kBlock) {
    const double tw = nd.tw[g];
    double e = nd.lb ? nd.lb[g] : 0.0;
    if (nd.expo_theta >= 0) e += expo * nd.l1[g];
    if (nd.n_basis > 0) {
      const double x = nd.us[g];
      int k;
      double tt;
      spline_locate(x, nd.lo, inv_dx, nd.n_basis, k, tt);
      const Taps b = cubic_taps(tt);
      const double* cf = s_theta + nd.coef_off + k;
      double v = cf[0] * b.b0 + cf[1] * b.b1 + cf[2] * b.b2 + cf[3] * b.b3;
      if ((nd.flags & (GWI_SPLINE_OUTSIDE_ZERO_EXPONENT | GWI_NORM_LINEAR_SPLINE)) && !((x >= nd.lo) && (x <= nd.hi))) v = 0.0;
      if (nd.flags & GWI_NORM_LINEAR_SPLINE) {  // BSpline.norm: trapz of the spline itself
        acc += tw * v;
        continue;
      }
      e += v;
    }
    if (tw != 0.0) acc += tw * fast_exp(e);
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) s_part[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) store_sys(out_slot, (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) __hip_atomic_store(stamp_slot, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- the scan's preloaded arguments ----------------------------------------------------------------
// The first 16 dwords of a kernel's SCALAR arguments can be placed in scalar registers by the command processor before a wave
// starts (kernel-argument preload; -mllvm -amdgpu-kernarg-preload-count=16; by-value structs are not covered).  The scan
// takes there exactly what its first column loads depend on -- the tile geometry and the first five column pointers -- so
// that those loads leave at wave entry instead of after a scalar round trip to the argument block in memory (0.7 us: a
// launch starts with invalidated caches), and the argument block's own first trip runs beside them instead of before them
// (tools/microbench/l2_resident.cpp: a streaming stand-in with 48 FMAs per sample 5.5 -> 4.7 us at config 2's size).
// One pointer serves both sample sets: every column is ONE allocation, posterior samples first, the injections
// `inj_off` elements behind (gwi_engine.hip: alloc_pair), so a workgroup's choice between the sets is an offset, not a load.
// Slot 0 is kappa; the terms' columns follow in term order (Term<K>::In holds one double per column); columns beyond the
// fifth slot are read from KArgs::pe_tcols as before.
constexpr int kHeadCols = 5;
struct ScanHead {  // 14 dwords: the kernel-argument pointer takes two of the sixteen user registers
  const double* col[kHeadCols];
  unsigned geom;    // n_ev (bits 0-19) | tiles_per_event (20-26) | n_norms (27-31)
  unsigned chunks;  // tile sizes of the two sample sets, 16 bits each (pack_chunk)
  unsigned n_pe, n_inj;
};
static_assert(sizeof(ScanHead) == 56, "fourteen dwords: what the command processor preloads next to the kernel-argument pointer");
constexpr unsigned kGeomEventBits = 20, kGeomTilesBits = 7;
// a tile size in 16 bits: exact below 32 768, in units of 256 samples above (tiles that large are whole trips: multiples of 256)
__host__ __device__ inline bool chunk_packs(long long c) { return c > 0 && (c < 32768 || (c % 256 == 0 && c / 256 < 32768)); }
__host__ __device__ inline unsigned pack_chunk(long long c) { return c < 32768 ? (unsigned)c : (0x8000u | (unsigned)(c / 256)); }
__host__ __device__ inline int unpack_chunk(unsigned v) { return (v & 0x8000u) ? (int)((v & 0x7fffu) * 256u) : (int)v; }
// where the injections start inside every column (elements): behind the posterior samples, on a 256-byte boundary
__host__ __device__ inline long long inj_offset(long long n_ev, long long n_pe) { return (n_ev * n_pe + 31) / 32 * 32; }

template <int SLOT, int T, int... Ks>
struct ColFill;
template <int SLOT, int T>
struct ColFill<SLOT, T> {
  __device__ static void run(const double* (*)[2], const double* const*, const KArgs&) {}
};
template <int SLOT, int T, int K, int... Rest>
struct ColFill<SLOT, T, K, Rest...> {
  static constexpr int kCols = (int)(sizeof(typename Term<K>::In) / sizeof(double));
  __device__ __forceinline__ static void run(const double* (*lc)[2], const double* const* head, const KArgs& a) {
#pragma unroll
    for (int k = 0; k < 2; ++k) lc[T][k] = k < kCols ? (SLOT + k < kHeadCols ? head[SLOT + k < kHeadCols ? SLOT + k : 0] : a.pe_tcols[T][k]) : nullptr;
    ColFill<SLOT + kCols, T + 1, Rest...>::run(lc, head, a);
  }
};

// ---- kernel-argument warm-up ---------------------------------------------------------------------
// The caches are invalidated when a launch starts, so a wave's first scalar load of every 64-byte line of the argument block
// goes to memory, and the compiler places those loads where their values are first used: behind branches, one dependent round
// trip after the other (header -> sizes -> PE-or-injection choice -> column pointers: 1.0 us from wave entry to the first
// column load at config 2, 1.8 at config 3 by the phase stamps).  One dword of every line the start-up and the first trip
// will read, loaded in ONE clause at wave entry, turns all but the first of those round trips into scalar-cache hits.
template <int N_TERMS, bool BATCHED>
struct KernargWarm {
  // byte ranges of the argument block -> one dword per 64-byte line: the column pointers (those beyond the preloaded ones come
  // from here); pointers, sizes and the terms' descriptors; the per-evaluation tail up to the last term's derived scalars; the
  // first sixteen hyper-parameters (a parametric model reads its scalars from there with scalar loads whose addresses come out
  // of the term descriptors: a second full round trip behind the first otherwise)
  static constexpr int kRanges = 4;
  static constexpr size_t lo(int r) {
    return r == 0 ? offsetof(KArgs, pe_tcols) : r == 1 ? offsetof(KArgs, kappa_pe) : r == 2 ? offsetof(KArgs, norm_seq) : offsetof(KArgs, theta);
  }
  static constexpr size_t hi(int r) {
    return r == 0   ? offsetof(KArgs, pe_tcols) + 16 * N_TERMS
           : r == 1 ? offsetof(KArgs, terms) + sizeof(TermD) * N_TERMS
           : r == 2 ? offsetof(KArgs, derived) + sizeof(double) * kMaxDerived * N_TERMS
                    : offsetof(KArgs, theta) + 128;
  }
  static constexpr int count() {
    int n = 0;
    for (int r = 0; r < kRanges; ++r)
      for (size_t o = lo(r) & ~(size_t)63; o < hi(r); o += 64) ++n;
    return n;
  }
  static constexpr size_t offset(int k) {
    int n = 0;
    for (int r = 0; r < kRanges; ++r)
      for (size_t o = lo(r) & ~(size_t)63; o < hi(r); o += 64) {
        if (n == k) return o;
        ++n;
      }
    return 0;
  }
  // single evaluations only: a batched launch runs many workgroups per CU one after the other, whose argument lines are in the
  // scalar cache after the first -- there the fourteen extra loads per wave cost the config-2 batch 11 % (4.65 -> 5.2 us per
  // evaluation at K = 16) for nothing
#ifndef GWI_AB_NO_KERNARG_WARM
  static constexpr int kN = BATCHED ? 0 : count();
#else
  static constexpr int kN = 0;
#endif
  int t[kN > 0 ? kN : 1];
  __device__ __forceinline__ void issue() {
    // the argument block inside the kernel-argument segment (behind the preloaded scalars); not &a: taking the address of the
    // by-value struct for an asm operand makes the compiler copy all of it to scratch.  Written out as asm: left to itself the
    // compiler widens, merges and serialises such loads (a wait between every two)
    const char __attribute__((address_space(4)))* w = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(ScanHead);
#pragma unroll
    for (int k = 0; k < kN; ++k) asm volatile("s_load_dword %0, %1, 0x0" : "=&s"(t[k]) : "s"(w + offset(k)));
  }
  // every destination stays allocated until here (nothing else can land in it while its load is in flight); the wait comes
  // after whatever the caller put in between (the first trip's column loads).  The compiler does not count these loads.
  __device__ __forceinline__ void settle() {
    if (kN > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < kN; ++k) asm volatile("" ::"s"(t[k]));
  }
};

// ---- the scan kernel -----------------------------------------------------------------------------
// grid = n_ev*tiles_per_event PE workgroups + n_inj_tiles injection workgroups.  A PE workgroup owns `chunk_pe` consecutive samples of ONE event, so its record
// belongs to that event's logsumexp; an injection workgroup owns `chunk_inj` consecutive
// injections.  Loads are coalesced: lane i of a wave reads element base+i of each column; each lane
// carries kU samples (256 apart) per trip so polynomial constants, the wave maximum and the loop
// overhead are shared between them.
constexpr int kRedChunk = 8;
constexpr int kRegularRepShift = 4;  // 16 gradient-row replicas in the regular (non-SAFE) scan kernels

#ifndef GWI_SCAN_WAVES_PER_EU
#define GWI_SCAN_WAVES_PER_EU 1
#endif
// SAFE (spline models only): the fallback instantiation with the two-pass sweep (last resort: a repeat that still misses)
// and the replay (deterministic) mode as run-time options -- and the batch index as one too, so that there is ONE such kernel per term sequence.  Carrying these
// options in the regular kernel cost it 50-90 VGPRs (config 5: 125 -> 213), i.e. one or two resident waves per SIMD.
template <bool WRITE_LOGW, bool BATCH, bool SAFE, int U, int... Ks>
__global__ __launch_bounds__(kBlock, GWI_SCAN_WAVES_PER_EU) void scan_kernel(const double* hc0, const double* hc1, const double* hc2, const double* hc3, const double* hc4, const unsigned h_geom,
                                                                             const unsigned h_chunks, const unsigned hu_n_pe, const unsigned hu_n_inj, const KArgs a) {
  using ChainT = Chain<U, Ks...>;
  constexpr int kU = U;
  // Models with spline terms ("shared" mode): the workgroup keeps ONE set of gradient rows in LDS ([coefficient][replica],
  // see spline_scatter) that all four waves add into, so every wave must weigh its samples against the SAME reference
  // exponent, known before the first sample is weighed.  It is the tile's exact maximum at the previous evaluation
  // (KArgs::tile_nref; a scalar load, no barrier, no wave maximum in the loop) applied as an exact power of two, and the
  // record is normalised so that its bits do not depend on it; the tile's true maximum is tracked on the side and, should
  // it lie outside the safe range around the reference, the host repeats the evaluation, which then finds the exact
  // reference in place (see n_ref below).  Parametric models keep the per-wave online maximum (registers only).
  constexpr bool kShared = ChainT::kSpline && !WRITE_LOGW;
  extern __shared__ double s_gacc[];
  __shared__ double s_theta[GWI_MAX_THETA];
  __shared__ double s_out[GWI_MAX_THETA];
  // spline models carry few scalar sums (their gradient lives in the s_gacc rows): a narrower staging
  // area leaves the LDS to those rows
  constexpr int kNVals = 2 + ChainT::kNumAcc;
#ifdef GWI_AB_OLD_RECORD  // A/B: the scalar sums of a workgroup through a transposed LDS staging area and one row-shift reduction per value
  constexpr int kChunk = ChainT::kSpline ? (kNVals < 4 ? kNVals : 4) : kRedChunk;
  __shared__ double s_red[kChunk][kBlock];
#else
  // the scalar sums of the four waves, eight per butterfly (wave_sum8): [wave][value]
  constexpr int kSumGroups = (kNVals + 7) / 8;
  __shared__ double s_part[kWaves][kSumGroups * 8];
#endif
  __shared__ double s_wrec[kWaves][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef GWI_STAMPS
  // diagnostic build: the stamps stay in registers and are written when the wave ends (stamp 4) -- reading the row pointer
  // out of the argument block at wave entry would put a memory round trip in front of everything the stamps are there to time
  unsigned long long stamp_v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define GWI_STAMP(k)                                                                                   \
  do {                                                                                                 \
    stamp_v[k] = __builtin_amdgcn_s_memrealtime();                                                     \
    if ((k) == 4 && lane == 0) {                                                                       \
      unsigned long long* stamp_row_ = a.stamps + ((long long)blockIdx.x * kWaves + wave) * 8;         \
      for (int q_ = 0; q_ < 8; ++q_) stamp_row_[q_] = stamp_v[q_];                                     \
    }                                                                                                  \
  } while (0)
#else
#define GWI_STAMP(k) \
  do {               \
  } while (0)
#endif
  GWI_STAMP(0);
#ifdef GWI_STAMPS
  {  // where the wave runs: HW_ID (wave slot, SIMD, CU, shader array / engine) and the XCD, for the placement report of tools/stamp_phases.py
    unsigned hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    stamp_v[5] = ((unsigned long long)xcc_id << 32) | hw_id;
  }
#endif
  // geometry and the first column pointers arrive in scalar registers (ScanHead); everything else comes from the argument block,
  // whose lines are requested now and waited for only after the first trip's column loads have left
  const int h_n_norms = (int)(h_geom >> (kGeomEventBits + kGeomTilesBits)), h_tiles = (int)((h_geom >> kGeomEventBits) & ((1u << kGeomTilesBits) - 1u));
  const int h_n_ev = (int)(h_geom & ((1u << kGeomEventBits) - 1u));
  const int h_chunk_pe = unpack_chunk(h_chunks & 0xffffu), h_chunk_inj = unpack_chunk(h_chunks >> 16);
  const long long h_n_pe = hu_n_pe, h_n_inj = hu_n_inj;
  // BATCH: blockIdx.y selects the hyper-parameter point; records of point k follow those of k-1 (a single launch of the SAFE
  // instantiation has one grid row)
  const int kb = (BATCH || SAFE) ? (int)blockIdx.y : 0;
  const int n_norm_blocks = WRITE_LOGW ? 0 : h_n_norms;
  // The argument block's lines are requested by hand-written scalar loads the compiler does not track (KernargWarm): their
  // destinations are safe only on a path that reaches warm.settle().  The normaliser workgroups below return without it, so the
  // loads are issued AFTER that branch (one scalar compare later).  Round 6 found them issued ahead of it: on the normaliser
  // path the compiler was free to reuse a destination register while its load was still in flight, and after a change of the
  // register allocation the high half of the completion stamp sat in one -- a normaliser stamp then came out as
  // (low dword of some pointer) << 32 | seq now and then, and the host waited ten seconds for a stamp that never matched.
  if ((int)blockIdx.x < n_norm_blocks) {  // wave-uniform, whole workgroup
    const bool batch_n = BATCH || (SAFE && a.tblocks != nullptr);
    const int j = blockIdx.x;
    if (blockIdx.x == 0 && kb == 0 && tid == 0) *a.seq_dev = a.norm_seq;  // the tail launches stamp their results with it
    norm_block(a.norms, batch_n ? a.tblocks[kb].theta : a.theta, a.n_theta, j, a.norm_out_host + kb * h_n_norms + j, a.norm_stamps_host + kb * h_n_norms + j, a.norm_seq, s_theta, &s_wrec[0][0]);
    return;
  }
  KernargWarm<(int)sizeof...(Ks), BATCH> warm;
  warm.issue();
  const int b = (int)blockIdx.x - n_norm_blocks;
  const int n_pe_blocks = h_n_ev * h_tiles;

  // ---- this workgroup's tile and its first trip's column loads
  long long start, end, base;
  if (b < n_pe_blocks) {
    const int e = b / h_tiles;
    const int t = b - e * h_tiles;
    start = (long long)t * h_chunk_pe;
    end = start + h_chunk_pe < h_n_pe ? start + h_chunk_pe : h_n_pe;
    base = (long long)e * h_n_pe;
  } else {
    const int t = b - n_pe_blocks;
    start = (long long)t * h_chunk_inj;
    end = start + h_chunk_inj < h_n_inj ? start + h_chunk_inj : h_n_inj;
    base = 0;
  }
  const long long col_base = b < n_pe_blocks ? base : inj_offset(h_n_ev, h_n_pe);  // where this workgroup's sample set starts inside every column
  // column pointers: the first kHeadCols slots from the preloaded arguments, the rest from the argument block (the generic
  // chain indexes the block's table at run time); one pointer serves both sample sets
  const double* const head_cols[kHeadCols] = {hc0, hc1, hc2, hc3, hc4};
  const double* lcols[sizeof...(Ks)][2];
  constexpr bool kGenericCols = ChainT::kGeneric;
  if constexpr (!kGenericCols) ColFill<1, 0, Ks...>::run(lcols, head_cols, a);
  Ctx ctx;
  ctx.a = &a;
  if constexpr (kGenericCols)
    ctx.tcols = a.pe_tcols;
  else
    ctx.tcols = (const double* const (*)[2])lcols;
  const double* kappa_col = hc0;

  double m = GWI_NEG_INF, s1 = 0.0, s2 = 0.0;
  ChainT chain;
  chain.init();

  // Trip structure: lane `lane` of the workgroup handles samples i, i + 256, ... (U of them) per trip;
  // every condition on (i - lane) is wave-uniform, every condition on (i - tid) workgroup-uniform.  Loads for the
  // NEXT trip are issued before the current trip is evaluated (register double buffer).
  double kap[2][kU];
  // sample positions are 32-bit offsets from the tile's first sample (`start`): every comparison below is a 32-bit one
  const int n_tile = (int)(end - start);
  auto issue_loads = [&](int buf, int i) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int iu = i + u * kBlock;
      if (iu - lane >= n_tile) continue;  // wave-uniform: no u-th sample for this wave
      const SIdx idx{col_base + start, (unsigned)(iu < n_tile ? iu : n_tile - 1) << 3};
      kap[buf][u] = gload(kappa_col, idx);
      chain.load(buf, u, 0, ctx, idx);
    }
  };
  const int i0 = tid;
  if (i0 - lane < n_tile) issue_loads(0, i0);  // the first trip's columns: nothing before them has waited for memory
  GWI_STAMP(7);  // diagnostic build: the first trip's loads are issued

  // ---- now the argument block
  warm.settle();
  GWI_STAMP(1);  // diagnostic build: the argument block's lines have arrived
  const int h_n_theta = a.n_theta;
  const bool batch = BATCH || (SAFE && a.tblocks != nullptr);
  const double* theta_src = batch ? a.tblocks[kb].theta : a.theta;
  if (!WRITE_LOGW && n_norm_blocks == 0 && blockIdx.x == 0 && kb == 0 && tid == 0) *a.seq_dev = a.norm_seq;  // (with normaliser workgroups, the first of them does it)
  double* const logw = b < n_pe_blocks ? a.logw_pe : a.logw_inj;

  // theta -> LDS only where lane-varying indices need it (spline coefficients, normaliser grids): one value per thread
  // (n_theta <= kBlock), in flight beside the columns; the LDS writes follow further down
  static_assert(GWI_MAX_THETA <= kBlock, "one hyper-parameter per thread in the theta staging");
  double theta_mine = 0.0, theta_next[3] = {0.0, 0.0, 0.0};  // theta[tid .. tid + 3]: this thread's entries of the power-basis table (spline_poly)
  if (ChainT::kSpline) {
    const int last = h_n_theta - 1;
    theta_mine = theta_src[tid < last ? tid : last];  // unconditional (clamped index)
#pragma unroll
    for (int j = 0; j < 3; ++j) theta_next[j] = theta_src[tid + 1 + j < last ? tid + 1 + j : last];
  }
  // Replicas per coefficient: the regular kernels are built for 16 (the four rows of a sample then sit at immediate
  // offsets of one LDS address: three address adds per spline term and sample less); the SAFE instantiation takes the
  // count at run time (64 in replay mode, whatever GWI_GACC_REP asks for)
  const int rep_shift = SAFE ? a.gacc_shift : kRegularRepShift;
  const int rep = 1 << rep_shift;
  const int n_rows = h_n_theta << rep_shift;  // doubles in the shared rows
  if (kShared)
    for (int p = tid; p < n_rows; p += kBlock) s_gacc[p] = 0.0;
  for (int p = tid; p < h_n_theta; p += kBlock) s_out[p] = 0.0;
  ctx.theta = theta_src;
  ctx.derived = batch ? a.tblocks[kb].derived : a.derived;
  ctx.coefs = s_theta;
  double* const s_poly = s_gacc + (kShared ? n_rows : 0);  // behind the gradient rows in the dynamic LDS (log-weight launches keep no rows)
  ctx.poly = s_poly;
  ctx.gacc = s_gacc + (lane & (rep - 1));
  ctx.rep_shift = rep_shift;

  // shared mode: the reference exponent of this tile, in binades, workgroup-uniform and in a SCALAR register: the tile's
  // exact maximum at the previous evaluation of this (handle, point) -- a sampler moves theta by a leapfrog step between
  // two evaluations, log-weights by a few units -- or 0 where there is none yet.  Weights are exp(l) 2^-n_ref with the
  // shift applied to the exponent field (fast_exp_shift), so every sum scales EXACTLY with the choice and the record below
  // is normalised to the tile's own S1: results do not depend, to the bit, on which reference was used.  What the choice
  // must guarantee is the range: the tile's true maximum (tracked per lane, one v_max per sample) has to lie within
  // kRefSlack binades of n_ref, else sums may have over- or underflowed and the workgroup asks for a repeat (redo_host),
  // which finds the exact maximum of THIS evaluation in tile_nref already: the repeat cannot miss.
  int n_ref = 0;
  int* nref_slot = nullptr;
  double lane_max = GWI_NEG_INF;   // shared mode: this lane's largest live exponent
  // S2 holds w^2 (w^4 in a squared pass): 2 x 430 (4 x 215) binades stay inside fp64's +-1022 with room for the sum
  const int ref_slack = a.square ? 215 : 430;
  // two-pass mode (SAFE, shared only; the last resort): pass 0 sweeps the tile for its exact maximum, pass 1 is the regular loop
  const int first_pass = (SAFE && kShared && a.two_pass) ? 0 : 1;
  if (kShared) {  // behind the column loads: the v_readfirstlane waits for this load, and the columns must not wait with it
    nref_slot = a.tile_nref + ((long long)(a.nref_row0 + kb) * a.nref_stride + b);
    const int prev = *nref_slot;
    n_ref = __builtin_amdgcn_readfirstlane(prev == kNoRef ? 0 : prev);
  }
  if (ChainT::kSpline) {
    if (tid < h_n_theta) {
      s_theta[tid] = theta_mine;
      spline_poly(theta_mine, theta_next[0], theta_next[1], theta_next[2], s_poly + tid);
    }
    __syncthreads();  // also covers the zeroing of the rows and of s_out above
  }
  for (int pass_ = first_pass; pass_ < 2; ++pass_) {
    if (pass_ != first_pass && i0 - lane < n_tile) issue_loads(0, i0);
#ifdef GWI_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // diagnostic: isolate the first trip's load latency
    GWI_STAMP(2);
#endif
    // shared mode iterates whole-workgroup trips in replay mode only (a barrier sits inside); otherwise a wave stops with its samples
    const bool wg_trips = SAFE && kShared && a.deterministic;
    for (int i = i0; wg_trips ? (i - tid < n_tile) : (i - lane < n_tile); i += kU * kBlock) {
      // loop-invariant mode flags, laundered so that the compiler keeps ONE copy of the loop body instead of one per
      // combination (unswitching): the branches on them are scalar and cost nothing next to the body
      int pass = pass_, det = (SAFE && kShared) ? a.deterministic : 0;
      if (SAFE) asm volatile("" : "+s"(pass), "+s"(det));
      const bool wave_has = i - lane < n_tile;  // wave-uniform
      const int i_next = i + kU * kBlock;
      const bool has_next = i_next - lane < n_tile;  // wave-uniform
      if (has_next) issue_loads(1, i_next);
      double ell[kU], lin[kU];
      bool live[kU];
      double mx_lane = GWI_NEG_INF;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int iu = i + u * kBlock;
        if ((kShared && !wave_has) || (u > 0 && iu - lane >= n_tile)) {  // wave-uniform: this wave has no u-th sample
          live[u] = false;
          ell[u] = GWI_NEG_INF;
          lin[u] = 0.0;
          continue;
        }
        const bool valid = iu < n_tile;
        lin[u] = 1.0;
        ell[u] = kap[0][u] + chain.eval(u, 0, ctx, lin[u]);
        if constexpr (WRITE_LOGW && ChainT::kAbsorb) lin[u] = chain.finish(u, 0, ctx, lin[u], 0.0, 0);  // the log-weight wants the plain density
        // NaN or +inf weights count as zero (tests/inference_test.py:172, 260); so do zero densities
        live[u] = valid && (ell[u] < GWI_POS_INF) && (ell[u] > GWI_NEG_INF) && (lin[u] > 0.0) && (lin[u] < GWI_POS_INF);
        if (!live[u]) ell[u] = GWI_NEG_INF;
        if (WRITE_LOGW) {
          if (valid) logw[base + start + iu] = live[u] ? ell[u] + log(lin[u]) : GWI_NEG_INF;
        }
        mx_lane = fmax(mx_lane, ell[u]);
      }
      if (kShared) {
        lane_max = fmax(lane_max, mx_lane);
        if (!(SAFE && pass == 0)) {
          // the weight of sample u: L exp(l) 2^-n_ref, the exponential folded into the chain's absorbing term where it has
          // one (which is evaluated here, last); a zero, overflowing or NaN density counts as 0
          auto weight = [&](int u) -> double {
            if constexpr (ChainT::kAbsorb) {
              const double f = chain.finish(u, 0, ctx, live[u] ? lin[u] : 0.0, ell[u], n_ref);
              return (f > 0.0 && f < GWI_POS_INF) ? f : 0.0;
            } else {
              return live[u] ? lin[u] * fast_exp_shift(ell[u], n_ref) : 0.0;
            }
          };
          // replay mode (det): the waves take turns, so every row slot receives its additions in one fixed order
          // (with rep = 64 a wave instruction never has two lanes on one address)
          const int n_turns = (SAFE && det) ? kWaves : 1;
          for (int turn = 0; turn < n_turns; ++turn) {
            if (wave_has && (!det || turn == wave)) {
#pragma unroll
              for (int u = 0; u < kU; ++u) {
                if (u > 0 && i + u * kBlock - lane >= n_tile) continue;
                double w = weight(u), wfac = 1.0;
                if constexpr (ChainT::kAbsorb) wfac = w > 0.0 ? 1.0 : 0.0;  // a rejected weight takes the absorbing term's pre-weighted states with it
                if (a.square) {
                  wfac = w;
                  w *= w;
                }
                s1 += w;
                s2 += w * w;
                if (ChainT::kSpline) ctx.wt = make_weight(w);
                chain.accumulate(u, 0, ctx, w, wfac);
              }
            }
            if (SAFE && det) __syncthreads();
          }
        }
      } else if (!WRITE_LOGW) {
        // the weight of sample u against the wave's reference exponent: L exp(l - ref)
        auto weight = [&](int u, double ref) -> double {
          if constexpr (ChainT::kAbsorb) {
            const double f = chain.finish(u, 0, ctx, live[u] ? lin[u] : 0.0, ell[u] - ref, 0);
            return (f > 0.0 && f < GWI_POS_INF) ? f : 0.0;
          } else {
            return live[u] ? lin[u] * fast_exp(ell[u] - ref) : 0.0;
          }
        };
        // The wave's reference exponent m is set by the first trip that holds a live sample and moves only when a later
        // sample outruns it by more than kRefSlack (any reference within that distance is exact to rounding):
        // the other trips pay one compare + ballot instead of the 20-instruction DPP maximum.
        constexpr double kRefSlack = 150.0;
        if (m == GWI_NEG_INF || __builtin_amdgcn_ballot_w64(mx_lane > m + kRefSlack) != 0) {  // wave-uniform
          const double mx = wave_max(mx_lane);
          if (mx > m) {  // move every running sum to the new reference exponent
            if (m != GWI_NEG_INF) {  // nothing accumulated yet on the first trip
              double sc = fast_exp(m - mx);
              if (a.square) sc *= sc;
              s1 *= sc;
              s2 *= sc * sc;
              chain.rescale(sc);
            }
            m = mx;
          }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          if (u > 0 && i + u * kBlock - lane >= n_tile) continue;
          double w = weight(u, m), wfac = 1.0;
          if constexpr (ChainT::kAbsorb) wfac = w > 0.0 ? 1.0 : 0.0;  // (as above: value and gradient stay consistent)
          if (a.square) {
            wfac = w;
            w *= w;
          }
          s1 += w;
          s2 += w * w;
          if (ChainT::kSpline) ctx.wt = make_weight(w);
          chain.accumulate(u, 0, ctx, w, wfac);
        }
      }
      if (has_next) {
        chain.advance();
#pragma unroll
        for (int u = 0; u < kU; ++u) kap[0][u] = kap[1][u];
      }
    }
    if (SAFE && kShared && pass_ == 0) {  // two-pass mode: the tile's exact maximum becomes the reference
      const double mx = wave_max(lane_max);
      if (lane == 0) s_wrec[wave][2] = mx;
      __syncthreads();
      const double mm = uniform(fmax(fmax(s_wrec[0][2], s_wrec[1][2]), fmax(s_wrec[2][2], s_wrec[3][2])));
      if (mm != GWI_NEG_INF) n_ref = __builtin_amdgcn_readfirstlane((int)__builtin_rint(fmin(fmax(mm, -7.0e5), 7.0e5) * kLog2e));
      __syncthreads();  // s_wrec[.][2] is used again below
    }
  }
  if (WRITE_LOGW) return;
  GWI_STAMP(3);

#ifdef GWI_AB_OLD_RECORD
  // ---- workgroup record: common exponent M, then a transposed LDS reduction of every scalar sum
  double M = GWI_NEG_INF, f = 1.0;
  if (kShared) {
    // every wave used the tile's reference n_ref; the tile's true maximum decides below whether that was good enough
    const double mx = wave_max(lane_max);
    if (lane == 0) s_wrec[wave][2] = mx;
  } else {
    if (lane == 0) s_wrec[wave][0] = m;
    __syncthreads();
    M = s_wrec[0][0];
#pragma unroll
    for (int w_ = 1; w_ < kWaves; ++w_) M = fmax(M, s_wrec[w_][0]);
    f = (m == GWI_NEG_INF) ? 0.0 : fast_exp(m - M);  // this wave's rescale factor
    if (a.square) f *= f;
  }

  GWI_STAMP(6);  // diagnostic build: the waves' references exchanged
  constexpr int kNV = 2 + ChainT::kNumAcc;
  double vals[kNV];
  int th[kNV];
#ifdef GWI_ABL_SCATTER_TO_REG
  s1 += 1e-300 * ctx.sink;
#endif
  vals[0] = s1 * f;
  vals[1] = s2 * f * f;
  th[0] = th[1] = -1;
  chain.collect(0, ctx, vals + 2, th + 2);
#pragma unroll
  for (int v = 2; v < kNV; ++v) vals[v] *= f;
  double* out = a.partials + ((long long)kb * (n_pe_blocks + a.n_inj_tiles) + b) * a.rec_stride;
#pragma unroll
  for (int v0 = 0; v0 < kNV; v0 += kChunk) {
    if (v0 > 0) __syncthreads();
#pragma unroll
    for (int v = v0; v < kNV && v < v0 + kChunk; ++v) s_red[v - v0][tid] = vals[v];
    __syncthreads();
    // wave w reduces values v0 + w, v0 + w + 4, ...: 4 strided reads (fixed order) + one DPP sum
#pragma unroll
    for (int v = v0; v < kNV && v < v0 + kChunk; ++v) {
      if (((v - v0) & (kWaves - 1)) != wave) continue;  // wave-uniform
      const double* row = s_red[v - v0];
      const double r = wave_sum((row[lane] + row[lane + 64]) + (row[lane + 128] + row[lane + 192]));
      if (lane == 0) {
        if (v < 2) {
          if (kShared)
            s_wrec[v][3] = r;  // S1 / S2 against n_ref: normalised below, once the whole workgroup can see S1
          else
            out[1 + v] = r;
        } else {
          unsafeAtomicAdd(&s_out[th[v]], r);  // several accumulators may feed one theta slot
        }
      }
    }
  }
  if (!kShared && tid == 0) out[0] = a.square ? 2.0 * M : M;
  __syncthreads();
  GWI_STAMP(7);  // diagnostic build: scalar sums reduced
  // Shared mode: the record is normalised to the tile's own S1 -- S1 in [1, 2), everything else scaled by the same exact
  // power of two, the exponent M a whole number of binades -- so that it is, to the bit, the record any other reference
  // n_ref would have produced (see n_ref above).
  int e_norm = 0;
  if (kShared) {
    const double S1 = s_wrec[0][3], S2 = s_wrec[1][3];
    const bool has_sum = S1 > 0.0 && S1 < GWI_POS_INF;
    if (has_sum) e_norm = ilogb(S1);
    if (tid == 0) {
      const double mm = fmax(fmax(s_wrec[0][2], s_wrec[1][2]), fmax(s_wrec[2][2], s_wrec[3][2]));
      const int n_max = (mm == GWI_NEG_INF) ? kNoRef : (int)__builtin_rint(fmin(fmax(mm, -7.0e5), 7.0e5) * kLog2e);
      *nref_slot = n_max;  // the next evaluation's reference (and, if this one has to be repeated, the repeat's: exact)
      const int dist = n_max - n_ref;
      if (n_max != kNoRef && (dist > ref_slack || dist < -ref_slack)) {
        __hip_atomic_store(a.redo_host, a.norm_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.redo_dev, a.norm_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      const int n_tot = (a.square ? 2 * n_ref : n_ref) + e_norm;
      out[0] = has_sum ? (double)n_tot * kLn2 : GWI_NEG_INF;
      out[1] = has_sum ? ldexp(S1, -e_norm) : 0.0;
      out[2] = has_sum ? ldexp(S2, -2 * e_norm) : 0.0;
    }
  }
  // gradient numerators: scalar sums from s_out, spline-coefficient sums from the shared rows (replicas in fixed order)
  for (int p = tid; p < a.n_theta; p += kBlock) {
    double g = s_out[p];
    if (kShared) {
      const double* rows = s_gacc + ((long)p << rep_shift);
      double gw = 0.0;
      for (int r = 0; r < rep; ++r) gw += rows[(r + p) & (rep - 1)];  // rotated start: the threads of a wave read different banks
      g = ldexp(g + gw, -e_norm);
    }
    out[kRecHeader + p] = g;
  }
#else
  // ---- workgroup record.  Every wave sums its lanes' scalar sums eight at a time (wave_sum8: no LDS, no barrier) and
  // leaves them in s_part[wave][.]; ONE barrier later they are added across the waves in wave order.  (Before: a transposed
  // LDS staging area, three barriers and one row-shift reduction per value -- 1.2 us of config 2's 4.8 us wave lifetime.)
  constexpr int kNV = kNVals;
  double vals[kSumGroups * 8];
  int th[kNV];
#ifdef GWI_ABL_SCATTER_TO_REG
  s1 += 1e-300 * ctx.sink;
#endif
  vals[0] = s1;
  vals[1] = s2;
  th[0] = th[1] = -1;
  chain.collect(0, ctx, vals + 2, th + 2);
#pragma unroll
  for (int v = kNV; v < kSumGroups * 8; ++v) vals[v] = 0.0;
#pragma unroll
  for (int g = 0; g < kSumGroups; ++g) {
    const double z = wave_sum8(vals + 8 * g);
    if ((lane & 7) == 0) s_part[wave][8 * g + (lane >> 3)] = z;
  }
  if (kShared) {
    // every wave used the tile's reference n_ref; the tile's true maximum decides below whether that was good enough
    const double mx = wave_max(lane_max);
    if (lane == 0) s_wrec[wave][2] = mx;
  } else {
    if (lane == 0) s_wrec[wave][0] = m;  // the wave's own reference exponent: its sums are relative to it
  }
  double* out = a.partials + ((long long)kb * (n_pe_blocks + a.n_inj_tiles) + b) * a.rec_stride;
  __syncthreads();  // shared mode: also every wave's additions to the gradient rows
  GWI_STAMP(6);     // diagnostic build: the waves' sums are in place
  if (!kShared) {
    // Parametric models: wave 0 finishes alone -- lane v (< kNV) moves value v of the four waves to the common exponent
    // M = max of their references and adds them in wave order; the other waves are done.
    if (wave != 0) {
      GWI_STAMP(4);
      return;
    }
    // lane l holds the reference of wave l & 3: ONE exponential gives the four rescale factors (a quad holds all of them),
    // quad broadcasts hand every lane all four -- one dependent chain where a loop over the waves would run four
    static_assert(kWaves == 4, "the quad broadcasts below assume four waves");
    const double my_m = s_wrec[lane & 3][0];
    const int vi = lane < kSumGroups * 8 ? lane : 0;
    double part[kWaves];
#pragma unroll
    for (int w_ = 0; w_ < kWaves; ++w_) part[w_] = s_part[w_][vi];
    double M = fmax(my_m, dpp_take<0xB1, 0xf>(my_m));  // quad_perm [1,0,3,2]
    M = fmax(M, dpp_take<0x4E, 0xf>(M));               // quad_perm [2,3,0,1]: the maximum of the four references, in every lane
    double f_mine = (my_m == GWI_NEG_INF) ? 0.0 : fast_exp(my_m - M);  // wave (l & 3)'s rescale factor
    if (a.square) f_mine *= f_mine;
    const double f0 = dpp_take<0x00, 0xf>(f_mine), f1 = dpp_take<0x55, 0xf>(f_mine), f2 = dpp_take<0xAA, 0xf>(f_mine), f3 = dpp_take<0xFF, 0xf>(f_mine);
    const double t1 = fma(part[3], f3, fma(part[2], f2, fma(part[1], f1, part[0] * f0)));                      // sums of w: in wave order
    const double t2 = fma(part[3], f3 * f3, fma(part[2], f2 * f2, fma(part[1], f1 * f1, part[0] * (f0 * f0))));  // S2 holds w^2
    const double tot = lane == 1 ? t2 : t1;
    if (lane == 0) out[0] = a.square ? 2.0 * M : M;
    if (lane < 2) out[1 + lane] = tot;
    // gradient numerators: lane v adds its total to the theta slot of accumulator v (several accumulators may feed one slot),
    // then the slots are read out -- LDS operations of ONE wave, executed in program order
    int slot = 0;
#pragma unroll
    for (int v = 2; v < kNV; ++v) slot = (lane == v) ? th[v] : slot;
    if (lane >= 2 && lane < kNV) unsafeAtomicAdd(&s_out[slot], tot);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (int p = lane; p < a.n_theta; p += 64) out[kRecHeader + p] = s_out[p];
    GWI_STAMP(4);
    return;
  }
  // Shared mode: the record is normalised to the tile's own S1 -- S1 in [1, 2), everything else scaled by the same exact
  // power of two, the exponent M a whole number of binades -- so that it is, to the bit, the record any other reference
  // n_ref would have produced (see n_ref above).
  auto total = [&](int v) { return (s_part[0][v] + s_part[1][v]) + (s_part[2][v] + s_part[3][v]); };
  const double S1 = total(0), S2 = total(1);
  const bool has_sum = S1 > 0.0 && S1 < GWI_POS_INF;
  const int e_norm = has_sum ? ilogb(S1) : 0;
  if (tid == 0) {
    const double mm = fmax(fmax(s_wrec[0][2], s_wrec[1][2]), fmax(s_wrec[2][2], s_wrec[3][2]));
    const int n_max = (mm == GWI_NEG_INF) ? kNoRef : (int)__builtin_rint(fmin(fmax(mm, -7.0e5), 7.0e5) * kLog2e);
    *nref_slot = n_max;  // the next evaluation's reference (and, if this one has to be repeated, the repeat's: exact)
    const int dist = n_max - n_ref;
    if (n_max != kNoRef && (dist > ref_slack || dist < -ref_slack)) {
      __hip_atomic_store(a.redo_host, a.norm_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(a.redo_dev, a.norm_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int n_tot = (a.square ? 2 * n_ref : n_ref) + e_norm;
    out[0] = has_sum ? (double)n_tot * kLn2 : GWI_NEG_INF;
    out[1] = has_sum ? ldexp(S1, -e_norm) : 0.0;
    out[2] = has_sum ? ldexp(S2, -2 * e_norm) : 0.0;
  }
  // gradient numerators: spline coefficients from the shared rows (replicas in fixed order), scalar parameters from the
  // waves' sums of the accumulators that feed them (few in a spline model: a compare per accumulator)
  for (int p = tid; p < a.n_theta; p += kBlock) {
    double g = 0.0;
#pragma unroll
    for (int v = 2; v < kNV; ++v)
      if (th[v] == p) g += total(v);
    const double* rows = s_gacc + ((long)p << rep_shift);
    double gw = 0.0;
    for (int r = 0; r < rep; ++r) gw += rows[(r + p) & (rep - 1)];  // rotated start: the threads of a wave read different banks
    out[kRecHeader + p] = ldexp(g + gw, -e_norm);
  }
#endif
  GWI_STAMP(4);
}

// ---- batched launches of PARAMETRIC chains: every sample is loaded once for all the points of the batch ---------------
// (Opt-in since round 6 -- GWI_PBATCH=1 or a row size: which of this kernel and the one-grid-row-per-point scan is faster
// depends on the box, within 10 %, and the default is the latter; gwi_engine.hip where h->pbatch is set.)
// scan_kernel<.., BATCH = true, ..> runs one grid row per hyper-parameter point: K points stream the catalog K times
// (through L2 / the Infinity Cache) and redo K times whatever a sample needs that does not depend on theta (PL+Peak:
// m1 = exp(log m1), a quarter of its exponentials; log(mmin / m1); the address arithmetic, the prologue, the argument
// block).  Here a workgroup owns ONE single-trip tile (<= U x 256 samples): its lanes load their samples' columns once,
// keep them in registers and loop over the `pbatch_pts` points of their grid row (blockIdx.y; KArgs::pbatch_pts of the
// k_batch points -- the host splits a batch over rows only as far as load balance needs: a tile x 16 points is a long
// workgroup, and 788 of them on 256 CUs leave a quarter of the chip idle at the end).  The loop over the points is a real
// loop, so the compiler hoists the theta-independent arithmetic out of it (seen in the disassembly of config 2's chain:
// three exponentials per sample and point inside the loop, m1 = exp(log m1) and the alpha = -1 branch's reciprocals ahead of it).  Per point a wave evaluates its samples against its
// own exact maximum (single trip: no running reference, no rescaling), sums eight values per butterfly (wave_sum8) and
// parks them in LDS; after ONE barrier wave w completes the records of points w, w + 4, ... exactly as scan_kernel's wave 0
// does for one.  Records, combine and final launches are those of the batched scan (same layout, same bits per point up to
// the order in which the four waves' sums are added, which is the same too).
// Hyper-parameters come through scalar loads from the point's ThetaBlock in device memory (wave-uniform addresses).
constexpr int kPbatchMaxPts = 16;  // points per workgroup (rows of the LDS staging area)
// samples per lane of the pbatch kernel that goes with a chain of U samples per lane
#ifndef GWI_PBATCH_U
#define GWI_PBATCH_U 0
#endif
constexpr int pbatch_u(int chain_u) { return GWI_PBATCH_U ? GWI_PBATCH_U : chain_u; }
template <int U, int... Ks>
__global__ __launch_bounds__(kBlock, GWI_SCAN_WAVES_PER_EU) void scan_pbatch_kernel(const double* hc0, const double* hc1, const double* hc2, const double* hc3, const double* hc4, const unsigned h_geom,
                                                                                    const unsigned h_chunks, const unsigned hu_n_pe, const unsigned hu_n_inj, const KArgs a) {
  using ChainT = Chain<U, Ks...>;
  static_assert(!ChainT::kSpline && !ChainT::kGeneric, "parametric chains only: spline models batch on gwi_mfma.h / the 4-tap kernel");
  constexpr int kU = U;
  constexpr int kNV = 2 + ChainT::kNumAcc;
  constexpr int kSumGroups = (kNV + 7) / 8;
  __shared__ double s_part[kWaves][kPbatchMaxPts][kSumGroups * 8];
  __shared__ double s_m[kWaves][kPbatchMaxPts];
  __shared__ double s_out[kWaves][GWI_MAX_THETA];  // per wave: theta slots of the record being completed
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h_n_norms = (int)(h_geom >> (kGeomEventBits + kGeomTilesBits)), h_tiles = (int)((h_geom >> kGeomEventBits) & ((1u << kGeomTilesBits) - 1u));
  const int h_n_ev = (int)(h_geom & ((1u << kGeomEventBits) - 1u));
  const int h_chunk_pe = unpack_chunk(h_chunks & 0xffffu), h_chunk_inj = unpack_chunk(h_chunks >> 16);
  const long long h_n_pe = hu_n_pe, h_n_inj = hu_n_inj;
  const int K = a.k_batch;
  // the first n_norms x K workgroups of grid row 0 integrate the normaliser grids, one (normaliser, point) each
  const int n_norm_blocks = h_n_norms * K;
  if ((int)blockIdx.x < n_norm_blocks) {  // wave-uniform, whole workgroup
    if (blockIdx.y != 0) return;
    const int kb = (int)blockIdx.x / h_n_norms, j = (int)blockIdx.x - kb * h_n_norms;
    if (blockIdx.x == 0 && tid == 0) *a.seq_dev = a.norm_seq;  // the tail launches stamp their results with it
    norm_block(a.norms, a.tblocks[kb].theta, a.n_theta, j, a.norm_out_host + kb * h_n_norms + j, a.norm_stamps_host + kb * h_n_norms + j, a.norm_seq, &s_out[0][0], &s_m[0][0]);
    return;
  }
  const int g_wg = (int)blockIdx.x - n_norm_blocks;
  const int n_pe_blocks = h_n_ev * h_tiles;
  const long long n_blocks_total = (long long)n_pe_blocks + a.n_inj_tiles;
  // ---- which (tile, point) UNITS this workgroup evaluates, in tile-major order u = tile x K + point.
  // rows mode (pbatch_pts > 0): tile blockIdx.x, the pbatch_pts points of grid row blockIdx.y.  Balanced mode (pbatch_pts = -G < 0,
  // one grid row of G scan workgroups): the launch has as many workgroups as the chip holds at once (or the next smaller count that gives them equal
  // shares) and workgroup g takes the g-th of G equal runs of the unit sequence -- a run crosses a tile boundary once or twice,
  // the tile is loaded when it changes.  No workgroup is left to run alone at the end of the launch, which at config 2's size
  // (788 tiles x 16 points on 1280 resident workgroups) was a third of the scan's time.
  unsigned u_next, u_end;
  if (a.pbatch_pts > 0) {
    const int pts = a.pbatch_pts < kPbatchMaxPts ? a.pbatch_pts : kPbatchMaxPts;
    const int k_first = (int)blockIdx.y * pts;
    u_next = (unsigned)g_wg * (unsigned)K + (unsigned)k_first;
    u_end = u_next + (unsigned)(K - k_first < pts ? K - k_first : pts);
  } else {
    // (the workgroup count comes with the arguments: gridDim would be an implicit kernel argument, which the AQL queue does not supply)
    const unsigned n_units = (unsigned)n_blocks_total * (unsigned)K, n_wg = (unsigned)(-a.pbatch_pts);
    const unsigned q = n_units / n_wg, r = n_units - q * n_wg, g = (unsigned)g_wg;
    u_next = g * q + (g < r ? g : r);
    u_end = u_next + q + (g < r ? 1u : 0u);
  }
  const double* const head_cols[kHeadCols] = {hc0, hc1, hc2, hc3, hc4};
  const double* lcols[sizeof...(Ks)][2];
  ColFill<1, 0, Ks...>::run(lcols, head_cols, a);
  Ctx ctx;
  ctx.a = &a;
  ctx.tcols = (const double* const (*)[2])lcols;
  ctx.coefs = nullptr;
  ctx.poly = nullptr;
  ctx.gacc = nullptr;
  ctx.rep_shift = 0;
  const double* kappa_col = hc0;
  ChainT chain;
  double kap[kU];
  if (n_norm_blocks == 0 && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) *a.seq_dev = a.norm_seq;
  for (int p = lane; p < a.n_theta; p += 64) s_out[wave][p] = 0.0;  // wave-private: no barrier needed before its own use
  // theta slot of every scalar sum (the same for every point)
  int th[kNV];
  {
    double unused[kSumGroups * 8];
    th[0] = th[1] = -1;
    chain.init();
    chain.collect(0, ctx, unused + 2, th + 2);
  }
  int b = -1, n_tile = 0;
  while (u_next < u_end) {  // one pass per SEGMENT: up to kPbatchMaxPts consecutive points of one tile (wave-uniform throughout)
    const int b_seg = (int)(u_next / (unsigned)K);
    const int k0 = (int)(u_next - (unsigned)b_seg * (unsigned)K);
    int n_k = K - k0;
    if ((unsigned)n_k > u_end - u_next) n_k = (int)(u_end - u_next);
    if (n_k > kPbatchMaxPts) n_k = kPbatchMaxPts;
    u_next += (unsigned)n_k;
    if (b_seg != b) {
      // ---- the tile's samples: ONE trip (the host sizes the tiles of this launch to <= kU x 256 samples), resident in registers
      b = b_seg;
      long long start, end, col_base;
      if (b < n_pe_blocks) {
        const int e = b / h_tiles;
        const int t = b - e * h_tiles;
        start = (long long)t * h_chunk_pe;
        end = start + h_chunk_pe < h_n_pe ? start + h_chunk_pe : h_n_pe;
        col_base = (long long)e * h_n_pe;
      } else {
        const int t = b - n_pe_blocks;
        start = (long long)t * h_chunk_inj;
        end = start + h_chunk_inj < h_n_inj ? start + h_chunk_inj : h_n_inj;
        col_base = inj_offset(h_n_ev, h_n_pe);
      }
      n_tile = (int)(end - start);
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int iu = tid + u * kBlock;
        if (iu - lane >= n_tile) continue;  // wave-uniform: no u-th sample for this wave
        const SIdx idx{col_base + start, (unsigned)(iu < n_tile ? iu : n_tile - 1) << 3};
        kap[u] = gload(kappa_col, idx);
        chain.load(0, u, 0, ctx, idx);
      }
    }
    for (int kk = 0; kk < n_k; ++kk) {
      const ThetaBlock* tb = a.tblocks + (k0 + kk);
      ctx.theta = tb->theta;
      ctx.derived = tb->derived;
      chain.init();
      double ell[kU], lin[kU];
      bool live[kU];
      double mx_lane = GWI_NEG_INF;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        const int iu = tid + u * kBlock;
        if (iu - lane >= n_tile) {  // wave-uniform: this wave has no u-th sample
          live[u] = false;
          ell[u] = GWI_NEG_INF;
          lin[u] = 0.0;
          continue;
        }
        lin[u] = 1.0;
        ell[u] = kap[u] + chain.eval(u, 0, ctx, lin[u]);
        // NaN or +inf weights count as zero (tests/inference_test.py:172, 260); so do zero densities
        live[u] = (iu < n_tile) && (ell[u] < GWI_POS_INF) && (ell[u] > GWI_NEG_INF) && (lin[u] > 0.0) && (lin[u] < GWI_POS_INF);
        if (!live[u]) ell[u] = GWI_NEG_INF;
        mx_lane = fmax(mx_lane, ell[u]);
      }
      // the wave's maximum for this point (to single precision, wave_max_coarse): every weight is <= (1 + 1e-7 |m|) x its linear part
#ifdef GWI_AB_PBATCH_NOMAX  // timing-only ablation: no cross-lane maximum (wrong reference)
      const double m = uniform(mx_lane);
#elif defined(GWI_AB_PBATCH_EXACT_MAX)  // A/B: round 5's double-precision maximum
      const double m = wave_max(mx_lane);
#else
      const double m = wave_max_coarse(mx_lane);  // within 6e-8 |m| of the wave's maximum: a reference, not a result
#endif
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (tid + u * kBlock - lane >= n_tile) continue;
        double w, wfac = 1.0;
        if constexpr (ChainT::kAbsorb) {
          const double f = chain.finish(u, 0, ctx, live[u] ? lin[u] : 0.0, ell[u] - m, 0);
          w = (f > 0.0 && f < GWI_POS_INF) ? f : 0.0;
          wfac = w > 0.0 ? 1.0 : 0.0;  // a rejected weight takes the absorbing term's pre-weighted states with it
        } else {
          w = live[u] ? lin[u] * fast_exp(ell[u] - m) : 0.0;
        }
        if (a.square) {
          wfac = w;
          w *= w;
        }
        s1 += w;
        s2 += w * w;
        chain.accumulate(u, 0, ctx, w, wfac);
      }
      double vals[kSumGroups * 8];
      int th_unused[kNV];
      vals[0] = s1;
      vals[1] = s2;
      chain.collect(0, ctx, vals + 2, th_unused + 2);
#pragma unroll
      for (int v = kNV; v < kSumGroups * 8; ++v) vals[v] = 0.0;
#pragma unroll
      for (int g = 0; g < kSumGroups; ++g) {
#ifdef GWI_AB_PBATCH_NOSUM  // timing-only ablation: no butterfly (every value still reaches LDS once: eight stores of 64 lanes)
#pragma unroll
        for (int v = 0; v < 8; ++v)
          if ((lane & 7) == v) s_part[wave][kk][8 * g + (lane >> 3)] = vals[8 * g + v];
#else
        const double z = wave_sum8(vals + 8 * g);
        if ((lane & 7) == 0) s_part[wave][kk][8 * g + (lane >> 3)] = z;
#endif
      }
      if (lane == 0) s_m[wave][kk] = m;
    }
    __syncthreads();
    // ---- records: wave w completes points w, w + 4, ... of this segment (scan_kernel's parametric epilogue, per point)
    static_assert(kWaves == 4, "the quad broadcasts below assume four waves");
    int slot = 0;
#pragma unroll
    for (int v = 2; v < kNV; ++v) slot = (lane == v) ? th[v] : slot;
    const int vi = lane < kSumGroups * 8 ? lane : 0;
    for (int kk = wave; kk < n_k; kk += kWaves) {
      const double my_m = s_m[lane & 3][kk];
      double part[kWaves];
#pragma unroll
      for (int w_ = 0; w_ < kWaves; ++w_) part[w_] = s_part[w_][kk][vi];
      double M = fmax(my_m, dpp_take<0xB1, 0xf>(my_m));  // quad_perm [1,0,3,2]
      M = fmax(M, dpp_take<0x4E, 0xf>(M));               // quad_perm [2,3,0,1]: the maximum of the four references, in every lane
      double f_mine = (my_m == GWI_NEG_INF) ? 0.0 : fast_exp(my_m - M);
      if (a.square) f_mine *= f_mine;
      const double f0 = dpp_take<0x00, 0xf>(f_mine), f1 = dpp_take<0x55, 0xf>(f_mine), f2 = dpp_take<0xAA, 0xf>(f_mine), f3 = dpp_take<0xFF, 0xf>(f_mine);
      const double t1 = fma(part[3], f3, fma(part[2], f2, fma(part[1], f1, part[0] * f0)));                      // sums of w: in wave order
      const double t2 = fma(part[3], f3 * f3, fma(part[2], f2 * f2, fma(part[1], f1 * f1, part[0] * (f0 * f0))));  // S2 holds w^2
      const double tot = lane == 1 ? t2 : t1;
      double* out = a.partials + ((long long)(k0 + kk) * n_blocks_total + b) * a.rec_stride;
      if (lane == 0) out[0] = a.square ? 2.0 * M : M;
      if (lane < 2) out[1 + lane] = tot;
      if (lane >= 2 && lane < kNV) unsafeAtomicAdd(&s_out[wave][slot], tot);  // several accumulators may feed one theta slot
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      for (int p = lane; p < a.n_theta; p += 64) {
        out[kRecHeader + p] = s_out[wave][p];
        s_out[wave][p] = 0.0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    }
    if (u_next < u_end) __syncthreads();  // the staging rows are reused by the next segment
  }  // segments
}

// ---- the tail as separate launches (GWI_FUSED_TAIL=0, and the reference point for A/B timing) -------
__global__ __launch_bounds__(kBlock) void combine_kernel(const TailArgs a) { combine_group(a, blockIdx.x, blockIdx.y, threadIdx.x); }

constexpr int kFinalThreads = 1024;
__global__ __launch_bounds__(kFinalThreads) void final_kernel(const TailArgs a) {
  __shared__ double s_tile[kFinalThreads];
  final_reduce<kFinalThreads>(a, blockIdx.y, blockIdx.x, threadIdx.x, s_tile);
}

// ---- batched launches: pull the K theta blocks from pinned host memory into device memory (one
//      workgroup per block, 16-byte loads over PCIe), in stream order ahead of the scan.  A
//      hipMemcpyAsync of the same 2 KiB x K costs more in copy-engine start-up than the whole scan. ----
__global__ __launch_bounds__(kBlock) void stage_theta_kernel(const ThetaBlock* host_src, ThetaBlock* dst) {
  static_assert(sizeof(ThetaBlock) % 16 == 0, "ThetaBlock is copied in 16-byte pieces");
  const double2* s = reinterpret_cast<const double2*>(host_src + blockIdx.x);
  double2* d = reinterpret_cast<double2*>(dst + blockIdx.x);
  for (int i = threadIdx.x; i < (int)(sizeof(ThetaBlock) / 16); i += kBlock) d[i] = s[i];
}

// ---- after the all-gather: copy the gathered records to pinned host memory and stamp completion -----
__global__ __launch_bounds__(kBlock) void publish_kernel(const double* gathered, double* host, int n, unsigned long long seq) {
  for (int i = threadIdx.x + 1; i < n; i += kBlock) store_sys(host + i, gathered[i]);
  publish_stamp(host, seq, threadIdx.x);
}

// ---- measured HBM bandwidth (gwi_hbm_bandwidth): a read-only sweep and a STREAM triad, 16-byte accesses, grid-stride ----
__global__ __launch_bounds__(kBlock) void bw_read_kernel(const double2* __restrict__ a, long long n2, double* out, int n_blocks /* = gridDim.x, passed explicitly: no implicit kernel arguments in this code object */) {
  // four independent 16-byte NON-TEMPORAL loads in flight per lane, 32 workgroups per CU: the best of the variants in
  // tools/microbench/hbm_read.hip (6.4 TB/s on the round-2 boxes; 5.7 with cached loads, 5.5 with 4 workgroups per CU)
  double s0 = 0.0, s1 = 0.0;
  const long long stride = (long long)n_blocks * kBlock;
  long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
  for (; i + 3 * stride < n2; i += 4 * stride) {
    double vx[4], vy[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      vx[u] = __builtin_nontemporal_load(&a[i + u * stride].x);
      vy[u] = __builtin_nontemporal_load(&a[i + u * stride].y);
    }
    s0 += (vx[0] + vx[1]) + (vx[2] + vx[3]);
    s1 += (vy[0] + vy[1]) + (vy[2] + vy[3]);
  }
  for (; i < n2; i += stride) {
    const double2 v = a[i];
    s0 += v.x;
    s1 += v.y;
  }
  const double w = wave_sum(s0 + s1);
  if ((threadIdx.x & 63) == 0) unsafeAtomicAdd(out + (blockIdx.x & 63), w);
}
__global__ __launch_bounds__(kBlock) void bw_triad_kernel(double2* __restrict__ a, const double2* __restrict__ b, const double2* __restrict__ c, double s, long long n2, int n_blocks) {
  const long long stride = (long long)n_blocks * kBlock;
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n2; i += stride) {
    const double2 vb = b[i], vc = c[i];
    a[i] = make_double2(vb.x + s * vc.x, vb.y + s * vc.y);
  }
}

}  // namespace gwi
