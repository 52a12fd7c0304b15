// gwi_mfma.h -- batched scan with the spline-coefficient gradient as an fp64 MFMA GEMM (gfx950 / CDNA4 only).
//
// Reference operation: the dense B-spline design-matrix contraction of the reference, forward
// `einsum("i...,i->...", design_matrix, coefs)` (gwinferno/interpolation.py:304, :393 on the matrices built at
// models/bsplines/single.py:56-57) and its transpose in reverse mode.  For ONE hyper-parameter point that contraction is
// a matrix-vector product with four non-zeros per column, which scan_kernel does with four taps per sample and LDS
// atomics.  For K = 16 hyper-parameter points per launch (vectorised chains, gwi_eval_batch) the reverse-mode half is a
// real GEMM,
//        G[basis, point] = sum_samples  B_basis(x_sample) * w[sample, point],
// and this kernel runs it on the matrix cores: v_mfma_f64_16x16x4_f64 with A = a 16-basis x 4-sample slab of the design
// matrix formed IN REGISTERS from the sample's four taps (never stored anywhere), B = the 4-sample x 16-point block of
// importance weights, D = the 16 x 16 gradient tile, which stays in registers for the whole tile of the catalog.  No LDS
// atomics, a fixed summation order (bit-reproducible), and the matrix pipe runs beside the vector pipe.
//
// Two phases per trip of a wavefront (64 samples), so that the memory traffic and the knot lookup are paid once per sample
// and not once per (sample, point):
//   phase A   lane = sample: column loads (coalesced, register-prefetched one trip ahead), knot interval k and fraction t of
//             every spline term, outside-domain flag -> a wavefront-private staging row in LDS (per spline term: k | flag, t;
//             per other term: its one or two column values; kappa first)
//   phase B   lane = (q = sample slot 0..3, j = point 0..15), sixteen rounds: read the sample's staged row (broadcast reads:
//             16 bytes per spline term), the four taps from t, dot with point j's coefficients (LDS), the parametric terms
//             with point j's scalars, exponential, running sums; then per 16-basis tile one MFMA: A = the lane's tap for
//             basis 16 tile + j (select on basis - k), B = the lane's weight.
// The staging carries (k, t) and not the taps: the kernel is bound by the LDS pipe the four SIMDs of a CU share (every
// wave instruction moves 64 lanes' worth of bytes, broadcast or not), and the coefficient gather alone is 32 bytes per lane,
// term and sample-point; recomputing the taps costs 11 vector instructions per term where reading them costs two more
// 16-byte reads (measured: taps staged 44.3 us per evaluation at config 5, K = 16).
// Operand / result layout of v_mfma_f64_16x16x4_f64 (confirmed on the hardware by tools/microbench/mfma_f64_layout.hip):
//   A  lane l holds A[i = l % 16][k = l / 16];  B  lane l holds B[k = l / 16][j = l % 16];
//   D  lane l, register r holds D[row = l / 16 + 4 r][col = l % 16].
// Reference exponents: one per (tile, point), the tile's exact maximum at the previous batched launch of the handle
// (KArgs::tile_nref rows 1..K), applied as an exact power of two; records normalised per point exactly as scan_kernel's.
#pragma once
#include "gwi_device.h"

namespace gwi {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kPts = 16;      // hyper-parameter points per wavefront (the N of the MFMA tile)
// build knobs of the matrix-core kernel (A/B: profiles/round4/EXPERIMENTS.md section 7)
#ifndef GWI_MFMA_STAGE
#define GWI_MFMA_STAGE 32     // samples per wavefront and trip in the staging rows
#endif
#ifndef GWI_MFMA_FWD_GROUP
#define GWI_MFMA_FWD_GROUP 4  // spline terms whose staged words and coefficients are in flight together in the forward half
#endif
#ifndef GWI_MFMA_WAVES
#define GWI_MFMA_WAVES 2      // resident waves per SIMD the register allocation aims at
#endif

// LDS reads of phase B go through a volatile pointer: left alone, the compiler pairs neighbouring 8-byte reads into
// ds_read2_b64, which the LDS serves as two 4 x 16-lane accesses (8 cycles per instruction = half the rate of two ds_read_b64
// at 2.3 each; MI355X_MICROARCH.md, LDS table), and this kernel's round is bound by the LDS pipe and the issue port in turn
// (profiles/round3/EXPERIMENTS.md section 6): 29 ds_read2_b64 + 9 ds_read_b64 per round of config 5 -> 67 ds_read_b64.
typedef const volatile double __attribute__((address_space(3))) * lds_ro;  // LDS address space spelled out: a volatile generic pointer would be read with flat loads

// the tap of basis function `first + d` for a sample whose non-zero bases start at `first`: b_d for d in 0..3, else 0
__device__ __forceinline__ double tap_select(const Taps& b, int d) {
  double v = 0.0;
  v = d == 0 ? b.b0 : v;
  v = d == 1 ? b.b1 : v;
  v = d == 2 ? b.b2 : v;
  v = d == 3 ? b.b3 : v;
  return v;
}

// ---- per-kind staging.  kDoubles = doubles a sample occupies in the staging row for this term.  Phase B runs in three
//      passes over the chain -- b_rows (every term's staged words requested), b_coefs (every spline term's four
//      coefficients requested, which needs k), b_value (the arithmetic) -- so that a round of the loop waits for the LDS
//      twice, not once or twice per term: with two resident waves per SIMD (the gradient tiles take 64 registers) there is
//      nothing to hide twenty LDS round trips per round behind (measured: 21 s_waitcnt per round, 2770 cycles per
//      64 sample-points whatever the vector instruction count).
template <int K>
struct Stage {  // every kind without spline coefficients: the term's column values travel as they are
  static constexpr bool kSpline = false;
  static constexpr int kDoubles = (int)(sizeof(typename Term<K>::In) / sizeof(double));
  struct Keep {
    typename Term<K>::In in;
    typename Term<K>::State st;      // of the round being evaluated (stage 1)
    typename Term<K>::State st_cur;  // of the round being accumulated (stage 2)
    __device__ void commit() { st_cur = st; }
  };
  __device__ static void phase_a(const TermD&, const typename Term<K>::In& in, double* row) {
    const double* src = reinterpret_cast<const double*>(&in);
#pragma unroll
    for (int i = 0; i < kDoubles; ++i) row[i] = src[i];
  }
  __device__ static void b_rows(const double* row_, Keep& kp) {
    double* dst = reinterpret_cast<double*>(&kp.in);
#ifdef GWI_AB_MFMA_READ2
    const double* row = row_;
#else
    lds_ro row = (lds_ro)row_;
#endif
#pragma unroll
    for (int i = 0; i < kDoubles; ++i) dst[i] = row[i];
  }
  __device__ static void b_coefs(const TermD&, const Ctx&, Keep&) {}
  __device__ static double b_value(const TermD& t, const double* d, const Ctx& c, Keep& kp, double& lin) { return Term<K>::eval(t, d, c, kp.in, kp.st, lin); }
};
struct SplineKeep {
  Taps b;        // (forward passes only: dead after b_value; accumulate gathers its operand from the staging row)
  double cf[4];  // (forward passes only)
  int k;
  double scale;  // what multiplies the weight in the B operand (1, or 1 / f for the linear spline)
  int k_cur;     // the same two of the round being accumulated (stage 2 of the software pipeline in scan_points_body)
  double scale_cur;
  __device__ void commit() {
    k_cur = k;
    scale_cur = scale;
  }
};
// staged per spline term: [0] = k, [1] = 0, [2..5] = the four taps (all 0 outside a zero-outside basis), [6] = 0.
// The zeros on both sides let the matrix-core path GATHER its A operand: the tap of basis 16 tile + j for a sample whose
// non-zero bases start at k is  row[2 + clamp(16 tile + j - k, -1, 4)]  -- one v_med3, one address add and one 8-byte LDS read
// per tile instead of four compares and eight selects on registers (96 of ~400 vector quad-cycles per 64 sample-points at
// config 5; the LDS pipe has the room: 19 % busy).
constexpr int kSplineStage = 7;
__device__ __forceinline__ void stage_spline(const TermD& t, double x, bool zero_outside, double* row) {
  int k;
  double tt;
  spline_locate_term(x, t, k, tt);  // x: the knot coordinate the engine keeps for spline terms (gwi_device.h: spline_locate_knot)
  Taps b = cubic_taps(tt);
  if (zero_outside && spline_outside(x, t)) b.b0 = b.b1 = b.b2 = b.b3 = 0.0;  // bases are 0 out there (interpolation.py:175)
  row[0] = __hiloint2double(0, k);
  row[1] = 0.0;
  row[2] = b.b0;
  row[3] = b.b1;
  row[4] = b.b2;
  row[5] = b.b3;
  row[6] = 0.0;
}
__device__ __forceinline__ void spline_rows(const double* row_, SplineKeep& kp) {
#ifdef GWI_AB_MFMA_READ2
  const double* row = row_;
#else
  lds_ro row = (lds_ro)row_;
#endif
  kp.k = __double2loint(row[0]);
#ifdef GWI_ABL_NO_TAPREAD  // timing-only ablation: the taps not read from the staging row
  kp.b.b0 = kp.b.b1 = kp.b.b2 = kp.b.b3 = 0.25;
#else
  kp.b.b0 = row[2];
  kp.b.b1 = row[3];
  kp.b.b2 = row[4];
  kp.b.b3 = row[5];
#endif
  kp.scale = 1.0;
}
__device__ __forceinline__ void spline_coefs(const TermD& t, const Ctx& c, SplineKeep& kp) {
#ifdef GWI_ABL_NO_COEF  // timing-only ablation: no coefficient gather
#pragma unroll
  for (int i = 0; i < 4; ++i) kp.cf[i] = 1.0 + kp.k;
#else
#ifdef GWI_AB_MFMA_READ2
  const double* cf = c.coefs + t.th0 + kp.k;
#else
  lds_ro cf = (lds_ro)(c.coefs + t.th0 + kp.k);
#endif
#pragma unroll
  for (int i = 0; i < 4; ++i) kp.cf[i] = cf[i];
#endif
}
__device__ __forceinline__ double spline_value(SplineKeep& kp) {
  return kp.cf[0] * kp.b.b0 + kp.cf[1] * kp.b.b1 + kp.cf[2] * kp.b.b2 + kp.cf[3] * kp.b.b3;
}
template <>
struct Stage<GWI_TERM_EXP_SPLINE> {
  static constexpr bool kSpline = true;
  static constexpr int kDoubles = kSplineStage;
  using Keep = SplineKeep;
  __device__ static void phase_a(const TermD& t, const typename Term<GWI_TERM_EXP_SPLINE>::In& in, double* row) {
    // LogY bases exclude outside samples through kappa (decided when the catalog was bound); zero-outside bases give exp(0)
    stage_spline(t, in.x0, (t.flags & GWI_SPLINE_OUTSIDE_ZERO_EXPONENT) != 0, row);
  }
  __device__ static void b_rows(const double* row, Keep& kp) { spline_rows(row, kp); }
  __device__ static void b_coefs(const TermD& t, const Ctx& c, Keep& kp) { spline_coefs(t, c, kp); }
  __device__ static double b_value(const TermD&, const double*, const Ctx&, Keep& kp, double&) { return spline_value(kp); }  // taps are 0 outside
};
template <>
struct Stage<GWI_TERM_LINEAR_SPLINE> {
  static constexpr bool kSpline = true;
  static constexpr int kDoubles = kSplineStage;
  using Keep = SplineKeep;
  __device__ static void phase_a(const TermD& t, const typename Term<GWI_TERM_LINEAR_SPLINE>::In& in, double* row) { stage_spline(t, in.x0, true, row); }
  __device__ static void b_rows(const double* row, Keep& kp) { spline_rows(row, kp); }
  __device__ static void b_coefs(const TermD& t, const Ctx& c, Keep& kp) { spline_coefs(t, c, kp); }
  __device__ static double b_value(const TermD&, const double*, const Ctx&, Keep& kp, double& lin) {
    const double f = spline_value(kp);
    kp.scale = f > 0.0 ? fast_rcp(f) : 0.0;  // dl / dc_k = B_k / f; f <= 0 (outside the domain too) makes the sample dead (lin > 0 test)
    lin *= f;
    return 0.0;
  }
};

// ---- compile-time chain.  Every entry is  kind + 100 * tiles:  tiles = 16-basis gradient tiles of a spline term
//      (n_basis <= 16 tiles, checked by the host), 0 for the other kinds. ------------------------------------------------
// ROWS: the gradient of the spline coefficients goes into LDS rows (scan_rows_kernel) instead of MFMA tiles.
template <bool ROWS, int... KTs>
struct MChain;
template <bool ROWS>
struct MChain<ROWS> {
  static constexpr int kNumAcc = 0, kTiles = 0, kRowDoubles = 0;
  __device__ void init() {}
  __device__ void load(int, const Ctx&, SIdx) {}
  __device__ void phase_a(int, const Ctx&, double*) {}
  __device__ void commit() {}
  template <int N>
  __device__ void g_rows(const double*) {}
  template <int N>
  __device__ void g_coefs(int, const Ctx&) {}
  template <int N>
  __device__ double g_value(int, const Ctx&, double&) { return 0.0; }
  template <int N>
  __device__ double forward_after(int, const Ctx&, const double*, double&) { return 0.0; }
  __device__ double forward(int, const Ctx&, const double*, double&) { return 0.0; }
  __device__ void gather(int, const double*) {}
  __device__ void accumulate(int, const Ctx&, double, int, double*, const double*) {}
  __device__ void collect(int, const Ctx&, double*, int*) {}
  template <class F>
  __device__ void for_each_tile(int, const Ctx&, F&&) {}
};
template <bool ROWS, int KT, int... Rest>
struct MChain<ROWS, KT, Rest...> {
  static constexpr int K = KT % 100, NT = ROWS ? 0 : KT / 100;
  using S = Stage<K>;
  using RestT = MChain<ROWS, Rest...>;
  static_assert(S::kSpline == (KT / 100 > 0), "spline kinds carry their tile count (kind + 100 * tiles), the others none");
  static constexpr int kNumAcc = Term<K>::kNumAcc + RestT::kNumAcc;
  static constexpr int kTiles = NT + RestT::kTiles;
  static constexpr int kRowDoubles = S::kDoubles + RestT::kRowDoubles;
  typename Term<K>::In in;     // phase A lanes: the column values of the trip about to be staged (loaded one trip ahead)
  typename S::Keep keep;       // phase B lanes: what the accumulation needs from the evaluation
  typename Term<K>::Acc acc;
  v4d tile[NT > 0 ? NT : 1];
  double aval[NT > 0 ? NT : 1];  // the A operands of the round being accumulated, gathered ahead of its matrix instructions
  RestT rest;
  __device__ void init() {
    Term<K>::init(acc);
#pragma unroll
    for (int t = 0; t < NT; ++t) tile[t] = v4d{0.0, 0.0, 0.0, 0.0};
    rest.init();
  }
  __device__ void load(int ti, const Ctx& c, SIdx idx) {
    Term<K>::load(c.tcols[ti], idx, in);
    rest.load(ti + 1, c, idx);
  }
  __device__ void phase_a(int ti, const Ctx& c, double* row) {
    S::phase_a(c.a->terms[ti], in, row);
    rest.phase_a(ti + 1, c, row + S::kDoubles);
  }
  __device__ void commit() {
    keep.commit();
    rest.commit();
  }
  // The forward half of a round, kFwdGroup terms at a time: every term of the group requests its staged words, then every
  // spline term of the group its four coefficients (which needs k), then the arithmetic -- the LDS round trips of a group
  // overlap, and only one group's operands (16 registers per spline term) are live at a time (all seven terms of config 5 at
  // once: 256 registers and spills)
  static constexpr int kFwdGroup = GWI_MFMA_FWD_GROUP;
  template <int N>
  __device__ void g_rows(const double* row) {
    if constexpr (N > 0) {
      S::b_rows(row, keep);
      rest.template g_rows<N - 1>(row + S::kDoubles);
    }
  }
  template <int N>
  __device__ void g_coefs(int ti, const Ctx& c) {
    if constexpr (N > 0) {
      S::b_coefs(c.a->terms[ti], c, keep);
      rest.template g_coefs<N - 1>(ti + 1, c);
    }
  }
  template <int N>
  __device__ double g_value(int ti, const Ctx& c, double& lin) {
    if constexpr (N > 0) {
      const double l = S::b_value(c.a->terms[ti], c.derived[ti], c, keep, lin);
      return l + rest.template g_value<N - 1>(ti + 1, c, lin);
    } else {
      return 0.0;
    }
  }
  template <int N>
  __device__ double forward_after(int ti, const Ctx& c, const double* row, double& lin) {  // skip N terms, then go on
    if constexpr (N > 0)
      return rest.template forward_after<N - 1>(ti + 1, c, row + S::kDoubles, lin);
    else
      return forward(ti, c, row, lin);
  }
  __device__ double forward(int ti, const Ctx& c, const double* row, double& lin) {
    g_rows<kFwdGroup>(row);
    g_coefs<kFwdGroup>(ti, c);
    const double l = g_value<kFwdGroup>(ti, c, lin);
    return l + forward_after<kFwdGroup>(ti, c, row, lin);
  }
  // The A operands of the current round's matrix instructions, requested from the staging rows as soon as k is known (the
  // caller fences the instruction scheduler behind this pass: left to itself it sinks every read to its matrix
  // instruction -- read, full LDS round trip, MFMA, eight times per round)
  __device__ void gather(int basis_lane, const double* row) {
    if constexpr (S::kSpline && !ROWS) {
      const int d0 = basis_lane - keep.k_cur;
#pragma unroll
      for (int t = 0; t < NT; ++t) aval[t] = row[2 + max(-1, min(16 * t + d0, 4))];  // v_med3_i32; row[1] and row[6] are 0
    }
    rest.gather(basis_lane, row + S::kDoubles);
  }
  // w = this lane's weight w[sample, point]; grow = this lane's gradient row in LDS (ROWS); row = the sample's staged words
  // for this term and the ones after it
  __device__ void accumulate(int ti, const Ctx& c, double w, int basis_lane, double* grow, const double* row) {
    if constexpr (S::kSpline) {
      const double bw = w * keep.scale_cur;
      if constexpr (ROWS) {
        double* g = grow + c.a->terms[ti].th0 + keep.k_cur;
        unsafeAtomicAdd(g, bw * row[2]);
        unsafeAtomicAdd(g + 1, bw * row[3]);
        unsafeAtomicAdd(g + 2, bw * row[4]);
        unsafeAtomicAdd(g + 3, bw * row[5]);
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#if defined(GWI_ABL_NO_MFMA)  // timing-only ablation (results wrong): the A operand still fetched, no matrix instruction
          tile[t][0] += aval[t] * bw;
#else
          tile[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aval[t], bw, tile[t], 0, 0, 0);
#endif
        }
      }
    } else {
      Term<K>::accumulate(c.a->terms[ti], c, w, keep.st_cur, acc);
    }
    rest.accumulate(ti + 1, c, w, basis_lane, grow, row + S::kDoubles);
  }
  __device__ void collect(int ti, const Ctx& c, double* vals, int* th) {
    Term<K>::collect(c.a->terms[ti], acc, vals, th);
    rest.collect(ti + 1, c, vals + Term<K>::kNumAcc, th + Term<K>::kNumAcc);
  }
  // f(theta offset of the tile's first coefficient, number of valid rows, the tile)
  template <class F>
  __device__ void for_each_tile(int ti, const Ctx& c, F&& f) {
    if constexpr (S::kSpline && !ROWS) {
      const TermD& t = c.a->terms[ti];
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) f(t.th0 + 16 * tl, t.n_basis - 16 * tl, tile[tl]);
    }
    rest.for_each_tile(ti + 1, c, f);
  }
};

// dynamic LDS of the two kernels in doubles (the host sizes the launch with it).  rows_rep = 0: scan_mfma_kernel; else the
// sample-slot replicas (4, 2 or 1) of scan_rows_kernel's gradient rows
__host__ __device__ inline size_t mfma_lds_doubles(int n_theta, int n_terms, int row_doubles, int rows_rep) {
  const size_t th_pad = (size_t)n_theta | 1, der_pad = (size_t)(n_terms * kMaxDerived) | 1;
  const size_t S = rows_rep ? 16 : GWI_MFMA_STAGE;
  size_t stage = (size_t)kWaves * S * (size_t)((row_doubles + 1) | 1);  // + kappa; odd stride
  if (rows_rep) {
    if (stage < 256) stage = 256;  // the scalar-sum staging of the epilogue aliases the rows
    return kPts * th_pad + kPts * der_pad + stage + (size_t)rows_rep * kPts * th_pad;
  }
  const size_t epilogue = (size_t)kPts * th_pad + (size_t)kWaves * 256;  // per-point output rows + the D staging: aliases the staging rows
  return kPts * th_pad + kPts * der_pad + (stage > epilogue ? stage : epilogue);
}

// ---- the kernel.  grid = (scan blocks [+ normaliser blocks], groups of 16 hyper-parameter points) x 256 threads. ------
template <bool ROWS, int... KTs>
__device__ __forceinline__ void scan_points_body(const KArgs& a) {
  using ChainT = MChain<ROWS, KTs...>;
  constexpr int kRow = ((ChainT::kRowDoubles + 1) | 1);  // doubles per staged sample (kappa first), odd
  constexpr int kStageS = ROWS ? 16 : GWI_MFMA_STAGE;    // samples per wavefront and trip (phase A lanes)
  extern __shared__ double s_dyn[];
  __shared__ double s_mx[kWaves][64];
  __shared__ int s_enorm[kPts];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, j = lane & (kPts - 1);
  const int group = blockIdx.y;
  const int kb_raw = group * kPts + j;
  const int kb = kb_raw < a.k_batch ? kb_raw : a.k_batch - 1;  // a ragged last group repeats the last point (results not written)
  const int th_pad = a.n_theta | 1, der_pad = (a.n_terms * kMaxDerived) | 1;
  double* const s_thetaK = s_dyn;                       // [16][th_pad]
  double* const s_derK = s_thetaK + kPts * th_pad;      // [16][der_pad]
  double* const s_rows = s_derK + kPts * der_pad;       // main loop: [kWaves][kStageS][kRow] staging rows
  // MFMA: the epilogue's [16][th_pad] gradient numerators per point and its [kWaves][16][16] D staging (also the scalar-sum
  // staging [256]) alias the staging rows.  ROWS: the gradient rows [rep][16][th_pad] follow the staging rows -- one row
  // per (sample slot q & (rep - 1), point j): the 16 lanes of a sample add to the same coefficient of 16 different rows
  // (odd stride: 16 different banks), the four samples of a wave instruction to different replicas, so the atomics never
  // meet on an address inside an instruction whatever the data; replica 0 becomes the output rows in the epilogue
  const int rows_rep = ROWS ? a.rows_rep : 0;
  const size_t stage_doubles = (size_t)kWaves * kStageS * kRow;
  double* const s_grad = s_rows + (stage_doubles < 256 ? 256 : stage_doubles);
  double* const s_outK = ROWS ? s_grad : s_rows;
  double* const s_stage = ROWS ? s_rows : s_outK + kPts * th_pad;
  if (ROWS)
    for (int p = tid; p < rows_rep * kPts * th_pad; p += kBlock) s_grad[p] = 0.0;

  if (blockIdx.x == 0 && group == 0 && tid == 0) *a.seq_dev = a.norm_seq;
  if ((int)blockIdx.x < a.n_norms) {  // grid normalisers of the group's points, one after the other
    __shared__ double s_ntheta[GWI_MAX_THETA];
    const int jn = blockIdx.x;
    for (int p = 0; p < kPts; ++p) {
      const int k = group * kPts + p;
      if (k >= a.k_batch) break;
      norm_block(a.norms, a.tblocks[k].theta, a.n_theta, jn, a.norm_out_host + k * a.n_norms + jn, a.norm_stamps_host + k * a.n_norms + jn, a.norm_seq, s_ntheta, &s_mx[0][0]);
      __syncthreads();
    }
    return;
  }
  const int b = (int)blockIdx.x - a.n_norms;
  const int n_pe_blocks = a.n_ev * a.tiles_per_event;

  // stage the 16 points' hyper-parameters and derived scalars
  for (int p = tid; p < kPts * a.n_theta; p += kBlock) {
    const int pt = p / a.n_theta, idx = p - pt * a.n_theta;
    const int k = group * kPts + pt;
    s_thetaK[pt * th_pad + idx] = a.tblocks[k < a.k_batch ? k : a.k_batch - 1].theta[idx];
  }
  for (int p = tid; p < kPts * a.n_terms * kMaxDerived; p += kBlock) {
    const int pt = p / (a.n_terms * kMaxDerived), idx = p - pt * (a.n_terms * kMaxDerived);
    const int k = group * kPts + pt;
    s_derK[pt * der_pad + idx] = (&a.tblocks[k < a.k_batch ? k : a.k_batch - 1].derived[0][0])[idx];
  }

  long long start, end, base;
  Ctx ctx;
  ctx.a = &a;
  ctx.theta = s_thetaK + j * th_pad;  // lane-varying: this lane's point
  ctx.derived = reinterpret_cast<const double(*)[kMaxDerived]>(s_derK + j * der_pad);
  ctx.coefs = s_thetaK + j * th_pad;
  ctx.gacc = nullptr;
  ctx.rep_shift = 0;
  if (b < n_pe_blocks) {
    const int e = b / a.tiles_per_event;
    const int t = b - e * a.tiles_per_event;
    start = (long long)t * a.chunk_pe;
    end = start + a.chunk_pe < a.n_pe ? start + a.chunk_pe : a.n_pe;
    base = (long long)e * a.n_pe;
    ctx.tcols = a.pe_tcols;
  } else {
    const int t = b - n_pe_blocks;
    start = (long long)t * a.chunk_inj;
    end = start + a.chunk_inj < a.n_inj ? start + a.chunk_inj : a.n_inj;
    base = 0;
    ctx.tcols = a.inj_tcols;
  }
  const double* kappa_col = b < n_pe_blocks ? a.kappa_pe : a.kappa_inj;
  const int n_tile = (int)(end - start);

  // the reference exponent of (this tile, this lane's point), in binades: the tile's maximum at the previous batched launch
  int* const nref_slot = a.tile_nref + ((long long)(a.nref_row0 + kb) * a.nref_stride + b);
  int n_ref = *nref_slot;
  n_ref = n_ref == kNoRef ? 0 : n_ref;
  const int ref_slack = a.square ? 215 : 430;

  ChainT chain;
  chain.init();
  double s1 = 0.0, s2 = 0.0, lane_max = GWI_NEG_INF;
  double* const my_rows = s_rows + (size_t)wave * kStageS * kRow;
  double* const grow = ROWS ? s_grad + (size_t)((q & (rows_rep - 1)) * kPts + j) * th_pad : nullptr;

  // a trip of the workgroup covers 256 samples: wave w, phase-A lane a -> sample  i + 64 w + a
  double kap = 0.0;
  auto issue_loads = [&](int i) {
    const int s = i + kStageS * wave + lane;
    const SIdx idx{base + start, (unsigned)(s < n_tile ? s : n_tile - 1) << 3};
    kap = gload(kappa_col, idx);
    chain.load(0, ctx, idx);
  };
  const bool a_lane = lane < kStageS;
  if (a_lane && kStageS * wave < n_tile) issue_loads(0);  // this wave's first samples
  __syncthreads();  // theta / derived staged
  for (int i = 0; i + kStageS * wave < n_tile; i += kWaves * kStageS) {  // wave-uniform: a wave stops with its samples
    // ---- phase A: stage the trip whose columns were loaded one trip ago, then request the next trip's
    if (a_lane) {
      double* row = my_rows + lane * kRow;
      const bool valid = i + kStageS * wave + lane < n_tile;
      row[0] = valid ? kap : GWI_NEG_INF;
      chain.phase_a(0, ctx, row + 1);
      const int i_next = i + kWaves * kStageS;
      if (i_next + kStageS * wave < n_tile) issue_loads(i_next);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's own LDS traffic is processed in order: the rows are there
    // ---- phase B: rounds of (4 samples x 16 points).  The A operands of the round's matrix instructions are requested as
    //      soon as the forward half has read k -- ahead of the exponential -- and the instruction scheduler is fenced there:
    //      left to itself it sinks every gather to its matrix instruction (read, full LDS round trip, MFMA, eight times a
    //      round).  A two-stage software pipeline over rounds on top of this measured the same (36.6 vs 36.4 us per evaluation
    //      at config 5) and is not kept.
#ifdef GWI_ABL_B_NONE  // timing-only ablation: phase A and the staging only
    constexpr int kRounds = 0;
#else
    constexpr int kRounds = kStageS / 4;
#endif
#pragma unroll 1
    for (int g = 0; g < kRounds; ++g) {
      const double* row = my_rows + (4 * g + q) * kRow;
      const double kappa = row[0];
      double lin = 1.0;
#ifdef GWI_ABL_NO_FWD  // timing-only ablation: no forward half
      double ell = kappa;
#else
      double ell = kappa + chain.forward(0, ctx, row + 1, lin);
#endif
      chain.commit();
      chain.gather(j, row + 1);
      __builtin_amdgcn_sched_barrier(0);
      const bool live = (ell < GWI_POS_INF) && (ell > GWI_NEG_INF) && (lin > 0.0) && (lin < GWI_POS_INF);
      if (!live) ell = GWI_NEG_INF;
      lane_max = fmax(lane_max, ell);
      double w = live ? lin * fast_exp_shift(ell, n_ref) : 0.0;
      if (a.square) w *= w;
      s1 += w;
      s2 += w * w;
      __builtin_amdgcn_sched_barrier(0);
      chain.accumulate(0, ctx, w, j, grow, row + 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows have been read before the next trip overwrites them
  }

  // ---- epilogue.  Tile maxima and reference check per point: lanes (w, q, j) -> point j
  s_mx[wave][lane] = lane_max;
  __syncthreads();  // also: every wave is done with the staging rows, which s_outK / s_stage alias
  if (tid < kPts) {
    double mm = GWI_NEG_INF;
#pragma unroll
    for (int w_ = 0; w_ < kWaves; ++w_)
#pragma unroll
      for (int q_ = 0; q_ < 4; ++q_) mm = fmax(mm, s_mx[w_][q_ * 16 + tid]);
    const int n_max = (mm == GWI_NEG_INF) ? kNoRef : (int)__builtin_rint(fmin(fmax(mm, -7.0e5), 7.0e5) * kLog2e);
    if (kb_raw < a.k_batch) {
      *nref_slot = n_max;
      const int dist = n_max - n_ref;
      if (n_max != kNoRef && (dist > ref_slack || dist < -ref_slack)) {
        __hip_atomic_store(a.redo_host, a.norm_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(a.redo_dev, a.norm_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (ROWS) {  // fold the sample-slot replicas into replica 0 (fixed order)
    for (int p = tid; p < kPts * th_pad; p += kBlock) {
      double g = s_grad[p];
      for (int r = 1; r < rows_rep; ++r) g += s_grad[(size_t)r * kPts * th_pad + p];
      s_grad[p] = g;
    }
  } else {
    for (int p = tid; p < kPts * th_pad; p += kBlock) s_outK[p] = 0.0;
  }

  // scalar sums: lanes (w, q, j) -> point j, fixed order over the 16 (w, q) partials
  constexpr int kNV = 2 + ChainT::kNumAcc;
  double vals[kNV];
  int th[kNV];
  vals[0] = s1;
  vals[1] = s2;
  th[0] = th[1] = -1;
  chain.collect(0, ctx, vals + 2, th + 2);
  const long long n_blocks = n_pe_blocks + a.n_inj_tiles;
  double* const out_j = a.partials + ((long long)kb_raw * n_blocks + b) * a.rec_stride;  // record of point j (valid iff kb_raw < k_batch)
  int e_norm = 0;  // threads 0..15: the point's normalisation exponent (record normalised to S1 in [1, 2), as scan_kernel's)
#pragma unroll
  for (int v = 0; v < kNV; ++v) {
    __syncthreads();
    s_stage[tid] = vals[v];
    __syncthreads();
    if (tid < kPts) {
      double r = 0.0;
#pragma unroll
      for (int w_ = 0; w_ < kWaves; ++w_)
#pragma unroll
        for (int q_ = 0; q_ < 4; ++q_) r += s_stage[w_ * 64 + q_ * 16 + tid];
      if (v == 0) {
        const bool has_sum = r > 0.0 && r < GWI_POS_INF;
        e_norm = has_sum ? ilogb(r) : 0;
        s_enorm[tid] = e_norm;
        if (kb_raw < a.k_batch) {
          out_j[0] = has_sum ? (double)((a.square ? 2 * n_ref : n_ref) + e_norm) * kLn2 : GWI_NEG_INF;
          out_j[1] = has_sum ? ldexp(r, -e_norm) : 0.0;
        }
      }
      if (v == 1 && kb_raw < a.k_batch) out_j[2] = (r > 0.0 && r < GWI_POS_INF) ? ldexp(r, -2 * e_norm) : 0.0;
      if (v >= 2) s_outK[tid * th_pad + th[v]] += r;  // tid == j here; several accumulators may feed one slot (in order)
    }
  }
  // gradient tiles: D of wave w -> staging [w][row][point]; thread (row, point) sums the four waves in order
  chain.for_each_tile(0, ctx, [&](int th0, int rows, const v4d& d) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) s_stage[wave * 256 + (q + 4 * r) * kPts + j] = d[r];  // D: lane (q, j), register r = row q + 4 r
    __syncthreads();
    const int row = tid >> 4, pt = tid & (kPts - 1);
    if (row < rows) {
      const double g = (s_stage[row * kPts + pt] + s_stage[256 + row * kPts + pt]) + (s_stage[512 + row * kPts + pt] + s_stage[768 + row * kPts + pt]);
      s_outK[pt * th_pad + th0 + row] += g;  // shared coefficient blocks (IID models) accumulate; one thread per slot per tile
    }
  });
  __syncthreads();
  for (int p = tid; p < kPts * a.n_theta; p += kBlock) {
    const int pt = p / a.n_theta, idx = p - pt * a.n_theta;
    const int k = group * kPts + pt;
    if (k < a.k_batch) a.partials[((long long)k * n_blocks + b) * a.rec_stride + kRecHeader + idx] = ldexp(s_outK[pt * th_pad + idx], -s_enorm[pt]);
  }
}

// the two instantiations of the body: gradient tiles on the matrix cores / gradient rows in LDS
template <int U_UNUSED, int... KTs>
__global__ __launch_bounds__(kBlock, GWI_MFMA_WAVES) void scan_mfma_kernel(const KArgs a) {
  scan_points_body<false, KTs...>(a);
}
template <int U_UNUSED, int... KTs>
__global__ __launch_bounds__(kBlock, 2) void scan_rows_kernel(const KArgs a) {
  scan_points_body<true, KTs...>(a);
}

}  // namespace gwi
