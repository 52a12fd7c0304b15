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
// atomics, a fixed summation order (bit-reproducible), and the matrix pipe runs beside the vector pipe that evaluates the
// densities.  The forward half stays in the 4-tap form: as a dense product it would spend one MFMA (32 cycles) per 4 basis
// functions per 16 samples where the taps need 4 FMAs per sample (measured: tools/microbench/mfma_f64_layout.hip gives the
// instruction's rate; DESIGN.md section 4 has the arithmetic and the measurement of this kernel against the tap kernel).
//
// Layout: a wavefront holds 4 samples x 16 hyper-parameter points: lane l = (q = l / 16: sample slot, j = l % 16: point).
//   B operand  lane (q, j) holds w[sample q, point j]                      -- the lane's own weight
//   A operand  lane (q, i) holds B_{16 tile + i}(x_{sample q})             -- the lane's own sample, basis = its j index
//   D          lane (q, j), register r holds G[16 tile + q + 4 r][point j]
// (layout confirmed on the hardware by tools/microbench/mfma_f64_layout.hip).  Per-sample work that does not depend on
// the hyper-parameters (column loads, knot interval, taps) is replicated across the 16 point-lanes of a sample.
#pragma once
#include "gwi_device.h"

namespace gwi {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kPts = 16;  // hyper-parameter points per wavefront (the N of the MFMA tile)

// the tap of basis function `first + d` for a sample whose non-zero bases start at `first`: b_d for d in 0..3, else 0
__device__ __forceinline__ double tap_select(const Taps& b, int d) {
  double v = 0.0;
  v = d == 0 ? b.b0 : v;
  v = d == 1 ? b.b1 : v;
  v = d == 2 ? b.b2 : v;
  v = d == 3 ? b.b3 : v;
  return v;
}

// ---- per-kind MFMA operands: one row slab of the (implicit) design matrix and the weight that multiplies it ----------
template <int K>
struct SplineOperands;
template <>
struct SplineOperands<GWI_TERM_EXP_SPLINE> {
  struct Prep {
    Taps b;
    int k;
  };
  __device__ static Prep prepare(const typename Term<GWI_TERM_EXP_SPLINE>::State& s) { return {cubic_taps(s.t), s.k}; }
  __device__ static double a(const Prep& p, int basis) { return p.k >= 0 ? tap_select(p.b, basis - p.k) : 0.0; }
  __device__ static double b(const typename Term<GWI_TERM_EXP_SPLINE>::State&, double w) { return w; }
};
template <>
struct SplineOperands<GWI_TERM_LINEAR_SPLINE> {
  struct Prep {
    Taps b;
    int k;
  };
  __device__ static Prep prepare(const typename Term<GWI_TERM_LINEAR_SPLINE>::State& s) { return {cubic_taps(s.t), s.k}; }
  __device__ static double a(const Prep& p, int basis) { return tap_select(p.b, basis - p.k); }
  __device__ static double b(const typename Term<GWI_TERM_LINEAR_SPLINE>::State& s, double w) { return w * s.inv_f; }  // dl/dc_k = B_k / f
};
template <>
struct SplineOperands<GWI_TERM_EXP_SPLINE_LERP> {
  struct Prep {
    Taps b0, b1;
    int k0, k1;
    double f;
  };
  __device__ static Prep prepare(const typename Term<GWI_TERM_EXP_SPLINE_LERP>::State& s) { return {cubic_taps(s.t0), cubic_taps(s.t1), s.k0, s.k1, s.f}; }
  __device__ static double a(const Prep& p, int basis) {
    const double a0 = p.k0 >= 0 ? tap_select(p.b0, basis - p.k0) : 0.0;
    const double a1 = p.k1 >= 0 ? tap_select(p.b1, basis - p.k1) : 0.0;
    return fma(p.f, a1 - a0, a0);  // the same blend of the two grid nodes as the value
  }
  __device__ static double b(const typename Term<GWI_TERM_EXP_SPLINE_LERP>::State&, double w) { return w; }
};

// ---- compile-time chain.  Every entry is  kind + 100 * tiles:  tiles = 16-basis gradient tiles of a spline term
//      (n_basis <= 16 tiles, checked by the host), 0 for the other kinds. ------------------------------------------------
template <int U, int... KTs>
struct MChain;
template <int U>
struct MChain<U> {
  static constexpr int kNumAcc = 0, kTiles = 0;
  __device__ void init() {}
  __device__ void load(int, int, int, const Ctx&, SIdx) {}
  __device__ void advance() {}
  __device__ double eval(int, int, const Ctx&, double&) { return 0.0; }
  __device__ void accumulate(int, int, const Ctx&, double, int) {}
  __device__ void collect(int, const Ctx&, double*, int*) {}
  template <class F>
  __device__ void for_each_tile(int, const Ctx&, F&&) {}
};
template <int U, int KT, int... Rest>
struct MChain<U, KT, Rest...> {
  static constexpr int K = KT % 100, NT = KT / 100;
  static constexpr bool kIsSpline = Term<K>::kSpline;
  static_assert(kIsSpline == (NT > 0), "spline kinds carry their tile count (kind + 100 * tiles), the others none");
  static constexpr int kNumAcc = Term<K>::kNumAcc + MChain<U, Rest...>::kNumAcc;
  static constexpr int kTiles = NT + MChain<U, Rest...>::kTiles;
  typename Term<K>::In in[2][U];
  typename Term<K>::State st[U];
  typename Term<K>::Acc acc;
  v4d tile[NT > 0 ? NT : 1];
  MChain<U, Rest...> rest;
  __device__ void init() {
    Term<K>::init(acc);
#pragma unroll
    for (int t = 0; t < NT; ++t) tile[t] = v4d{0.0, 0.0, 0.0, 0.0};
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
    const double l = Term<K>::eval(c.a->terms[ti], c.derived[ti], c, in[0][u], st[u], lin);
    return l + rest.eval(u, ti + 1, c, lin);
  }
  // w = this lane's weight w[sample, point]; basis_lane = lane & 15 (the lane's row inside a gradient tile)
  __device__ void accumulate(int u, int ti, const Ctx& c, double w, int basis_lane) {
    if constexpr (kIsSpline) {
      const auto prep = SplineOperands<K>::prepare(st[u]);
      const double bw = SplineOperands<K>::b(st[u], w);
#pragma unroll
      for (int t = 0; t < NT; ++t) tile[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(SplineOperands<K>::a(prep, 16 * t + basis_lane), bw, tile[t], 0, 0, 0);
    } else {
      Term<K>::accumulate(c.a->terms[ti], c, w, st[u], acc);
    }
    rest.accumulate(u, ti + 1, c, w, basis_lane);
  }
  __device__ void collect(int ti, const Ctx& c, double* vals, int* th) {
    Term<K>::collect(c.a->terms[ti], acc, vals, th);
    rest.collect(ti + 1, c, vals + Term<K>::kNumAcc, th + Term<K>::kNumAcc);
  }
  // f(theta offset of the tile's first coefficient, number of valid rows, the tile)
  template <class F>
  __device__ void for_each_tile(int ti, const Ctx& c, F&& f) {
    if constexpr (kIsSpline) {
      const TermD& t = c.a->terms[ti];
#pragma unroll
      for (int tl = 0; tl < NT; ++tl) f(t.th0 + 16 * tl, t.n_basis - 16 * tl, tile[tl]);
    }
    rest.for_each_tile(ti + 1, c, f);
  }
};

// ---- the kernel.  grid = (scan blocks [+ normaliser blocks], groups of 16 hyper-parameter points) x 256 threads.
//      Dynamic LDS: theta and derived scalars of the group's 16 points ([16][n_theta | 1] + [16][n_terms * kMaxDerived | 1]
//      doubles), then the per-point output rows [16][n_theta] and a 16 x 16 x 4-wave staging tile. ------------------------
template <int U, int... KTs>
// two wavefronts per SIMD: left alone the compiler takes 284 registers for the config-5 sequence (one wave per SIMD, nothing to
// hide a dependent fp64 chain behind); bounded to 256 it needs 217 and spills nothing
__global__ __launch_bounds__(kBlock, 2) void scan_mfma_kernel(const KArgs a) {
  using ChainT = MChain<U, KTs...>;
  constexpr int kU = U;
  extern __shared__ double s_dyn[];
  __shared__ double s_mx[kWaves][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, j = lane & (kPts - 1);
  const int group = blockIdx.y;
  const int kb_raw = group * kPts + j;
  const int kb = kb_raw < a.k_batch ? kb_raw : a.k_batch - 1;  // a ragged last group repeats the last point (results not written)
  const int th_pad = a.n_theta | 1, der_pad = (a.n_terms * kMaxDerived) | 1;
  double* const s_thetaK = s_dyn;                       // [16][th_pad]
  double* const s_derK = s_thetaK + kPts * th_pad;      // [16][der_pad]
  double* const s_outK = s_derK + kPts * der_pad;       // [16][n_theta]: gradient numerators per point
  double* const s_stage = s_outK + kPts * a.n_theta;    // [kWaves][16][16] (also the scalar-sum staging [vals][256])

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
  for (int p = tid; p < kPts * a.n_theta; p += kBlock) s_outK[p] = 0.0;
  __syncthreads();

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

  ChainT chain;
  chain.init();
  double s1 = 0.0, s2 = 0.0;
  double m_ref = GWI_NEG_INF;  // reference exponent of (this tile, this lane's point): fixed at the first live trip
  int all_set = 0, over = 0;   // workgroup-uniform: every point has its reference; some sample outran one by > slack
  constexpr double kRefSlack = 150.0;

  // a trip of the workgroup covers 16 U samples: wave w, slot q, unroll u -> sample  first + 16 u + 4 w + q
  double kap[2][kU];
  auto issue_loads = [&](int buf, long long first) {
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const long long s = first + 16 * u + 4 * wave + q;
      const SIdx idx{base + start, (unsigned)((s < end ? s : end - 1) - start) << 3};
      kap[buf][u] = gload(kappa_col, idx);
      chain.load(buf, u, 0, ctx, idx);
    }
  };
  issue_loads(0, start);
  int trip = 0;
  for (long long first = start; first < end; first += 16 * kU, ++trip) {  // workgroup-uniform
    const long long next = first + 16 * kU;
    if (next < end) issue_loads(1, next);
    double ell[kU], lin[kU];
    bool live[kU];
    double mx_lane = GWI_NEG_INF;
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const bool valid = first + 16 * u + 4 * wave + q < end;
      lin[u] = 1.0;
      ell[u] = kap[0][u] + chain.eval(u, 0, ctx, lin[u]);
      live[u] = valid && (ell[u] < GWI_POS_INF) && (ell[u] > GWI_NEG_INF) && (lin[u] > 0.0) && (lin[u] < GWI_POS_INF);
      if (!live[u]) ell[u] = GWI_NEG_INF;
      mx_lane = fmax(mx_lane, ell[u]);
    }
    if (!all_set) {  // workgroup-uniform: some point of the group has not seen a live sample yet
      double* mx_slot = &s_mx[0][0];  // single buffer: two barriers bracket its use (only while references are being fixed)
      mx_slot[wave * 64 + lane] = mx_lane;
      __syncthreads();
      double mm = GWI_NEG_INF;
#pragma unroll
      for (int w_ = 0; w_ < kWaves; ++w_)
#pragma unroll
        for (int q_ = 0; q_ < 4; ++q_) mm = fmax(mm, mx_slot[w_ * 64 + q_ * 16 + j]);
      if (m_ref == GWI_NEG_INF) m_ref = mm;  // per point; never moves once set
      all_set = __builtin_amdgcn_ballot_w64(m_ref == GWI_NEG_INF) == 0;  // the same in every wave: all read the same 16 maxima
      __syncthreads();
    } else if (__builtin_amdgcn_ballot_w64(mx_lane > m_ref + kRefSlack) != 0) {
      over = 1;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      double w = (live[u] && m_ref != GWI_NEG_INF) ? lin[u] * fast_exp(ell[u] - m_ref) : 0.0;
      if (a.square) w *= w;
      s1 += w;
      s2 += w * w;
      chain.accumulate(u, 0, ctx, w, j);
    }
    if (next < end) {
      chain.advance();
#pragma unroll
      for (int u = 0; u < kU; ++u) kap[0][u] = kap[1][u];
    }
  }
  if (over && lane == 0) {
    __hip_atomic_store(a.redo_host, a.norm_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(a.redo_dev, a.norm_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }

  // ---- epilogue.  Scalar sums: lanes (w, q, j) -> point j, fixed order over the 16 (w, q) partials.
  constexpr int kNV = 2 + ChainT::kNumAcc;
  double vals[kNV];
  int th[kNV];
  vals[0] = s1;
  vals[1] = s2;
  th[0] = th[1] = -1;
  chain.collect(0, ctx, vals + 2, th + 2);
  const long long n_blocks = n_pe_blocks + a.n_inj_tiles;
  double* const out_j = a.partials + ((long long)kb_raw * n_blocks + b) * a.rec_stride;  // record of point j (valid iff kb_raw < k_batch)
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
      if (v == 0 && kb_raw < a.k_batch) out_j[1] = r;
      if (v == 1 && kb_raw < a.k_batch) out_j[2] = r;
      if (v >= 2) s_outK[tid * a.n_theta + th[v]] += r;  // tid == j here; several accumulators may feed one slot (in order)
    }
  }
  if (tid < kPts && kb_raw < a.k_batch) out_j[0] = a.square ? 2.0 * m_ref : m_ref;
  // gradient tiles: D of wave w -> staging [w][row][point]; thread (row, point) sums the four waves in order
  chain.for_each_tile(0, ctx, [&](int th0, int rows, const v4d& d) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) s_stage[wave * 256 + (q + 4 * r) * kPts + j] = d[r];  // D: lane (q, j), register r = row q + 4 r
    __syncthreads();
    const int row = tid >> 4, pt = tid & (kPts - 1);
    if (row < rows) {
      const double g = (s_stage[row * kPts + pt] + s_stage[256 + row * kPts + pt]) + (s_stage[512 + row * kPts + pt] + s_stage[768 + row * kPts + pt]);
      s_outK[pt * a.n_theta + th0 + row] += g;  // shared coefficient blocks (IID models) accumulate; one thread per slot per tile
    }
  });
  __syncthreads();
  for (int p = tid; p < kPts * a.n_theta; p += kBlock) {
    const int pt = p / a.n_theta, idx = p - pt * a.n_theta;
    const int k = group * kPts + pt;
    if (k < a.k_batch) a.partials[((long long)k * n_blocks + b) * a.rec_stride + kRecHeader + idx] = s_outK[p];
  }
}

}  // namespace gwi
