// gwi_device.h -- device side of the population-likelihood engine (gfx950 / CDNA4 only).
//
// One fused "scan" launch streams the catalog columns once and produces, per workgroup, a partial
// record (running max m, S1 = sum e^{l-m}, S2 = sum e^{2(l-m)}, G[p] = sum e^{l-m} dl/dtheta_p);
// extra workgroups of the same launch integrate the grid normalisers.  Two tiny launches then
// combine records per event and over events.  Everything is fp64.
//
// Reference arithmetic being replaced (paths relative to the reference root):
//   per-sample densities      gwinferno/distributions.py:100-162, models/parametric/parametric.py:27-145
//   B-spline projection       gwinferno/interpolation.py:293-304, 381-394 (dense GEMV in the reference;
//                             here 4 taps per sample recomputed in registers, uniform knots :98-106)
//   masked scatter            models/bsplines/single.py:77-109 (here: kappa = -inf)
//   reductions                pipeline/analysis.py:50-136
#pragma once
#include <hip/hip_runtime.h>

#include "gwi_engine.h"

namespace gwi {

constexpr int kBlock = 256;            // 4 wavefronts of 64
constexpr int kWaves = kBlock / 64;
constexpr int kMaxDerived = 6;
constexpr int kRecHeader = 3;          // m, S1, S2 precede the gradient numerators in a record

struct TermD {
  int kind, col0, col1, n_basis;
  int th0, th1, th2, th3;  // EXP_SPLINE: th0 = coef_off
  int flags, pad;
  double p0, p1, p2;  // EXP_SPLINE: lo, hi, 1/dx of the spline coordinate
};

struct NormD {
  int n_pts, expo_theta, n_basis, coef_off, flags, pad;
  double expo_add, lo, hi;
  const double* tw;
  const double* lb;
  const double* l1;
  const double* us;
};

struct KArgs {
  const double* const* pe_cols;
  const double* const* inj_cols;
  const NormD* norms;
  double* partials;   // [n_scan_blocks][rec_stride]
  double* norm_out;   // [n_norms]
  double* logw_pe;    // only for the log-weight variant
  double* logw_inj;
  long long n_pe;     // samples per event
  long long n_inj;
  int n_ev, tiles_per_event, chunk_pe, n_inj_tiles, chunk_inj, n_norms;
  int n_terms, n_theta, kappa_col, rec_stride;
  TermD terms[GWI_MAX_TERMS];
  double derived[GWI_MAX_TERMS][kMaxDerived];
  double theta[GWI_MAX_THETA];
};
static_assert(sizeof(KArgs) <= 4096, "kernel argument block must fit the 4 KiB kernarg segment");

#define GWI_NEG_INF (-__builtin_huge_val())
#define GWI_POS_INF (__builtin_huge_val())

// ---- wave-level reductions (64 lanes) -----------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// ---- uniform cubic B-spline: 4 taps from the fractional knot coordinate --------------------
// Knots are uniform (interpolation.py:98-106), so with u = (x - lo) / dx the only non-zero bases
// at x are B_k..B_{k+3}, k = floor(u), and their values depend on t = u - k alone.  x == hi is
// assigned to the last interval with t = 1, which reproduces the (1/6, 2/3, 1/6) taps the
// reference's half-open order-1 pieces give there.
struct Taps {
  double b0, b1, b2, b3;
};
__device__ __forceinline__ Taps cubic_taps(double t) {
  const double omt = 1.0 - t;
  const double t2 = t * t, t3 = t2 * t;
  Taps r;
  r.b0 = omt * omt * omt * (1.0 / 6.0);
  r.b1 = (3.0 * t3 - 6.0 * t2 + 4.0) * (1.0 / 6.0);
  r.b2 = (-3.0 * t3 + 3.0 * t2 + 3.0 * t + 1.0) * (1.0 / 6.0);
  r.b3 = t3 * (1.0 / 6.0);
  return r;
}
__device__ __forceinline__ void spline_locate(double x, double lo, double inv_dx, int n_basis, int& k, double& t) {
  const double u = (x - lo) * inv_dx;
  const int last = n_basis - 4;  // index of the last interval
  int kk = (int)floor(u);
  kk = kk < 0 ? 0 : (kk > last ? last : kk);
  k = kk;
  t = u - (double)kk;
}

// ---- evaluation context --------------------------------------------------------------------
struct Ctx {
  const KArgs* a;
  const double* theta;          // LDS copy of theta
  double* gacc;                 // this wave's LDS gradient-numerator row [n_theta]
  const double* const* cols;    // column table of the sample set this workgroup scans
};

// ---- term library --------------------------------------------------------------------------
// Each term contributes l_t = log f_t(x; theta) without its sample-independent normaliser (those
// are added on the host: they cancel in log_l and its gradient), caches what the gradient needs in
// State, and after the weight w = e^{l-m} is known adds w * dl/dtheta into register accumulators
// (scalar hyper-parameters) or the wave's LDS row (spline coefficients).
template <int K>
struct Term;

// x^alpha on fixed [lo,hi] (distributions.py:100-119); log-normaliser is sample independent.
template <>
struct Term<GWI_TERM_POWERLAW> {
  struct State {
    double lx;
  };
  struct Acc {
    double g0;
  };
  __device__ static double eval(const TermD& t, const double*, const Ctx& c, long long idx, State& s) {
    s.lx = c.cols[t.col0][idx];
    return c.theta[t.th0] * s.lx;
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) { a.g0 += w * s.lx; }
  __device__ static void init(Acc& a) { a.g0 = 0; }
  __device__ static void rescale(Acc& a, double sc) { a.g0 *= sc; }
  __device__ static void flush(const TermD& t, const Ctx& c, Acc& a, int lane) {
    const double r = wave_sum(a.g0);
    if (lane == 0) c.gacc[t.th0] += r;
  }
};

// (1-lam) A x^alpha + lam Cn exp(-(x-mu)^2/(2 sig^2))  (parametric.py:49-53)
// derived: d0=log A, d1=dlogA/dalpha, d2=log Cn, d3=dlogCn/dmu, d4=dlogCn/dsig
template <>
struct Term<GWI_TERM_PLPEAK> {
  struct State {
    double da, dmu, dsg, dlam;
  };
  struct Acc {
    double g[4];
  };
  __device__ static double eval(const TermD& t, const double* d, const Ctx& c, long long idx, State& s) {
    const double x = c.cols[t.col0][idx];
    const double lx = c.cols[t.col1][idx];
    const double alpha = c.theta[t.th0], mu = c.theta[t.th1], sg = c.theta[t.th2], lam = c.theta[t.th3];
    const double inv_s2 = 1.0 / (sg * sg);
    const double dx = x - mu;
    const double e_pl = exp(alpha * lx + d[0]);
    const double e_tn = exp(-0.5 * dx * dx * inv_s2 + d[2]);
    const double P = (1.0 - lam) * e_pl, T = lam * e_tn;
    const double p = P + T;
    const double ip = 1.0 / p;
    s.da = P * (lx + d[1]) * ip;
    s.dmu = T * (dx * inv_s2 + d[3]) * ip;
    s.dsg = T * (dx * dx * inv_s2 / sg + d[4]) * ip;
    s.dlam = (e_tn - e_pl) * ip;
    return log(p);
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) {
    a.g[0] += w * s.da;
    a.g[1] += w * s.dmu;
    a.g[2] += w * s.dsg;
    a.g[3] += w * s.dlam;
  }
  __device__ static void init(Acc& a) { a.g[0] = a.g[1] = a.g[2] = a.g[3] = 0; }
  __device__ static void rescale(Acc& a, double sc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a.g[j] *= sc;
  }
  __device__ static void flush(const TermD& t, const Ctx& c, Acc& a, int lane) {
    const int th[4] = {t.th0, t.th1, t.th2, t.th3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const double r = wave_sum(a.g[j]);
      if (lane == 0) c.gacc[th[j]] += r;
    }
  }
};

// q^beta (1+beta)/(1 - r^(1+beta)), r = mmin/m1  (distributions.py:111-116 with low = mmin/m1)
template <>
struct Term<GWI_TERM_POWERLAW_RATIO> {
  struct State {
    double db;
  };
  struct Acc {
    double g0;
  };
  __device__ static double eval(const TermD& t, const double*, const Ctx& c, long long idx, State& s) {
    const double lq = c.cols[t.col0][idx];
    const double lr = t.p0 - c.cols[t.col1][idx];  // log(mmin/m1) <= 0 for every non-excluded sample
    const double beta = c.theta[t.th0];
    const double b1 = 1.0 + beta;
    if (b1 == 0.0) {  // alpha == -1 branch of the reference: 1/log(high/low)
      s.db = lq - 0.5 * lr;
      return -lq - log(-lr);
    }
    const double em1 = expm1(b1 * lr);  // r^(1+beta) - 1
    const double denom = -em1;          // 1 - r^(1+beta)
    s.db = lq + 1.0 / b1 + (em1 + 1.0) * lr / denom;
    return beta * lq + log(b1 / denom);
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) { a.g0 += w * s.db; }
  __device__ static void init(Acc& a) { a.g0 = 0; }
  __device__ static void rescale(Acc& a, double sc) { a.g0 *= sc; }
  __device__ static void flush(const TermD& t, const Ctx& c, Acc& a, int lane) {
    const double r = wave_sum(a.g0);
    if (lane == 0) c.gacc[t.th0] += r;
  }
};

// Beta(a; alpha, beta): (alpha-1) log a + (beta-1) log(1-a) - betaln  (distributions.py:160-161)
template <>
struct Term<GWI_TERM_BETA> {
  struct State {
    double la, l1;
  };
  struct Acc {
    double g[2];
  };
  __device__ static double eval(const TermD& t, const double*, const Ctx& c, long long idx, State& s) {
    s.la = c.cols[t.col0][idx];
    s.l1 = c.cols[t.col1][idx];
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
  __device__ static void flush(const TermD& t, const Ctx& c, Acc& a, int lane) {
    const double r0 = wave_sum(a.g[0]), r1 = wave_sum(a.g[1]);
    if (lane == 0) {
      c.gacc[t.th0] += r0;
      c.gacc[t.th1] += r1;
    }
  }
};

// (1-xi)/2 + xi Cn exp(-(ct-1)^2/(2 sig^2))  (parametric.py:84-86); derived: d0=log Cn, d1=dlogCn/dsig
template <>
struct Term<GWI_TERM_TILT_MIXTURE> {
  struct State {
    double dxi, dsg;
  };
  struct Acc {
    double g[2];
  };
  __device__ static double eval(const TermD& t, const double* d, const Ctx& c, long long idx, State& s) {
    const double ct = c.cols[t.col0][idx];
    const double xi = c.theta[t.th0], sg = c.theta[t.th1];
    const double inv_s2 = 1.0 / (sg * sg);
    const double dx = ct - 1.0;
    const double e_tn = exp(-0.5 * dx * dx * inv_s2 + d[0]);
    const double p = 0.5 * (1.0 - xi) + xi * e_tn;
    const double ip = 1.0 / p;
    s.dxi = (e_tn - 0.5) * ip;
    s.dsg = xi * e_tn * (dx * dx * inv_s2 / sg + d[1]) * ip;
    return log(p);
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
  __device__ static void flush(const TermD& t, const Ctx& c, Acc& a, int lane) {
    const double r0 = wave_sum(a.g[0]), r1 = wave_sum(a.g[1]);
    if (lane == 0) {
      c.gacc[t.th0] += r0;
      c.gacc[t.th1] += r1;
    }
  }
};

// TN(x; mu, sig, lo, hi) alone (distributions.py:136-143); normaliser is sample independent.
template <>
struct Term<GWI_TERM_TRUNCNORM> {
  struct State {
    double dmu, dsg;
  };
  struct Acc {
    double g[2];
  };
  __device__ static double eval(const TermD& t, const double*, const Ctx& c, long long idx, State& s) {
    const double x = c.cols[t.col0][idx];
    const double mu = c.theta[t.th0], sg = c.theta[t.th1];
    const double inv_s2 = 1.0 / (sg * sg);
    const double dx = x - mu;
    s.dmu = dx * inv_s2;
    s.dsg = dx * dx * inv_s2 / sg;
    return -0.5 * dx * dx * inv_s2;
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
  __device__ static void flush(const TermD& t, const Ctx& c, Acc& a, int lane) {
    const double r0 = wave_sum(a.g[0]), r1 = wave_sum(a.g[1]);
    if (lane == 0) {
      c.gacc[t.th0] += r0;
      c.gacc[t.th1] += r1;
    }
  }
};

// (1+z)^(lamb-1) (parametric.py:126-127); dVc/dz is in kappa, the grid normaliser on the host side.
template <>
struct Term<GWI_TERM_POWERLAW_REDSHIFT> {
  struct State {
    double l1pz;
  };
  struct Acc {
    double g0;
  };
  __device__ static double eval(const TermD& t, const double*, const Ctx& c, long long idx, State& s) {
    s.l1pz = c.cols[t.col0][idx];
    return (c.theta[t.th0] - 1.0) * s.l1pz;
  }
  __device__ static void accumulate(const TermD&, const Ctx&, double w, const State& s, Acc& a) { a.g0 += w * s.l1pz; }
  __device__ static void init(Acc& a) { a.g0 = 0; }
  __device__ static void rescale(Acc& a, double sc) { a.g0 *= sc; }
  __device__ static void flush(const TermD& t, const Ctx& c, Acc& a, int lane) {
    const double r = wave_sum(a.g0);
    if (lane == 0) c.gacc[t.th0] += r;
  }
};

// exp(sum_k c_k B_k(x))  (interpolation.py:381-394 / :293-304 + exp, spline_perturbation.py:352)
// p0 = lo, p1 = hi, p2 = 1/dx of the spline coordinate; th0 = theta offset of c_0.
template <>
struct Term<GWI_TERM_EXP_SPLINE> {
  struct State {
    double t;
    int k;  // -1: outside the domain of a zero-outside basis (factor 1, no gradient)
  };
  struct Acc {};
  __device__ static double eval(const TermD& t, const double*, const Ctx& c, long long idx, State& s) {
    const double x = c.cols[t.col0][idx];
    int k;
    double tt;
    spline_locate(x, t.p0, t.p2, t.n_basis, k, tt);
    const double* cf = c.theta + t.th0 + k;
    const Taps b = cubic_taps(tt);
    double v = cf[0] * b.b0 + cf[1] * b.b1 + cf[2] * b.b2 + cf[3] * b.b3;
    if (t.flags & GWI_SPLINE_OUTSIDE_ZERO_EXPONENT) {
      // BSpline / LogXBSpline bases are 0 outside the closed domain (interpolation.py:175)
      if (!((x >= t.p0) && (x <= t.p1))) {
        k = -1;
        v = 0.0;
      }
    }
    s.t = tt;
    s.k = k;
    return v;
  }
  __device__ static void accumulate(const TermD& t, const Ctx& c, double w, const State& s, Acc&) {
    if (s.k >= 0 && w != 0.0) {
      const Taps b = cubic_taps(s.t);
      double* g = c.gacc + t.th0 + s.k;
      unsafeAtomicAdd(g + 0, w * b.b0);
      unsafeAtomicAdd(g + 1, w * b.b1);
      unsafeAtomicAdd(g + 2, w * b.b2);
      unsafeAtomicAdd(g + 3, w * b.b3);
    }
  }
  __device__ static void init(Acc&) {}
  __device__ static void rescale(Acc&, double) {}
  __device__ static void flush(const TermD&, const Ctx&, Acc&, int) {}
};

// ---- compile-time chain of terms --------------------------------------------------------------
template <int... Ks>
struct Chain;
template <>
struct Chain<> {
  __device__ void init() {}
  __device__ double eval(int, const Ctx&, long long) { return 0.0; }
  __device__ void accumulate(int, const Ctx&, double) {}
  __device__ void rescale(double) {}
  __device__ void flush(int, const Ctx&, int) {}
};
template <int K, int... Rest>
struct Chain<K, Rest...> {
  typename Term<K>::State st;
  typename Term<K>::Acc acc;
  Chain<Rest...> rest;
  __device__ void init() {
    Term<K>::init(acc);
    rest.init();
  }
  __device__ double eval(int ti, const Ctx& c, long long idx) {
    return Term<K>::eval(c.a->terms[ti], c.a->derived[ti], c, idx, st) + rest.eval(ti + 1, c, idx);
  }
  __device__ void accumulate(int ti, const Ctx& c, double w) {
    Term<K>::accumulate(c.a->terms[ti], c, w, st, acc);
    rest.accumulate(ti + 1, c, w);
  }
  __device__ void rescale(double sc) {
    Term<K>::rescale(acc, sc);
    rest.rescale(sc);
  }
  __device__ void flush(int ti, const Ctx& c, int lane) {
    Term<K>::flush(c.a->terms[ti], c, acc, lane);
    rest.flush(ti + 1, c, lane);
  }
};

// ---- grid normaliser workgroup (interpolation.py:280-291, parametric.py:123-124,
//      spline_perturbation.py:323-336): Z = sum_g tw_g exp(lb_g + (theta+add) l1_g + spline(us_g))
__device__ inline void norm_block(const KArgs& a, int j, const double* s_theta, double* s_red) {
  const NormD nd = a.norms[j];
  const int tid = threadIdx.x;
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
      if ((nd.flags & GWI_SPLINE_OUTSIDE_ZERO_EXPONENT) && !((x >= nd.lo) && (x <= nd.hi))) v = 0.0;
      e += v;
    }
    if (tw != 0.0) acc += tw * exp(e);
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) s_red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) a.norm_out[j] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// ---- the scan kernel -----------------------------------------------------------------------------
// grid = n_ev*tiles_per_event PE workgroups + n_inj_tiles injection workgroups + n_norms normaliser
// workgroups.  A PE workgroup owns `chunk_pe` consecutive samples of ONE event, so its record
// belongs to that event's logsumexp; an injection workgroup owns `chunk_inj` consecutive
// injections.  Loads are coalesced: lane i of a wave reads element base+i of each column.
template <bool WRITE_LOGW, int... Ks>
__global__ __launch_bounds__(kBlock) void scan_kernel(const KArgs a) {
  __shared__ double s_theta[GWI_MAX_THETA];
  __shared__ double s_gacc[kWaves][GWI_MAX_THETA];
  __shared__ double s_wrec[kWaves][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int n_pe_blocks = a.n_ev * a.tiles_per_event;
  const int n_scan_blocks = n_pe_blocks + a.n_inj_tiles;

  for (int p = tid; p < a.n_theta; p += kBlock) s_theta[p] = a.theta[p];
  for (int p = tid; p < kWaves * GWI_MAX_THETA; p += kBlock) (&s_gacc[0][0])[p] = 0.0;
  __syncthreads();

  if (b >= n_scan_blocks) {
    norm_block(a, b - n_scan_blocks, s_theta, &s_wrec[0][0]);
    return;
  }

  long long start, end, base;
  Ctx ctx;
  ctx.a = &a;
  ctx.theta = s_theta;
  ctx.gacc = s_gacc[wave];
  double* logw;
  if (b < n_pe_blocks) {
    const int e = b / a.tiles_per_event;
    const int t = b - e * a.tiles_per_event;
    start = (long long)t * a.chunk_pe;
    end = start + a.chunk_pe < a.n_pe ? start + a.chunk_pe : a.n_pe;
    base = (long long)e * a.n_pe;
    ctx.cols = a.pe_cols;
    logw = a.logw_pe;
  } else {
    const int t = b - n_pe_blocks;
    start = (long long)t * a.chunk_inj;
    end = start + a.chunk_inj < a.n_inj ? start + a.chunk_inj : a.n_inj;
    base = 0;
    ctx.cols = a.inj_cols;
    logw = a.logw_inj;
  }
  const double* kappa_col = ctx.cols[a.kappa_col];

  double m = GWI_NEG_INF, s1 = 0.0, s2 = 0.0;
  Chain<Ks...> chain;
  chain.init();

  // the loop condition is wave-uniform: i - lane is the same for every lane of a wave
  for (long long i = start + tid; i - lane < end; i += kBlock) {
    const bool valid = i < end;
    const long long idx = base + (valid ? i : end - 1);
    double ell = kappa_col[idx] + chain.eval(0, ctx, idx);
    // NaN or +inf weights count as zero (tests/inference_test.py:172, 260)
    if (!valid || !(ell < GWI_POS_INF)) ell = GWI_NEG_INF;
    if (WRITE_LOGW) {
      if (valid) logw[idx] = ell;
      continue;
    }
    const double mx = wave_max(ell);
    if (mx > m) {  // wave-uniform: move every running sum to the new reference exponent
      const double sc = exp(m - mx);
      s1 *= sc;
      s2 *= sc * sc;
      chain.rescale(sc);
      for (int p = lane; p < a.n_theta; p += 64) ctx.gacc[p] *= sc;
      m = mx;
    }
    const double w = (ell == GWI_NEG_INF) ? 0.0 : exp(ell - m);
    s1 += w;
    s2 += w * w;
    chain.accumulate(0, ctx, w);
  }
  if (WRITE_LOGW) return;

  chain.flush(0, ctx, lane);
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  if (lane == 0) {
    s_wrec[wave][0] = m;
    s_wrec[wave][1] = s1;
    s_wrec[wave][2] = s2;
  }
  __syncthreads();
  double M = s_wrec[0][0];
#pragma unroll
  for (int w_ = 1; w_ < kWaves; ++w_) M = fmax(M, s_wrec[w_][0]);
  double f[kWaves];
#pragma unroll
  for (int w_ = 0; w_ < kWaves; ++w_) f[w_] = (s_wrec[w_][0] == GWI_NEG_INF) ? 0.0 : exp(s_wrec[w_][0] - M);
  double* out = a.partials + (long long)b * a.rec_stride;
  if (tid == 0) {
    double S1 = 0.0, S2 = 0.0;
#pragma unroll
    for (int w_ = 0; w_ < kWaves; ++w_) {
      S1 += f[w_] * s_wrec[w_][1];
      S2 += f[w_] * f[w_] * s_wrec[w_][2];
    }
    out[0] = M;
    out[1] = S1;
    out[2] = S2;
  }
  for (int p = tid; p < a.n_theta; p += kBlock) {
    double g = 0.0;
#pragma unroll
    for (int w_ = 0; w_ < kWaves; ++w_) g += f[w_] * s_gacc[w_][p];
    out[kRecHeader + p] = g;
  }
}

// ---- stage 2: combine the tile records of one event (blocks 0..n_ev-1) or of the injection set
//      (block n_ev) with a common reference exponent ----------------------------------------------
struct CombineArgs {
  const double* partials;
  double* ev_out;     // [n_ev][4]: logsumexp (= log sum_j w_ij, no -log N_pe), log n_eff, variance, S1
  double* ev_grad;    // [n_ev][n_theta]: G_p / S1
  double* inj_out;    // [4]: M, S1, S2
  double* inj_grad;   // [n_theta]: G_p relative to M
  int n_ev, tiles_per_event, n_inj_tiles, n_theta, rec_stride;
  double n_pe;
};

__global__ __launch_bounds__(kBlock) void combine_kernel(const CombineArgs a) {
  __shared__ double s_red[kWaves];
  __shared__ double s_M;
  const int tid = threadIdx.x;
  const int e = blockIdx.x;
  const bool is_inj = e == a.n_ev;
  const int n_tiles = is_inj ? a.n_inj_tiles : a.tiles_per_event;
  const long long first = is_inj ? (long long)a.n_ev * a.tiles_per_event : (long long)e * a.tiles_per_event;
  const double* rec = a.partials + first * a.rec_stride;

  double mx = GWI_NEG_INF;
  for (int t = tid; t < n_tiles; t += kBlock) mx = fmax(mx, rec[(long long)t * a.rec_stride]);
  mx = wave_max(mx);
  if ((tid & 63) == 0) s_red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) s_M = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
  __syncthreads();
  const double M = s_M;

  // S1, S2 by thread 0 in tile order (deterministic); gradient numerators by thread p
  double S1 = 0.0, S2 = 0.0;
  if (tid == 0) {
    for (int t = 0; t < n_tiles; ++t) {
      const double* r = rec + (long long)t * a.rec_stride;
      const double f = (r[0] == GWI_NEG_INF) ? 0.0 : exp(r[0] - M);
      S1 += f * r[1];
      S2 += f * f * r[2];
    }
    s_red[0] = S1;
  }
  __syncthreads();
  S1 = s_red[0];
  for (int p = tid; p < a.n_theta; p += kBlock) {
    double g = 0.0;
    for (int t = 0; t < n_tiles; ++t) {
      const double* r = rec + (long long)t * a.rec_stride;
      const double f = (r[0] == GWI_NEG_INF) ? 0.0 : exp(r[0] - M);
      g += f * r[kRecHeader + p];
    }
    if (is_inj)
      a.inj_grad[p] = g;
    else
      a.ev_grad[(long long)e * a.n_theta + p] = S1 > 0.0 ? g / S1 : 0.0;
  }
  if (tid == 0) {
    if (is_inj) {
      a.inj_out[0] = M;
      a.inj_out[1] = S1;
      a.inj_out[2] = S2;
    } else {
      // analysis.py:78-87: logBF = logsumexp - log N_pe (constant added on the host),
      // log n_eff = 2 logsumexp(l) - logsumexp(2 l), variance = 1/n_eff - 1/N_pe
      const double log_s1 = log(S1);
      const double log_neff = 2.0 * log_s1 - log(S2);
      double* o = a.ev_out + (long long)e * 4;
      o[0] = log_s1 + M;
      o[1] = log_neff;
      o[2] = 1.0 / exp(log_neff) - 1.0 / a.n_pe;
      o[3] = S1;
    }
  }
}

// ---- stage 3: reduce over events and publish this device's record to pinned host memory -----------
// record layout (doubles): see RecordLayout in gwi_engine.hip
struct FinalArgs {
  const double* ev_out;
  const double* ev_grad;
  const double* inj_out;
  const double* inj_grad;
  const double* norm_out;
  double* record;       // device-visible pinned host buffer
  double* ev_host;      // [3][n_ev] pinned host: logsumexp, log n_eff, variance
  int n_ev, n_theta, n_norms;
  unsigned long long seq;  // written last to record[0] as a completion stamp
};

__global__ __launch_bounds__(kBlock) void final_kernel(const FinalArgs a) {
  __shared__ double s_sum[kWaves], s_var[kWaves], s_min[kWaves];
  const int tid = threadIdx.x;
  double sum = 0.0, var = 0.0, mn = GWI_POS_INF;
  for (int e = tid; e < a.n_ev; e += kBlock) {
    const double* o = a.ev_out + (long long)e * 4;
    sum += o[0];
    var += o[2];
    // jnp.min(jnp.nan_to_num(logn_effs)) (analysis.py:295): NaN -> 0, +-inf -> +-max double
    double le = o[1];
    if (le != le) le = 0.0;
    le = fmin(fmax(le, -1.7976931348623157e308), 1.7976931348623157e308);
    mn = fmin(mn, le);
    a.ev_host[e] = o[0];
    a.ev_host[a.n_ev + e] = o[1];
    a.ev_host[2 * a.n_ev + e] = o[2];
  }
  sum = wave_sum(sum);
  var = wave_sum(var);
  mn = -wave_max(-mn);
  if ((tid & 63) == 0) {
    s_sum[tid >> 6] = sum;
    s_var[tid >> 6] = var;
    s_min[tid >> 6] = mn;
  }
  double* r = a.record;
  const int off_norm = 8, off_gpe = off_norm + a.n_norms, off_ginj = off_gpe + a.n_theta;
  for (int p = tid; p < a.n_theta; p += kBlock) {
    double g = 0.0;
    for (int e = 0; e < a.n_ev; ++e) g += a.ev_grad[(long long)e * a.n_theta + p];
    r[off_gpe + p] = g;
    r[off_ginj + p] = a.inj_grad[p];
  }
  for (int j = tid; j < a.n_norms; j += kBlock) r[off_norm + j] = a.norm_out[j];
  __syncthreads();
  if (tid == 0) {
    r[1] = (s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]);
    r[2] = (s_var[0] + s_var[1]) + (s_var[2] + s_var[3]);
    r[3] = fmin(fmin(s_min[0], s_min[1]), fmin(s_min[2], s_min[3]));
    r[4] = a.inj_out[0];
    r[5] = a.inj_out[1];
    r[6] = a.inj_out[2];
    r[7] = (double)a.n_ev;
  }
  __syncthreads();
  __threadfence_system();
  if (tid == 0) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(r), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

}  // namespace gwi
