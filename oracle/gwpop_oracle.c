/* gwpop_oracle.c -- CPU ORACLE / BASELINE, TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C (OpenMP) restatement of the hierarchical population likelihood hot path for the flat model
 * description of include/gwi_engine.h: per-sample densities (gwinferno/distributions.py:100-162,
 * models/parametric/parametric.py:27-145), uniform cubic B-spline projection
 * (gwinferno/interpolation.py:98-149, 293-317, 381-394), grid normalisers (:280-291,
 * parametric.py:123-124, spline_perturbation.py:323-336), importance-sampling reductions
 * (pipeline/analysis.py:50-136) and the assembly of log_l with its cuts (:259-319), value AND gradient.
 * It is pinned against the golden vectors of the unmodified reference through
 * tests/test_c_oracle.py, and is what bench.py times as `cpu_baseline` (kind "port") on all host cores.
 * Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may load it; the product never does.
 *
 * Scalar straight-line code on purpose: one sample at a time, libm transcendentals, a running
 * maximum with rescaling -- no relation to the GPU kernels' organisation.  Parallelism (round 4): ONE OpenMP region over
 * blocks of GWO_BLOCK consecutive samples -- (event, block) pairs and injection blocks alike, so that a 69-event catalog
 * keeps 128 host threads busy -- each block reduced on its own and the blocks merged afterwards in block order (the result
 * does not depend on the thread count).  A sample's d(log w)/d theta is kept as a short list of (index, value) pairs: a
 * B-spline term touches four coefficients, not all n_theta.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/gwi_engine.h"

#define NEG_BIG (-1.7976931348623157e308)

#define GWO_BLOCK 512      /* samples per work unit; fixed, so that the summation order does not depend on the thread count */
#define GWO_MAX_PAIRS 160  /* (index, value) pairs one sample can produce: 12 terms x (2 grid nodes x 4 taps) at most, + scalars */

typedef struct {
  double m, s1, s2;
  double* g; /* [n_theta] sum w dl/dtheta, in units of e^m */
  double* h; /* [n_theta] sum w^2 dl/dtheta, in units of e^2m (analysis.py:270-271: the gradient of n_eff needs it), or NULL */
} acc_t;

typedef struct {
  int n;
  int idx[GWO_MAX_PAIRS];
  double val[GWO_MAX_PAIRS];
} pairs_t;

static inline void push(pairs_t* d, int i, double v) {
  d->idx[d->n] = i;
  d->val[d->n] = v;
  d->n++;
}

static void acc_init(acc_t* a, double* g, double* h, int n_theta) {
  a->m = -INFINITY;
  a->s1 = a->s2 = 0.0;
  a->g = g;
  a->h = h;
  memset(g, 0, sizeof(double) * n_theta);
  if (h) memset(h, 0, sizeof(double) * n_theta);
}

static void acc_rescale(acc_t* a, double new_m, int n_theta) {
  if (a->m == -INFINITY) {
    a->m = new_m;
    return;
  }
  const double sc = exp(a->m - new_m);
  a->s1 *= sc;
  a->s2 *= sc * sc;
  for (int p = 0; p < n_theta; ++p) a->g[p] *= sc;
  if (a->h)
    for (int p = 0; p < n_theta; ++p) a->h[p] *= sc * sc;
  a->m = new_m;
}

static void acc_merge(acc_t* dst, const acc_t* src, int n_theta) {
  if (src->m == -INFINITY) return;
  if (src->m > dst->m) acc_rescale(dst, src->m, n_theta);
  const double f = exp(src->m - dst->m);
  dst->s1 += f * src->s1;
  dst->s2 += f * f * src->s2;
  for (int p = 0; p < n_theta; ++p) dst->g[p] += f * src->g[p];
  if (dst->h)
    for (int p = 0; p < n_theta; ++p) dst->h[p] += f * f * src->h[p];
}

static void pl_lognorm(double alpha, double lo, double hi, double* la, double* dla) {
  const double a1 = 1.0 + alpha, llo = log(lo), lhi = log(hi);
  if (a1 == 0.0) {
    *la = -log(lhi - llo);
    *dla = -0.5 * (lhi + llo);
    return;
  }
  const double ph = pow(hi, a1), pw = pow(lo, a1);
  *la = log(a1 / (ph - pw)); /* distributions.py:115 */
  *dla = 1.0 / a1 - (ph * lhi - pw * llo) / (ph - pw);
}

static void tn_lognorm(double mu, double sg, double lo, double hi, double* lc, double* dmu, double* dsg) {
  const double r2 = sqrt(2.0), a = (lo - mu) / sg, b = (hi - mu) / sg;
  const double dphi = 0.5 * (1.0 + erf(b / r2)) - 0.5 * (1.0 + erf(a / r2)); /* distributions.py:138-140 */
  const double c = 1.0 / sqrt(2.0 * M_PI);
  const double pa = exp(-0.5 * a * a) * c, pb = exp(-0.5 * b * b) * c;
  *lc = -log(sg) - 0.5 * log(2.0 * M_PI) - log(dphi);
  *dmu = (pb - pa) / (sg * dphi);
  *dsg = -1.0 / sg + (b * pb - a * pa) / (sg * dphi);
}

static void taps(double t, double b[4]) {
  const double o = 1.0 - t;
  b[0] = o * o * o / 6.0;
  b[1] = (3.0 * t * t * t - 6.0 * t * t + 4.0) / 6.0;
  b[2] = (-3.0 * t * t * t + 3.0 * t * t + 3.0 * t + 1.0) / 6.0;
  b[3] = t * t * t / 6.0;
}

static int locate(double x, double lo, double hi, int n_basis, double* t) {
  const int n_int = n_basis - 3;
  const double u = (x - lo) * ((double)n_int / (hi - lo));
  int k = (int)floor(u);
  if (k < 0) k = 0;
  if (k > n_int - 1) k = n_int - 1;
  *t = u - (double)k;
  return k;
}

/* log weight (without sample-independent constants) and d(log weight)/d theta of ONE sample */
static double sample_logw(const gwi_spec* sp, const double* const* cols, int64_t idx, const double* th, const double (*der)[8], pairs_t* d) {
  d->n = 0;
  double ell = cols[sp->kappa_col][idx];
  for (int t = 0; t < sp->n_terms && ell > -INFINITY; ++t) {
    const gwi_term* tm = &sp->terms[t];
    const double x0_raw = cols[tm->cols[0]][idx];
    const double x0 = tm->kind == GWI_TERM_PLPEAK ? exp(x0_raw) : x0_raw;
    switch (tm->kind) {
      case GWI_TERM_POWERLAW:
        ell += th[tm->theta[0]] * x0;
        push(d, tm->theta[0], x0);
        break;
      case GWI_TERM_PLPEAK: { /* one column: log x (include/gwi_engine.h); x0 below is x = exp(log x) */
        const double lx = x0_raw;
        const double al = th[tm->theta[0]], mu = th[tm->theta[1]], sg = th[tm->theta[2]], lam = th[tm->theta[3]];
        const double epl = exp(al * lx + der[t][0]), etn = exp(-0.5 * (x0 - mu) * (x0 - mu) / (sg * sg) + der[t][2]);
        const double P = (1.0 - lam) * epl, T = lam * etn, p = P + T;
        ell += log(p);
        push(d, tm->theta[0], P * (lx + der[t][1]) / p);
        push(d, tm->theta[1], T * ((x0 - mu) / (sg * sg) + der[t][3]) / p);
        push(d, tm->theta[2], T * ((x0 - mu) * (x0 - mu) / (sg * sg * sg) + der[t][4]) / p);
        push(d, tm->theta[3], (etn - epl) / p);
        break;
      }
      case GWI_TERM_SMOOTH: { /* distributions.py:16-21: 1/(1+exp(d/y + d/(y-d))) for every y = x - xmin */
        const double dl = th[tm->theta[0]], y = x0;
        const double S = 1.0 / (1.0 + exp(dl / y + dl / (y - dl)));
        ell += log(S);
        push(d, tm->theta[0], -(1.0 - S) * (1.0 / y + y / ((y - dl) * (y - dl))));
        break;
      }
      case GWI_TERM_PLPEAK_SMOOTH: { /* parametric.py:49-53 with delta; coef_off = theta index of delta */
        const double lx = cols[tm->cols[1]][idx];
        const double al = th[tm->theta[0]], mu = th[tm->theta[1]], sg = th[tm->theta[2]], lam = th[tm->theta[3]], dl = th[tm->coef_off];
        const double y = x0 - tm->p[0];
        const double S = 1.0 / (1.0 + exp(dl / y + dl / (y - dl)));
        const double epl = exp(al * lx + der[t][0]) * S, etn = exp(-0.5 * (x0 - mu) * (x0 - mu) / (sg * sg) + der[t][2]);
        const double P = (1.0 - lam) * epl, T = lam * etn, p = P + T;
        ell += log(p);
        push(d, tm->theta[0], P * (lx + der[t][1]) / p);
        push(d, tm->theta[1], T * ((x0 - mu) / (sg * sg) + der[t][3]) / p);
        push(d, tm->theta[2], T * ((x0 - mu) * (x0 - mu) / (sg * sg * sg) + der[t][4]) / p);
        push(d, tm->theta[3], (etn - epl) / p);
        push(d, tm->coef_off, P * (-(1.0 - S) * (1.0 / y + y / ((y - dl) * (y - dl)))) / p);
        break;
      }
      case GWI_TERM_POWERLAW_RATIO: {
        /* cols[1] = log m1.  Under GWI_RATIO_LOGM_FROM_SPLINE it is the coordinate column of the model's m1 spline, which on
         * the HOST side -- what this checker is handed -- holds that spline's coordinate x = log m1 itself (only the device
         * copy is converted to knot coordinates, gwi_engine.hip: spline_knot_kernel): the same read either way. */
        const double lr = tm->p[0] - cols[tm->cols[1]][idx], beta = th[tm->theta[0]], b1 = 1.0 + beta;
        if (b1 == 0.0) {
          ell += -x0 - log(-lr);
          push(d, tm->theta[0], x0 - 0.5 * lr);
        } else {
          const double E = exp(b1 * lr);
          ell += beta * x0 + log(b1 / (1.0 - E));
          push(d, tm->theta[0], x0 + 1.0 / b1 + E * lr / (1.0 - E));
        }
        break;
      }
      case GWI_TERM_BETA: {
        const double l1 = cols[tm->cols[1]][idx];
        ell += (th[tm->theta[0]] - 1.0) * x0 + (th[tm->theta[1]] - 1.0) * l1;
        push(d, tm->theta[0], x0);
        push(d, tm->theta[1], l1);
        break;
      }
      case GWI_TERM_TILT_MIXTURE: {
        const double xi = th[tm->theta[0]], sg = th[tm->theta[1]];
        const double e = exp(-0.5 * (x0 - 1.0) * (x0 - 1.0) / (sg * sg) + der[t][0]);
        const double p = 0.5 * (1.0 - xi) + xi * e;
        ell += log(p);
        push(d, tm->theta[0], (e - 0.5) / p);
        push(d, tm->theta[1], xi * e * ((x0 - 1.0) * (x0 - 1.0) / (sg * sg * sg) + der[t][1]) / p);
        break;
      }
      case GWI_TERM_TILT_JOINT: {
        const double x1 = cols[tm->cols[1]][idx], xi = th[tm->theta[0]], sg = th[tm->theta[1]];
        const double r2 = (x0 - 1.0) * (x0 - 1.0) + (x1 - 1.0) * (x1 - 1.0);
        const double A = exp(-0.5 * r2 / (sg * sg) + 2.0 * der[t][0]);
        const double p = 0.25 * (1.0 - xi) + xi * A;
        ell += log(p);
        push(d, tm->theta[0], (A - 0.25) / p);
        push(d, tm->theta[1], xi * A * (r2 / (sg * sg * sg) + 2.0 * der[t][1]) / p);
        break;
      }
      case GWI_TERM_TRUNCNORM: {
        const double mu = th[tm->theta[0]], sg = th[tm->theta[1]];
        ell += -0.5 * (x0 - mu) * (x0 - mu) / (sg * sg);
        push(d, tm->theta[0], (x0 - mu) / (sg * sg));
        push(d, tm->theta[1], (x0 - mu) * (x0 - mu) / (sg * sg * sg));
        break;
      }
      case GWI_TERM_POWERLAW_REDSHIFT:
        ell += (th[tm->theta[0]] - 1.0) * x0;
        push(d, tm->theta[0], x0);
        break;
      case GWI_TERM_POWERLAW_BOUNDS: { /* numpyro_distributions.py:127-136: -inf outside [minimum, maximum] (bounds included) */
        const double xv = cols[tm->cols[1]][idx];
        if ((xv < th[tm->theta[1]]) || (xv > th[tm->theta[2]])) {
          ell = -INFINITY;
          break;
        }
        ell += th[tm->theta[0]] * x0;
        push(d, tm->theta[0], x0);
        break;
      }
      case GWI_TERM_EXP_SPLINE_LERP: { /* numpyro_distributions.py:273, :296-301: interp(value, grid, cs . grid_dmat) */
        const gwi_norm* nm = &sp->norms[tm->norm];
        int j = (int)x0;
        if (j < 0) j = 0;
        if (j > nm->n_pts - 2) j = nm->n_pts - 2;
        const double f = x0 - (double)j;
        const double wts[2] = {1.0 - f, f};
        double val = 0.0;
        for (int e = 0; e < 2 && ell > -INFINITY; ++e) {
          if (wts[e] == 0.0) continue;
          const double sx = nm->us[j + e];
          double tt, b[4];
          const int inside = (sx >= tm->p[0]) && (sx <= tm->p[1]);
          if (!inside) {
            if (!(tm->flags & GWI_SPLINE_OUTSIDE_ZERO_EXPONENT)) ell = -INFINITY; /* log-Y basis: lpdf = -inf at that grid point */
            continue;
          }
          const int k = locate(sx, tm->p[0], tm->p[1], tm->n_basis, &tt);
          taps(tt, b);
          const double* c = th + tm->coef_off + k;
          val += wts[e] * (c[0] * b[0] + c[1] * b[1] + c[2] * b[2] + c[3] * b[3]);
          for (int q = 0; q < 4; ++q) push(d, tm->coef_off + k + q, wts[e] * b[q]);
        }
        if (ell > -INFINITY) ell += val;
        break;
      }
      case GWI_TERM_EXP_SPLINE:
      case GWI_TERM_LINEAR_SPLINE: {
        double tt, b[4];
        const int inside = (x0 >= tm->p[0]) && (x0 <= tm->p[1]);
        const int k = locate(x0, tm->p[0], tm->p[1], tm->n_basis, &tt);
        taps(tt, b);
        const double* c = th + tm->coef_off + k;
        const double v = c[0] * b[0] + c[1] * b[1] + c[2] * b[2] + c[3] * b[3];
        if (tm->kind == GWI_TERM_EXP_SPLINE) {
          if ((tm->flags & GWI_SPLINE_OUTSIDE_ZERO_EXPONENT) && !inside) break; /* basis 0 outside: factor 1 */
          ell += v;
          for (int j = 0; j < 4; ++j) push(d, tm->coef_off + k + j, b[j]);
        } else {
          if (!inside || !(v > 0.0)) {
            ell = -INFINITY;
            break;
          }
          ell += log(v);
          for (int j = 0; j < 4; ++j) push(d, tm->coef_off + k + j, b[j] / v);
        }
        break;
      }
      default: ell = NAN;
    }
  }
  if (!(ell < INFINITY)) ell = -INFINITY; /* NaN / +inf weights count as zero (tests/inference_test.py:172) */
  return ell;
}

/* one block of consecutive samples reduced on its own (running maximum with rescaling) */
static void scan_range(const gwi_spec* sp, const double* const* cols, int64_t lo, int64_t hi, const double* th, const double (*der)[8], acc_t* out) {
  const int n_theta = sp->n_theta;
  pairs_t d;
  for (int64_t i = lo; i < hi; ++i) {
    const double ell = sample_logw(sp, cols, i, th, der, &d);
    if (ell == -INFINITY) continue;
    if (ell > out->m) acc_rescale(out, ell, n_theta);
    const double w = exp(ell - out->m);
    out->s1 += w;
    out->s2 += w * w;
    for (int k = 0; k < d.n; ++k) out->g[d.idx[k]] += w * d.val[k];
    if (out->h)
      for (int k = 0; k < d.n; ++k) out->h[d.idx[k]] += w * w * d.val[k];
  }
}

static double grid_norm(const gwi_norm* nm, const double* th) {
  double z = 0.0;
  for (int g = 0; g < nm->n_pts; ++g) {
    double e = nm->lb ? nm->lb[g] : 0.0;
    if (nm->expo_theta >= 0) e += (th[nm->expo_theta] + nm->expo_add) * nm->l1[g];
    if (nm->n_basis > 0) {
      double tt, b[4];
      const double x = nm->us[g];
      const int k = locate(x, nm->lo, nm->hi, nm->n_basis, &tt);
      taps(tt, b);
      const double* c = th + nm->coef_off + k;
      double v = c[0] * b[0] + c[1] * b[1] + c[2] * b[2] + c[3] * b[3];
      if ((nm->spline_flags & (GWI_SPLINE_OUTSIDE_ZERO_EXPONENT | GWI_NORM_LINEAR_SPLINE)) && !((x >= nm->lo) && (x <= nm->hi))) v = 0.0;
      if (nm->spline_flags & GWI_NORM_LINEAR_SPLINE) {
        z += nm->tw[g] * v;
        continue;
      }
      e += v;
    }
    if (nm->tw[g] != 0.0) z += nm->tw[g] * exp(e);
  }
  return z;
}

/* One value-and-gradient evaluation.  Outputs as gwi_eval (include/gwi_engine.h). */
int gwo_eval(const gwi_spec* sp, const double* const* pe_cols, int64_t n_ev, int64_t n_pe, const double* const* inj_cols, int64_t n_inj,
             const double* th, const gwi_options* opt, gwi_summary* out, double* grad, double* log_bfs, double* log_neffs, double* variances,
             double* norms_out, int n_threads) {
  const int n_theta = sp->n_theta;
  double der[GWI_MAX_TERMS][8];
  memset(der, 0, sizeof(der));
  double host_const = 0.0;
  for (int t = 0; t < sp->n_terms; ++t) {
    const gwi_term* tm = &sp->terms[t];
    double a, b, c;
    switch (tm->kind) {
      case GWI_TERM_POWERLAW:
        if (!(tm->flags & GWI_POWERLAW_UNNORMALISED)) {
          pl_lognorm(th[tm->theta[0]], tm->p[0], tm->p[1], &a, &b);
          host_const += a;
        }
        break;
      case GWI_TERM_POWERLAW_BOUNDS:
        pl_lognorm(th[tm->theta[0]], th[tm->theta[1]], th[tm->theta[2]], &a, &b);
        if (th[tm->theta[0]] == -1.0) a = -log(th[tm->theta[2]] / th[tm->theta[1]]); /* numpyro_distributions.py:130 as written */
        host_const += a;
        break;
      case GWI_TERM_PLPEAK_SMOOTH:
      case GWI_TERM_PLPEAK:
        pl_lognorm(th[tm->theta[0]], tm->p[0], tm->p[1], &der[t][0], &der[t][1]);
        tn_lognorm(th[tm->theta[1]], th[tm->theta[2]], tm->p[0], tm->p[1], &der[t][2], &der[t][3], &der[t][4]);
        break;
      case GWI_TERM_BETA: host_const -= lgamma(th[tm->theta[0]]) + lgamma(th[tm->theta[1]]) - lgamma(th[tm->theta[0]] + th[tm->theta[1]]); break;
      case GWI_TERM_TILT_MIXTURE:
      case GWI_TERM_TILT_JOINT: tn_lognorm(1.0, th[tm->theta[1]], -1.0, 1.0, &der[t][0], &a, &der[t][1]); break;
      case GWI_TERM_TRUNCNORM:
        tn_lognorm(th[tm->theta[0]], th[tm->theta[1]], tm->p[0], tm->p[1], &a, &b, &c);
        host_const += a;
        break;
      default: break;
    }
  }
  double norms[GWI_MAX_NORMS];
  for (int j = 0; j < sp->n_norms; ++j) norms[j] = grid_norm(&sp->norms[j], th);
  double log_const = host_const;
  for (int t = 0; t < sp->n_terms; ++t)
    if (sp->terms[t].norm >= 0) log_const -= log(norms[sp->terms[t].norm]);
  if (norms_out) memcpy(norms_out, norms, sizeof(double) * sp->n_norms);

#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
  /* ONE parallel region over blocks of GWO_BLOCK samples: blocks [0, n_ev * nb_pe) are (event, block) pairs, the rest injection
   * blocks; then a fixed-order merge per event (parallel over events) and over the injection blocks (serial: a few hundred
   * records).  h (sum w^2 dl) only when the marginalised selection term needs it. */
  const int need_h = opt->marginalize_selection && grad;
  const int64_t nb_pe = (n_pe + GWO_BLOCK - 1) / GWO_BLOCK, nb_inj = (n_inj + GWO_BLOCK - 1) / GWO_BLOCK;
  const int64_t n_units = n_ev * nb_pe + nb_inj;
  const size_t stride = (size_t)n_theta * (need_h ? 2 : 1);
  acc_t* units = (acc_t*)malloc(sizeof(acc_t) * (size_t)(n_units ? n_units : 1));
  double* slab = (double*)malloc(sizeof(double) * stride * (size_t)(n_units ? n_units : 1));
#pragma omp parallel for schedule(dynamic, 2)
  for (int64_t u = 0; u < n_units; ++u) {
    double* g = slab + (size_t)u * stride;
    acc_init(&units[u], g, need_h ? g + n_theta : NULL, n_theta);
    if (u < n_ev * nb_pe) {
      const int64_t e = u / nb_pe, b = u - e * nb_pe;
      const int64_t lo = e * n_pe + b * GWO_BLOCK, hi = e * n_pe + ((b + 1) * GWO_BLOCK < n_pe ? (b + 1) * GWO_BLOCK : n_pe);
      scan_range(sp, pe_cols, lo, hi, th, der, &units[u]);
    } else {
      const int64_t b = u - n_ev * nb_pe;
      scan_range(sp, inj_cols, b * GWO_BLOCK, (b + 1) * GWO_BLOCK < n_inj ? (b + 1) * GWO_BLOCK : n_inj, th, der, &units[u]);
    }
  }
  double* ev = (double*)malloc(sizeof(double) * (size_t)(n_ev ? n_ev : 1) * (3 + n_theta));
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < n_ev; ++e) {
    double gbuf[2 * GWI_MAX_THETA];
    acc_t a;
    acc_init(&a, gbuf, NULL, n_theta);
    for (int64_t b = 0; b < nb_pe; ++b) acc_merge(&a, &units[e * nb_pe + b], n_theta); /* block order */
    double* row = ev + e * (3 + n_theta);
    const double log_s1 = log(a.s1);
    row[0] = log_s1 + a.m;                 /* logsumexp */
    row[1] = 2.0 * log_s1 - log(a.s2);     /* log n_eff (analysis.py:79) */
    row[2] = 1.0 / exp(row[1]) - 1.0 / (double)n_pe;
    for (int p = 0; p < n_theta; ++p) row[3 + p] = a.s1 > 0.0 ? a.g[p] / a.s1 : 0.0;
  }
  double inj_g[GWI_MAX_THETA], inj_h[GWI_MAX_THETA];
  acc_t inj;
  acc_init(&inj, inj_g, need_h ? inj_h : NULL, n_theta);
  for (int64_t b = 0; b < nb_inj; ++b) acc_merge(&inj, &units[n_ev * nb_pe + b], n_theta);
  free(units);
  free(slab);

  /* assembly: analysis.py:259-319 */
  const double n_obs = opt->n_obs, n_tot = opt->total_inj;
  double sum_lse = 0.0, sum_var = 0.0, min_lneff = INFINITY;
  for (int64_t e = 0; e < n_ev; ++e) {
    const double* row = ev + e * (3 + n_theta);
    sum_lse += row[0];
    sum_var += row[2];
    double le = row[1];
    if (le != le) le = 0.0;
    le = fmin(fmax(le, NEG_BIG), -NEG_BIG);
    min_lneff = fmin(min_lneff, le);
    if (log_bfs) log_bfs[e] = row[0] - log((double)n_pe) + log_const;
    if (log_neffs) log_neffs[e] = row[1];
    if (variances) variances[e] = row[2];
  }
  gwi_summary s;
  memset(&s, 0, sizeof(s));
  s.log_norm_const = log_const;
  s.sum_logBFs = sum_lse + (double)n_ev * (log_const - log((double)n_pe));
  const double log_mu = log(inj.s1) + inj.m - log(n_tot) + log_const;
  const double log_neff_inj = 2.0 * log(inj.s1) - log(inj.s2 - inj.s1 * inj.s1 / n_tot);
  const double var_mu = 1.0 / exp(log_neff_inj) - 1.0 / n_tot;
  s.log_det_eff = log_mu;
  s.log_nEff_inj = log_neff_inj;
  s.variance_log_detection_efficiency = var_mu;
  s.min_log_nEff = min_lneff;
  s.surveyed_hypervolume_norm = sp->vt_norm >= 0 ? norms[sp->vt_norm] : NAN;
  double lde = log_mu;
  int cut = 0;
  if (opt->marginalize_selection) lde -= (3.0 + n_obs) / (2.0 * exp(log_neff_inj));
  if (opt->min_neff_cut && !(log_neff_inj >= log(4.0 * n_obs))) lde = INFINITY;
  s.selection_factor = isinf(lde) ? NEG_BIG : -n_obs * lde;
  if (isinf(lde)) cut = 1;
  double log_l = s.selection_factor + s.sum_logBFs;
  if (isnan(log_l)) {
    log_l = NEG_BIG;
    cut = 1;
  } else if (isinf(log_l)) {
    log_l = log_l > 0 ? -NEG_BIG : NEG_BIG;
    cut = 1;
  }
  s.log_l = log_l;
  if (opt->min_neff_cut && exp(min_lneff) <= n_obs) {
    log_l = NEG_BIG;
    cut = 1;
  }
  s.variance_log_likelihood = n_obs * n_obs * var_mu + sum_var;
  if (opt->max_variance_cut && !(s.variance_log_likelihood <= 1.0)) {
    log_l = NEG_BIG;
    cut = 1;
  }
  s.log_likelihood = log_l;
  if (out) *out = s;
  if (grad) {
    for (int p = 0; p < n_theta; ++p) {
      double g = 0.0;
      for (int64_t e = 0; e < n_ev; ++e) g += ev[e * (3 + n_theta) + 3 + p];
      grad[p] = cut ? 0.0 : g - n_obs * (inj.s1 > 0.0 ? inj.g[p] / inj.s1 : 0.0);
      if (opt->marginalize_selection && !cut && inj.s1 > 0.0) {
        /* lde = log mu - c / n_eff, c = (3 + N_obs) / 2 (analysis.py:271); log n_eff = 2 log S1 - log V, V = S2 - S1^2 / N_tot;
         * dS1 = G = sum w dl, dS2 = 2 H, H = sum w^2 dl  =>  d(-N_obs lde) carries  -N_obs (c / n_eff) dlog n_eff */
        const double V = inj.s2 - inj.s1 * inj.s1 / n_tot;
        const double dlog_neff = 2.0 * inj.g[p] / inj.s1 - (2.0 * inj.h[p] - 2.0 * inj.s1 * inj.g[p] / n_tot) / V;
        grad[p] -= n_obs * (3.0 + n_obs) / (2.0 * exp(log_neff_inj)) * dlog_neff;
      }
    }
  }
  free(ev);
  return 0;
}

int gwo_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
