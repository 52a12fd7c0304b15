/* The C ABI on its own (no Python, no torch): a power-law population in one parameter x on [lo, hi],
 *     w(x; alpha) = x^alpha (1+alpha)/(hi^(1+alpha) - lo^(1+alpha)) / prior(x),
 * 4 "events" x 1000 posterior samples and 3000 found injections, evaluated by gwi_eval and checked against
 * a plain double loop in this file (value of log_l, per-event log Bayes factors, d log_l / d alpha); then the
 * library's NUTS (gwi_nuts_engine) samples alpha under a Normal(0, 5) prior and its posterior mean and width are
 * checked against quadrature of the same posterior on a grid.  A second engine is built from the RAW x and prior
 * arrays with gwi_create_ingest (logarithm, truncation mask and -log prior computed by the library's setup kernel on
 * the device) and must give the same likelihood.
 *   gcc -O2 -Iinclude examples/c_abi_example.c -o c_abi_example -Lgwinferno_amd/_lib -lgwi_engine \
 *       -Wl,-rpath,$PWD/gwinferno_amd/_lib -lm && ./c_abi_example                                        */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gwi_engine.h"
#include "gwi_sampler.h"

#define N_EV 4
#define N_PE 1000
#define N_INJ 3000

static double urand(unsigned long long* s) { /* xorshift64*, [0, 1) */
  *s ^= *s >> 12;
  *s ^= *s << 25;
  *s ^= *s >> 27;
  return (double)((*s * 2685821657736338717ULL) >> 11) / 9007199254740992.0;
}

int main(void) {
  const double lo = 5.0, hi = 100.0, alpha = -2.3, total_inj = 60000.0;
  static double logx_pe[N_EV * N_PE], kap_pe[N_EV * N_PE], logx_inj[N_INJ], kap_inj[N_INJ];
  static double x_pe[N_EV * N_PE], prior_pe[N_EV * N_PE], x_inj[N_INJ], prior_inj[N_INJ]; /* the raw catalog, for gwi_create_ingest */
  unsigned long long seed = 88172645463325252ULL;
  for (int i = 0; i < N_EV * N_PE; ++i) {
    const double x = 4.0 + 110.0 * urand(&seed); /* some samples fall outside [lo, hi] */
    x_pe[i] = x;
    prior_pe[i] = 0.01 * x;
    logx_pe[i] = log(x);
    kap_pe[i] = (x < lo || x > hi) ? -INFINITY : -log(0.01 * x); /* kappa = -log prior, -inf = excluded */
  }
  for (int i = 0; i < N_INJ; ++i) {
    const double x = lo + (hi - lo) * urand(&seed);
    x_inj[i] = x;
    prior_inj[i] = 1.0 / (hi - lo);
    logx_inj[i] = log(x);
    kap_inj[i] = -log(1.0 / (hi - lo));
  }

  gwi_spec spec;
  memset(&spec, 0, sizeof(spec));
  spec.abi_version = GWI_ABI_VERSION;
  spec.n_cols = 2; /* column 0 = log x, column 1 = kappa */
  spec.kappa_col = 1;
  spec.n_theta = 1;
  spec.n_terms = 1;
  spec.n_norms = 0;
  spec.vt_norm = -1;
  spec.terms[0].kind = GWI_TERM_POWERLAW;
  spec.terms[0].cols[0] = 0;
  spec.terms[0].cols[1] = 0;
  spec.terms[0].theta[0] = 0;
  spec.terms[0].norm = -1;
  spec.terms[0].p[0] = lo;
  spec.terms[0].p[1] = hi;

  const double* pe_cols[2] = {logx_pe, kap_pe};
  const double* inj_cols[2] = {logx_inj, kap_inj};
  gwi_handle h = NULL;
  gwi_status st = gwi_create(&spec, pe_cols, N_EV, N_PE, inj_cols, N_INJ, GWI_DEVICE_CURRENT, &h);
  if (st != GWI_OK) {
    fprintf(stderr, "gwi_create failed (%d): %s\n", (int)st, h ? gwi_last_error(h) : "no device");
    return 2;
  }
  gwi_options opt = {(double)N_EV, total_inj, 0, 0, 0, 0};
  gwi_summary s;
  double grad[1], log_bfs[N_EV], log_neffs[N_EV], variances[N_EV];
  st = gwi_eval(h, &alpha, &opt, &s, grad, log_bfs, log_neffs, variances, NULL);
  if (st != GWI_OK) {
    fprintf(stderr, "gwi_eval failed (%d): %s\n", (int)st, gwi_last_error(h));
    return 2;
  }

  /* the same engine from the raw columns: the setup program below is what a binding emits for
   *   col0 = log x;   kappa = where((x < lo) | (x > hi), -inf, -log prior)                                   */
  const gwi_ingest_op prog[] = {
      {GWI_ING_LOAD, 0, 0, 0, 0, 0, 0.0},  /* r0 = x */
      {GWI_ING_LOG, 1, 0, 0, 0, 0, 0.0},   /* r1 = log x */
      {GWI_ING_LOAD, 2, 1, 0, 0, 0, 0.0},  /* r2 = prior */
      {GWI_ING_LOG, 2, 2, 0, 0, 0, 0.0},
      {GWI_ING_NEG, 2, 2, 0, 0, 0, 0.0},   /* r2 = -log prior */
      {GWI_ING_CONST, 3, 0, 0, 0, 0, lo},
      {GWI_ING_LT, 3, 0, 3, 0, 0, 0.0},    /* r3 = x < lo */
      {GWI_ING_CONST, 4, 0, 0, 0, 0, hi},
      {GWI_ING_GT, 4, 0, 4, 0, 0, 0.0},    /* r4 = x > hi */
      {GWI_ING_OR, 3, 3, 4, 0, 0, 0.0},
      {GWI_ING_CONST, 4, 0, 0, 0, 0, -INFINITY},
      {GWI_ING_WHERE, 2, 3, 4, 2, 0, 0.0}, /* r2 = r3 ? -inf : r2 */
      {GWI_ING_STORE, 0, 1, 0, 0, 0, 0.0},
      {GWI_ING_STORE, 1, 2, 0, 0, 0, 0.0},
  };
  const int32_t f64x2[2] = {GWI_DTYPE_F64, GWI_DTYPE_F64};
  const void* src_pe[2] = {x_pe, prior_pe};
  const void* src_inj[2] = {x_inj, prior_inj};
  const gwi_ingest_program ing_pe = {(int32_t)(sizeof(prog) / sizeof(prog[0])), 5, 2, 0, prog, src_pe, f64x2, NULL, NULL};
  const gwi_ingest_program ing_inj = {(int32_t)(sizeof(prog) / sizeof(prog[0])), 5, 2, 0, prog, src_inj, f64x2, NULL, NULL};
  gwi_handle h2 = NULL;
  st = gwi_create_ingest(&spec, &ing_pe, N_EV, N_PE, &ing_inj, N_INJ, GWI_DEVICE_CURRENT, &h2);
  if (st != GWI_OK) {
    fprintf(stderr, "gwi_create_ingest failed (%d): %s\n", (int)st, h2 ? gwi_last_error(h2) : "no device");
    return 2;
  }
  gwi_summary s2;
  static double kap_back[N_EV * N_PE];
  if (gwi_eval(h2, &alpha, &opt, &s2, NULL, NULL, NULL, NULL, NULL) != GWI_OK || gwi_read_column(h2, 1, 1, kap_back) != GWI_OK) return 2;
  int same_mask = 1;
  for (int i = 0; i < N_EV * N_PE; ++i) same_mask = same_mask && ((kap_back[i] == -INFINITY) == (kap_pe[i] == -INFINITY));
  const double e_ingest = fabs(s2.log_likelihood - s.log_likelihood) / fabs(s.log_likelihood);
  printf("device setup (gwi_create_ingest): log_l %.12f, rel. diff to the host-prepared engine %.2e, masks %s\n", s2.log_likelihood, e_ingest, same_mask ? "equal" : "DIFFER");
  gwi_destroy(h2);

  /* the same, written out: log_l = sum_i log(mean_j w_ij) - N_ev log(sum_j w_j / N_tot) */
  const double b1 = 1.0 + alpha, den = pow(hi, b1) - pow(lo, b1);
  const double log_norm = log(b1 / den), dlog_norm = 1.0 / b1 - (pow(hi, b1) * log(hi) - pow(lo, b1) * log(lo)) / den;
  double log_l = 0.0, dlog_l = 0.0, worst_bf = 0.0;
  for (int e = 0; e < N_EV; ++e) {
    double sw = 0.0, sg = 0.0;
    for (int j = 0; j < N_PE; ++j) {
      const int i = e * N_PE + j;
      if (kap_pe[i] == -INFINITY) continue;
      const double w = exp(alpha * logx_pe[i] + log_norm + kap_pe[i]);
      sw += w;
      sg += w * (logx_pe[i] + dlog_norm);
    }
    const double lbf = log(sw / N_PE);
    worst_bf = fmax(worst_bf, fabs(lbf - log_bfs[e]));
    log_l += lbf;
    dlog_l += sg / sw;
  }
  double sw = 0.0, sg = 0.0;
  for (int j = 0; j < N_INJ; ++j) {
    const double w = exp(alpha * logx_inj[j] + log_norm + kap_inj[j]);
    sw += w;
    sg += w * (logx_inj[j] + dlog_norm);
  }
  log_l -= N_EV * log(sw / total_inj);
  dlog_l -= N_EV * sg / sw;

  const double e_val = fabs(s.log_likelihood - log_l) / fabs(log_l), e_grad = fabs(grad[0] - dlog_l) / fmax(1.0, fabs(dlog_l));
  printf("log_l engine %.12f  loop %.12f  rel.err %.2e | dlog_l/dalpha engine %.10f loop %.10f err %.2e | max |dlogBF| %.2e\n", s.log_likelihood, log_l, e_val, grad[0],
         dlog_l, e_grad, worst_bf);

  /* posterior of alpha by quadrature (engine likelihood x Normal(0, 5) prior) ... */
  double z0 = 0.0, z1 = 0.0, z2 = 0.0, top = -INFINITY;
  static double lp_grid[2400];
  for (int k = 0; k < 2400; ++k) {
    const double a = -22.005 + 0.01 * k;
    gwi_summary sk;
    if (gwi_eval(h, &a, &opt, &sk, NULL, NULL, NULL, NULL, NULL) != GWI_OK) return 2;
    lp_grid[k] = sk.log_likelihood - 0.5 * a * a / 25.0;
    top = fmax(top, lp_grid[k]);
  }
  for (int k = 0; k < 2400; ++k) {
    const double a = -22.005 + 0.01 * k, w = exp(lp_grid[k] - top);
    z0 += w;
    z1 += w * a;
    z2 += w * a * a;
  }
  const double q_mean = z1 / z0, q_sd = sqrt(z2 / z0 - q_mean * q_mean);
  /* ... and by the library's sampler: one chain, 300 warm-up + 2000 draws */
  const gwi_param_prior prior = {GWI_BIJECT_IDENTITY, 0, 0.0, 0.0, 5.0};
  const gwi_nuts_options nopt = {300, 2000, 8, 0, 0.8, 2025};
  static double draws[2000];
  gwi_nuts_result nres;
  st = gwi_nuts_engine(&h, 1, 1, &opt, &prior, NULL, 0, &alpha, &nopt, draws, NULL, NULL, &nres);
  if (st != GWI_OK) {
    fprintf(stderr, "gwi_nuts_engine failed (%d): %s\n", (int)st, gwi_last_error(h));
    return 2;
  }
  double m1 = 0.0, m2 = 0.0;
  for (int k = 0; k < 2000; ++k) m1 += draws[k] / 2000.0;
  for (int k = 0; k < 2000; ++k) m2 += (draws[k] - m1) * (draws[k] - m1) / 2000.0;
  printf("alpha | data: quadrature %.4f +- %.4f, NUTS %.4f +- %.4f (%lld evaluations, accept %.2f, %d divergent)\n", q_mean, q_sd, m1, sqrt(m2), (long long)nres.n_evals,
         nres.accept_rate, (int)nres.n_divergent);
  const int nuts_ok = fabs(m1 - q_mean) < 0.2 * q_sd && fabs(sqrt(m2) / q_sd - 1.0) < 0.2;
  gwi_destroy(h);
  if (e_val < 1e-11 && e_grad < 1e-10 && worst_bf < 1e-11 && nuts_ok && e_ingest < 1e-13 && same_mask) {
    printf("OK\n");
    return 0;
  }
  printf("MISMATCH\n");
  return 1;
}
