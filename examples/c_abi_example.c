/* The C ABI on its own (no Python, no torch): a power-law population in one parameter x on [lo, hi],
 *     w(x; alpha) = x^alpha (1+alpha)/(hi^(1+alpha) - lo^(1+alpha)) / prior(x),
 * 4 "events" x 1000 posterior samples and 3000 found injections, evaluated by gwi_eval and checked against
 * a plain double loop in this file (value of log_l, per-event log Bayes factors, d log_l / d alpha).
 *   gcc -O2 -Iinclude examples/c_abi_example.c -o c_abi_example -Lgwinferno_amd/_lib -lgwi_engine \
 *       -Wl,-rpath,$PWD/gwinferno_amd/_lib -lm && ./c_abi_example                                        */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gwi_engine.h"

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
  unsigned long long seed = 88172645463325252ULL;
  for (int i = 0; i < N_EV * N_PE; ++i) {
    const double x = 4.0 + 110.0 * urand(&seed); /* some samples fall outside [lo, hi] */
    logx_pe[i] = log(x);
    kap_pe[i] = (x < lo || x > hi) ? -INFINITY : -log(0.01 * x); /* kappa = -log prior, -inf = excluded */
  }
  for (int i = 0; i < N_INJ; ++i) {
    const double x = lo + (hi - lo) * urand(&seed);
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
  gwi_destroy(h);
  if (e_val < 1e-11 && e_grad < 1e-10 && worst_bf < 1e-11) {
    printf("OK\n");
    return 0;
  }
  printf("MISMATCH\n");
  return 1;
}
