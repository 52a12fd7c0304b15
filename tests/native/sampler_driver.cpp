// TEST INFRASTRUCTURE (CPU): drives gwinferno_amd/csrc/gwi_sampler.cpp on its own under AddressSanitizer /
// UndefinedBehaviorSanitizer / ThreadSanitizer builds (tests/test_sanitizers_cpu.py).  The engine entry the
// sampler calls, gwi_eval_sequence, is replaced here by a correlated-Gaussian log-likelihood so that no GPU and no
// HIP runtime is involved: what is exercised is the sampler's own memory and thread behaviour -- tree building,
// adaptation, bijectors, penalties, one host thread per chain, and the lock-step scheduler (chains on stacks of their own,
// switched with swapcontext, which both sanitizers intercept).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "gwi_sampler.h"

namespace {
struct FakeEngine {
  int n;
  std::vector<double> mean, prec;  // precision matrix, row-major
  long long calls = 0;
};
}  // namespace

extern "C" gwi_status gwi_eval_sequence(gwi_handle h, const double* thetas, int32_t n, const gwi_options*, double* log_likelihoods, double* grads, int32_t, float*) {
  FakeEngine& e = *reinterpret_cast<FakeEngine*>(h);
  for (int s = 0; s < n; ++s) {
    const double* th = thetas + (size_t)s * e.n;
    double q = 0.0;
    for (int i = 0; i < e.n; ++i) {
      double r = 0.0;
      for (int j = 0; j < e.n; ++j) r += e.prec[(size_t)i * e.n + j] * (th[j] - e.mean[j]);
      if (grads) grads[(size_t)s * e.n + i] = -r;
      q += (th[i] - e.mean[i]) * r;
    }
    log_likelihoods[s] = -0.5 * q;
    ++e.calls;
  }
  return GWI_OK;
}

extern "C" gwi_status gwi_pin_thread_to_engine(gwi_handle) { return GWI_ERR_UNSUPPORTED; }

// the two halves of a batched evaluation (lock-step chains): the points are kept at begin and evaluated at end
namespace {
struct PendingBatch {
  gwi_handle h = nullptr;
  std::vector<double> thetas;
  int k = 0;
};
PendingBatch g_pending[4];
PendingBatch& pending_of(gwi_handle h) {
  for (auto& p : g_pending)
    if (p.h == h || p.h == nullptr) {
      p.h = h;
      return p;
    }
  return g_pending[0];
}
}  // namespace
extern "C" gwi_status gwi_eval_batch_begin(gwi_handle h, const double* thetas, int32_t k, const gwi_options*, int32_t, int32_t) {
  FakeEngine& e = *reinterpret_cast<FakeEngine*>(h);
  PendingBatch& p = pending_of(h);
  p.thetas.assign(thetas, thetas + (size_t)k * e.n);
  p.k = k;
  return GWI_OK;
}
extern "C" gwi_status gwi_eval_batch_end(gwi_handle h, gwi_summary* summaries, double* grads, double*, double*, double*, double*) {
  PendingBatch& p = pending_of(h);
  std::vector<double> ll(p.k);
  const gwi_status st = gwi_eval_sequence(h, p.thetas.data(), p.k, nullptr, ll.data(), grads, 0, nullptr);
  for (int j = 0; j < p.k; ++j) {
    std::memset(&summaries[j], 0, sizeof(gwi_summary));
    summaries[j].log_likelihood = ll[j];
  }
  return st;
}

static int32_t banana_batch(void*, int32_t k, const int32_t*, const double* xs, double* lps, double* grads) {
  for (int j = 0; j < k; ++j) {
    const double* x = xs + 2 * j;
    const double a = x[1] - x[0] * x[0];
    lps[j] = -0.5 * x[0] * x[0] - 2.0 * a * a;
    grads[2 * j] = -x[0] + 8.0 * a * x[0];
    grads[2 * j + 1] = -4.0 * a;
  }
  return 0;
}

static int32_t banana(void*, const double* x, double* lp, double* g) {  // a curved 2-d target for the callback entry
  const double a = x[1] - x[0] * x[0];
  *lp = -0.5 * x[0] * x[0] - 2.0 * a * a;
  g[0] = -x[0] + 8.0 * a * x[0];
  g[1] = -4.0 * a;
  return 0;
}

int main() {
  // 1. callback entry
  {
    const double x0[2] = {0.1, 0.2};
    gwi_nuts_options o = {200, 400, 8, 0, 0.8, 7};
    std::vector<double> samples(400 * 2), lp(400);
    std::vector<int32_t> depth(400);
    gwi_nuts_result r;
    if (gwi_nuts_run(banana, nullptr, 2, x0, &o, samples.data(), lp.data(), depth.data(), &r) != GWI_OK) return 1;
    double m = 0;
    for (int i = 0; i < 400; ++i) m += samples[2 * i] / 400;
    std::printf("banana: mean x0 %.3f accept %.2f evals %lld\n", m, r.accept_rate, (long long)r.n_evals);
    if (!(std::fabs(m) < 0.5 && r.accept_rate > 0.5)) return 2;
    if (gwi_nuts_run(nullptr, nullptr, 2, x0, &o, samples.data(), nullptr, nullptr, nullptr) != GWI_ERR_INVALID) return 3;
  }
  // 2. engine entry: 4 chains in 4 threads, every feature of the target (interval + positive bijectors, a Normal
  //    prior, a second-difference penalty)
  {
    const int n = 6, chains = 4;
    std::vector<FakeEngine> eng(chains);
    std::vector<gwi_handle> handles;
    for (auto& e : eng) {
      e.n = n;
      e.mean = {0.3, 2.0, -1.0, 0.5, 0.0, 1.0};
      e.prec.assign((size_t)n * n, 0.0);
      for (int i = 0; i < n; ++i) {
        e.prec[(size_t)i * n + i] = 2.0 + i;
        if (i + 1 < n) e.prec[(size_t)i * n + i + 1] = e.prec[(size_t)(i + 1) * n + i] = 0.4;
      }
      handles.push_back(reinterpret_cast<gwi_handle>(&e));
    }
    gwi_param_prior pri[n];
    for (int i = 0; i < n; ++i) pri[i] = {GWI_BIJECT_IDENTITY, 0, 0.0, 0.0, 5.0};
    pri[0] = {GWI_BIJECT_INTERVAL, 0, 0.0, 1.0, INFINITY};
    pri[1] = {GWI_BIJECT_POSITIVE, 0, 0.0, 0.0, 10.0};
    const gwi_smoothing_penalty pen = {2, 4, 2, 0, 0.5};
    gwi_options lopt;
    std::memset(&lopt, 0, sizeof(lopt));
    std::vector<double> u0((size_t)chains * n, 0.0);
    gwi_nuts_options o = {150, 300, 7, 0, 0.8, 11};
    std::vector<double> samples((size_t)chains * 300 * n), lp((size_t)chains * 300);
    std::vector<int32_t> depth((size_t)chains * 300);
    std::vector<gwi_nuts_result> res(chains);
    if (gwi_nuts_engine(handles.data(), chains, n, &lopt, pri, &pen, 1, u0.data(), &o, samples.data(), lp.data(), depth.data(), res.data()) != GWI_OK) return 4;
    for (int c = 0; c < chains; ++c) {
      double m0 = 0, m1 = 0;
      for (int k = 0; k < 300; ++k) {
        const double* s = &samples[((size_t)c * 300 + k) * n];
        if (!(s[0] > 0.0 && s[0] < 1.0 && s[1] > 0.0)) return 5;  // bijector ranges
        m0 += s[0] / 300;
        m1 += s[1] / 300;
      }
      std::printf("chain %d: mean theta0 %.3f theta1 %.3f accept %.2f evals %lld (engine calls %lld)\n", c, m0, m1, res[c].accept_rate, (long long)res[c].n_evals, eng[c].calls);
      if (res[c].n_evals != eng[c].calls || !(res[c].accept_rate > 0.5)) return 6;
    }
    const gwi_smoothing_penalty bad = {4, 4, 1, 0, 1.0};  // runs past n_theta
    if (gwi_nuts_engine(handles.data(), chains, n, &lopt, pri, &bad, 1, u0.data(), &o, samples.data(), nullptr, nullptr, nullptr) != GWI_ERR_INVALID) return 7;
  }
  // 3. lock-step chains (every chain on a stack of its own, ucontext): the callback entry against gwi_nuts_run chain by chain,
  //    then the engine entry with two groups of three chains against the threaded entry's target
  {
    const int chains = 5, ns = 60;
    std::vector<double> x0((size_t)chains * 2);
    for (int c = 0; c < chains; ++c) {
      x0[2 * c] = 0.1 * c;
      x0[2 * c + 1] = 0.2;
    }
    gwi_nuts_options o = {80, ns, 8, 0, 0.8, 21};
    std::vector<double> samples((size_t)chains * ns * 2), lp((size_t)chains * ns), alone((size_t)ns * 2);
    std::vector<int32_t> depth((size_t)chains * ns);
    std::vector<gwi_nuts_result> res(chains);
    if (gwi_nuts_run_lockstep(banana_batch, nullptr, 2, chains, x0.data(), &o, samples.data(), lp.data(), depth.data(), res.data()) != GWI_OK) return 8;
    for (int c = 0; c < chains; ++c) {
      gwi_nuts_options oc = o;
      oc.seed = o.seed + 1000ULL * c;
      gwi_nuts_result r;
      if (gwi_nuts_run(banana, nullptr, 2, &x0[2 * c], &oc, alone.data(), nullptr, nullptr, &r) != GWI_OK) return 9;
      if (std::memcmp(alone.data(), &samples[(size_t)c * ns * 2], sizeof(double) * ns * 2) != 0 || r.n_evals != res[c].n_evals) return 10;
    }
    double st[6];
    gwi_nuts_lockstep_stats(st);
    std::printf("lock step: %d chains, %.0f batches of %.2f points\n", chains, st[0], st[1] / st[0]);
  }
  {
    const int n = 6, groups = 2, per = 3, chains = groups * per, ns = 80;
    std::vector<FakeEngine> eng(groups);
    std::vector<gwi_handle> handles;
    for (auto& e : eng) {
      e.n = n;
      e.mean = {0.3, 2.0, -1.0, 0.5, 0.0, 1.0};
      e.prec.assign((size_t)n * n, 0.0);
      for (int i = 0; i < n; ++i) e.prec[(size_t)i * n + i] = 2.0 + i;
      handles.push_back(reinterpret_cast<gwi_handle>(&e));
    }
    gwi_param_prior pri[n];
    for (int i = 0; i < n; ++i) pri[i] = {GWI_BIJECT_IDENTITY, 0, 0.0, 0.0, 5.0};
    pri[0] = {GWI_BIJECT_INTERVAL, 0, 0.0, 1.0, INFINITY};
    pri[1] = {GWI_BIJECT_POSITIVE, 0, 0.0, 0.0, 10.0};
    const gwi_smoothing_penalty pen = {2, 4, 2, 0, 0.5};
    gwi_options lopt;
    std::memset(&lopt, 0, sizeof(lopt));
    std::vector<double> u0((size_t)chains * n, 0.0);
    gwi_nuts_options o = {60, ns, 7, 0, 0.8, 11};
    std::vector<double> samples((size_t)chains * ns * n), one((size_t)ns * n);
    std::vector<gwi_nuts_result> res(chains);
    if (gwi_nuts_engine_lockstep(handles.data(), groups, per, n, &lopt, pri, &pen, 1, u0.data(), &o, samples.data(), nullptr, nullptr, res.data()) != GWI_OK) return 11;
    for (int c = 0; c < chains; ++c) {  // chain c == the threaded entry's single chain with its seed (same target, same arithmetic)
      gwi_nuts_options oc = o;
      oc.seed = o.seed + 1000ULL * c;
      gwi_nuts_result r;
      if (gwi_nuts_engine(handles.data(), 1, n, &lopt, pri, &pen, 1, u0.data(), &oc, one.data(), nullptr, nullptr, &r) != GWI_OK) return 12;
      if (std::memcmp(one.data(), &samples[(size_t)c * ns * n], sizeof(double) * ns * n) != 0 || r.n_evals != res[c].n_evals) return 13;
    }
    std::printf("lock step on engines: %d x %d chains equal the chains run alone\n", groups, per);
    // 4. a QUEUE of chains: 11 chains over 2 groups x 2 slots -- a chain that ends hands its slot on; every chain still draws
    //    what it draws alone, in whichever group and next to whichever chains it ran
    const int queued = 11, slots = 2;
    std::vector<double> uq((size_t)queued * n, 0.0), sq((size_t)queued * ns * n);
    for (int c = 0; c < queued; ++c)
      for (int i = 2; i < n; ++i) uq[(size_t)c * n + i] = 0.15 * c - 0.05 * i;
    std::vector<gwi_nuts_result> rq(queued);
    if (gwi_nuts_engine_queue(handles.data(), groups, slots, queued, n, &lopt, pri, &pen, 1, uq.data(), &o, sq.data(), nullptr, nullptr, rq.data()) != GWI_OK) return 14;
    double st[6];
    gwi_nuts_lockstep_stats(st);
    for (int c = 0; c < queued; ++c) {
      gwi_nuts_options oc = o;
      oc.seed = o.seed + 1000ULL * c;
      gwi_nuts_result r;
      if (gwi_nuts_engine(handles.data(), 1, n, &lopt, pri, &pen, 1, &uq[(size_t)c * n], &oc, one.data(), nullptr, nullptr, &r) != GWI_OK) return 15;
      if (std::memcmp(one.data(), &sq[(size_t)c * ns * n], sizeof(double) * ns * n) != 0 || r.n_evals != rq[c].n_evals) return 16;
    }
    if (st[1] / st[0] > slots || st[1] / st[0] < 1.6) return 17;  // never more than `slots` per batch, and close to full
    std::printf("queue of chains on engines: %d chains over %d x %d slots, %.0f batches of %.2f points\n", queued, groups, slots, st[0], st[1] / st[0]);
  }
  std::printf("OK\n");
  return 0;
}
