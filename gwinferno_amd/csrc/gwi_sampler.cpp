// gwi_sampler.cpp -- host-side No-U-Turn sampler of the engine library (declared in include/gwi_sampler.h).
//
// The reference drives its model with numpyro.infer.NUTS (examples/utils.py:63-85): jit-compiled host code
// around one value_and_grad per leapfrog step.  This is the native counterpart for environments without
// JAX/NumPyro, so that sampling runs at the engine's evaluation rate instead of a Python interpreter's:
// multinomial NUTS with the generalised U-turn criterion (Betancourt 2017; the scheme of Stan / NumPyro),
// dual-averaging step size (Hoffman & Gelman 2014, alg. 5) and the windowed warm-up of Stan / NumPyro for the diagonal mass
// matrix (an initial fast buffer, slow windows of doubling length each ending in a regularised variance estimate, a re-found
// step size and a restarted dual averaging, a final fast buffer: warmup_schedule below).
// gwinferno_amd/sampling.py holds the same algorithm in NumPy (the two are tested against the same targets).
//
//   gwi_nuts_run      any target given as a C callback (log-probability and gradient)
//   gwi_nuts_engine   target = engine log-likelihood + Normal priors + P-spline difference penalties
//                     (pipeline/utils.py:163-216) on constrained parameters mapped by interval / positive
//                     bijectors; chains run in one host thread each, every chain on its own engine handle
//   gwi_nuts_run_lockstep / gwi_nuts_engine_lockstep
//                     K chains advanced together, ONE batched evaluation (gwi_eval_batch: the several-points-per-launch
//                     kernels) per leapfrog step of all of them -- numpyro's chain_method="vectorized".  Every chain runs
//                     the unchanged run_nuts on a stack of its own (ucontext); its target hands the point to the scheduler
//                     and is resumed with the batch's result.  One host thread, G groups of chains (an engine each) in
//                     flight: group g's launches run while the others' chains do their host arithmetic.
#include "gwi_sampler.h"

#include <ucontext.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

using Vec = std::vector<double>;

struct Target {
  gwi_target_fn fn;
  void* user;
  int dim;
  long long n_evals = 0;
  bool failed = false;
  // evaluates at x; returns log-probability, writes the gradient
  double operator()(const Vec& x, Vec& grad) {
    double lp = 0.0;
    ++n_evals;
    if (fn(user, x.data(), &lp, grad.data()) != 0) failed = true;
    return lp;
  }
};

struct State {
  Vec th, p, g;
  double lp = 0.0;
};

double dot_w(const Vec& a, const Vec& w, const Vec& b) {  // sum a_i w_i b_i
  double s = 0.0;
  for (size_t i = 0; i < a.size(); ++i) s += a[i] * w[i] * b[i];
  return s;
}
double kinetic(const Vec& p, const Vec& inv_mass) { return 0.5 * dot_w(p, inv_mass, p); }

void leapfrog(Target& t, const State& a, double eps, const Vec& inv_mass, State& b) {
  const int d = t.dim;
  b.th.resize(d);
  b.p.resize(d);
  b.g.resize(d);
  for (int i = 0; i < d; ++i) {
    b.p[i] = a.p[i] + 0.5 * eps * a.g[i];
    b.th[i] = a.th[i] + eps * inv_mass[i] * b.p[i];
  }
  b.lp = t(b.th, b.g);
  for (int i = 0; i < d; ++i) b.p[i] += 0.5 * eps * b.g[i];
}

bool usable(double lp, const Vec& g) {
  if (!(std::isfinite(lp) && lp > -1e300)) return false;
  for (double v : g)
    if (!std::isfinite(v)) return false;
  return true;
}

struct Tree {
  State edge;      // outer end of the subtree
  Vec p_first;     // momentum at its inner end
  Vec prop_th, prop_g;
  double prop_lp = 0.0;
  double logw = -std::numeric_limits<double>::infinity();
  Vec rho;
  bool turning = false, diverging = false;
  double sum_alpha = 0.0;
  int n_alpha = 0;
};

struct Sampler {
  Target& t;
  std::mt19937_64 rng;
  std::normal_distribution<double> normal{0.0, 1.0};
  std::uniform_real_distribution<double> unif{0.0, 1.0};
  Vec inv_mass;
  Sampler(Target& t_, unsigned long long seed) : t(t_), rng(seed), inv_mass(t_.dim, 1.0) {}

  double log_unif() { return std::log(unif(rng)); }

  // Hoffman & Gelman (2014), algorithm 4
  double find_step_size(const State& s, double eps) {
    State a = s;
    for (int i = 0; i < t.dim; ++i) a.p[i] = normal(rng) / std::sqrt(inv_mass[i]);
    const double h0 = -a.lp + kinetic(a.p, inv_mass);
    auto log_ratio = [&](double e) {
      State b;
      leapfrog(t, a, e, inv_mass, b);
      const double h1 = -b.lp + kinetic(b.p, inv_mass);
      return (std::isfinite(h1) && b.lp > -1e300) ? h0 - h1 : -std::numeric_limits<double>::infinity();
    };
    const double dir = log_ratio(eps) > std::log(0.5) ? 1.0 : -1.0;
    for (int it = 0; it < 50; ++it) {
      if (!(dir * log_ratio(eps) > -dir * std::log(2.0))) break;
      eps *= std::pow(2.0, dir);
    }
    return eps;
  }

  void build(const State& from, int direction, int depth, double eps, double h0, Tree& out) {
    const int d = t.dim;
    if (depth == 0) {
      leapfrog(t, from, direction * eps, inv_mass, out.edge);
      const double h1 = -out.edge.lp + kinetic(out.edge.p, inv_mass);
      const bool ok = std::isfinite(h1) && usable(out.edge.lp, out.edge.g);
      const double dl = ok ? h0 - h1 : -std::numeric_limits<double>::infinity();
      out.p_first = out.edge.p;
      out.prop_th = out.edge.th;
      out.prop_g = out.edge.g;
      out.prop_lp = out.edge.lp;
      out.logw = dl;
      out.rho = out.edge.p;
      out.turning = false;
      out.diverging = !ok || dl < -1000.0;
      out.sum_alpha = ok ? std::fmin(1.0, std::exp(std::fmin(0.0, dl))) : 0.0;
      out.n_alpha = 1;
      return;
    }
    Tree a;
    build(from, direction, depth - 1, eps, h0, a);
    if (a.diverging || a.turning) {
      out = std::move(a);
      return;
    }
    Tree b;
    build(a.edge, direction, depth - 1, eps, h0, b);
    const double m = std::fmax(a.logw, b.logw);
    const double logw = std::isfinite(m) ? m + std::log(std::exp(a.logw - m) + std::exp(b.logw - m)) : m;
    out.prop_th = a.prop_th;
    out.prop_g = a.prop_g;
    out.prop_lp = a.prop_lp;
    if (!(b.diverging || b.turning) && log_unif() < b.logw - logw) {  // uniform over the subtree
      out.prop_th = b.prop_th;
      out.prop_g = b.prop_g;
      out.prop_lp = b.prop_lp;
    }
    out.rho.resize(d);
    for (int i = 0; i < d; ++i) out.rho[i] = a.rho[i] + b.rho[i];
    out.p_first = a.p_first;
    // generalised U-turn test: summed momentum against the velocities at the two ends of this subtree
    out.turning = b.turning || dot_w(out.rho, inv_mass, out.p_first) <= 0.0 || dot_w(out.rho, inv_mass, b.edge.p) <= 0.0;
    out.diverging = b.diverging;
    out.sum_alpha = a.sum_alpha + b.sum_alpha;
    out.n_alpha = a.n_alpha + b.n_alpha;
    out.logw = logw;
    out.edge = std::move(b.edge);
  }
};

// End indices (exclusive, in warm-up iterations) of the slow adaptation windows, and the first iteration of the slow phase:
// Stan's schedule as NumPyro builds it (numpyro.infer.hmc_util.build_adaptation_schedule): 75 fast iterations, windows of
// 25, 50, 100, ... (the last one stretched to the end of the slow phase), 50 fast iterations; shrunk to 15 % / 75 % / 10 % when
// the warm-up is shorter than 150 iterations; no mass-matrix adaptation below 20.
struct WarmupSchedule {
  int slow_start = 0;
  std::vector<int> window_end;
};
WarmupSchedule warmup_schedule(int n) {
  WarmupSchedule w;
  if (n < 20) {
    w.slow_start = n;
    return w;
  }
  int init = 75, term = 50, base = 25;
  if (init + base + term > n) {
    init = (int)(0.15 * n);
    term = (int)(0.1 * n);
    base = n - init - term;
  }
  w.slow_start = init;
  const int slow_end = n - term;
  int start = init, size = base;
  while (start < slow_end) {
    int end = start + size;
    if (end + 2 * size > slow_end) end = slow_end;  // the next window would not fit: this one takes the rest
    w.window_end.push_back(end);
    start = end;
    size *= 2;
  }
  return w;
}

int run_nuts(Target& t, const double* x0, const gwi_nuts_options& o, double* samples, double* logp, int32_t* tree_depth, gwi_nuts_result* res) {
  const int d = t.dim;
  Sampler s(t, o.seed);
  State cur;
  cur.th.assign(x0, x0 + d);
  cur.p.assign(d, 0.0);
  cur.g.assign(d, 0.0);
  cur.lp = t(cur.th, cur.g);
  if (t.failed) return 1;
  if (!usable(cur.lp, cur.g)) return 2;  // zero likelihood (a cut, or outside the support) or a non-finite gradient at the start
  double eps = s.find_step_size(cur, 0.1);
  double mu = std::log(10 * eps), log_eps_bar = 0.0, h_bar = 0.0;
  const double gamma = 0.05, t0 = 10.0, kappa = 0.75;
  int da_count = 0;
  const WarmupSchedule sched = warmup_schedule(o.n_warmup);
  size_t next_window = 0;
  Vec w_mean(d, 0.0), w_m2(d, 0.0);  // Welford accumulators of the current slow window
  int w_count = 0;
  double acc_sum = 0.0;
  int n_div = 0;
  const int max_depth = o.max_tree_depth > 0 ? o.max_tree_depth : 10;
  for (int it = 0; it < o.n_warmup + o.n_samples && !t.failed; ++it) {
    for (int i = 0; i < d; ++i) cur.p[i] = s.normal(s.rng) / std::sqrt(s.inv_mass[i]);
    const double h0 = -cur.lp + kinetic(cur.p, s.inv_mass);
    State left = cur, right = cur;
    Vec prop_th = cur.th, prop_g = cur.g;
    double prop_lp = cur.lp;
    double logw = 0.0;
    Vec rho = cur.p;
    double sum_alpha = 0.0;
    int n_alpha = 0, depth = 0;
    bool diverged = false;
    while (depth < max_depth) {
      const int direction = s.unif(s.rng) < 0.5 ? 1 : -1;
      Tree sub;
      s.build(direction == 1 ? right : left, direction, depth, eps, h0, sub);
      sum_alpha += sub.sum_alpha;
      n_alpha += sub.n_alpha;
      if (sub.diverging) {
        diverged = true;
        break;
      }
      if (sub.turning) break;
      if (s.log_unif() < sub.logw - logw) {  // biased progressive sampling across doublings
        prop_th = sub.prop_th;
        prop_g = sub.prop_g;
        prop_lp = sub.prop_lp;
      }
      const double m = std::fmax(logw, sub.logw);
      logw = m + std::log(std::exp(logw - m) + std::exp(sub.logw - m));
      for (int i = 0; i < d; ++i) rho[i] += sub.rho[i];
      if (direction == 1)
        right = std::move(sub.edge);
      else
        left = std::move(sub.edge);
      ++depth;
      if (dot_w(rho, s.inv_mass, left.p) <= 0.0 || dot_w(rho, s.inv_mass, right.p) <= 0.0) break;
    }
    cur.th = prop_th;
    cur.g = prop_g;
    cur.lp = prop_lp;
    const double acc = sum_alpha / (n_alpha > 0 ? n_alpha : 1);
    if (it < o.n_warmup) {
      const int m = it + 1;
      ++da_count;
      h_bar = (1 - 1.0 / (da_count + t0)) * h_bar + (o.target_accept - acc) / (da_count + t0);
      const double log_eps = mu - std::sqrt((double)da_count) / gamma * h_bar;
      const double w = std::pow((double)da_count, -kappa);
      log_eps_bar = w * log_eps + (1 - w) * log_eps_bar;
      eps = std::exp(log_eps);
      if (it >= sched.slow_start && next_window < sched.window_end.size()) {  // slow phase: this draw feeds the window's variance
        ++w_count;
        for (int i = 0; i < d; ++i) {
          const double delta = cur.th[i] - w_mean[i];
          w_mean[i] += delta / w_count;
          w_m2[i] += delta * (cur.th[i] - w_mean[i]);
        }
        if (m == sched.window_end[next_window]) {  // end of a window: new metric, re-found step size, dual averaging restarted
          if (w_count > 1)
            for (int i = 0; i < d; ++i) {
              const double var = w_m2[i] / (w_count - 1);
              // Stan's shrinkage towards the unit metric: (n / (n + 5)) var + 1e-3 (5 / (n + 5))
              const double reg = (w_count / (w_count + 5.0)) * var + 1e-3 * (5.0 / (w_count + 5.0));
              s.inv_mass[i] = (std::isfinite(reg) && reg > 0.0) ? reg : 1.0;
            }
          std::fill(w_mean.begin(), w_mean.end(), 0.0);
          std::fill(w_m2.begin(), w_m2.end(), 0.0);
          w_count = 0;
          ++next_window;
          eps = s.find_step_size(cur, std::exp(log_eps_bar));
          mu = std::log(10 * eps);
          log_eps_bar = 0.0;
          h_bar = 0.0;
          da_count = 0;
        }
      }
      if (m == o.n_warmup && da_count > 0) eps = std::exp(log_eps_bar);
    } else {
      const int k = it - o.n_warmup;
      std::memcpy(samples + (size_t)k * d, cur.th.data(), sizeof(double) * d);
      if (logp) logp[k] = cur.lp;
      if (tree_depth) tree_depth[k] = depth;
      acc_sum += acc;
      n_div += diverged ? 1 : 0;  // post-warm-up transitions only, as numpyro reports them
    }
  }
  if (res) {
    res->accept_rate = o.n_samples > 0 ? acc_sum / o.n_samples : 0.0;
    res->step_size = eps;
    res->n_evals = t.n_evals;
    res->n_divergent = n_div;
  }
  return t.failed ? 1 : 0;
}

// ---- engine target: log-likelihood + priors in unconstrained coordinates (mirror of sampling.make_target) ----
struct EngineTarget {
  gwi_handle h;
  gwi_options lopt;
  const gwi_param_prior* priors;
  const gwi_smoothing_penalty* pens;
  int n_pens, n_theta;
  Vec theta, dth, dlogj, grad_ll, d1, d2;
  // lock-step chains: the likelihood at e.theta comes from the group's batched launch instead of a blocking gwi_eval
  int (*eval_hook)(void* hook_user, const double* theta, double* ll, double* grad_ll) = nullptr;
  void* hook_user = nullptr;
};

int engine_target(void* user, const double* u, double* logp, double* grad) {
  EngineTarget& e = *static_cast<EngineTarget*>(user);
  const int n = e.n_theta;
  double logj = 0.0;
  for (int i = 0; i < n; ++i) {
    const gwi_param_prior& pr = e.priors[i];
    if (pr.kind == GWI_BIJECT_INTERVAL) {
      const double sig = 1.0 / (1.0 + std::exp(-u[i])), width = pr.hi - pr.lo;
      e.theta[i] = pr.lo + width * sig;
      e.dth[i] = width * sig * (1 - sig);
      e.dlogj[i] = 1 - 2 * sig;
      logj += std::log(e.dth[i]);
    } else if (pr.kind == GWI_BIJECT_POSITIVE) {
      e.theta[i] = std::exp(u[i]);
      e.dth[i] = e.theta[i];
      e.dlogj[i] = 1.0;
      logj += u[i];
    } else if (pr.kind == GWI_BIJECT_FIXED) {  // pinned parameter; its coordinate is an independent N(0,1) dummy
      e.theta[i] = pr.lo;
      e.dth[i] = 0.0;
      e.dlogj[i] = -u[i];
      logj += -0.5 * u[i] * u[i];
    } else {
      e.theta[i] = u[i];
      e.dth[i] = 1.0;
      e.dlogj[i] = 0.0;
    }
  }
  // one blocking evaluation; the sequence entry picks gwi_eval_sharded when the handle carries a communicator
  // (every rank then runs the same chain from the same seed: the exchanged records make the bits identical)
  double ll = 0.0;
  if (e.eval_hook) {
    if (e.eval_hook(e.hook_user, e.theta.data(), &ll, e.grad_ll.data()) != 0) return 1;
  } else if (gwi_eval_sequence(e.h, e.theta.data(), 1, &e.lopt, &ll, e.grad_ll.data(), 0, nullptr) != GWI_OK) {
    return 1;
  }
  double lp = ll + logj;
  for (int i = 0; i < n; ++i) {
    double g = e.grad_ll[i];
    const double sg = e.priors[i].sigma;
    if (std::isfinite(sg) && sg > 0) {  // Normal(0, sigma) on the constrained value
      lp += -0.5 * e.theta[i] * e.theta[i] / (sg * sg);
      g += -e.theta[i] / (sg * sg);
    }
    grad[i] = g;
  }
  for (int k = 0; k < e.n_pens; ++k) {  // -0.5 tau ||D^deg c||^2 (models/bsplines/smoothing.py:8-28)
    const gwi_smoothing_penalty& pn = e.pens[k];
    e.d1.assign(e.theta.begin() + pn.offset, e.theta.begin() + pn.offset + pn.count);
    for (int r = 0; r < pn.degree; ++r) {
      for (size_t i = 0; i + 1 < e.d1.size(); ++i) e.d1[i] = e.d1[i + 1] - e.d1[i];
      e.d1.pop_back();
    }
    double ss = 0.0;
    for (double v : e.d1) ss += v * v;
    lp += -0.5 * pn.tau * ss;
    e.d2 = e.d1;  // (D^deg)^T d: apply the transposed difference `degree` times
    for (int r = 0; r < pn.degree; ++r) {
      Vec nx(e.d2.size() + 1);
      nx[0] = -e.d2[0];
      for (size_t i = 1; i < e.d2.size(); ++i) nx[i] = e.d2[i - 1] - e.d2[i];
      nx[e.d2.size()] = e.d2.back();
      e.d2.swap(nx);
    }
    for (int i = 0; i < pn.count; ++i) grad[pn.offset + i] -= pn.tau * e.d2[i];
  }
  for (int i = 0; i < n; ++i) grad[i] = grad[i] * e.dth[i] + e.dlogj[i];
  *logp = lp;
  return 0;
}

// ---- lock-step chains ------------------------------------------------------------------------------------------------
// A chain = the unchanged run_nuts on a stack of its own.  Its target (fiber_eval) stores the point, switches to the
// scheduler, and finds the value and gradient in place when it is switched back to.
struct Lockstep;
struct Fiber {
  Lockstep* owner = nullptr;
  int chain = 0, group = 0;
  ucontext_t ctx{};
  std::unique_ptr<char[]> stack;
  bool done = false, waiting = false;
  int rc = 0;
  const double* req_x = nullptr;  // the point this chain wants evaluated (dim doubles, owned by the chain)
  double* out_lp = nullptr;
  double* out_grad = nullptr;
  void (*body)(Fiber&) = nullptr;
  void* body_user = nullptr;
};

struct LockstepBackend {  // K points of one group: issue, then collect (value + gradient per point)
  virtual ~LockstepBackend() = default;
  virtual int begin(int group, int k, const int32_t* chains, const double* xs) = 0;
  virtual int end(int group, int k, double* lps, double* grads) = 0;
};

thread_local double last_stats[6] = {0, 0, 0, 0, 0, 0};  // gwi_nuts_lockstep_stats

struct Lockstep {
  int dim = 0;
  ucontext_t sched{};
  bool failed = false;
  std::vector<Fiber> fibers;
  static constexpr size_t kStack = 512 * 1024;  // run_nuts recurses to max_tree_depth with a few hundred bytes per frame (vectors live on the heap)

  static void entry(unsigned lo, unsigned hi) {
    Fiber& f = *reinterpret_cast<Fiber*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    f.body(f);
    f.done = true;  // returning switches to uc_link = the scheduler
  }
  // the chain's target: hand the point over, sleep until the group's batch is back
  static int fiber_eval(Fiber& f, const double* x, double* lp, double* grad) {
    if (f.owner->failed) return 1;
    f.req_x = x;
    f.out_lp = lp;
    f.out_grad = grad;
    f.waiting = true;
    swapcontext(&f.ctx, &f.owner->sched);
    return f.owner->failed ? 1 : 0;
  }
  void resume(Fiber& f) {
    f.waiting = false;
    swapcontext(&sched, &f.ctx);
  }
  void start(Fiber& f) {
    f.stack.reset(new char[kStack]);
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack.get();
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = &sched;
    const uintptr_t p = reinterpret_cast<uintptr_t>(&f);
    makecontext(&f.ctx, (void (*)())entry, 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
    resume(f);  // runs to its first evaluation (or to its end)
  }

  // slots <= 0: chains of group g = fibers with .group == g, all started at once.  slots > 0: a QUEUE of chains -- every group
  // runs at most `slots` chains at a time and a chain that ends hands its slot to the next chain that has not started yet
  // (whichever group asks first), so the batches stay full until the queue is empty; a chain's draws do not depend on when or
  // where it ran.  Returns 0, or 1 when the backend failed (every chain then unwinds).
  int run(LockstepBackend& be, int n_groups, int slots = 0) {
    struct Group {
      std::vector<int> members, live;
      std::vector<int32_t> ids;
      Vec xs, lps, grads;
      bool pending = false;
    };
    std::vector<Group> groups(n_groups);
    size_t next_queued = fibers.size();
    auto top_up = [&](int g) {  // queue mode: finished chains leave the group, queued ones take their slots
      Group& G = groups[g];
      size_t kept = 0;
      for (int i : G.members)
        if (!fibers[i].done) G.members[kept++] = i;
      G.members.resize(kept);
      while ((int)G.members.size() < slots && next_queued < fibers.size()) {
        Fiber& f = fibers[next_queued];
        f.group = g;
        start(f);  // runs to its first evaluation (or to its end: a starting point without likelihood)
        if (!f.done) G.members.push_back((int)next_queued);
        ++next_queued;
      }
    };
    if (slots > 0) {
      next_queued = 0;
      for (int g = 0; g < n_groups; ++g) top_up(g);
    } else {
      for (size_t i = 0; i < fibers.size(); ++i) groups[fibers[i].group].members.push_back((int)i);
      for (Fiber& f : fibers) start(f);
    }
    auto issue = [&](int g) {  // gather the points of the group's waiting chains and launch them
      Group& G = groups[g];
      if (slots > 0) top_up(g);
      G.live.clear();
      G.ids.clear();
      for (int i : G.members)
        if (!fibers[i].done && fibers[i].waiting) {
          G.live.push_back(i);
          G.ids.push_back(fibers[i].chain);
        }
      G.pending = false;
      if (G.live.empty() || failed) return;
      const size_t k = G.live.size();
      G.xs.resize(k * dim);
      G.lps.resize(k);
      G.grads.resize(k * dim);
      for (size_t j = 0; j < k; ++j) std::memcpy(G.xs.data() + j * dim, fibers[G.live[j]].req_x, sizeof(double) * dim);
      if (be.begin(g, (int)k, G.ids.data(), G.xs.data()) != 0) {
        failed = true;
        return;
      }
      G.pending = true;
    };
    // GWI_LOCKSTEP_STATS=1: where the wall time went (stderr): collecting (mostly waiting for the GPU), the chains' own
    // arithmetic between two evaluations, gathering + issuing
    const bool print_stats = std::getenv("GWI_LOCKSTEP_STATS") != nullptr;
    const bool stats = true;  // (five clock reads per batch of evaluations)
    using clk = std::chrono::steady_clock;
    double t_end = 0, t_host = 0, t_issue = 0;
    long long n_batches = 0, n_points = 0;
    auto since = [](clk::time_point a) { return std::chrono::duration<double>(clk::now() - a).count(); };
    const clk::time_point t_all = clk::now();
    for (int g = 0; g < n_groups; ++g) issue(g);
    for (bool any = true; any;) {
      any = false;
      for (int g = 0; g < n_groups; ++g) {
        Group& G = groups[g];
        if (!G.pending) continue;
        any = true;
        const size_t k = G.live.size();
        clk::time_point t0 = clk::now();
        if (be.end(g, (int)k, G.lps.data(), G.grads.data()) != 0) failed = true;
        if (stats) {
          t_end += since(t0);
          t0 = clk::now();
          ++n_batches;
          n_points += (long long)k;
        }
        G.pending = false;
        for (size_t j = 0; j < k; ++j) {  // hand the results over; each chain runs on to its next evaluation (G.live is rebuilt by issue() below, not before)
          Fiber& f = fibers[G.live[j]];
          if (!failed) {
            *f.out_lp = G.lps[j];
            std::memcpy(f.out_grad, G.grads.data() + j * dim, sizeof(double) * dim);
          }
          resume(f);
        }
        if (stats) {
          t_host += since(t0);
          t0 = clk::now();
        }
        issue(g);
        if (stats) t_issue += since(t0);
      }
    }
    last_stats[0] = (double)n_batches;
    last_stats[1] = (double)n_points;
    last_stats[2] = t_end;
    last_stats[3] = t_host;
    last_stats[4] = t_issue;
    last_stats[5] = since(t_all);
    if (print_stats && n_batches > 0)
      std::fprintf(stderr, "[gwi lockstep] %lld batches, %.2f points each; per batch: collect %.1f us, chains %.1f us, issue %.1f us; wall %.3f s\n", n_batches,
                   (double)n_points / n_batches, 1e6 * t_end / n_batches, 1e6 * t_host / n_batches, 1e6 * t_issue / n_batches, since(t_all));
    if (failed)  // let every chain that is still asleep unwind (its target now returns 1 without switching)
      for (Fiber& f : fibers)
        while (!f.done) resume(f);
    return failed ? 1 : 0;
  }
};

struct CallbackBackend : LockstepBackend {  // an arbitrary batched target: evaluated at collect time
  gwi_batch_target_fn fn;
  void* user;
  int dim;
  std::vector<int32_t> ids;
  Vec xs;
  int begin(int, int k, const int32_t* chains, const double* x) override {
    ids.assign(chains, chains + k);
    xs.assign(x, x + (size_t)k * dim);
    return 0;
  }
  int end(int, int k, double* lps, double* grads) override { return fn(user, k, ids.data(), xs.data(), lps, grads) != 0; }
};

struct EngineBackend : LockstepBackend {  // group g = engine handles[g]; the points are constrained hyper-parameters
  const gwi_handle* handles;
  gwi_options lopt;
  std::vector<gwi_summary> summaries;
  int begin(int g, int k, const int32_t*, const double* thetas) override { return gwi_eval_batch_begin(handles[g], thetas, k, &lopt, 1, 0) != GWI_OK; }
  int end(int g, int k, double* lls, double* grads) override {
    summaries.resize(k);
    if (gwi_eval_batch_end(handles[g], summaries.data(), grads, nullptr, nullptr, nullptr, nullptr) != GWI_OK) return 1;
    for (int j = 0; j < k; ++j) lls[j] = summaries[j].log_likelihood;
    return 0;
  }
};

void constrain_draws(double* out, size_t ns, int n_theta, const gwi_param_prior* priors) {  // unconstrained draws -> hyper-parameters
  for (size_t k = 0; k < ns; ++k)
    for (int i = 0; i < n_theta; ++i) {
      double& v = out[k * n_theta + i];
      if (priors[i].kind == GWI_BIJECT_INTERVAL)
        v = priors[i].lo + (priors[i].hi - priors[i].lo) / (1.0 + std::exp(-v));
      else if (priors[i].kind == GWI_BIJECT_POSITIVE)
        v = std::exp(v);
      else if (priors[i].kind == GWI_BIJECT_FIXED)
        v = priors[i].lo;
    }
}

}  // namespace

extern "C" {

gwi_status gwi_nuts_run(gwi_target_fn fn, void* user, int32_t dim, const double* x0, const gwi_nuts_options* opt, double* samples, double* logp, int32_t* tree_depth,
                        gwi_nuts_result* result) {
  if (!fn || dim < 1 || !x0 || !opt || !samples || opt->n_warmup < 0 || opt->n_samples < 0) return GWI_ERR_INVALID;
  Target t{fn, user, dim};
  const int rc = run_nuts(t, x0, *opt, samples, logp, tree_depth, result);
  return rc == 0 ? GWI_OK : (rc == 2 ? GWI_ERR_INVALID : GWI_ERR_HIP);
}

gwi_status gwi_nuts_engine(const gwi_handle* handles, int32_t n_chains, int32_t n_theta, const gwi_options* lopt, const gwi_param_prior* priors, const gwi_smoothing_penalty* pens,
                           int32_t n_pens, const double* u0, const gwi_nuts_options* opt, double* samples, double* logp, int32_t* tree_depth, gwi_nuts_result* results) {
  if (!handles || n_chains < 1 || n_theta < 1 || !lopt || !priors || !u0 || !opt || !samples) return GWI_ERR_INVALID;
  for (int k = 0; k < n_pens; ++k)
    if (!pens || pens[k].offset < 0 || pens[k].count < 2 || pens[k].offset + pens[k].count > n_theta || pens[k].degree < 1 || pens[k].degree >= pens[k].count) return GWI_ERR_INVALID;
  std::vector<int> rc(n_chains, 0);
  auto chain = [&](int c) {
    if (n_chains > 1) (void)gwi_pin_thread_to_engine(handles[c]);  // worker threads: stay on the GPU's side of the machine
    EngineTarget e{handles[c], *lopt, priors, pens, n_pens, n_theta, Vec(n_theta), Vec(n_theta), Vec(n_theta), Vec(n_theta), {}, {}};
    Target t{engine_target, &e, n_theta};
    gwi_nuts_options o = *opt;
    o.seed = opt->seed + 1000ULL * (unsigned long long)c;
    const size_t ns = (size_t)opt->n_samples;
    double* out = samples + (size_t)c * ns * n_theta;
    rc[c] = run_nuts(t, u0 + (size_t)c * n_theta, o, out, logp ? logp + (size_t)c * ns : nullptr, tree_depth ? tree_depth + (size_t)c * ns : nullptr, results ? results + c : nullptr);
    if (rc[c] == 0) constrain_draws(out, ns, n_theta, priors);
  };
  if (n_chains == 1) {
    chain(0);
  } else {  // one host thread per chain: every chain has its own engine handle (stream, buffers, catalog copy)
    std::vector<std::thread> workers;
    for (int c = 0; c < n_chains; ++c) workers.emplace_back(chain, c);
    for (auto& w : workers) w.join();
  }
  for (int c = 0; c < n_chains; ++c)
    if (rc[c] != 0) return rc[c] == 2 ? GWI_ERR_INVALID : GWI_ERR_HIP;
  return GWI_OK;
}

void gwi_nuts_lockstep_stats(double* out6) {
  if (out6) std::memcpy(out6, last_stats, sizeof(last_stats));
}

gwi_status gwi_nuts_run_lockstep(gwi_batch_target_fn fn, void* user, int32_t dim, int32_t n_chains, const double* x0, const gwi_nuts_options* opt, double* samples,
                                 double* logp, int32_t* tree_depth, gwi_nuts_result* results) {
  return gwi_nuts_run_queue(fn, user, dim, n_chains, 0, x0, opt, samples, logp, tree_depth, results);
}

gwi_status gwi_nuts_run_queue(gwi_batch_target_fn fn, void* user, int32_t dim, int32_t n_chains, int32_t slots, const double* x0, const gwi_nuts_options* opt, double* samples,
                              double* logp, int32_t* tree_depth, gwi_nuts_result* results) {
  if (!fn || dim < 1 || n_chains < 1 || slots < 0 || !x0 || !opt || !samples || opt->n_warmup < 0 || opt->n_samples < 0) return GWI_ERR_INVALID;
  struct Job {
    const double* x0;
    gwi_nuts_options o;
    double *samples, *logp;
    int32_t* depth;
    gwi_nuts_result* res;
    int dim;
  };
  std::vector<Job> jobs(n_chains);
  Lockstep ls;
  ls.dim = dim;
  ls.fibers.resize(n_chains);
  const size_t ns = (size_t)opt->n_samples;
  for (int c = 0; c < n_chains; ++c) {
    gwi_nuts_options o = *opt;
    o.seed = opt->seed + 1000ULL * (unsigned long long)c;
    jobs[c] = Job{x0 + (size_t)c * dim, o, samples + (size_t)c * ns * dim, logp ? logp + (size_t)c * ns : nullptr, tree_depth ? tree_depth + (size_t)c * ns : nullptr,
                  results ? results + c : nullptr, dim};
    Fiber& f = ls.fibers[c];
    f.owner = &ls;
    f.chain = c;
    f.group = 0;
    f.body_user = &jobs[c];
    f.body = [](Fiber& fb) {
      Job& j = *static_cast<Job*>(fb.body_user);
      Target t{[](void* u, const double* x, double* lp, double* g) -> int32_t { return Lockstep::fiber_eval(*static_cast<Fiber*>(u), x, lp, g); }, &fb, j.dim};
      fb.rc = run_nuts(t, j.x0, j.o, j.samples, j.logp, j.depth, j.res);
    };
  }
  CallbackBackend be;
  be.fn = fn;
  be.user = user;
  be.dim = dim;
  const int failed = ls.run(be, 1, slots);
  if (failed) return GWI_ERR_HIP;
  for (int c = 0; c < n_chains; ++c)
    if (ls.fibers[c].rc != 0) return ls.fibers[c].rc == 2 ? GWI_ERR_INVALID : GWI_ERR_HIP;
  return GWI_OK;
}

gwi_status gwi_nuts_engine_lockstep(const gwi_handle* handles, int32_t n_groups, int32_t chains_per_group, int32_t n_theta, const gwi_options* lopt, const gwi_param_prior* priors,
                                    const gwi_smoothing_penalty* pens, int32_t n_pens, const double* u0, const gwi_nuts_options* opt, double* samples, double* logp,
                                    int32_t* tree_depth, gwi_nuts_result* results) {
  if (n_groups < 1 || chains_per_group < 1) return GWI_ERR_INVALID;
  return gwi_nuts_engine_queue(handles, n_groups, chains_per_group, n_groups * chains_per_group, n_theta, lopt, priors, pens, n_pens, u0, opt, samples, logp, tree_depth, results);
}

gwi_status gwi_nuts_engine_queue(const gwi_handle* handles, int32_t n_groups, int32_t slots_per_group, int32_t n_chains, int32_t n_theta, const gwi_options* lopt,
                                 const gwi_param_prior* priors, const gwi_smoothing_penalty* pens, int32_t n_pens, const double* u0, const gwi_nuts_options* opt, double* samples,
                                 double* logp, int32_t* tree_depth, gwi_nuts_result* results) {
  const int chains_per_group = slots_per_group;
  if (!handles || n_groups < 1 || slots_per_group < 1 || n_chains < 1 || n_theta < 1 || !lopt || !priors || !u0 || !opt || !samples) return GWI_ERR_INVALID;
  for (int k = 0; k < n_pens; ++k)
    if (!pens || pens[k].offset < 0 || pens[k].count < 2 || pens[k].offset + pens[k].count > n_theta || pens[k].degree < 1 || pens[k].degree >= pens[k].count) return GWI_ERR_INVALID;
  // exactly n_groups x slots chains: the static assignment (chain c in group c / slots, all started at once); otherwise a queue
  const bool queued = n_chains != n_groups * slots_per_group;
  struct Job {
    EngineTarget e;
    const double* u0;
    gwi_nuts_options o;
    double *samples, *logp;
    int32_t* depth;
    gwi_nuts_result* res;
  };
  std::vector<Job> jobs(n_chains);
  Lockstep ls;
  ls.dim = n_theta;
  ls.fibers.resize(n_chains);
  const size_t ns = (size_t)opt->n_samples;
  for (int c = 0; c < n_chains; ++c) {
    gwi_nuts_options o = *opt;
    o.seed = opt->seed + 1000ULL * (unsigned long long)c;
    Fiber& f = ls.fibers[c];
    f.owner = &ls;
    f.chain = c;
    f.group = queued ? 0 : c / chains_per_group;  // (queued chains learn their group when they start)
    Job& j = jobs[c];
    j.e = EngineTarget{handles[f.group], *lopt, priors, pens, n_pens, n_theta, Vec(n_theta), Vec(n_theta), Vec(n_theta), Vec(n_theta), {}, {}};
    // the chain's likelihood evaluation = a slot in its group's batched launch
    j.e.eval_hook = [](void* u, const double* theta, double* ll, double* g) -> int { return Lockstep::fiber_eval(*static_cast<Fiber*>(u), theta, ll, g); };
    j.e.hook_user = &f;
    j.u0 = u0 + (size_t)c * n_theta;
    j.o = o;
    j.samples = samples + (size_t)c * ns * n_theta;
    j.logp = logp ? logp + (size_t)c * ns : nullptr;
    j.depth = tree_depth ? tree_depth + (size_t)c * ns : nullptr;
    j.res = results ? results + c : nullptr;
    f.body_user = &j;
    f.body = [](Fiber& fb) {
      Job& jb = *static_cast<Job*>(fb.body_user);
      Target t{engine_target, &jb.e, jb.e.n_theta};
      fb.rc = run_nuts(t, jb.u0, jb.o, jb.samples, jb.logp, jb.depth, jb.res);
    };
  }
  EngineBackend be;
  be.handles = handles;
  be.lopt = *lopt;
  const int failed = ls.run(be, n_groups, queued ? slots_per_group : 0);
  if (failed) return GWI_ERR_HIP;
  for (int c = 0; c < n_chains; ++c) {
    if (ls.fibers[c].rc != 0) return ls.fibers[c].rc == 2 ? GWI_ERR_INVALID : GWI_ERR_HIP;
    constrain_draws(samples + (size_t)c * ns * n_theta, ns, n_theta, priors);
  }
  return GWI_OK;
}

}  // extern "C"
