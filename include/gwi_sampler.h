/* gwi_sampler.h -- C ABI of the engine library's host-side No-U-Turn sampler.
 *
 * Stands in for the sampler the reference drives its model with, numpyro.infer.NUTS / MCMC
 * (examples/utils.py:63-85; tests/inference_test.py:367-411), in environments without JAX/NumPyro: every
 * leapfrog step is one gwi_eval (value + gradient), and nothing but C++ runs between two evaluations.
 * Multinomial NUTS, generalised U-turn criterion, dual-averaging step size, one diagonal mass-matrix update
 * during warm-up.  gwinferno_amd/sampling.py is the NumPy statement of the same algorithm. */
#ifndef GWI_SAMPLER_H
#define GWI_SAMPLER_H

#include "gwi_engine.h"

#ifdef __cplusplus
extern "C" {
#endif

/* log-probability and gradient of the target at x[dim]; returns 0, or non-zero to abort the run */
typedef int32_t (*gwi_target_fn)(void* user, const double* x, double* log_prob, double* grad);

typedef struct {
  int32_t n_warmup, n_samples;
  int32_t max_tree_depth; /* <= 0: 10 */
  int32_t reserved;
  double target_accept;   /* 0.8 is the usual choice */
  uint64_t seed;
} gwi_nuts_options;

typedef struct {
  double accept_rate, step_size;
  int64_t n_evals;
  int32_t n_divergent, reserved; /* divergent transitions after warm-up */
} gwi_nuts_result;

/* Status: GWI_ERR_INVALID also when a chain's starting point has zero probability (log-probability -inf, NaN or the
 * nan_to_num(-inf) value a likelihood cut returns) or a non-finite gradient; GWI_ERR_HIP when the target failed.
 *
 * One chain on an arbitrary target.  samples[n_samples][dim], log_prob[n_samples] and tree_depth[n_samples]
 * (the last two nullable) receive the post-warm-up draws. */
gwi_status gwi_nuts_run(gwi_target_fn fn, void* user, int32_t dim, const double* x0, const gwi_nuts_options* opt, double* samples, double* log_prob,
                        int32_t* tree_depth, gwi_nuts_result* result);

/* Change of variables per hyper-parameter (what numpyro's biject_to(support) does for Uniform / HalfNormal sites,
 * e.g. examples/simple_powerlaw_peak_example.py:52-77) and an optional Normal(0, sigma) prior on its constrained value. */
enum {
  GWI_BIJECT_IDENTITY = 0,
  GWI_BIJECT_INTERVAL = 1, /* (lo, hi) via the logistic map */
  GWI_BIJECT_POSITIVE = 2, /* exp */
  GWI_BIJECT_FIXED = 3     /* theta pinned to `lo` (pipeline/utils.py:213-214: z_cs[0] = 0); the coordinate is a N(0,1) dummy */
};
typedef struct {
  int32_t kind, reserved;
  double lo, hi;  /* GWI_BIJECT_INTERVAL */
  double sigma;   /* Normal(0, sigma) prior; +inf (or <= 0): flat */
} gwi_param_prior;

/* P-spline penalty  -0.5 tau ||Delta^degree theta[offset .. offset+count)||^2  (models/bsplines/smoothing.py:8-28,
 * callers pipeline/utils.py:163-216) */
typedef struct {
  int32_t offset, count, degree, reserved;
  double tau;
} gwi_smoothing_penalty;

/* n_chains chains, chain c on engine handles[c] (its own stream, buffers and catalog copy) in a host thread of its
 * own; target = log-likelihood (gwi_eval with *lopt) + priors + penalties + log-Jacobian, sampled in unconstrained
 * coordinates from u0[c][n_theta].  samples[c][n_samples][n_theta] are returned in CONSTRAINED coordinates;
 * log_prob / tree_depth [c][n_samples] and results[c] are nullable.  Chain c uses seed opt->seed + 1000 c.
 * A handle with a communicator (gwi_comm_init) is evaluated with gwi_eval_sharded: call this on every rank with
 * the same arguments (one chain per process) -- all ranks then walk the same trajectory, bit for bit. */
gwi_status gwi_nuts_engine(const gwi_handle* handles, int32_t n_chains, int32_t n_theta, const gwi_options* lopt, const gwi_param_prior* priors,
                           const gwi_smoothing_penalty* penalties, int32_t n_penalties, const double* u0, const gwi_nuts_options* opt, double* samples, double* log_prob,
                           int32_t* tree_depth, gwi_nuts_result* results);

/* Lock-step chains -- numpyro's chain_method="vectorized" (examples/utils.py:63-85 with num_chains > 1 on one device): all
 * chains are advanced together and every leapfrog step of all of them is ONE batched evaluation.
 *
 * On an arbitrary batched target: fn evaluates the k points xs[k][dim] of the chains chain_ids[k] (k shrinks as chains
 * finish; chains whose trees need fewer steps in an iteration start their next iteration -- nothing idles) and writes
 * log_probs[k], grads[k][dim].  x0[n_chains][dim]; outputs as gwi_nuts_run's with a leading chain axis; chain c uses seed
 * opt->seed + 1000 c and draws, bit for bit, what gwi_nuts_run draws with that seed on the same target. */
typedef int32_t (*gwi_batch_target_fn)(void* user, int32_t k, const int32_t* chain_ids, const double* xs, double* log_probs, double* grads);
gwi_status gwi_nuts_run_lockstep(gwi_batch_target_fn fn, void* user, int32_t dim, int32_t n_chains, const double* x0, const gwi_nuts_options* opt, double* samples,
                                 double* log_prob, int32_t* tree_depth, gwi_nuts_result* results);

/* ... and on engines: n_groups * chains_per_group chains, the chains of group g on handles[g] through gwi_eval_batch_begin /
 * gwi_eval_batch_end (chains_per_group <= that engine's max_batch: 16 unless GWI_MAX_BATCH says otherwise) -- the batched
 * kernels (several points per workgroup for parametric chains, the matrix-core kernel for spline models).  One host thread:
 * while group g's launches run, the chains of the other groups do their host arithmetic and issue theirs, so two or three
 * groups keep the GPU busy.  Target, arguments and outputs as gwi_nuts_engine (u0[n_chains][n_theta], chain c in group
 * c / chains_per_group).  Handles without a communicator. */
gwi_status gwi_nuts_engine_lockstep(const gwi_handle* handles, int32_t n_groups, int32_t chains_per_group, int32_t n_theta, const gwi_options* lopt,
                                    const gwi_param_prior* priors, const gwi_smoothing_penalty* penalties, int32_t n_penalties, const double* u0, const gwi_nuts_options* opt,
                                    double* samples, double* log_prob, int32_t* tree_depth, gwi_nuts_result* results);

/* A QUEUE of chains on the batched kernels: n_chains chains (any number) over n_groups x slots_per_group slots.  Every group runs
 * at most slots_per_group chains at a time; a chain that has drawn its last sample hands its slot to the next chain that has not
 * started yet, so the batches stay full until the queue is empty instead of shrinking as the quick chains finish (chains started
 * from different points need up to 15 x different numbers of evaluations: gwi_nuts_lockstep_stats).  Chain c still draws, bit for
 * bit, what gwi_nuts_run draws with seed opt->seed + 1000 c -- when and in which group it ran does not enter.  With n_chains ==
 * n_groups x slots_per_group this is gwi_nuts_engine_lockstep (static assignment).  numpyro's counterpart: num_chains >
 * device count under chain_method="parallel" (examples/utils.py:62, 118), run in rounds there.
 * gwi_nuts_run_queue: the same for an arbitrary batched target (one group; slots = 0: every chain at once). */
gwi_status gwi_nuts_engine_queue(const gwi_handle* handles, int32_t n_groups, int32_t slots_per_group, int32_t n_chains, int32_t n_theta, const gwi_options* lopt,
                                 const gwi_param_prior* priors, const gwi_smoothing_penalty* penalties, int32_t n_penalties, const double* u0, const gwi_nuts_options* opt,
                                 double* samples, double* log_prob, int32_t* tree_depth, gwi_nuts_result* results);
gwi_status gwi_nuts_run_queue(gwi_batch_target_fn fn, void* user, int32_t dim, int32_t n_chains, int32_t slots, const double* x0, const gwi_nuts_options* opt, double* samples,
                              double* log_prob, int32_t* tree_depth, gwi_nuts_result* results);

/* Of the calling thread's last lock-step run: out6 = { batched evaluations made, points in them (their quotient is the mean
 * batch size: chains whose trees need fewer steps finish earlier and the batches shrink -- a run whose chains need very
 * different numbers of evaluations gains little from lock step), seconds collecting (mostly waiting for the GPU), seconds in the
 * chains' own arithmetic, seconds gathering + issuing, wall seconds }.  GWI_LOCKSTEP_STATS=1 prints the same to stderr. */
void gwi_nuts_lockstep_stats(double* out6);

#ifdef __cplusplus
}
#endif
#endif /* GWI_SAMPLER_H */
