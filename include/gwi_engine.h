/* gwi_engine.h -- C ABI of the MI355X-native hierarchical population-likelihood engine.
 *
 * GWInferno (the reference) has no FFI layer: its hot path is a set of Python calling
 * conventions.  This header is the C-ABI seam a binding would target; every
 * entry point names the reference interface it stands in for (paths relative to the reference
 * root).  The library behind it is gwinferno_amd/_lib/libgwi_engine.so (hand-written HIP for
 * gfx950); there is NO CPU fallback -- without a usable GPU gwi_create() fails with
 * GWI_ERR_NO_DEVICE.
 *
 * Data model
 *   A *catalog* is two sample sets that share one column schema:
 *     PE set         n_ev x n_pe samples, event-major (pedict[param] : (N_ev, N_pe),
 *                    gwinferno/pipeline/utils.py:82-84)
 *     injection set  n_inj samples        (injdict[param] : (N_inj,), pipeline/utils.py:86)
 *   Columns are float64 and hold per-sample quantities that do not depend on the
 *   hyper-parameters (log m1, log q, log(1+z), spline coordinates, ...).  One column, `kappa`,
 *   holds log(dVc/dz) - log(prior) with -inf for samples any static truncation excludes
 *   (models/bsplines/single.py:54-55, distributions.py:119,143,162, parametric.py:141-145,
 *   tests/inference_test.py:172).
 *   A *model* is a product of terms (gwi_term); each term reads <= 2 columns and a few entries of
 *   the flat hyper-parameter vector theta.  Grid normalisers (interpolation.py:280-291,
 *   parametric.py:123-124, spline_perturbation.py:323-336) are described by gwi_norm.
 *
 * Threading: one handle = one device + one HIP stream; gwi_eval* are not re-entrant per handle;
 * distinct handles are independent.  Ownership: the caller owns every host buffer passed in or
 * out; the engine copies inputs during gwi_create and owns all device memory.  Errors: integer
 * status, never C++ exceptions; non-finite likelihood values are VALUES (reference semantics,
 * jnp.nan_to_num at pipeline/analysis.py:280-315), not errors.
 */
#ifndef GWI_ENGINE_H
#define GWI_ENGINE_H

#ifndef __HIPCC_RTC__ /* hipRTC defines the fixed-width integer types itself and has no system headers */
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define GWI_ABI_VERSION 3 /* 2: GWI_TERM_PLPEAK takes ONE column (log x); 3: GWI_MAX_NORMS 8 -> 12, GWI_MAX_COLS 16 -> 32 (gwi_spec grows) */
#define GWI_MAX_TERMS 12
#define GWI_MAX_THETA 256
#define GWI_MAX_NORMS 12 /* one grid normaliser per term at most */
#define GWI_MAX_COLS 32  /* twelve terms x two columns + kappa fit */

typedef int32_t gwi_status;
enum {
  GWI_OK = 0,
  GWI_ERR_INVALID = -1,     /* malformed spec / argument */
  GWI_ERR_NO_DEVICE = -2,   /* no usable gfx950 device: the engine never falls back to the CPU */
  GWI_ERR_HIP = -3,         /* a HIP runtime call failed; see gwi_last_error() */
  GWI_ERR_UNSUPPORTED = -4, /* gradient of marginalize_selection via gwi_combine; host placement unknown */
  GWI_ERR_TIMEOUT = -5
};

/* Term kinds.  `cols[]` index the catalog's column table, `theta[]` index the flat
 * hyper-parameter vector, `p[]` are fixed constants. */
enum {
  /* x^alpha on the fixed interval [lo,hi]           distributions.py:100-119 (scalar bounds)
   * cols[0]=log x; theta[0]=alpha; p[0]=lo, p[1]=hi */
  GWI_TERM_POWERLAW = 1,
  /* (1-lam) PL(x;alpha,lo,hi) + lam TN(x;mpp,sigpp,lo,hi)   parametric.py:49-53 (delta=None)
   * cols[0]=log x (the kernels form x = exp(log x) for the Gaussian component themselves: one exponential per sample instead
   * of a second 8-byte column -- config 2 then streams its algorithmic 32 B per sample, not 40); theta = alpha, mpp, sigpp, lam;
   * p[0]=lo, p[1]=hi */
  GWI_TERM_PLPEAK = 2,
  /* q^beta on [mmin/m1, 1]: powerlaw_pdf(q, beta, mmin/m1, 1)   parametric.py:28,40; separable.py:364
   * cols[0]=log q, cols[1]=log m1; theta[0]=beta; p[0]=log(mmin).
   * With GWI_RATIO_LOGM_FROM_SPLINE in flags, cols[1] is the coordinate column of a spline term in log m1 of the same model
   * (BSplinePrimaryPowerlawRatio, separable.py:295-365: the LogXLogYBSpline of m1) and p[1], p[2], p[3] = that spline's lo,
   * (n_basis - 3) / (hi - lo), (hi - lo) / (n_basis - 3): the engine keeps that column as the knot coordinate u and this term
   * forms log m1 = lo + u dx itself -- one column per sample less (config 3: 64 B per sample, the algorithmic figure, not 72).
   * The caller guarantees that every sample alive for this term lies inside that spline's domain (the spline's own mask). */
  GWI_TERM_POWERLAW_RATIO = 3,
  /* Beta(a; alpha, beta) on [0, 1]                  distributions.py:146-162, parametric.py:63-81
   * cols[0]=log a, cols[1]=log(1-a); theta = alpha, beta */
  GWI_TERM_BETA = 4,
  /* (1-xi)/2 + xi TN(ct; 1, sigma, -1, 1)           parametric.py:84-86
   * cols[0]=cos tilt; theta = xi, sigma */
  GWI_TERM_TILT_MIXTURE = 5,
  /* dVc/dz (1+z)^(lamb-1) / Z(lamb)                 parametric.py:112-145
   * cols[0]=log(1+z) (log dVc/dz lives in kappa); theta[0]=lamb; norm = grid normaliser id */
  GWI_TERM_POWERLAW_REDSHIFT = 6,
  /* exp(sum_k c_k B_k(x)) [/ Z(c)] on uniform cubic B-splines   interpolation.py:360-449,
   * models/bsplines/single.py:77-109, spline_perturbation.py:352
   * cols[0]=spline coordinate (x or log x); coef_off/n_basis; p[0]=lo, p[1]=hi of the coordinate;
   * flags: GWI_SPLINE_OUTSIDE_ZERO_EXPONENT; norm = grid normaliser id or -1 */
  GWI_TERM_EXP_SPLINE = 7,
  /* truncated normal TN(x; mu, sigma, lo, hi)       distributions.py:122-143 (log=False)
   * cols[0]=x; theta = mu, sigma; p[0]=lo, p[1]=hi */
  GWI_TERM_TRUNCNORM = 8,
  /* sum_k c_k B_k(x) [/ Z(c)]: linear-Y B-spline density (BSpline basis)   interpolation.py:293-317,
   * models/bsplines/single.py:199-318 (chi_eff, chi_p)
   * cols[0]=spline coordinate; coef_off/n_basis; p[0]=lo, p[1]=hi; norm = grid normaliser (linear) or -1.
   * Non-positive values of the spline count as zero density. */
  GWI_TERM_LINEAR_SPLINE = 9,
  /* (1-xi)/4 + xi TN(ct1;1,sigma,-1,1) TN(ct2;1,sigma,-1,1)   parametric.py:97-102 (default_spin_tilt)
   * cols = cos tilt 1, cos tilt 2; theta = xi, sigma */
  GWI_TERM_TILT_JOINT = 10,
  /* low-mass taper as the reference evaluates it: 1 / (1 + exp(d/(x-xmin) + d/(x-xmin-d))) for EVERY x
   * (distributions.py:16-21: the second `where` condition holds for all x), e.g. smooth(delta, q m1, mmin)
   * of plpeak_primary_ratio_pdf (parametric.py:39-46).  cols[0] = x - xmin; theta[0] = d (delta) */
  GWI_TERM_SMOOTH = 11,
  /* (1-lam) PL(x) smooth(delta, x, lo) + lam TN(x): plpeak_primary_pdf with delta (parametric.py:49-53).
   * cols[0]=x, cols[1]=log x; theta = alpha, mpp, sigpp, lam; coef_off = theta index of delta; p[0]=lo, p[1]=hi */
  GWI_TERM_PLPEAK_SMOOTH = 12,
  /* x^alpha on [lo,hi] with the BOUNDS hyper-parameters too: Powerlaw.log_prob with sampled minimum / maximum
   * (numpyro_distributions.py:101-136; examples/config_files/config.yml:8-25).  cols[0]=log x, cols[1]=x; theta = alpha, lo, hi.
   * x < lo | x > hi is excluded (a sample exactly on a bound is inside); the gradient w.r.t. lo / hi is 0 (the normaliser cancels in log_l and the
   * truncation itself is piecewise constant), as reverse-mode differentiation of the reference gives. */
  GWI_TERM_POWERLAW_BOUNDS = 13,
  /* exp(interp(x, grid, lpdfs)) / Z, lpdfs_g = sum_k c_k B_k(us_g): BSplineDistribution.log_prob
   * (numpyro_distributions.py:266-293).  cols[0] = fractional grid index of the sample (j + f, clamped to the grid as
   * np.interp holds the end values); coef_off/n_basis; p[0]=lo, p[1]=hi of the spline coordinate; norm = the grid
   * normaliser whose `us` table (spline coordinate per grid point) and trapezoid weights define grid and Z (required);
   * flags: GWI_SPLINE_OUTSIDE_ZERO_EXPONENT as for EXP_SPLINE (otherwise a grid point outside [lo,hi] has lpdf -inf) */
  GWI_TERM_EXP_SPLINE_LERP = 14
};

/* POWERLAW flag: bare x^alpha with no normaliser and no truncation (the (m2/m1)^beta pairing factor,
 * models/bsplines/separable.py:609-613, :703). */
#define GWI_POWERLAW_UNNORMALISED 2
#define GWI_RATIO_LOGM_FROM_SPLINE 8 /* GWI_TERM_POWERLAW_RATIO: see there */
/* gwi_norm.spline_flags bit: the integrand is the linear spline itself, Z = sum_g tw_g sum_k c_k B_k
 * (BSpline.norm, interpolation.py:280-291), not exp(...) of it. */
#define GWI_NORM_LINEAR_SPLINE 4

/* EXP_SPLINE flag: outside [lo,hi] the basis is 0 (BSpline/LogXBSpline.bases,
 * interpolation.py:175) so the factor is exp(0)=1, instead of the sample being excluded
 * (LogY bases, :407,:449 -- those are folded into kappa by the caller). */
#define GWI_SPLINE_OUTSIDE_ZERO_EXPONENT 1

typedef struct {
  int32_t kind;
  int32_t cols[2];
  int32_t theta[4];
  int32_t n_basis;  /* EXP_SPLINE */
  int32_t coef_off; /* EXP_SPLINE: theta offset of c_0 */
  int32_t flags;
  int32_t norm;     /* index into gwi_spec.norms of the normaliser dividing this term, or -1 */
  int32_t reserved;
  double p[4];
} gwi_term;

/* Grid normaliser  Z = sum_g tw[g] * exp( lb[g] + (theta[expo_theta]+expo_add) * l1[g]
 *                                        + sum_k theta[coef_off+k] B_k(us[g]) )
 * tw = trapezoid weights of the reference's grid (0 where the reference's integrand is masked). */
typedef struct {
  int32_t n_pts;
  int32_t expo_theta; /* -1: no power-law factor */
  int32_t n_basis;    /* 0: no spline factor */
  int32_t coef_off;
  int32_t spline_flags;
  int32_t reserved;
  double expo_add;
  double lo, hi;      /* spline coordinate domain */
  const double* tw;
  const double* lb;   /* may be NULL (zeros) */
  const double* l1;   /* may be NULL iff expo_theta < 0 */
  const double* us;   /* may be NULL iff n_basis == 0 */
} gwi_norm;

typedef struct {
  int32_t abi_version; /* GWI_ABI_VERSION */
  int32_t n_cols;
  int32_t kappa_col;
  int32_t n_theta;
  int32_t n_terms;
  int32_t n_norms;
  int32_t vt_norm;     /* normaliser reported as `surveyed_hypervolume` (analysis.py:267), or -1 */
  int32_t reserved;
  gwi_term terms[GWI_MAX_TERMS];
  gwi_norm norms[GWI_MAX_NORMS];
} gwi_spec;

/* Likelihood options == keyword arguments of hierarchical_likelihood (analysis.py:139-163). */
typedef struct {
  double n_obs;        /* Nobs (global number of events) */
  double total_inj;    /* total_inj */
  int32_t marginalize_selection;
  int32_t min_neff_cut;
  int32_t max_variance_cut;
  int32_t reserved;
} gwi_options;

/* Scalar results of one evaluation; names follow the numpyro sites of analysis.py:260-319. */
typedef struct {
  double log_likelihood;      /* numpyro.factor("log_likelihood") :319 (after every cut) */
  double log_l;               /* site "log_l" :284-291 */
  double sum_logBFs;          /* :282 */
  double selection_factor;    /* :278-281 */
  double log_det_eff;         /* log mu before marginalisation / cuts :259 */
  double log_nEff_inj;        /* :260 */
  double variance_log_detection_efficiency; /* :265 */
  double variance_log_likelihood;           /* :305-308 */
  double min_log_nEff;        /* min_i log n_eff_i (:295) */
  double surveyed_hypervolume_norm; /* Z of spec.vt_norm (raw; the site divides by 1e9, x Tobs) */
  double log_norm_const;      /* sum of sample-independent log-normalisers folded out of the scan */
  double reserved[5];
} gwi_summary;

typedef struct gwi_engine* gwi_handle;

/* device argument of gwi_create: the current HIP device, or a host-only handle that owns no device
 * memory and supports only gwi_prepare_combine / gwi_combine / gwi_partial_len (used by ranks that
 * merely assemble gathered records, and by the CPU test-suite). */
#define GWI_DEVICE_CURRENT (-1)
#define GWI_DEVICE_HOST_ONLY (-2)

/* Build an engine for one catalog + model.  Stands in for the reference's model construction
 * (Base1DBSplineModel.__init__, single.py:35-58; PowerlawRedshiftModel.__init__,
 * parametric.py:113-121): copies the columns to HBM once.  `pe_cols[c]` has n_ev*n_pe entries
 * (event-major), `inj_cols[c]` n_inj.  device < 0 selects the current HIP device. */
gwi_status gwi_create(const gwi_spec* spec, const double* const* pe_cols, int64_t n_ev, int64_t n_pe,
                      const double* const* inj_cols, int64_t n_inj, int32_t device, gwi_handle* out);

/* ---- setup on the device ---------------------------------------------------------
 * The reference prepares its per-sample, hyper-parameter-independent quantities eagerly on the host when the model
 * objects are constructed: validity masks (models/bsplines/single.py:54-55, distributions.py:119,143,162,
 * parametric.py:141-145), logarithms of the mass / ratio / redshift columns (interpolation.py:357,447), dVc/dz per
 * sample by linear interpolation into the comoving-distance table (cosmology.py:95-120, parametric.py:116) and the
 * division by the sampling prior (examples/simple_bspline_example.py:58-71).  gwi_create_ingest() takes the RAW catalog
 * columns (the arrays of pedict / injdict, float64 or float32) and a small register program per sample set that
 * describes those operations; one HIP kernel evaluates the program for every sample and writes the engine's columns
 * (kappa included) straight into HBM.  The program is straight-line code over `n_regs` fp64 registers (booleans are
 * 0.0 / 1.0), the same for every sample:
 *
 *   LOAD  dst <- sources[a][i]           CONST dst <- k
 *   LOG LOG1P NEG ABS NOT SQRT ISFINITE  dst <- f(r[a])
 *   ADD SUB MUL DIV LT GT LE GE AND OR   dst <- r[a] (op) r[b]         (IEEE fp64, never fused)
 *   WHERE dst <- r[a] != 0 ? r[b] : r[c]
 *   INTERP dst <- linear interpolation of r[a] in (tables[b], tables[c]), end values held outside (numpy.interp /
 *                 jnp.interp as used at cosmology.py:111-120); NaN in, NaN out
 *   GRIDINDEX dst <- j + f, the piece j and weight f numpy.interp would use on tables[b]
 *   STORE  column dst <- r[a]
 *
 * Every step except LOG / LOG1P reproduces the host (NumPy) evaluation of the same program to the bit. */
enum {
  GWI_ING_LOAD = 0, GWI_ING_CONST = 1,
  GWI_ING_LOG = 2, GWI_ING_LOG1P = 3, GWI_ING_NEG = 4, GWI_ING_ABS = 5, GWI_ING_NOT = 6, GWI_ING_SQRT = 7, GWI_ING_ISFINITE = 8,
  GWI_ING_ADD = 10, GWI_ING_SUB = 11, GWI_ING_MUL = 12, GWI_ING_DIV = 13, GWI_ING_LT = 14, GWI_ING_GT = 15, GWI_ING_LE = 16,
  GWI_ING_GE = 17, GWI_ING_AND = 18, GWI_ING_OR = 19,
  GWI_ING_WHERE = 20, GWI_ING_INTERP = 21, GWI_ING_GRIDINDEX = 22, GWI_ING_STORE = 23
};
#define GWI_INGEST_MAX_REGS 64
#define GWI_INGEST_MAX_SOURCES 32
#define GWI_INGEST_MAX_TABLES 16
enum { GWI_DTYPE_F64 = 0, GWI_DTYPE_F32 = 1 };

typedef struct gwi_ingest_op {
  int32_t op, dst, a, b, c, reserved;
  double k;
} gwi_ingest_op;

typedef struct gwi_ingest_program {
  int32_t n_ops, n_regs, n_sources, n_tables;
  const gwi_ingest_op* ops;
  const void* const* sources;      /* host arrays, one value per sample of the set, in sample order */
  const int32_t* source_dtype;     /* GWI_DTYPE_F64 | GWI_DTYPE_F32 per source */
  const double* const* tables;     /* host arrays */
  const int64_t* table_len;
} gwi_ingest_program;

/* gwi_create() with the columns computed on the device: `pe` / `inj` must STORE every column 0 .. spec->n_cols-1 of
 * their sample set.  Raw sources are uploaded once (float32 ones as float32) and released after the kernel ran. */
gwi_status gwi_create_ingest(const gwi_spec* spec, const gwi_ingest_program* pe, int64_t n_ev, int64_t n_pe,
                             const gwi_ingest_program* inj, int64_t n_inj, int32_t device, gwi_handle* out);

/* The ingest kernel on its own: run `prog` over n samples on `device` and copy the n_cols columns it stores back to
 * the host (cols[c] has n entries).  What the parity test of the setup path compares with the host evaluation. */
gwi_status gwi_ingest_columns(const gwi_ingest_program* prog, int64_t n, int32_t n_cols, double* const* cols, int32_t device);

/* Copy column `col` of an engine's resident catalog back to the host (`pe_side` != 0: n_ev * n_pe entries, else n_inj).
 * Resident means what the kernels read: a column that only GWI_TERM_EXP_SPLINE / GWI_TERM_LINEAR_SPLINE terms with one set of
 * knots read holds the KNOT coordinate (x - p[0]) * (n_basis - 3) / (p[1] - p[0]) of the spline coordinate x the caller handed
 * over (clamped into [0, n_basis - 3) for exponentiated splines without GWI_SPLINE_OUTSIDE_ZERO_EXPONENT), computed once at
 * gwi_create / gwi_create_ingest; every other column is returned as it was handed over or ingested. */
gwi_status gwi_read_column(gwi_handle h, int32_t pe_side, int32_t col, double* out);

/* One value-and-gradient evaluation == one execution of the user's NumPyro model body ending in
 * hierarchical_likelihood(...) (analysis.py:139-319) under jit(value_and_grad)
 * (tests/inference_test.py:320-326).  theta has spec.n_theta entries.  Nullable outputs:
 * grad[n_theta] = d log_likelihood / d theta; log_bfs / log_neffs / variances [n_ev] = sites
 * "logBFs", "log_nEffs", "variance_log_BFs"; norms[n_norms] = normaliser values Z.
 * A non-finite theta yields the reference's NaN branch: log_likelihood = nan_to_num(-inf), zero gradient
 * (analysis.py:287-289).  With opt->marginalize_selection and a gradient requested, a second launch set
 * with squared weights supplies sum_j w_j^2 dl_j/dtheta (analysis.py:270-271). */
gwi_status gwi_eval(gwi_handle h, const double* theta, const gwi_options* opt, gwi_summary* summary,
                    double* grad, double* log_bfs, double* log_neffs, double* variances, double* norms);

/* The same in two halves, for several handles in flight on one GPU (independent chains whose trajectory
 * lengths differ cannot be batched in lock step, but their evaluations can overlap): gwi_eval_begin() does the
 * host prelude and issues the launches of handle h, gwi_eval_end() waits for and assembles that evaluation.
 * gwi_eval == begin + end.  One evaluation per handle may be pending; every handle has its own stream,
 * buffers and copy of the catalog. */
gwi_status gwi_eval_begin(gwi_handle h, const double* theta, const gwi_options* opt, int32_t want_grad);
gwi_status gwi_eval_end(gwi_handle h, gwi_summary* summary, double* grad, double* log_bfs, double* log_neffs,
                        double* variances, double* norms);

/* k_batch hyper-parameter points in ONE set of launches (blockIdx.y = point): what vectorised
 * multi-chain NUTS evaluates per step.  thetas[k_batch][n_theta] row-major; outputs are arrays of
 * k_batch entries (summaries[k], grads[k][n_theta], log_bfs[k][n_ev], ..., norms[k][n_norms]; any may
 * be NULL).  The catalog is streamed from HBM once per workgroup tile and re-read from L2/Infinity
 * Cache for the other points; launch, combine and host latencies are paid once per batch.
 * k_batch <= GWI_MAX_BATCH (environment, default 16, at most 64). */
gwi_status gwi_eval_batch(gwi_handle h, const double* thetas, int32_t k_batch, const gwi_options* opt, gwi_summary* summaries,
                          double* grads, double* log_bfs, double* log_neffs, double* variances, double* norms);

/* The same in two halves, like gwi_eval_begin / gwi_eval_end: begin does the host prelude and issues the launches of the K points
 * (after, on a spline model's first batch, the blocking measurement of its two batched kernels; and after the blocking
 * squared-weight pass when opt->marginalize_selection and want_grad), end waits for and assembles them.  A blocking batch leaves
 * the GPU to its combine / final launches and the host for a third of its time: two or three sets in flight -- one handle each,
 * ONE host thread -- fill it (config 2, K = 16: 243 k -> 315-360 k evaluations per second).  want_events: the per-event sites
 * will be asked for at gwi_eval_batch_end.  One evaluation or batch per handle may be pending. */
gwi_status gwi_eval_batch_begin(gwi_handle h, const double* thetas, int32_t k_batch, const gwi_options* opt, int32_t want_grad, int32_t want_events);
gwi_status gwi_eval_batch_end(gwi_handle h, gwi_summary* summaries, double* grads, double* log_bfs, double* log_neffs, double* variances, double* norms);

/* Which kernel a batched launch of k_batch points would use: "taps" (one grid row per point, 4-tap gradient into LDS rows;
 * the default) or "mfma" (GWI_BATCH_MFMA=1 at gwi_create, models with spline terms whose term sequence and basis counts
 * have a matrix-core instantiation, k_batch >= 9: the spline-coefficient gradient as a v_mfma_f64_16x16x4 GEMM over 16
 * points per wavefront, gwinferno_amd/csrc/gwi_mfma.h).  Both are kept because both are measured: see DESIGN.md.
 * Models without spline terms: "rows-per-point" (one grid row per point, scan_kernel BATCH: the default since round 6) or
 * "pbatch" (GWI_PBATCH=1 or a row size GWI_PBATCH_PTS: scan_pbatch_kernel, every sample loaded once for the points a workgroup
 * draws; which of the two is faster depends on the box, within 10 %: profiles/round6/EXPERIMENTS.md section 5). */
const char* gwi_batch_path(gwi_handle h, int32_t k_batch);
/* Spline models that have both batched kernels: which one runs follows a STATIC rule by default (matrix cores from 9 points per
 * launch on, up to 8 gradient tiles; otherwise the 4-tap kernel) -- the two kernels sum in different orders, so the same model on
 * the same catalog must not get one or the other from a race of wall times.  The environment can name a path (GWI_BATCH_MFMA,
 * GWI_BATCH_ROWS) or, with GWI_BATCH_AUTOTUNE=1, ask for a measurement on the engine's FIRST batched launch of >= 9 points (three
 * evaluation sets of each on the caller's own points, host theta -> host results; the faster stays; per handle; last-bit results
 * then depend on which kernel won).  gwi_batch_path answers with the static rule, or with the measured choice after that launch;
 * this returns whether a measurement has been made and the best microseconds per evaluation set of either kernel. */
gwi_status gwi_batch_calibration(gwi_handle h, int32_t* measured, double* mfma_us, double* taps_us);
/* A spline model whose kinds and basis counts have no ahead-of-time matrix-core instantiation gets one compiled at run time
 * (gwinferno_amd/csrc/gwi_jit.h) on its first batched launch of >= 9 points (at gwi_create with GWI_BATCH_MFMA=1): this says what
 * happened -- "compiled jit-mfma:6,207,107 ..." or why not (more than eight 16-basis gradient tiles, a term without a matrix-core
 * form, register spills, hipRTC missing); "" before the attempt and for models that have an ahead-of-time instantiation. */
const char* gwi_batch_kernel_note(gwi_handle h);

/* Per-sample log-weights log(p(theta|Lambda)/prior) (-inf for excluded samples), the arrays the
 * reference passes to hierarchical_likelihood as pe_weights / inj_weights (tests/inference_test.py:
 * 174-175).  Diagnostic / parity entry point; not used on the sampling path. */
gwi_status gwi_log_weights(gwi_handle h, const double* theta, double* pe_logw, double* inj_logw);

/* Multi-GPU (one process per GPU): each rank's engine holds a contiguous block of events and a
 * slice of the injections.  gwi_eval_partial() runs the scan and leaves this rank's partial
 * record (gwi_partial_len() doubles) in `record`; the caller exchanges records (RCCL all-gather
 * over xGMI) and every rank calls gwi_combine() on the gathered buffer. */
int64_t gwi_partial_len(gwi_handle h);
gwi_status gwi_eval_partial(gwi_handle h, const double* theta, double* record_host, double* log_bfs,
                            double* log_neffs, double* variances);
/* Host-only: recompute the sample-independent constants of `theta` that gwi_combine folds in
 * (gwi_eval_partial does this implicitly). */
gwi_status gwi_prepare_combine(gwi_handle h, const double* theta);
gwi_status gwi_combine(gwi_handle h, const double* records, int32_t n_ranks, const gwi_options* opt,
                       gwi_summary* summary, double* grad, double* norms);

/* In-engine collective (one process per GPU, RCCL over xGMI).  gwi_comm_unique_id() fills 128 bytes
 * on ONE rank (ncclGetUniqueId); the caller distributes them (any out-of-band channel) and every rank
 * calls gwi_comm_init().  `rccl_path` names the librccl to dlopen (NULL: "librccl.so.1").  Afterwards
 * gwi_eval_sharded() == gwi_eval_partial + ncclAllGather of the records on the engine's own stream +
 * gwi_combine, with no host round trip between the scan and the exchange; log_bfs / log_neffs /
 * variances are this rank's events. */
gwi_status gwi_comm_unique_id(const char* rccl_path, void* id128);
gwi_status gwi_comm_init(gwi_handle h, const char* rccl_path, const void* id128, int32_t rank, int32_t world);
gwi_status gwi_eval_sharded(gwi_handle h, const double* theta, const gwi_options* opt, gwi_summary* summary,
                            double* grad, double* log_bfs, double* log_neffs, double* variances, double* norms);

/* Single-node exchange without a collective launch: the ranks of ONE node publish their partial records into a POSIX
 * shared-memory segment (`name`, created by whichever rank gets there first; unlink it with gwi_shm_comm_unlink once every
 * rank has attached) and poll each other's sequence stamps -- the records are ~1 KiB and already end up in host memory, so
 * the exchange costs a few cache-line transfers between host cores instead of a collective's launch + small-message latency.
 * After gwi_shm_comm_init(), gwi_eval_sharded() evaluates this rank's shard through the engine's regular (AQL) fast path
 * and exchanges through the segment; it takes precedence over a communicator set up with gwi_comm_init().  Works on
 * host-only handles too (gwi_shm_exchange with caller-made records: the CPU test-suite).  Every rank must issue the same
 * sequence of exchanges.  Same partitioning contract as above (pipeline/analysis.py:78-86, :126-134). */
gwi_status gwi_shm_comm_init(gwi_handle h, const char* name, int32_t rank, int32_t world);
gwi_status gwi_shm_comm_unlink(const char* name);
/* publish `record` (gwi_partial_len() doubles) as this rank's, wait for every rank's, copy them to gathered[world][len] */
gwi_status gwi_shm_exchange(gwi_handle h, const double* record, double* gathered);

/* Diagnostic: the launch geometry gwi_create chose -- out = {PE tile size, injection tile size, tiles per event, injection
 * tiles, scan workgroups per hyper-parameter point, injection groups of the combine launch} (samples / counts). */
gwi_status gwi_launch_geometry(gwi_handle h, int32_t out[6]);

/* Timing of the most recent gwi_eval*: milliseconds between the start/stop HIP events attached to each
 * launch on the engine's stream ([0]=scan kernel, [1]=per-event combine, [2]=final reduce; 0 when the
 * host does the final sum). */
gwi_status gwi_last_kernel_ms(gwi_handle h, float ms[3]);
/* Per-launch kernel timing (off by default: it adds host overhead).  1: kernel begin/end of every launch of an
 * evaluation -- dispatch timestamps of the engine's AQL queue where that is active, HIP events attached to the launches
 * otherwise; 2: the same, forced through the HIP stream (A/B against the AQL path); 0: off. */
gwi_status gwi_set_timing(gwi_handle h, int32_t enabled);

/* Diagnostic: run `n_iter` sequential gwi_eval calls from C (no binding overhead) and return the
 * mean seconds per evaluation; separates host-language overhead from launch + device time. */
gwi_status gwi_selftime(gwi_handle h, const double* theta, const gwi_options* opt, int32_t n_iter, double* seconds_per_eval);

/* n sequential, blocking evaluations (value + gradient) at the given points thetas[n][n_theta] -- the inner loop of
 * a sampler (examples/utils.py:63-85 runs it inside one XLA program) with no host-language binding between two
 * evaluations; uses gwi_eval_sharded when gwi_comm_init has been called on the handle.  log_likelihoods[n];
 * grads[n][n_theta] nullable.  kernel_ms[n][3] (nullable): launch durations [scan, combine, final] of every
 * `timing_every`-th evaluation (event timing switched on for those only), -1 for the others. */
gwi_status gwi_eval_sequence(gwi_handle h, const double* thetas, int32_t n, const gwi_options* opt, double* log_likelihoods, double* grads, int32_t timing_every,
                             float* kernel_ms);

/* The same loop, recording the wall-clock seconds of every evaluation (host theta in -> host results out) into
 * seconds[n]: the latency distribution (median, p5/p95) of SURVEY.md section 8(d). */
gwi_status gwi_eval_latencies(gwi_handle h, const double* thetas, int32_t n, const gwi_options* opt, double* seconds);

/* Models with spline terms: the scan weighs a tile's samples against a reference exponent known before the tile's first
 * sample -- the tile's exact maximum at the previous evaluation of the handle (of the same point of a batch), applied as an
 * exact power of two so that results do not depend on it to the bit.  When the tile's true maximum turns out more than
 * 2^430 (2^215 in a squared-weight pass) away from it -- the first evaluation of a handle whose log-weights lie that far
 * from 0, or a jump in theta that moves a tile's weights by ~300 e-folds -- the evaluation is repeated once; the failed
 * attempt has left the exact maxima behind, so the repeat cannot miss.  This counts the repeated evaluations (a sampler
 * moves theta by a leapfrog step between two evaluations: 0 in any ordinary run, whatever the prior width). */
int64_t gwi_two_pass_repeats(gwi_handle h);

/* Host tuning: restrict the CALLING thread to the CPUs next to the engine's GPU (the local_cpulist of its PCI function,
 * intersected with the thread's current affinity).  Every evaluation is a few PCIe round trips driven by that thread.
 * GWI_ERR_UNSUPPORTED (and no change) when sysfs does not say. */
gwi_status gwi_pin_thread_to_engine(gwi_handle h);
/* the same by device index (GWI_DEVICE_CURRENT = the current HIP device), e.g. BEFORE engines are created, so that their
 * pinned host buffers are first touched on that side too */
gwi_status gwi_pin_thread_to_device(int32_t device);

/* Measured HBM bandwidth of the device, the number SURVEY.md section 8(d) asks to report next to the vendor figure the
 * roofline is normalised against: a read-only sweep (sum of one array: what the scan kernel's traffic looks like) and a
 * STREAM triad a = b + s c, each over arrays of n_doubles (>= 64 Mi doubles recommended: beyond the 256 MB Infinity Cache),
 * best of `iters` launches timed with HIP events.  GB/s = bytes moved / time (triad: 24 B per element).  No reference
 * counterpart (a measurement aid). */
gwi_status gwi_hbm_bandwidth(int32_t device, int64_t n_doubles, int32_t iters, double* read_gbs, double* triad_gbs);

/* How plain evaluations are dispatched: "aql: active" (AQL packets into a user-mode queue of the engine's own,
 * gwinferno_amd/csrc/gwi_aql.h: 0.4 us of host time per launch instead of 3.5) or the reason the HIP stream is used. */
const char* gwi_dispatch_info(gwi_handle h);

/* Name of the scan kernel this engine runs: the compiled term chain ("plq+plz+spline5", ...; gwi_kernel_variant_name) or
 * "generic (run-time term loop)" -- any product of <= GWI_MAX_TERMS terms has a kernel (the reference's model function
 * multiplies whatever densities the user picks: tests/inference_test.py:256-260, examples/simple_bspline_example.py:58-71);
 * products outside the ahead-of-time set get a chain compiled at gwi_create ("jit:...", below); the generic kernel (2-2.7 x the
 * scan time) runs only where that is impossible. */
const char* gwi_scan_kernel_name(gwi_handle h);

/* ---- scan chains compiled at run time (gwinferno_amd/csrc/gwi_jit.h) ---------------------------------------------------
 * The reference's user model multiplies whatever densities the user picks (tests/inference_test.py:256-260,
 * models/bsplines/separable.py:295-778).  A product of terms whose kind sequence has no ahead-of-time scan kernel gets one
 * at gwi_create: the scan template instantiated for exactly that sequence by hipRTC (gfx950, the flags of the library's
 * own build, from the headers embedded in the library), kept as a code object under $GWI_JIT_CACHE (default
 * ~/.cache/gwinferno_amd).  Without hipRTC (or with GWI_JIT=0) such models run the generic kernel.  The cache is trusted only as
 * far as it is the caller's: a directory that is a link, belongs to another user or is writable by group / others is skipped
 * (next candidate, else compile in this process only, said once on stderr); a cache file is a regular 0600 file of this user
 * carrying a digest of (kinds, samples per lane, kernel names, code object) and is compiled over, never loaded, when it does not
 * match (tests/test_jit_cache_cpu.py).
 *
 * gwi_jit_compile(): compile (or find in the cache) the chain of `kinds` (GWI_TERM_* numbers, ascending) with
 * `samples_per_lane` (1 | 2) samples per lane -- needs no GPU (samples_per_lane = 0: the batched matrix-core kernel of a spline
 * model instead, `kinds` then being kind + 100 x 16-basis gradient tiles of each term, as gwi_batch_kernel_note names it).  path_out (nullable, path_cap bytes) receives the cache file
 * ("" when no cache directory is both writable and trusted), or the reason on failure; compile_seconds = hipRTC time spent by THIS call chain
 * (0 when the code object came from the cache), from_cache = 1 then.  GWI_ERR_UNSUPPORTED: hipRTC missing / compilation failed.
 * gwi_jit_info(): whether this engine's scan kernel was compiled at run time, what that cost this process and whether the
 * disk cache supplied it; note = why the generic kernel runs where it does ("" otherwise).  All outputs nullable. */
gwi_status gwi_jit_compile(const int32_t* kinds, int32_t n_kinds, int32_t samples_per_lane, char* path_out, int64_t path_cap, double* compile_seconds,
                           int32_t* from_cache);
gwi_status gwi_jit_info(gwi_handle h, int32_t* compiled_at_run_time, double* compile_seconds, int32_t* from_cache, const char** note);

const char* gwi_last_error(gwi_handle h);
void gwi_destroy(gwi_handle h);

/* Library-level queries usable without a GPU. */
int32_t gwi_abi_version(void);
int32_t gwi_kernel_variants(void);         /* number of compiled term sequences */
const char* gwi_kernel_variant_name(int32_t i);

#ifdef __cplusplus
}
#endif
#endif /* GWI_ENGINE_H */
