/* stan4bart_amd.h — C-ABI of the MI355X-native stan4bart Gibbs hot path.
 *
 * This is the drop-in boundary for the reference's `.Call` layer (reference
 * src/init.cpp:1215-1229 registration table).  Every entry point below names the reference
 * routine it replaces.  The reference passes R SEXPs (S4 dbarts objects + named lists); here the
 * same information travels as plain structs of scalars and pointers — no R, no torch types.
 * An R shim (`INTEGRATION.md`) unpacks the SEXPs into these structs and forwards.
 *
 * Conventions
 *   - all matrices are column-major (R layout); all pointers are host pointers unless the
 *     struct says otherwise; inputs are copied at create time (reference copies X, y, CSR parts:
 *     src/stan_sampler.cpp:35-65,197-249) except `offset`, which is copied too (the reference
 *     borrows it, src/init.cpp:1033 — copying removes the lifetime hazard).
 *   - every function returns 0 on success, non-zero on failure; `s4b_last_error()` returns the
 *     message the R shim would hand to Rf_error (reference: Rf_error longjmp, src/init.cpp:319).
 *   - one sampler == one chain == one GPU (reference: one chain per R worker process,
 *     R/stan4bart_fit.R:527-533).  A sampler is not thread-safe; different samplers are independent.
 *
 * The symbol prefix is configurable so the CPU oracle (oracle/, test infrastructure only) can
 * export the identical interface as `orc_*` for parity tests.
 */
#ifndef STAN4BART_AMD_H
#define STAN4BART_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifndef S4B_PREFIX
#define S4B_PREFIX s4b_
#endif
#define S4B_CAT_(a, b) a##b
#define S4B_CAT(a, b) S4B_CAT_(a, b)
#define S4B_FN(name) S4B_CAT(S4B_PREFIX, name)

#ifdef __cplusplus
extern "C" {
#endif

/* revision of the struct layouts below; s4b_bart_control.interface_version must carry it (checked by create()) */
#define S4B_INTERFACE_VERSION 6

typedef struct s4b_sampler s4b_sampler; /* reference: `Sampler`, src/init.cpp:124-173, held in an externalptr */

/* dbartsControl + dbartsModel fields the reference sets (R/stan4bart_fit.R:436-479) */
typedef struct {
  int32_t n_trees;      /* control@n.trees (bart_args$n.trees)                       */
  int32_t n_thin;       /* control@n.thin = skip.bart (R/stan4bart_fit.R:438)         */
  int32_t keep_trees;   /* control@keepTrees                                          */
  int32_t node_capacity;/* 0 = default (256): max node slots per tree on the device   */
  double base, power;   /* cgm(power = 2, base = 0.95)                                */
  double k;             /* normal(k = 2): the fixed k, or the value a modeled k starts from */
  double node_scale;    /* model@node.scale: 0.5 continuous, 3.0 binary (:477-479)    */
  double birth_or_death_prob, swap_prob, change_prob, birth_prob; /* dbarts: .5 .1 .4 .5 */
  const double* split_probs; /* cgm(split.probs = ): NULL (predictors equally likely) or one positive weight per predictor, any scale:
                              * a rule's predictor is drawn with probability weight / sum of the weights of the predictors still
                              * available at the node, and the tree prior carries the same term (R/stan4bart_fit.R:466-475;
                              * tests/testthat/test-09-bartArgs.R:20).  Tree updates then run on the persistent sweep's k_sweep_sp
                              * where that applies, else on the two-kernel path.                                                */
  int32_t use_quantiles;     /* dbartsControl(useQuantiles = ) through bart_args (R/stan4bart_fit.R:440-444): 0 uniform cut points
                              * between the column extremes, 1 cut points from the distinct values (n_cuts is then a maximum)    */
  int32_t interface_version; /* must be S4B_INTERFACE_VERSION: create() refuses anything else.  This struct and s4b_results have grown fields over the revisions
                              * (k_hyper_*, bart_k) and carry no size field: a caller compiled against an older header would pass short structs, and
                              * run() would read past their end.  The word was `reserved` (0) in those headers, so such a caller is refused at
                              * create() — before any of the newer fields is read — instead of being served garbage. */
  /* normal(k = chi(degreesOfFreedom, scale)) — reference R/stan4bart.R:202 ("allow calls like bart_args = list(k = chi(2, Inf))"),
   * R/stan4bart_fit.R:460-465, tests/testthat/test-09-bartArgs.R:32; model@node.hyperprior, `kPrior->isFixed` src/init.cpp:272,731.
   * k_hyper_df <= 0: k is fixed (`k` above).  k_hyper_df > 0: k is a parameter with prior density k^(df - 1) exp(-k^2 / (2 scale^2))
   * (k_hyper_scale may be +Inf) that is redrawn once per sweep, after the trees (and the latents of a binary response), from its
   * conditional given the leaf values (k^2 ~ Gamma((df + #leaves) / 2, rate (T sum mu^2 / node_scale^2 + 1 / scale^2) / 2), one rgamma from
   * R's stream); its draws come back in s4b_results.bart_k. */
  double k_hyper_df, k_hyper_scale;
} s4b_bart_control;

/* dbartsData slots (R/lme4_functions.R:176, R/stan4bart_fit.R:449-451) */
typedef struct {
  int64_t n;            /* numObservations                                            */
  int32_t p;            /* numPredictors                                              */
  int32_t reserved;
  const double* x;      /* n x p, data@x                                              */
  const int32_t* n_cuts;/* p, data@n.cuts (default 100 each)                          */
  int64_t n_test;       /* numTestObservations                                        */
  const double* x_test; /* n_test x p or NULL, data@x.test                            */
} s4b_bart_data;

/* the `stanData` list: same names as dataNames[] (reference src/stan_sampler.cpp:67-80) */
typedef struct {
  int64_t N; int32_t K;
  int32_t is_binary, has_intercept, has_weights;
  int32_t prior_dist, prior_dist_for_aux;           /* 0 none,1 normal,2 student_t,3 hs,4 hs_plus,5 laplace,6 lasso,7 product_normal / aux: 0..2, 3 exponential */
  const double* X;            /* N x K, column-centred fixed-effect design (R/rstanarm_functions.R:420-446) */
  const double* y;            /* N */
  const double* weights;      /* N or NULL */
  const double* prior_scale;  /* K */
  const double* prior_mean;   /* K */
  const double* prior_df;     /* K */
  double prior_scale_for_aux, prior_mean_for_aux, prior_df_for_aux;
  int32_t t;                  /* number of grouping terms */
  int32_t q;                  /* columns of Z */
  int32_t len_theta_L, len_concentration, len_regularization, reserved;
  const int32_t* p;           /* t */
  const int32_t* l;           /* t */
  const double* shape;        /* t */
  const double* scale;        /* t */
  const double* concentration;   /* len_concentration */
  const double* regularization;  /* len_regularization */
  int64_t num_non_zero;
  const double* w;            /* num_non_zero: CSR values of Z  */
  const int32_t* v;           /* num_non_zero: 0-based columns  */
  const int32_t* u;           /* N + 1: 0-based row starts      */
  /* coefficient prior families beyond normal / student_t (continuous.stan:124-144, 298-322, 382-414):
   * prior_dist 3 hs, 4 hs_plus, 5 laplace, 6 lasso, 7 product_normal */
  double global_prior_df, global_prior_scale, slab_df, slab_scale;   /* hs, hs_plus */
  const int32_t* num_normals; /* K (>= 2 each), product_normal only, else NULL */
} s4b_stan_data;

/* the `stanControl` list with the reference defaults (src/stan_sampler.cpp:395-458) */
typedef struct {
  uint32_t seed;              /* required */
  int32_t skip;               /* <=0 : NA -> max(1,(2000 - warmup)/1000) (src/init.cpp:206-209) */
  double init_r;              /* 2.0 */
  double adapt_gamma, adapt_delta, adapt_kappa, adapt_t0;   /* .05 .8 .75 10 */
  uint32_t adapt_init_buffer, adapt_term_buffer, adapt_window, reserved;  /* 75 50 25 */
  double stepsize, stepsize_jitter;  /* 1, 0 */
  int32_t max_treedepth;      /* 10 */
  int32_t hmc_mode;           /* extension: 0 = sufficient-statistic (Gram) gradient, 1 = per-leapfrog O(N) kernels */
} s4b_stan_control;

/* per-iteration callback (src/init.cpp:849-911); a non-zero return stops the run after that iteration (status 1, "stopped by the
 * per-iteration callback") — the reference's callback is R code whose error unwinds run() */
typedef int (*s4b_callback_fn)(void* user, const double* yhat_train, const double* yhat_test,
                               const double* stan_pars, int32_t num_pars);
/* progress / cancellation hook of run(): called before iteration `iter` (1-based) of `num_iter` for iter = 1 and whenever iter is
 * a multiple of common_control.refresh (every iteration when refresh <= 0); a non-zero return stops the run (status 1,
 * "interrupted ...").  With a hook installed run() prints nothing itself: the hook's owner prints the reference's
 * "starting warmup ..." and "iter k / n" lines (the R shim does, shim/init_shim.cpp).
 * Reference: "iter k / n" lines (src/init.cpp:752-754) and R_CheckUserInterrupt per transition (src/stan_sampler.hpp:44-48). */
typedef int (*s4b_progress_fn)(void* user, int32_t iter, int32_t num_iter, int32_t is_warmup);

/* the `commonControl` list (src/init.cpp:199-202, 1015-1051) */
typedef struct {
  int32_t warmup, iter, verbose, refresh;
  int32_t is_binary;
  int32_t offset_type;        /* 0 default,1 fixef,2 ranef,3 bart,4 parametric (src/init.cpp:83-97) */
  int32_t keep_fits;
  int32_t device;             /* extension: HIP device ordinal for this chain */
  const double* offset;       /* N or NULL: user offset */
  const double* bart_offset_init; /* N or NULL */
  double sigma_init;          /* > 0, default 1 */
  s4b_callback_fn callback;   /* NULL or per-iteration callback (src/init.cpp:849-911) */
  void* callback_user;
} s4b_common_control;

/* caller-allocated result buffers of one run() (layouts: src/stan_sampler.cpp:577-596, src/bart_util.cpp:13-81).
 * num_samples = keep_fits ? num_iter : 1; any pointer may be NULL to skip that output. */
typedef struct {
  double* stan;          /* num_pars x num_samples */
  double* bart_sigma;    /* num_samples            */
  double* bart_train;    /* n x num_samples        */
  double* bart_test;     /* n_test x num_samples   */
  int32_t* bart_varcount;/* p x num_samples        */
  double* bart_k;        /* num_samples: the draws of a modeled k (k_hyper_df > 0; reference result element "k", src/bart_util.cpp:17-26,60-64,75-76);
                          * with a fixed k the buffer, if given, is filled with that value */
} s4b_results;

/* R's generator state as 625 words {mti, mt[0..623]} == .Random.seed[2:626]
 * (reference brackets every entry with GetRNGstate/PutRNGstate: src/init.cpp:259,298,750,919) */
#define S4B_R_RNG_WORDS 625

const char* S4B_FN(last_error)(void);

/* stan4bart_create(bartControl, bartData, bartModel, stanData, stanControl, commonControl) — src/init.cpp:190-310 */
int S4B_FN(create)(const s4b_bart_control* bart_control, const s4b_bart_data* bart_data,
                   const s4b_stan_data* stan_data, const s4b_stan_control* stan_control,
                   const s4b_common_control* common_control, const uint32_t* r_rng_state,
                   s4b_sampler** out);

/* stan4bart_run(sampler, numIter, isWarmup, resultsType) — src/init.cpp:678-965; results_type 0 both,1 bart,2 stan */
int S4B_FN(run)(s4b_sampler* s, int32_t num_iter, int32_t is_warmup, int32_t results_type, s4b_results* out);

/* stan4bart_disengageAdaptation — src/init.cpp:995-1004 */
int S4B_FN(disengage_adaptation)(s4b_sampler* s);

/* stan4bart_printInitialSummary — src/init.cpp:971-993 (writes to stdout) */
int S4B_FN(print_initial_summary)(s4b_sampler* s);

/* stan4bart_getParametricMean — src/init.cpp:332-347: X beta + Z b of the last Stan draw, N doubles */
int S4B_FN(get_parametric_mean)(s4b_sampler* s, double* out);

/* stan4bart_getBARTDataRange — src/init.cpp:316-330: {min, max} of the response rescaling */
int S4B_FN(get_bart_data_range)(s4b_sampler* s, double out[2]);

/* PutRNGstate()/GetRNGstate() counterparts */
int S4B_FN(get_r_rng_state)(s4b_sampler* s, uint32_t* state);
int S4B_FN(set_r_rng_state)(s4b_sampler* s, const uint32_t* state);

/* sizes: num_pars (rows of the stan result), n, n_test, p, n_trees */
int S4B_FN(get_dims)(s4b_sampler* s, int64_t dims[5]);

/* row names of the stan result (src/stan_sampler.cpp:476-489): writes a '\n'-joined list */
int S4B_FN(get_stan_par_names)(s4b_sampler* s, char* buf, size_t cap);

/* stan4bart_getTrees(current = TRUE) — src/init.cpp:514-671 flattened-tree layout for the live trees:
 * preorder per tree; var >= 0: internal node (value = cut point), var = -1: leaf (value = mu on the
 * rescaled scale).  Returns the node count through *num_nodes; arrays may be NULL to query the size. */
int S4B_FN(get_trees)(s4b_sampler* s, int64_t cap, int32_t* tree, int32_t* n_obs, int32_t* var,
                      int32_t* split, double* value, int64_t* num_nodes);

/* stan4bart_getTrees(current = FALSE) — src/init.cpp:514-671 over the draws kept while sampling (keep_trees; also on a
 * stored sampler): sample < 0 selects every kept draw, otherwise that draw (0-based).  Same layout as get_trees plus
 * the draw index of every node. */
int S4B_FN(get_kept_trees)(s4b_sampler* s, int64_t sample, int64_t cap, int32_t* sample_index, int32_t* tree, int32_t* n_obs,
                           int32_t* var, int32_t* split, double* value, int64_t* num_nodes);

/* stan4bart_getTrees(chainIndices, sampleIndices, treeIndices, current = FALSE) — src/init.cpp:514-671 with its index vectors:
 * 0-based draw / tree indices (NULL = all).  A sampler is one chain; the R shim loops over chainIndices. */
int S4B_FN(get_kept_trees_indexed)(s4b_sampler* s, const int32_t* sample_idx, int64_t num_samples, const int32_t* tree_idx, int64_t num_trees,
                                   int64_t cap, int32_t* sample_index, int32_t* tree, int32_t* n_obs, int32_t* var, int32_t* split, double* value,
                                   int64_t* num_nodes);

/* stan4bart_printTrees(chainIndices, sampleIndices, treeIndices) — src/init.cpp:448-512: prints the selected kept trees to stdout.
 * The reference forwards to dbarts' printer, whose text format is not part of the reference tree: the layout here is this
 * library's own (one node per line, indented by depth). */
int S4B_FN(print_trees)(s4b_sampler* s, const int32_t* sample_idx, int64_t num_samples, const int32_t* tree_idx, int64_t num_trees);

/* progress / cancellation hook (see s4b_progress_fn); fn = NULL removes it */
int S4B_FN(set_progress)(s4b_sampler* s, s4b_progress_fn fn, void* user);

/* stan4bart_exportBARTState — src/init.cpp:409-416 (+ R/stan4bart_fit.R:572-580): the trees kept while sampling (keep_trees),
 * the cut points and the response scales as one relocatable byte string, so that a chain fitted in another process can be
 * predicted from.  Call with buf = NULL (or cap too small) to learn the size. */
int S4B_FN(export_bart_state)(s4b_sampler* s, void* buf, int64_t cap, int64_t* size);

/* stan4bart_createStoredBARTSampler — src/init.cpp:418-446: a sampler that only holds an exported state; it supports
 * predict_bart, export_bart_state, get_dims and free, everything else fails with a message. */
int S4B_FN(create_stored_bart_sampler)(const void* state, int64_t size, int32_t device, s4b_sampler** out);

/* stan4bart_predictBART(storedSampler, x_test, offset_test = NULL) — src/init.cpp:354-403, with the rescaling of
 * R/generics.R:671-674 folded in: BART fit of every kept draw at new predictor rows, on the data scale (probit: the
 * latent scale).  Trees are kept for the non-warmup runs of a sampler created with bart_control.keep_trees = 1
 * (reference: keepTrees is switched on only for the sampling phase, src/init.cpp:216-221,737-744).
 * out is n_test x num_samples (column-major); pass out = NULL to query num_samples. */
int S4B_FN(predict_bart)(s4b_sampler* s, const double* x_test, int64_t n_test, double* out, int64_t* num_samples);
/* the same with the reference's third argument: offset_test (n_test doubles, or NULL) is added to every draw's prediction */
int S4B_FN(predict_bart_offset)(s4b_sampler* s, const double* x_test, int64_t n_test, const double* offset_test, double* out, int64_t* num_samples);

/* The state of a chain BETWEEN TWO GIBBS ITERATIONS as one relocatable byte string: what the next iteration starts from.  It is
 * the hook of the teacher-forced parity tests (state of one implementation injected into the other before every compared
 * transition) and lets a chain continue in another sampler created from the same data; it is not an archive of a fit: the
 * kept trees of keep_trees (export_bart_state carries those), the draws already returned, the tree-move trace and the
 * counters of get_counters / get_nuts_stats stay with the sampler that produced them.  set_state validates the blob
 * (dimensions, tree shapes, rule ranges, finite values, positive scales) before anything is changed.  No reference routine
 * does this: the reference can only re-run a chain from its seed (R/stan4bart_fit.R:33-60); what the blob holds is the
 * state the reference keeps between two iterations of src/init.cpp:752-917 —
 *   NUTS: current point, step size, dual-averaging scalars (stepsize_adaptation.hpp:10-65), window counters
 *   (windowed_adaptation.hpp:29-110), Welford accumulators and inverse metric (var_adaptation.hpp:17-46), ecuyer1988 state;
 *   BART: trees + leaf values, total fit per observation, offset, response scale, sigma, probit latents, R's generator.
 * Layout (native endianness, every block 8-byte aligned), in this order:
 *   s4b_state_header
 *   double  q[D], inv_metric[D], welford_mean[D], welford_m2[D]
 *   double  nuts[6]     = {stepsize, da_mu, da_counter, da_s_bar, da_x_bar, welford_n}
 *   double  last_row[7] = lp__, accept_stat__, stepsize__, treedepth__, n_leapfrog__, divergent__, energy__ of the last draw
 *   uint32  win[8]      = {num_warmup, init_buffer, term_buffer, base_window, window_counter, next_window, window_size, adapting}
 *   uint32  ecuyer[2]
 *   uint32  r_rng[626]  = {mti, mt[624], 0}
 *   double  scale[4]    = {min, max, range, sigma on the data scale}
 *   double  offset[n], total_fits[n] (sum of the tree fits on the rescaled scale), then latents[n] (probit only: latent
 *           response with the offset removed)
 *   per tree: int32 num_nodes, num_leaves; int32 node[num_nodes][2] in preorder ({var, split} internal, {-1, n_obs} leaf);
 *             double mu[num_leaves] in DFS order
 * get_state: call with buf = NULL (or cap too small) to learn the size. */
#define S4B_STATE_MAGIC 0x53423453u /* "S4BS" */
typedef struct {
  uint32_t magic, version;   /* S4B_STATE_MAGIC, 1 */
  int64_t n;
  int32_t n_trees, num_unconstrained, is_binary, p;
  int64_t reserved[2];       /* [0]: bit pattern of the current k (a double) when k is modeled, else 0 */
} s4b_state_header;
int S4B_FN(get_state)(s4b_sampler* s, void* buf, int64_t cap, int64_t* size);
int S4B_FN(set_state)(s4b_sampler* s, const void* buf, int64_t size);

/* diagnostics used by the parity tests: per-tree-update trace records of 5 int32
 * {type 0 birth 1 death 2 swap 3 change, status 1/0/-1, var, split, num_leaves} */
int S4B_FN(set_trace)(s4b_sampler* s, int32_t enable);
int S4B_FN(get_trace)(s4b_sampler* s, int64_t cap_records, int32_t* out, int64_t* num_records);
/* DFS leaf rank of every training observation in tree t (n int32) */
int S4B_FN(get_leaf_assignment)(s4b_sampler* s, int32_t tree, int32_t* out);
/* Not a reference routine.  Hint that `chains` samplers share this sampler's device (R/stan4bart_fit.R:515-533 runs the chains of
 * one fit in parallel workers; here they can be host threads on one GPU).  Where the persistent sweep applies (set_tree_path) the hint
 * changes nothing since round 5: samplers of one process take turns on the device, launches of other processes are sorted out by the
 * roll call at the start of every persistent launch, and the aggregate rate is higher than on the per-tree kernels.
 * Elsewhere — n > 1.04e6, observation weights together with cgm(split.probs) — the fused launch keeps every CU busy with one register-heavy workgroup,
 * which is fastest for a chain that has the device to itself; with three or more chains per device the sampler switches to the
 * two-kernel tree update, which leaves room for the other chains' kernels (higher aggregate rate).  The
 * same chain either way: identical tree moves and generator stream, floating-point values equal up to the summation order of the
 * per-bin sums (1e-15 relative).  May be called at any time between runs. */
int S4B_FN(set_device_sharing)(s4b_sampler* s, int32_t chains);
/* Not a reference routine.  Which device code runs a tree update: 0 automatic (default), 1 two kernels per tree (k_tree + k_control),
 * 2 one fused launch per tree (k_step), 4 persistent (k_sweep: ONE launch per sweep, the residual in the registers of the pass waves, bin
 * partials exchanged through order-free integer atomics), 5 persistent with a streaming pass (k_sweep_stream: the same launch, the pass
 * waves read and write the residual per tree; measured slower than 1 / 2 at every size and never chosen automatically).  The automatic
 * choice is 4 wherever it applies — at most 16 observations per pass thread (n <= 1 044 480 on 256 compute units); observation weights
 * and cgm(split.probs) have their own instantiations of the launch since round 6 (k_sweep_w, k_sweep_sp), the two together do not —,
 * else 2 up to n ~ 4e6, else 1 (with split.probs always 1).  A request the sampler cannot honour
 * (more than 255 quads per thread for 2, the conditions above for 4 / 5) falls back to the next path down; get_tree_path reports the path
 * in effect.  The same chain on every path (see set_device_sharing).  May be called at any time between runs.
 * get_tree_path: out[0] = the request, out[1] = the path in effect (1, 2, 4 or 5).  (3 was the lagged launch of an earlier revision:
 * removed, the value is rejected.) */
int S4B_FN(set_tree_path)(s4b_sampler* s, int32_t path);
int S4B_FN(get_tree_path)(s4b_sampler* s, int32_t out[2]);
/* counters: {log-density gradient evaluations, tree updates, device kernel launches} */
int S4B_FN(get_counters)(s4b_sampler* s, int64_t out[3]);
/* diagnostics of the one-launch O(N) sums of the Stan block (k_stan_fused): out = {evaluations, evaluations repeated in plain
 * doubles because the fixed-point range check failed (first evaluation, rescaled response, trajectory far outside the typical set)} */
int S4B_FN(get_fused_stats)(s4b_sampler* s, int64_t out[2]);
/* persistent tree path only (zeros otherwise): out = {sweeps run while the persistent path was in effect — the ones a busy device sent to
 * k_step launches (get_sweep_busy and the back-off after it) included —, of which handed over to k_step launches part-way because a
 * tree outgrew the 64 node slots of the wave-register control path} since creation */
int S4B_FN(get_sweep_stats)(s4b_sampler* s, int64_t out[2]);
/* persistent tree path only (zeros otherwise), since creation: out = {persistent launches that ran (roll call passed), tree updates
 * decided inside them, steps whose bin statistics were published BEFORE the verdict of the tree before (speculation), steps
 * of those whose verdict bore the speculation out} */
int S4B_FN(get_sweep_spec)(s4b_sampler* s, int64_t out[4]);
/* persistent tree path only (zero otherwise): persistent launches that found the device shared (not every workgroup of the launch became
 * resident within 200 us: somebody else's kernels held compute units).  Such a launch changes nothing; its sweep ran as k_step launches and
 * so do the next 16, 32, ... sweeps before the persistent launch is tried again.  (The reference's chain fan-out starts one worker process
 * per chain, R/stan4bart_fit.R:498-533: on a box with fewer GPUs than chains those processes share devices.) */
int S4B_FN(get_sweep_busy)(s4b_sampler* s, int64_t* out);
/* extension: how the Stan block evaluates the O(N) part of the log density (stan_control.hmc_mode): 0 = sufficient statistics
 * gathered once per Gibbs iteration, 1 = one device evaluation per leapfrog (the reference's cost model).  A sampler created
 * with mode 0 may be switched to 1 and back between runs (same posterior, same draws up to rounding); one created with mode 1
 * has no Gram matrix and stays in mode 1. */
int S4B_FN(set_hmc_mode)(s4b_sampler* s, int32_t mode);
int S4B_FN(get_hmc_mode)(s4b_sampler* s, int32_t* mode);

/* NUTS totals over all transitions since creation: {transitions, sum of treedepth__, sum of n_leapfrog__, divergent transitions}
 * (the per-draw values are columns 4-6 of the stan result; the totals let a caller that keeps no per-iteration output, keep_fits =
 * FALSE, report mean tree depth and leapfrogs per iteration) */
int S4B_FN(get_nuts_stats)(s4b_sampler* s, double out[4]);

/* measurement hook (no reference counterpart): runs `n_sweeps` extra BART sweeps with HIP events recorded on the
 * sampler's own stream around every kernel launch and returns, per kernel class
 * {stats, control, apply}: out[0..2] = average launch duration in microseconds, out[3..5] = launches timed,
 * out[6] = wall microseconds per sweep (events around the whole sweep, no per-launch events), out[7] = n. */
int S4B_FN(profile_sweep)(s4b_sampler* s, int32_t n_sweeps, double out[8]);
/* extension (measurement): plain streaming kernels over n_doubles doubles on `device` — out[0] GB/s of a read-only pass,
 * out[1] GB/s of an in-place read + write pass (the residual's access pattern), out[2..3] their best times in us */
int S4B_FN(stream_probe)(int32_t device, int64_t n_doubles, int32_t reps, double out[4]);

/* extension (measurement): HIP-event timing of the per-leapfrog O(N) sums of the hmc_mode 1 path at the current draw.
 * out[0] us per evaluation (kernels), out[1] us including the result fetch, out[2] launches per evaluation, out[3] N,
 * out[4] algorithmic bytes per evaluation N (8K + 12z + 20) (SURVEY §8d B_lf) */
int S4B_FN(profile_leapfrog)(s4b_sampler* s, int32_t n_evals, double out[8]);

/* TEST HOOK (no reference counterpart; off by default): hook 1, value k > 0 — every k-th persistent launch of this sampler finds the roll call
 * of its launch already decided "device busy" (what a launch sees that shares the device with another process's kernels), so that the busy
 * fallback of the persistent path runs under the parity tests without a second process; value 0 switches it off.  Other hooks are rejected. */
int S4B_FN(set_test_hook)(s4b_sampler* s, int32_t hook, int64_t value);

/* finalizer of the externalptr — src/init.cpp:1152-1165 */
void S4B_FN(free)(s4b_sampler* s);

#ifdef __cplusplus
}
#endif
#endif /* STAN4BART_AMD_H */
