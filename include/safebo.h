/*
 * safebo.h -- C ABI of libsafebo.so: the MI355X (gfx950) candidate-sweep engine for
 * SafeOpt / GoOSE safe Bayesian optimisation.
 *
 * The reference project (dleeim/Safe-Bayesian-Optimization) is pure Python/JAX and has no
 * FFI seam of its own; the seam this library sits behind is the one SURVEY.md section 8(b) names:
 *
 *   state crossing the seam  = the `inference_datasets` dict      models/GP_Safe.py:16-23, 236-245
 *                              + `bound`, `b`                     models/SafeOpt.py:12-13
 *   calls replaced           = GP.GP_inference (batched)          models/GP_Safe.py:310-352
 *                              BO.mean / ucb / lcb (batched)      models/SafeOpt.py:29-45, test/test_SafeOpt.py:337
 *                              BO.Minimizer / BO.Expander         models/SafeOpt.py:53-66, 90-124
 *                              BO.minimize_obj_lcb / Target /
 *                              explore_safeset                    models/GoOSE.py:63-67, 80-119
 *
 * Conventions
 *   - every entry point returns an int status (SBO_OK == 0, errors < 0) and never throws;
 *     sbo_last_error() returns a thread-local message for the last failing call.
 *   - host pointers are borrowed for the duration of the call only (the library copies);
 *     arrays are C-contiguous, row-major.  Model arrays are always passed as double; the
 *     `dtype` tag selects the arithmetic the kernels run in (the arrays are rounded once on upload).
 *   - one sbo_ctx per process and per GPU (one process per GPU, ranks joined with sbo_comm_init);
 *     a ctx is single-caller; calls return after the device work they issued has completed unless
 *     the entry point says "asynchronous".
 *   - flat candidate index g: for grids, axis 0 is the fastest axis
 *     (jnp.meshgrid 'xy' + ravel, test/test_SafeOpt.py:325-334); indices reported in results are
 *     GLOBAL flat indices (shard offset included).
 *   - there is no CPU fallback: without a HIP device sbo_init fails with SBO_E_HIP.
 */
#ifndef SAFEBO_H
#define SAFEBO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBO_ABI_VERSION 3
#define SBO_MAX_D 8        /* input dimension limit (reference problems use d = 2)          */
#define SBO_MAX_Q 8        /* modelled outputs: objective + constraints                      */
#define SBO_MAX_N 2048     /* observations                                                   */

enum sbo_status {
  SBO_OK = 0,
  SBO_E_INVALID = -1,         /* bad argument (maps to ValueError, as models/GP_Safe.py:134-137) */
  SBO_E_NOMEM = -2,
  SBO_E_HIP = -3,             /* HIP runtime error / no device                                   */
  SBO_E_NO_MODEL = -4,        /* sweep before sbo_model_set                                      */
  SBO_E_NO_CANDIDATES = -5,   /* sweep before sbo_candidates_*                                   */
  SBO_E_EMPTY_SAFE_SET = -6,  /* S_t is empty on this candidate set (all ranks)                  */
  SBO_E_COMM = -7,            /* RCCL error                                                      */
  SBO_E_UNSUPPORTED = -8
};

enum sbo_dtype { SBO_F64 = 0, SBO_F32 = 1 };

/* which matrix the variance contraction uses */
enum sbo_factor {
  SBO_FACTOR_INVK = 0,  /* caller's invK (models/GP_Safe.py:231-232): alpha = invK (y - mp) uses it as given, the
                           variance uses its triangular factor M (M^T M = invK):  k^T invK k = ||M k||^2        */
  SBO_FACTOR_CHOL = 1   /* library builds K + (sn2 + float32 eps) I = L L^T itself and contracts with M = L^-1:
                           var = sf2 - || L^-1 k ||^2  (used when invK == NULL)                            */
};

enum sbo_bound_kind { SBO_MEAN = 0, SBO_UCB = 1, SBO_LCB = 2, SBO_VAR = 3 };

enum sbo_mask {
  SBO_MASK_S = 0,   /* safe set                 models/SafeOpt.py:57-59                                   */
  SBO_MASK_U = 1,   /* "fully unsafe" witnesses models/SafeOpt.py:73-77, 109                              */
  SBO_MASK_M = 2,   /* potential minimisers     models/SafeOpt.py:62                                      */
  SBO_MASK_G = 3,   /* expanders G_c  (index c = 1..q-1)  models/SafeOpt.py:85-88, 111                    */
  SBO_MASK_O = 4    /* GoOSE optimistic set O_c (c = 1..q-1)  models/GoOSE.py:93-101                      */
};

typedef struct sbo_ctx sbo_ctx;

typedef struct sbo_sweep_opts {
  double b;                        /* confidence multiplier beta, models/SafeOpt.py:13                    */
  int32_t reference_quirk_L_index; /* 1: every constraint uses L_{q-1} (models/SafeOpt.py:110 loop leak);
                                      0: constraint c uses L_c                                            */
  int32_t want_masks;              /* 1: keep S/U/M/G (or O) masks in HBM for sbo_masks_get               */
  int32_t posterior_ready;         /* 1: reuse mean/var of the last sbo_posterior_run on these candidates */
  int32_t lean;                    /* the caller wants the sweep's result only -- sets, indices, counts, u*, the constraints' L.  1: the
                                      sweep may leave mean / var unwritten where no later stage of it reads them (the objective on posterior
                                      tiles without a safe candidate: u*, M and the arg-max reductions are over S only,
                                      models/SafeOpt.py:47-66); 2: it need not even evaluate them there (the result is the same).  A lean
                                      SafeOpt sweep of a model with constraints reports L[0] = 0: the objective's Lipschitz key is read by
                                      no sweep of the reference (its expanders use the constraints' keys, models/SafeOpt.py:110,
                                      models/GoOSE.py:100), and the interpolating posteriors then leave its gradient fields out.
                                      sbo_posterior_get / sbo_bounds / a posterior_ready sweep behind a lean sweep run K1 again.  0 (the
                                      default): the whole posterior is evaluated and stays resident, as models/GP_Safe.py:310-352 returns it */
} sbo_sweep_opts;

typedef struct sbo_safeopt_result {
  /* Minimizer(): argmax_{M_t} var_0, returns (x, sqrt(var_0))           models/SafeOpt.py:53-66          */
  int64_t minimizer_index;
  double  minimizer_x[SBO_MAX_D];
  double  minimizer_std;
  /* Expander(): per constraint argmax_{G_c} var_0, most uncertain kept   models/SafeOpt.py:90-124         */
  int64_t expander_index_c[SBO_MAX_Q];   /* [c-1], -1 when G_c is empty                                   */
  double  expander_std_c[SBO_MAX_Q];
  int32_t expander_best_c;               /* constraint index of the kept expander, 0 when none            */
  int64_t expander_index;
  double  expander_x[SBO_MAX_D];
  double  expander_std;
  int32_t choose_minimizer;              /* std_min > std_exp, test/test_SafeOpt.py:153                   */
  double  u_star;                        /* min_{S_t} ucb_0, models/SafeOpt.py:47-51, 61                  */
  double  L[SBO_MAX_Q];                  /* max over candidates of ||grad MEAN_i||_inf, SafeOpt.py:79-83  */
  int64_t count_S, count_U, count_M;
  int64_t count_G[SBO_MAX_Q];            /* [c-1]                                                         */
  int64_t n_exact_rechecks;              /* expander decisions that fell in the +1e-8 ambiguity band and
                                            were re-decided by exhaustive evaluation                      */
  /* Guard band of an approximating posterior kernel (K1b: Chebyshev core, K1t: Chebyshev-node interpolation; zero for the exact
   * kernels): decisions of the first pass that the band +- (sbo_profile.guard_dm, guard_dv) left open; when non-zero the
   * candidates concerned were re-evaluated by the exact kernel and the set phase ran again, so that every mask and index
   * returned is that of the exact posterior                                                                              */
  int64_t guard_band;                    /* decisions inside the band on the first pass (0: the first pass is the result)  */
  int64_t guard_rechecks;                /* candidates re-evaluated by the exact kernel                                     */
  int32_t guard_passes;                  /* set-phase passes of the re-evaluation (0: none was needed)                      */
  int32_t reserved_g;
} sbo_safeopt_result;

typedef struct sbo_goose_result {
  int64_t safe_min_index;                /* argmin_{S_t} lcb_0            models/GoOSE.py:63-67            */
  double  safe_min_x[SBO_MAX_D];
  double  safe_min_lcb;
  int64_t target_index_c[SBO_MAX_Q];     /* per constraint argmin_{O_c} lcb_0, -1 when empty   :82-112    */
  double  target_lcb_c[SBO_MAX_Q];
  int32_t target_best_c;
  int64_t target_index;
  double  target_x[SBO_MAX_D];
  double  target_lcb;
  int64_t explore_index;                 /* argmin_{S_t} ||x - target||_2  models/GoOSE.py:116-119        */
  double  explore_x[SBO_MAX_D];
  int32_t choose_safe_min;               /* min_safe_lcb <= target_lcb, test/test_GoOSE.py:158            */
  double  L[SBO_MAX_Q];
  int64_t count_S, count_U;
  int64_t count_O[SBO_MAX_Q];
  int64_t n_exact_rechecks;              /* candidates decided by the exhaustive reference predicate (expanders + coverage) */
  int64_t guard_band, guard_rechecks;    /* as in sbo_safeopt_result                                                        */
  int32_t guard_passes, reserved_g;
} sbo_goose_result;

typedef struct sbo_tr_result {
  int64_t index;                 /* argmin_{S_t and ||x - x_0|| <= r} lcb_0, -1 when that set is empty (models/GP_TR.py:43-51) */
  double  x[SBO_MAX_D];
  double  lcb;
  int64_t count_S, count_T;      /* |S_t|, |S_t intersected with the ball|                                                  */
  int64_t guard_band, guard_rechecks;    /* as in sbo_safeopt_result                                                        */
  int32_t guard_passes, reserved_g;
} sbo_tr_result;

/* per-kernel device time of the last sweep / posterior call, measured with HIP events on the
 * library's stream (feeds bench.py's roofline.achieved) */
typedef struct sbo_profile {
  double posterior_ms;     /* K1: fused cross-covariance + contraction + mean/var                          */
  double classify_ms;      /* K3: bounds, S/U masks, u* (with "set_fuse" the merge of the partials and the M mask
                              ride in the expander's first launches and are counted there)  (these three: 0 unless option */
  double expander_ms;      /* K4: distance transform + G_c / O_c decisions  "phase_events" is 1 -- an event  */
  double argreduce_ms;     /* K5: masked arg-max / arg-min                   costs a ~6 us bubble per record) */
  double comm_ms;          /* RCCL collectives (option "comm_events": an event pair around every call -- bubbles, so off by default) */
  double total_ms;         /* first launch to last completion                                              */
  double posterior_flops;  /* algorithmic flops of the K1 launch(es): q (n^2 + (2d+10) n) per candidate    */
  int64_t candidates;      /* candidates swept by this rank                                                */
  int32_t posterior_launches;
  int32_t posterior_kernel;      /* which K1 ran last: 1 generic, 2 generic chunked, 3 separable tables (K1g), 4 bilinear GEMMs (K1b),
                                    5 Chebyshev-node interpolation on 3-D / 4-D grids (K1t), 6 the same two GEMMs as 4 on coefficients
                                    interpolated from exactly evaluated Chebyshev nodes: a model's first sweep on a 2-D grid (K1i) */
  double posterior_executed_flops; /* matrix-core flops the last K1 launch(es) actually issued (K1b: far below the algorithmic count) */
  double posterior_setup_ms;     /* host time of the last per-(model, grid) table build of K1b, 0 when none was needed        */
  int64_t fp64_rechecks;         /* dtype SBO_F32 SafeOpt sweeps: candidates whose fp32 bounds could not decide S / U / u* / M / the
                                    minimiser and were re-evaluated in fp64 (option "fp64_recheck"); 0 otherwise                  */
  double recheck_ms;             /* device time of that step (band reductions, flagging, fp64 posterior of the list, scatter)    */
  double set_phase_ms;           /* K1 stop event -> end of the sweep: what the set phase (K3-K5 + collectives) adds to K1          */
  int32_t host_syncs;            /* host waits on the device inside the last sweep call (1 = the result read-back only)          */
  int32_t comm_calls;            /* collectives of the last sweep; comm_ms is their event-timed sum when option "comm_events" is 1 */
  int64_t comm_bytes;            /* multi-rank sweeps: bytes this rank handed to the collectives of the last sweep (send side)     */
  /* guard band of the posterior kernel that ran last (un-normalised units; zero for the exact kernels K1 / K1c / K1g)             */
  double guard_dm[SBO_MAX_Q], guard_dv[SBO_MAX_Q], guard_rl[SBO_MAX_Q];   /* |mean - exact|, |var - exact|, relative band of L       */
  double guard_ms;               /* device + host time of the last sweep's re-evaluation (0: none was needed)                      */
  int32_t halo_reruns;           /* multi-rank: set phases run again because SOME rank's speculative halo window was too narrow
                                    (the decision is global; host_syncs / comm_* include the discarded pass)                        */
  int32_t set_path;              /* set phase of the last SafeOpt sweep: 0 byte masks, 1 column words written by the GEMM posterior (r05)    */
  /* Standing audit of the guard band (r05; option "guard_audit" = samples per sweep, default 4096): behind every SafeOpt sweep on K1b /
   * K1i with a caller's invK, a rotating sample of the candidates is re-evaluated with the reference formula (models/GP_Safe.py:341-343)
   * on a side stream and compared with what the posterior kernel stored.  Cumulative over the context's life (a finished audit is
   * collected by the next call): value pairs compared, pairs whose |difference| exceeded the band (guard_dm / guard_dv) -- the claim
   * every mask and index rests on: must stay 0 --, and the largest deviation seen in units of the band.                             */
  int64_t guard_audit_samples, guard_audit_violations;
  double guard_audit_worst;
  /* What the band is made of (r05, K1b / K1i): the ANALYTIC part -- the truncations the plan makes, carried through the posterior formula
   * (csrc/guard.hip: k_gb_band; csrc/bilinear.hip: k_gb_band_i) -- and the largest deviation seen at the plan's 144 probe points, which
   * only CHECKS it: guard_dm = analytic + rounding floor, or 1e300 (plan not trusted, every sweep re-evaluates exactly) when
   * 4 x probe exceeds that.  Zero for the exact kernels and for K1t (whose band is 16 x 2048 probes, measured).                       */
  double guard_analytic_dm[SBO_MAX_Q], guard_analytic_dv[SBO_MAX_Q], guard_probe_dm[SBO_MAX_Q], guard_probe_dv[SBO_MAX_Q];
} sbo_profile;

/* ---- library / context ------------------------------------------------------------------- */
int sbo_version(void);
const char* sbo_last_error(void);
int sbo_device_count(int* count);
int sbo_init(int device_id, sbo_ctx** out);
int sbo_shutdown(sbo_ctx* ctx);
int sbo_synchronize(sbo_ctx* ctx);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ---------------------------------------- */
#define SBO_COMM_ID_BYTES 128
int sbo_comm_unique_id(void* id_out /* SBO_COMM_ID_BYTES, filled on rank 0 and sent to peers */);
/* world_size == 1 with id == NULL: no communicator (the collectives are identities and are skipped); with an id a
 * one-rank RCCL communicator is built all the same (see option "comm_selftest").  On failure the context stays
 * single-rank and usable. */
int sbo_comm_init(sbo_ctx* ctx, int world_size, int rank, const void* id);
int sbo_comm_barrier(sbo_ctx* ctx);
/* Rehearsal transport for test rigs where the ranks cannot form an RCCL communicator (e.g. two ranks sharing
 * the one GPU of a test box, which RCCL refuses): the same collectives, staged through host memory and carried
 * by caller-supplied functions (the tests use torch.distributed/gloo).  elem: 0 = uint64, 1 = double;
 * op: 0 = sum, 1 = max, 2 = min.  Not a production path: RCCL over xGMI is. */
typedef int (*sbo_relay_allreduce_fn)(void* user, void* buf, int64_t count, int elem, int op);
typedef int (*sbo_relay_allgather_fn)(void* user, const void* send, void* recv, int64_t bytes_per_rank);
int sbo_comm_init_relay(sbo_ctx* ctx, int world_size, int rank, sbo_relay_allreduce_fn allreduce,
                        sbo_relay_allgather_fn allgather, void* user);

/* ---- model state = inference_datasets (models/GP_Safe.py:236-245) -------------------------- */
/* hypopt is [d+2, q]: rows 0..d-1 = log ell_a, row d = log sigma_f, row d+1 = log sigma_n, consumed as
 * exp(2 h) (models/GP_Safe.py:338).  invK is q stacked [n, n] matrices or NULL (see sbo_factor).
 * kernel must be "RBF" (models/GP_Safe.py:159-162), anything else is SBO_E_INVALID. */
int sbo_model_set(sbo_ctx* ctx, int dtype, const char* kernel, int n, int d, int q,
                  const double* X_mean, const double* X_std, const double* Y_mean, const double* Y_std,
                  const double* X_norm, const double* Y_norm, const double* hypopt, const double* invK);

/* The same with invK as the reference holds it: `invKopt`, a list of q separate [n, n] arrays (models/GP_Safe.py:231-232,
 * 244) -- no stacking copy on the host.  invK_list == NULL as above. */
int sbo_model_set_list(sbo_ctx* ctx, int dtype, const char* kernel, int n, int d, int q, const double* X_mean,
                       const double* X_std, const double* Y_mean, const double* Y_std, const double* X_norm,
                       const double* Y_norm, const double* hypopt, const double* const* invK_list);

/* ---- candidates (resident in HBM until replaced) ------------------------------------------- */
/* One more observation (normalised coordinates / outputs, the caller's frozen X_mean, X_std, Y_mean, Y_std) under the
 * hyper-parameters of the last sbo_model_set: the lower factor gains one row and alpha is updated in O(n^2) on the
 * device.  SURVEY.md 8(f) rank 2 -- an opt-in fast path; the reference itself refits and renormalises on every sample
 * (models/GP_Safe.py:283-304).  Fails with SBO_E_UNSUPPORTED at n = SBO_MAX_N. */
int sbo_model_append(sbo_ctx* ctx, const double* x_norm_new, const double* y_norm_new);

/* explicit list: points[N, d] of doubles (dtype SBO_F64) or floats (SBO_F32); first_index = global flat
 * index of points[0] (shard offset). */
int sbo_candidates_points(sbo_ctx* ctx, const void* points, int points_dtype, int64_t n_local, int d,
                          int64_t first_index);
/* implicit tensor grid: x_a(i) = lo_a + i (hi_a - lo_a)/(count_a - 1), last = hi_a; this rank sweeps the
 * flat range [first_index, first_index + n_local). No HBM bytes are read for candidates. */
int sbo_candidates_grid(sbo_ctx* ctx, int d, const double* lo, const double* hi, const int64_t* count,
                        int64_t first_index, int64_t n_local);

/* same grid, sharded over the ranks of sbo_comm_init by whole hyper-planes of the slowest axis (rank r owns
 * planes [r P / W, (r+1) P / W)); returns this rank's flat range.  Required for multi-rank sweeps. */
int sbo_candidates_grid_sharded(sbo_ctx* ctx, int d, const double* lo, const double* hi, const int64_t* count,
                                int64_t* first_index_out, int64_t* n_local_out);

/* ---- hot path ------------------------------------------------------------------------------ */
/* K1: GP_inference for every local candidate; mean/var stay in HBM. */
int sbo_posterior_run(sbo_ctx* ctx);
/* copy out as [n_local, q] arrays of the model dtype (either pointer may be NULL) */
int sbo_posterior_get(sbo_ctx* ctx, void* mean_out, void* var_out);
/* batched BO.mean/ucb/lcb(points, index): out[n_local] of the model dtype (runs K1 if needed) */
int sbo_bounds(sbo_ctx* ctx, double b, int index, int kind, void* out);
/* full SafeOpt / GoOSE iteration on the resident candidates */
int sbo_sweep_safeopt(sbo_ctx* ctx, const sbo_sweep_opts* opts, sbo_safeopt_result* result);
int sbo_sweep_goose(sbo_ctx* ctx, const sbo_sweep_opts* opts, sbo_goose_result* result);
/* trust-region acquisition of models/GP_TR.py:43-51 on the resident candidates: x_0[d] centre, r radius */
int sbo_sweep_tr(sbo_ctx* ctx, const sbo_sweep_opts* opts, const double* x_0, double r, sbo_tr_result* result);
/* BO.explore_safeset(target) for a target of the caller's choosing (models/GoOSE.py:116-119): the candidate of the LAST sweep's safe set
 * closest to target[d] (scipy cdist's Euclidean distance, ties -> lowest flat index); x_out[SBO_MAX_D] may be NULL.
 * SBO_E_EMPTY_SAFE_SET when S_t is empty.  (sbo_sweep_goose answers the same question for its own target in its result.) */
int sbo_explore_safeset(sbo_ctx* ctx, const double* target, int64_t* index_out, double* x_out);
/* uint8 mask [n_local] of the last sweep (opts.want_masks); c is the constraint index for G / O */
int sbo_masks_get(sbo_ctx* ctx, int which, int c, uint8_t* out);

/* ---- model fit ------------------------------------------------- */
/* negative_loglikelihood (models/GP_Safe.py:169-192) for P hyper-parameter vectors at once:
 * hyper[P, d+2] rows = (log ell_0.., log sigma_f, log sigma_n); X_norm[n, d]; y[n] = one column of Y_norm;
 * out[p] = y^T K_p^-1 y + log|K_p| with K_p = sf2 exp(-1/2 D) + (sn2 + 1e-8) I (+inf if not positive definite).
 * This is the objective SciPy DE evaluates at models/GP_Safe.py:224, one population per call. */
int sbo_nll_batch(sbo_ctx* ctx, int n, int d, const double* X_norm, const double* y, int P, const double* hyper,
                  double* out);

/* The whole differential-evolution search of that objective on the device (models/GP_Safe.py:205-224: SciPy DE, default
 * best1bin strategy, mutation dithered in [0.5, 1), recombination 0.7; deferred updating here).  init_pop[P, d+2] is the
 * initial population (SciPy uses a Latin hypercube over the bounds lo / hi [d+2]); stops after maxiter generations or when
 * std(energies) <= atol + tol |mean(energies)|.  best_x[d+2], *best_energy, *generations are written on return; the
 * caller may polish best_x (SciPy does, with L-BFGS-B). */
int sbo_fit_de(sbo_ctx* ctx, int n, int d, const double* X_norm, const double* y, int P, const double* lo, const double* hi,
               const double* init_pop, uint64_t seed, int maxiter, double tol, double atol, double* best_x, double* best_energy,
               int* generations);

/* ---- plant evaluation ------------------------------------------ */
/* The reference's William-Otto reactor (problems/WilliamOttoReactor_Problem.py:19-93), noise-free, for n input rows
 * u[n, 2] = (Fb, Tr): out[n, 3] = (get_objective, get_constraint1, get_constraint2), each the steady state of the six
 * mass balances reached from x0 = 0.1 (the reference calls scipy fsolve once per point and output). */
int sbo_plant_wo(sbo_ctx* ctx, int64_t n, const double* u, double* out);

/* ---- measurement --------------------------------------------------------------------------- */
int sbo_profile_get(sbo_ctx* ctx, sbo_profile* out);
/* Diagnostics / test knobs (every default is the measured-best path; DESIGN.md "Options" has the A/B record of each):
 *   "posterior_path"   0 auto | 1 generic single-phase kernel (K1) | 2 generic chunked kernel (K1c)
 *   "bilinear"         1: fp64 2-D grids run the posterior as two GEMMs on Chebyshev coefficients -- a model's first sweep with a caller's
 *                      invK from exactly evaluated nodes (K1i), later sweeps from the reduced-basis plan (K1b) when the bases qualify;
 *                      2: K1b's plan from the first sweep on; 0: K1g
 *   "cheb_tol_e17"     K1b / K1i: coefficients below this x 1e-17 of the largest are not run (default 400 = 4e-15)
 *   "tensor_cheb"      1: fp64 3-D / 4-D grids interpolate the exact posterior from a tensor grid of Chebyshev nodes (K1t); 0: K1g
 *   "tensor_guess_pct" K1t test hook: scales the first guess of the node counts (a short guess exercises the probe's retry)
 *   "fuse_classify"    one-constraint K1b sweeps take S / U from the posterior's mean epilogue: 1 | 0 | -1 auto (default)
 *   "chol_async"       1: a caller's invK is contracted as given by K1b and its reverse Cholesky factor built off the critical path
 *   "scan_blocks"      1: blocked last-axis scans of the distance transforms; 0: step by step (A/B checker)
 *   "scan_waves"       1: open candidates of the verdict kernels are listed and scanned by groups of 16 lanes (8 | 32 | 64: lanes); 0: own thread
 *   "goose_pairs"      1: GoOSE coverage by pruned pair evaluation on grids too (A/B checker of the power transform)
 *   "col_path"         1 (auto): one-constraint SafeOpt sweeps of one rank on 2-D grids of whole 64 x 128 posterior tiles -- at least four tiles
 *                      per CU and output -- run their set phase on column words written by the GEMM posterior's epilogue
 *                      (sets_colpath.inc.hpp); 2: on every grid of that shape; 0: the byte-mask pipeline (A/B checker)
 *   "grad_defer"       first sweep of a model on fp64 2-D grids (K1i): where the Lipschitz keys' gradient phases run.  0: inside the posterior
 *                      launches, behind the gate's kernels (r04); 1 (default): in a launch of their own on a side stream beside the posterior
 *                      launches -- behind the gate on large grids, on every tile without a gate on small ones; 2: always without, 3: always with
 *   "col_overlap"      1: on that path the expander chain (distance transform, verdicts) runs on a second stream beside the objective's
 *                      posterior launch; 0: every kernel on the main stream
 *   "set_fuse"         1: 2-D grids of one rank share launches between independent set-phase kernels; 0: one launch per kernel
 *   "set_lanes"        1: constraints of a one-rank sweep alternate between two streams; 0: one after the other
 *   "exact_lazy"       1: one-constraint sweeps launch the exhaustive recheck only when in-band verdicts were listed; 2: always that late path (test); 0: eager
 *   "result_mirror"    1: the last kernel of a one-rank sweep writes the result block into pinned host memory; 0: a copy behind it
 *   "phase_events"     1: events between the set phases too (sbo_profile.classify_ms / expander_ms / argreduce_ms; ~6 us bubble each)
 *   "halo_spec"        1: ranks > 1 size their transform windows from the previous sweep's keys (device-checked); 0: wait for this sweep's
 *   "comm_events"      1: an event pair around every collective (sbo_profile.comm_ms)
 *   "comm_selftest"    1: a one-rank communicator still sends C1 / C2 / C3 through RCCL (results must not change)
 *   "fp64_recheck"     1: fp32 models re-evaluate in fp64 every candidate their 1e-4 contract cannot decide; 0: masks of the fp32 posterior
 *   "guard_audit"      samples per audited sweep of the standing audit of the guard band (sbo_profile.guard_audit_*; default 1024); 0: off
 *   "guard_audit_scale_ppm" test hook: the audit compares against the band x value / 1e6 (default 1000000); setting it clears the counts
 *   "guard_audit_every" one sweep in this many carries an audit (default 16; the first sweep after setting it does).  An audit shares
 *                      the card with the sweep it follows (~35 us of config H's set phase at 1024 samples, n = 512): 1 audits every sweep
 *   "guard_band"       1: sweeps on an approximating posterior (K1b / K1i / K1t) count the decisions inside its band and re-evaluate exactly
 *                      when there are any; 0: masks of the approximating posterior as they come; 2: the re-evaluation on every sweep (test) */
int sbo_set_option(sbo_ctx* ctx, const char* key, int64_t value);

#ifdef __cplusplus
}
#endif
#endif /* SAFEBO_H */
