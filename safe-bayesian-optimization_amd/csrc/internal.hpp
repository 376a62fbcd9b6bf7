// internal.hpp -- shared declarations of libsafebo.so (host side + kernel argument blocks).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/safebo.h"

namespace sbo {

// The direct reference of the guard band (guard.hip: k_ref_list) keeps 16 cross-covariance vectors of npad doubles + 32 KB of partial
// sums in LDS: 160 KB at npad = 1024.  Larger models do not take the GEMM posteriors with a caller's matrix (they run K1g, whose
// factor is made on demand) -- ADVICE r04.
constexpr int kGuardRefMaxNpad = 1008;
constexpr int kMaxD = SBO_MAX_D;
constexpr int kMaxQ = SBO_MAX_Q;
constexpr int kGbMirrorOffset = 7680;   // pinned landing block h_back: [0, 4096) end-of-sweep read-back | 4096 plan records | 7168 audit counts | 7680 guard band

// Model constants handed to kernels by value (kernarg segment).  Everything is kept in double; the
// fp32 kernels round on use.  Rows a1/a4 of SURVEY.md section 8 (models/GP_Safe.py:236-245, 326-347).
struct ModelConst {
  int n, npad, d, dpad, q, factor;
  double X_mean[kMaxD], X_std[kMaxD], X_rstd[kMaxD];
  double Y_mean[kMaxQ], Y_std[kMaxQ];
  double sf2[kMaxQ];            // exp(2 h[d])                       GP_Safe.py:338
  double sn2[kMaxQ];            // exp(2 h[d+1]) + float32 eps       GP_Safe.py:229 (used by the model build / append only)
  double mp[kMaxQ];             // prior mean, -2 Y_mean/Y_std, [0]=0 GP_Safe.py:331-332
  double vinv[kMaxQ][kMaxD];    // ell^-1/2 = exp(-h[a])             GP_Safe.py:112
  double inv_ell[kMaxQ][kMaxD]; // 1/ell = exp(-2 h[a])
};

// Candidate description (explicit list or implicit tensor grid), by value.
struct CandSpec {
  int kind;        // 0 explicit points, 1 grid
  int d;
  int pts_dtype;   // SBO_F64 / SBO_F32 for explicit points
  int pad;
  const void* pts; // device pointer [n_local, d]
  long long n_local, first;
  double lo[kMaxD], hi[kMaxD], step[kMaxD];
  long long count[kMaxD];
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

// Column-word form of the masks of a 2-D grid (r05): word [s][i] holds the bits of column i (axis 0) for the 64 rows
// 64 s .. 64 s + 63 of axis 1 -- exactly what one 64 x 128 tile of the GEMM posterior (k_bpost) knows about a column, so the
// constraint's mean epilogue emits the S / U words itself (128 coalesced 8-byte stores per tile instead of 16384 byte
// stores), and every set-phase kernel of the column path (sets_colpath.inc.hpp) reads 1/8 of the bytes.  Usum[i]: bit s set
// iff Uw[s][i] != 0 (the carries of the column distance transform in two dependent loads).
struct ColBits {
  unsigned long long* Sw;
  unsigned long long* Uw;
  unsigned long long* Usum;
  unsigned long long* slots;   // [kColSlotFields][kColSlots]: the tiles' partial scalars, merged by atomics as the workgroups end
};
// One workgroup merging the 2 x 2048 partial rows of config H took 15 us at the head of the set phase.  Instead a tile's workgroup
// adds / maxes / mins its scalars into slot (tile mod 64) of each field (64 addresses per field: no queueing in L2), and whoever
// needs a scalar reduces 64 words with one load per lane.  The finals reset the block for the next sweep.
constexpr int kColSlots = 64;
enum ColSlotField { kSlotUmin = 0 /* min key of ucb_0 over S */, kSlotS, kSlotU, kSlotB /* counts */, kSlotVmin0, kSlotVmin1 /* min keys */,
                    kSlotRmax1 /* max key */, kSlotL0, kSlotL1 /* max of the gradient norms' bit patterns */,
                    kSlotRowMask /* word s: bit t set iff tile t of tile row s holds a safe candidate */, kColSlotFields };
__host__ __device__ inline bool col_slot_is_min(int f) { return f == kSlotUmin || f == kSlotVmin0 || f == kSlotVmin1; }
// what a k_bpost launch is told beyond its operands (r05: one launch per output on the column path)
struct PostExtra {
  int o0;        // first output of this launch (blockIdx.z counts from it)
  int q;         // outputs of the model (layout of the gradient gate's tables)
  int lean;      // 1: an objective tile without a safe candidate does not store its mean / var (nobody reads them)
  long long fstride; // fused classification of several constraints: stride of the S / U byte planes (constraint o writes plane o - 1; 0: one constraint, the masks themselves)
  int nograd;    // 1: the gradient phases (Lipschitz keys) run in a launch of their own (k_bpost<.., 3>, K1i's deferred gate): none here
  ColBits cb;    // Sw == nullptr: no bit words (byte masks or no classification at all)
};

// K1b (bilinear.hip): device tables of the reduced-basis posterior on a 2-D grid, valid for one (model, candidates) pair
struct BilinearPlan {
  bool valid = false;    // built (or found unusable) for the current model and candidates
  bool usable = false;   // the bases qualified: the posterior runs as two GEMMs
  int KB0 = 0, KB1 = 0;  // 16-blocks of the pair indices of axis 0 / axis 1
  int r0u = 0, ncs0 = 0, nrb = 0;
  int KS0 = 0, KBm = 0, KSm = 0, KBm2 = 0;   // k-steps of the variance phase; k-blocks / k-steps of the mean phases
  size_t sVA = 0, sSBf = 0;
  long long nlines_pad = 0;
  size_t sP0f = 0, sP1A = 0, sT4f = 0, sBtA = 0;   // per-output strides (elements)
  int r0[kMaxQ] = {0}, r1[kMaxQ] = {0};
  double setup_ms = 0.0;
  int* eff = nullptr;    // device: per-output counts the kernels run to (k_cheb_trunc), followed by the truncation tails (q doubles)
  const double* gtmax = nullptr;              // device: largest gradient samples per k_bpost tile + slacks (k_bl_gradcoarse), or none
  const unsigned long long* gkey = nullptr;   // device: the grid's largest gradient samples
  bool band_ready = false;   // the guard band of this plan has been measured (guard.hip: guard_band_bilinear)
};

// K1i: the first sweep of a model by interpolation from Chebyshev nodes (bilinear.hip)
struct InterpPlan {
  bool valid = false;    // enqueued (or found not applicable) for model `serial` and the current grid
  bool usable = false;
  bool used = false;     // a sweep has run on it: the next one builds K1b's own plan
  bool band_ready = false;
  unsigned long long serial = 0;
  int Dn = 0, KB = 0, ncs0 = 0, nrb = 0;
  size_t sT4f = 0, sBtA = 0;
  int* eff = nullptr;    // device: per coefficient set (4 per output) the counts the kernels run to, then the truncation tails
  const double* gtmax = nullptr;              // which tiles run the gradient phases (as BilinearPlan's)
  const unsigned long long* gkey = nullptr;
  double *grad_S0 = nullptr, *grad_Vb = nullptr, *grad_gt = nullptr;
  unsigned long long* grad_key = nullptr;
  // the plan as a HIP graph: captured when a signature repeats, replayed with the next model's parameter block
  struct Sig {
    CandSpec cs;
    int Dn, q, n, npad, dpad, guard, a_ld, gate;
    double cheb_tol;
    const void* ptr[32];
  } sig;
  bool sig_valid = false, graph_ok = false;
  void* exec = nullptr;                       // hipGraphExec_t
  bool grad_deferred = false;                 // the gate's kernels run on stream3 beside the plan's tail: the posterior launches carry no
                                              // gradient phases, a launch of those alone follows the gate there (launch_posterior_interp)
};
using InterpSig = InterpPlan::Sig;

}  // namespace sbo

struct sbo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // side stream: the K1b axis bases of a new model run next to its factorisation
  hipStream_t stream_audit = nullptr;   // lowest priority: the standing audit of the guard band (guard.hip) fills what the sweeps leave idle
  hipStream_t stream4 = nullptr;   // the deferred factorisation of a caller's invK (chol_async): off the critical path of a model change
  hipStream_t stream3 = nullptr;   // spare high-priority stream (drained with the others)
  int n_cu = 256;
  // model
  bool has_model = false;
  int dtype = SBO_F64;
  sbo::ModelConst mc{};
  sbo::DevBuf Fpk;     // [q][ntri*4*64] packed MFMA A-fragments of the contraction matrix
  sbo::DevBuf As;      // [q][npad][dpad]  X_norm * ell^-1/2
  sbo::DevBuf sqA;     // [q][npad]        sum_a As^2
  sbo::DevBuf alpha;   // [q][npad]        invK (Y_norm - mp)
  sbo::DevBuf Xn;      // [npad][dpad]     X_norm (for the mean gradient)
  sbo::DevBuf E0f;     // grid path: axis-0 kernel table in B-fragment order [q][tiles0][npad/4][64]
  sbo::DevBuf Er;      // grid path: tables of the remaining axes [q][sum_a count_a][npad]
  sbo::DevBuf AXg;     // grid path: alpha_j (1, Xn_j) rows in fragment-slot order [q][npad][1 + dpad]
  size_t fpk_stride = 0;  // elements per output in Fpk
  std::vector<double> h_Xnorm;   // host copies used by the exact-recheck / result decoding
  sbo::DevBuf Fplain;            // [q][f_cap][f_cap] fp64 lower factor M, row-major (K1b table build, sbo_model_append)
  sbo::DevBuf alpha64;           // [q][a_ld] fp64 alpha
  int f_cap = 0;                 // leading dimension / capacity of Fplain: n after a build, n + 256.. once an append has grown it
  int a_ld = 0;                  // stride of alpha64: npad after a build, f_cap after an append
  sbo::DevBuf mwork;             // model build workspace (uploads, fp64 copies of the derived arrays, the factorisation's scratch)
  // Caller's invK on a K1b-capable grid (option chol_async): the GEMM posterior's tables contract with invK itself -- packed
  // here as full matrix-core images, exactly the matrix of models/GP_Safe.py:341-343 -- so the reverse Cholesky factor M (needed
  // by the O(n^2) kernels K1g / K1 / K1c and by sbo_model_append only) is built on stream4 while the caller goes on; whoever
  // needs it calls factor_sync first, which also delivers the positive-definiteness verdict
  sbo::DevBuf invk_img;            // [q][npad / 16][npad / 16][256] A images of the full invK (fp64)
  bool invk_img_valid = false;
  bool invk_w_valid = false;       // the uploaded invK of the current model is still in the build workspace (images can be packed later)
  bool factor_todo = false;        // the chain has not been enqueued yet (sbo_model_set does that last: model_factor_enqueue)
  bool factor_pending = false;     // the factor chain of the current model is (possibly) still running; ev_factor marks its end
  hipEvent_t ev_factor = nullptr, ev_w = nullptr;
  int chol_async = 1;
  int exact_lazy = 1;      // one-constraint SafeOpt sweeps on one rank: k_expander_exact only when the result block reports in-band candidates
  double cheb_tol = 4e-15; // ... relative size below which trailing coefficients are not run (option cheb_tol_e17, in units of 1e-17)
  sbo::BilinearPlan bl;
  sbo::DevBuf bl_P0f, bl_P1A, bl_T4f, bl_BtA, bl_SBf, bl_VA, bl_small, bl_work, bl_cheb;
  // K1t (tensor.hip): fp64 grids of three / four axes by Chebyshev interpolation from exact node values
  int tensor_cheb = 1;             // option: 0 = always K1g
  int tensor_guess_pct = 100;      // option (test hook): scales the first guess of the node counts; a short guess exercises the probe's second attempt
  const double* k1g_axc = nullptr; // K1g launch arguments of launch_posterior_on_axes (explicit axis positions, gradient output)
  void* k1g_grad = nullptr;
  bool tensor_busy = false;        // the exact node / probe launch of K1t is running through launch_posterior
  sbo::DevBuf tn_pts, tn_vals, tn_work, tn_W0t, tn_W1t, tn_probe, tn_scr;
  sbo::DevBuf tn_gather;     // ranks > 1: this rank's slab of the node tensors | every rank's (all-gather)
  sbo::DevBuf tn_W[SBO_MAX_D];
  size_t tn_work_half = 0;
  double tn_flops = 0.0;           // multiply-add flops of the last interpolation pass
  bool tn_valid = false, tn_usable = false;     // plan decided for (tn_model, grid below) / it passed its accuracy probe
  unsigned long long tn_model = 0;
  long long tn_first = 0, tn_nlocal = 0, tn_count[4] = {0, 0, 0, 0};
  double tn_lo[4] = {0, 0, 0, 0}, tn_hi[4] = {0, 0, 0, 0};
  int tn_level[4] = {0, 0, 0, 0};
  int tn_dn[4] = {0, 0, 0, 0};     // node counts of the plan in use
  sbo::DevBuf tn_tail;             // K1t: [2 q][d] keys of the node tensors' coefficient tails per axis (k_t_fiber_tail)
  double tn_band[7 * SBO_MAX_Q] = {0};   // the plan's guard band (dm | dv | rl | analytic dm | dv | probe dm | dv per output)
  int tn_bump = 0;                 // ladder steps added to the first guess on this grid (a previous model's plan needed its second attempt)
  // Guard band of the approximating posteriors K1b / K1t (device_common.hpp: GuardBand; guard.hip)
  int guard_band = 1;              // option: 1 count + re-evaluate exactly when the count is non-zero; 0 off; 2 re-evaluate on every sweep (test)
  bool gb_active = false;          // the posterior in mean / var came from an approximating kernel; `gb` holds (or will hold, in stream order) its band
  bool gb_off = false;             // a recheck's inner sweep on fully refined values: no band
  bool gb_slow = false;            // the re-evaluation path of the SafeOpt sweep is running (Lipschitz keys exact, lists in use)
  sbo::DevBuf gb;                  // GuardBand of the resident posterior
  long long guard_first = 0;       // decisions the first pass of the running sweep left open
  // Standing audit of the band (r05): behind every K1b / K1i posterior launch of a sweep a rotating sample of the candidates is
  // re-evaluated with the reference formula on stream2 -- off the critical path -- and compared with what the posterior kernel
  // stored: a deviation beyond the band is a VIOLATION of the claim the sweeps' exactness rests on (counted, reported in sbo_profile)
  int guard_audit = 1024;          // option: samples per audited sweep, 0 = off
  int guard_audit_every = 16;      // option: one sweep in this many is audited (the context's first one is)
  long long audit_tick = 0;
  double audit_scale = 1.0;        // option guard_audit_scale_ppm (tests): the audit compares against band x this -- a way to see it fire
  sbo::DevBuf audit_pts, audit_val, audit_part, audit_cnt;
  hipEvent_t ev_audit[2]{};        // the sample has been taken (mean / var may be overwritten) / the audit has finished
  bool audit_pending = false;
  unsigned long long audit_offset = 0;
  long long audit_samples = 0, audit_violations = 0;
  double audit_worst = 0.0;        // largest deviation seen, in units of the band
  bool gb_mirrored = false;        // the plan's band kernel also wrote the block to pinned host memory (h_back + kGbMirrorOffset): valid once a sweep has synchronised
  bool gb_host_valid = false;      // gb_host mirrors `gb` (read back on demand by sbo_profile_get; dropped when a plan writes the block)
  double gb_host[7 * SBO_MAX_Q] = {0};   // dm | dv | rl | analytic dm | dv | largest probe deviation dm | dv
  sbo::DevBuf gb_pts, gb_vals;     // re-evaluation: coordinates and exact values of the listed candidates
  sbo::DevBuf gb_part;             // k_ref_list: per-column-chunk partial sums
  sbo::DevBuf gb_probe;            // K1b: probe indices / coordinates / reference values of the running plan
  sbo::DevBuf list_scr;            // key scratch of launch_posterior_on_list
  unsigned long long gb_plan_model = 0;   // K1b: the model whose band `gb` holds (0: none)
  long long gb_plan_first = -1, gb_plan_n = -1;
  const double* invk_plain = nullptr;     // the caller's invK as uploaded, [q][n][n] (valid while invk_w_valid)
  sbo::DevBuf bl_basis;            // K1b: the 2 q axis bases (U, Chebyshev series, ranks) and the workspace of their kernel
  bool bl_basis_ok = false;        // bases enqueued for (bl_basis_serial, bl_basis_ab); their (ok, r, rc) records land at h_back + 4096
  unsigned long long bl_basis_serial = 0;
  double bl_basis_ab[4] = {0, 0, 0, 0};
  unsigned long long model_serial = 0;   // bumped by every sbo_model_set / sbo_model_append
  // candidates
  bool has_cand = false;
  sbo::CandSpec cs{};
  sbo::DevBuf pts;
  // posterior workspace, SoA [q][n_local] of the model dtype
  sbo::DevBuf mean, var;
  bool posterior_valid = false;
  int last_k1 = 0;               // kernel family of the last posterior launch (sbo_profile.posterior_kernel)
  double last_k1_flops = 0.0;    // matrix-core flops it issued
  sbo::DevBuf Lmax;    // [kMaxQ] uint64 keys: max ||grad MEAN_i||_inf over the candidates
  // set workspace
  sbo::DevBuf maskS, maskU, maskM, maskG, maskO;   // uint8 [n_local] (G/O: [(q-1)][n_local])
  sbo::DevBuf dist2;   // double [n_local] distance-transform scratch (x2 for ping-pong)
  sbo::DevBuf dist2b;
  sbo::DevBuf fitbuf, fitwork;   // hyper-parameter objective: inputs/outputs and the P x n x n factor workspace
  sbo::DevBuf coarse;  // coarse U mask + its distance transform (expander pre-decision)
  sbo::DevBuf scal;    // small device scalar block (keys, counters, arg-reduce results)
  // Second lane of the set phase (models with two or more constraints, one rank): the per-constraint chains of a sweep are
  // independent, so every other constraint is enqueued on stream2 with its own scratch and its own copy of the scalar block
  // (the chains only write its recheck / scan counters).  The host swaps these in and out of the fields above around the
  // calls that enqueue a lane-1 constraint (sets.hip: lane_swap).
  struct SetLane {
    sbo::DevBuf dist2, dist2b, coarse, blockmin, blockmax, scanlist, amb, gw, runmeta, scal;
    bool amb_clean = false;
  } lane1;
  int set_lanes = 1;       // 0: all constraints on the main stream, one after the other
  sbo::DevBuf partial; // arg-reduce per-block partials
  sbo::DevBuf amb;     // ambiguous-index list for the exact recheck
  sbo::DevBuf runmeta; // GoOSE: per-run bounding boxes / radii of the coverage search
  sbo::DevBuf scanlist; // candidates left open by the coarse expander decision (wave-per-candidate scan)
  sbo::DevBuf blockmax; // per-block largest source weight along axis 0 (blocked axis-0 pass of the power transform)
  sbo::DevBuf blockmin; // per-block minima along the last axis (blocked last-axis scans)
  sbo::DevBuf gw;      // GoOSE: source weights (ucb_c on sources, -inf elsewhere), T [max shard]
  sbo::InterpPlan bi;   // K1i plan (first sweep of a model)
  sbo::DevBuf bi_params; // ... its per-model parameter block (device) and the pinned staging the graph's copy node reads
  void* h_bi_params = nullptr;
  void* ev_bi_params = nullptr;   // hipEvent_t: the plan's copy of the block (and everything before it on the main stream) has run
  sbo::DevBuf bl_grad;  // K1b: which tiles run the gradient phases (per plan)
  sbo::DevBuf bl_lpart; // K1b: per-wave Lipschitz partials of k_bpost
  sbo::DevBuf cpart;   // per-workgroup partials of k_classify, field-major [kClassifyRow][cpart_cap]
  int cpart_cap = 0;   // row capacity the last writer of cpart laid its rows out with
  long long comm_bytes = 0;   // collectives of the running sweep: bytes handed over (send side), calls, and -- option comm_events --
  int comm_calls = 0;         // an event pair per call (comm_ev, created on first use) whose elapsed times sbo_profile.comm_ms sums
  int comm_events = 0;
  int comm_nev = 0;
  double comm_host_ms = 0.0;  // (relay transport: wall clock of the staged collectives)
  hipEvent_t comm_ev[16]{};
  int host_syncs = 0;  // host waits on the device inside the running sweep call (sbo_profile.host_syncs)
  // classification fused into the posterior (K1b, one constraint): a sweep sets fuse_request / fuse_b before it enqueues the
  // posterior; fuse_rows > 0 afterwards = the kernel wrote S / U and that many partial rows at the head of cpart
  sbo::DevBuf fuseS, fuseU;   // [(q - 1)][N] byte planes of a fused classification of several constraints (r05)
  int fuse_request = 0;    // 0 no, 1 yes, 2 when the GEMM launch is large enough for it to pay (option fuse_classify = -1)
  double fuse_b = 0.0;
  int fuse_rows = 0;
  // Column path (r05, sets_colpath.inc.hpp): a one-rank SafeOpt sweep of a one-constraint model on a 2-D grid of whole
  // 64 x 128 tiles asks (col_request) for the classification as column words; the GEMM posterior then runs one launch per
  // output -- constraint first: S / U words, |S| per tile; objective second: u* and min var_0 over S from its own epilogue --
  // and says so (col_active).  The words stay resident for sbo_masks_get (masks_bits: expanded to bytes on demand).
  bool col_request = false, col_active = false;
  int sweep_lean = 0;      // the running SafeOpt sweep is lean: the objective's Lipschitz key L_0 (no sweep reads it) is not computed by K1t / K1i
  int col_lean = 0;        // the running request: objective tiles without a safe candidate need not store mean / var
  bool slots_clean = false;// the slot block holds its neutral elements (the finals of the last column sweep reset it)
  bool usum_dirty = false; // Usum holds bits of an earlier launch (cleared by the column path's second kernel; by a memset after a failure)
  int col_path = 1;        // option: 0 = never
  int col_overlap = 1;     // option: 1 = the expander chain runs on stream3 beside the objective's posterior launch; 0: one stream
  sbo::DevBuf cbS, cbU, cbM, cbG, cbUsum;    // column words [H / 64][W]; Usum [W]
  sbo::DevBuf col_slots;                     // ColBits::slots
  sbo::DevBuf col_img, col_bmin;             // column distance image u16 [H][W], block minima u16 [H][W / 32]
  sbo::DevBuf col_cimg, col_cbmin;           // the same on the 8 x 8 cells
  long long col_ckey = 0;                    // the grid the padding of col_cbmin was laid out for
  sbo::DevBuf col_fin;                       // the objective's scalars, the finals' tickets and intermediate rows (4 KB)
  hipEvent_t ev_col[2]{};      // fork (the constraint's posterior launch has finished) / join (the expander chain on stream3 has)
  hipEvent_t ev_grad[4]{};     // K1i's deferred tail: fork (plan: the series are in place) / stage 1 has run / the keys are merged / the band is written
  bool grad_pending = false;   // a deferred gradient launch is in flight on stream3: whoever reads the Lipschitz partials elsewhere waits for ev_grad[2]
  int grad_defer = 1;          // option: 0 = the gate stays in front of the posterior launch (r04)
  bool col_forked = false;     // the constraint's launch of the running posterior carried ev_col[0]
  bool masks_bits = false;     // the masks of the last sweep live in the column words (byte buffers stale)
  bool col_G_bytes = false;    // ... except G, which the exhaustive recheck finished in byte form
  // Lipschitz keys of K1b: a sweep sets lmax_defer before it enqueues the posterior; the posterior then leaves its per-wave
  // partials (lmax_per_out per output) for the sweep's k_classify_final / k_edt_axis0_pair to merge (lmax_pending)
  bool lmax_defer = false;
  bool lmax_pending = false;
  int lmax_per_out = 0;
  sbo::DevBuf Wfull;   // multi-rank GoOSE: source weights of the whole grid (all-gathered), T [grid_total]
  sbo::DevBuf Uwin;    // multi-rank: U mask of the expander transform's window (own planes + halo), uint8
  long long uwin_first = 0, uwin_n = 0;   // flat range the window covers
  sbo::DevBuf ubits;                      // multi-rank: own U bits (gather_words words) followed by every rank's (all-gathered)
  long long gather_words = 0;             // words per rank in ubits
  sbo::DevBuf gather;  // multi-rank: all-gather receive buffer [world][max_local]
  sbo::DevBuf xch;     // multi-rank: small exchange buffers (C1 keys, C3 rows)
  sbo::DevBuf shard_first;  // device copy of first_of[]
  long long grid_total = 0; // candidates in the whole grid (all ranks)
  bool sharded = false;     // candidates were set with the canonical plane sharding
  std::vector<long long> first_of;   // [world + 1] flat offsets of the rank shards
  unsigned long long* h_c1 = nullptr;     // pinned host copy of the C1 keys (global u*, L, radius) of the running sweep
  // Speculative halo (ranks > 1, option halo_spec): the host sizes the transform window of constraint c from the keys of the
  // PREVIOUS sweep (with a margin) instead of waiting for this sweep's; the device checks the guess against the keys it
  // gathered (SweepScalars::halo_short) and a short guess reruns the set phase the waiting way.  -1: no guess yet.
  long long halo_guess[SBO_MAX_Q] = {-1, -1, -1, -1, -1, -1, -1, -1};
  int halo_spec = 1;
  int halo_reruns = 0;            // set phases of the current sweep call discarded for a short window (global decision)
  bool in_halo_rerun = false;
  bool c1_pending = false;                // the read-back of h_c1 has been enqueued (event ev[5]) but not yet waited for
  void* h_stage = nullptr;                // pinned staging of the K1b table build's single upload
  size_t h_stage_bytes = 0;
  unsigned char* h_back = nullptr;        // pinned host landing area of the end-of-sweep read-back (scalars + Lipschitz keys)
  int last_sweep = 0;  // 1 safeopt, 2 goose (what the masks hold)
  bool masks_valid = false;
  bool k1_stop_attached = false;   // the last K1 kernel carries ev[1] as its stop event (no separate record, which costs a ~6 us bubble)
  bool amb_clean = false;   // the recheck / scan counters of the scalar block are still zero (no k_reset_amb needed)
  // profile
  sbo_profile prof{};
  hipEvent_t ev[8]{};
  hipEvent_t ev_join[SBO_MAX_Q]{};   // phase marks of the fp32 recheck
  // options
  int scan_waves = 1;      // 1: candidates the coarse bounds leave open are scanned one wave each (0: by their own thread)
  int scan_blocks = 1;     // 0: step-by-step last-axis scans (A/B against the blocked form)
  int result_mirror = 1;   // SafeOpt sweeps on one rank: the last kernel writes the results into the pinned host block itself (0: a copy behind it)
  int set_fuse = 1;        // 2-D grids: independent set-phase kernels share launches (k_edt_axis0_pair, k_set_mid); 0: one launch each
  int fuse_classify = -1;  // one-constraint sweeps on the K1b path take their S / U bytes from the posterior kernel's mean epilogue: 1 always, 0 never, -1 (default) when the launch has at least four workgroups per CU (r03, sqrt-free sign tests: config H -40 us, config B +-0)
  int goose_pairs = 0;     // 1: GoOSE coverage by pruned pair evaluation on grids too (A/B against the transform)
  int phase_events = 0;    // 1: events between the set phases too (classify / expander / arg-reduce times in sbo_profile)
  int bilinear = 1;        // 1: fp64 2-D grids run the posterior as two GEMMs in a reduced basis when the bases qualify (K1b)
  int posterior_path = 0;  // 0 auto (separable tables on aligned grids), 1 force the generic exp() kernel
  // fp32 models: an fp64 twin of the model (same arrays, double images) that re-evaluates the candidates the fp32 bounds
  // cannot decide (option fp64_recheck); it shares this context's streams and pinned areas and holds explicit lists only
  sbo_ctx* shadow = nullptr;
  bool is_shadow = false;
  int fp64_recheck = 1;
  sbo::DevBuf rc_mean, rc_var;   // double [q][n_local]: the fp32 posterior widened, flagged entries replaced by fp64 values
  sbo::DevBuf rc_list;           // flagged candidate indices (long long) + counters (64-byte head)
  sbo::DevBuf rc_refined;        // uint8 [n_local]: this candidate's entries of rc_mean / rc_var are fp64 values
  bool rc_active = false;        // the running set phase works on a partly refined fp32 posterior (verdicts carry bands)
  // comm
  void* comm = nullptr;  // ncclComm_t
  int world = 1, rank = 0;
  int comm_selftest = 0; // 1: a one-rank world still sends C1 / C2 / C3 through its communicator (test of the RCCL calls on one GPU)
  // rehearsal transport (tests on a 1-GPU box): collectives staged through host callbacks instead of RCCL
  sbo_relay_allreduce_fn relay_allreduce = nullptr;
  sbo_relay_allgather_fn relay_allgather = nullptr;
  void* relay_user = nullptr;
};

namespace sbo {
// the sweeps take their multi-rank path (pack, collective, unpack, host merge): more than one rank, or the self-test
inline bool multi_rank(const sbo_ctx* c) { return c->world > 1 || c->comm_selftest; }
int fail(int code, const std::string& msg);
int hip_fail(hipError_t e, const char* what);
int ensure(DevBuf& b, size_t bytes);
hipError_t stream_wait(const sbo_ctx* c, hipStream_t st);   // hipStreamSynchronize, polling first (option spin_wait)
void release(DevBuf& b);
void drain_streams(sbo_ctx* c);
int factor_sync(sbo_ctx* c);      // wait for a deferred factorisation (sbo_ctx::factor_pending); SBO_E_INVALID when invK was not positive definite   // after a failed call: wait for whatever it left on the main and side streams

// launchers implemented in the .hip files -----------------------------------------------------
int launch_posterior(sbo_ctx* c);
int launch_bound(sbo_ctx* c, double b, int index, int kind, void* dev_out);
int model_build(sbo_ctx* c, const double* const* host_invK, const double* X_norm, const double* Y_norm);
int model_prep(sbo_ctx* c, const double* X_norm);
int model_factor_enqueue(sbo_ctx* c);
int model_pack_invk(sbo_ctx* c);         // images of the caller's invK for the K1b tables, when the grid arrived after the model   // the deferred factor chain of a caller's invK, behind everything on the critical path
int model_append(sbo_ctx* c, const std::vector<double>& kvec /*[q][n]*/, const double* kappa, const double* rho);
int model_repack(sbo_ctx* c);
bool bilinear_applicable(const sbo_ctx* c);
int bilinear_basis_enqueue(sbo_ctx* c, hipStream_t st, bool force_big);
bool interp_applicable(const sbo_ctx* c);
int interp_setup(sbo_ctx* c);
int launch_posterior_interp(sbo_ctx* c);
int guard_probe_gradients(sbo_ctx* c, hipStream_t st, double* ppts, double* grad_out, const sbo::ModelConst* mcp = nullptr);
bool guard_reference_is_direct(const sbo_ctx* c);
int bilinear_setup(sbo_ctx* c);
int launch_posterior_bilinear(sbo_ctx* c);
bool tensor_applicable(const sbo_ctx* c);
int launch_posterior_tensor(sbo_ctx* c, bool* declined);
int launch_posterior_on_axes(sbo_ctx* c, int d, const long long* count, const double* axc, double* mean_out, double* var_out, double* grad_out,
                             unsigned long long* lmax);
// the exact posterior of the generic kernel on an explicit fp64 list [N][d] (device), into caller-given arrays [q][N]
int launch_posterior_on_list(sbo_ctx* c, const double* pts, long long N, double* mean_out, double* var_out);
// guard.hip: the exact evaluator that the band of the resident posterior was measured against, on a list; K1b's band (device,
// enqueued behind the first posterior launch of a plan); a host-known band (K1t) or "none" (exact kernels)
int guard_exact_list(sbo_ctx* c, const double* pts, long long N, double* mean_out, double* var_out);
int guard_exact_grad_list(sbo_ctx* c, const double* pts, long long N, double* grad_out /* [q][d][N] */);
int guard_probe_reference(sbo_ctx* c, hipStream_t side, double** ref_m, double** ref_v, const sbo::ModelConst* mcp = nullptr);       // -> gb_probe: [q][P] each; K1b's own values follow at + 2 q P
// what K1b's band kernel needs to carry the axis bases' truncation through the posterior formula (guard.hip: k_gb_band)
struct GbAnalytic {
  const double* axis_eps;   // [2 q][2]: largest of the last four Chebyshev coefficients of the axis factors | sqrt(rc) x largest residual row norm of the basis
  const double* alpha;      // [q][a_ld]
  int a_ld, n;
  int rc[sbo::kMaxQ][2];         // Chebyshev degree of the axis factors
  const double* Xn;         // [n][dpad] normalised observations, and the normalised interval of each grid axis (a0, b0, a1, b1)
  int dpad, pad;
  double ab[4];
  const double* ref_g;      // [q][2][P] exact gradient components of the mean at the probe points (a lower bound on the Lipschitz keys)
};
int guard_band_from_probes(sbo_ctx* c, const double* pm, const double* pv, const double* ref_m, const double* ref_v, const double* tail, const GbAnalytic& an);
int guard_band_host(sbo_ctx* c, const double* dm, const double* dv, const double* rl, const double* parts = nullptr /* analytic dm | dv | probe dm | dv, kMaxQ each */);
int guard_audit_enqueue(sbo_ctx* c, int first_output);   // behind the posterior launch of a sweep (first_output 1: a lean sweep left the objective's values incomplete)
void guard_audit_harvest(sbo_ctx* c, bool wait);         // collect a finished audit's counts (wait: block until it has finished)
}  // namespace sbo

#define SBO_HIP(x)                                                   \
  do {                                                               \
    hipError_t e__ = (x);                                            \
    if (e__ != hipSuccess) return sbo::hip_fail(e__, #x);            \
  } while (0)
