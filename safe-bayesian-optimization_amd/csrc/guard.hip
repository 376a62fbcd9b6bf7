// guard.hip -- the guard band of the approximating posteriors (r04): exact evaluators on candidate lists and the band of K1b.
//
// K1b (bilinear.hip: two GEMMs on a Chebyshev core) and K1t (tensor.hip: interpolation from Chebyshev nodes) do not evaluate
// GP_inference (models/GP_Safe.py:310-352) candidate by candidate; their values differ from an exact fp64 evaluation by a
// truncation / interpolation error (measured 1e-13 .. 2e-12 normalised on the BASELINE models).  The sweeps decide masks and
// indices from those values, so every plan carries a BAND (device_common.hpp: GuardBand): per output the largest deviation of
// mean and var from the exact evaluator, measured at probe points when the plan is built, times a safety factor, plus the
// analytic truncation tail where there is one.  The set phase counts the decisions the band leaves open
// (SweepScalars::n_guard); a non-zero count sends the sweep through sets_recheck.inc.hpp, which re-evaluates the candidates
// concerned HERE and runs the set phase again.
//
// "Exact" is the evaluator the band was measured against, so that a re-evaluated value and its neighbours' approximate values
// are consistent to within the band:
//   K1b on a caller's invK (chol_async): the reference formula itself, (k^T invK) k with the matrix as given
//        (models/GP_Safe.py:341-343) -- k_ref_list, no factor of invK is needed (it may not exist yet);
//   otherwise (library Cholesky, K1t): the generic fp64 kernel K1 with the model's factor (launch_posterior_on_list).
#include <algorithm>
#include <cmath>
#include <cstring>
#include "internal.hpp"
#include "device_common.hpp"

namespace sbo {

constexpr int kRefPer = 16;         // candidates per workgroup of k_ref_list (a matrix element is read once for all of them: the
                                    // kernel is bound by the L2 traffic of re-reading the matrix per group of candidates)
constexpr double kInfBand = 1.0e300; // a probe that is not finite: everything is "inside the band"

__device__ __forceinline__ long long block_sum_i64(long long v) {        // valid in thread 0
  __shared__ long long sh[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  long long r = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) r += sh[w];
  return r;
}

// The reference formula on a list: mean_i = mp_i + k . alpha_i, var_i = max(0, sf2 - (k^T invK) k) with the caller's matrix as
// given (row-major [n][n]), k from the expanded distance (models/GP_Safe.py:112-119, 166, 326-347).  A workgroup takes kRefPer
// candidates, one output and a chunk of `ccols` matrix columns (a multiple of 64): the k vectors go to LDS; lane l of wave w
// accumulates (k^T invK)_j for column j = chunk base + 64 t + l over the rows of its quarter of the matrix -- the loads of a row
// are coalesced across the lanes and none depends on another (a wave per row with a reduction per row kept ONE row in flight: 440 us
// for 256 probes at n = 512) --, the four quarters meet in LDS, and the chunk's share of (k^T invK) k and of k . alpha is written
// as a partial; k_ref_finish sums the chunks.  pts == nullptr: the candidates are K1b's probe points (gb_probe_index).
template <int D>
__global__ __launch_bounds__(256) void k_ref_list(const ModelConst mc_val, const ModelConst* __restrict__ mcp /* non-null: the model's constants in
                                                  device memory (a plan replayed as a HIP graph: bilinear.hip) */, const CandSpec cs, const double* __restrict__ pts, long long N,
                                                  long long nlines, const double* __restrict__ As, const double* __restrict__ sqA,
                                                  const double* __restrict__ alpha, int ald, const double* __restrict__ Mx, size_t mstride,
                                                  int ccols, double* __restrict__ part /* [chunks][2][q][N] */) {
  extern __shared__ double kv[];                 // [kRefPer][npad] | [4][64][kRefPer] quarter sums
  const ModelConst& mc = mcp ? *mcp : mc_val;
  const int o = blockIdx.y, chunk = blockIdx.z, n = mc.n, npad = mc.npad, q = mc.q;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double* wq = kv + (size_t)kRefPer * npad;
  const long long c0 = (long long)blockIdx.x * kRefPer;
  for (int e = tid; e < kRefPer * npad; e += blockDim.x) {
    const int cc = e / npad, j = e % npad;
    const long long ci = c0 + cc;
    double v = 0.0;
    if (ci < N && j < n) {
      double x[D];
      if (pts) {
#pragma unroll
        for (int a = 0; a < D; ++a) x[a] = a < mc.d ? pts[(size_t)ci * mc.d + a] : 0.0;
      } else {
        cand_coords<D>(cs, gb_probe_index(cs, nlines, (int)ci), x);
      }
      double dot = 0.0, sqb = 0.0;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const double bq = a < mc.d ? ((x[a] - mc.X_mean[a]) / mc.X_std[a]) * mc.vinv[o][a] : 0.0;
        dot += As[((size_t)o * npad + j) * D + a] * bq;
        sqb += bq * bq;
      }
      v = mc.sf2[o] * exp(-0.5 * ((-2.0 * dot + sqA[(size_t)o * npad + j]) + sqb));
    }
    kv[e] = v;
  }
  __syncthreads();
  const double* Mo = Mx + (size_t)o * mstride;
  const int rq = (n + 3) / 4, i0 = wave * rq, i1 = (i0 + rq < n) ? i0 + rq : n;     // this wave's rows
  double quad[kRefPer], ms[kRefPer];
#pragma unroll
  for (int cc = 0; cc < kRefPer; ++cc) { quad[cc] = 0.0; ms[cc] = 0.0; }
  for (int jb = chunk * ccols; jb < (chunk + 1) * ccols && jb < n; jb += 64) {
    const int j = jb + lane;
    double w[kRefPer];
#pragma unroll
    for (int cc = 0; cc < kRefPer; ++cc) w[cc] = 0.0;
    if (j < n) {
      const double* col = Mo + j;
#pragma unroll 32
      for (int i = i0; i < i1; ++i) {                      // (a quarter of the rows: 32 loads in flight per lane)
        const double mij = col[(size_t)i * n];
#pragma unroll
        for (int cc = 0; cc < kRefPer; ++cc) w[cc] = fma(kv[cc * npad + i], mij, w[cc]);
      }
    }
    __syncthreads();                               // (the previous block of columns has been summed)
#pragma unroll
    for (int cc = 0; cc < kRefPer; ++cc) wq[(wave * 64 + lane) * kRefPer + cc] = w[cc];
    __syncthreads();
    if (wave == 0 && j < n) {
      const double aj = alpha[(size_t)o * ald + j];
#pragma unroll
      for (int cc = 0; cc < kRefPer; ++cc) {
        const double wj = (wq[lane * kRefPer + cc] + wq[(64 + lane) * kRefPer + cc]) + (wq[(128 + lane) * kRefPer + cc] + wq[(192 + lane) * kRefPer + cc]);
        quad[cc] = fma(wj, kv[cc * npad + j], quad[cc]);
        ms[cc] = fma(aj, kv[cc * npad + j], ms[cc]);
      }
    }
  }
  if (wave == 0) {
#pragma unroll
    for (int cc = 0; cc < kRefPer; ++cc) {
      const double qs = wave_sum(quad[cc]), m_ = wave_sum(ms[cc]);
      if (lane == 0 && c0 + cc < N) {
        part[(((size_t)chunk * 2 + 0) * q + o) * N + c0 + cc] = qs;
        part[(((size_t)chunk * 2 + 1) * q + o) * N + c0 + cc] = m_;
      }
    }
  }
}
__global__ __launch_bounds__(256) void k_ref_finish(const ModelConst mc_val, const ModelConst* __restrict__ mcp, long long N, int chunks,
                                                    const double* __restrict__ part, double* __restrict__ mean_out, double* __restrict__ var_out) {
  const ModelConst& mc = mcp ? *mcp : mc_val;
  const int q = mc.q;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < N * q; e += (long long)gridDim.x * blockDim.x) {
    const int o = (int)(e / N);
    const long long i = e % N;
    double q_ = 0.0, m_ = 0.0;
    for (int ch = 0; ch < chunks; ++ch) {
      q_ += part[(((size_t)ch * 2 + 0) * q + o) * N + i];
      m_ += part[(((size_t)ch * 2 + 1) * q + o) * N + i];
    }
    double var = mc.sf2[o] - q_;                                        // models/GP_Safe.py:343
    var = var > 0.0 ? var : 0.0;
    const double mean = mc.mp[o] + m_;                                  // :342
    mean_out[(size_t)o * N + i] = mean * mc.Y_std[o] + mc.Y_mean[o];    // :346
    var_out[(size_t)o * N + i] = var * (mc.Y_std[o] * mc.Y_std[o]);     // :347
  }
}

// coordinates of K1b's probe points (for the generic kernel, which takes explicit lists)
__global__ void k_gb_probe_pts(const CandSpec cs, long long nlines, double* __restrict__ pts) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= kGbProbes) return;
  double x[2];
  cand_coords<2>(cs, gb_probe_index(cs, nlines, p), x);
  pts[2 * p] = x[0];
  pts[2 * p + 1] = x[1];
}

// signed components of the gradient of the un-normalised mean at listed points (the analytic form of jax.grad(self.mean),
// models/SafeOpt.py:68-71): out[(o d + a) N + i]; a thread per point
template <int D>
__global__ __launch_bounds__(256) void k_grad_list(const ModelConst mc, const double* __restrict__ pts, long long N,
                                                   const double* __restrict__ As, const double* __restrict__ sqA,
                                                   const double* __restrict__ alpha, const double* __restrict__ Xn, double* __restrict__ out) {
  const int n = mc.n, npad = mc.npad;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
    double xn[D];
#pragma unroll
    for (int a = 0; a < D; ++a) xn[a] = a < mc.d ? (pts[(size_t)i * mc.d + a] - mc.X_mean[a]) / mc.X_std[a] : 0.0;
    for (int o = 0; o < mc.q; ++o) {
      double bq[D], sqb = 0.0;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        bq[a] = a < mc.d ? xn[a] * mc.vinv[o][a] : 0.0;
        sqb += bq[a] * bq[a];
      }
      double s0 = 0.0, sa[D];
#pragma unroll
      for (int a = 0; a < D; ++a) sa[a] = 0.0;
      for (int j = 0; j < n; ++j) {
        double dot = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) dot += As[((size_t)o * npad + j) * D + a] * bq[a];
        const double w = alpha[(size_t)o * npad + j] * (mc.sf2[o] * exp(-0.5 * ((-2.0 * dot + sqA[(size_t)o * npad + j]) + sqb)));
        s0 += w;
#pragma unroll
        for (int a = 0; a < D; ++a) sa[a] += w * Xn[(size_t)j * D + a];
      }
#pragma unroll
      for (int a = 0; a < D; ++a)
        if (a < mc.d) out[((size_t)o * mc.d + a) * N + i] = mc.Y_std[o] * (sa[a] - xn[a] * s0) * mc.inv_ell[o][a] * mc.X_rstd[a];
    }
  }
}

// the same for a SHORT list (the probe points of a plan): a wave per point, lanes over the observations (a thread per point walks
// q n exponentials: 0.25 ms for 144 points at n = 512)
template <int D>
__global__ __launch_bounds__(256) void k_grad_list_w(const ModelConst mc_val, const ModelConst* __restrict__ mcp, const double* __restrict__ pts, long long N,
                                                     const double* __restrict__ As, const double* __restrict__ sqA,
                                                     const double* __restrict__ alpha, const double* __restrict__ Xn, double* __restrict__ out) {
  const ModelConst& mc = mcp ? *mcp : mc_val;
  const int n = mc.n, npad = mc.npad, lane = threadIdx.x & 63;
  const long long i = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  double xn[D];
#pragma unroll
  for (int a = 0; a < D; ++a) xn[a] = a < mc.d ? (pts[(size_t)i * mc.d + a] - mc.X_mean[a]) / mc.X_std[a] : 0.0;
  for (int o = 0; o < mc.q; ++o) {
    double bq[D], sqb = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      bq[a] = a < mc.d ? xn[a] * mc.vinv[o][a] : 0.0;
      sqb += bq[a] * bq[a];
    }
    double s0 = 0.0, sa[D];
#pragma unroll
    for (int a = 0; a < D; ++a) sa[a] = 0.0;
    for (int j = lane; j < n; j += 64) {
      double dot = 0.0;
#pragma unroll
      for (int a = 0; a < D; ++a) dot += As[((size_t)o * npad + j) * D + a] * bq[a];
      const double w = alpha[(size_t)o * npad + j] * (mc.sf2[o] * exp(-0.5 * ((-2.0 * dot + sqA[(size_t)o * npad + j]) + sqb)));
      s0 += w;
#pragma unroll
      for (int a = 0; a < D; ++a) sa[a] += w * Xn[(size_t)j * D + a];
    }
    s0 = wave_sum(s0);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      const double s = wave_sum(sa[a]);
      if (lane == 0 && a < mc.d) out[((size_t)o * mc.d + a) * N + i] = mc.Y_std[o] * (s - xn[a] * s0) * mc.inv_ell[o][a] * mc.X_rstd[a];
    }
  }
}

// K1b's band from its probes: one workgroup.  ref_m / ref_v [q][P]: the exact evaluator at the probe points; pm / pv: K1b's own
// representation evaluated there (bilinear.hip: k_gb_probe_k1b -- the sums the two GEMMs form, in another order); tail[o]: sum of
// the Chebyshev coefficients of the variance's quadratic form that the kernels do not run (normalised variance units;
// k_cheb_trunc).  rl: K1b's Lipschitz keys come from the same reduced-basis mean whose values are probed here -- the gradient sums
// are exact GEMMs on it --; 1e-9 relative is three decades above what the parity tests measure against K1g (1e-12).
// K1b's band (r05).  Analytic part: K1b replaces the axis factors f_aj(x) = exp(-(x - x_j)^2 / (2 ell_a)) of the kernel vector
// k_j(x) = sf2 f_0j(x_0) f_1j(x_1) by members of a reduced basis of degree-(rc - 1) Chebyshev series: uniformly on the axis
//   |f~ - f| <= eps_a = kGbTailFactor t4_a  (Chebyshev cut: the interpolation error is at most twice the dropped tail, which continues
//                                            the last four coefficients -- t4: their largest -- at a ratio <= 0.8)
//                     + res_a                (rank cut: sqrt(rc) x the largest residual row norm the pivot loop left, k_bl_basis)
//                     + 4 rc eps             (evaluating the series),
// so every k_j moves by at most eta = sf2 (eps_0 + eps_1 + eps_0 eps_1) and, e = k~ - k,
//   |mean~ - mean| = ys |e . alpha|                    <= ys eta ||alpha||_1
//   |var~  - var | = ys^2 |2 e . (A k) + e . A e|      <= ys^2 (2 eta sqrt(n) sqrt(sf2 / sn2) + n eta^2 / sn2)
//   |grad_a~ - grad_a| = ys / (ell_a X_std_a) |sum_j alpha_j (xn_ja - x_a) e_j|  <= ys / (ell_a X_std_a) eta sum_j |alpha_j| (|xn_ja| + max |x_a|)
//   (the gradient sums are bilinear forms of the same factors, not derivatives of the approximant); relative to the Lipschitz key
//   through the largest exact component at the probe points, which the key is not smaller than
// with A = invK: ||A k||_2^2 = k^T A^2 k <= ||A||_2 k^T A k <= sf2 / sn2, because K = sf2 exp(..) + sn2 I has no eigenvalue below
// sn2 and the exact variance is not negative (models/GP_Safe.py:226-232, 343).  [A caller's matrix that is NOT that inverse is
// outside this bound -- the probes below are what notices.]  Plus the dropped coefficients of the 2-D core (tail) and a rounding
// floor, plus the measured rounding level of the plan (kGbSafety x the probes' largest deviation); the check: see GuardBand.
constexpr double kGbTailFactor = 8.0;
__global__ __launch_bounds__(256) void k_gb_band(const ModelConst mc, const double* __restrict__ pm, const double* __restrict__ pv,
                                                 const double* __restrict__ ref_m, const double* __restrict__ ref_v,
                                                 const double* __restrict__ tail, const GbAnalytic an, GuardBand* gb,
                                                 GuardBand* gb_mirror /* pinned host copy for sbo_profile_get (no read-back per plan) */) {
  __shared__ double sh[4][8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int o = 0; o < mc.q; ++o) {
    double em = 0.0, ev = 0.0, am = 0.0, av = 0.0, a1 = 0.0, ax0 = 0.0, ax1 = 0.0, gmax = 0.0;
    bool bad = false;
    for (int p = tid; p < kGbProbes; p += blockDim.x) {
      const double m = pm[(size_t)o * kGbProbes + p], v = pv[(size_t)o * kGbProbes + p];
      const double rm = ref_m[(size_t)o * kGbProbes + p], rv = ref_v[(size_t)o * kGbProbes + p];
      const double dm = fabs(m - rm), dv = fabs(v - rv);
      bad = bad || !(dm < kInfBand) || !(dv < kInfBand);
      em = fmax(em, dm);
      ev = fmax(ev, dv);
      am = fmax(am, fabs(rm));
      av = fmax(av, fabs(rv));
    }
    if (an.alpha)
      for (int j = tid; j < an.n; j += blockDim.x) {
        const double aj = fabs(an.alpha[(size_t)o * an.a_ld + j]);
        a1 += aj;
        if (an.Xn) { ax0 += aj * fabs(an.Xn[(size_t)j * an.dpad]); ax1 += aj * fabs(an.Xn[(size_t)j * an.dpad + 1]); }
      }
    if (an.ref_g)
      for (int p = tid; p < 2 * kGbProbes; p += blockDim.x) gmax = fmax(gmax, fabs(an.ref_g[(size_t)o * 2 * kGbProbes + p]));
    if (bad) { em = kInfBand; ev = kInfBand; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      em = fmax(em, __shfl_xor(em, off));
      ev = fmax(ev, __shfl_xor(ev, off));
      am = fmax(am, __shfl_xor(am, off));
      av = fmax(av, __shfl_xor(av, off));
      a1 += __shfl_xor(a1, off);
      ax0 += __shfl_xor(ax0, off);
      ax1 += __shfl_xor(ax1, off);
      gmax = fmax(gmax, __shfl_xor(gmax, off));
    }
    __syncthreads();
    if (lane == 0) { sh[wave][0] = em; sh[wave][1] = ev; sh[wave][2] = am; sh[wave][3] = av; sh[wave][4] = a1; sh[wave][5] = ax0; sh[wave][6] = ax1; sh[wave][7] = gmax; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w) {
        em = fmax(em, sh[w][0]); ev = fmax(ev, sh[w][1]); am = fmax(am, sh[w][2]); av = fmax(av, sh[w][3]);
        a1 += sh[w][4]; ax0 += sh[w][5]; ax1 += sh[w][6]; gmax = fmax(gmax, sh[w][7]);
      }
      const double ys = mc.Y_std[o], eps = 2.220446049250313e-16, sf2 = mc.sf2[o], sn2 = mc.sn2[o];
      double an_m = 0.0, an_v = (tail ? tail[o] : 0.0) * ys * ys, an_g = 0.0;
      if (an.axis_eps) {
        double ea[2];
        for (int a = 0; a < 2; ++a)
          ea[a] = kGbTailFactor * an.axis_eps[2 * (2 * o + a)] + an.axis_eps[2 * (2 * o + a) + 1] + 4.0 * an.rc[o][a] * eps;
        const double eta = sf2 * (ea[0] + ea[1] + ea[0] * ea[1]);
        an_m = ys * eta * a1;
        an_v += ys * ys * (2.0 * eta * sqrt((double)an.n * sf2 / sn2) + (double)an.n * eta * eta / sn2);
        const double g0 = ys * mc.inv_ell[o][0] * mc.X_rstd[0] * eta * (ax0 + fmax(fabs(an.ab[0]), fabs(an.ab[1])) * a1);
        const double g1 = ys * mc.inv_ell[o][1] * mc.X_rstd[1] * eta * (ax1 + fmax(fabs(an.ab[2]), fabs(an.ab[3])) * a1);
        an_g = fmax(g0, g1);
      }
      const double fl_m = 64.0 * eps * fmax(am, fabs(mc.Y_mean[o]) + ys), fl_v = 64.0 * eps * fmax(av, sf2 * ys * ys);
      gb->an_m[o] = an_m; gb->an_v[o] = an_v; gb->pr_m[o] = em; gb->pr_v[o] = ev;
      const bool inf = !(em < kInfBand) || !(an_m < kInfBand) || !(an_v < kInfBand);
      const bool distrust = em > an_m + gb_round_mean(an.n, sf2, a1, ys) || ev > an_v + gb_round_var(an.n, sf2, sn2, ys);
      gb->dm[o] = (inf || distrust) ? kInfBand : an_m + kGbSafety * em + fl_m;
      gb->dv[o] = (inf || distrust) ? kInfBand : an_v + kGbSafety * ev + fl_v;
      gb->rl[o] = 1e-9 + (an_g > 0.0 ? (gmax > 0.0 ? an_g / gmax : 1e-3) : 0.0);
      if (gb_mirror) {
        gb_mirror->dm[o] = gb->dm[o]; gb_mirror->dv[o] = gb->dv[o]; gb_mirror->rl[o] = gb->rl[o];
        gb_mirror->an_m[o] = an_m; gb_mirror->an_v[o] = an_v; gb_mirror->pr_m[o] = em; gb_mirror->pr_v[o] = ev;
      }
    }
    __syncthreads();
  }
}

// ---- host ----------------------------------------------------------------------------------------------------------------
// can the reference formula run with the caller's matrix (no factor needed)?
static bool ref_direct(const sbo_ctx* c) {
  return c->mc.factor == SBO_FACTOR_INVK && c->chol_async && c->invk_w_valid && c->invk_plain != nullptr && c->dtype == SBO_F64 && !c->is_shadow &&
         c->mc.npad <= kGuardRefMaxNpad;
}

template <int D>
static int launch_ref(sbo_ctx* c, hipStream_t st, const double* pts, long long N, long long nlines, double* mean_out, double* var_out,
                      const ModelConst* mcp = nullptr, DevBuf* part_buf = nullptr /* the audit's own scratch (it runs beside the main stream) */) {
  const ModelConst& mc = c->mc;
  const int q = mc.q;
  const long long groups = (N + kRefPer - 1) / kRefPer;
  // column chunks: as many as it takes to put ~2 workgroups on every CU (a short list is latency-bound: 128 dependent-free loads
  // per lane and chunk of 64 columns at n = 512), one chunk for long lists
  const int nblk = (mc.n + 63) / 64;
  int chunks = (int)std::max<long long>(1, std::min<long long>(nblk, (2ll * c->n_cu) / std::max<long long>(1, groups * q)));
  const int ccols = ((nblk + chunks - 1) / chunks) * 64;
  chunks = (mc.n + ccols - 1) / ccols;
  int rc;
  DevBuf& pb = part_buf ? *part_buf : c->gb_part;
  if ((rc = ensure(pb, sizeof(double) * 2 * (size_t)chunks * q * (size_t)N))) return rc;
  const size_t lds = sizeof(double) * ((size_t)kRefPer * mc.npad + 256 * kRefPer);
  auto kern = k_ref_list<D>;
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)groups, (unsigned)q, (unsigned)chunks), dim3(256), lds, st, mc, mcp, c->cs, pts, N, nlines,
                     (const double*)c->As.p, (const double*)c->sqA.p, (const double*)c->alpha64.p, c->a_ld, c->invk_plain,
                     (size_t)mc.n * mc.n, ccols, (double*)pb.p);
  hipLaunchKernelGGL(k_ref_finish, dim3((unsigned)std::max<long long>(1, std::min<long long>((N * q + 255) / 256, 1024))), dim3(256), 0, st,
                     mc, mcp, N, chunks, (const double*)pb.p, mean_out, var_out);
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

int guard_exact_list(sbo_ctx* c, const double* pts, long long N, double* mean_out, double* var_out) {
  if (N <= 0) return SBO_OK;
  if ((c->last_k1 == 4 || c->last_k1 == 6) && ref_direct(c)) {
    switch (c->mc.dpad) {
      case 2: return launch_ref<2>(c, c->stream, pts, N, 0, mean_out, var_out);
      case 4: return launch_ref<4>(c, c->stream, pts, N, 0, mean_out, var_out);
      default: return launch_ref<8>(c, c->stream, pts, N, 0, mean_out, var_out);
    }
  }
  return launch_posterior_on_list(c, pts, N, mean_out, var_out);
}

int guard_exact_grad_list(sbo_ctx* c, const double* pts, long long N, double* grad_out) {
  if (N <= 0) return SBO_OK;
  if (c->dtype != SBO_F64) return fail(SBO_E_UNSUPPORTED, "internal: gradient list is an fp64 path");
  const unsigned nb = (unsigned)std::max<long long>(1, std::min<long long>((N + 255) / 256, 4096));
  switch (c->mc.dpad) {
    case 2: hipLaunchKernelGGL(k_grad_list<2>, dim3(nb), dim3(256), 0, c->stream, c->mc, pts, N, (const double*)c->As.p, (const double*)c->sqA.p, (const double*)c->alpha.p, (const double*)c->Xn.p, grad_out); break;
    case 4: hipLaunchKernelGGL(k_grad_list<4>, dim3(nb), dim3(256), 0, c->stream, c->mc, pts, N, (const double*)c->As.p, (const double*)c->sqA.p, (const double*)c->alpha.p, (const double*)c->Xn.p, grad_out); break;
    default: hipLaunchKernelGGL(k_grad_list<8>, dim3(nb), dim3(256), 0, c->stream, c->mc, pts, N, (const double*)c->As.p, (const double*)c->sqA.p, (const double*)c->alpha.p, (const double*)c->Xn.p, grad_out); break;
  }
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

// a band the host knows (K1t's probe), or all-zero (nullptr arguments)
int guard_band_host(sbo_ctx* c, const double* dm, const double* dv, const double* rl, const double* parts) {
  int rc;
  if ((rc = ensure(c->gb, sizeof(GuardBand)))) return rc;
  c->gb_host_valid = false;
  c->gb_mirrored = false;
  GuardBand hb;
  memset(&hb, 0, sizeof(hb));
  for (int o = 0; o < c->mc.q && dm; ++o) {
    hb.dm[o] = dm[o]; hb.dv[o] = dv[o]; hb.rl[o] = rl[o];
    if (parts) { hb.an_m[o] = parts[o]; hb.an_v[o] = parts[kMaxQ + o]; hb.pr_m[o] = parts[2 * kMaxQ + o]; hb.pr_v[o] = parts[3 * kMaxQ + o]; }
  }
  // (pageable source: the runtime stages it before the call returns)
  SBO_HIP(hipMemcpyAsync(c->gb.p, &hb, sizeof(hb), hipMemcpyHostToDevice, c->stream));
  for (int o = 0; o < SBO_MAX_Q; ++o) {
    c->gb_host[o] = hb.dm[o]; c->gb_host[SBO_MAX_Q + o] = hb.dv[o]; c->gb_host[2 * SBO_MAX_Q + o] = hb.rl[o];
    c->gb_host[3 * SBO_MAX_Q + o] = hb.an_m[o]; c->gb_host[4 * SBO_MAX_Q + o] = hb.an_v[o]; c->gb_host[5 * SBO_MAX_Q + o] = hb.pr_m[o]; c->gb_host[6 * SBO_MAX_Q + o] = hb.pr_v[o];
  }
  c->gb_host_valid = true;
  return SBO_OK;
}

// K1b's band, built with the plan (bilinear_setup): the exact evaluator at the probe points -- on `side`, the stream the plan's
// axis tables are made on, beside the GEMM chain of the Chebyshev core -- and, once K1b's own values at the probes are there
// (pm / pv, bilinear.hip), the band from the deviations.  Everything on the device, in stream order: nothing waits for the host,
// and the band is in place before the plan's first posterior launch (whose fused classification reads it).
bool guard_reference_is_direct(const sbo_ctx* c) { return ref_direct(c); }
int guard_probe_reference(sbo_ctx* c, hipStream_t side, double** ref_m, double** ref_v, const ModelConst* mcp) {
  const int q = c->mc.q;
  const long long nlines = c->cs.n_local / c->cs.count[0];
  int rc;
  if ((rc = ensure(c->gb, sizeof(GuardBand)))) return rc;
  c->gb_host_valid = false;
  // (ref_m | ref_v | the plan's own values pm | pv | the probe list [P][2] | exact gradient components [q][2][P])
  if ((rc = ensure(c->gb_probe, sizeof(double) * (6 * (size_t)q + 2) * kGbProbes))) return rc;
  *ref_m = (double*)c->gb_probe.p;
  *ref_v = *ref_m + (size_t)q * kGbProbes;
  if (ref_direct(c)) return launch_ref<2>(c, side, nullptr, kGbProbes, nlines, *ref_m, *ref_v, mcp);
  // (library Cholesky, or a caller's invK without chol_async: the factor is there, the generic kernel takes the probe list --
  // on the main stream: it is the context's launcher)
  double* ppts = *ref_m + 4 * (size_t)q * kGbProbes;
  hipLaunchKernelGGL(k_gb_probe_pts, dim3(1), dim3(kGbProbes), 0, c->stream, c->cs, nlines, ppts);
  return launch_posterior_on_list(c, ppts, kGbProbes, *ref_m, *ref_v);
}
// the probe points as a list [P][d] and the exact gradient components of the mean there ([q][d][P]), on `st` (2-D grids)
int guard_probe_gradients(sbo_ctx* c, hipStream_t st, double* ppts, double* grad_out, const ModelConst* mcp) {
  if (c->dtype != SBO_F64 || c->mc.dpad != 2) return fail(SBO_E_UNSUPPORTED, "internal: probe gradients are an fp64 2-D path");
  hipLaunchKernelGGL(k_gb_probe_pts, dim3(1), dim3(kGbProbes), 0, st, c->cs, c->cs.n_local / c->cs.count[0], ppts);
  hipLaunchKernelGGL(k_grad_list_w<2>, dim3((kGbProbes + 3) / 4), dim3(256), 0, st, c->mc, mcp, (const double*)ppts, (long long)kGbProbes,
                     (const double*)c->As.p, (const double*)c->sqA.p, (const double*)c->alpha.p, (const double*)c->Xn.p, grad_out);
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}
int guard_band_from_probes(sbo_ctx* c, const double* pm, const double* pv, const double* ref_m, const double* ref_v, const double* tail, const GbAnalytic& an) {
  hipLaunchKernelGGL(k_gb_band, dim3(1), dim3(256), 0, c->stream, c->mc, pm, pv, ref_m, ref_v, tail, an, (GuardBand*)c->gb.p,
                     (GuardBand*)(c->h_back + kGbMirrorOffset));
  c->gb_mirrored = true;
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

// ---- standing audit of the guard band (r05) --------------------------------------------------------------------------------
// sample k of a sweep: candidate (offset + k stride) mod N -- a stride coprime to the grid sizes in use scatters the sample over the
// grid, the offset moves on with every sweep --; its coordinates for the reference formula and the values the posterior kernel stored
template <int D>
__global__ __launch_bounds__(256) void k_audit_pick(const CandSpec cs, int P, unsigned long long offset, unsigned long long stride, int q,
                                                    const double* __restrict__ mean, const double* __restrict__ var, double* __restrict__ pts,
                                                    double* __restrict__ apx /* [2][q][P] */) {
  const long long N = cs.n_local;
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < P; k += gridDim.x * blockDim.x) {
    const long long g = (long long)((offset + (unsigned long long)k * stride) % (unsigned long long)N);
    double x[D];
    cand_coords<D>(cs, g, x);
    for (int a = 0; a < cs.d; ++a) pts[(size_t)k * cs.d + a] = x[a];
    for (int o = 0; o < q; ++o) {
      apx[(size_t)o * P + k] = mean[(size_t)o * N + g];
      apx[(size_t)(q + o) * P + k] = var[(size_t)o * N + g];
    }
  }
}
// cnt: [0] violations, [1] samples (value pairs compared), [2] bit pattern of the largest deviation in units of the band
__global__ __launch_bounds__(256) void k_audit_compare(int P, int q, int o_first, const double* __restrict__ apx, const double* __restrict__ ref_m,
                                                       const double* __restrict__ ref_v, const GuardBand* __restrict__ gb,
                                                       unsigned long long* __restrict__ cnt, double scale /* 1, or the test hook's factor on the band */) {
  long long viol = 0, smp = 0;
  double worst = 0.0;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < P * (q - o_first); e += gridDim.x * blockDim.x) {
    const int o = o_first + e / P, k = e % P;
    const double dm = fabs(apx[(size_t)o * P + k] - ref_m[(size_t)o * P + k]), dv = fabs(apx[(size_t)(q + o) * P + k] - ref_v[(size_t)o * P + k]);
    const double bm = gb->dm[o] * scale, bv = gb->dv[o] * scale;
    const double rm = dm / bm, rv = dv / bv;
    const bool bad = !(dm <= bm) || !(dv <= bv);                            // (NaN counts)
    viol += bad;
    ++smp;
    const double r = rm > rv ? rm : rv;
    worst = r > worst ? r : (r != r ? 1e300 : worst);
  }
  viol = block_sum_i64(viol);
  smp = block_sum_i64(smp);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) worst = fmax(worst, __shfl_xor(worst, off));
  if ((threadIdx.x & 63) == 0 && worst > 0.0) atomicMax(&cnt[2], (unsigned long long)__double_as_longlong(worst));
  if (threadIdx.x == 0) {
    if (viol) atomicAdd(&cnt[0], (unsigned long long)viol);
    atomicAdd(&cnt[1], (unsigned long long)smp);
  }
}

void guard_audit_harvest(sbo_ctx* c, bool wait) {
  if (!c->audit_pending) return;
  if (wait) (void)hipEventSynchronize(c->ev_audit[1]);
  else if (hipEventQuery(c->ev_audit[1]) != hipSuccess) return;
  const unsigned long long* hc = (const unsigned long long*)(c->h_back + 7168);
  c->audit_violations += (long long)hc[0];
  c->audit_samples += (long long)hc[1];
  double w;
  memcpy(&w, &hc[2], 8);
  c->audit_worst = std::max(c->audit_worst, w);
  c->audit_pending = false;
}

int guard_audit_enqueue(sbo_ctx* c, int first_output) {
  if (c->guard_audit <= 0 || !c->gb_active || !c->guard_band || c->is_shadow || !(c->last_k1 == 4 || c->last_k1 == 6) || !ref_direct(c) || c->mc.dpad != 2 ||
      c->cs.n_local <= 0 || !c->stream_audit || first_output >= c->mc.q)
    return SBO_OK;
  // (the audit shares the card with the sweep it follows -- ~35 us of a config-H sweep's set phase for 1024 samples at n = 512 --, so
  // one sweep in guard_audit_every carries one)
  if (c->audit_tick++ % std::max(1, c->guard_audit_every)) return SBO_OK;
  guard_audit_harvest(c, false);
  if (c->audit_pending) return SBO_OK;                 // (the previous audit is still running: this sweep's is skipped, none queues up)
  const int P = c->guard_audit, q = c->mc.q;
  int rc;
  if ((rc = ensure(c->audit_pts, sizeof(double) * (size_t)P * 2)) || (rc = ensure(c->audit_val, sizeof(double) * (size_t)P * q * 4)) ||
      (rc = ensure(c->audit_cnt, 64)))
    return rc;
  double* apx = (double*)c->audit_val.p;
  double* ref_m = apx + (size_t)2 * q * P;
  double* ref_v = ref_m + (size_t)q * P;
  hipStream_t st = c->stream_audit;
  // the sample is taken behind the posterior (its stop event ev[1] -- no record of its own on the main stream: that would be a bubble
  // in the sweep) ...
  SBO_HIP(hipStreamWaitEvent(st, c->ev[1], 0));
  SBO_HIP(hipMemsetAsync(c->audit_cnt.p, 0, 64, st));
  const unsigned long long stride = 1000003ull;
  hipLaunchKernelGGL(k_audit_pick<2>, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, c->cs, P, c->audit_offset, stride, q, (const double*)c->mean.p,
                     (const double*)c->var.p, (double*)c->audit_pts.p, apx);
  c->audit_offset += (unsigned long long)P * stride + 17ull;
  // ... and before anything overwrites mean / var again (the next posterior launch waits for this event)
  SBO_HIP(hipEventRecord(c->ev_audit[0], st));
  if ((rc = launch_ref<2>(c, st, (const double*)c->audit_pts.p, P, 0, ref_m, ref_v, nullptr, &c->audit_part))) return rc;
  hipLaunchKernelGGL(k_audit_compare, dim3(8), dim3(256), 0, st, P, q, first_output, (const double*)apx, (const double*)ref_m, (const double*)ref_v,
                     (const GuardBand*)c->gb.p, (unsigned long long*)c->audit_cnt.p, c->audit_scale);
  SBO_HIP(hipMemcpyAsync(c->h_back + 7168, c->audit_cnt.p, 24, hipMemcpyDeviceToHost, st));
  SBO_HIP(hipEventRecord(c->ev_audit[1], st));
  SBO_HIP(hipGetLastError());
  c->audit_pending = true;
  return SBO_OK;
}

}  // namespace sbo
