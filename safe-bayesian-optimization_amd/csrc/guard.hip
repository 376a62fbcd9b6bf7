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
//   K1b on a caller's invK (chol_async): the reference formula itself, k^T invK k with the matrix as given
//        (models/GP_Safe.py:341-343) -- k_ref_list, no factor of invK is needed (it may not exist yet);
//   otherwise (library Cholesky, K1t): the generic fp64 kernel K1 with the model's factor (launch_posterior_on_list).
#include <algorithm>
#include <cmath>
#include <cstring>
#include "internal.hpp"
#include "device_common.hpp"

namespace sbo {

constexpr int kRefPer = 4;          // candidates per workgroup of k_ref_list (the matrix is read once for all of them)
constexpr int kGbProbe1 = 16;       // K1b: probe positions per axis (a 16 x 16 tensor of near-Chebyshev grid positions)
constexpr int kGbProbes = kGbProbe1 * kGbProbe1;
constexpr double kGbSafety = 16.0;  // band = safety x the largest probe deviation (+ truncation tail + rounding floor)
constexpr double kInfBand = 1.0e300; // a probe that is not finite: everything is "inside the band"

// local index of K1b probe p on the resident grid: positions nearest to the Chebyshev extrema of each axis (ends included --
// a polynomial surrogate errs most there)
__device__ __forceinline__ long long gb_probe_index(const CandSpec& cs, long long nlines, int p) {
  const int i0 = p % kGbProbe1, i1 = p / kGbProbe1;
  const long long c0 = cs.count[0];
  const long long x0 = (long long)llrint(0.5 * (1.0 - cospi((double)i0 / (double)(kGbProbe1 - 1))) * (double)(c0 - 1));
  const long long x1 = (long long)llrint(0.5 * (1.0 - cospi((double)i1 / (double)(kGbProbe1 - 1))) * (double)(nlines - 1));
  return x1 * c0 + x0;
}

// The reference formula on a list: mean_i = mp_i + k . alpha_i, var_i = max(0, sf2 - quad) with quad = k^T invK k (MODE 0: the
// caller's matrix as given, row-major [n][ld]) or ||M k||^2 (MODE 1: the lower-triangular factor, M^T M = invK), k from the
// expanded distance (models/GP_Safe.py:112-119, 166, 326-347).  One workgroup per (kRefPer candidates, output): the k vectors in
// LDS, a wave per matrix row with its lanes along the row (coalesced), kRefPer dot products per row read.
// pts == nullptr: the candidates are K1b's probe points of the resident grid (gb_probe_index).
template <int D, int MODE>
__global__ __launch_bounds__(256) void k_ref_list(const ModelConst mc, const CandSpec cs, const double* __restrict__ pts, long long N,
                                                  long long nlines, const double* __restrict__ As, const double* __restrict__ sqA,
                                                  const double* __restrict__ alpha, int ald, const double* __restrict__ Mx, size_t mstride,
                                                  int ld, double* __restrict__ mean_out, double* __restrict__ var_out) {
  extern __shared__ double kv[];                 // [kRefPer][npad]
  __shared__ double red[4][kRefPer];
  __shared__ double msum[4][kRefPer];
  const int o = blockIdx.y, n = mc.n, npad = mc.npad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long c0 = (long long)blockIdx.x * kRefPer;
  // k vectors
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
  double quad[kRefPer], ms[kRefPer];
#pragma unroll
  for (int cc = 0; cc < kRefPer; ++cc) { quad[cc] = 0.0; ms[cc] = 0.0; }
  for (int i = wave; i < n; i += 4) {
    const double* row = Mo + (size_t)i * ld;
    const int jend = MODE == 1 ? i + 1 : n;
    double s[kRefPer];
#pragma unroll
    for (int cc = 0; cc < kRefPer; ++cc) s[cc] = 0.0;
    for (int j = lane; j < jend; j += 64) {
      const double mij = row[j];
#pragma unroll
      for (int cc = 0; cc < kRefPer; ++cc) s[cc] = fma(mij, kv[cc * npad + j], s[cc]);
    }
#pragma unroll
    for (int cc = 0; cc < kRefPer; ++cc) {
      const double w = wave_sum(s[cc]);
      quad[cc] += MODE == 1 ? w * w : kv[cc * npad + i] * w;
    }
  }
  // mean: k . alpha (wave 0's lanes along j), then the waves' quad shares
  if (wave == 0) {
    double s[kRefPer];
#pragma unroll
    for (int cc = 0; cc < kRefPer; ++cc) s[cc] = 0.0;
    for (int j = lane; j < n; j += 64) {
      const double aj = alpha[(size_t)o * ald + j];
#pragma unroll
      for (int cc = 0; cc < kRefPer; ++cc) s[cc] = fma(aj, kv[cc * npad + j], s[cc]);
    }
#pragma unroll
    for (int cc = 0; cc < kRefPer; ++cc) ms[cc] = wave_sum(s[cc]);
  }
  if (lane == 0) {
#pragma unroll
    for (int cc = 0; cc < kRefPer; ++cc) { red[wave][cc] = quad[cc]; msum[wave][cc] = ms[cc]; }
  }
  __syncthreads();
  if (tid < kRefPer && c0 + tid < N) {
    const double q_ = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    double var = mc.sf2[o] - q_;                                        // models/GP_Safe.py:343
    var = var > 0.0 ? var : 0.0;
    const double mean = mc.mp[o] + msum[0][tid];                        // :342
    mean_out[(size_t)o * N + c0 + tid] = mean * mc.Y_std[o] + mc.Y_mean[o];      // :346
    var_out[(size_t)o * N + c0 + tid] = var * (mc.Y_std[o] * mc.Y_std[o]);       // :347
  }
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

// K1b's band from its probes: one workgroup.  ref_m / ref_v [q][P]: the exact evaluator at the probe points; mean / var: the
// posterior K1b has just written; tail[o]: sum of the Chebyshev coefficients of the variance's quadratic form that the kernels
// do not run (normalised variance units; k_cheb_trunc).  rl: K1b's Lipschitz keys come from the same reduced-basis mean whose
// values are probed here -- the gradient sums are exact GEMMs on it --; 1e-9 relative is three decades above what the parity
// tests measure against K1g (1e-12).
__global__ __launch_bounds__(256) void k_gb_band(const ModelConst mc, const CandSpec cs, long long nlines, const double* __restrict__ mean,
                                                 const double* __restrict__ var, const double* __restrict__ ref_m,
                                                 const double* __restrict__ ref_v, const double* __restrict__ tail, GuardBand* gb) {
  __shared__ double sh[4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long n = cs.n_local;
  for (int o = 0; o < mc.q; ++o) {
    double em = 0.0, ev = 0.0, am = 0.0, av = 0.0;
    bool bad = false;
    for (int p = tid; p < kGbProbes; p += blockDim.x) {
      const long long g = gb_probe_index(cs, nlines, p);
      const double m = mean[(size_t)o * n + g], v = var[(size_t)o * n + g];
      const double rm = ref_m[(size_t)o * kGbProbes + p], rv = ref_v[(size_t)o * kGbProbes + p];
      const double dm = fabs(m - rm), dv = fabs(v - rv);
      bad = bad || !(dm < kInfBand) || !(dv < kInfBand);
      em = fmax(em, dm);
      ev = fmax(ev, dv);
      am = fmax(am, fabs(rm));
      av = fmax(av, fabs(rv));
    }
    if (bad) { em = kInfBand; ev = kInfBand; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      em = fmax(em, __shfl_xor(em, off));
      ev = fmax(ev, __shfl_xor(ev, off));
      am = fmax(am, __shfl_xor(am, off));
      av = fmax(av, __shfl_xor(av, off));
    }
    __syncthreads();
    if (lane == 0) { sh[wave][0] = em; sh[wave][1] = ev; sh[wave][2] = am; sh[wave][3] = av; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w) { em = fmax(em, sh[w][0]); ev = fmax(ev, sh[w][1]); am = fmax(am, sh[w][2]); av = fmax(av, sh[w][3]); }
      const double ys = mc.Y_std[o], eps = 2.220446049250313e-16;
      gb->dm[o] = kGbSafety * em + 64.0 * eps * fmax(am, fabs(mc.Y_mean[o]) + ys);
      gb->dv[o] = kGbSafety * ev + (tail ? tail[o] : 0.0) * ys * ys + 64.0 * eps * fmax(av, mc.sf2[o] * ys * ys);
      gb->rl[o] = 1e-9;
    }
  }
}

// ---- host ----------------------------------------------------------------------------------------------------------------
// can the reference formula run with the caller's matrix (no factor needed)?
static bool ref_direct(const sbo_ctx* c) {
  return c->mc.factor == SBO_FACTOR_INVK && c->chol_async && c->invk_w_valid && c->invk_plain != nullptr && c->dtype == SBO_F64 && !c->is_shadow;
}

template <int D>
static int launch_ref(sbo_ctx* c, const double* pts, long long N, long long nlines, double* mean_out, double* var_out, bool direct) {
  const ModelConst& mc = c->mc;
  const size_t lds = sizeof(double) * kRefPer * (size_t)mc.npad;
  const dim3 grid((unsigned)((N + kRefPer - 1) / kRefPer), (unsigned)mc.q);
  if (direct) {
    auto kern = k_ref_list<D, 0>;
    SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, c->stream, mc, c->cs, pts, N, nlines, (const double*)c->As.p, (const double*)c->sqA.p,
                       (const double*)c->alpha64.p, c->a_ld, c->invk_plain, (size_t)mc.n * mc.n, mc.n, mean_out, var_out);
  } else {
    auto kern = k_ref_list<D, 1>;
    SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, c->stream, mc, c->cs, pts, N, nlines, (const double*)c->As.p, (const double*)c->sqA.p,
                       (const double*)c->alpha64.p, c->a_ld, (const double*)c->Fplain.p, (size_t)c->f_cap * c->f_cap, c->f_cap, mean_out, var_out);
  }
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

int guard_exact_list(sbo_ctx* c, const double* pts, long long N, double* mean_out, double* var_out) {
  if (N <= 0) return SBO_OK;
  if (c->last_k1 == 4 && ref_direct(c)) {
    switch (c->mc.dpad) {
      case 2: return launch_ref<2>(c, pts, N, 0, mean_out, var_out, true);
      case 4: return launch_ref<4>(c, pts, N, 0, mean_out, var_out, true);
      default: return launch_ref<8>(c, pts, N, 0, mean_out, var_out, true);
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
int guard_band_host(sbo_ctx* c, const double* dm, const double* dv, const double* rl) {
  int rc;
  if ((rc = ensure(c->gb, sizeof(GuardBand)))) return rc;
  GuardBand hb;
  memset(&hb, 0, sizeof(hb));
  for (int o = 0; o < c->mc.q && dm; ++o) { hb.dm[o] = dm[o]; hb.dv[o] = dv[o]; hb.rl[o] = rl[o]; }
  // (pageable source: the runtime stages it before the call returns)
  SBO_HIP(hipMemcpyAsync(c->gb.p, &hb, sizeof(hb), hipMemcpyHostToDevice, c->stream));
  return SBO_OK;
}

// K1b: behind the posterior launch of a plan that has no band yet -- the exact evaluator at the probe points, then the band
// from the deviations (both on the device, in stream order; nothing waits for the host)
int guard_band_bilinear(sbo_ctx* c) {
  const ModelConst& mc = c->mc;
  const CandSpec& cs = c->cs;
  const int q = mc.q;
  const long long nlines = cs.n_local / cs.count[0];
  int rc;
  if ((rc = ensure(c->gb, sizeof(GuardBand)))) return rc;
  if ((rc = ensure(c->gb_probe, sizeof(double) * 2 * (size_t)q * kGbProbes))) return rc;
  double* ref_m = (double*)c->gb_probe.p;
  double* ref_v = ref_m + (size_t)q * kGbProbes;
  const bool direct = ref_direct(c);
  if (!direct && (rc = factor_sync(c))) return rc;          // (library Cholesky: the factor is there; a caller's invK without chol_async: it is waited for)
  if ((rc = launch_ref<2>(c, nullptr, kGbProbes, nlines, ref_m, ref_v, direct))) return rc;
  const double* tail = c->bl.eff ? reinterpret_cast<const double*>(c->bl.eff + 4 * q) : nullptr;
  hipLaunchKernelGGL(k_gb_band, dim3(1), dim3(256), 0, c->stream, mc, cs, nlines, (const double*)c->mean.p, (const double*)c->var.p,
                     (const double*)ref_m, (const double*)ref_v, tail, (GuardBand*)c->gb.p);
  SBO_HIP(hipGetLastError());
  c->gb_plan_model = c->model_serial;
  c->gb_plan_first = cs.first;
  c->gb_plan_n = cs.n_local;
  return SBO_OK;
}

}  // namespace sbo
