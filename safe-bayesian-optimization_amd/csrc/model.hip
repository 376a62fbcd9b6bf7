// model.hip -- SURVEY.md section 8(f) rank 1 (model build): the per-model factorisation on the device.
//
// What models/GP_Safe.py:229-245 leaves in `inference_datasets` is invK (or, when the caller passes none, the
// hyper-parameters to build K = sf2 exp(-1/2 D) + (sn2 + float32 eps) I from).  The sweep kernels contract with a
// LOWER-triangular M, M^T M = invK (K1g / K1c / K1, and the table build of K1b), and the mean uses
// alpha = invK (Y_norm - mp).  Here, one workgroup per output:
//   invK given : alpha = invK rhs with the matrix exactly as passed; M from the "UL" factorisation of the symmetrised
//                invK: reverse both index orders, factor J invK J = U~^T U~ (U~ upper), M = J U~ J.
//   invK absent: K from the expanded distance (models/GP_Safe.py:119), K = U^T U, M = L^-1 = U^-T by row-wise
//                substitution, alpha = M^T (M rhs).
// The factor is then packed into the matrix-core fragment images every K1 kernel reads (Fpk).  Small models (n < 256)
// run in one workgroup per output; larger ones use the blocked multi-workgroup form below (n = 512: 1.4 ms, n = 2048:
// 9 ms per model, against ~95 ms / seconds of host Cholesky).
#include <cmath>
#include <cstring>
#include <limits>
#include <type_traits>
#include <vector>
#include "device_common.hpp"

namespace sbo {

// In-place K = U^T U on the upper triangle of a row-major n x n matrix, one workgroup (the loop of fit.hip without the
// right-hand side).  `bad` is set when a pivot is not positive.  With E != nullptr (zero-initialised, row-major) the
// identity rides along as n extra right-hand sides, exactly as y does in fit.hip, and E ends up as L^-1 (L = U^T).
__device__ void factor_utu(double* __restrict__ U, int n, int* bad, double* sh_piv, double* __restrict__ E = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  for (int j = 0; j < n; ++j) {
    if (tid == 0) {
      const double piv = U[(size_t)j * n + j];
      if (!(piv > 0.0)) *bad = 1;
      *sh_piv = piv > 0.0 ? sqrt(piv) : 1.0;
    }
    __syncthreads();
    const double ljj = *sh_piv;
    double* uj = U + (size_t)j * n;
    for (int i = j + 1 + tid; i < n; i += blockDim.x) uj[i] /= ljj;
    if (tid == 0) uj[j] = ljj;
    double* ej = E ? E + (size_t)j * n : nullptr;
    if (E) {
      for (int cidx = tid; cidx < j; cidx += blockDim.x) ej[cidx] /= ljj;
      if (tid == 0) ej[j] = 1.0 / ljj;
    }
    __syncthreads();
    // trailing update, row k from column k on; a wave keeps two rows in flight (the loop is bound by memory latency)
    for (int k = j + 1 + wave; k < n; k += 2 * nw) {
      const int k2 = k + nw;
      const double f = uj[k];
      double* uk = U + (size_t)k * n;
      if (k2 < n) {
        const double f2 = uj[k2];
        double* uk2 = U + (size_t)k2 * n;
        for (int i = k + lane; i < n; i += 64) {
          const double v = uj[i];
          uk[i] -= f * v;
          if (i >= k2) uk2[i] -= f2 * v;
        }
      } else {
        for (int i = k + lane; i < n; i += 64) uk[i] -= f * uj[i];
      }
      if (E) {                                  // rows k (and k2) of the inverse, columns 0..j
        double* ek = E + (size_t)k * n;
        for (int cidx = lane; cidx <= j; cidx += 64) ek[cidx] -= f * ej[cidx];
        if (k2 < n) {
          const double f2 = uj[k2];
          double* ek2 = E + (size_t)k2 * n;
          for (int cidx = lane; cidx <= j; cidx += 64) ek2[cidx] -= f2 * ej[cidx];
        }
      }
    }
    __syncthreads();
  }
}

// mode 0: W = invK [q][n][n] given.  mode 1: build K from As / sqA.  Outputs: F [q][n][n] lower triangle (row-major),
// alpha [q][npad], bad[q].  `work` [q][n][n] scratch.
__global__ __launch_bounds__(1024) void k_model_build(int mode, int n, int npad, int dpad, int d, const double* __restrict__ W,
                                                     const double* __restrict__ As, const double* __restrict__ sqA,
                                                     const double* __restrict__ rhs, const ModelConst mc, double* __restrict__ work,
                                                     double* __restrict__ F, double* __restrict__ alpha, int* __restrict__ bad) {
  __shared__ double sh_piv;
  __shared__ int sh_bad;
  const int o = blockIdx.x, tid = threadIdx.x;
  double* U = work + (size_t)o * n * n;
  double* Fo = F + (size_t)o * n * n;
  const double* r = rhs + (size_t)o * n;
  if (tid == 0) sh_bad = 0;
  __syncthreads();
  if (mode == 0) {
    const double* Wo = W + (size_t)o * n * n;
    // alpha straight from the caller's inverse (GP_Safe.py:342)
    for (int i = tid; i < n; i += blockDim.x) {
      double s = 0.0;
      for (int j = 0; j < n; ++j) s += Wo[(size_t)i * n + j] * r[j];
      alpha[(size_t)o * npad + i] = s;
    }
    // reversed, symmetrised copy (upper triangle)
    for (long long idx = tid; idx < (long long)n * n; idx += blockDim.x) {
      const int i = (int)(idx / n), k = (int)(idx % n);
      if (k < i) continue;
      const int ri = n - 1 - i, rk = n - 1 - k;
      U[idx] = 0.5 * (Wo[(size_t)ri * n + rk] + Wo[(size_t)rk * n + ri]);
    }
    __syncthreads();
    factor_utu(U, n, &sh_bad, &sh_piv);
    // M = J U~ J  (lower triangular):  M[i][j] = U~[n-1-i][n-1-j], j <= i
    for (long long idx = tid; idx < (long long)n * n; idx += blockDim.x) {
      const int i = (int)(idx / n), j = (int)(idx % n);
      Fo[idx] = j <= i ? U[(size_t)(n - 1 - i) * n + (n - 1 - j)] : 0.0;
    }
  } else {
    const double sf2 = mc.sf2[o], sn2 = mc.sn2[o];
    const double* Ao = As + (size_t)o * npad * dpad;
    const double* so = sqA + (size_t)o * npad;
    // K[i][k] = sf2 exp(-1/2 dist(i, k)) + sn2 [i == k], dist as GP_Safe.py:119 evaluates it; the lower element (i >= k)
    // is the one a lower Cholesky reads, stored here at the transposed (upper) position
    for (long long idx = tid; idx < (long long)n * n; idx += blockDim.x) {
      const int k = (int)(idx / n), i = (int)(idx % n);      // upper position (k, i), k <= i
      if (i < k) continue;
      double dot = 0.0;
      for (int a = 0; a < d; ++a) dot += Ao[(size_t)i * dpad + a] * Ao[(size_t)k * dpad + a];
      const double dist = (-2.0 * dot + so[i]) + so[k];
      U[idx] = sf2 * exp(-0.5 * dist) + (i == k ? sn2 : 0.0);
    }
    for (long long idx = tid; idx < (long long)n * n; idx += blockDim.x) Fo[idx] = 0.0;
    __syncthreads();
    factor_utu(U, n, &sh_bad, &sh_piv, Fo);       // Fo = L^-1 = M
    // alpha = M^T (M rhs)
    double* t = U;                                            // scratch: the factor is no longer needed
    for (int i = tid; i < n; i += blockDim.x) {
      double s = 0.0;
      for (int j = 0; j <= i; ++j) s += Fo[(size_t)i * n + j] * r[j];
      t[i] = s;
    }
    __syncthreads();
    for (int j = tid; j < n; j += blockDim.x) {
      double s = 0.0;
      for (int i = j; i < n; ++i) s += Fo[(size_t)i * n + j] * t[i];
      alpha[(size_t)o * npad + j] = s;
    }
  }
  __syncthreads();
  if (tid == 0) bad[o] = sh_bad;
}

// ---- blocked, multi-workgroup form of the same factorisation (n >= kBlockedFrom) ---------------------------------------------
// Right-looking U^T U with panels of kPB rows; per panel two kinds of launches, every output (blockIdx.y) at once:
//   k_chol_panel : every workgroup factors the kPB x kPB diagonal block in LDS (redundantly: it is tiny), then applies
//                  Ukk^-T to its share of the columns to the right (the panel rows of U) and, with E, to the columns of
//                  the inverse companion left of and inside the panel (E = L^-1 rides along exactly as in factor_utu);
//   k_chol_update: rank-kPB update of the trailing matrix (upper triangle) and of the trailing rows of E, tiled 32 x 32.
constexpr int kPB = 32;

__device__ __forceinline__ double readlane_f64(double v, int l) {      // l: wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
constexpr int kBlockedFrom = 96;   // smallest n factorised by the blocked form (below: one workgroup does everything)

// mode 0 front end, coalesced (the all-in-one k_chol_prep below reads invK by columns: 133 us at n = 512):
//   k_invk_alpha: alpha = invK rhs with the caller's matrix as given (GP_Safe.py:342), one wave per row;
//   k_invk_reverse: U = upper triangle of J sym(invK) J in 32 x 32 tiles, the transposed partner of a tile through LDS
__global__ __launch_bounds__(256) void k_invk_alpha(int n, int npad, const double* __restrict__ W, const double* __restrict__ rhs,
                                                    double* __restrict__ alpha) {
  const int o = blockIdx.y, lane = threadIdx.x & 63;
  const double* Wo = W + (size_t)o * n * n;
  const double* r = rhs + (size_t)o * n;
  for (int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < n; i += (gridDim.x * blockDim.x) >> 6) {
    double s = 0.0;
    for (int j = lane; j < n; j += 64) s += Wo[(size_t)i * n + j] * r[j];
    s = wave_sum(s);
    if (lane == 0) alpha[(size_t)o * npad + i] = s;
  }
}
__global__ __launch_bounds__(256) void k_invk_reverse(int n, const double* __restrict__ W, double* __restrict__ work) {
  __shared__ double Ta[32][33], Tb[32][33];
  const int o = blockIdx.z, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const double* Wo = W + (size_t)o * n * n;
  double* U = work + (size_t)o * n * n;
  const int i0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
  if (k0 + 31 < i0) {                                   // tile entirely below the diagonal: zeros
    for (int rr = ty; rr < 32; rr += 8)
      if (i0 + rr < n && k0 + tx < n) U[(size_t)(i0 + rr) * n + k0 + tx] = 0.0;
    return;
  }
  // element (i, k) of the tile needs W[n-1-i][n-1-k] and W[n-1-k][n-1-i]: the blocks W[R..][C..] and W[C..][R..] with
  // R = n-1-i0-31, C = n-1-k0-31 (rows / columns clipped at 0), read row-wise
  const int R = n - 1 - i0 - 31, Cc = n - 1 - k0 - 31;
  for (int rr = ty; rr < 32; rr += 8) {
    const int ra = R + rr, ca = Cc + tx;
    Ta[rr][tx] = (ra >= 0 && ra < n && ca >= 0 && ca < n) ? Wo[(size_t)ra * n + ca] : 0.0;
    const int rb = Cc + rr, cb = R + tx;
    Tb[rr][tx] = (rb >= 0 && rb < n && cb >= 0 && cb < n) ? Wo[(size_t)rb * n + cb] : 0.0;
  }
  __syncthreads();
  for (int rr = ty; rr < 32; rr += 8) {
    const int i = i0 + rr, k = k0 + tx;
    if (i < n && k < n) {
      // W[n-1-i][n-1-k] = Ta[31 - rr][31 - tx];  W[n-1-k][n-1-i] = Tb[31 - tx][31 - rr]
      U[(size_t)i * n + k] = k >= i ? 0.5 * (Ta[31 - rr][31 - tx] + Tb[31 - tx][31 - rr]) : 0.0;
    }
  }
}

__global__ __launch_bounds__(256) void k_chol_prep(int mode, int n, int npad, int dpad, int d, const double* __restrict__ W,
                                                   const double* __restrict__ As, const double* __restrict__ sqA,
                                                   const double* __restrict__ rhs, const ModelConst mc, double* __restrict__ work,
                                                   double* __restrict__ F, double* __restrict__ alpha) {
  const int o = blockIdx.y;
  double* U = work + (size_t)o * n * n;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x, gstride = (long long)gridDim.x * blockDim.x;
  if (mode == 0) {
    const double* Wo = W + (size_t)o * n * n;
    const double* r = rhs + (size_t)o * n;
    for (long long i = gid; i < n; i += gstride) {                       // alpha from the caller's inverse as given
      double s = 0.0;
      for (int j = 0; j < n; ++j) s += Wo[(size_t)i * n + j] * r[j];
      alpha[(size_t)o * npad + i] = s;
    }
    for (long long idx = gid; idx < (long long)n * n; idx += gstride) { // reversed, symmetrised copy (upper triangle)
      const int i = (int)(idx / n), k = (int)(idx % n);
      const int ri = n - 1 - i, rk = n - 1 - k;
      U[idx] = k >= i ? 0.5 * (Wo[(size_t)ri * n + rk] + Wo[(size_t)rk * n + ri]) : 0.0;
    }
  } else {
    const double sf2 = mc.sf2[o], sn2 = mc.sn2[o];
    const double* Ao = As + (size_t)o * npad * dpad;
    const double* so = sqA + (size_t)o * npad;
    double* Fo = F + (size_t)o * n * n;
    for (long long idx = gid; idx < (long long)n * n; idx += gstride) {
      const int k = (int)(idx / n), i = (int)(idx % n);                  // upper position (k, i)
      double v = 0.0;
      if (i >= k) {
        double dot = 0.0;
        for (int a = 0; a < d; ++a) dot += Ao[(size_t)i * dpad + a] * Ao[(size_t)k * dpad + a];
        v = sf2 * exp(-0.5 * ((-2.0 * dot + so[i]) + so[k])) + (i == k ? sn2 : 0.0);
      }
      U[idx] = v;
      Fo[idx] = 0.0;                                                     // E starts empty (its unit diagonal is implicit)
    }
  }
}

// ---- r03: one launch per panel ------------------------------------------------------------------------------------------
// The two-launch form of round 2 (a panel kernel + a trailing update, removed in r04) took 16 x (19-25 + 7-10) us at n = 512, and what a panel
// launch waits for is the pivot chain of its 32 x 32 diagonal block: IEEE sqrt + IEEE division per pivot and ~90 lane
// broadcasts per step on ONE wave).  k_chol_step is the same right-looking factorisation with the trailing update one
// launch late ("look-ahead"): the launch of panel kb
//   role A (workgroups [0, nA)): applies the PENDING rank-32 update of panel kb - 32 to the diagonal block and to its own
//          columns of the panel rows, factors the block with all four waves in LDS (four elements per thread, one barrier
//          per pivot; the pivot's reciprocal square root by v_rsq_f64 + two Newton steps instead of sqrt and division),
//          then the forward substitution of its columns, a row per lane, multiplying by the stored reciprocals;
//   role B (the rest): the pending update of panel kb - 32 on the rows BELOW the current panel (k_chol_update's tiles).
// Every element receives the same updates in the same order as before (sum over a panel's 32 rows, then one subtraction);
// the factors differ from the two-launch form only through the reciprocal pivots (a few ulp).
__device__ __forceinline__ double rsqrt_nr(double x) {
  double r = __builtin_amdgcn_rsq(x);                       // v_rsq_f64: ~2^-23 relative
  const double h = 0.5 * x;
  r = fma(r, fma(-h * r, r, 0.5), r);                       // r (1.5 - h r^2), twice: ~2^-45, then rounding
  r = fma(r, fma(-h * r, r, 0.5), r);
  return r;
}

__global__ __launch_bounds__(256) void k_chol_step(double* __restrict__ work, double* __restrict__ F, int with_E, int n, int kb, int nA,
                                                   int ti, int* __restrict__ bad, double* __restrict__ Dg) {
  __shared__ double D[kPB][kPB + 1];      // role A: working copy of the diagonal block (upper triangle)
  __shared__ double Uf[kPB][kPB + 1];     // role A: its factor
  __shared__ double P[kPB][kPB + 1];      // role A: P[t][r] = U[kbp + t][kb + r];  role B: the tile's rows of the pending panel
  __shared__ double Qx[kPB][kPB + 1];     // role B: the tile's columns of the pending panel
  __shared__ double rinv_sh[kPB];
  const int o = blockIdx.y, tid = threadIdx.x, bid = blockIdx.x;
  double* U = work + (size_t)o * n * n;
  double* E = F + (size_t)o * n * n;
  const int kw = n - kb < kPB ? n - kb : kPB;
  const int kbp = kb - kPB;                                  // the pending panel (kb > 0): always a full one
  if (bid >= nA) {
    // ---- role B: pending update of panel kbp on rows >= kb + kw (k_chol_update's arithmetic) ----
    int b = bid - nA, part = 0;
    if (b >= ti * ti) { b -= ti * ti; part = 1; }
    const int bx = part == 0 ? b % ti : b / ti, by = part == 0 ? b / ti : b % ti;
    const int i0 = kb + kw + by * 32;
    const int x0 = part == 0 ? kb + kw + bx * 32 : bx * 32;
    if (i0 >= n) return;
    if (part == 0 && x0 + 31 < i0) return;                   // tile entirely below the diagonal
    const double* Q = part == 0 ? U : E;
    for (int idx = tid; idx < kPB * 32; idx += blockDim.x) {
      const int r = idx / 32, t = idx % 32;
      P[r][t] = i0 + t < n ? U[(size_t)(kbp + r) * n + i0 + t] : 0.0;
      const int xc = x0 + t;
      const bool xin = part == 0 ? xc < n : xc < kb;
      // rows of E inside a panel are lower triangular: entries right of their diagonal are zero
      Qx[r][t] = (xin && (part == 0 || xc <= kbp + r)) ? Q[(size_t)(kbp + r) * n + xc] : 0.0;
    }
    __syncthreads();
    const int tx = tid % 32, ty = tid / 32;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int il = ty + 8 * rr, i = i0 + il, xc = x0 + tx;
      if (i >= n) continue;
      if (part == 0 ? (xc >= n || xc < i) : (xc >= kb)) continue;
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < kPB; ++r) acc += P[r][il] * Qx[r][tx];
      double* dst = (part == 0 ? U : E) + (size_t)i * n + xc;
      *dst -= acc;
    }
    return;
  }
  // ---- role A ----
  for (int idx = tid; idx < kPB * kPB; idx += blockDim.x) {
    const int r = idx / kPB, cc = idx % kPB;
    D[r][cc] = (r < kw && cc < kw && cc >= r) ? U[(size_t)(kb + r) * n + kb + cc] : 0.0;
    P[r][cc] = (kb > 0 && cc < kw) ? U[(size_t)(kbp + r) * n + kb + cc] : 0.0;
  }
  __syncthreads();
  const int cc = tid & 31, r0 = tid >> 5;                    // this thread's four elements: column cc, rows r0 + 8 m
  if (kb > 0) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int r = r0 + 8 * m;
      if (r <= cc && cc < kw) {
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < kPB; ++t) acc += P[t][r] * P[t][cc];
        D[r][cc] -= acc;
      }
    }
    __syncthreads();
  }
  bool notpd = false;
  for (int j = 0; j < kw; ++j) {
    const double piv = D[j][j];
    notpd = notpd || !(piv > 0.0);
    const double x = piv > 0.0 ? piv : 1.0;
    const double rv = rsqrt_nr(x);
    double ljj = x * rv;
    ljj = fma(0.5 * rv, fma(-ljj, ljj, x), ljj);             // one correction step: sqrt(x) to ~1 ulp
    const double ujc = D[j][cc] * rv;                        // entry of the scaled pivot row in this thread's column
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int r = r0 + 8 * m;
      if (r > j && cc >= r) D[r][cc] -= (D[j][r] * rv) * ujc;
      if (r == j) Uf[j][cc] = cc > j ? ujc : (cc == j ? ljj : 0.0);
    }
    if (tid == 0) rinv_sh[j] = rv;
    __syncthreads();
  }
  for (int idx = tid; idx < kPB * kPB; idx += blockDim.x)     // rows / columns beyond a short last panel
    if (idx / kPB >= kw || idx % kPB >= kw) Uf[idx / kPB][idx % kPB] = 0.0;
  if (tid >= kw && tid < kPB) rinv_sh[tid] = 1.0;
  if (tid == 0 && notpd) bad[o] = 1;
  __syncthreads();
  // the factored block goes to the side array k_chol_finish reads (never back into U: see k_chol_panel)
  if (bid == 0) {
    double* dg = Dg + ((size_t)o * ((n + kPB - 1) / kPB) + kb / kPB) * (kPB * kPB);
    for (int idx = tid; idx < kPB * kPB; idx += blockDim.x) dg[idx] = Uf[idx / kPB][idx % kPB];
  }
  // columns [kb + kw, n) of U, then (with E) [0, kb + kw) of E: a column per half wave, a row per lane
  const int nright = n - kb - kw;
  const int ncols = nright + (with_E ? kb + kw : 0);
  const int lane = tid & 63, r = lane & 31, half = lane >> 5;
  double dcol[kPB];
#pragma unroll
  for (int t = 0; t < kPB; ++t) dcol[t] = Uf[t][r];
  const double myrinv = rinv_sh[r];
  const int pairs_total = (ncols + 1) / 2;
  for (int pr = bid * (blockDim.x >> 6) + (tid >> 6); pr < pairs_total; pr += nA * (blockDim.x >> 6)) {
    const int xcol = 2 * pr + half;
    const bool live = xcol < ncols && r < kw;
    const bool isU = xcol < nright;
    const int col = isU ? kb + kw + xcol : xcol - nright;
    double* base = isU ? U : E;
    double acc = 0.0;
    if (live) acc = (!isU && col >= kb) ? (col - kb == r ? 1.0 : 0.0) : base[(size_t)(kb + r) * n + col];
    if (kb > 0) {
      // pending update: acc -= sum_t P[t][r] X[kbp + t][col]; lane t of a half fetches X[kbp + t][col] (rows of E inside the
      // pending panel are zero right of their diagonal, and so for every column of the current panel's own block)
      double xr = 0.0;
      if (xcol < ncols && (isU || col <= kbp + r)) xr = base[(size_t)(kbp + r) * n + col];
      double pend = 0.0;
#pragma unroll
      for (int t = 0; t < kPB; ++t) {
        const double x0 = readlane_f64(xr, t), x1 = readlane_f64(xr, 32 + t);
        pend += P[t][r] * (half ? x1 : x0);
      }
      if (live && (isU || col < kb)) acc -= pend;
    }
#pragma unroll
    for (int t = 0; t < kPB; ++t) {
      if (t < kw) {
        if (r == t) acc = acc * myrinv;
        const double v0 = readlane_f64(acc, t), v1 = readlane_f64(acc, 32 + t);
        const double vt = half ? v1 : v0;
        if (r > t) acc -= dcol[t] * vt;
      }
    }
    if (live && (isU || col <= kb + r)) base[(size_t)(kb + r) * n + col] = acc;
  }
}

__global__ __launch_bounds__(256) void k_chol_finish(int mode, int n, int npad, const double* __restrict__ rhs,
                                                     double* __restrict__ work, double* __restrict__ F, double* __restrict__ alpha,
                                                     const double* __restrict__ Dg) {
  const int o = blockIdx.y;
  double* U = work + (size_t)o * n * n;
  double* Fo = F + (size_t)o * n * n;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x, gstride = (long long)gridDim.x * blockDim.x;
  if (mode == 0) {
    for (long long idx = gid; idx < (long long)n * n; idx += gstride) {   // M = J U~ J
      const int i = (int)(idx / n), j = (int)(idx % n);
      const int ur = n - 1 - i, uc = n - 1 - j;                           // position in U~ (diagonal blocks: the side array)
      double v = 0.0;
      if (j <= i)
        v = ur / kPB == uc / kPB ? Dg[((size_t)o * ((n + kPB - 1) / kPB) + ur / kPB) * (kPB * kPB) + (ur % kPB) * kPB + uc % kPB]
                                 : U[(size_t)ur * n + uc];
      Fo[idx] = v;
    }
  } else {
    // alpha = M^T (M rhs), M = Fo (lower).  t = M rhs goes through the (now free) first row of `work`
    const double* r = rhs + (size_t)o * n;
    if (blockIdx.x == 0) {
      for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double s = 0.0;
        for (int j = 0; j <= i; ++j) s += Fo[(size_t)i * n + j] * r[j];
        U[i] = s;
      }
      __syncthreads();
      for (int j = threadIdx.x; j < n; j += blockDim.x) {
        double s = 0.0;
        for (int i = j; i < n; ++i) s += Fo[(size_t)i * n + j] * U[i];
        alpha[(size_t)o * npad + j] = s;
      }
    }
  }
}

// lower-triangular factor [q][n][n] -> matrix-core A-fragment images [q][tri(I, J)][256] in the model dtype
template <typename T>
__global__ __launch_bounds__(256) void k_pack_factor(const double* __restrict__ F, int n, int ld, size_t ostride, int nb,
                                                     size_t fpk_stride, T* __restrict__ Fpk) {
  const int o = blockIdx.y;
  const size_t ntri = (size_t)nb * (nb + 1) / 2;
  const double* Fo = F + (size_t)o * ostride;
  T* dst = Fpk + (size_t)o * fpk_stride;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < ntri * 256; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t blk = idx >> 8;
    const int e = (int)(idx & 255);
    // block (I, J) of the triangle from its linear index
    int I = (int)((sqrt(8.0 * (double)blk + 1.0) - 1.0) * 0.5);
    while ((size_t)(I + 1) * (I + 2) / 2 <= blk) ++I;
    while ((size_t)I * (I + 1) / 2 > blk) --I;
    const int J = (int)(blk - (size_t)I * (I + 1) / 2);
    int r, k, kk;
    MM<T>::unpack_pos(e, r, k, kk);
    const int row = 16 * I + r, col = 16 * J + MM<T>::jslot(kk, k);
    dst[idx] = (row < n && col < n && col <= row) ? (T)Fo[(size_t)row * ld + col] : T(0);
  }
}

// the caller's invK [q][n][n] (row-major, as given) -> full A images [q][nb][nb][256] for the table GEMM of the K1b plan
__global__ __launch_bounds__(256) void k_pack_full(const double* __restrict__ W, int n, int nb, double* __restrict__ img) {
  const int o = blockIdx.y;
  const double* Wo = W + (size_t)o * n * n;
  double* dst = img + (size_t)o * nb * nb * 256;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)nb * nb * 256; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t blk = idx >> 8;
    const int e = (int)(idx & 255), I = (int)(blk / nb), J = (int)(blk % nb);
    int r, k, kk;
    MM<double>::unpack_pos(e, r, k, kk);
    const int row = 16 * I + r, col = 16 * J + MM<double>::jslot(kk, k);
    dst[idx] = (row < n && col < n) ? Wo[(size_t)row * n + col] : 0.0;
  }
}

// alpha [q][sstride] fp64 -> [q][dstride] in the model dtype (first n entries of each output, zero padding)
template <typename T>
__global__ void k_cast_alpha(const double* __restrict__ src, int sstride, int n, int q, int dstride, T* __restrict__ dst) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < q * dstride; i += gridDim.x * blockDim.x) {
    const int o = i / dstride, j = i % dstride;
    dst[i] = j < n ? (T)src[(size_t)o * sstride + j] : T(0);
  }
}

// One more observation with frozen hyper-parameters and normalisation.  With k = K(X, x_new),
// kappa = sf2 + sn2 and u = invK k = M^T (M k), s = kappa - k.u, the inverse of the bordered matrix is
// [[invK + u u^T / s, -u / s], [-u^T / s, 1 / s]]: the lower factor simply gains the row (-u^T / sqrt(s), 1 / sqrt(s)),
// and alpha_new = (alpha + u (k.alpha - rho) / s, (rho - k.alpha) / s).  O(n^2), one workgroup per output.
__global__ __launch_bounds__(1024) void k_model_append(int n, int ld, double* __restrict__ F, double* __restrict__ alpha,
                                                       const double* __restrict__ kvec, const double* __restrict__ kappa,
                                                       const double* __restrict__ rho, double* __restrict__ scratch,
                                                       int* __restrict__ bad) {
  __shared__ double red[32];
  __shared__ double sh_s, sh_ka;
  const int o = blockIdx.x, tid = threadIdx.x;
  double* M = F + (size_t)o * ld * ld;
  double* al = alpha + (size_t)o * ld;
  const double* k = kvec + (size_t)o * n;
  double* t = scratch + (size_t)o * 2 * ld;
  double* u = t + ld;
  for (int i = tid; i < n; i += blockDim.x) {            // t = M k
    double s = 0.0;
    for (int j = 0; j <= i; ++j) s += M[(size_t)i * ld + j] * k[j];
    t[i] = s;
  }
  __syncthreads();
  double ku = 0.0, ka = 0.0;
  for (int j = tid; j < n; j += blockDim.x) {            // u = M^T t ;  partial sums of k.u and k.alpha
    double s = 0.0;
    for (int i = j; i < n; ++i) s += M[(size_t)i * ld + j] * t[i];
    u[j] = s;
    ku += k[j] * s;
    ka += k[j] * al[j];
  }
  auto block_sum = [&](double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
  };
  ku = block_sum(ku);
  ka = block_sum(ka);
  if (tid == 0) {
    const double s = kappa[o] - ku;
    if (!(s > 0.0)) bad[o] = 1;
    sh_s = s > 0.0 ? s : 1.0;
    sh_ka = ka;
  }
  __syncthreads();
  const double s = sh_s, rs = 1.0 / sqrt(s), coef = (sh_ka - rho[o]) / s;
  for (int j = tid; j < n; j += blockDim.x) {
    M[(size_t)n * ld + j] = -u[j] * rs;
    al[j] += u[j] * coef;
  }
  if (tid == 0) {
    M[(size_t)n * ld + n] = rs;
    al[n] = (rho[o] - sh_ka) / s;
  }
  for (int j = n + 1 + tid; j < ld; j += blockDim.x) M[(size_t)n * ld + j] = 0.0;
}

// Derived model arrays from the uploaded X_norm [n][d] / Y_norm [n][q] (models/GP_Safe.py:112-116, 331-342):
//   As[o][j][a] = X_norm[j][a] ell_a^-1/2, sqA[o][j] = sum_a As^2 (model dtype for the K1 kernels, fp64 for the build),
//   Xn[j][a] (model dtype, for the mean gradient), rhs[o][j] = Y_norm[j][o] - mp_o (fp64, only with Y_norm).
// Rows j >= n of the padded arrays are zero.
template <typename T>
__global__ __launch_bounds__(256) void k_model_prep(const ModelConst mc, const double* __restrict__ Xh, const double* __restrict__ Yh,
                                                    T* __restrict__ As, T* __restrict__ sqA, T* __restrict__ Xn, double* __restrict__ As64,
                                                    double* __restrict__ sq64, double* __restrict__ rhs) {
  const int n = mc.n, npad = mc.npad, d = mc.d, D = mc.dpad, q = mc.q;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < q * npad; i += gridDim.x * blockDim.x) {
    const int o = i / npad, j = i - o * npad;
    double s_ = 0.0;
    for (int a = 0; a < D; ++a) {
      double v = 0.0, x = 0.0;
      if (j < n && a < d) {
        x = Xh[(size_t)j * d + a];
        v = x * mc.vinv[o][a];                                          // GP_Safe.py:115
        s_ += v * v;
      }
      As[(size_t)i * D + a] = (T)v;
      As64[(size_t)i * D + a] = v;
      if (o == 0) Xn[(size_t)j * D + a] = (T)x;
    }
    sqA[i] = (T)s_;
    sq64[i] = s_;
    if (Yh && j < n) rhs[(size_t)o * n + j] = Yh[(size_t)j * q + o] - mc.mp[o];
  }
}

// Upload X_norm (and Y_norm) through the pinned staging area and derive the model arrays on the device.  Layout of
// c->mwork (doubles): XY [n d + n q] | As64 | sq64 | rhs | sf2 sn2 | W | work | Dg | bad (ints)
struct ModelWork {
  double *XY, *As64, *sq64, *rhs, *sfsn, *W, *work, *Dg;
  int* bad;
};
static int model_work(sbo_ctx* c, bool with_W, ModelWork& w) {
  const ModelConst& mc = c->mc;
  const size_t n = mc.n, q = mc.q, npad = mc.npad, nn = n * n;
  const size_t nXY = n * mc.d + n * q, nAs = q * npad * mc.dpad, nsq = q * npad, nrhs = q * n;
  const size_t ndiag = q * ((n + kPB - 1) / kPB) * (kPB * kPB);
  const size_t total = nXY + nAs + nsq + nrhs + 2 * q + (with_W ? q * nn : 0) + q * nn + ndiag + q + 8;
  int rc = ensure(c->mwork, sizeof(double) * total);
  if (rc) return rc;
  w.XY = (double*)c->mwork.p;
  w.As64 = w.XY + nXY;
  w.sq64 = w.As64 + nAs;
  w.rhs = w.sq64 + nsq;
  w.sfsn = w.rhs + nrhs;
  w.W = w.sfsn + 2 * q;
  w.work = w.W + (with_W ? q * nn : 0);
  w.Dg = w.work + q * nn;
  w.bad = (int*)(w.Dg + ndiag);
  return SBO_OK;
}
static int stage_ensure(sbo_ctx* c, size_t bytes) {
  if (c->h_stage_bytes >= bytes) return SBO_OK;
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  c->h_stage = nullptr;
  c->h_stage_bytes = 0;
  const size_t want = bytes + bytes / 2 + 4096;
  if (hipHostMalloc(&c->h_stage, want, hipHostMallocDefault) != hipSuccess) return fail(SBO_E_HIP, "hipHostMalloc (model staging)");
  c->h_stage_bytes = want;
  return SBO_OK;
}
template <typename T>
static int model_prep_t(sbo_ctx* c, const double* X_norm, const double* Y_norm, const ModelWork& w) {
  const ModelConst& mc = c->mc;
  const size_t n = mc.n, q = mc.q, npad = mc.npad;
  int rc;
  if ((rc = stage_ensure(c, sizeof(double) * (n * mc.d + n * q + 2 * q)))) return rc;
  double* hs = (double*)c->h_stage;
  std::memcpy(hs, X_norm, sizeof(double) * n * mc.d);
  if (Y_norm) std::memcpy(hs + n * mc.d, Y_norm, sizeof(double) * n * q);
  SBO_HIP(hipMemcpyAsync(w.XY, hs, sizeof(double) * (n * mc.d + (Y_norm ? n * q : 0)), hipMemcpyHostToDevice, c->stream));
  if ((rc = ensure(c->As, sizeof(T) * q * npad * mc.dpad))) return rc;
  if ((rc = ensure(c->sqA, sizeof(T) * q * npad))) return rc;
  if ((rc = ensure(c->Xn, sizeof(T) * npad * mc.dpad))) return rc;
  hipLaunchKernelGGL((k_model_prep<T>), dim3((unsigned)((q * npad + 255) / 256)), dim3(256), 0, c->stream, mc, (const double*)w.XY,
                     Y_norm ? (const double*)(w.XY + n * mc.d) : (const double*)nullptr, (T*)c->As.p, (T*)c->sqA.p, (T*)c->Xn.p, w.As64,
                     w.sq64, w.rhs);
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}
int model_prep(sbo_ctx* c, const double* X_norm) {         // (after an append: the derived arrays with the new row)
  ModelWork w;
  int rc = model_work(c, false, w);
  if (rc) return rc;
  rc = c->dtype == SBO_F64 ? model_prep_t<double>(c, X_norm, nullptr, w) : model_prep_t<float>(c, X_norm, nullptr, w);
  if (rc) return rc;
  SBO_HIP(hipStreamSynchronize(c->stream));                 // (the staging area is free again)
  return SBO_OK;
}

// The factorisation proper on stream `fs`: front end (mode 0: reversed symmetrised copy of invK -- alpha is made by the caller of
// this function from the matrix as given --; mode 1: K from the expanded distance), the panels, the finish, the fragment images.
template <typename T>
static int factor_chain_enqueue(sbo_ctx* c, const ModelWork& w, int mode, hipStream_t fs) {
  const ModelConst& mc = c->mc;
  const int n = mc.n, npad = mc.npad, q = mc.q, nb = npad / 16;
  const size_t nn = (size_t)n * n;
  double* dF = (double*)c->Fplain.p;
  double* dalpha = (double*)c->alpha64.p;
  if (n >= kBlockedFrom) {
    // blocked multi-workgroup factorisation: the single-workgroup loop is bound by the latency of its own updates
    if (mode == 0) {
      hipLaunchKernelGGL(k_invk_reverse, dim3((unsigned)((n + 31) / 32), (unsigned)((n + 31) / 32), q), dim3(256), 0, fs, n,
                         (const double*)w.W, w.work);
    } else {
      hipLaunchKernelGGL(k_chol_prep, dim3(256, q), dim3(256), 0, fs, mode, n, npad, mc.dpad, mc.d, (const double*)w.W,
                         (const double*)w.As64, (const double*)w.sq64, (const double*)w.rhs, mc, w.work, dF, dalpha);
    }
    {
      // one launch per panel: the pending update of the previous panel rides in the panel's own launch (k_chol_step)
      for (int kb = 0; kb < n; kb += kPB) {
        const int kw = std::min(kPB, n - kb);
        const int ncols = (n - kb - kw) + (mode ? kb + kw : 0);
        const int nA = std::max(1, ((ncols + 1) / 2 + 3) / 4);
        const int ti = kb > 0 ? (n - kb - kw + 31) / 32 : 0;
        const int nB = ti * ti + (mode && kb > 0 ? ((kb + 31) / 32) * ti : 0);
        hipLaunchKernelGGL(k_chol_step, dim3((unsigned)(nA + nB), q), dim3(256), 0, fs, w.work, dF, mode, n, kb, nA, ti, w.bad, w.Dg);
      }
    }
    hipLaunchKernelGGL(k_chol_finish, dim3(256, q), dim3(256), 0, fs, mode, n, npad, (const double*)w.rhs, w.work, dF, dalpha,
                       (const double*)w.Dg);
  } else {
    hipLaunchKernelGGL(k_model_build, dim3(q), dim3(1024), 0, fs, mode, n, npad, mc.dpad, mc.d, (const double*)w.W,
                       (const double*)w.As64, (const double*)w.sq64, (const double*)w.rhs, mc, w.work, dF, dalpha, w.bad);
  }
  const size_t ntri = (size_t)nb * (nb + 1) / 2;
  // (k_pack_factor writes every element of the images, zeros included; only the over-read padding needs clearing)
  SBO_HIP(hipMemsetAsync((T*)c->Fpk.p + (size_t)q * c->fpk_stride, 0, sizeof(T) * 512, fs));
  hipLaunchKernelGGL((k_pack_factor<T>), dim3((unsigned)std::min<size_t>((ntri * 256 + 255) / 256, 4096), q), dim3(256), 0, fs,
                     (const double*)dF, n, n, nn, nb, c->fpk_stride, (T*)c->Fpk.p);
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

// Device side of sbo_model_set.  host_invK may be NULL (the library factors K itself).  On success Fpk (dtype T), alpha
// (dtype T), the derived arrays As / sqA / Xn and the fp64 factor / alpha (Fplain, alpha64: leading dimension n / npad
// until an append grows them) are in place.  One host synchronisation, at the end (the positive-definiteness verdict);
// when a K1b-capable grid is resident the axis bases of the new model are built meanwhile on the second stream.
template <typename T>
static int model_build_t(sbo_ctx* c, const double* const* host_invK /* q matrices, or nullptr */, const double* X_norm, const double* Y_norm) {
  const ModelConst& mc = c->mc;
  const int n = mc.n, npad = mc.npad, q = mc.q, nb = npad / 16;
  const size_t nn = (size_t)n * n;
  int rc;
  if (c->factor_pending) {                // (the previous model's deferred factor chain works in the buffers reused below)
    SBO_HIP(hipEventSynchronize(c->ev_factor));
    c->factor_pending = false;
  }
  c->invk_img_valid = false;
  c->invk_w_valid = false;
  c->factor_todo = false;
  ModelWork w;
  if ((rc = model_work(c, host_invK != nullptr, w))) return rc;
  if ((rc = ensure(c->Fplain, sizeof(double) * (size_t)q * nn))) return rc;
  if ((rc = ensure(c->alpha64, sizeof(double) * (size_t)q * npad))) return rc;
  c->f_cap = n;
  c->a_ld = npad;
  double* dalpha = (double*)c->alpha64.p;
  if ((rc = model_prep_t<T>(c, X_norm, Y_norm, w))) return rc;
  bool eager_basis = false;
  // (r04) a caller's invK on a K1b-capable grid: the model's first sweep runs on interpolated node values (K1i, bilinear.hip), which
  // needs no bases -- they are made when the model is swept a second time
  const bool interp_first = bilinear_applicable(c) && c->bilinear == 1 && host_invK != nullptr && n >= kBlockedFrom && c->chol_async &&
                            !c->is_shadow && c->stream4 && std::is_same<T, double>::value && c->mc.npad % 16 == 0;
  if (bilinear_applicable(c) && !interp_first) {
    // the bases need X_norm only: next to the factorisation, on the second stream -- and ahead of the upload of invK
    // (4 MB at n = 512, ~0.15 ms of pageable copies the host sits in): their pivot loop is the longest chain of a model change
    SBO_HIP(hipEventRecord(c->ev[6], c->stream));
    SBO_HIP(hipStreamWaitEvent(c->stream2, c->ev[6], 0));
    if ((rc = bilinear_basis_enqueue(c, c->stream2, false))) return rc;
    eager_basis = true;
  }
  if (host_invK) {                        // (the reference keeps them as a list of q arrays: one copy each, no stacking on the host)
    for (int o = 0; o < q; ++o)
      SBO_HIP(hipMemcpyAsync(w.W + (size_t)o * nn, host_invK[o], sizeof(double) * nn, hipMemcpyHostToDevice, c->stream));
    c->invk_w_valid = std::is_same<T, double>::value && !c->is_shadow;
    c->invk_plain = w.W;
  }
  SBO_HIP(hipMemsetAsync(dalpha, 0, sizeof(double) * (size_t)q * npad, c->stream));
  SBO_HIP(hipMemsetAsync(w.bad, 0, sizeof(int) * q, c->stream));
  const int mode = host_invK ? 0 : 1;
  // A caller's invK on a K1b-capable grid: the GEMM posterior's tables take invK itself (k_pack_full), so the reverse
  // Cholesky factor is only needed by the O(n^2) kernels and by sbo_model_append -- it is built on stream4 and nobody waits
  // for it here (n = 512: 16 dependent panel launches, ~0.43 ms off the critical path of a model change).  Its launches are
  // not even enqueued here (the host needs ~0.1 ms for them): the next sweep does that while it waits for its own result
  // (model_factor_enqueue), a consumer of the factor before that enqueues and waits (factor_sync), and a model replaced
  // before anyone asked never factors at all.
  const bool deferred = mode == 0 && n >= kBlockedFrom && c->chol_async && !c->is_shadow && (eager_basis || interp_first) && c->stream4 &&
                        std::is_same<T, double>::value;
  const size_t ntri = (size_t)nb * (nb + 1) / 2;
  c->fpk_stride = ntri * 4 * 64;
  if ((rc = ensure(c->Fpk, sizeof(T) * ((size_t)q * c->fpk_stride + 512)))) return rc;   // + padding: the K1g pipeline over-reads
  if ((rc = ensure(c->alpha, sizeof(T) * (size_t)q * npad))) return rc;
  if (n >= kBlockedFrom && mode == 0) {
    hipLaunchKernelGGL(k_invk_alpha, dim3((unsigned)((n + 3) / 4), q), dim3(256), 0, c->stream, n, npad, (const double*)w.W,
                       (const double*)w.rhs, dalpha);
    if (deferred) {
      if ((rc = ensure(c->invk_img, sizeof(double) * (size_t)q * npad * npad))) return rc;
      hipLaunchKernelGGL(k_pack_full, dim3((unsigned)std::min<size_t>(((size_t)nb * nb * 256 + 255) / 256, 4096), q), dim3(256), 0, c->stream,
                         (const double*)w.W, n, nb, (double*)c->invk_img.p);
      c->invk_img_valid = true;
      SBO_HIP(hipEventRecord(c->ev_w, c->stream));
      hipLaunchKernelGGL((k_cast_alpha<T>), dim3(16), dim3(256), 0, c->stream, (const double*)dalpha, npad, n, q, npad, (T*)c->alpha.p);
      SBO_HIP(hipGetLastError());
      c->factor_todo = true;
      // (the bases' records come back with this synchronisation; a K1i-first model has none to wait for -- r05: the host goes straight on
      // to enqueue the plan while these kernels run, ~30 us of a model change)
      if (eager_basis) {
        SBO_HIP(stream_wait(c, c->stream));
        SBO_HIP(stream_wait(c, c->stream2));
      }
      return SBO_OK;                      // (the verdict on invK comes with the factor: factor_sync)
    }
  }
  if ((rc = factor_chain_enqueue<T>(c, w, mode, c->stream))) return rc;
  hipLaunchKernelGGL((k_cast_alpha<T>), dim3(16), dim3(256), 0, c->stream, (const double*)dalpha, npad, n, q, npad, (T*)c->alpha.p);
  SBO_HIP(hipGetLastError());
  int* hbad = (int*)(c->h_back + 4608);
  SBO_HIP(hipMemcpyAsync(hbad, w.bad, sizeof(int) * q, hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(stream_wait(c, c->stream));
  if (eager_basis) SBO_HIP(stream_wait(c, c->stream2));
  for (int o = 0; o < q; ++o)
    if (hbad[o]) return fail(SBO_E_INVALID, host_invK ? "invK is not positive definite" : "K + sn2 I is not positive definite");
  return SBO_OK;
}

// A grid that arrived after the model: the images of the caller's invK for the K1b tables are packed now, from the upload that
// still sits in the build workspace -- so that the GEMM posterior contracts with invK as given whatever the order of the calls.
int model_pack_invk(sbo_ctx* c) {
  if (c->invk_img_valid) return SBO_OK;
  if (!c->invk_w_valid) return fail(SBO_E_INVALID, "internal: the uploaded invK is gone");
  ModelWork w;
  int rc = model_work(c, true, w);
  if (rc) return rc;
  const int n = c->mc.n, npad = c->mc.npad, q = c->mc.q, nb = npad / 16;
  if ((rc = ensure(c->invk_img, sizeof(double) * (size_t)q * npad * npad))) return rc;
  hipLaunchKernelGGL(k_pack_full, dim3((unsigned)std::min<size_t>(((size_t)nb * nb * 256 + 255) / 256, 4096), q), dim3(256), 0, c->stream,
                     (const double*)w.W, n, nb, (double*)c->invk_img.p);
  SBO_HIP(hipGetLastError());
  c->invk_img_valid = true;
  return SBO_OK;
}

// The deferred factor chain of a caller's invK (see model_build_t): enqueued on stream4 behind the upload of invK (ev_w),
// ended by ev_factor; the positive-definiteness flags land in the pinned block for factor_sync.
int model_factor_enqueue(sbo_ctx* c) {
  if (!c->factor_todo) return SBO_OK;
  c->factor_todo = false;
  ModelWork w;
  int rc = model_work(c, true, w);
  if (rc) return rc;
  SBO_HIP(hipStreamWaitEvent(c->stream4, c->ev_w, 0));
  if ((rc = factor_chain_enqueue<double>(c, w, 0, c->stream4))) return rc;
  SBO_HIP(hipMemcpyAsync(c->h_back + 4864, w.bad, sizeof(int) * c->mc.q, hipMemcpyDeviceToHost, c->stream4));
  SBO_HIP(hipEventRecord(c->ev_factor, c->stream4));
  c->factor_pending = true;
  return SBO_OK;
}

// Re-pack Fpk / alpha of the model dtype from the resident fp64 factor (after an append; mc.n, mc.npad already updated).
template <typename T>
static int model_repack_t(sbo_ctx* c) {
  const ModelConst& mc = c->mc;
  const int n = mc.n, npad = mc.npad, q = mc.q, nb = npad / 16, cap = c->f_cap;
  const size_t ntri = (size_t)nb * (nb + 1) / 2;
  int rc;
  c->fpk_stride = ntri * 4 * 64;
  if ((rc = ensure(c->Fpk, sizeof(T) * ((size_t)q * c->fpk_stride + 512)))) return rc;
  SBO_HIP(hipMemsetAsync(c->Fpk.p, 0, sizeof(T) * ((size_t)q * c->fpk_stride + 512), c->stream));
  hipLaunchKernelGGL((k_pack_factor<T>), dim3((unsigned)std::min<size_t>((ntri * 256 + 255) / 256, 4096), q), dim3(256), 0, c->stream,
                     (const double*)c->Fplain.p, n, cap, (size_t)cap * cap, nb, c->fpk_stride, (T*)c->Fpk.p);
  if ((rc = ensure(c->alpha, sizeof(T) * (size_t)q * npad))) return rc;
  hipLaunchKernelGGL((k_cast_alpha<T>), dim3(16), dim3(256), 0, c->stream, (const double*)c->alpha64.p, c->a_ld, n, q, npad, (T*)c->alpha.p);
  SBO_HIP(hipGetLastError());
  SBO_HIP(hipStreamSynchronize(c->stream));
  return SBO_OK;
}
int model_repack(sbo_ctx* c) { return c->dtype == SBO_F64 ? model_repack_t<double>(c) : model_repack_t<float>(c); }

// kvec [q][n] cross-covariances of the new point, kappa[q] = sf2 + sn2, rho[q] = y_norm_new - mp; appends row n.
int model_append(sbo_ctx* c, const std::vector<double>& kvec, const double* kappa, const double* rho) {
  const int n = c->mc.n, q = c->mc.q;
  int rc;
  if (n + 1 > c->f_cap || c->a_ld != c->f_cap) {
    // a freshly built model keeps its factor tight (leading dimension n): make room for 256 more rows
    const int cap = std::min(SBO_MAX_N, (c->mc.npad + 256 + 127) / 128 * 128);
    if (n + 1 > cap) return fail(SBO_E_UNSUPPORTED, "model is at its capacity: rebuild it with sbo_model_set");
    DevBuf F2, a2;
    if ((rc = ensure(F2, sizeof(double) * (size_t)q * cap * cap))) return rc;
    if ((rc = ensure(a2, sizeof(double) * (size_t)q * cap))) { release(F2); return rc; }
    hipError_t e = hipMemsetAsync(F2.p, 0, sizeof(double) * (size_t)q * cap * cap, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(a2.p, 0, sizeof(double) * (size_t)q * cap, c->stream);
    for (int o = 0; o < q && e == hipSuccess; ++o) {
      e = hipMemcpy2DAsync((double*)F2.p + (size_t)o * cap * cap, sizeof(double) * cap, (const double*)c->Fplain.p + (size_t)o * c->f_cap * c->f_cap,
                           sizeof(double) * c->f_cap, sizeof(double) * n, n, hipMemcpyDeviceToDevice, c->stream);
      if (e == hipSuccess)
        e = hipMemcpyAsync((double*)a2.p + (size_t)o * cap, (const double*)c->alpha64.p + (size_t)o * c->a_ld, sizeof(double) * n,
                           hipMemcpyDeviceToDevice, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { release(F2); release(a2); return hip_fail(e, "growing the resident factor"); }
    release(c->Fplain);
    release(c->alpha64);
    c->Fplain = F2;
    c->alpha64 = a2;
    c->f_cap = cap;
    c->a_ld = cap;
  }
  const int cap = c->f_cap;
  if ((rc = ensure(c->mwork, sizeof(double) * ((size_t)q * n + 2 * (size_t)q + 2 * (size_t)q * cap) + sizeof(int) * q))) return rc;
  double* dk = (double*)c->mwork.p;
  double* dkappa = dk + (size_t)q * n;
  double* drho = dkappa + q;
  double* dscratch = drho + q;
  int* dbad = (int*)(dscratch + 2 * (size_t)q * cap);
  SBO_HIP(hipMemcpyAsync(dk, kvec.data(), sizeof(double) * (size_t)q * n, hipMemcpyHostToDevice, c->stream));
  SBO_HIP(hipMemcpyAsync(dkappa, kappa, sizeof(double) * q, hipMemcpyHostToDevice, c->stream));
  SBO_HIP(hipMemcpyAsync(drho, rho, sizeof(double) * q, hipMemcpyHostToDevice, c->stream));
  SBO_HIP(hipMemsetAsync(dbad, 0, sizeof(int) * q, c->stream));
  hipLaunchKernelGGL(k_model_append, dim3(q), dim3(1024), 0, c->stream, n, cap, (double*)c->Fplain.p, (double*)c->alpha64.p,
                     (const double*)dk, (const double*)dkappa, (const double*)drho, dscratch, dbad);
  SBO_HIP(hipGetLastError());
  std::vector<int> hbad(q, 0);
  SBO_HIP(hipMemcpyAsync(hbad.data(), dbad, sizeof(int) * q, hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipStreamSynchronize(c->stream));
  for (int o = 0; o < q; ++o)
    if (hbad[o]) return fail(SBO_E_INVALID, "appended observation makes K singular (duplicate point without noise?)");
  return SBO_OK;
}

int model_build(sbo_ctx* c, const double* const* host_invK, const double* X_norm, const double* Y_norm) {
  const int rc = c->dtype == SBO_F64 ? model_build_t<double>(c, host_invK, X_norm, Y_norm) : model_build_t<float>(c, host_invK, X_norm, Y_norm);
  if (rc != SBO_OK) drain_streams(c);     // (the bases may still be running on the second stream)
  return rc;
}

}  // namespace sbo
