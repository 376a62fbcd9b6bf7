// bilinear_host.hpp -- host-side numerics of the bilinear (reduced-basis) posterior on 2-D tensor grids.
// Pure C++ (no HIP) so that it can be exercised on a CPU-only machine.
//
// On a tensor grid the RBF-ARD cross-covariance separates, k_j(x0, x1) = sf2 e0_j(x0) e1_j(x1), and each factor family
// {e_j(.)}_j is a set of Gaussians restricted to the axis interval: numerically it spans only r ~ 20-50 directions, however
// many observations there are.  With e0 = U0 S0(x0), e1 = U1 S1(x1) (U orthonormal n x r, S the r coordinates of a grid
// position) the posterior of models/GP_Safe.py:342-343 becomes
//     k^T invK k      = sf2^2 sum_{pp',ss'} T4[pp',ss'] (S0_p S0_p')(x0) (S1_s S1_s')(x1)        T4 = Z^T invK Z,
//     k^T alpha       = sf2   sum_{p,s}     Ma[p,s]     S0_p(x0) S1_s(x1)                        Z_j,(p,s) = U0_jp U1_js
// i.e. two dense GEMMs whose inner dimension r(r+1)/2 does not depend on n.  The bases come from a Chebyshev
// interpolant of each Gaussian on the axis interval (coefficients to below 1e-16) followed by an SVD of the
// coefficient matrix; directions with singular value <= 1e-16 sigma_max are dropped.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <thread>
#include <vector>

namespace sbo {
namespace bl {

constexpr double kPi = 3.14159265358979323846;
constexpr int kMaxCheb = 128;    // Chebyshev degree limit of the interpolant (beyond: the separable-table kernel is used)
constexpr int kMaxRank = 64;     // basis size limit per axis (inner dimension of the GEMMs: r (r + 1) / 2 <= 2080)

// One-sided Jacobi SVD (Hestenes) of G [rows x cols], column-major (column c at G + c*rows).  On return the columns of G
// are U_c sigma_c (mutually orthogonal), V [cols x cols] column-major holds the right singular vectors, sig the norms.
inline void jacobi_svd(double* G, int rows, int cols, std::vector<double>& V, std::vector<double>& sig) {
  V.assign((size_t)cols * cols, 0.0);
  for (int c = 0; c < cols; ++c) V[(size_t)c * cols + c] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < cols - 1; ++p) {
      double* gp = G + (size_t)p * rows;
      for (int q = p + 1; q < cols; ++q) {
        double* gq = G + (size_t)q * rows;
        double a = 0, b = 0, g = 0;
        for (int i = 0; i < rows; ++i) { a += gp[i] * gp[i]; b += gq[i] * gq[i]; g += gp[i] * gq[i]; }
        if (g == 0.0 || std::fabs(g) <= 1e-16 * std::sqrt(a * b)) continue;
        rotated = true;
        const double zeta = (b - a) / (2.0 * g);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < rows; ++i) {
          const double x = gp[i], y = gq[i];
          gp[i] = cs * x - sn * y;
          gq[i] = sn * x + cs * y;
        }
        double* vp = &V[(size_t)p * cols];
        double* vq = &V[(size_t)q * cols];
        for (int i = 0; i < cols; ++i) {
          const double x = vp[i], y = vq[i];
          vp[i] = cs * x - sn * y;
          vq[i] = sn * x + cs * y;
        }
      }
    }
    if (!rotated) break;
  }
  sig.assign(cols, 0.0);
  for (int c = 0; c < cols; ++c) {
    double a = 0;
    for (int i = 0; i < rows; ++i) a += G[(size_t)c * rows + i] * G[(size_t)c * rows + i];
    sig[c] = std::sqrt(a);
  }
}

struct AxisBasis {
  int r = 0, rc = 0;
  double a = 0.0, b = 0.0; // interval of the axis (normalised coordinates) the Chebyshev series lives on
  std::vector<double> U;   // [n x r] column-major, orthonormal columns
  std::vector<double> S;   // [r x count] row-major: coordinates of every grid position of the axis (when tabulated)
  std::vector<double> Vs;  // [r x rc]: Chebyshev coefficients of coordinate p;  S[p][i] = sig[p] sum_c Vs[p][c] T_c(xi_i)
  std::vector<double> sig; // [r]
};

// Basis of the family f_j(xn) = exp(-1/2 (xn vinv - As_j)^2), j < n, on the positions xn[0..count).
// Returns false when the interpolant does not converge within kMaxCheb terms or the rank exceeds kMaxRank.
// tabulate = false leaves S empty: the caller evaluates the series itself (the device build does, from Vs / sig / a / b).
inline bool axis_basis(int n, const double* As_col, double vinv, const double* xn, int count, AxisBasis& out,
                       bool tabulate = true) {
  double a = xn[0], b = xn[0];
  for (int i = 1; i < count; ++i) { a = std::min(a, xn[i]); b = std::max(b, xn[i]); }
  if (!(b > a)) return false;
  std::vector<double> coef;
  int rc = 0;
  for (int trial : {32, 48, 64, 96, kMaxCheb}) {
    rc = trial;
    // samples at the Chebyshev points of the first kind, coefficients by the discrete cosine sum
    std::vector<double> fn((size_t)n * rc), cth((size_t)rc * rc);
    for (int k = 0; k < rc; ++k) {
      const double th = kPi * (k + 0.5) / rc;
      const double x = 0.5 * (std::cos(th) * (b - a) + (a + b));
      for (int j = 0; j < n; ++j) {
        const double df = x * vinv - As_col[j];
        fn[(size_t)j * rc + k] = std::exp(-0.5 * (df * df));
      }
      for (int p = 0; p < rc; ++p) cth[(size_t)p * rc + k] = std::cos(p * th);
    }
    coef.assign((size_t)rc * n, 0.0);   // column-major [n x rc]: column p = coefficient p of every f_j
    double tail = 0.0;
    // the last four coefficients first: a trial that has not converged is dropped without the other rc - 4 sums
    for (int pass = 0; pass < 2; ++pass) {
      const int p0 = pass == 0 ? rc - 4 : 0, p1 = pass == 0 ? rc : rc - 4;
      for (int p = p0; p < p1; ++p)
        for (int j = 0; j < n; ++j) {
          double s = 0;
          for (int k = 0; k < rc; ++k) s += fn[(size_t)j * rc + k] * cth[(size_t)p * rc + k];
          s *= (p == 0 ? 1.0 : 2.0) / rc;
          coef[(size_t)p * n + j] = s;
          if (p >= rc - 4) tail = std::max(tail, std::fabs(s));
        }
      if (pass == 0 && tail > 1e-14) break;
    }
    if (tail <= 1e-14) break;   // (the computed coefficients bottom out at ~1e-15; they fall by > 10x per term there)
    if (trial == kMaxCheb) return false;
  }
  // SVD of the coefficient matrix.  Tall case: reduce with Householder QR first, rotate the small triangular factor.
  std::vector<double> V, sig;
  std::vector<double> Ucols;   // [n x rc] column-major, columns = U_c sigma_c
  if (n > rc) {
    std::vector<double> A = coef;                       // overwritten by the reflectors / R
    std::vector<double> beta(rc, 0.0);
    for (int k = 0; k < rc; ++k) {
      double* ck = &A[(size_t)k * n];
      double nrm = 0;
      for (int i = k; i < n; ++i) nrm += ck[i] * ck[i];
      nrm = std::sqrt(nrm);
      if (nrm == 0.0) continue;
      const double alpha = ck[k] >= 0 ? -nrm : nrm;
      const double v0 = ck[k] - alpha;
      // v = (v0, ck[k+1..]) ; H = I - beta v v^T, beta = 2 / (v^T v)
      double vtv = v0 * v0;
      for (int i = k + 1; i < n; ++i) vtv += ck[i] * ck[i];
      beta[k] = vtv > 0 ? 2.0 / vtv : 0.0;
      for (int c2 = k + 1; c2 < rc; ++c2) {
        double* cc = &A[(size_t)c2 * n];
        double s = v0 * cc[k];
        for (int i = k + 1; i < n; ++i) s += ck[i] * cc[i];
        s *= beta[k];
        cc[k] -= s * v0;
        for (int i = k + 1; i < n; ++i) cc[i] -= s * ck[i];
      }
      ck[k] = alpha;
      // keep v below the diagonal (normalised so that the stored part is v / 1 with v0 remembered separately)
      // store v0 in a side array by scaling: v := v / v0 (v0 != 0 by construction), beta := beta v0^2
      if (v0 != 0.0) {
        for (int i = k + 1; i < n; ++i) ck[i] /= v0;
        beta[k] *= v0 * v0;
      }
    }
    std::vector<double> R((size_t)rc * rc, 0.0);        // column-major upper triangle
    for (int c2 = 0; c2 < rc; ++c2)
      for (int i = 0; i <= c2; ++i) R[(size_t)c2 * rc + i] = A[(size_t)c2 * n + i];
    jacobi_svd(R.data(), rc, rc, V, sig);
    // U sigma = Q (R V): apply the reflectors H_1 .. H_rc (in reverse order) to the padded columns
    Ucols.assign((size_t)rc * n, 0.0);
    for (int c2 = 0; c2 < rc; ++c2)
      for (int i = 0; i < rc; ++i) Ucols[(size_t)c2 * n + i] = R[(size_t)c2 * rc + i];
    for (int k = rc - 1; k >= 0; --k) {
      if (beta[k] == 0.0) continue;
      const double* ck = &A[(size_t)k * n];
      for (int c2 = 0; c2 < rc; ++c2) {
        double* uc = &Ucols[(size_t)c2 * n];
        double s = uc[k];
        for (int i = k + 1; i < n; ++i) s += ck[i] * uc[i];
        s *= beta[k];
        uc[k] -= s;
        for (int i = k + 1; i < n; ++i) uc[i] -= s * ck[i];
      }
    }
  } else {
    Ucols = coef;
    jacobi_svd(Ucols.data(), n, rc, V, sig);
  }
  std::vector<int> order(rc);
  for (int c2 = 0; c2 < rc; ++c2) order[c2] = c2;
  std::sort(order.begin(), order.end(), [&](int x, int y) { return sig[x] > sig[y]; });
  const double smax = sig[order[0]];
  int r = 0;
  while (r < rc && sig[order[r]] > 1e-16 * smax) ++r;
  if (r < 1 || r > kMaxRank) return false;
  out.r = r;
  out.rc = rc;
  out.U.assign((size_t)n * r, 0.0);
  for (int p = 0; p < r; ++p) {
    const int c2 = order[p];
    const double inv = 1.0 / sig[c2];
    for (int j = 0; j < n; ++j) out.U[(size_t)p * n + j] = Ucols[(size_t)c2 * n + j] * inv;
  }
  out.a = a;
  out.b = b;
  out.Vs.assign((size_t)r * rc, 0.0);
  out.sig.assign(r, 0.0);
  for (int p = 0; p < r; ++p) {
    out.sig[p] = sig[order[p]];
    for (int c2 = 0; c2 < rc; ++c2) out.Vs[(size_t)p * rc + c2] = V[(size_t)order[p] * rc + c2];
  }
  out.S.clear();
  if (!tabulate) return true;
  // S[p][i] = sigma_p sum_c V[c][p] T_c(xi_i), Chebyshev values by the three-term recurrence
  out.S.assign((size_t)r * count, 0.0);
  std::vector<double> Tc(rc);
  for (int i = 0; i < count; ++i) {
    double xi = (2.0 * xn[i] - (a + b)) / (b - a);
    xi = std::max(-1.0, std::min(1.0, xi));
    Tc[0] = 1.0;
    if (rc > 1) Tc[1] = xi;
    for (int c2 = 2; c2 < rc; ++c2) Tc[c2] = 2.0 * xi * Tc[c2 - 1] - Tc[c2 - 2];
    for (int p = 0; p < r; ++p) {
      const double* vp = &V[(size_t)order[p] * rc];
      double s = 0;
      for (int c2 = 0; c2 < rc; ++c2) s += vp[c2] * Tc[c2];
      out.S[(size_t)p * count + i] = sig[order[p]] * s;
    }
  }
  return true;
}

// run body(first, last) over [0, total) on a few host threads (the T4 build is ~1.4e8 multiply-adds at n = 512)
template <typename F>
inline void parallel_ranges(int total, F body) {
  unsigned hw = std::thread::hardware_concurrency();
  const int nt = (int)std::max(1u, std::min(hw ? hw : 1u, 8u));
  if (nt == 1 || total < 2) { body(0, total); return; }
  std::vector<std::thread> th;
  const int chunk = (total + nt - 1) / nt;
  for (int t = 0; t < nt; ++t) {
    const int a = t * chunk, b = std::min(total, a + chunk);
    if (a < b) th.emplace_back([=]() { body(a, b); });
  }
  for (auto& x : th) x.join();
}

// symmetric pair index: pairs (p <= p') enumerated row by row; K = r (r + 1) / 2
inline int pair_count(int r) { return r * (r + 1) / 2; }

// T4qq [K0 x K1] row-major (symmetrised over p <-> p', scaled by `scale`), Mb[(1 + d)][r0 x r1] row-major.
//   M     : [n x n] row-major lower-triangular factor with M^T M = invK (or L^-1)
//   beta  : (1 + d) weight vectors of length n  (alpha, alpha * Xn_0, alpha * Xn_1 ..)
inline void build_forms(int n, const double* M, const AxisBasis& b0, const AxisBasis& b1, double scale, int nbeta,
                        const double* const* beta, double beta_scale, std::vector<double>& T4qq, std::vector<double>& Mb) {
  const int r0 = b0.r, r1 = b1.r, R = r0 * r1;
  // C = M Z, Z_j,(p,s) = U0_jp U1_js ; stored column-major [n x R]
  std::vector<double> Z((size_t)n * R), C((size_t)n * R, 0.0);
  for (int p = 0; p < r0; ++p)
    for (int s = 0; s < r1; ++s) {
      double* z = &Z[(size_t)(p * r1 + s) * n];
      for (int j = 0; j < n; ++j) z[j] = b0.U[(size_t)p * n + j] * b1.U[(size_t)s * n + j];
    }
  parallel_ranges(R, [&](int c_lo, int c_hi) {
    for (int c = c_lo; c < c_hi; ++c) {
      const double* z = &Z[(size_t)c * n];
      double* cc = &C[(size_t)c * n];
      for (int i = 0; i < n; ++i) {
        const double* mi = M + (size_t)i * n;
        double s = 0;
        for (int j = 0; j <= i; ++j) s += mi[j] * z[j];
        cc[i] = s;
      }
    }
  });
  // G = C^T C (upper half), G[(p,s),(p',s')]
  std::vector<double> G((size_t)R * R, 0.0);
  // (rows interleaved over the threads: row c has R - c entries)
  parallel_ranges(8, [&](int t_lo, int t_hi) {
    for (int t = t_lo; t < t_hi; ++t)
      for (int c = t; c < R; c += 8)
        for (int c2 = c; c2 < R; ++c2) {
          const double* x = &C[(size_t)c * n];
          const double* y = &C[(size_t)c2 * n];
          double s = 0;
          for (int i = 0; i < n; ++i) s += x[i] * y[i];
          G[(size_t)c * R + c2] = s;
          G[(size_t)c2 * R + c] = s;
        }
  });
  const int K0 = pair_count(r0), K1 = pair_count(r1);
  T4qq.assign((size_t)K0 * K1, 0.0);
  int k0 = 0;
  for (int p = 0; p < r0; ++p)
    for (int pp = p; pp < r0; ++pp, ++k0) {
      int k1 = 0;
      for (int s = 0; s < r1; ++s)
        for (int ss = s; ss < r1; ++ss, ++k1) {
          // 1/2 (T[p,s,p',s'] + T[p',s,p,s']) -- the (p,s) <-> (p',s') symmetry of G covers the other two arrangements
          const double t = 0.5 * (G[(size_t)(p * r1 + s) * R + (pp * r1 + ss)] + G[(size_t)(pp * r1 + s) * R + (p * r1 + ss)]);
          T4qq[(size_t)k0 * K1 + k1] = scale * t;
        }
    }
  Mb.assign((size_t)nbeta * R, 0.0);
  for (int bi = 0; bi < nbeta; ++bi)
    for (int c = 0; c < R; ++c) {
      const double* z = &Z[(size_t)c * n];
      double s = 0;
      for (int j = 0; j < n; ++j) s += beta[bi][j] * z[j];
      Mb[(size_t)bi * R + c] = beta_scale * s;
    }
}

// Mb[b][p r1 + s] = beta_scale sum_j beta_b[j] U0_jp U1_js   (the bilinear forms of the mean and its gradient sums)
inline void mean_forms(int n, const AxisBasis& b0, const AxisBasis& b1, int nbeta, const double* const* beta, double beta_scale,
                       std::vector<double>& Mb) {
  const int r0 = b0.r, r1 = b1.r, R = r0 * r1;
  Mb.assign((size_t)nbeta * R, 0.0);
  for (int bi = 0; bi < nbeta; ++bi)
    for (int p = 0; p < r0; ++p)
      for (int s = 0; s < r1; ++s) {
        const double* u0 = &b0.U[(size_t)p * n];
        const double* u1 = &b1.U[(size_t)s * n];
        double acc = 0;
        for (int j = 0; j < n; ++j) acc += beta[bi][j] * u0[j] * u1[j];
        Mb[(size_t)bi * R + (size_t)p * r1 + s] = beta_scale * acc;
      }
}

// (p, p') of every pair index k, pairs enumerated row by row
inline void pair_map(int r, std::vector<int>& map) {
  map.clear();
  for (int p = 0; p < r; ++p)
    for (int pp = p; pp < r; ++pp) { map.push_back(p); map.push_back(pp); }
}

// P[(p <= p')][i] = w S_p(i) S_p'(i), w = 1 on the diagonal and 2 off it; row-major [K x count]
inline void pair_table(const AxisBasis& b, int count, std::vector<double>& P) {
  const int r = b.r, K = pair_count(r);
  P.assign((size_t)K * count, 0.0);
  int k = 0;
  for (int p = 0; p < r; ++p)
    for (int pp = p; pp < r; ++pp, ++k) {
      const double w = p == pp ? 1.0 : 2.0;
      const double* sp = &b.S[(size_t)p * count];
      const double* sq = &b.S[(size_t)pp * count];
      double* dst = &P[(size_t)k * count];
      for (int i = 0; i < count; ++i) dst[i] = w * sp[i] * sq[i];
    }
}

}  // namespace bl
}  // namespace sbo
