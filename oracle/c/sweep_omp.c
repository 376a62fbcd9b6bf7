/* CPU restatement of the SafeOpt candidate sweep in C + OpenMP -- TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/__init__.py:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/).
 *
 * Same formulation as oracle/gp_oracle.py, which restates the reference line by line:
 *   distance   models/GP_Safe.py:98-120   expanded form  -2 Xa.Ya^T + sum Xa^2 + sum Ya^2
 *   covariance models/GP_Safe.py:143-167  sf2 exp(-1/2 dist)
 *   posterior  models/GP_Safe.py:310-352  mean = mp + (k^T invK)(Y - mp),  var = max(0, sf2 - (k^T invK) k), un-normalised
 *   bounds     models/SafeOpt.py:34-45    mean -/+ b sqrt(var)
 *   sets       models/SafeOpt.py:47-66    S_t = all lcb_c >= 0, u* = min_S ucb_0, M_t = S & lcb_0 <= u*, arg-max var_0 on M
 * What differs from NumPy is the order of the sums inside k^T invK (a plain loop over the observations here, a BLAS
 * kernel there): results agree to rounding (tests/test_oracle.py), not bit for bit.
 *
 * Candidates are the points of an implicit grid (axis 0 fastest, last point of an axis = hi exactly), processed in
 * blocks of CB columns: K block [n][CB], T = invK^T K as n rank-1 updates per row (the inner loop runs over the CB
 * candidates, so it vectorises without reassociating any sum).  PARITY UNPINNED like the rest of the oracle.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

#define CB 32
#define MAXD 8

typedef struct {
  int n, d, q;
  const double* X_norm;   /* [n][d] */
  const double* Y_norm;   /* [n][q] */
  const double* invK;     /* [q][n][n] */
  const double* hyp;      /* [d + 2][q]  log ell_a, log sf, log sn */
  const double* X_mean;   /* [d] */
  const double* X_std;
  const double* Y_mean;   /* [q] */
  const double* Y_std;
} sweep_model;

int sweep_omp_threads(void) { return omp_get_max_threads(); }
void sweep_omp_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

/* mean / var of candidates [first, first + N) of the grid; out arrays [N][q] (NULL: not stored).
 * S / U / M masks (NULL: not stored), result: res[0] = |S|, res[1] = |M|, res[2] = arg-max index (-1: none), ustar. */
int sweep_omp_safeopt(const sweep_model* m, const double* lo, const double* hi, const long long* count, long long first, long long N,
                      double b, double* mean_out, double* var_out, uint8_t* S_out, uint8_t* U_out, uint8_t* M_out, long long* res,
                      double* ustar_out) {
  const int n = m->n, d = m->d, q = m->q;
  if (d > MAXD || q > 8) return 1;
  double step[MAXD];
  for (int a = 0; a < d; ++a) step[a] = count[a] > 1 ? (hi[a] - lo[a]) / (double)(count[a] - 1) : 0.0;
  /* per-output constants and scaled observations */
  double* Xa = (double*)malloc(sizeof(double) * (size_t)q * n * d);
  double* sqa = (double*)malloc(sizeof(double) * (size_t)q * n);
  double* rhs = (double*)malloc(sizeof(double) * (size_t)q * n);
  double ell[8][MAXD], sf2[8], mp[8];
  for (int i = 0; i < q; ++i) {
    for (int a = 0; a < d; ++a) ell[i][a] = exp(2.0 * m->hyp[(size_t)a * q + i]);
    sf2[i] = exp(2.0 * m->hyp[(size_t)d * q + i]);
    mp[i] = i == 0 ? 0.0 : (-2.0 * m->Y_mean[i]) / m->Y_std[i];          /* models/GP_Safe.py:331-332 */
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
      for (int a = 0; a < d; ++a) {
        const double v = m->X_norm[(size_t)j * d + a] * pow(ell[i][a], -0.5);
        Xa[((size_t)i * n + j) * d + a] = v;
        s += v * v;
      }
      sqa[(size_t)i * n + j] = s;
      rhs[(size_t)i * n + j] = m->Y_norm[(size_t)j * q + i] - mp[i];
    }
  }
  double* lcb0 = (double*)malloc(sizeof(double) * (size_t)N);
  double* ucb0 = (double*)malloc(sizeof(double) * (size_t)N);
  double* var0 = (double*)malloc(sizeof(double) * (size_t)N);
  uint8_t* Sm = S_out ? S_out : (uint8_t*)malloc((size_t)N);
  double ustar = INFINITY;
  long long cS = 0;
#pragma omp parallel
  {
    double* K = (double*)malloc(sizeof(double) * (size_t)n * CB);
    double* T = (double*)malloc(sizeof(double) * (size_t)n * CB);
    double ul = INFINITY;
    long long cl = 0;
#pragma omp for schedule(dynamic, 64)
    for (long long blk = 0; blk < (N + CB - 1) / CB; ++blk) {
      const long long g0 = blk * CB;
      const int nc = (int)(N - g0 < CB ? N - g0 : CB);
      double xn[CB][MAXD];
      for (int c = 0; c < nc; ++c) {
        long long f = first + g0 + c;
        for (int a = 0; a < d; ++a) {
          const long long i = f % count[a];
          f /= count[a];
          const double x = (i == count[a] - 1 && count[a] > 1) ? hi[a] : lo[a] + (double)i * step[a];
          xn[c][a] = (x - m->X_mean[a]) / m->X_std[a];                     /* models/GP_Safe.py:326 */
        }
      }
      uint8_t s_all[CB], u_all[CB];
      for (int c = 0; c < nc; ++c) { s_all[c] = 1; u_all[c] = 1; }
      double l0[CB], u0[CB], v0[CB];
      for (int i = 0; i < q; ++i) {
        double ya[CB][MAXD], sqy[CB];
        for (int c = 0; c < nc; ++c) {
          double s = 0.0;
          for (int a = 0; a < d; ++a) { ya[c][a] = xn[c][a] * pow(ell[i][a], -0.5); s += ya[c][a] * ya[c][a]; }
          sqy[c] = s;
        }
        for (int j = 0; j < n; ++j) {
          const double* xj = Xa + ((size_t)i * n + j) * d;
          for (int c = 0; c < nc; ++c) {
            double dot = 0.0;
            for (int a = 0; a < d; ++a) dot += xj[a] * ya[c][a];
            const double dist = -2.0 * dot + sqa[(size_t)i * n + j] + sqy[c];   /* :119 */
            K[(size_t)j * CB + c] = sf2[i] * exp(-0.5 * dist);                 /* :165-166 */
          }
          for (int c = nc; c < CB; ++c) K[(size_t)j * CB + c] = 0.0;
        }
        memset(T, 0, sizeof(double) * (size_t)n * CB);
        const double* iK = m->invK + (size_t)i * n * n;
        for (int l = 0; l < n; ++l) {                                        /* T[j][c] = sum_l invK[l][j] K[l][c] */
          const double* kl = K + (size_t)l * CB;
          for (int j = 0; j < n; ++j) {
            const double a_ = iK[(size_t)l * n + j];
            double* tj = T + (size_t)j * CB;
#pragma omp simd
            for (int c = 0; c < CB; ++c) tj[c] += a_ * kl[c];
          }
        }
        double ms[CB], qs[CB];
        for (int c = 0; c < CB; ++c) { ms[c] = 0.0; qs[c] = 0.0; }
        for (int j = 0; j < n; ++j) {
          const double r = rhs[(size_t)i * n + j];
          const double* tj = T + (size_t)j * CB;
          const double* kj = K + (size_t)j * CB;
#pragma omp simd
          for (int c = 0; c < CB; ++c) { ms[c] += tj[c] * r; qs[c] += tj[c] * kj[c]; }
        }
        for (int c = 0; c < nc; ++c) {
          double v = sf2[i] - qs[c];                                          /* :343 */
          v = v > 0.0 ? v : 0.0;
          const double mean = (mp[i] + ms[c]) * m->Y_std[i] + m->Y_mean[i];   /* :342, :346 */
          const double var = v * m->Y_std[i] * m->Y_std[i];                   /* :347 */
          if (mean_out) mean_out[(size_t)(g0 + c) * q + i] = mean;
          if (var_out) var_out[(size_t)(g0 + c) * q + i] = var;
          const double sd = b * sqrt(var);
          const double lcb = mean - sd, ucb = mean + sd;                      /* models/SafeOpt.py:37, 43 */
          if (i == 0) { l0[c] = lcb; u0[c] = ucb; v0[c] = var; }
          else { s_all[c] &= lcb >= 0.0; u_all[c] &= lcb <= 0.0; }           /* :57-59, :73-77 */
        }
      }
      for (int c = 0; c < nc; ++c) {
        Sm[g0 + c] = s_all[c];
        if (U_out) U_out[g0 + c] = u_all[c];
        lcb0[g0 + c] = l0[c]; ucb0[g0 + c] = u0[c]; var0[g0 + c] = v0[c];
        if (s_all[c]) { ++cl; if (u0[c] < ul) ul = u0[c]; }
      }
    }
#pragma omp critical
    { if (ul < ustar) ustar = ul; cS += cl; }
    free(K);
    free(T);
  }
  /* M_t and the arg-max of var_0 over it (first maximum wins, as numpy.argmax) */
  long long cM = 0, best = -1;
  double bv = -INFINITY;
#pragma omp parallel
  {
    long long cl = 0, bl = -1;
    double bvl = -INFINITY;
#pragma omp for schedule(static)
    for (long long g = 0; g < N; ++g) {
      const uint8_t mm = Sm[g] && lcb0[g] <= ustar;                           /* models/SafeOpt.py:62 */
      if (M_out) M_out[g] = mm;
      if (mm) { ++cl; if (var0[g] > bvl) { bvl = var0[g]; bl = g; } }
    }
#pragma omp critical
    { cM += cl; if (bl >= 0 && (bvl > bv || (bvl == bv && bl < best))) { bv = bvl; best = bl; } }
  }
  res[0] = cS; res[1] = cM; res[2] = best >= 0 ? first + best : -1;
  *ustar_out = ustar;
  if (!S_out) free(Sm);
  free(lcb0); free(ucb0); free(var0); free(Xa); free(sqa); free(rhs);
  return 0;
}
