// C shim over bilinear_host.hpp for the CPU unit test (tests/test_bilinear_host.py builds it with g++; not part of libsafebo.so)
#include "bilinear_host.hpp"
#include <cstring>

extern "C" {

int blt_axis(int n, const double* As_col, double vinv, const double* xn, int count, int* r, int* rc, double* U, double* S) {
  sbo::bl::AxisBasis b;
  if (!sbo::bl::axis_basis(n, As_col, vinv, xn, count, b)) return 1;
  *r = b.r;
  *rc = b.rc;
  memcpy(U, b.U.data(), sizeof(double) * b.U.size());
  memcpy(S, b.S.data(), sizeof(double) * b.S.size());
  return 0;
}

int blt_forms(int n, const double* M, int r0, const double* U0, int r1, const double* U1, double scale, int nbeta,
              const double* betas, double beta_scale, double* T4qq, double* Mb) {
  sbo::bl::AxisBasis b0, b1;
  b0.r = r0; b0.U.assign(U0, U0 + (size_t)n * r0);
  b1.r = r1; b1.U.assign(U1, U1 + (size_t)n * r1);
  std::vector<const double*> bp(nbeta);
  for (int i = 0; i < nbeta; ++i) bp[i] = betas + (size_t)i * n;
  std::vector<double> T, Mbv;
  sbo::bl::build_forms(n, M, b0, b1, scale, nbeta, bp.data(), beta_scale, T, Mbv);
  memcpy(T4qq, T.data(), sizeof(double) * T.size());
  memcpy(Mb, Mbv.data(), sizeof(double) * Mbv.size());
  return 0;
}

int blt_pairs(int r, int count, const double* S, double* P) {
  sbo::bl::AxisBasis b;
  b.r = r;
  b.S.assign(S, S + (size_t)r * count);
  std::vector<double> Pv;
  sbo::bl::pair_table(b, count, Pv);
  memcpy(P, Pv.data(), sizeof(double) * Pv.size());
  return 0;
}
}
