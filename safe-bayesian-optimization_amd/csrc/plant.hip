// plant.hip -- SURVEY.md section 8(f) rank 4: the William-Otto reactor plant of the reference, batched on the device.
//
// Reference: problems/WilliamOttoReactor_Problem.py:19-45 (six steady-state mass balances of the CSTR), :47-93 (objective
// and the two constraints at the root found by scipy.optimize.fsolve from x0 = 0.1).  The reference evaluates one point per
// call inside Python loops (its 100 x 100 contour table takes 10 000 fsolve calls, utils/utils_WilliamOttoReactor.py:24-55);
// here every input row gets one thread running a damped Newton iteration on the analytic 6 x 6 Jacobian in registers.
#include <cmath>
#include "internal.hpp"

namespace sbo {

__device__ __forceinline__ void wo_balances(const double* w, double Fb, double k1, double k2, double k3, double* f, double (*J)[6]) {
  constexpr double FA = 1.8275, VR = 2105.2;
  const double xa = w[0], xb = w[1], xc = w[2], xp = w[3], xe = w[4], xg = w[5];
  const double Fr = FA + Fb, a = -Fr / VR;
  f[0] = (FA - Fr * xa - VR * xa * xb * k1) / VR;
  f[1] = (Fb - Fr * xb - VR * xa * xb * k1 - VR * xb * xc * k2) / VR;
  f[2] = -Fr * xc / VR + 2 * xa * xb * k1 - 2 * xb * xc * k2 - xc * xp * k3;
  f[3] = -Fr * xp / VR + xb * xc * k2 - 0.5 * xp * xc * k3;
  f[4] = -Fr * xe / VR + 2 * xb * xc * k2;
  f[5] = -Fr * xg / VR + 1.5 * xp * xc * k3;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < 6; ++j) J[i][j] = 0.0;
  J[0][0] = a - xb * k1;  J[0][1] = -xa * k1;
  J[1][0] = -xb * k1;     J[1][1] = a - xa * k1 - xc * k2;      J[1][2] = -xb * k2;
  J[2][0] = 2 * xb * k1;  J[2][1] = 2 * xa * k1 - 2 * xc * k2;  J[2][2] = a - 2 * xb * k2 - xp * k3;  J[2][3] = -xc * k3;
  J[3][1] = xc * k2;      J[3][2] = xb * k2 - 0.5 * xp * k3;    J[3][3] = a - 0.5 * xc * k3;
  J[4][1] = 2 * xc * k2;  J[4][2] = 2 * xb * k2;                J[4][4] = a;
  J[5][2] = 1.5 * xp * k3; J[5][3] = 1.5 * xc * k3;             J[5][5] = a;
}

// out[i] = (objective, constraint 1, constraint 2) for u[i] = (Fb, Tr), noise-free
__global__ __launch_bounds__(64) void k_plant_wo(const double* __restrict__ u, long long n, double* __restrict__ out) {
  constexpr double FA = 1.8275;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double Fb = u[2 * i], Tr = u[2 * i + 1];
  const double k1 = 1.6599e6 * exp(-6666.7 / (Tr + 273));
  const double k2 = 7.2177e8 * exp(-8333.3 / (Tr + 273));
  const double k3 = 2.6745e12 * exp(-11111.0 / (Tr + 273));
  double w[6] = {0.1, 0.1, 0.1, 0.1, 0.1, 0.1};                 // fsolve's start, WilliamOttoReactor_Problem.py:49
  for (int it = 0; it < 60; ++it) {
    double f[6], J[6][6], s[6];
    wo_balances(w, Fb, k1, k2, k3, f, J);
    // solve J s = -f: Gaussian elimination with partial pivoting (fully unrolled, everything in registers)
#pragma unroll
    for (int r = 0; r < 6; ++r) s[r] = -f[r];
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      int piv = c;
      double best = fabs(J[c][c]);
#pragma unroll
      for (int r = c + 1; r < 6; ++r) {
        const double v = fabs(J[r][c]);
        if (v > best) { best = v; piv = r; }
      }
#pragma unroll
      for (int r = c + 1; r < 6; ++r) {
        if (r == piv) {
#pragma unroll
          for (int k = 0; k < 6; ++k) { const double t = J[c][k]; J[c][k] = J[r][k]; J[r][k] = t; }
          const double t = s[c]; s[c] = s[r]; s[r] = t;
        }
      }
      const double inv = 1.0 / J[c][c];
#pragma unroll
      for (int r = c + 1; r < 6; ++r) {
        const double m = J[r][c] * inv;
#pragma unroll
        for (int k = c; k < 6; ++k) J[r][k] -= m * J[c][k];
        s[r] -= m * s[c];
      }
    }
#pragma unroll
    for (int r = 5; r >= 0; --r) {
      double acc = s[r];
#pragma unroll
      for (int k = r + 1; k < 6; ++k) acc -= J[r][k] * s[k];
      s[r] = acc / J[r][r];
    }
    // damping: stay in the non-negative orthant (the states are mass fractions)
    double lam = 1.0;
    for (int h = 0; h < 30; ++h) {
      bool bad = false;
#pragma unroll
      for (int r = 0; r < 6; ++r) bad = bad || (w[r] + lam * s[r] < 0.0);
      if (!bad) break;
      lam *= 0.5;
    }
    double mx = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      w[r] += lam * s[r];
      mx = fmax(mx, fabs(lam * s[r]));
    }
    if (mx < 1e-15) break;
  }
  const double Fr = FA + Fb;
  const double fx = 1043.38 * w[3] * Fr + 20.92 * w[4] * Fr - 79.23 * FA - 118.34 * Fb;      // :58
  out[3 * i] = -fx;
  out[3 * i + 1] = 0.12 - w[0];                                                                // :75
  out[3 * i + 2] = 0.08 - w[5];                                                                // :89
}

}  // namespace sbo

using namespace sbo;

extern "C" int sbo_plant_wo(sbo_ctx* c, int64_t n, const double* u, double* out) {
  if (!c) return fail(SBO_E_INVALID, "ctx is NULL");
  if (n < 0 || (n > 0 && (!u || !out))) return fail(SBO_E_INVALID, "bad n / NULL array");
  if (n == 0) return SBO_OK;
  SBO_HIP(hipSetDevice(c->device));
  int rc = ensure(c->fitbuf, sizeof(double) * (size_t)n * 5);
  if (rc) return rc;
  double* du = (double*)c->fitbuf.p;
  double* dout = du + (size_t)n * 2;
  SBO_HIP(hipMemcpyAsync(du, u, sizeof(double) * (size_t)n * 2, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_plant_wo, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, (const double*)du, (long long)n, dout);
  SBO_HIP(hipGetLastError());
  SBO_HIP(hipMemcpyAsync(out, dout, sizeof(double) * (size_t)n * 3, hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipStreamSynchronize(c->stream));
  return SBO_OK;
}
