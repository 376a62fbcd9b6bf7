// fit.hip -- SURVEY.md section 8(f) rank 1: the hyper-parameter objective of the model fit, batched on the device.
//
// negative_loglikelihood (models/GP_Safe.py:169-192) for P hyper-parameter vectors at once -- what a population-based
// optimiser (the reference uses SciPy differential evolution, models/GP_Safe.py:224) evaluates every generation:
//     W = exp(2 h[:d]), sf2 = exp(2 h[d]), sn2 = exp(2 h[d+1])
//     K = sf2 exp(-1/2 D_W(X, X)) + (sn2 + 1e-8) I ;  K = (K + K^T)/2 ;  K = L L^T
//     NLL = y^T K^-1 y + log|K| = ||L^-1 y||^2 + 2 sum log L_ii          (no 1/2, no constant: :190)
// One workgroup per population member.  The factor is built as K = U^T U with U upper triangular stored row-major so
// that every inner loop walks contiguous memory; the right-hand side y rides along as an extra column, so the forward
// solve costs no extra synchronisation.  Matrices live in an HBM workspace (L2-resident at the reference's sizes).
#include <cmath>
#include "device_common.hpp"

namespace sbo {

__global__ __launch_bounds__(1024) void k_nll_batch(int n, int d, const double* __restrict__ X, const double* __restrict__ y,
                                                   const double* __restrict__ hyper, double* __restrict__ work,
                                                   double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* Xa = reinterpret_cast<double*>(smem);      // [n][d]  X * W^-1/2
  double* sq = Xa + (size_t)n * d;                   // [n]
  double* z = sq + n;                                // [n]     running right-hand side / solution
  __shared__ double sh_piv;
  __shared__ int sh_bad;
  const int p = blockIdx.x, tid = threadIdx.x;
  const double* h = hyper + (size_t)p * (d + 2);
  double* U = work + (size_t)p * n * n;
  const double sf2 = exp(2.0 * h[d]);
  const double jit = exp(2.0 * h[d + 1]) + 1e-8;                        // GP_Safe.py:184
  for (int idx = tid; idx < n * d; idx += blockDim.x) {
    const int a = idx % d;
    Xa[idx] = X[idx] * pow(exp(2.0 * h[a]), -0.5);                      // GP_Safe.py:112-115
  }
  if (tid == 0) sh_bad = 0;
  __syncthreads();
  for (int i = tid; i < n; i += blockDim.x) {
    double s = 0.0;
    for (int a = 0; a < d; ++a) s += Xa[i * d + a] * Xa[i * d + a];
    sq[i] = s;
    z[i] = y[i];
  }
  __syncthreads();
  // upper triangle of the symmetrised covariance
  for (long long idx = tid; idx < (long long)n * n; idx += blockDim.x) {
    const int i = (int)(idx / n), k = (int)(idx % n);
    if (k < i) continue;
    double dot = 0.0;
    for (int a = 0; a < d; ++a) dot += Xa[i * d + a] * Xa[k * d + a];
    const double d1 = (-2.0 * dot + sq[i]) + sq[k];                    // dist[i, k] as GP_Safe.py:119 evaluates it
    const double d2 = (-2.0 * dot + sq[k]) + sq[i];                    // dist[k, i]
    double v = (sf2 * exp(-0.5 * d1) + sf2 * exp(-0.5 * d2)) * 0.5;    // (K + K^T) / 2, GP_Safe.py:185
    if (i == k) v = sf2 * exp(-0.5 * d1) + jit;
    U[(size_t)i * n + k] = v;
  }
  __syncthreads();
  double logdet = 0.0, zz = 0.0;
  const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  for (int j = 0; j < n; ++j) {
    if (tid == 0) {
      const double piv = U[(size_t)j * n + j];
      if (!(piv > 0.0)) sh_bad = 1;
      sh_piv = piv > 0.0 ? sqrt(piv) : 1.0;
    }
    __syncthreads();
    const double ljj = sh_piv;
    double* uj = U + (size_t)j * n;
    for (int i = j + 1 + tid; i < n; i += blockDim.x) uj[i] /= ljj;     // row j of U
    if (tid == 0) {
      uj[j] = ljj;
      z[j] = z[j] / ljj;
      logdet += log(ljj);
      zz += z[j] * z[j];
    }
    __syncthreads();
    const double zj = z[j];
    for (int i = j + 1 + tid; i < n; i += blockDim.x) z[i] -= uj[i] * zj;   // forward substitution rides along
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
    }
    __syncthreads();
  }
  if (tid == 0) out[p] = sh_bad ? INFINITY : zz + 2.0 * logdet;         // GP_Safe.py:187-190
}

}  // namespace sbo

using namespace sbo;

extern "C" int sbo_nll_batch(sbo_ctx* c, int n, int d, const double* X_norm, const double* y, int P, const double* hyper,
                             double* out) {
  if (!c) return fail(SBO_E_INVALID, "ctx is NULL");
  if (n < 1 || n > SBO_MAX_N || d < 1 || d > SBO_MAX_D || P < 1) return fail(SBO_E_INVALID, "n, d or P out of range");
  if (!X_norm || !y || !hyper || !out) return fail(SBO_E_INVALID, "NULL argument");
  SBO_HIP(hipSetDevice(c->device));
  const size_t lds = sizeof(double) * ((size_t)n * d + 2 * (size_t)n);
  if (lds > 150 * 1024) return fail(SBO_E_UNSUPPORTED, "n * d too large for the fit kernel's LDS staging");
  int rc;
  const size_t in_bytes = sizeof(double) * ((size_t)n * d + n + (size_t)P * (d + 2) + P);
  if ((rc = ensure(c->fitbuf, in_bytes))) return rc;
  if ((rc = ensure(c->fitwork, sizeof(double) * (size_t)P * n * n))) return rc;
  double* dX = (double*)c->fitbuf.p;
  double* dy = dX + (size_t)n * d;
  double* dh = dy + n;
  double* dout = dh + (size_t)P * (d + 2);
  SBO_HIP(hipMemcpyAsync(dX, X_norm, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, c->stream));
  SBO_HIP(hipMemcpyAsync(dy, y, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  SBO_HIP(hipMemcpyAsync(dh, hyper, sizeof(double) * (size_t)P * (d + 2), hipMemcpyHostToDevice, c->stream));
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_nll_batch), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // 1024 threads per member for the larger matrices (more rows of the trailing update in flight); small ones keep 256
  hipLaunchKernelGGL(k_nll_batch, dim3(P), dim3(n >= 96 ? 1024 : 256), lds, c->stream, n, d, (const double*)dX, (const double*)dy,
                     (const double*)dh, (double*)c->fitwork.p, dout);
  SBO_HIP(hipGetLastError());
  SBO_HIP(hipMemcpyAsync(out, dout, sizeof(double) * P, hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipStreamSynchronize(c->stream));
  return SBO_OK;
}
