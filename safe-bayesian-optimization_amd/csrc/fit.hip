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
#include <vector>
#include "device_common.hpp"

namespace sbo {

// NLL of one hyper-parameter vector h[d + 2], evaluated by the whole workgroup (result valid in thread 0).
// U: this member's n x n workspace; smem: [n d + 2 n] doubles.
__device__ double nll_member(int n, int d, const double* __restrict__ X, const double* __restrict__ y, const double* h,
                             double* __restrict__ U, double* smem_d) {
  double* Xa = smem_d;                               // [n][d]  X * W^-1/2
  double* sq = Xa + (size_t)n * d;                   // [n]
  double* z = sq + n;                                // [n]     running right-hand side / solution
  __shared__ double sh_piv;
  __shared__ int sh_bad;
  const int tid = threadIdx.x;
  const double sf2 = exp(2.0 * h[d]);
  const double jit = exp(2.0 * h[d + 1]) + 1e-8;                        // GP_Safe.py:184
  __syncthreads();                                                      // (the buffers may still be read by a previous call)
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
  return sh_bad ? INFINITY : zz + 2.0 * logdet;                         // GP_Safe.py:187-190 (thread 0's value)
}

__global__ __launch_bounds__(1024) void k_nll_batch(int n, int d, const double* __restrict__ X, const double* __restrict__ y,
                                                    const double* __restrict__ hyper, double* __restrict__ work,
                                                    double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int p = blockIdx.x;
  const double v = nll_member(n, d, X, y, hyper + (size_t)p * (d + 2), work + (size_t)p * n * n, reinterpret_cast<double*>(smem));
  if (threadIdx.x == 0) out[p] = v;
}

// ---- differential evolution on the device --------------------------------------------------------------------------
// One generation of SciPy's default strategy as the reference calls it (models/GP_Safe.py:224: best1bin, dithered
// mutation, recombination 0.7) with deferred updating: workgroup i builds the trial vector of member i from the current
// population, evaluates its NLL and keeps the better of (parent, trial) in the next population.
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {     // splitmix64 finaliser
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ double u01(unsigned long long seed, unsigned gen, unsigned member, unsigned draw) {
  const unsigned long long k = mix64(seed ^ mix64(((unsigned long long)gen << 32) | member) ^ mix64(0xda3e39cb94b95bdbull + draw));
  return (double)(k >> 11) * (1.0 / 9007199254740992.0);
}

__global__ __launch_bounds__(1024) void k_de_step(int n, int d, const double* __restrict__ X, const double* __restrict__ y, int P,
                                                  const double* __restrict__ pop, const double* __restrict__ energy,
                                                  double* __restrict__ pop_next, double* __restrict__ energy_next,
                                                  const double* __restrict__ lo, const double* __restrict__ hi, double F, double CR,
                                                  unsigned long long seed, unsigned gen, double* __restrict__ work) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ double trial[SBO_MAX_D + 2];
  const int i = blockIdx.x, D = d + 2;
  if (threadIdx.x == 0) {
    int best = 0;                                    // best of the current generation (lowest index on ties)
    for (int p = 1; p < P; ++p)
      if (energy[p] < energy[best]) best = p;
    // two distinct members other than i (scipy _select_samples: a random permutation without the candidate)
    int r0 = (int)(u01(seed, gen, i, 0) * (P - 1));
    int r1 = (int)(u01(seed, gen, i, 1) * (P - 2));
    if (r0 >= i) ++r0;
    int lo_ = r0 < i ? r0 : i, hi_ = r0 < i ? i : r0;
    if (r1 >= lo_) ++r1;
    if (r1 >= hi_) ++r1;
    const int fill = (int)(u01(seed, gen, i, 2) * D);
    for (int a = 0; a < D; ++a) {
      double v = pop[(size_t)i * D + a];
      if (a == fill || u01(seed, gen, i, 8 + a) < CR) {
        v = pop[(size_t)best * D + a] + F * (pop[(size_t)r0 * D + a] - pop[(size_t)r1 * D + a]);
        if (v < lo[a] || v > hi[a]) v = lo[a] + u01(seed, gen, i, 64 + a) * (hi[a] - lo[a]);   // scipy re-draws out-of-bounds entries
      }
      trial[a] = v;
    }
  }
  __syncthreads();
  const double e = nll_member(n, d, X, y, trial, work + (size_t)i * n * n, reinterpret_cast<double*>(smem));
  if (threadIdx.x == 0) {
    const bool take = e <= energy[i];                // scipy: "if energy <= self.population_energies[candidate]"
    for (int a = 0; a < D; ++a) pop_next[(size_t)i * D + a] = take ? trial[a] : pop[(size_t)i * D + a];
    energy_next[i] = take ? e : energy[i];
  }
}

}  // namespace sbo

using namespace sbo;

static unsigned long long mix64_host(unsigned long long x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}

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

// Differential evolution of the NLL over the box [lo, hi]^(d+2) entirely on the device:
// init_pop[P, d + 2] is the caller's initial population (SciPy draws a Latin hypercube), `seed` keys the counter-based
// generator of the trial vectors, F is dithered per generation in [0.5, 1) as SciPy's default mutation=(0.5, 1) does.
// Stops after maxiter generations or when std(energies) <= atol + tol |mean(energies)| (SciPy's criterion, checked
// every 8 generations).  Returns the best member, its NLL and the generations run.
extern "C" int sbo_fit_de(sbo_ctx* c, int n, int d, const double* X_norm, const double* y, int P, const double* lo, const double* hi,
                          const double* init_pop, uint64_t seed, int maxiter, double tol, double atol, double* best_x,
                          double* best_energy, int* generations) {
  if (!c) return fail(SBO_E_INVALID, "ctx is NULL");
  if (n < 1 || n > SBO_MAX_N || d < 1 || d > SBO_MAX_D || P < 4 || maxiter < 0) return fail(SBO_E_INVALID, "n, d, P or maxiter out of range");
  if (!X_norm || !y || !lo || !hi || !init_pop || !best_x || !best_energy) return fail(SBO_E_INVALID, "NULL argument");
  SBO_HIP(hipSetDevice(c->device));
  const int D = d + 2;
  const size_t lds = sizeof(double) * ((size_t)n * d + 2 * (size_t)n);
  if (lds > 150 * 1024) return fail(SBO_E_UNSUPPORTED, "n * d too large for the fit kernel's LDS staging");
  int rc;
  const size_t in_elems = (size_t)n * d + n + 2 * (size_t)P * D + 2 * (size_t)P + 2 * (size_t)D;
  if ((rc = ensure(c->fitbuf, sizeof(double) * in_elems))) return rc;
  if ((rc = ensure(c->fitwork, sizeof(double) * (size_t)P * n * n))) return rc;
  double* dX = (double*)c->fitbuf.p;
  double* dy = dX + (size_t)n * d;
  double* dpop[2] = {dy + n, dy + n + (size_t)P * D};
  double* den[2] = {dpop[1] + (size_t)P * D, dpop[1] + (size_t)P * D + P};
  double* dlo = den[1] + P;
  double* dhi = dlo + D;
  SBO_HIP(hipMemcpyAsync(dX, X_norm, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, c->stream));
  SBO_HIP(hipMemcpyAsync(dy, y, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
  SBO_HIP(hipMemcpyAsync(dpop[0], init_pop, sizeof(double) * (size_t)P * D, hipMemcpyHostToDevice, c->stream));
  SBO_HIP(hipMemcpyAsync(dlo, lo, sizeof(double) * D, hipMemcpyHostToDevice, c->stream));
  SBO_HIP(hipMemcpyAsync(dhi, hi, sizeof(double) * D, hipMemcpyHostToDevice, c->stream));
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_nll_batch), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_de_step), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int threads = n >= 96 ? 1024 : 256;
  hipLaunchKernelGGL(k_nll_batch, dim3(P), dim3(threads), lds, c->stream, n, d, (const double*)dX, (const double*)dy,
                     (const double*)dpop[0], (double*)c->fitwork.p, den[0]);
  std::vector<double> he(P);
  int cur = 0, gen = 0;
  unsigned long long fstate = mix64_host(seed);
  for (; gen < maxiter; ++gen) {
    fstate = mix64_host(fstate);
    const double F = 0.5 + 0.5 * ((double)(fstate >> 11) * (1.0 / 9007199254740992.0));
    hipLaunchKernelGGL(k_de_step, dim3(P), dim3(threads), lds, c->stream, n, d, (const double*)dX, (const double*)dy, P,
                       (const double*)dpop[cur], (const double*)den[cur], dpop[cur ^ 1], den[cur ^ 1], (const double*)dlo,
                       (const double*)dhi, F, 0.7, (unsigned long long)seed, (unsigned)gen, (double*)c->fitwork.p);
    cur ^= 1;
    if ((gen & 7) == 7 || gen + 1 == maxiter) {
      SBO_HIP(hipMemcpyAsync(he.data(), den[cur], sizeof(double) * P, hipMemcpyDeviceToHost, c->stream));
      SBO_HIP(hipStreamSynchronize(c->stream));
      double mean = 0, var = 0;
      bool finite = true;
      for (double e : he) { mean += e; finite = finite && std::isfinite(e); }
      mean /= P;
      for (double e : he) var += (e - mean) * (e - mean);
      if (finite && std::sqrt(var / P) <= atol + tol * std::fabs(mean)) { ++gen; break; }
    }
  }
  SBO_HIP(hipGetLastError());
  std::vector<double> hp((size_t)P * D);
  SBO_HIP(hipMemcpyAsync(he.data(), den[cur], sizeof(double) * P, hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipMemcpyAsync(hp.data(), dpop[cur], sizeof(double) * (size_t)P * D, hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipStreamSynchronize(c->stream));
  int best = 0;
  for (int p = 1; p < P; ++p)
    if (he[p] < he[best]) best = p;
  for (int a = 0; a < D; ++a) best_x[a] = hp[(size_t)best * D + a];
  *best_energy = he[best];
  if (generations) *generations = gen;
  return SBO_OK;
}
