// posterior.hip -- K1: fused GP posterior for a tile of candidates (gfx950, wave64, MFMA 16x16x4).
//
// Restates, for P = 16*S candidates per workgroup and one modelled output per blockIdx.y,
//   k      = sf2 exp(-1/2 dist(X_norm, xn))            models/GP_Safe.py:98-120, 146-167
//   mean   = mp + k . alpha,  alpha = invK (Y_norm-mp)  models/GP_Safe.py:342
//   var    = max(0, sf2 - k^T invK k)                   models/GP_Safe.py:343
//   MEAN   = mean Y_std + Y_mean, VAR = var Y_std^2      models/GP_Safe.py:346-347
//   |dMEAN/dx|_inf (for the Lipschitz bound)            models/SafeOpt.py:68-71
// The cross-covariance tile K*[n, P] is generated straight into LDS in MFMA B-fragment order and never
// touches HBM.  The quadratic form is the block-triangular GEMM  T = F K*  on v_mfma_{f64,f32}_16x16x4
// (F = invK folded to its lower triangle, or L^-1), followed by a per-candidate row reduction
// sum_i k_i t_i (or sum_i t_i^2) inside the wave, then across the four waves through LDS.
#include <algorithm>
#include "device_common.hpp"

namespace sbo {

constexpr int kWaves = 4;  // waves per workgroup (256 threads)

template <typename T> __device__ __forceinline__ T exp_t(T x);
template <> __device__ __forceinline__ double exp_t<double>(double x) { return exp(x); }
template <> __device__ __forceinline__ float exp_t<float>(float x) { return expf(x); }

// Row-block schedule: wave w owns row blocks  w, 2W-1-w, 2W+w, 4W-1-w, ...  (W = kWaves) so that the
// triangular work  sum (I+1)  is balanced whenever the block count is a multiple of 2W.
__device__ __forceinline__ int row_block_of(int r, int w) {
  const int base = (r >> 1) * (2 * kWaves);
  return (r & 1) ? base + 2 * kWaves - 1 - w : base + w;
}

template <typename T, int S, int D>
__global__ __launch_bounds__(256) void k_posterior(const ModelConst mc, const CandSpec cs,
                                                   const T* __restrict__ Fpk, size_t fpk_stride,
                                                   const T* __restrict__ As, const T* __restrict__ sqA,
                                                   const T* __restrict__ alpha, const T* __restrict__ Xn,
                                                   T* __restrict__ mean_out, T* __restrict__ var_out,
                                                   unsigned long long* __restrict__ Lmax) {
  using acc_t = typename MM<T>::acc_t;
  constexpr int P = 16 * S;
  constexpr int NPARTS = kWaves / S;  // waves sharing one 16-candidate strip during generation
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nfr = mc.npad >> 2;                    // B-fragments per strip
  T* Kf = reinterpret_cast<T*>(smem);              // [S][nfr][64]
  T* qpart = Kf + (size_t)S * nfr * 64;            // [kWaves][P]
  T* mpart = qpart + kWaves * P;                   // [NPARTS][P][1 + D]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int pp = lane & 15, slot = lane >> 4;
  const int out = blockIdx.y;
  const long long tile0 = (long long)blockIdx.x * P;

  // ---------------- phase 1: cross-covariance fragments + mean / gradient dot products ----------------
  {
    const int s = wave % S, part = wave / S;
    long long g = tile0 + s * 16 + pp;
    if (g >= cs.n_local) g = cs.n_local - 1;  // clamp: computed, never stored
    double xr[D];
    cand_coords<D>(cs, g, xr);
    T B[D];
    T sqB = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      const T xn = ((T)xr[a] - (T)mc.X_mean[a]) / (T)mc.X_std[a];   // GP_Safe.py:326 (true division)
      B[a] = xn * (T)mc.vinv[out][a];                                // GP_Safe.py:116
      sqB += B[a] * B[a];
    }
    const T sf2 = (T)mc.sf2[out];
    const T* As_o = As + (size_t)out * mc.npad * D;
    const T* sqA_o = sqA + (size_t)out * mc.npad;
    const T* al_o = alpha + (size_t)out * mc.npad;
    T m0 = 0;
    T ms[D];
#pragma unroll
    for (int a = 0; a < D; ++a) ms[a] = 0;
    const int jj0 = part * nfr / NPARTS, jj1 = (part + 1) * nfr / NPARTS;
    T* kf_s = Kf + (size_t)s * nfr * 64 + lane;
    for (int jj = jj0; jj < jj1; ++jj) {
      const int j = ((jj >> 2) << 4) + MM<T>::jslot(jj & 3, slot);
      T dot = 0;
#pragma unroll
      for (int a = 0; a < D; ++a) dot = fma(As_o[j * D + a], B[a], dot);
      const T dist = (T(-2) * dot + sqA_o[j]) + sqB;                 // GP_Safe.py:119 (expanded form)
      const T k = sf2 * exp_t<T>(T(-0.5) * dist);                    // GP_Safe.py:166
      const T w = al_o[j] * k;
      m0 += w;
#pragma unroll
      for (int a = 0; a < D; ++a) ms[a] = fma(w, Xn[j * D + a], ms[a]);
      kf_s[(size_t)jj * 64] = k;
    }
    // reduce over the four k-slots (lanes l, l^16, l^32, l^48 hold the same candidate)
    m0 += __shfl_xor(m0, 16);
    m0 += __shfl_xor(m0, 32);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      ms[a] += __shfl_xor(ms[a], 16);
      ms[a] += __shfl_xor(ms[a], 32);
    }
    if (slot == 0) {
      T* mp_ = mpart + ((size_t)part * P + s * 16 + pp) * (1 + D);
      mp_[0] = m0;
#pragma unroll
      for (int a = 0; a < D; ++a) mp_[1 + a] = ms[a];
    }
  }
  __syncthreads();

  // ---------------- phase 2: block-triangular contraction on the matrix cores ----------------
  {
    const int nb = mc.npad >> 4;
    const T* F_o = Fpk + (size_t)out * fpk_stride + lane;
    T quad[S];
#pragma unroll
    for (int s = 0; s < S; ++s) quad[s] = 0;
    for (int r = 0;; ++r) {
      const int I = row_block_of(r, wave);
      if (I >= nb) break;
      acc_t acc[S];
#pragma unroll
      for (int s = 0; s < S; ++s) acc[s] = acc_t{0, 0, 0, 0};
      const T* fp = F_o + (size_t)(I * (I + 1) / 2) * 4 * 64;   // fragments of blocks (I, 0..I), 4 k-steps each
      for (int J = 0; J <= I; ++J) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int st = J * 4 + kk;
          const T a = fp[(size_t)st * 64];
#pragma unroll
          for (int s = 0; s < S; ++s) {
            const T b = Kf[((size_t)s * nfr + st) * 64 + lane];
            acc[s] = MM<T>::mfma(a, b, acc[s]);
          }
        }
      }
      // row reduction: register r_ of acc sits on the lane holding k of the same observation (B-fragment I*4+r_)
#pragma unroll
      for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int r_ = 0; r_ < 4; ++r_) {
          const T t = acc[s][r_];
          const T kv = (mc.factor == SBO_FACTOR_INVK) ? Kf[((size_t)s * nfr + I * 4 + r_) * 64 + lane] : t;
          quad[s] = fma(kv, t, quad[s]);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
      T v = quad[s];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (slot == 0) qpart[wave * P + s * 16 + pp] = v;
    }
  }
  __syncthreads();

  // ---------------- epilogue: one thread per candidate ----------------
  if (tid < 64) {
    T gn = 0;
    const long long g = tile0 + tid;
    const bool valid = (tid < P) && (g < cs.n_local);
    if (valid) {
      T quad = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) quad += qpart[w * P + tid];
      T m0 = 0;
      T ms[D];
#pragma unroll
      for (int a = 0; a < D; ++a) ms[a] = 0;
#pragma unroll
      for (int p_ = 0; p_ < NPARTS; ++p_) {
        const T* mp_ = mpart + ((size_t)p_ * P + tid) * (1 + D);
        m0 += mp_[0];
#pragma unroll
        for (int a = 0; a < D; ++a) ms[a] += mp_[1 + a];
      }
      const T sf2 = (T)mc.sf2[out], ystd = (T)mc.Y_std[out];
      const T mean = (T)mc.mp[out] + m0;                        // GP_Safe.py:342
      T var = sf2 - quad;                                       // GP_Safe.py:343
      var = var > T(0) ? var : T(0);
      mean_out[(size_t)out * cs.n_local + g] = add_rn(mul_rn(mean, ystd), (T)mc.Y_mean[out]);   // :346
      var_out[(size_t)out * cs.n_local + g] = mul_rn(var, mul_rn(ystd, ystd));                  // :347
      // gradient of the un-normalised mean w.r.t. raw x (analytic form of jax.grad(self.mean), SafeOpt.py:68-71)
      double xr[D];
      cand_coords<D>(cs, g, xr);
#pragma unroll
      for (int a = 0; a < D; ++a) {
        if (a < mc.d) {
          const T xn = ((T)xr[a] - (T)mc.X_mean[a]) / (T)mc.X_std[a];
          T ga = ystd * (ms[a] - xn * m0) * (T)mc.inv_ell[out][a] / (T)mc.X_std[a];
          ga = ga < 0 ? -ga : ga;
          gn = ga > gn ? ga : gn;
        }
      }
    }
    // wave max -> one atomic per workgroup (values are >= 0, so the raw bit pattern orders correctly)
    double gd = (double)gn;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double other = __shfl_xor(gd, o);
      gd = other > gd ? other : gd;
    }
    if (tid == 0) atomicMax(&Lmax[out], (unsigned long long)__double_as_longlong(gd));
  }
}

// ---- elementwise helpers on the SoA workspace ------------------------------------------------------
// BO.mean / ucb / lcb batched (models/SafeOpt.py:29-45): value = mean +/- b sqrt(var), unfused.
template <typename T>
__global__ void k_bound(const T* __restrict__ mean, const T* __restrict__ var, long long n, T b, int kind,
                        T* __restrict__ out) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    const T m = mean[g], v = var[g];
    T r;
    if (kind == SBO_MEAN) r = m;
    else if (kind == SBO_VAR) r = v;
    else {
      const T sd = mul_rn(b, sqrt_rn(v));
      r = (kind == SBO_UCB) ? add_rn(m, sd) : sub_rn(m, sd);
    }
    out[g] = r;
  }
}

template <typename T>
__global__ void k_soa_to_aos(const T* __restrict__ soa, long long n, int q, T* __restrict__ aos) {
  const long long total = n * q;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long g = t / q;
    const int i = (int)(t - g * q);
    aos[t] = soa[(size_t)i * n + g];
  }
}

// ---- launchers ---------------------------------------------------------------------------------------
template <typename T, int S, int D>
static int launch_posterior_t(sbo_ctx* c) {
  const ModelConst& mc = c->mc;
  constexpr int P = 16 * S;
  const int nfr = mc.npad / 4;
  const size_t lds = sizeof(T) * ((size_t)S * nfr * 64 + kWaves * P + (kWaves / S) * P * (1 + D));
  auto kern = k_posterior<T, S, D>;
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long tiles = (c->cs.n_local + P - 1) / P;
  if (tiles > 0x7fffffffLL) return fail(SBO_E_UNSUPPORTED, "too many candidate tiles for one launch");
  dim3 grid((unsigned)tiles, (unsigned)mc.q), block(256);
  hipLaunchKernelGGL(kern, grid, block, lds, c->stream, mc, c->cs, (const T*)c->Fpk.p, c->fpk_stride,
                     (const T*)c->As.p, (const T*)c->sqA.p, (const T*)c->alpha.p, (const T*)c->Xn.p,
                     (T*)c->mean.p, (T*)c->var.p, (unsigned long long*)c->Lmax.p);
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

template <typename T, int S>
static int launch_posterior_d(sbo_ctx* c) {
  switch (c->mc.dpad) {
    case 2: return launch_posterior_t<T, S, 2>(c);
    case 4: return launch_posterior_t<T, S, 4>(c);
    case 8: return launch_posterior_t<T, S, 8>(c);
  }
  return fail(SBO_E_UNSUPPORTED, "unsupported padded dimension");
}

template <typename T>
static int launch_posterior_s(sbo_ctx* c) {
  // strips per workgroup: the K* tile [npad, 16 S] must fit the 160 KiB LDS with room for two workgroups
  // per CU when it can (latency hiding across the generation / contraction phases).
  const size_t per_strip = sizeof(T) * (size_t)c->mc.npad * 16;
  const size_t budget = 144 * 1024;
  if (per_strip * 4 <= budget) return launch_posterior_d<T, 4>(c);
  if (per_strip * 2 <= budget) return launch_posterior_d<T, 2>(c);
  if (per_strip <= budget) return launch_posterior_d<T, 1>(c);
  return fail(SBO_E_UNSUPPORTED, "n too large for the LDS-resident cross-covariance tile");
}

int launch_posterior(sbo_ctx* c) {
  SBO_HIP(hipMemsetAsync(c->Lmax.p, 0, sizeof(unsigned long long) * kMaxQ, c->stream));
  return c->dtype == SBO_F64 ? launch_posterior_s<double>(c) : launch_posterior_s<float>(c);
}

int launch_bound(sbo_ctx* c, double b, int index, int kind, void* dev_out) {
  const long long n = c->cs.n_local;
  const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
  if (c->dtype == SBO_F64) {
    const double* m = (const double*)c->mean.p + (size_t)index * n;
    const double* v = (const double*)c->var.p + (size_t)index * n;
    hipLaunchKernelGGL(k_bound<double>, dim3(blocks), dim3(256), 0, c->stream, m, v, n, b, kind, (double*)dev_out);
  } else {
    const float* m = (const float*)c->mean.p + (size_t)index * n;
    const float* v = (const float*)c->var.p + (size_t)index * n;
    hipLaunchKernelGGL(k_bound<float>, dim3(blocks), dim3(256), 0, c->stream, m, v, n, (float)b, kind, (float*)dev_out);
  }
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

int launch_soa_to_aos(sbo_ctx* c, const void* soa, void* aos) {
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  const int blocks = (int)std::min<long long>((n * q + 255) / 256, 4096);
  if (c->dtype == SBO_F64)
    hipLaunchKernelGGL(k_soa_to_aos<double>, dim3(blocks), dim3(256), 0, c->stream, (const double*)soa, n, q, (double*)aos);
  else
    hipLaunchKernelGGL(k_soa_to_aos<float>, dim3(blocks), dim3(256), 0, c->stream, (const float*)soa, n, q, (float*)aos);
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

}  // namespace sbo
