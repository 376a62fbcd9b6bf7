// posterior.hip -- K1: fused GP posterior for a tile of candidates (gfx950, wave64, MFMA 16x16x4).
//
// Restates, for P = 16*S candidates per workgroup and one modelled output per blockIdx.y,
//   k      = sf2 exp(-1/2 dist(X_norm, xn))            models/GP_Safe.py:98-120, 146-167
//   mean   = mp + k . alpha,  alpha = invK (Y_norm-mp)  models/GP_Safe.py:342
//   var    = max(0, sf2 - k^T invK k)                   models/GP_Safe.py:343
//   MEAN   = mean Y_std + Y_mean, VAR = var Y_std^2      models/GP_Safe.py:346-347
//   |dMEAN/dx|_inf (for the Lipschitz bound)            models/SafeOpt.py:68-71
// The cross-covariance tile K*[n, P] is generated straight into LDS in MFMA B-fragment order and never
// touches HBM.  The quadratic form k^T invK k = ||M k||^2 (M lower triangular with M^T M = invK, i.e. L^-1)
// is the block-triangular GEMM  T = M K*  on the matrix cores (device_common.hpp, MM<T>), followed by a
// per-candidate reduction  sum_i t_i^2  inside the wave, then across the four waves through LDS.
#include <algorithm>
#include <cstring>
#include <type_traits>
#include "device_common.hpp"

namespace sbo {

constexpr int kWaves = 4;  // waves per workgroup (256 threads)

template <typename T> __device__ __forceinline__ T exp_t(T x);
template <> __device__ __forceinline__ double exp_t<double>(double x) { return exp(x); }
// fp32 path: hardware exp2 (v_exp_f32, ~1e-6 relative) -- the fp32 parity bar is 1e-4 and K* is regenerated per chunk pair
template <> __device__ __forceinline__ float exp_t<float>(float x) { return __expf(x); }

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
    const T* F_o = Fpk + (size_t)out * fpk_stride;
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
          const typename MM<T>::a_t a = MM<T>::load_a(fp + (size_t)J * 256, lane, kk);
#pragma unroll
          for (int s = 0; s < S; ++s) {
            const T b = Kf[((size_t)s * nfr + st) * 64 + lane];
            acc[s] = MM<T>::mfma(a, b, acc[s]);
          }
        }
      }
      // row reduction: every accumulator element of this lane is a t_i of candidate lane&15
#pragma unroll
      for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int r_ = 0; r_ < 4; ++r_) quad[s] = fma(acc[s][r_], acc[s][r_], quad[s]);
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

// =====================================================================================================
// K1c: generic candidates (explicit lists, ragged shards), any n.  The K* tile of all n observations does not fit
// LDS once n is large (n = 2048 x 64 candidates x 4 B = 512 KiB), so the triangle is walked in chunks of CB
// row/column blocks: for row chunk Ic every wave keeps RW row blocks x 4 strips of accumulators in registers and
// the column chunks Jc <= Ic are generated into LDS one after the other (K* of a column chunk is therefore
// re-generated for every row chunk at or above it -- about (nc+1)/2 times on average -- which costs VALU exp() work
// but keeps 64 candidates per workgroup, i.e. four MFMAs per A-fragment load instead of one).
// =====================================================================================================
template <typename T> struct ChunkCfg;
template <> struct ChunkCfg<float> { static constexpr int CB = 16, RW = 8; };    // 64 KiB K* chunk, 128 acc VGPRs
template <> struct ChunkCfg<double> { static constexpr int CB = 8, RW = 2; };    // 64 KiB K* chunk, 64 acc VGPRs

template <typename T, int D>
__global__ __launch_bounds__(256) void k_posterior_chunked(const ModelConst mc, const CandSpec cs,
                                                           const T* __restrict__ Fpk, size_t fpk_stride,
                                                           const T* __restrict__ As, const T* __restrict__ sqA,
                                                           const T* __restrict__ alpha, const T* __restrict__ Xn,
                                                           T* __restrict__ mean_out, T* __restrict__ var_out,
                                                           unsigned long long* __restrict__ Lmax) {
  using acc_t = typename MM<T>::acc_t;
  using a_t = typename MM<T>::a_t;
  constexpr int S = kWaves, P = 16 * S;
  constexpr int CB = ChunkCfg<T>::CB, RW = ChunkCfg<T>::RW, RC = RW * kWaves;
  static_assert(RC % CB == 0, "a row chunk spans whole column chunks");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  T* Kf = reinterpret_cast<T*>(smem);              // [S][CB * 4][64]  K* fragments of the current column chunk
  T* qpart = Kf + (size_t)S * CB * 4 * 64;         // [kWaves][P]
  T* mpart = qpart + kWaves * P;                   // [P][1 + D]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pp = lane & 15, slot = lane >> 4;
  const int out = blockIdx.y;
  const long long tile0 = (long long)blockIdx.x * P;
  const int nb = mc.npad >> 4;
  const int ncr = (nb + RC - 1) / RC;              // row chunks
  const int ncc = (nb + CB - 1) / CB;              // column chunks

  // this lane's candidate for the generation phases: strip = wave, candidate pp, k-slot `slot`
  long long gcand = tile0 + wave * 16 + pp;
  if (gcand >= cs.n_local) gcand = cs.n_local - 1;  // clamp: computed, never stored
  T B[D];
  T sqB = 0;
  {
    double xr[D];
    cand_coords<D>(cs, gcand, xr);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      const T xn = ((T)xr[a] - (T)mc.X_mean[a]) / (T)mc.X_std[a];   // GP_Safe.py:326
      B[a] = xn * (T)mc.vinv[out][a];                                // GP_Safe.py:116
      sqB += B[a] * B[a];
    }
  }
  const T sf2 = (T)mc.sf2[out];
  const T* As_o = As + (size_t)out * mc.npad * D;
  const T* sqA_o = sqA + (size_t)out * mc.npad;
  const T* al_o = alpha + (size_t)out * mc.npad;
  const T* F_o = Fpk + (size_t)out * fpk_stride;
  // this lane's B operand of the cross-term products (fp32 generation below): coordinate 4 ks + slot of its candidate
  T bsel[(D + 3) / 4];
#pragma unroll
  for (int ks = 0; ks < (D + 3) / 4; ++ks) {
    T v = 0;
#pragma unroll
    for (int a = 0; a < D; ++a) v = (a == 4 * ks + slot) ? B[a] : v;
    bsel[ks] = v;
  }
  T m0 = 0, ms[D];
#pragma unroll
  for (int a = 0; a < D; ++a) ms[a] = 0;
  T quad[S];
#pragma unroll
  for (int s = 0; s < S; ++s) quad[s] = 0;

  for (int Ic = 0; Ic < ncr; ++Ic) {
    acc_t acc[RW][S];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int s = 0; s < S; ++s) acc[r][s] = acc_t{0, 0, 0, 0};
    const bool with_dots = (Ic == ncr - 1);         // the last row chunk sees every column chunk exactly once
    const int last_row = ((Ic + 1) * RC < nb ? (Ic + 1) * RC : nb) - 1;
    const int Jc_end = last_row / CB < ncc ? last_row / CB : ncc - 1;   // last column chunk a row of this chunk reaches
    for (int Jc = 0; Jc <= Jc_end; ++Jc) {
      // ---- generate K* fragments of column chunk Jc (strip = wave) ----
      __syncthreads();                              // previous chunk's readers are done
      const int Jlo = Jc * CB, Jhi = (Jlo + CB < nb) ? Jlo + CB : nb;
      T* kf_w = Kf + (size_t)wave * CB * 4 * 64 + lane;
      if constexpr (std::is_same<T, float>::value) {
        // (r04) the cross terms A_j . B of a whole 16 x 16 block (observations x this wave's candidates) on the matrix cores: K = D
        // coordinates = one or two 16x16x4 steps.  Element i of the result at lane (slot, pp) is row 4 slot + i = jslot(i, slot) --
        // exactly what k-step i of the block's K* fragment wants from this lane, so the values go to LDS where they are.  Leaves
        // ~6 vector instructions per element (distance, exponential, store) of ~30: on config E the exp() generation was 22 % of
        // the kernel's issue slots beside 66 % for the contraction.
        constexpr int KD = (D + 3) / 4;
        for (int Jb = Jlo; Jb < Jhi; ++Jb) {
          f4_t dotv = f4_t{0, 0, 0, 0};
#pragma unroll
          for (int ks = 0; ks < KD; ++ks) {
            const int dim = 4 * ks + slot;
            const float av = dim < D ? (float)As_o[(Jb * 16 + pp) * D + dim] : 0.f;
            dotv = MM<float>::mfma(av, (float)bsel[ks], dotv);
          }
          const int j0 = Jb * 16 + 4 * slot;
          const f4_t sq = *reinterpret_cast<const f4_t*>(sqA_o + j0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float dist = (-2.f * dotv[i] + sq[i]) + (float)sqB;         // GP_Safe.py:119 (expanded form)
            const float k = (float)sf2 * exp_t<float>(-0.5f * dist);          // GP_Safe.py:166
            kf_w[(size_t)((Jb - Jlo) * 4 + i) * 64] = (T)k;
            if (with_dots) {
              const int j = j0 + i;
              const T w = al_o[j] * (T)k;
              m0 += w;
#pragma unroll
              for (int a = 0; a < D; ++a) ms[a] = fma(w, Xn[j * D + a], ms[a]);
            }
          }
        }
      } else {
        for (int jl = 0; jl < (Jhi - Jlo) * 4; ++jl) {
          const int jj = Jlo * 4 + jl;
          const int j = ((jj >> 2) << 4) + MM<T>::jslot(jj & 3, slot);
          T dot = 0;
#pragma unroll
          for (int a = 0; a < D; ++a) dot = fma(As_o[j * D + a], B[a], dot);
          const T dist = (T(-2) * dot + sqA_o[j]) + sqB;               // GP_Safe.py:119 (expanded form)
          const T k = sf2 * exp_t<T>(T(-0.5) * dist);                  // GP_Safe.py:166
          kf_w[(size_t)jl * 64] = k;
          if (with_dots) {
            const T w = al_o[j] * k;
            m0 += w;
#pragma unroll
            for (int a = 0; a < D; ++a) ms[a] = fma(w, Xn[j * D + a], ms[a]);
          }
        }
      }
      __syncthreads();
      // ---- contraction: my RW row blocks of chunk Ic against the column blocks of chunk Jc ----
      // A-fragments of column block J + 1 are fetched while block J is multiplied (reads past a row's last block
      // stay inside the padded buffer)
      a_t a_nx[RW][4];
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        const int I = Ic * RC + wave + kWaves * r;
        const int Ic_ = I < nb ? I : nb - 1;
        MM<T>::load_a4(F_o + ((size_t)Ic_ * (Ic_ + 1) / 2 + Jlo) * 256, lane, a_nx[r]);
      }
      for (int J = Jlo; J < Jhi; ++J) {
        a_t a_cur[RW][4];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
          const int I = Ic * RC + wave + kWaves * r;
          const int Ic_ = I < nb ? I : nb - 1;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) a_cur[r][kk] = a_nx[r][kk];
          MM<T>::load_a4(F_o + ((size_t)Ic_ * (Ic_ + 1) / 2 + J + 1) * 256, lane, a_nx[r]);
        }
        auto kstep = [&](auto kk_tag) {
          constexpr int kk = decltype(kk_tag)::value;
          T bfr[S];
#pragma unroll
          for (int s = 0; s < S; ++s) bfr[s] = Kf[((size_t)s * CB * 4 + (J - Jlo) * 4 + kk) * 64 + lane];
#pragma unroll
          for (int r = 0; r < RW; ++r) {
            const int I = Ic * RC + wave + kWaves * r;   // rows interleaved over the waves (diagonal chunk balance)
            if constexpr (std::is_same<T, double>::value) {
              if (I < nb && J < I) {
#pragma unroll
                for (int s = 0; s < S; ++s) acc[r][s] = MM<T>::mfma(a_cur[r][kk], bfr[s], acc[r][s]);
              } else if (J == I) {                       // diagonal block: zero 4x4 sub-blocks skipped
#pragma unroll
                for (int s = 0; s < S; ++s) acc[r][s] = MM<T>::template mfma_diag<kk>(a_cur[r][kk], bfr[s], acc[r][s]);
              }
            } else {
              if (I < nb && J <= I) {
#pragma unroll
                for (int s = 0; s < S; ++s) acc[r][s] = MM<T>::mfma(a_cur[r][kk], bfr[s], acc[r][s]);
              }
            }
          }
        };
        kstep(std::integral_constant<int, 0>{});
        kstep(std::integral_constant<int, 1>{});
        kstep(std::integral_constant<int, 2>{});
        kstep(std::integral_constant<int, 3>{});
      }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
      for (int s = 0; s < S; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) quad[s] = fma(acc[r][s][e], acc[r][s][e], quad[s]);
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
    T* mp_ = mpart + (size_t)(wave * 16 + pp) * (1 + D);
    mp_[0] = m0;
#pragma unroll
    for (int a = 0; a < D; ++a) mp_[1 + a] = ms[a];
  }
#pragma unroll
  for (int s = 0; s < S; ++s) {
    T v = quad[s];
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    if (slot == 0) qpart[wave * P + s * 16 + pp] = v;
  }
  __syncthreads();

  if (tid < 64) {
    T gn = 0;
    const long long g = tile0 + tid;
    if (g < cs.n_local) {
      T quad_ = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) quad_ += qpart[w * P + tid];
      const T* mp_ = mpart + (size_t)tid * (1 + D);
      const T m0_ = mp_[0];
      const T ystd = (T)mc.Y_std[out];
      const T mean = (T)mc.mp[out] + m0_;                        // GP_Safe.py:342
      T var = sf2 - quad_;                                       // GP_Safe.py:343
      var = var > T(0) ? var : T(0);
      mean_out[(size_t)out * cs.n_local + g] = add_rn(mul_rn(mean, ystd), (T)mc.Y_mean[out]);   // :346
      var_out[(size_t)out * cs.n_local + g] = mul_rn(var, mul_rn(ystd, ystd));                  // :347
      double xr[D];
      cand_coords<D>(cs, g, xr);
#pragma unroll
      for (int a = 0; a < D; ++a) {
        if (a < mc.d) {
          const T xn = ((T)xr[a] - (T)mc.X_mean[a]) / (T)mc.X_std[a];
          T ga = ystd * (mp_[1 + a] - xn * m0_) * (T)mc.inv_ell[out][a] / (T)mc.X_std[a];
          ga = ga < 0 ? -ga : ga;
          gn = ga > gn ? ga : gn;
        }
      }
    }
    double gd = (double)gn;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double other = __shfl_xor(gd, o);
      gd = other > gd ? other : gd;
    }
    if (tid == 0) atomicMax(&Lmax[out], (unsigned long long)__double_as_longlong(gd));
  }
}

// =====================================================================================================
// K1g: the same posterior for candidates on a tensor grid, using the separability of the RBF-ARD kernel:
//   k(x, X_j) = sf2 prod_a exp(-1/2 ((xn_a - X_ja)/l_a)^2) = E0[g0][j] * Erest[(g1, g2, ..)][j].
// Per-axis tables (count_a x n entries, built by k_build_tables at every launch) replace the N*n exp()
// evaluations; a B-fragment is one multiply of two table entries, formed in registers right before the MFMA
// that consumes it, so the cross-covariance tile never exists in LDS either.  Workgroup tile = 16 consecutive
// axis-0 positions x S lines of the remaining axes.
// (Differs from the expanded-distance form of models/GP_Safe.py:119 by a few ulp in k; parity bar is 1e-10.)
// =====================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void k_build_tables(const ModelConst mc, const CandSpec cs, const T* __restrict__ As,
                                                      int D, T* __restrict__ E0f, size_t e0_stride, T* __restrict__ Er,
                                                      size_t er_stride_o, size_t er_off1, size_t er_off2, size_t er_off3,
                                                      size_t er_off4, size_t er_off5, size_t er_off6, size_t er_off7,
                                                      const double* __restrict__ axc /* nullptr, or explicit axis positions (K1t's
                                                      Chebyshev nodes): axis a at offset sum_{a' < a} count[a'] */) {
  const int o = blockIdx.y;
  const int nfr = mc.npad >> 2;
  const long long cnt0 = cs.count[0];
  const long long ntile0 = (cnt0 + 15) / 16;
  const long long n0 = ntile0 * nfr * 64;
  long long total = n0;
  for (int a = 1; a < cs.d; ++a) total += cs.count[a] * mc.npad;
  const size_t er_off[8] = {0, er_off1, er_off2, er_off3, er_off4, er_off5, er_off6, er_off7};
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int a, j;
    long long i;
    T* dst;
    if (t < n0) {
      const int lane = (int)(t & 63);
      const long long fr = t >> 6;
      const int jj = (int)(fr % nfr);
      const long long t0 = fr / nfr;
      a = 0;
      i = t0 * 16 + (lane & 15);
      if (i > cnt0 - 1) i = cnt0 - 1;
      j = ((jj >> 2) << 4) + MM<T>::jslot(jj & 3, lane >> 4);
      dst = E0f + (size_t)o * e0_stride + t;
    } else {
      long long u = t - n0;
      a = 1;
      while (u >= cs.count[a] * mc.npad) { u -= cs.count[a] * mc.npad; ++a; }
      i = u / mc.npad;
      j = (int)(u % mc.npad);
      dst = Er + (size_t)o * er_stride_o + er_off[a] + u;
    }
    const long long cnt = cs.count[a];
    double x = (i == cnt - 1 && cnt > 1) ? cs.hi[a] : cs.lo[a] + (double)i * cs.step[a];
    if (axc) {
      long long off = 0;
      for (int a2 = 0; a2 < a; ++a2) off += cs.count[a2];
      x = axc[off + i];
    }
    const T xn = ((T)x - (T)mc.X_mean[a]) / (T)mc.X_std[a];
    const T diff = xn * (T)mc.vinv[o][a] - As[((size_t)o * mc.npad + j) * D + a];
    *dst = exp_t<T>(T(-0.5) * (diff * diff));
  }
}

constexpr __host__ __device__ int NC_AX(int D) { return 1 + D; }

struct GridTables {
  size_t e0_stride;      // elements per output in E0f
  size_t er_stride_o;    // elements per output in Er
  size_t er_off[kMaxD];  // element offset of axis a's table inside one output's Er block
  long long nlines;      // local lines (n_local / count0)
  long long line0;       // global index of the first local line
};

// which row block carries the mean / gradient dot products of column block J
__device__ __forceinline__ int top_row_of_wave(int w, int nb) {
  // largest I < nb with row_block_of(r, w) == I
  int best = -1;   // the schedule is non-decreasing in r, so the last row below nb is the largest
  for (int r = 0;; ++r) {
    const int I = row_block_of(r, w);
    if (I >= nb) break;
    best = I;
  }
  return best;
}

// (three workgroups per CU asked for at D = 4: that instance needs 175 registers unconstrained, 7 above the limit for the
// occupancy the D = 2 instance runs at)
template <typename T, int S, int D, bool AX>
__global__ __launch_bounds__(256, (D == 4 ? 3 : 1)) void k_posterior_grid(const ModelConst mc, const CandSpec cs, const GridTables gt,
                                                        const T* __restrict__ Fpk, size_t fpk_stride,
                                                        const T* __restrict__ E0f, const T* __restrict__ Er,
                                                        const T* __restrict__ AXg, unsigned int ntiles,
                                                        T* __restrict__ mean_out, T* __restrict__ var_out,
                                                        unsigned long long* __restrict__ Lmax,
                                                        const double* __restrict__ axc, T* __restrict__ grad_out /* both nullptr,
                                                        or: explicit axis positions, signed gradient components [q][d][n_local] */) {
  using acc_t = typename MM<T>::acc_t;
  using a_t = typename MM<T>::a_t;
  constexpr int P = 16 * S;
  constexpr int NC = 2 + D;                       // per-candidate partial sums: quad, m0, ms[D]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int npad = mc.npad, nfr = npad >> 2, nb = npad >> 4;
  T* E1f = reinterpret_cast<T*>(smem);            // [S][npad]   sf2 * prod_{a>=1} table, in fragment-slot order
  T* part = E1f + (size_t)S * npad + 8;           // [NC][kWaves * 4][P] partial sums per (wave, k-slot); +8: prefetch slack
  __shared__ unsigned int line_row[2][S][kMaxD];  // per strip: row index into each remaining-axis table (x2: the
                                                  // epilogue of tile i overlaps phase 0 of tile i+1)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pp = lane & 15, slot = lane >> 4;
  const int out = blockIdx.y;
  const unsigned int cnt0 = (unsigned int)cs.count[0];
  const unsigned int ntile0 = (cnt0 + 15u) / 16u;

  // Which row block carries the mean / gradient dot products of column block J: the top row of wave J % W while
  // J <= tmin (the smallest of the waves' top rows), row J itself above that.  All of it is wave-uniform.
  const int my_top = top_row_of_wave(wave, nb);
  int tmin = nb;
  for (int w = 0; w < kWaves; ++w) {
    const int t = top_row_of_wave(w, nb);
    tmin = t < tmin ? t : tmin;
  }
  const T* F_o = Fpk + (size_t)out * fpk_stride;
  const T* AX_o = AXg + (size_t)out * npad * NC_AX(D) + slot * NC_AX(D);
  double gmax = 0.0;                              // running max of |grad MEAN|_inf (threads of wave 0)

  // persistent over tiles: a workgroup lives for many tiles, so dispatch gaps and per-launch setup are paid once
  int par = 0;
  for (unsigned int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
    const unsigned int t0 = tile % ntile0;
    const unsigned int lt = tile / ntile0;        // tile of S lines

    // ---------------- phase 0: per-line factors into LDS ----------------
    if (tid >= 64 && tid < 64 + S) {
      const int s = tid - 64;
      long long line = (long long)lt * S + s;
      if (line >= gt.nlines) line = gt.nlines - 1;
      unsigned long long f = (unsigned long long)(gt.line0 + line);   // global line index -> (i1, i2, ...)
      for (int a = 1; a < cs.d; ++a) {
        const unsigned long long c = (unsigned long long)cs.count[a];
        line_row[par][s][a] = (unsigned int)(f % c);
        f /= c;
      }
    }
    __syncthreads();   // also orders the previous tile's epilogue (reads of `part`) before this tile's writes
    for (int idx = tid; idx < npad; idx += 256) {
      const int jj = idx >> 2, sl = idx & 3;
      const int j = ((jj >> 2) << 4) + MM<T>::jslot(jj & 3, sl);
#pragma unroll
      for (int s = 0; s < S; ++s) {
        T v = (T)mc.sf2[out];
        for (int a = 1; a < cs.d; ++a)
          v *= Er[(size_t)out * gt.er_stride_o + gt.er_off[a] + (size_t)line_row[par][s][a] * npad + j];
        E1f[s * npad + idx] = v;
      }
    }
    __syncthreads();

    // ---------------- phase 2: block-triangular contraction, B-fragments formed on the fly ----------------
    const T* E0_t = E0f + (size_t)out * gt.e0_stride + (size_t)t0 * nfr * 64;   // wave-uniform base, lane added per load
    T quad[S], m0[S], ms[S][D];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      quad[s] = 0;
      m0[s] = 0;
#pragma unroll
      for (int a = 0; a < D; ++a) ms[s][a] = 0;
    }
    for (int r = 0;; ++r) {
      const int I = row_block_of(r, wave);
      if (I >= nb) break;
      acc_t acc[S];
#pragma unroll
      for (int s = 0; s < S; ++s) acc[s] = acc_t{0, 0, 0, 0};
      // software pipeline: the two fragment loads of step st+1 are in flight behind the MFMAs of step st.  The
      // load after the last step of a row reads the next fragment in memory (both buffers carry padding), so the
      // loop has no clamp and only pointer bumps.
      const T* fp = F_o + (size_t)(I * (I + 1) / 2) * 256;   // packed (I, J) blocks of this row, 256 elements each
      const T* ep = E0_t;
      a_t a_nx = MM<T>::load_a(fp, lane, 0);
      T e_nx = ep[lane];
      const T* e1p = E1f + slot;                   // per-line factors of step st at e1p[s * npad + st * 4]
      T e1_nx[S];
#pragma unroll
      for (int s = 0; s < S; ++s) e1_nx[s] = e1p[s * npad];
      const bool is_top = (I == my_top);
      // one k-step: consume the prefetched operands, launch the next step's loads, then the matrix-core step
      // (KK = -1: an off-diagonal block; KK = 0..3: k-step KK of the diagonal block, whose zero 4x4 sub-blocks are skipped)
      auto kstep = [&](auto kk_tag, auto diag_tag, const int J, const bool dots) {
        constexpr int kk = decltype(kk_tag)::value;
        constexpr bool diag = decltype(diag_tag)::value;
        const a_t a = a_nx;
        T b[S];
#pragma unroll
        for (int s = 0; s < S; ++s) b[s] = e_nx * e1_nx[s];
        // operands of the next step (global fragments and LDS factors) fly behind this step's MFMAs; the reads
        // past the row's last step stay inside the padded buffers / the LDS allocation
        ep += 64;
        e1p += 4;
        a_nx = (kk < 3) ? MM<T>::load_a(fp, lane, kk + 1) : MM<T>::load_a(fp + 256, lane, 0);
        if (kk == 3) fp += 256;
        e_nx = ep[lane];
#pragma unroll
        for (int s = 0; s < S; ++s) e1_nx[s] = e1p[s * npad];
#pragma unroll
        for (int s = 0; s < S; ++s) {
          if constexpr (diag) acc[s] = MM<T>::template mfma_diag<kk>(a, b[s], acc[s]);
          else acc[s] = MM<T>::mfma(a, b[s], acc[s]);
        }
        if (dots) {
          const T* ax = AX_o + (J * 16 + kk * 4) * NC_AX(D);
#pragma unroll
          for (int s = 0; s < S; ++s) {
            m0[s] = fma(ax[0], b[s], m0[s]);
#pragma unroll
            for (int a_ = 0; a_ < D; ++a_) ms[s][a_] = fma(ax[1 + a_], b[s], ms[s][a_]);
          }
        }
      };
      using std::integral_constant;
      for (int J = 0; J < I; ++J) {
        const bool dots = (J <= tmin) ? (is_top && (J % kWaves) == wave) : false;
        kstep(integral_constant<int, 0>{}, integral_constant<bool, false>{}, J, dots);
        kstep(integral_constant<int, 1>{}, integral_constant<bool, false>{}, J, dots);
        kstep(integral_constant<int, 2>{}, integral_constant<bool, false>{}, J, dots);
        kstep(integral_constant<int, 3>{}, integral_constant<bool, false>{}, J, dots);
      }
      {
        const bool dots = (I <= tmin) ? (is_top && (I % kWaves) == wave) : true;
        kstep(integral_constant<int, 0>{}, integral_constant<bool, true>{}, I, dots);
        kstep(integral_constant<int, 1>{}, integral_constant<bool, true>{}, I, dots);
        kstep(integral_constant<int, 2>{}, integral_constant<bool, true>{}, I, dots);
        kstep(integral_constant<int, 3>{}, integral_constant<bool, true>{}, I, dots);
      }
#pragma unroll
      for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int r_ = 0; r_ < 4; ++r_) quad[s] = fma(acc[s][r_], acc[s][r_], quad[s]);
      }
    }
    // every lane parks its partial sums; 16 (wave, k-slot) partials per candidate are added in a fixed order
    {
      const int row = (wave * 4 + slot) * P + pp;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        part[(size_t)0 * kWaves * 4 * P + row + s * 16] = quad[s];
        part[(size_t)1 * kWaves * 4 * P + row + s * 16] = m0[s];
#pragma unroll
        for (int a = 0; a < D; ++a) part[(size_t)(2 + a) * kWaves * 4 * P + row + s * 16] = ms[s][a];
      }
    }
    __syncthreads();

    // ---------------- epilogue: one thread per candidate (the other waves move on to the next tile) ----
    if (tid < P) {
      const int s = tid >> 4, p0 = tid & 15;
      const unsigned int g0 = t0 * 16 + p0;
      const long long line = (long long)lt * S + s;
      const bool valid = (g0 < cnt0) && (line < gt.nlines);
      if (valid) {
        const long long g = line * cnt0 + g0;
        T sums[NC];
#pragma unroll
        for (int c_ = 0; c_ < NC; ++c_) {
          T acc_ = 0;
#pragma unroll
          for (int w = 0; w < kWaves * 4; ++w) acc_ += part[((size_t)c_ * kWaves * 4 + w) * P + tid];
          sums[c_] = acc_;
        }
        const T sf2 = (T)mc.sf2[out], ystd = (T)mc.Y_std[out];
        const T mean = (T)mc.mp[out] + sums[1];
        T var = sf2 - sums[0];
        var = var > T(0) ? var : T(0);
        mean_out[(size_t)out * cs.n_local + g] = mean * ystd + (T)mc.Y_mean[out];
        var_out[(size_t)out * cs.n_local + g] = var * (ystd * ystd);
        // gradient of the un-normalised mean; the grid indices of this candidate are (g0, line_row[s][1..])
        T gn = 0;
#pragma unroll
        for (int a = 0; a < D; ++a) {
          if (a < mc.d) {
            const unsigned int ia = a == 0 ? g0 : line_row[par][s][a];
            const long long cnt = cs.count[a];
            double x = (ia == cnt - 1 && cnt > 1) ? cs.hi[a] : cs.lo[a] + (double)ia * cs.step[a];
            if (AX) {
              long long off = 0;
              for (int a2 = 0; a2 < a; ++a2) off += cs.count[a2];
              x = axc[off + ia];
            }
            const T rstd = (T)mc.X_rstd[a];
            const T xn = ((T)x - (T)mc.X_mean[a]) * rstd;
            T ga = ystd * (sums[2 + a] - xn * sums[1]) * (T)mc.inv_ell[out][a] * rstd;
            if (AX) grad_out[((size_t)out * mc.d + a) * cs.n_local + g] = ga;
            ga = ga < 0 ? -ga : ga;
            gn = ga > gn ? ga : gn;
          }
        }
        gmax = (double)gn > gmax ? (double)gn : gmax;
      }
    }
  }
  if (tid < (P < 64 ? 64 : P)) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double other = __shfl_xor(gmax, o);
      gmax = other > gmax ? other : gmax;
    }
    if (lane == 0) atomicMax(&Lmax[out], (unsigned long long)__double_as_longlong(gmax));
  }
}

// alpha_j * (1, Xn_j0, ..) in fragment-slot order [q][npad][1 + D] (the mean / gradient dot-product rows)
template <typename T>
__global__ void k_build_ax(const ModelConst mc, const T* __restrict__ alpha, const T* __restrict__ Xn, int D,
                           T* __restrict__ AX) {
  const int o = blockIdx.y;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < mc.npad; idx += gridDim.x * blockDim.x) {
    const int jj = idx >> 2, sl = idx & 3;
    const int j = ((jj >> 2) << 4) + MM<T>::jslot(jj & 3, sl);
    const T al = alpha[(size_t)o * mc.npad + j];
    T* dst = AX + ((size_t)o * mc.npad + idx) * (1 + D);
    dst[0] = al;
    for (int a = 0; a < D; ++a) dst[1 + a] = al * Xn[j * D + a];
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

template <typename T, int D>
static int launch_posterior_chunked_t(sbo_ctx* c) {
  const ModelConst& mc = c->mc;
  constexpr int P = 16 * kWaves, CB = ChunkCfg<T>::CB;
  const size_t lds = sizeof(T) * ((size_t)kWaves * CB * 4 * 64 + kWaves * P + (size_t)P * (1 + D));
  auto kern = k_posterior_chunked<T, D>;
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long tiles = (c->cs.n_local + P - 1) / P;
  if (tiles > 0x7fffffffLL) return fail(SBO_E_UNSUPPORTED, "too many candidate tiles for one launch");
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles, (unsigned)mc.q), dim3(256), lds, c->stream, mc, c->cs, (const T*)c->Fpk.p,
                     c->fpk_stride, (const T*)c->As.p, (const T*)c->sqA.p, (const T*)c->alpha.p, (const T*)c->Xn.p,
                     (T*)c->mean.p, (T*)c->var.p, (unsigned long long*)c->Lmax.p);
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

template <typename T>
static int launch_posterior_chunked(sbo_ctx* c) {
  switch (c->mc.dpad) {
    case 2: return launch_posterior_chunked_t<T, 2>(c);
    case 4: return launch_posterior_chunked_t<T, 4>(c);
    case 8: return launch_posterior_chunked_t<T, 8>(c);
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

template <typename T, int D, int S>
static int launch_posterior_grid_ts(sbo_ctx* c) {
  constexpr int P = 16 * S;
  const ModelConst& mc = c->mc;
  const CandSpec& cs = c->cs;
  const int npad = mc.npad, nfr = npad / 4, q = mc.q, d = cs.d;
  const long long cnt0 = cs.count[0];
  const long long ntile0 = (cnt0 + 15) / 16;
  GridTables gt;
  memset(&gt, 0, sizeof(gt));
  gt.e0_stride = (size_t)ntile0 * nfr * 64;
  size_t off = 0;
  for (int a = 1; a < d; ++a) {
    gt.er_off[a] = off;
    off += (size_t)cs.count[a] * npad;
  }
  gt.er_stride_o = off;
  gt.nlines = cs.n_local / cnt0;
  gt.line0 = cs.first / cnt0;
  int rc;
  if ((rc = ensure(c->E0f, sizeof(T) * (gt.e0_stride * q + 512)))) return rc;   // + padding: the pipeline over-reads one fragment
  if ((rc = ensure(c->Er, sizeof(T) * std::max<size_t>(gt.er_stride_o, 1) * q))) return rc;
  const long long total = (long long)gt.e0_stride + (long long)gt.er_stride_o;
  hipLaunchKernelGGL((k_build_tables<T>), dim3((unsigned)std::min<long long>((total + 255) / 256, 4096), q), dim3(256), 0,
                     c->stream, mc, cs, (const T*)c->As.p, mc.dpad, (T*)c->E0f.p, gt.e0_stride, (T*)c->Er.p, gt.er_stride_o,
                     gt.er_off[1], gt.er_off[2], gt.er_off[3], gt.er_off[4], gt.er_off[5], gt.er_off[6], gt.er_off[7], c->k1g_axc);
  if ((rc = ensure(c->AXg, sizeof(T) * (size_t)q * npad * (1 + D)))) return rc;
  hipLaunchKernelGGL((k_build_ax<T>), dim3(1, q), dim3(256), 0, c->stream, mc, (const T*)c->alpha.p, (const T*)c->Xn.p, D,
                     (T*)c->AXg.p);
  const size_t lds = sizeof(T) * ((size_t)S * npad + 8 + (size_t)(2 + D) * kWaves * 4 * P);
  // (explicit axis positions + gradient output: its own instance, fp64 grids of up to four axes only -- the plain one keeps its registers)
  const bool ax = c->k1g_axc != nullptr;
  if (ax && !(sizeof(T) == 8 && D == 4 && S == 4)) return fail(SBO_E_UNSUPPORTED, "internal: explicit axes are an fp64 path of three / four axes");
  auto kern = ax ? k_posterior_grid<T, S, D, (sizeof(T) == 8 && D == 4 && S == 4)> : k_posterior_grid<T, S, D, false>;
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long tiles = ntile0 * ((gt.nlines + S - 1) / S);
  if (tiles > 0x7fffffffLL) return fail(SBO_E_UNSUPPORTED, "too many candidate tiles for one launch");
  // persistent grid: a few workgroups per CU (as many as registers / LDS admit), each looping over tiles
  int per_cu = 0;
  SBO_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds));
  per_cu = std::max(1, std::min(per_cu, 4));
  // the q outputs are separate grid rows: share the CU slots between them
  const long long wgs = std::min<long long>(tiles, ((long long)c->n_cu * per_cu + q - 1) / q);
  hipLaunchKernelGGL(kern, dim3((unsigned)std::max<long long>(wgs, 1), (unsigned)q), dim3(256), lds, c->stream, mc, cs, gt,
                     (const T*)c->Fpk.p, c->fpk_stride, (const T*)c->E0f.p, (const T*)c->Er.p, (const T*)c->AXg.p,
                     (unsigned int)tiles, (T*)c->mean.p, (T*)c->var.p, (unsigned long long*)c->Lmax.p, c->k1g_axc, (T*)c->k1g_grad);
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

template <typename T, int D>
static int launch_posterior_grid_t(sbo_ctx* c) {
  return launch_posterior_grid_ts<T, D, 4>(c);
}

template <typename T>
static int launch_posterior_grid(sbo_ctx* c) {
  switch (c->mc.dpad) {
    case 2: return launch_posterior_grid_t<T, 2>(c);
    case 4: return launch_posterior_grid_t<T, 4>(c);
    case 8: return launch_posterior_grid_t<T, 8>(c);
  }
  return fail(SBO_E_UNSUPPORTED, "unsupported padded dimension");
}

// the separable path needs whole axis-0 lines in the shard
static bool grid_path_ok(const sbo_ctx* c) {
  const CandSpec& cs = c->cs;
  if (c->posterior_path != 0) return false;
  if (cs.kind != 1 || cs.n_local <= 0) return false;
  const long long cnt0 = cs.count[0];
  return cs.first % cnt0 == 0 && cs.n_local % cnt0 == 0;
}

int launch_posterior(sbo_ctx* c) {
  c->gb_active = false;                    // (the approximating paths K1b / K1t switch their guard band on themselves)
  // block-triangular contraction as issued: npad (npad + 16) / 2 multiply-adds per candidate and output
  const double tri_flops = (double)c->mc.q * c->mc.npad * (c->mc.npad + 16.0) * (double)c->cs.n_local;
  if (grid_path_ok(c)) {
    // fp64 2-D grids: two GEMMs in a reduced basis (K1b) when the axis bases qualify, else the separable tables (K1g)
    if (bilinear_applicable(c)) {
      int rc;
      // the first sweep of a model: interpolation from Chebyshev nodes (K1i, enqueued by sbo_model_set -- or here, when the grid
      // came after the model); K1b's plan is built when the same model is swept again
      if (!c->bl.valid && interp_applicable(c)) {
        if (!(c->bi.valid && c->bi.serial == c->model_serial) && (rc = interp_setup(c))) return rc;
        if (c->bi.usable && !c->bi.used) {
          c->last_k1 = 6;
          return launch_posterior_interp(c);
        }
      }
      if (!c->bl.valid && (rc = bilinear_setup(c))) return rc;
      if (c->bl.usable) {
        c->last_k1 = 4;
        return launch_posterior_bilinear(c);       // (writes the Lipschitz keys itself)
      }
    }
    // fp64 grids of three / four axes: exact values at Chebyshev nodes, interpolated to the grid (K1t) when the plan qualifies
    if (!c->tensor_busy && tensor_applicable(c)) {
      bool declined = true;
      const int rct = launch_posterior_tensor(c, &declined);
      if (rct || !declined) return rct;
    }
    { const int rcf = factor_sync(c); if (rcf) return rcf; }     // (the O(n^2) kernels contract with the factor images)
    SBO_HIP(hipMemsetAsync(c->Lmax.p, 0, sizeof(unsigned long long) * kMaxQ, c->stream));
    c->last_k1 = 3;
    c->last_k1_flops = tri_flops;
    return c->dtype == SBO_F64 ? launch_posterior_grid<double>(c) : launch_posterior_grid<float>(c);
  }
  // generic candidates: the single-phase kernel while the whole K* tile of 64 candidates fits LDS with two workgroups
  // per CU, the chunked kernel beyond that (and on request: posterior_path 2)
  { const int rcf = factor_sync(c); if (rcf) return rcf; }
  SBO_HIP(hipMemsetAsync(c->Lmax.p, 0, sizeof(unsigned long long) * kMaxQ, c->stream));
  const size_t tile_bytes = (c->dtype == SBO_F64 ? 8u : 4u) * (size_t)c->mc.npad * 64;
  c->last_k1_flops = tri_flops;
  c->last_k1 = (c->posterior_path == 2 || tile_bytes > 64 * 1024) ? 2 : 1;
  if (c->posterior_path == 2 || tile_bytes > 64 * 1024)
    return c->dtype == SBO_F64 ? launch_posterior_chunked<double>(c) : launch_posterior_chunked<float>(c);
  return c->dtype == SBO_F64 ? launch_posterior_s<double>(c) : launch_posterior_s<float>(c);
}

// K1g on a tensor grid with explicit axis positions (fp64 models; K1t's Chebyshev nodes), into caller-given arrays: mean / var
// [q][N], signed gradient components of the mean [q][d][N], N = prod count, axis 0 fastest.  The context's candidate description
// and posterior buffers are swapped for the call; `lmax` receives the Lipschitz keys of this point set.
int launch_posterior_on_axes(sbo_ctx* c, int d, const long long* count, const double* axc, double* mean_out, double* var_out, double* grad_out,
                             unsigned long long* lmax) {
  if (c->dtype != SBO_F64) return fail(SBO_E_UNSUPPORTED, "internal: explicit axes are an fp64 path");
  { const int rcf = factor_sync(c); if (rcf) return rcf; }
  const CandSpec keep_cs = c->cs;
  const DevBuf keep_m = c->mean, keep_v = c->var, keep_l = c->Lmax;
  memset(&c->cs, 0, sizeof(c->cs));
  c->cs.kind = 1;
  c->cs.d = d;
  long long N = 1;
  for (int a = 0; a < d; ++a) { c->cs.count[a] = count[a]; N *= count[a]; }
  for (int a = d; a < kMaxD; ++a) c->cs.count[a] = 1;
  c->cs.n_local = N;
  c->mean.p = mean_out;
  c->var.p = var_out;
  c->Lmax.p = lmax;
  c->k1g_axc = axc;
  c->k1g_grad = grad_out;
  int rc = hipMemsetAsync(lmax, 0, sizeof(unsigned long long) * kMaxQ, c->stream) == hipSuccess ? SBO_OK : fail(SBO_E_HIP, "memset of the key scratch");
  if (!rc) rc = launch_posterior_grid<double>(c);
  c->k1g_axc = nullptr;
  c->k1g_grad = nullptr;
  c->cs = keep_cs;
  c->mean = keep_m;
  c->var = keep_v;
  c->Lmax = keep_l;
  return rc;
}

// the exact posterior of the generic kernel on an explicit fp64 list, into caller-given arrays (the context's candidate
// description and posterior buffers are swapped for the call; its plans and flags are left as they were)
int launch_posterior_on_list(sbo_ctx* c, const double* pts, long long N, double* mean_out, double* var_out) {
  if (c->dtype != SBO_F64) return fail(SBO_E_UNSUPPORTED, "internal: exact lists are an fp64 path");
  const CandSpec keep_cs = c->cs;
  const DevBuf keep_m = c->mean, keep_v = c->var, keep_l = c->Lmax;
  const int keep_k1 = c->last_k1;
  const double keep_flops = c->last_k1_flops;
  const bool keep_gb = c->gb_active, keep_busy = c->tensor_busy;
  int rc;
  if ((rc = ensure(c->list_scr, 512))) return rc;
  memset(&c->cs, 0, sizeof(c->cs));
  c->cs.kind = 0;
  c->cs.d = keep_cs.d;
  c->cs.pts_dtype = SBO_F64;
  c->cs.pts = pts;
  c->cs.n_local = N;
  c->cs.first = 0;
  c->mean.p = mean_out;
  c->var.p = var_out;
  c->Lmax.p = c->list_scr.p;                   // (the Lipschitz keys of the list are of no interest)
  c->tensor_busy = true;                       // (launch_posterior must not come back to the tensor path)
  rc = launch_posterior(c);
  c->tensor_busy = keep_busy;
  c->cs = keep_cs;
  c->mean = keep_m;
  c->var = keep_v;
  c->Lmax = keep_l;
  c->last_k1 = keep_k1;
  c->last_k1_flops = keep_flops;
  c->gb_active = keep_gb;
  return rc;
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
