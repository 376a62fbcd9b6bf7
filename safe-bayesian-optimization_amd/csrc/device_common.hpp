// device_common.hpp -- device helpers shared by the kernels of libsafebo.so (gfx950 only).
#pragma once
#include "internal.hpp"

namespace sbo {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));

// 16x16x4 MFMA per compute type.  A/B operands: one scalar per lane, lane l holds A[i = l&15][k = l>>4]
// and B[k = l>>4][j = l&15].  C/D: f32 row = 4*(l>>4) + r, f64 row = (l>>4) + 4*r, col = l&15
// (cdna_hip_programming.md section 3; verified with exact integer data by tools/mfma_probe.hip).
// jslot(kk, slot) is the observation offset inside a 16-block that k-step kk / k-slot `slot` carries; it is
// chosen per type so that accumulator register r of block I sits on the lane that also holds B-fragment
// (J = I, kk = r), i.e. t_i and k_i meet in one lane with no shuffle.
template <typename T> struct MM;
template <> struct MM<double> {
  using acc_t = d4_t;
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __host__ __device__ __forceinline__ int jslot(int kk, int slot) { return 4 * kk + slot; }
};
template <> struct MM<float> {
  using acc_t = f4_t;
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __host__ __device__ __forceinline__ int jslot(int kk, int slot) { return 4 * slot + kk; }
};

// unfused arithmetic where the oracle's rounding sequence is part of the contract
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ double sub_rn(double a, double b) { return __dsub_rn(a, b); }
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float sub_rn(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ double sqrt_rn(double a) { return __dsqrt_rn(a); }
__device__ __forceinline__ float sqrt_rn(float a) { return __fsqrt_rn(a); }

// order-preserving map double -> uint64 (for atomicMin / atomicMax on signed values)
__host__ __device__ __forceinline__ unsigned long long ord_key(double v) {
  unsigned long long b;
#if defined(__HIP_DEVICE_COMPILE__)
  b = (unsigned long long)__double_as_longlong(v);
#else
  __builtin_memcpy(&b, &v, 8);
#endif
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ __forceinline__ double ord_val(unsigned long long k) {
  unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  double v;
#if defined(__HIP_DEVICE_COMPILE__)
  v = __longlong_as_double((long long)b);
#else
  __builtin_memcpy(&v, &b, 8);
#endif
  return v;
}

// Candidate coordinates of local candidate g (raw, un-normalised), always in double.
// Grid: x_a(i) = lo_a + i*step_a, last point = hi_a exactly (oracle.grid_axes), flat index axis 0 fastest
// (test/test_SafeOpt.py:325-334).
template <int D>
__device__ __forceinline__ void cand_coords(const CandSpec& cs, long long g, double (&x)[D]) {
  if (cs.kind == 0) {
    if (cs.pts_dtype == SBO_F64) {
      const double* p = (const double*)cs.pts + g * cs.d;
#pragma unroll
      for (int a = 0; a < D; ++a) x[a] = (a < cs.d) ? p[a] : 0.0;
    } else {
      const float* p = (const float*)cs.pts + g * cs.d;
#pragma unroll
      for (int a = 0; a < D; ++a) x[a] = (a < cs.d) ? (double)p[a] : 0.0;
    }
  } else {
    long long f = cs.first + g;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      if (a < cs.d) {
        const long long cnt = cs.count[a];
        const long long i = f % cnt;
        f /= cnt;
        x[a] = (i == cnt - 1 && cnt > 1) ? cs.hi[a] : __dadd_rn(cs.lo[a], __dmul_rn((double)i, cs.step[a]));
      } else {
        x[a] = 0.0;
      }
    }
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

}  // namespace sbo
