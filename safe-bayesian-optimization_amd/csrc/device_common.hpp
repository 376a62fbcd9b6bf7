// device_common.hpp -- device helpers shared by the kernels of libsafebo.so (gfx950 only).
#pragma once
#include "internal.hpp"

namespace sbo {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));

// Matrix-core step per compute type: one call multiplies a 16(rows) x 4(k) A-fragment with a 4(k) x 16(candidates)
// B-fragment and accumulates 16 x 16 results, 4 per lane.  B lane map (both types): lane l holds
// B[k = l>>4][candidate = l&15]; every accumulator element of lane l belongs to candidate l&15 (which row it is
// does not matter: the epilogue sums squares over all rows).
//   float : v_mfma_f32_16x16x4_f32 (measured 155 TFLOP/s = the f32 peak); A lane map A[row = l&15][k = l>>4].
//   double: v_mfma_f64_16x16x4_f64 issues only every ~105 cycles on gfx950 (47.9 TFLOP/s measured,
//           profiles/r01_mfma_probe.txt) while v_mfma_f64_4x4x4_4b_f64 sustains 72-75 TFLOP/s, so the f64 step is
//           four 4x4x4 instructions.  Lane layout of that form (found with one-hot data, tools/mfma_probe4.hip):
//           A[blk][i][k] at lane 16k+4blk+i, B[blk][k][j] at 16k+4blk+j, D[blk][i][j] at 16i+4blk+j, four
//           independent 4x4x4 products.  Instruction t gives every block the SAME four rows 4t..4t+3 of the
//           A-fragment (its four lane-quads read identical addresses, so the load still moves 128 unique bytes)
//           and block blk the candidates 4blk..4blk+3 of the natural B-fragment: no lane movement at all.
// jslot(kk, slot) is the observation offset inside a 16-block carried by k-step kk / k-slot `slot`;
// pack_pos(r, k, kk) is where element (row r, k-slot k, k-step kk) of a 16x16 block sits in its 256-element packed
// image, chosen so that one lane's operands are contiguous (f64: the 4 row-groups of a k-step = 32 bytes, two
// dwordx4 loads; f32: the 4 k-steps of a block = 16 bytes, one dwordx4 load).
template <typename T> struct MM;
template <> struct MM<float> {
  using acc_t = f4_t;
  using a_t = float;
  // block = the 256 packed elements of one (I, J) block; a lane's four k-steps are contiguous (one dwordx4)
  static __device__ __forceinline__ a_t load_a(const float* block, int lane, int kk) { return block[lane * 4 + kk]; }
  static __device__ __forceinline__ void load_a4(const float* block, int lane, a_t (&a)[4]) {
    const f4_t v = *reinterpret_cast<const f4_t*>(block + lane * 4);
    a[0] = v[0]; a[1] = v[1]; a[2] = v[2]; a[3] = v[3];
  }
  static __device__ __forceinline__ acc_t mfma(a_t a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // k-step kk of a diagonal block: the 16x16x4 form cannot skip the zero part of the triangle
  template <int KK>
  static __device__ __forceinline__ acc_t mfma_diag(a_t a, float b, acc_t c) { return mfma(a, b, c); }
  static __host__ __device__ __forceinline__ int jslot(int kk, int slot) { return 4 * slot + kk; }
  static __host__ __device__ __forceinline__ int pack_pos(int r, int k, int kk) { return (k * 16 + r) * 4 + kk; }
  static __host__ __device__ __forceinline__ void unpack_pos(int e, int& r, int& k, int& kk) { kk = e & 3; r = (e >> 2) & 15; k = e >> 6; }
};
template <> struct MM<double> {
  using acc_t = d4_t;
  using a_t = d4_t;   // rows {i, 4+i, 8+i, 12+i} (i = lane&3) at k-slot lane>>4
  static __device__ __forceinline__ a_t load_a(const double* block, int lane, int kk) {
    return *reinterpret_cast<const d4_t*>(block + kk * 64 + (((lane >> 4) << 2) + (lane & 3)) * 4);
  }
  static __device__ __forceinline__ void load_a4(const double* block, int lane, a_t (&a)[4]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) a[kk] = load_a(block, lane, kk);
  }
  static __device__ __forceinline__ acc_t mfma(a_t a, double b, acc_t c) {
    c[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], b, c[0], 0, 0, 0);
    c[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], b, c[1], 0, 0, 0);
    c[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[2], b, c[2], 0, 0, 0);
    c[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[3], b, c[3], 0, 0, 0);
    return c;
  }
  // k-step kk of a DIAGONAL block of the lower-triangular factor: row group t (rows 4t..4t+3) times column group kk
  // (columns 4kk..4kk+3) is identically zero for kk > t, so those 4x4x4 products are skipped (6 of 16 per block)
  template <int KK>
  static __device__ __forceinline__ acc_t mfma_diag(a_t a, double b, acc_t c) {
    if (KK <= 0) c[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], b, c[0], 0, 0, 0);
    if (KK <= 1) c[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], b, c[1], 0, 0, 0);
    if (KK <= 2) c[2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[2], b, c[2], 0, 0, 0);
    c[3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[3], b, c[3], 0, 0, 0);
    return c;
  }
  static __host__ __device__ __forceinline__ int jslot(int kk, int slot) { return 4 * kk + slot; }
  static __host__ __device__ __forceinline__ int pack_pos(int r, int k, int kk) { return kk * 64 + (k * 4 + (r & 3)) * 4 + (r >> 2); }
  static __host__ __device__ __forceinline__ void unpack_pos(int e, int& r, int& k, int& kk) {
    kk = e >> 6;
    const int rem = e & 63;
    k = rem >> 4;
    r = ((rem >> 2) & 3) + 4 * (rem & 3);
  }
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

// Sign of lcb = fl(m - sd), sd = fl(b fl(sqrt v)) (models/SafeOpt.py:40-45) WITHOUT the square root where that is safe.
// A difference of two doubles has the sign of the exact difference, so lcb >= 0 <=> m >= sd and lcb <= 0 <=> m <= sd, and sd
// lies within 2.0001 * 2^-53 (relative) of b sqrt(v): comparing m^2 with b^2 v decides every candidate whose bound is not
// within a few ulp of zero; only those take the IEEE square root (a dozen dependent f64 instructions on the datapath the
// matrix cores share).  The decisions are the reference's bit for bit.  bb = fl(b b).
struct LcbSign {
  bool ge, le;      // lcb >= 0 (the S test, models/SafeOpt.py:57-59), lcb <= 0 (the U test, :73-77)
};
__device__ __forceinline__ LcbSign lcb_sign(double m, double v, double b, double bb) {
  if (b >= 0.0 && v >= 0.0) {
    if (m < 0.0) return LcbSign{false, true};                       // sd >= 0 > m
    const double P = m * m, Q = bb * v;                              // three roundings between them: relative 3 * 2^-53
    constexpr double c = 1.0 + 0x1p-48, tiny = 1e-250, huge = 1e300;
    if (P < huge && Q < huge) {
      if (P > tiny && P >= Q * c) return LcbSign{true, false};      // m > sd for sure
      if (Q > tiny && P * c <= Q) return LcbSign{false, true};      // m < sd for sure
    }
  }
  const double sd = mul_rn(b, sqrt_rn(v));
  return LcbSign{m >= sd, m <= sd};
}
// ---- guard band of an approximating posterior (r04) ---------------------------------------------------------------------------
// K1b (Chebyshev core) and K1t (Chebyshev-node interpolation) deliver mean / var within (dm[o], dv[o]) of the exact fp64 kernel
// (un-normalised units; measured per plan against exactly evaluated probe points, plus the analytic truncation tail) and the
// Lipschitz keys within a relative rl[o].  Every decision of a sweep that such a deviation could move -- the sign of a
// constraint's lcb, lcb_0 <= u*, an arg-max / arg-min, an expander / optimistic-set verdict -- is COUNTED on the fast path
// (SweepScalars::n_guard); a sweep with a non-zero count re-evaluates the candidates concerned with the exact kernel and runs
// its set phase again (sets_recheck.inc.hpp), so that the masks and indices it returns are those of the exact posterior.
// The block lives in device memory: K1b computes its band on the device (no host round trip in a model change).
// r05: the band has two parts.  an_m / an_v: what the plan can BOUND -- the truncations it makes (Chebyshev degree and rank of the
// axis bases, dropped coefficients of the 2-D series, the aliasing estimate of an interpolant), carried through the posterior formula.
// kGbSafety x pr_m / pr_v (the largest deviation at the plan's probe points) + a floor: the ROUNDING of this plan's sums, which is
// measured, because its a-priori bound is useless as a band -- gamma_n sum |alpha_j k_j| for the mean and gamma_2n |k|^T |A| |k| for the
// variance, 1e-11 and 1e-7 on config H, five orders above what the sums actually lose; the reference's own jnp.dot is only defined
// to that bound (models/GP_Safe.py:341-343).  That a-priori bound is the CHECK: a probe deviation that truncation bound + worst-case
// rounding do not explain means the plan errs in a way nothing here describes, and it is not used at all (band = infinite: every
// sweep on it re-evaluates with the exact kernel).
struct GuardBand {
  double dm[kMaxQ], dv[kMaxQ], rl[kMaxQ];
  double an_m[kMaxQ], an_v[kMaxQ], pr_m[kMaxQ], pr_v[kMaxQ];
};
constexpr double kGbSafety = 16.0;
// worst-case rounding of the reference formula's sums in any order (un-normalised units): n eps sum_j |alpha_j k_j| <= n eps sf2 ||alpha||_1
// and 2 n eps |k|^T |A| |k| <= 2 n eps ||k||_2^2 || |A| ||_2 <= 2 n eps (n sf2^2) (sqrt(n) / sn2)
__device__ __forceinline__ double gb_round_mean(int n, double sf2, double a1, double ys) { return (double)n * 2.220446049250313e-16 * sf2 * a1 * ys; }
__device__ __forceinline__ double gb_round_var(int n, double sf2, double sn2, double ys) {
  return 2.0 * (double)n * 2.220446049250313e-16 * ((double)n * sf2 * sf2) * (sqrt((double)n) / sn2) * ys * ys;
}
// K1b's probe points: a kGbProbe1 x kGbProbe1 tensor of the grid positions nearest to the Chebyshev extrema of each axis (ends
// included -- a polynomial surrogate errs most there); local index of probe p on the resident grid
constexpr int kGbProbe1 = 12;
constexpr int kGbProbes = kGbProbe1 * kGbProbe1;
__device__ __forceinline__ void gb_probe_xy(const CandSpec& cs, long long nlines, int p, long long& x0, long long& x1) {
  const int i0 = p % kGbProbe1, i1 = p / kGbProbe1;
  x0 = (long long)llrint(0.5 * (1.0 - cospi((double)i0 / (double)(kGbProbe1 - 1))) * (double)(cs.count[0] - 1));
  x1 = (long long)llrint(0.5 * (1.0 - cospi((double)i1 / (double)(kGbProbe1 - 1))) * (double)(nlines - 1));
}
__device__ __forceinline__ long long gb_probe_index(const CandSpec& cs, long long nlines, int p) {
  long long x0, x1;
  gb_probe_xy(cs, nlines, p, x0, x1);
  return x1 * cs.count[0] + x0;
}
// |sqrt(v') - sqrt(v)| for |v' - v| <= dv (both clipped at zero): dv / sqrt(v) while v >= dv, sqrt(dv) below
__device__ __forceinline__ double gb_dsqrt(double v, double dv) {
  return v >= dv ? dv / __dsqrt_rn(v) : __dsqrt_rn(dv);
}
// Could a deviation of (dm, dv) move the sign of lcb = m - b sqrt(v) (either test of LcbSign)?  Sqrt-free and conservative: with
// P = m^2, Q = b^2 v the sign of m - b sqrt(v) for m >= 0 is that of P - Q, and P - Q moves by at most 2 |m| dm + dm^2 + b^2 dv;
// for m < -dm the sign is settled (b sqrt(v) >= 0), and a flagged candidate there is a false alarm.  The band's floor (64 eps of
// the values' scale, guard.hip) is ~50 x the rounding of P and Q themselves; the constants carry another 10 %.
struct LcbBand {
  double c1, c0;      // |P - Q| <= |m| c1 + c0  <=>  "near"
};
__device__ __forceinline__ LcbBand lcb_band(double bb, double dm, double dv) { return LcbBand{2.2 * dm, 1.1 * fma(dm, dm, bb * dv)}; }
// lcb_sign with the band test sharing its products (the fused epilogue of the GEMM posterior pays for every f64 instruction)
__device__ __forceinline__ LcbSign lcb_sign_gb(double m, double v, double b, double bb, const LcbBand& lb, int& near) {
  const double P = m * m, Q = bb * v;
  const double gap = P > Q ? P - Q : Q - P, am = m < 0 ? -m : m;
  near += !(gap > fma(am, lb.c1, lb.c0));                              // (NaN operands count as near)
  if (b >= 0.0 && v >= 0.0) {
    if (m < 0.0) return LcbSign{false, true};                         // sd >= 0 > m
    constexpr double c = 1.0 + 0x1p-48, tiny = 1e-250, huge = 1e300;
    if (P < huge && Q < huge) {
      if (P > tiny && P >= Q * c) return LcbSign{true, false};        // m > sd for sure
      if (Q > tiny && P * c <= Q) return LcbSign{false, true};        // m < sd for sure
    }
  }
  const double sd = mul_rn(b, sqrt_rn(v));
  return LcbSign{m >= sd, m <= sd};
}

// bounds of ucb = fl(m + fl(b fl(sqrt v))) from a single-precision square root: the reductions over ucb (largest ucb_c over
// S, smallest ucb_0 over S) evaluate the exact bound only for candidates that could move the running extremum
__device__ __forceinline__ double ucb_upper(double m, double v, double b) {
  const float sf = __fsqrt_rn((float)v);                            // (v beyond the float range: inf -- the exact path is taken)
  const double s_up = (double)sf * (1.0 + 0x1p-20) + 1e-18;
  const double x = m + b * s_up;
  return x + (x < 0 ? -x : x) * 0x1p-50;
}
__device__ __forceinline__ double ucb_lower(double m, double v, double b) {
  const float sf = __fsqrt_rn((float)v);
  double s_lo = (double)sf * (1.0 - 0x1p-20) - 1e-18;
  s_lo = s_lo > 0.0 ? s_lo : 0.0;
  if (!(sf < 3.0e38f)) s_lo = 0.0;                                  // (inf / NaN: no information)
  const double x = m + b * s_lo;
  return x - (x < 0 ? -x : x) * 0x1p-50;
}

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

// Lipschitz key of output o from K1b's per-wave partials: Lmax[o] = max (values >= 0, so the bit pattern orders them);
// one workgroup of 256 threads, `sh`: four doubles of LDS
__device__ __forceinline__ void lmax_reduce_body(int o, double* sh, const double* __restrict__ Lpart, int per_out,
                                                 unsigned long long* __restrict__ Lmax) {
  double g = 0.0;
  for (int i = threadIdx.x; i < per_out; i += blockDim.x) {
    const double v = Lpart[(size_t)o * per_out + i];
    g = v > g ? v : g;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double other = __shfl_xor(g, off);
    g = other > g ? other : g;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = g;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) g = sh[w] > g ? sh[w] : g;
    Lmax[o] = (unsigned long long)__double_as_longlong(g);
  }
}

}  // namespace sbo
