// sets.hip -- K3/K4/K5: safe-set classification, minimiser / expander sets and masked arg-reductions.
//
// Discretised form of the reference's constrained optimisation problems:
//   S  = {g : lcb_i(g) >= 0 for all constraints i}                       models/SafeOpt.py:57-59
//   u* = min_S ucb_0,  M = {g in S : lcb_0(g) <= u*}                      models/SafeOpt.py:47-51, 61-62
//   U  = {h : lcb_i(h) <= 0 for all constraints i}                        models/SafeOpt.py:73-77, 109
//   G_c= {g in S : exists h in U, ucb_c(g) - L ||x_g - x_h + 1e-8|| >= 0}  models/SafeOpt.py:85-88, 111
//   acquisition: argmax var_0 over M and over each G_c                    models/SafeOpt.py:55, 65-66, 92, 117-124
// These passes are HBM-bound (a few bytes per candidate); the arithmetic that decides a mask bit is written
// with unfused multiplies/adds in the oracle's order so the masks are reproducible bit for bit from a given
// mean/var.  The expander query "is some fully-unsafe point within ucb/L of g" is answered with an exact
// separable Euclidean distance transform of the U mask on the grid, bounded by the largest radius that can
// matter; decisions that fall inside the rounding band of the reference's "+1e-8" are re-decided by
// exhaustive evaluation of the reference expression, so the transform never changes a mask bit.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <hip/hip_ext.h>
#include "device_common.hpp"

namespace sbo {

int comm_allreduce_max_u64(sbo_ctx* c, unsigned long long* dev, int count);
int comm_allreduce_min_u64(sbo_ctx* c, unsigned long long* dev, int count);
int comm_allreduce_sum_f64(sbo_ctx* c, double* dev, int count);
int comm_allgather_bytes(sbo_ctx* c, const void* send, void* recv, size_t bytes_per_rank);

constexpr double kInfD = 1.0e300;
constexpr int kArgSlots = 16;

// small device-resident scalar block of one sweep.  The fields up to kScalHost are what the host reads back (from pinned,
// uncached memory: every cache line of it costs the sweep's tail), the rest stays on the device.
struct SweepScalars {
  unsigned long long ustar_key;            // min over S of ord_key(ucb_0)
  unsigned long long rmax_key[kMaxQ];      // max over S of ord_key(ucb_c)
  long long count_S, count_U, count_M;
  long long count_set[kMaxQ];              // |G_c| or |O_c|
  long long n_amb, n_amb_total;
  long long n_scan;                        // candidates handed from the coarse decision to the wave-per-candidate scan
  long long halo_short;                    // ranks > 1: a speculative transform window was narrower than this sweep's keys need
  double arg_val[kArgSlots];
  long long arg_idx[kArgSlots];
  // Guard band of an approximating posterior (device_common.hpp: GuardBand); all zero when the posterior kernel is exact.
  // n_guard: decisions of this sweep inside the band, from the classification (set by its merge) and the verdict kernels
  // (atomics: rare); guard_slot[s]: the same from slot s of the final reductions (M-band members, an undecided arg-reduction);
  // arg_d: the band of a slot's winner
  double arg_d[kArgSlots];
  long long n_guard;
  long long guard_slot[kArgSlots];
  // ---- device only ----
  unsigned long long ticket;               // workgroups of k_goose_finals that have finished their slot (zeroed with the block)
  long long n_guard_cls;                   // the classification's share of n_guard, the value it starts from
  long long guard_nb0;                     // slot 0's share that is not its arg-reduction (members of M inside the band): ranks > 1 sum it
  unsigned long long vmin_key[kMaxQ];      // min over S of ord_key(var_o): bounds the band of a bound's square root on S
  double gb_du[kMaxQ];                     // band of ucb_o / lcb_o on S: dm_o + b |d sqrt(var_o)| at the smallest variance over S
  double gb_rl[kMaxQ];                     // relative band of the Lipschitz keys
  double arg_e1[kArgSlots], arg_e2[kArgSlots];   // arg-reductions: the two extreme far ends over all candidates and the first one's
  long long arg_ei[kArgSlots];             //   candidate (ranks > 1: merged by the host from the C3 rows)
};
constexpr size_t kScalHost = offsetof(SweepScalars, ticket);

// Result of a masked arg-reduction over values known to +- d.  (v, i): the winner, ties -> lowest flat index; d: its band;
// e1 / e2: the two most extreme FAR ends over all candidates (v + d for an arg-max: how high could it be; v - d for an
// arg-min), ei: the candidate e1 belongs to.  The reduction is settled when no candidate other than the winner has a far end
// beyond the winner's near end (arg_near); with d = 0 everywhere that is "no exact tie", which the index rule decides.
struct Best {
  double v;
  long long i;   // global flat index, -1 = none
  double d;
  double e1, e2;
  long long ei;
};
template <bool MAX>
__host__ __device__ __forceinline__ Best best_none() { return Best{0.0, -1, 0.0, MAX ? -kInfD : kInfD, MAX ? -kInfD : kInfD, -1}; }

template <bool MAX>
__host__ __device__ __forceinline__ bool better(const Best& a, const Best& b) {
  if (a.i < 0) return false;
  if (b.i < 0) return true;
  if (MAX ? (a.v > b.v) : (a.v < b.v)) return true;
  return a.v == b.v && a.i < b.i;   // ties -> lowest flat index
}
template <bool MAX>
__host__ __device__ __forceinline__ void ends_take(Best& B, double end, long long i) {
  if (MAX ? (end > B.e1) : (end < B.e1)) { B.e2 = B.e1; B.e1 = end; B.ei = i; }
  else if (MAX ? (end > B.e2) : (end < B.e2)) B.e2 = end;
}
// candidate i with value v +- d
template <bool MAX>
__host__ __device__ __forceinline__ void best_take(Best& B, double v, double d, long long i) {
  if (B.i < 0 || (MAX ? (v > B.v) : (v < B.v)) || (v == B.v && i < B.i)) { B.v = v; B.i = i; B.d = d; }
  ends_take<MAX>(B, MAX ? v + d : v - d, i);
}
template <bool MAX>
__host__ __device__ __forceinline__ Best best_merge(const Best& a, const Best& b) {
  Best r = a;
  if (better<MAX>(b, a)) { r.v = b.v; r.i = b.i; r.d = b.d; }
  ends_take<MAX>(r, b.e1, b.ei);           // (no-op for an empty b: its ends are the sentinels)
  if (MAX ? (b.e2 > r.e2) : (b.e2 < r.e2)) r.e2 = b.e2;
  return r;
}
template <bool MAX>
__host__ __device__ __forceinline__ bool arg_near(const Best& w) {
  if (w.i < 0) return false;
  const double other = w.ei == w.i ? w.e2 : w.e1;          // the most extreme far end among the OTHER candidates
  return MAX ? !(other < w.v - w.d) : !(other > w.v + w.d);
}

template <bool MAX>
__device__ __forceinline__ Best wave_best(Best x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Best y;
    y.v = __shfl_xor(x.v, o);
    y.i = __shfl_xor(x.i, o);
    y.d = __shfl_xor(x.d, o);
    y.e1 = __shfl_xor(x.e1, o);
    y.e2 = __shfl_xor(x.e2, o);
    y.ei = __shfl_xor(x.ei, o);
    x = best_merge<MAX>(x, y);
  }
  return x;
}

template <bool MAX>
__device__ __forceinline__ Best block_best(Best x) {
  __shared__ Best sh[16];
  x = wave_best<MAX>(x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = x;
  __syncthreads();
  if (wave == 0) {
    Best y = lane < nw ? sh[lane] : best_none<MAX>();
    y = wave_best<MAX>(y);
    x = y;
  }
  return x;   // valid in wave 0
}

// The same reduction for values whose band d is ONE number (var_0 +- dv: the arg-max reductions of a SafeOpt sweep): a thread
// only tracks the winner and the runner-up VALUE (in e2), half the fields to carry through the shuffles; block_best_uni puts
// the general form back together (e1 = the winner's own far end, e2 = the runner-up's).
template <bool MAX>
__device__ __forceinline__ void best_take_uni(Best& B, double v, long long i) {
  if (B.i < 0 || (MAX ? (v > B.v) : (v < B.v)) || (v == B.v && i < B.i)) {
    if (B.i >= 0) B.e2 = MAX ? (B.v > B.e2 ? B.v : B.e2) : (B.v < B.e2 ? B.v : B.e2);
    B.v = v;
    B.i = i;
  } else {
    B.e2 = MAX ? (v > B.e2 ? v : B.e2) : (v < B.e2 ? v : B.e2);
  }
}
template <bool MAX>
__device__ __forceinline__ Best wave_best_uni(Best x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Best y;
    y.v = __shfl_xor(x.v, o);
    y.i = __shfl_xor(x.i, o);
    y.e2 = __shfl_xor(x.e2, o);
    const bool yb = better<MAX>(y, x);
    const double lose = yb ? x.v : y.v;                      // the loser's value joins the runner-up contest (when it exists)
    const bool lose_on = yb ? x.i >= 0 : y.i >= 0;
    double e2 = MAX ? (y.e2 > x.e2 ? y.e2 : x.e2) : (y.e2 < x.e2 ? y.e2 : x.e2);
    if (lose_on) e2 = MAX ? (lose > e2 ? lose : e2) : (lose < e2 ? lose : e2);
    if (yb) { x.v = y.v; x.i = y.i; }
    x.e2 = e2;
  }
  return x;
}
template <bool MAX>
__device__ __forceinline__ Best block_best_uni(Best x, double d) {
  __shared__ Best shu[16];
  x = wave_best_uni<MAX>(x);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) shu[wave] = x;
  __syncthreads();
  if (wave == 0) {
    Best y = lane < nw ? shu[lane] : best_none<MAX>();
    y = wave_best_uni<MAX>(y);
    // back to the general form: far ends = value +- d
    y.d = d;
    y.e1 = y.i >= 0 ? (MAX ? y.v + d : y.v - d) : (MAX ? -kInfD : kInfD);
    y.ei = y.i;
    if (MAX ? y.e2 > -kInfD : y.e2 < kInfD) y.e2 = MAX ? y.e2 + d : y.e2 - d;
    x = y;
  }
  return x;   // valid in wave 0
}

__device__ __forceinline__ long long block_sum_ll(long long v) {
  __shared__ long long sh[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  long long r = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < nw; ++w) r += sh[w];
  return r;   // valid in thread 0
}

template <bool MAX>
__device__ __forceinline__ unsigned long long block_ext_u64(unsigned long long v) {
  __shared__ unsigned long long sh[16];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long y = __shfl_xor(v, o);
    v = MAX ? (y > v ? y : v) : (y < v ? y : v);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  if (threadIdx.x == 0)
    for (int w = 1; w < nw; ++w) v = MAX ? (sh[w] > v ? sh[w] : v) : (sh[w] < v ? sh[w] : v);
  return v;   // valid in thread 0
}

template <typename T>
__device__ __forceinline__ void lcb_ucb(T m, T v, T b, T& lcb, T& ucb) {
  const T sd = mul_rn(b, sqrt_rn(v));    // models/SafeOpt.py:37, 43: mean -/+ b*sqrt(var)
  lcb = sub_rn(m, sd);
  ucb = add_rn(m, sd);
}

// a workgroup's partial row of the classification: u* key, |S|, |U|, decisions inside the guard band, min-variance keys over S
// per output (~0: none), radius keys per constraint.  Stored field-major -- part[field * pcap + row], pcap = the row capacity
// of the buffer --, so that the one workgroup that merges the rows reads every field with coalesced loads (row-major rows of
// 160 bytes made that merge the longest job of the launch it shares: +14 us on config H).
constexpr int kClassifyRow = 4 + 2 * kMaxQ;
constexpr int kRowVmin = 4, kRowRmax = 4 + kMaxQ;

// ---- K3a: S / U masks, u* --------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_classify(const T* __restrict__ mean, const T* __restrict__ var, long long n,
                                                  int q, T b, uint8_t* __restrict__ S, uint8_t* __restrict__ U,
                                                  unsigned long long* __restrict__ part /* [kClassifyRow][pcap] */, int pcap,
                                                  const GuardBand* __restrict__ gb /* nullptr: the posterior is exact */) {
  __shared__ unsigned long long rmax_sh[kMaxQ];   // max over S of ucb_c: bounds the expander search radius
  if (threadIdx.x < kMaxQ) rmax_sh[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long umin = ~0ull;
  long long cS = 0, cU = 0, cB = 0;
  // guard band (fp64 approximating posteriors): sign tests that the band could move are counted, and the smallest variance over
  // S is kept per output (it bounds the band of sqrt(var) on S for the kernels that follow)
  LcbBand lband[kMaxQ];
  double vmin[kMaxQ];
  int cBi = 0;
#pragma unroll
  for (int c = 0; c < kMaxQ; ++c) {
    lband[c] = (gb && c < q) ? lcb_band((double)b * (double)b, gb->dm[c], gb->dv[c]) : LcbBand{0.0, 0.0};
    vmin[c] = kInfD;
  }
  // constraints first (S / U bits, ucb_c for the radius keys); the objective's mean / var are read for safe candidates
  // only -- S is a fifth of config B's grid, so the kernel streams (q - 1) / q of the posterior plus that fifth
  // fp64: the sign of every lcb_c without the square root (lcb_sign), and the exact ucb_c -- for the radius keys -- only of
  // safe candidates whose cheap upper bound beats the workgroup's running maximum
  constexpr bool kFast = std::is_same<T, double>::value;
  const double bb = (double)b * (double)b;
  auto constraints = [&](const T* mv /* [2 * q]: mean_c, var_c (entries of output 0 unused) */, T* ucbc) {
    bool s_ = true, u = true;
#pragma unroll
    for (int c = 1; c < kMaxQ; ++c) {
      if (c < q) {
        if constexpr (kFast) {
          const LcbSign sg = gb ? lcb_sign_gb((double)mv[2 * c], (double)mv[2 * c + 1], (double)b, bb, lband[c], cBi)
                                : lcb_sign((double)mv[2 * c], (double)mv[2 * c + 1], (double)b, bb);
          s_ = s_ && sg.ge;
          u = u && sg.le;
        } else {
          T lcb;
          lcb_ucb(mv[2 * c], mv[2 * c + 1], b, lcb, ucbc[c]);   // one sqrt per (candidate, constraint)
          s_ = s_ && (lcb >= T(0));
          u = u && (lcb <= T(0));
        }
      }
    }
    cS += s_;
    cU += u;
    return (unsigned)(s_ ? 1u : 0u) | (unsigned)(u ? 2u : 0u);
  };
  auto objective = [&](T m0, T v0, const T* mv, const T* ucbc) {       // a safe candidate: u* key and radius keys
    if (gb) {
      vmin[0] = fmin(vmin[0], (double)v0);
#pragma unroll
      for (int c = 1; c < kMaxQ; ++c)
        if (c < q) vmin[c] = fmin(vmin[c], (double)mv[2 * c + 1]);
    }
    bool need = true;
    if constexpr (kFast) need = !(ord_key(ucb_lower((double)m0, (double)v0, (double)b)) >= umin);
    if (need) {
      T lcb, ucb;
      lcb_ucb(m0, v0, b, lcb, ucb);
      const unsigned long long k = ord_key((double)ucb);
      umin = k < umin ? k : umin;
    }
#pragma unroll
    for (int c = 1; c < kMaxQ; ++c) {
      if (c < q) {
        if constexpr (kFast) {
          if (!(ord_key(ucb_upper((double)mv[2 * c], (double)mv[2 * c + 1], (double)b)) <= rmax_sh[c])) {
            T lcb, ucb;
            lcb_ucb(mv[2 * c], mv[2 * c + 1], b, lcb, ucb);
            const unsigned long long kc = ord_key((double)ucb);
            if (kc > rmax_sh[c]) atomicMax(&rmax_sh[c], kc);
          }
        } else {
          const unsigned long long kc = ord_key((double)ucbc[c]);
          if (kc > rmax_sh[c]) atomicMax(&rmax_sh[c], kc);
        }
      }
    }
  };
  // two consecutive candidates per thread: the mean / var streams are read with 16-byte (fp64) / 8-byte (fp32) loads
  typedef T T2 __attribute__((ext_vector_type(2)));
  const long long npair = n >> 1;
  const bool aligned = (n & 1) == 0;               // output c starts at c * n elements: pairs stay aligned only for even n
  if (aligned) {
    // a contiguous chunk of pairs per workgroup (the four streams of a workgroup then walk four DRAM pages, not 4 x the
    // number of workgroups interleaved)
    const long long chunk = ((npair + gridDim.x - 1) / gridDim.x + blockDim.x - 1) / blockDim.x * blockDim.x;
    const long long pend = (blockIdx.x + 1) * chunk < npair ? (blockIdx.x + 1) * chunk : npair;
    for (long long pi = blockIdx.x * chunk + threadIdx.x; pi < pend; pi += blockDim.x) {
      T mv0[2 * kMaxQ], mv1[2 * kMaxQ], uc0[kMaxQ], uc1[kMaxQ];
#pragma unroll
      for (int c = 1; c < kMaxQ; ++c) {
        if (c < q) {
          const T2 m2 = *reinterpret_cast<const T2*>(mean + (size_t)c * n + 2 * pi);
          const T2 v2 = *reinterpret_cast<const T2*>(var + (size_t)c * n + 2 * pi);
          mv0[2 * c] = m2[0]; mv0[2 * c + 1] = v2[0];
          mv1[2 * c] = m2[1]; mv1[2 * c + 1] = v2[1];
        }
      }
      const unsigned r0 = constraints(mv0, uc0), r1 = constraints(mv1, uc1);
      if ((r0 | r1) & 1u) {
        const T2 m2 = *reinterpret_cast<const T2*>(mean + 2 * pi);
        const T2 v2 = *reinterpret_cast<const T2*>(var + 2 * pi);
        if (r0 & 1u) objective(m2[0], v2[0], mv0, uc0);
        if (r1 & 1u) objective(m2[1], v2[1], mv1, uc1);
      }
      *reinterpret_cast<unsigned short*>(S + 2 * pi) = (unsigned short)((r0 & 1u) | ((r1 & 1u) << 8));
      *reinterpret_cast<unsigned short*>(U + 2 * pi) = (unsigned short)(((r0 >> 1) & 1u) | (((r1 >> 1) & 1u) << 8));
    }
  } else {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
      T mv[2 * kMaxQ], uc[kMaxQ];
#pragma unroll
      for (int c = 1; c < kMaxQ; ++c)
        if (c < q) { mv[2 * c] = mean[(size_t)c * n + g]; mv[2 * c + 1] = var[(size_t)c * n + g]; }
      const unsigned r = constraints(mv, uc);
      if (r & 1u) objective(mean[g], var[g], mv, uc);
      S[g] = (uint8_t)(r & 1u);
      U[g] = (uint8_t)((r >> 1) & 1u);
    }
  }
  umin = block_ext_u64<false>(umin);
  cS = block_sum_ll(cS);
  cU = block_sum_ll(cU);
  unsigned long long vk[kMaxQ];
#pragma unroll
  for (int c = 0; c < kMaxQ; ++c) vk[c] = ~0ull;
  if (gb) {
    cB = block_sum_ll((long long)cBi);
#pragma unroll
    for (int c = 0; c < kMaxQ; ++c)
      if (c < q) vk[c] = block_ext_u64<false>(vmin[c] < kInfD ? ord_key(vmin[c]) : ~0ull);
  }
  __syncthreads();
  // per-workgroup partials, merged by k_classify_final (atomics of every workgroup on one cache line serialise in L2:
  // ~10 ns each, which was most of this kernel's time)
  unsigned long long* row = part + blockIdx.x;
  if (threadIdx.x == 0) {
    row[0] = umin;
    row[(size_t)1 * pcap] = (unsigned long long)cS;
    row[(size_t)2 * pcap] = (unsigned long long)cU;
    row[(size_t)3 * pcap] = (unsigned long long)cB;
#pragma unroll
    for (int c = 0; c < kMaxQ; ++c) row[(size_t)(kRowVmin + c) * pcap] = vk[c];
  }
  if (threadIdx.x < kMaxQ) row[(size_t)(kRowRmax + threadIdx.x) * pcap] = rmax_sh[threadIdx.x];
}

// start of a sweep's scalar block: cleared, then u*, |S|, |U| and the radius keys merged from k_classify's partials
struct FinalJob {
  bool pending = false;
  const unsigned long long* part = nullptr;   // k_classify's rows (field-major)
  int nparts = 0, pcap = 0, q = 0;
  SweepScalars* sc = nullptr;
  const double* Lpart = nullptr;              // K1b's Lipschitz partials still to be merged (nullptr: Lmax is final)
  int per_out = 0;
  unsigned long long* Lmax = nullptr;
  SweepScalars* sc_copy = nullptr;            // the second lane's block: a snapshot of the merged scalars (nullptr: one lane)
  const GuardBand* gb = nullptr;              // band of an approximating posterior (nullptr: exact)
  double b = 0.0;                             // the sweep's confidence multiplier (enters the band of the bounds)
};
__device__ __forceinline__ void classify_final_body(const unsigned long long* __restrict__ part, int nparts, int pcap, int q, SweepScalars* sc,
                                                    const double* __restrict__ Lpart, int per_out, unsigned long long* Lmax,
                                                    SweepScalars* sc_copy, const GuardBand* gb, double b) {
  __shared__ double lsh[4];
  if (Lpart)
    for (int o = 0; o < q; ++o) lmax_reduce_body(o, lsh, Lpart, per_out, Lmax);
  unsigned long long* w = reinterpret_cast<unsigned long long*>(sc);       // (the struct is a multiple of 8 bytes)
  for (unsigned i = threadIdx.x; i < sizeof(SweepScalars) / 8; i += blockDim.x) w[i] = 0ull;
  __syncthreads();
  unsigned long long umin = ~0ull, rmax[kMaxQ], vmin[kMaxQ];
  long long cS = 0, cU = 0, cB = 0;
#pragma unroll
  for (int c = 0; c < kMaxQ; ++c) { rmax[c] = 0ull; vmin[c] = ~0ull; }
  // (one workgroup walks all rows: four rows' loads in flight per thread and only the q columns that carry anything; the
  // fields are stored field-major, so a load instruction of the workgroup reads 2 KB in a row)
  for (int i0 = threadIdx.x; i0 < nparts; i0 += 4 * blockDim.x) {
    unsigned long long v0[4], v1[4], v2[4], vr[4][kMaxQ];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + j * blockDim.x;
      const bool on = i < nparts;
      const unsigned long long* row = part + (on ? i : i0);
      v0[j] = on ? row[0] : ~0ull;
      v1[j] = on ? row[(size_t)1 * pcap] : 0ull;
      v2[j] = on ? row[(size_t)2 * pcap] : 0ull;
#pragma unroll
      for (int c = 1; c < kMaxQ; ++c) vr[j][c] = (on && c < q) ? row[(size_t)(kRowRmax + c) * pcap] : 0ull;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      umin = v0[j] < umin ? v0[j] : umin;
      cS += (long long)v1[j];
      cU += (long long)v2[j];
#pragma unroll
      for (int c = 1; c < kMaxQ; ++c) rmax[c] = vr[j][c] > rmax[c] ? vr[j][c] : rmax[c];
    }
    if (gb) {                                   // (the band's columns: read only when a band is in force)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = i0 + j * blockDim.x;
        if (i >= nparts) continue;
        const unsigned long long* row = part + i;
        cB += (long long)row[(size_t)3 * pcap];
#pragma unroll
        for (int c = 0; c < kMaxQ; ++c)
          if (c < q) { const unsigned long long k = row[(size_t)(kRowVmin + c) * pcap]; vmin[c] = k < vmin[c] ? k : vmin[c]; }
      }
    }
  }
  umin = block_ext_u64<false>(umin);
  cS = block_sum_ll(cS);
  cU = block_sum_ll(cU);
#pragma unroll
  for (int c = 1; c < kMaxQ; ++c)
    if (c < q) rmax[c] = block_ext_u64<true>(rmax[c]);
  if (gb) {
    cB = block_sum_ll(cB);
#pragma unroll
    for (int c = 0; c < kMaxQ; ++c)
      if (c < q) vmin[c] = block_ext_u64<false>(vmin[c]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    sc->ustar_key = umin;
    sc->count_S = cS;
    sc->count_U = cU;
#pragma unroll
    for (int c = 1; c < kMaxQ; ++c) sc->rmax_key[c] = c < q ? rmax[c] : 0ull;
    if (gb) {
      // the band of every bound on S from the smallest variance over S (the band of a square root grows as its argument shrinks)
      sc->n_guard = cB;
      sc->n_guard_cls = cB;
#pragma unroll
      for (int c = 0; c < kMaxQ; ++c) {
        if (c < q) {
          sc->vmin_key[c] = vmin[c];
          // (no safe candidate on this rank: nothing here to bound -- a variance of "zero" would put sqrt(dv) ~ 1e-6 into the
          // max over the ranks: 119 members of M inside that band on config H at four ranks, every sweep a second pass)
          const bool any = vmin[c] != ~0ull;
          const double vm = any ? fmax(0.0, ord_val(vmin[c])) : 0.0;
          sc->gb_du[c] = gb->dm[c] + (any ? b * gb_dsqrt(vm, gb->dv[c]) : 0.0);
          sc->gb_rl[c] = gb->rl[c];
        }
      }
    }
  }
  if (threadIdx.x < kArgSlots) sc->arg_idx[threadIdx.x] = -1;
  if (sc_copy) {
    // The second lane's block is written HERE, not copied later on the lane's own stream: the main lane goes on to count
    // its recheck / scan candidates in `sc`, and a later copy would pick those counters up half-way.
    __syncthreads();
    const unsigned long long* from = reinterpret_cast<const unsigned long long*>(sc);
    unsigned long long* to = reinterpret_cast<unsigned long long*>(sc_copy);
    for (unsigned i = threadIdx.x; i < sizeof(SweepScalars) / 8; i += blockDim.x) to[i] = from[i];
  }
}
__global__ __launch_bounds__(256) void k_classify_final(const unsigned long long* __restrict__ part, int nparts, int pcap, int q,
                                                        SweepScalars* sc, const double* __restrict__ Lpart, int per_out,
                                                        unsigned long long* Lmax, SweepScalars* sc_copy, const GuardBand* gb, double b) {
  classify_final_body(part, nparts, pcap, q, sc, Lpart, per_out, Lmax, sc_copy, gb, b);
}

// Objective pass of a classification whose S / U bytes came out of the posterior kernel (K1b, one constraint): u* = min over
// S of ucb_0, one partial row per workgroup behind the posterior's rows (mask-driven loop as k_minimizer)
__device__ __forceinline__ bool tile_byte(unsigned long long w, int k, int lane);
// (r05) several constraints: the posterior kernel's workgroups wrote one S / U byte plane per constraint (a workgroup sees one output);
// here the planes are AND-ed into the S / U masks on the way (models/SafeOpt.py:57-59: safe = every constraint's lcb >= 0), and the
// populations counted
struct PlaneAnd {
  const uint8_t* Sp;      // [np][stride] planes, nullptr: S below is the mask itself
  const uint8_t* Up;
  long long stride;
  int np;
  uint8_t* Uout;
};
template <typename T>
__device__ __forceinline__ unsigned long long ustar_partial_body(int bid, int nwg, const T* __restrict__ mean0, const T* __restrict__ var0,
                                                                 long long n, T b, const uint8_t* __restrict__ S, double& vmin0,
                                                                 const PlaneAnd pa = PlaneAnd{nullptr, nullptr, 0, 0, nullptr}, long long* cS = nullptr,
                                                                 long long* cU = nullptr) {
  unsigned long long umin = ~0ull;
  vmin0 = kInfD;                                    // smallest var_0 over S seen by this thread (guard band, see k_classify)
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)bid * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)nwg * blockDim.x) >> 6;
  const bool planes = pa.Sp != nullptr;
  uint8_t* const Sw_ = const_cast<uint8_t*>(S);        // (planes: S is written here)
  const bool al8 = (((uintptr_t)S) & 7) == 0 && (!planes || (((((uintptr_t)pa.Sp) | ((uintptr_t)pa.Up) | ((uintptr_t)pa.Uout)) & 7) == 0 && (pa.stride & 7) == 0));
  const long long ntiles = al8 ? n / 512 : 0;
  long long nS = 0, nU = 0;
  auto take = [&](T m, T v) {
    // (fp64: the exact bound only when the cheap lower bound could beat the running minimum)
    if (std::is_same<T, double>::value && ord_key(ucb_lower((double)m, (double)v, (double)b)) >= umin) return;
    T lcb, ucb;
    lcb_ucb(m, v, b, lcb, ucb);
    const unsigned long long k = ord_key((double)ucb);
    umin = k < umin ? k : umin;
  };
  // (the mask words of four of this wave's tiles are requested together: the tile loop is a chain of dependent loads -- mask word,
  // then the gathered posterior -- and most tiles end at the first link)
  for (long long t0 = wave; t0 < ntiles; t0 += 4 * nwaves) {
    unsigned long long w4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long t = t0 + j * nwaves;
      if (!planes) {
        w4[j] = t < ntiles ? ((const unsigned long long*)(S + t * 512))[lane] : 0ull;
      } else {
        unsigned long long ws = 0ull, wu = 0ull;
        if (t < ntiles) {
          ws = ((const unsigned long long*)(pa.Sp + t * 512))[lane];
          wu = ((const unsigned long long*)(pa.Up + t * 512))[lane];
          for (int c = 1; c < pa.np; ++c) {
            ws &= ((const unsigned long long*)(pa.Sp + (size_t)c * pa.stride + t * 512))[lane];
            wu &= ((const unsigned long long*)(pa.Up + (size_t)c * pa.stride + t * 512))[lane];
          }
          ((unsigned long long*)(Sw_ + t * 512))[lane] = ws;
          ((unsigned long long*)(pa.Uout + t * 512))[lane] = wu;
          nS += __popcll(ws);                     // (bytes are 0 / 1)
          nU += __popcll(wu);
        }
        w4[j] = ws;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
    const long long t = t0 + j * nwaves;
    const long long base = t * 512;
    const unsigned long long w = w4[j];
    if (__ballot(w != 0ull) == 0ull) continue;
    T mu[8], va[8];
    bool set[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      set[k] = tile_byte(w, k, lane);
      const long long g = base + k * 64 + lane;
      mu[k] = set[k] ? mean0[g] : (T)0;
      va[k] = set[k] ? var0[g] : (T)0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (set[k]) { vmin0 = fmin(vmin0, (double)va[k]); take(mu[k], va[k]); }
    }
  }
  for (long long g = ntiles * 512 + (long long)bid * blockDim.x + threadIdx.x; g < n; g += (long long)nwg * blockDim.x) {
    bool sg;
    if (planes) {
      uint8_t ss = 1, uu = 1;
      for (int c = 0; c < pa.np; ++c) { ss &= pa.Sp[(size_t)c * pa.stride + g]; uu &= pa.Up[(size_t)c * pa.stride + g]; }
      Sw_[g] = ss;
      pa.Uout[g] = uu;
      nS += ss;
      nU += uu;
      sg = ss != 0;
    } else {
      sg = S[g] != 0;
    }
    if (sg) { const T v = var0[g]; vmin0 = fmin(vmin0, (double)v); take(mean0[g], v); }
  }
  if (cS) { *cS = nS; *cU = nU; }
  return block_ext_u64<false>(umin);   // valid in thread 0
}
template <typename T>
__global__ __launch_bounds__(256) void k_classify_obj(const T* __restrict__ mean0, const T* __restrict__ var0, long long n, T b,
                                                      const uint8_t* __restrict__ S, unsigned long long* __restrict__ part /* first row of this pass */,
                                                      int pcap) {
  double vmin0;
  const unsigned long long umin = ustar_partial_body<T>((int)blockIdx.x, (int)gridDim.x, mean0, var0, n, b, S, vmin0);
  const unsigned long long vk = block_ext_u64<false>(vmin0 < kInfD ? ord_key(vmin0) : ~0ull);
  __shared__ unsigned long long keys[2];
  if (threadIdx.x == 0) { keys[0] = umin; keys[1] = vk; }
  __syncthreads();
  unsigned long long* row = part + blockIdx.x;
  if (threadIdx.x < kClassifyRow) {
    const int t = threadIdx.x;
    row[(size_t)t * pcap] = t == 0 ? keys[0] : (t == kRowVmin ? keys[1] : ((t > kRowVmin && t < kRowRmax) ? ~0ull : 0ull));
  }
}

// the same behind a posterior launch that wrote one byte plane per constraint (r05): S / U = the planes AND-ed, |S|, |U|, u*
template <typename T>
__global__ __launch_bounds__(256) void k_classify_and(const T* __restrict__ mean0, const T* __restrict__ var0, long long n, T b, const PlaneAnd pa,
                                                      uint8_t* __restrict__ S, unsigned long long* __restrict__ part, int pcap) {
  double vmin0;
  long long cS = 0, cU = 0;
  const unsigned long long umin = ustar_partial_body<T>((int)blockIdx.x, (int)gridDim.x, mean0, var0, n, b, S, vmin0, pa, &cS, &cU);
  const unsigned long long vk = block_ext_u64<false>(vmin0 < kInfD ? ord_key(vmin0) : ~0ull);
  cS = block_sum_ll(cS);
  cU = block_sum_ll(cU);
  __shared__ unsigned long long keys[4];
  if (threadIdx.x == 0) { keys[0] = umin; keys[1] = vk; keys[2] = (unsigned long long)cS; keys[3] = (unsigned long long)cU; }
  __syncthreads();
  unsigned long long* row = part + blockIdx.x;
  if (threadIdx.x < kClassifyRow) {
    const int t = threadIdx.x;
    row[(size_t)t * pcap] = t == 0 ? keys[0] : (t == 1 ? keys[2] : (t == 2 ? keys[3] : (t == kRowVmin ? keys[1] : ((t > kRowVmin && t < kRowRmax) ? ~0ull : 0ull))));
  }
}

// Mask-driven loops of K3b / K5.  A wave takes tiles of 512 consecutive candidates: every lane reads eight mask bytes
// as one word, tiles without a set byte cost nothing more, and for the others the value loads (eight per lane, coalesced
// across the wave, predicated on the candidate's own byte fetched by a shuffle) are all issued before the first
// comparison.  The remainder (n mod 512, or everything when the mask is not 8-byte aligned) runs one candidate per lane.
__device__ __forceinline__ bool tile_byte(unsigned long long w, int k, int lane) {
  const unsigned long long wk = __shfl(w, k * 8 + (lane >> 3));
  return ((wk >> (8 * (lane & 7))) & 0xffull) != 0ull;
}

// ---- K3b + K5: M mask and arg-max of var_0 over M -----------------------------------------------------
template <typename T>
__device__ __forceinline__ void minimizer_body(int bid, int nwg, const T* __restrict__ mean0, const T* __restrict__ var0, long long n,
                                               long long first, T b, const uint8_t* __restrict__ S, uint8_t* __restrict__ M,
                                               unsigned long long ustar_key, Best* partial, const SweepScalars* sc,
                                               const GuardBand* __restrict__ gb) {
  const T ustar = (T)ord_val(ustar_key);
  Best best = best_none<true>();
  long long cM = 0, cB = 0;
  // guard band: u* and every lcb_0 on S are known to +- gb_du[0], so |lcb_0 - u*| <= 2 gb_du[0] leaves "lcb_0 <= u*" open;
  // var_0 is known to +- dv[0] (the arg-max's band)
  const double dM = gb ? 2.0 * sc->gb_du[0] * (1.0 + 0x1p-40) : -1.0, dv0 = gb ? gb->dv[0] : 0.0;
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)bid * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)nwg * blockDim.x) >> 6;
  const long long ntiles = (((uintptr_t)S) & 7) == 0 ? n / 512 : 0;
  for (long long t = wave; t < ntiles; t += nwaves) {
    const long long base = t * 512;
    const unsigned long long w = ((const unsigned long long*)(S + base))[lane];
    if (__ballot(w != 0ull) == 0ull) {
#pragma unroll
      for (int k = 0; k < 8; ++k) M[base + k * 64 + lane] = 0;
      continue;
    }
    T mu[8], va[8];
    bool set[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      set[k] = tile_byte(w, k, lane);
      const long long g = base + k * 64 + lane;
      mu[k] = set[k] ? mean0[g] : (T)0;
      va[k] = set[k] ? var0[g] : (T)0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const long long g = base + k * 64 + lane;
      bool m = false;
      if (set[k]) {
        T lcb, ucb;
        lcb_ucb(mu[k], va[k], b, lcb, ucb);
        m = lcb <= ustar;                     // models/SafeOpt.py:62
        const double gap = (double)lcb - (double)ustar;
        cB += (gap < 0 ? -gap : gap) <= dM;
      }
      M[g] = m;
      if (m) {
        ++cM;
        best_take_uni<true>(best, (double)va[k], first + g);
      }
    }
  }
  for (long long g = ntiles * 512 + (long long)bid * blockDim.x + threadIdx.x; g < n; g += (long long)nwg * blockDim.x) {
    bool m = false;
    if (S[g]) {
      T lcb, ucb;
      lcb_ucb(mean0[g], var0[g], b, lcb, ucb);
      m = lcb <= ustar;                       // models/SafeOpt.py:62
      const double gap = (double)lcb - (double)ustar;
      cB += (gap < 0 ? -gap : gap) <= dM;
    }
    M[g] = m;
    if (m) {
      ++cM;
      best_take_uni<true>(best, (double)var0[g], first + g);
    }
  }
  best = block_best_uni<true>(best, dv0);
  cM = block_sum_ll(cM);
  cB = block_sum_ll(cB);
  if (threadIdx.x == 0) {
    partial[bid] = best;
    ((long long*)(partial + nwg))[bid] = cM;   // summed by the finals (an atomic per workgroup on one counter serialises)
    ((long long*)(partial + nwg))[nwg + bid] = cB;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_minimizer(const T* __restrict__ mean0, const T* __restrict__ var0, long long n,
                                                   long long first, T b, const uint8_t* __restrict__ S,
                                                   uint8_t* __restrict__ M, SweepScalars* sc, Best* partial, const GuardBand* gb) {
  minimizer_body<T>((int)blockIdx.x, (int)gridDim.x, mean0, var0, n, first, b, S, M, sc->ustar_key, partial, sc, gb);
}

// value sources of the masked arg-reductions: an array, or the value computed for the candidates whose mask byte is set
// only (no pass over all candidates to fill an array first)
// (`d`: the band of the value under the guard band of an approximating posterior -- bind() loads the band's scalars once)
template <typename T>
struct ValArray {          // var_0
  static constexpr bool kUniform = true;   // one band for every value (uni())
  const T* p;
  double dv;
  __device__ __forceinline__ double uni() const { return dv; }
  __device__ __forceinline__ void bind(const GuardBand* gb) { dv = gb ? gb->dv[0] : 0.0; }
  __device__ __forceinline__ T operator()(long long g, double& d) const { d = dv; return p[g]; }
};
template <typename T>
struct ValLcb {            // lcb_0 = mean_0 - b sqrt(var_0), models/GoOSE.py:72, models/GP_TR.py:45
  static constexpr bool kUniform = false;  // the band of a bound depends on the candidate's variance
  __device__ __forceinline__ double uni() const { return 0.0; }
  const T* m;
  const T* v;
  T b;
  double dm, dv;
  __device__ __forceinline__ void bind(const GuardBand* gb) { dm = gb ? gb->dm[0] : 0.0; dv = gb ? gb->dv[0] : 0.0; }
  __device__ __forceinline__ T operator()(long long g, double& d) const {
    T lcb, ucb;
    const T vv = v[g];
    lcb_ucb(m[g], vv, b, lcb, ucb);
    d = dv > 0.0 ? dm + (double)b * gb_dsqrt((double)vv, dv) : dm;
    return lcb;
  }
};
template <typename T, int D>
struct ValDist {           // Euclidean distance to the target, as scipy.spatial.distance.cdist computes it (models/GoOSE.py:117)
  static constexpr bool kUniform = true;
  __device__ __forceinline__ double uni() const { return 0.0; }
  CandSpec cs;
  const double* target;
  __device__ __forceinline__ void bind(const GuardBand*) {}
  __device__ __forceinline__ T operator()(long long g, double& d) const {
    d = 0.0;                               // (geometry: exact)
    double x[D];
    cand_coords<D>(cs, g, x);
    double ss = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      if (a < cs.d) {
        const double df = x[a] - target[a];
        ss += df * df;
      }
    }
    return (T)sqrt(ss);
  }
};

// generic masked arg-max / arg-min of a value source, plus the mask population
template <typename T, bool MAX, typename V>
__device__ __forceinline__ void arg_masked_body(int bid, int nwg, const V& val_in, const uint8_t* __restrict__ mask, long long n,
                                                long long first, Best* partial, const GuardBand* __restrict__ gb) {
  V val = val_in;
  val.bind(gb);
  Best best = best_none<MAX>();
  long long cnt = 0;
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)bid * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)nwg * blockDim.x) >> 6;
  const long long ntiles = (((uintptr_t)mask) & 7) == 0 ? n / 512 : 0;
  for (long long t = wave; t < ntiles; t += nwaves) {
    const long long base = t * 512;
    const unsigned long long w = ((const unsigned long long*)(mask + base))[lane];
    if (__ballot(w != 0ull) == 0ull) continue;
    T v[8];
    double dd[8];
    bool set[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      set[k] = tile_byte(w, k, lane);
      dd[k] = 0.0;
      v[k] = set[k] ? val(base + k * 64 + lane, dd[k]) : (T)0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (set[k]) {
        ++cnt;
        if constexpr (V::kUniform) best_take_uni<MAX>(best, (double)v[k], first + base + k * 64 + lane);
        else best_take<MAX>(best, (double)v[k], dd[k], first + base + k * 64 + lane);
      }
    }
  }
  for (long long g = ntiles * 512 + (long long)bid * blockDim.x + threadIdx.x; g < n; g += (long long)nwg * blockDim.x) {
    if (mask[g]) {
      ++cnt;
      double d = 0.0;
      const T v = val(g, d);
      if constexpr (V::kUniform) best_take_uni<MAX>(best, (double)v, first + g);
      else best_take<MAX>(best, (double)v, d, first + g);
    }
  }
  if constexpr (V::kUniform) best = block_best_uni<MAX>(best, val.uni());
  else best = block_best<MAX>(best);
  cnt = block_sum_ll(cnt);
  if (threadIdx.x == 0) {
    partial[bid] = best;
    ((long long*)(partial + nwg))[bid] = cnt;
    ((long long*)(partial + nwg))[nwg + bid] = 0;
  }
}
template <typename T, bool MAX, typename V>
__global__ __launch_bounds__(256) void k_arg_masked(const V val, const uint8_t* __restrict__ mask, long long n,
                                                    long long first, Best* partial, const GuardBand* gb) {
  arg_masked_body<T, MAX, V>((int)blockIdx.x, (int)gridDim.x, val, mask, n, first, partial, gb);
}
// the same for several masks in one launch: blockIdx.y = s selects slot slot0 + s -- mask0 for slot 0, masks + (slot - 1) n
// otherwise -- and region slot of the partials (stride pstride bytes); the reductions are independent of each other
template <typename T, bool MAX, typename V>
__global__ __launch_bounds__(256) void k_arg_masked_multi(const V val, const uint8_t* __restrict__ mask0, const uint8_t* __restrict__ masks,
                                                          long long n, long long first, unsigned char* pbase, size_t pstride, int slot0,
                                                          const GuardBand* gb) {
  const int slot = slot0 + (int)blockIdx.y;
  const uint8_t* mask = slot == 0 ? mask0 : masks + (size_t)(slot - 1) * n;
  arg_masked_body<T, MAX, V>((int)blockIdx.x, (int)gridDim.x, val, mask, n, first, reinterpret_cast<Best*>(pbase + pstride * (size_t)slot), gb);
}

// A region of partials: Best[nparts], then the workgroups' mask populations long long[nparts], then their guard-band counts
// long long[nparts] (members of M inside the band of lcb_0 <= u*; zero for the plain arg-reductions).
__host__ __device__ __forceinline__ size_t partial_stride(int nparts) { return (sizeof(Best) + 2 * sizeof(long long)) * (size_t)nparts; }
// merge of one region by a workgroup; the results are valid in thread 0
template <bool MAX>
__device__ __forceinline__ void merge_region(const Best* __restrict__ partial, int nparts, Best& best, long long& cnt, long long& nb) {
  best = best_none<MAX>();
  cnt = 0;
  nb = 0;
  const long long* pc = (const long long*)(partial + nparts);
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
    best = best_merge<MAX>(best, partial[i]);
    cnt += pc[i];
    nb += pc[nparts + i];
  }
  best = block_best<MAX>(best);
  cnt = block_sum_ll(cnt);
  nb = block_sum_ll(nb);
}
// thread 0 of a merging workgroup: the slot's results, and what the guard band leaves open in it (gb_on: a band is in force)
template <bool MAX>
__device__ __forceinline__ void store_slot(SweepScalars* sc, int slot, const Best& best, long long nb, bool gb_on) {
  sc->arg_val[slot] = best.v;
  sc->arg_idx[slot] = best.i;
  sc->arg_d[slot] = best.d;
  sc->arg_e1[slot] = best.e1;
  sc->arg_e2[slot] = best.e2;
  sc->arg_ei[slot] = best.ei;
  sc->guard_slot[slot] = gb_on ? nb + (arg_near<MAX>(best) ? 1 : 0) : 0;
  if (slot == 0) sc->guard_nb0 = gb_on ? nb : 0;
}
template <bool MAX>
__global__ __launch_bounds__(256) void k_arg_final(const Best* __restrict__ partial, int nparts, SweepScalars* sc, int slot,
                                                   long long* count, int gb_on) {
  Best best;
  long long cnt, nb;
  merge_region<MAX>(partial, nparts, best, cnt, nb);
  if (threadIdx.x == 0) {
    store_slot<MAX>(sc, slot, best, nb, gb_on != 0);
    if (count) *count += cnt;
  }
}

// The final reductions of a sweep in one launch: workgroup s merges region s of the partials (region stride `stride`
// bytes).  SafeOpt (MAX): s = 0 minimiser -> slot 0 and |M|, s = c >= 1 expanders of constraint c -> slot c and |G_c|.
// GoOSE (!MAX): s = 0 arg-min of lcb_0 over S_t -> slot 0 (no count), s = c >= 1 over O_c -> slot c and |O_c|.
template <bool MAX>
__global__ __launch_bounds__(256) void k_sweep_finals(const unsigned char* __restrict__ regions, size_t stride, int nparts,
                                                      SweepScalars* sc, const SweepScalars* lane1 /* nullptr, or the second lane's block */,
                                                      unsigned char* mirror /* nullptr, or the host's pinned landing area */,
                                                      const unsigned long long* Lkeys, int gb_on) {
  if (lane1 && blockIdx.x == 0 && threadIdx.x == 0) {
    sc->n_amb_total += lane1->n_amb_total;
    sc->n_guard += lane1->n_guard - lane1->n_guard_cls;      // (the lane's block started as a snapshot: its own additions only)
  }
  // `mirror`: the results go straight to the pinned host block the read-back would have filled (SweepScalars at 0, the
  // Lipschitz keys at 3072) -- every workgroup its own slot, workgroup 0 the fields earlier kernels finished --, and the
  // sweep's end event rides on this launch: no copy kernel and no barrier packet behind the last kernel.
  SweepScalars* hm = reinterpret_cast<SweepScalars*>(mirror);
  if (mirror && blockIdx.x == 0) {
    if (threadIdx.x == 0) {
      hm->ustar_key = sc->ustar_key;
      hm->count_S = sc->count_S;
      hm->count_U = sc->count_U;
      hm->n_amb = sc->n_amb;
      hm->n_amb_total = sc->n_amb_total;
      hm->n_scan = sc->n_scan;
      hm->n_guard = sc->n_guard;
    }
    if (threadIdx.x < kMaxQ) {
      hm->rmax_key[threadIdx.x] = sc->rmax_key[threadIdx.x];
      reinterpret_cast<unsigned long long*>(mirror + 3072)[threadIdx.x] = Lkeys[threadIdx.x];
    }
    if (threadIdx.x >= 64 && threadIdx.x < 64 + kArgSlots && (int)threadIdx.x - 64 >= (int)gridDim.x) hm->guard_slot[threadIdx.x - 64] = 0;
  }
  const int slot = blockIdx.x;
  Best best;
  long long cnt, nb;
  merge_region<MAX>(reinterpret_cast<const Best*>(regions + (size_t)slot * stride), nparts, best, cnt, nb);
  if (threadIdx.x == 0) {
    store_slot<MAX>(sc, slot, best, nb, gb_on != 0);
    if (slot == 0) { if (MAX) sc->count_M += cnt; }
    else sc->count_set[slot - 1] += cnt;
    if (mirror) {
      hm->arg_val[slot] = best.v;
      hm->arg_idx[slot] = best.i;
      hm->arg_d[slot] = best.d;
      hm->guard_slot[slot] = sc->guard_slot[slot];
      if (slot == 0) hm->count_M = sc->count_M;
      else hm->count_set[slot - 1] = sc->count_set[slot - 1];
    }
  }
}

// (a second merge of the finals after a late recheck: the counts are accumulated with +=, so they start over)
__global__ void k_sweep_clear_slot(SweepScalars* sc, int nslots) {
  sc->count_M = 0;
  for (int s = 0; s < nslots; ++s) sc->count_set[s] = 0;
}

#include "sets_expander.inc.hpp"

// The middle of the set phase of a 2-D grid sweep in one launch: three jobs that need the classification's scalars (u*,
// the radius keys) and the axis-0 passes but not each other -- the last-axis scan of the coarse transform (first in the
// launch: its threads walk the longest chains of dependent loads), the M mask with its arg-max partials, the block minima
// of the fine transform.  One after the other they took 11 + 10 + 7 us on config B, none of them filling the GPU.
template <typename T>
struct MidJobs {
  const SweepScalars* sc;
  const unsigned long long* Lkeys;
  int ns;                                  // coarse scan: workgroups, in / out, cells, stride, count, step, constraint, L index
  const double* dc_in;
  double* dc_out;
  long long nc, cstride;
  int ccnt;
  double hc;
  int cidx, lidx;
  double cap_extra;
  int nb;                                  // minimiser (0: not in this launch)
  const T* mean0;
  const T* var0;
  long long n, first;
  T b;
  const uint8_t* S;
  uint8_t* M;
  Best* partial;
  const GuardBand* gb;
  int nm;                                  // block minima (0: none); din: the axis-0 image (doubles, or step counts with h0)
  double h0;
  const double* din;
  long long stride;
  int cnt, blk;
  double* bmin;
};
template <typename T, bool U16>
__global__ __launch_bounds__(256) void k_set_mid(const MidJobs<T> j) {
  __shared__ double part[4][64];
  const int bid = (int)blockIdx.x;
  if (bid < j.ns)
    edt_scan_body(bid, j.ns, j.dc_in, j.dc_out, j.nc, j.cstride, j.ccnt, j.hc, j.sc, j.cidx, j.Lkeys, j.lidx, 0, j.cap_extra);
  else if (bid < j.ns + j.nb)
    minimizer_body<T>(bid - j.ns, j.nb, j.mean0, j.var0, j.n, j.first, j.b, j.S, j.M, j.sc->ustar_key, j.partial, j.sc, j.gb);
  else if (U16)
    block_min_body(bid - j.ns - j.nb, j.nm, part, DistU16{reinterpret_cast<const unsigned short*>(j.din), j.h0}, j.stride, j.cnt, j.blk, j.bmin);
  else
    block_min_body(bid - j.ns - j.nb, j.nm, part, DistF64{j.din, 0.0}, j.stride, j.cnt, j.blk, j.bmin);
}

#include "sets_exchange.inc.hpp"
#include "sets_goose.inc.hpp"

// Tail of a single-rank GoOSE sweep, two launches instead of four and no copy behind them (r03):
// k_goose_finals = k_sweep_finals<false> (a workgroup per slot) whose last workgroup to finish also does k_pick_target's job;
// k_arg_final_mirror = k_arg_final<false> of the explore slot, which then writes the whole result block and the Lipschitz keys
// into the host's pinned landing area and carries the sweep's end event (as k_sweep_finals does for SafeOpt).
template <int D>
__global__ __launch_bounds__(256) void k_goose_finals(const unsigned char* __restrict__ regions, size_t stride, int nparts, int q, SweepScalars* sc,
                                                      const SweepScalars* lane1, const CandSpec cs, double* __restrict__ target, int gb_on) {
  // workgroup s merges slot s (as k_sweep_finals<false>); the one that finishes last chooses the target
  if (lane1 && blockIdx.x == 0 && threadIdx.x == 0) {
    sc->n_amb_total += lane1->n_amb_total;
    sc->n_guard += lane1->n_guard - lane1->n_guard_cls;
  }
  const int slot = blockIdx.x;
  Best best;
  long long cnt, nb;
  merge_region<false>(reinterpret_cast<const Best*>(regions + (size_t)slot * stride), nparts, best, cnt, nb);
  if (threadIdx.x == 0) {
    store_slot<false>(sc, slot, best, nb, gb_on != 0);
    __hip_atomic_store(&sc->arg_val[slot], best.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&sc->arg_idx[slot], best.i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&sc->arg_d[slot], best.d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (slot > 0) sc->count_set[slot - 1] += cnt;
    __threadfence();
    const unsigned long long t = atomicAdd(&sc->ticket, 1ull);
    if (t == (unsigned long long)(q - 1)) {      // every slot is in memory (models/GoOSE.py:110-112, the first minimum wins)
      __threadfence();
      __hip_atomic_store(&sc->ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int best_c = 0;
      double bestv = 0.0, bestd = 0.0;
      long long besti = -1;
      for (int cc = 1; cc < q; ++cc) {
        const long long ai = __hip_atomic_load(&sc->arg_idx[cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double av = __hip_atomic_load(&sc->arg_val[cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ai >= 0 && (best_c == 0 || av < bestv)) { best_c = cc; bestv = av; besti = ai; bestd = __hip_atomic_load(&sc->arg_d[cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      }
      // the choice among the constraints' targets is itself a decision the band can leave open
      if (gb_on && best_c) {
        for (int cc = 1; cc < q; ++cc) {
          if (cc == best_c) continue;
          const long long ai = __hip_atomic_load(&sc->arg_idx[cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const double av = __hip_atomic_load(&sc->arg_val[cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const double ad = __hip_atomic_load(&sc->arg_d[cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // (the same candidate in two optimistic sets ties with itself under any posterior: the first minimum wins either way)
          if (ai >= 0 && ai != besti && !(av - ad > bestv + bestd)) { atomicAdd((unsigned long long*)&sc->n_guard, 1ull); break; }
        }
      }
      double x[D];
#pragma unroll
      for (int a = 0; a < D; ++a) x[a] = 0.0;
      if (best_c) cand_coords<D>(cs, besti - cs.first, x);
#pragma unroll
      for (int a = 0; a < D; ++a) target[a] = x[a];
    }
  }
}
__global__ __launch_bounds__(256) void k_arg_final_mirror(const Best* __restrict__ partial, int nparts, SweepScalars* sc, int slot,
                                                          unsigned char* mirror, const unsigned long long* __restrict__ Lkeys, int gb_on) {
  Best best;
  long long cnt, nb;
  merge_region<false>(partial, nparts, best, cnt, nb);
  if (threadIdx.x == 0) store_slot<false>(sc, slot, best, nb, gb_on != 0);
  __syncthreads();
  const unsigned long long* from = reinterpret_cast<const unsigned long long*>(sc);
  unsigned long long* to = reinterpret_cast<unsigned long long*>(mirror);
  for (unsigned i = threadIdx.x; i < kScalHost / 8; i += blockDim.x) to[i] = from[i];     // (the host's part of the block)
  if (threadIdx.x < kMaxQ) reinterpret_cast<unsigned long long*>(mirror + 3072)[threadIdx.x] = Lkeys[threadIdx.x];
}

// ---- host orchestration -------------------------------------------------------------------------------
static int reduce_blocks(const sbo_ctx* c) {
  const long long n = c->cs.n_local;
  return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, (long long)c->n_cu * 4));   // (per-block reduction tails cost more than extra grid-stride turns: x4 measured best)
}

// mask buffers of a sweep; called before the posterior is enqueued (K1b may write S / U itself) -- `b` is the sweep's
// confidence multiplier, handed to the posterior with the request to classify
static int sweep_masks(sbo_ctx* c, double b, bool may_fuse) {
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  int rc;
  long long npad_shard = n;                  // ranks > 1: the U mask is all-gathered with the largest shard's size
  for (size_t r = 0; r + 1 < c->first_of.size() && multi_rank(c) && c->sharded; ++r)
    npad_shard = std::max(npad_shard, c->first_of[r + 1] - c->first_of[r]);
  if ((rc = ensure(c->maskS, (size_t)n))) return rc;
  if ((rc = ensure(c->maskU, (size_t)npad_shard))) return rc;
  if ((rc = ensure(c->maskM, (size_t)n))) return rc;
  if ((rc = ensure(c->maskG, (size_t)n * std::max(1, q - 1)))) return rc;
  c->masks_bits = false;
  c->col_G_bytes = false;
  c->fuse_request = (may_fuse && q >= 2) ? (c->fuse_classify < 0 ? 2 : c->fuse_classify) : 0;
  if (c->fuse_request && q > 2) {            // (one S / U byte plane per constraint: bilinear.hip, k_classify_and)
    if ((rc = ensure(c->fuseS, (size_t)n * (q - 1))) || (rc = ensure(c->fuseU, (size_t)n * (q - 1)))) return rc;
  }
  c->lmax_defer = may_fuse;        // (every sweep merges K1b's Lipschitz partials in its k_classify_final)
  c->lmax_pending = false;
  c->fuse_b = b;
  c->fuse_rows = 0;
  return SBO_OK;
}

// (second lane of the set phase, see sbo_ctx::lane1 and lane_swap below)
static bool lanes_on(const sbo_ctx* c) { return c->set_lanes && !multi_rank(c) && c->mc.q >= 3 && c->cs.n_local > 0 && c->stream2; }

// the guard band in force for the running sweep: the posterior in the mean / var buffers came from an approximating kernel
// (K1b / K1t) -- nullptr for the exact kernels and for fp32 models (whose own recheck covers them)
static const GuardBand* gb_of(const sbo_ctx* c) {
  return (c->gb_active && !c->gb_off && c->guard_band && c->dtype == SBO_F64 && c->gb.p) ? (const GuardBand*)c->gb.p : nullptr;
}
static void launch_final(sbo_ctx* c, FinalJob* fj) {
  if (!fj || !fj->pending) return;
  fj->pending = false;
  // (with a second lane the fork event rides on this launch as its stop event: a separate record costs the stream a bubble)
  hipExtLaunchKernelGGL(k_classify_final, dim3(1), dim3(256), 0, c->stream, nullptr, fj->sc_copy ? c->ev_join[4] : nullptr, 0, fj->part,
                        fj->nparts, fj->pcap, fj->q, fj->sc, fj->Lpart, fj->per_out, fj->Lmax, fj->sc_copy, fj->gb, fj->b);
}

// `defer`: the merge of the classification's partials is handed back instead of launched (SafeOpt on one rank: it rides in
// the expander's first launch, which reads the U mask only)
template <typename T>
static int sweep_common_front(sbo_ctx* c, const sbo_sweep_opts* o, FinalJob* defer = nullptr) {
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  int rc;
  if ((rc = ensure(c->scal, sizeof(SweepScalars)))) return rc;
  const int nb = reduce_blocks(c);
  if ((rc = ensure(c->partial, partial_stride(nb)))) return rc;
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  FinalJob fj;
  fj.pending = true;
  fj.q = q;
  fj.sc = sc;
  fj.gb = gb_of(c);
  fj.b = o->b;
  if (lanes_on(c)) {
    if ((rc = ensure(c->lane1.scal, 4096))) return rc;
    fj.sc_copy = (SweepScalars*)c->lane1.scal.p;
  }
  if (c->lmax_pending) {
    fj.Lpart = (const double*)c->bl_lpart.p;
    fj.per_out = c->lmax_per_out;
    fj.Lmax = (unsigned long long*)c->Lmax.p;
    c->lmax_pending = false;
  }
  // four workgroups per CU measured best (2: 24.6 us, 4: 21.5, 8: 27.1 on config B's 4 M candidates; on config C's 1 M: 1024 /
  // 512 / 256 workgroups 0.1455 / 0.1463 / 0.1530 ms per sweep -- fewer is not better there either)
  int ncb = std::max(1, c->n_cu * 4);
  if (c->fuse_rows > 0 && n > 0) {
    // S / U bytes, |S|, |U| and the radius key came out of the posterior kernel (which sized the row buffer: cpart_cap): only u*
    // is left, over the safe candidates
    const int nob = std::max(1, c->n_cu * 4);
    unsigned long long* rows = (unsigned long long*)c->cpart.p;
    if (q > 2) {
      const PlaneAnd pa{(const uint8_t*)c->fuseS.p, (const uint8_t*)c->fuseU.p, n, q - 1, (uint8_t*)c->maskU.p};
      hipLaunchKernelGGL(k_classify_and<T>, dim3((unsigned)nob), dim3(256), 0, c->stream, (const T*)c->mean.p, (const T*)c->var.p, n, (T)o->b, pa,
                         (uint8_t*)c->maskS.p, rows + c->fuse_rows, c->cpart_cap);
    } else
    hipLaunchKernelGGL(k_classify_obj<T>, dim3((unsigned)nob), dim3(256), 0, c->stream, (const T*)c->mean.p, (const T*)c->var.p, n, (T)o->b,
                       (const uint8_t*)c->maskS.p, rows + c->fuse_rows, c->cpart_cap);
    fj.part = (const unsigned long long*)rows;
    fj.nparts = c->fuse_rows + nob;
    fj.pcap = c->cpart_cap;
    if (defer) *defer = fj;
    else launch_final(c, &fj);
    c->amb_clean = true;
    SBO_HIP(hipGetLastError());
    return SBO_OK;
  }
  if ((rc = ensure(c->cpart, sizeof(unsigned long long) * kClassifyRow * (size_t)ncb))) return rc;
  c->cpart_cap = (int)(c->cpart.bytes / (sizeof(unsigned long long) * kClassifyRow));
  if (n > 0)
    hipLaunchKernelGGL((k_classify<T>), dim3((unsigned)ncb), dim3(256), 0, c->stream, (const T*)c->mean.p, (const T*)c->var.p, n, q,
                       (T)o->b, (uint8_t*)c->maskS.p, (uint8_t*)c->maskU.p, (unsigned long long*)c->cpart.p, c->cpart_cap, gb_of(c));
  fj.part = (const unsigned long long*)c->cpart.p;
  fj.nparts = n > 0 ? ncb : 0;
  fj.pcap = c->cpart_cap;
  if (defer) *defer = fj;
  else launch_final(c, &fj);
  c->amb_clean = true;
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

template <typename T, int D>
static int launch_exact(sbo_ctx* c, const sbo_sweep_opts* o, int cidx, int lidx, uint8_t* G) {
  const long long n = c->cs.n_local;
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  const T* mean_c = (const T*)c->mean.p + (size_t)cidx * n;
  const T* var_c = (const T*)c->var.p + (size_t)cidx * n;
  CandSpec csU = c->cs;          // the witness set: every candidate that can matter (ranks > 1: the transform's window)
  const uint8_t* Uall = (const uint8_t*)c->maskU.p;
  if (multi_rank(c)) {
    csU.first = c->uwin_first;
    csU.n_local = c->uwin_n;
    Uall = (const uint8_t*)c->Uwin.p;
  }
  RcExp rx;
  memset(&rx, 0, sizeof(rx));
  if (gb_of(c) && !c->rc_active) {        // (fast path of an approximating posterior: the listed candidates' verdicts are judged against its band)
    rx.gb_c = cidx;
    rx.gb_l = c->gb_slow ? -1 : lidx;
  }
  hipLaunchKernelGGL((k_expander_exact<T, D>), dim3(1024), dim3(256), 0, c->stream, c->cs, csU, mean_c, var_c, (T)o->b,
                     Uall, (const unsigned long long*)c->Lmax.p, lidx, sc,
                     (const long long*)c->amb.p, G, rx);
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

template <typename T>
static int launch_exact_d(sbo_ctx* c, const sbo_sweep_opts* o, int cidx, int lidx, uint8_t* G) {
  switch (c->mc.dpad) {
    case 2: return launch_exact<T, 2>(c, o, cidx, lidx, G);
    case 4: return launch_exact<T, 4>(c, o, cidx, lidx, G);
    case 8: return launch_exact<T, 8>(c, o, cidx, lidx, G);
  }
  return fail(SBO_E_UNSUPPORTED, "unsupported padded dimension");
}

static int sweep_exchange_wait(sbo_ctx* c);

// Halo of a rank's transform window in hyper-planes of the slowest axis: witnesses (expanders) / sources (GoOSE) further than
// the largest radius rmax / L cannot change a verdict.  One formula for the host (keys read back) and the device (the check
// of a speculative window against the keys of the running sweep).
__host__ __device__ __forceinline__ long long halo_planes(double L, double rmax, double hl, long long planes_total) {
  long long H = planes_total;
  if (L > 0 && hl > 0) {
    const double cap = rmax / L * 1.000001 + 1e-6;
    const double hp = ceil(cap / hl) + 2.0;
    if (hp < (double)planes_total) H = (long long)hp;
  }
  return H;
}
__global__ void k_halo_check(SweepScalars* sc, const unsigned long long* Lkeys, int lidx, int cidx, double hl, long long H_used,
                             long long planes_total) {
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const double rmax = sc->rmax_key[cidx] ? ord_val(sc->rmax_key[cidx]) : 0.0;
  if (halo_planes(L, rmax, hl, planes_total) > H_used) sc->halo_short = 1;
}
// the window's halo for constraint cidx: the guess from the previous sweep (checked on the device), or the keys of this one
// (the host waits for their read-back)
static int halo_for(sbo_ctx* c, int cidx, int lidx, double hl, long long planes_total, long long* H_out) {
  if (c->halo_spec && c->halo_guess[cidx] >= 0) {
    long long H = std::min(planes_total, c->halo_guess[cidx]);
    // (test hook: this rank alone guesses one plane -- tests/test_gpu_parity.py: the rerun must be every rank's decision)
    static const bool test_short = getenv("SBO_TEST_HALO_SHORT") != nullptr;
    if (test_short) H = std::min<long long>(H, 1);
    hipLaunchKernelGGL(k_halo_check, dim3(1), dim3(1), 0, c->stream, (SweepScalars*)c->scal.p, (const unsigned long long*)c->Lmax.p, lidx, cidx,
                       hl, H, planes_total);
    *H_out = H;
    return SBO_OK;
  }
  int rc;
  if ((rc = sweep_exchange_wait(c))) return rc;
  double L, rmax = 0.0;
  memcpy(&L, &c->h_c1[1 + lidx], 8);
  if (c->h_c1[1 + kMaxQ + cidx]) rmax = ord_val(c->h_c1[1 + kMaxQ + cidx]);
  *H_out = halo_planes(L, rmax, hl, planes_total);
  return SBO_OK;
}
// after a sweep: next sweep's guesses from this sweep's global keys (a quarter more, so that a slowly growing radius stays inside)
static void halo_learn(sbo_ctx* c, const SweepScalars& h, const unsigned long long* Lk, int quirk) {
  if (!multi_rank(c) || c->cs.kind != 1) return;
  const int q = c->mc.q, d = c->cs.d;
  const double hl = d >= 2 ? c->cs.step[d - 1] : c->cs.step[0];
  const long long planes_total = c->cs.count[d - 1];
  for (int cc = 1; cc < q; ++cc) {
    const int lidx = quirk ? q - 1 : cc;
    double L;
    memcpy(&L, &Lk[lidx], 8);
    const double rmax = h.rmax_key[cc] ? ord_val(h.rmax_key[cc]) : 0.0;
    const long long H = halo_planes(L, rmax, hl, planes_total);
    c->halo_guess[cc] = H >= planes_total ? planes_total : std::min(planes_total, H + H / 4 + 2);
  }
}

// G_c for constraint cidx (1..q-1) into G[n]
static void launch_edt_axis0(sbo_ctx* c, const uint8_t* U, long long nlines, int count0, double h0, double* D, hipStream_t st = nullptr) {
  if (!st) st = c->stream;
  if ((count0 & 63) == 0 && count0 <= 4096 && ((uintptr_t)U & 7) == 0)
    hipLaunchKernelGGL(k_edt_axis0_waves, dim3((unsigned)std::max<long long>(1, std::min<long long>((nlines + 3) / 4, (long long)c->n_cu * 16))), dim3(256), 0,
                       st, U, nlines, count0, h0, D);
  else if (count0 <= kAxis0Max && count0 >= 128)
    hipLaunchKernelGGL(k_edt_axis0_wg<false>, dim3((unsigned)std::min<long long>(nlines, 1 << 20)), dim3(256), 0, st, U, nlines,
                       count0, h0, D, CoarseGrid{});
  else
    hipLaunchKernelGGL(k_edt_axis0, dim3((unsigned)((nlines + 3) / 4)), dim3(256), 0, st, U, nlines, count0, h0, D);
}

// ---- second lane of the set phase (see sbo_ctx::lane1) ----------------------------------------------------------------
static void lane_swap(sbo_ctx* c) {
  auto& l = c->lane1;
  std::swap(c->stream, c->stream2);
  std::swap(c->dist2, l.dist2);
  std::swap(c->dist2b, l.dist2b);
  std::swap(c->coarse, l.coarse);
  std::swap(c->blockmin, l.blockmin);
  std::swap(c->blockmax, l.blockmax);
  std::swap(c->scanlist, l.scanlist);
  std::swap(c->amb, l.amb);
  std::swap(c->gw, l.gw);
  std::swap(c->runmeta, l.runmeta);
  std::swap(c->scal, l.scal);
  std::swap(c->amb_clean, l.amb_clean);
}
// after the classification's scalars are final on the main stream: lane 1 waits for them and takes its copy of the block
// after the classification's scalars are final on the main stream (k_classify_final has written lane 1's snapshot of them)
static int lanes_fork(sbo_ctx* c) {
  SBO_HIP(hipStreamWaitEvent(c->stream2, c->ev_join[4], 0));          // (recorded by the k_classify_final launch)
  c->lane1.amb_clean = true;
  return SBO_OK;
}
static int lanes_join(sbo_ctx* c) {
  SBO_HIP(hipEventRecord(c->ev_join[5], c->stream2));
  SBO_HIP(hipStreamWaitEvent(c->stream, c->ev_join[5], 0));
  return SBO_OK;
}
struct LaneScope {          // enqueue-time view of lane 1 (odd lanes swap the context's stream / scratch in, and back out)
  sbo_ctx* c;
  bool on;
  LaneScope(sbo_ctx* c_, bool on_) : c(c_), on(on_) { if (on) lane_swap(c); }
  ~LaneScope() { if (on) lane_swap(c); }
};

constexpr long long kListExpanderMax = 1ll << 21;     // explicit lists: largest candidate set with exhaustive expander sets

// the minimiser launch of a SafeOpt sweep, held back so that the first constraint's expander can take it into k_set_mid
struct MinimizerJob {
  bool pending = false;
  int nb = 0;
  Best* partial = nullptr;
  FinalJob fin;                 // the classification's merge, when it too waits for the expander's first launch
};
template <typename T>
static void launch_minimizer(sbo_ctx* c, const sbo_sweep_opts* o, MinimizerJob* mj) {
  if (mj) launch_final(c, &mj->fin);
  if (!mj || !mj->pending) return;
  mj->pending = false;
  hipLaunchKernelGGL((k_minimizer<T>), dim3(mj->nb), dim3(256), 0, c->stream, (const T*)c->mean.p, (const T*)c->var.p, c->cs.n_local,
                     (long long)c->cs.first, (T)o->b, (const uint8_t*)c->maskS.p, (uint8_t*)c->maskM.p, (SweepScalars*)c->scal.p,
                     mj->partial, gb_of(c));
}

// `lazy_exact`: the exhaustive recheck of in-band candidates (k_expander_exact) is NOT launched -- the caller looks at the
// sweep's n_amb afterwards and runs it (and everything behind it) only when something was listed, which is rare
template <typename T>
static int expander_set(sbo_ctx* c, const sbo_sweep_opts* o, int cidx, uint8_t* G, MinimizerJob* mj = nullptr, bool lazy_exact = false) {
  const long long n = c->cs.n_local;
  if (n == 0) { launch_minimizer<T>(c, o, mj); return SBO_OK; }
  const int q = c->mc.q;
  const int lidx = o->reference_quirk_L_index ? q - 1 : cidx;   // models/SafeOpt.py:110 (loop-leaked i)
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  const T* mean_c = (const T*)c->mean.p + (size_t)cidx * n;
  const T* var_c = (const T*)c->var.p + (size_t)cidx * n;
  const int nb = reduce_blocks(c);
  int rc;
  if ((rc = ensure(c->amb, sizeof(long long) * (size_t)n))) return rc;
  if (!c->amb_clean) hipLaunchKernelGGL(k_reset_amb, dim3(1), dim3(1), 0, c->stream, sc);   // (k_classify_final left the counters at zero)
  c->amb_clean = false;
  const int d_ = c->cs.d;
  long long plane = 1;                       // candidates per step of the slowest axis
  for (int a = 0; a < d_ - 1; ++a) plane *= c->cs.count[a];
  const bool plane_aligned = c->cs.kind == 1 && c->cs.first % plane == 0 && n % plane == 0;
  if (c->cs.kind == 1 && plane_aligned) {
    const int d = d_;
    // Window of the transform: this rank's hyper-planes of the slowest axis plus, with ranks > 1, a halo of
    // ceil(cap / h) planes on either side taken from the all-gathered U mask (cap = largest radius that can matter,
    // from the keys of collective C1).  Witnesses further away cannot change a verdict, so the window is exact.
    const long long planes_total = multi_rank(c) ? c->cs.count[d - 1] : n / plane;
    long long p0 = multi_rank(c) ? c->cs.first / plane : 0, p1 = p0 + n / plane;
    const long long own0 = p0;
    if (multi_rank(c)) {
      const double hl = d >= 2 ? c->cs.step[d - 1] : c->cs.step[0];
      long long H;
      if ((rc = halo_for(c, cidx, lidx, hl, planes_total, &H))) return rc;
      p0 = std::max(0ll, p0 - H);
      p1 = std::min(planes_total, p1 + H);
    }
    const long long wplanes = p1 - p0;
    const long long nt = wplanes * plane;
    const long long goff = (own0 - p0) * plane;                 // own candidates start here inside the window
    const uint8_t* Uall = (const uint8_t*)c->maskU.p;
    if (multi_rank(c)) {
      // bytes of the window only, out of the all-gathered bit words (the gather itself was queued ahead of the host's wait)
      if ((rc = ensure(c->Uwin, (size_t)nt))) return rc;
      const unsigned long long* recvw = (const unsigned long long*)c->ubits.p + c->gather_words + kC1Head;   // (past the own block and a head)
      hipLaunchKernelGGL(k_unpack_shards, dim3((unsigned)std::min<long long>((nt + 255) / 256, 1 << 16)), dim3(256), 0, c->stream, recvw,
                         c->gather_words, c->world, (const long long*)c->shard_first.p, p0 * plane, nt, (uint8_t*)c->Uwin.p);
      c->uwin_first = p0 * plane;
      c->uwin_n = nt;
      Uall = (const uint8_t*)c->Uwin.p;
    }
    if ((rc = ensure(c->dist2, sizeof(double) * (size_t)nt))) return rc;
    if (d > 2 && (rc = ensure(c->dist2b, sizeof(double) * (size_t)nt))) return rc;
    const int count0 = d >= 2 ? (int)c->cs.count[0] : (int)nt;  // d == 1: the window is one line
    const long long nlines = nt / count0;
    // coarse transform of the window (tiny): lets most candidates decide without the per-candidate scan
    CoarseGrid cg;
    memset(&cg, 0, sizeof(cg));
    cg.d = d;
    bool coarse_ok = d >= 2 && nt >= (1ll << 16);
    long long nc = 1;
    double h2 = 0.0, hmax = 0.0;
    for (int a = 0; a < d; ++a) {
      cg.count[a] = a == d - 1 ? wplanes : c->cs.count[a];
      cg.ccount[a] = (cg.count[a] + kCoarse - 1) / kCoarse;
      nc *= cg.ccount[a];
      h2 += c->cs.step[a] * c->cs.step[a];
      hmax = std::max(hmax, c->cs.step[a]);
      if (cg.count[a] < 4 * kCoarse) coarse_ok = false;
    }
    double* dc0 = nullptr;
    double* dc1 = nullptr;
    uint8_t* Uc = nullptr;
    const int cc0 = (int)cg.ccount[0];
    const long long clines = nc / cc0;
    if (coarse_ok) {
      cg.enabled = 1;
      cg.delta = (kCoarse - 1) * std::sqrt(h2) * (1.0 + 1e-9);
      const size_t cbytes = ((size_t)nc * (1 + 2 * sizeof(double)) + 64 + 255) / 256 * 256;
      if ((rc = ensure(c->coarse, cbytes))) return rc;
      dc0 = (double*)c->coarse.p;
      dc1 = dc0 + nc;
      Uc = (uint8_t*)(dc1 + nc);
    }
    const double cap_extra = 2.0 * cg.delta + 2.0 * kCoarse * hmax;
    const int last_cnt_ = d >= 2 ? (int)wplanes : 1;
    const int blk_ = last_cnt_ >= 8192 ? 64 : 32;
    const bool want_bmin = d >= 2 && last_cnt_ >= 4 * blk_ && c->scan_blocks;
    // 2-D grids: the two axis-0 passes share a launch, and so do the coarse last-axis scan, the minimiser and the block minima
    const bool paired = d == 2 && coarse_ok && c->set_fuse && count0 <= kAxis0Max && count0 >= 128 && cc0 >= 128;
    bool bmin_done = false, u16 = false;
    double* din = (double*)c->dist2.p;
    double* dout = (double*)c->dist2b.p;
    long long stride = count0;
    if (paired) {
      // fine lines of whole words, up to 4096 positions: a wave per line
      const int wave_lines = ((count0 & 63) == 0 && count0 <= 4096 && nlines < (1ll << 22)) ? 1 : 0;
      const int ncoarse = (int)std::min<long long>(clines, 1 << 20);
      // (wave form: four workgroups of 39 KB LDS fit a CU; no more fine workgroups than are resident beside the coarse ones)
      const int nfine = wave_lines ? (int)std::min<long long>((nlines + 3) / 4, std::max<long long>(c->n_cu, 4ll * c->n_cu - ncoarse - 1))
                                   : (int)std::min<long long>(nlines, 1 << 20);
      FinalJob fin;
      if (mj && mj->fin.pending) {
        fin = mj->fin;
        mj->fin.pending = false;
      }
      // the fine image as 16-bit step counts (0xffff: no U point on the line) -- only when its readers are the block minima
      // and the list scan, which decode it: without the list (short last axis, options scan_blocks / scan_waves = 0) the
      // verdict kernel scans the image itself and expects squared distances as doubles
      u16 = count0 < 65535 && want_bmin && blk_ <= 64 && c->scan_waves;
      if (u16)
        hipLaunchKernelGGL(k_edt_axis0_pair<true>, dim3((unsigned)(nfine + ncoarse + (fin.pending ? 1 : 0))), dim3(256), 0, c->stream, Uall,
                           nlines, count0, c->cs.step[0], din, nfine, ncoarse, clines, cc0, c->cs.step[0] * kCoarse, dc0, cg, fin, wave_lines);
      else
        hipLaunchKernelGGL(k_edt_axis0_pair<false>, dim3((unsigned)(nfine + ncoarse + (fin.pending ? 1 : 0))), dim3(256), 0, c->stream, Uall,
                           nlines, count0, c->cs.step[0], din, nfine, ncoarse, clines, cc0, c->cs.step[0] * kCoarse, dc0, cg, fin, wave_lines);
      MidJobs<T> j;
      memset(&j, 0, sizeof(j));
      j.sc = sc;
      j.Lkeys = (const unsigned long long*)c->Lmax.p;
      j.ns = (int)std::min<long long>((nc + 255) / 256, 1 << 16);
      j.dc_in = dc0;
      j.dc_out = dc1;
      j.nc = nc;
      j.cstride = cc0;
      j.ccnt = (int)cg.ccount[1];
      j.hc = c->cs.step[1] * kCoarse;
      j.cidx = cidx;
      j.lidx = lidx;
      j.cap_extra = cap_extra;
      if (mj && mj->pending) {
        mj->pending = false;
        j.nb = mj->nb;
        j.mean0 = (const T*)c->mean.p;
        j.var0 = (const T*)c->var.p;
        j.n = n;
        j.first = (long long)c->cs.first;
        j.b = (T)o->b;
        j.S = (const uint8_t*)c->maskS.p;
        j.M = (uint8_t*)c->maskM.p;
        j.partial = mj->partial;
        j.gb = gb_of(c);
      }
      if (want_bmin) {
        const int nblocks = (last_cnt_ + blk_ - 1) / blk_;
        if ((rc = ensure(c->blockmin, sizeof(double) * (size_t)nblocks * (size_t)stride))) return rc;
        j.nm = (int)std::min<long long>((stride + 63) / 64 * nblocks, 1 << 20);
        j.din = din;
        j.stride = stride;
        j.cnt = last_cnt_;
        j.blk = blk_;
        j.bmin = (double*)c->blockmin.p;
        bmin_done = true;
      }
      j.h0 = c->cs.step[0];
      if (u16) hipLaunchKernelGGL((k_set_mid<T, true>), dim3((unsigned)(j.ns + j.nb + j.nm)), dim3(256), 0, c->stream, j);
      else hipLaunchKernelGGL((k_set_mid<T, false>), dim3((unsigned)(j.ns + j.nb + j.nm)), dim3(256), 0, c->stream, j);
      cg.Dc = dc1;
    } else {
      launch_minimizer<T>(c, o, mj);
      launch_edt_axis0(c, Uall, nlines, count0, c->cs.step[0], din);
      for (int a = 1; a < d - 1; ++a) {
        hipLaunchKernelGGL(k_edt_scan, dim3((unsigned)std::min<long long>((nt + 255) / 256, 1 << 20)), dim3(256), 0, c->stream,
                           (const double*)din, dout, nt, stride, (int)c->cs.count[a], c->cs.step[a],
                           (const SweepScalars*)sc, cidx, (const unsigned long long*)c->Lmax.p, lidx, 0, 0.0);
        std::swap(din, dout);
        stride *= c->cs.count[a];
      }
      if (coarse_ok) {
        if (cc0 <= kAxis0Max && cc0 >= 128) {
          // the coarse axis-0 pass forms the cells' bits from the fine mask itself
          hipLaunchKernelGGL(k_edt_axis0_wg<true>, dim3((unsigned)std::min<long long>(clines, 1 << 20)), dim3(256), 0, c->stream, Uall,
                             clines, cc0, c->cs.step[0] * kCoarse, dc0, cg);
        } else {
          hipLaunchKernelGGL(k_coarsen_mask, dim3((unsigned)std::min<long long>((nc + 255) / 256, 1 << 16)), dim3(256), 0, c->stream,
                             Uall, cg, nc, Uc);
          launch_edt_axis0(c, (const uint8_t*)Uc, clines, cc0, c->cs.step[0] * kCoarse, dc0);
        }
        long long cstride = cc0;
        for (int a = 1; a < d; ++a) {
          hipLaunchKernelGGL(k_edt_scan, dim3((unsigned)std::min<long long>((nc + 255) / 256, 1 << 16)), dim3(256), 0, c->stream,
                             (const double*)dc0, dc1, nc, cstride, (int)cg.ccount[a], c->cs.step[a] * kCoarse, (const SweepScalars*)sc,
                             cidx, (const unsigned long long*)c->Lmax.p, lidx, 0, cap_extra);
          std::swap(dc0, dc1);
          cstride *= cg.ccount[a];
        }
        cg.Dc = dc0;
      }
    }
    double xscale = 0.0;
    for (int a = 0; a < d; ++a) xscale = std::max(xscale, std::max(std::fabs(c->cs.lo[a]), std::fabs(c->cs.hi[a])));
    const int last_cnt = d >= 2 ? (int)wplanes : 1;
    const double last_h = d >= 2 ? c->cs.step[d - 1] : 0.0;
    {
      const double* bmin = bmin_done ? (const double*)c->blockmin.p : nullptr;
      const int blk = blk_;
      if (!bmin_done && want_bmin) {
        const long long nb_ = (long long)((last_cnt + blk - 1) / blk) * stride;
        if ((rc = ensure(c->blockmin, sizeof(double) * (size_t)nb_))) return rc;
        hipLaunchKernelGGL(k_block_min, dim3((unsigned)std::min<long long>((stride + 63) / 64 * ((last_cnt + blk - 1) / blk), 1 << 20)), dim3(256), 0, c->stream,
                           (const double*)din, stride, last_cnt, blk, (double*)c->blockmin.p);
        bmin = (const double*)c->blockmin.p;
      }
      long long* slist = nullptr;
      if (bmin && blk <= 64 && c->scan_waves) {
        if ((rc = ensure(c->scanlist, 2 * sizeof(long long) * (size_t)n))) return rc;   // (candidate, ucb) pairs
        slist = (long long*)c->scanlist.p;
      }
      const long long len0 = d >= 2 ? count0 : n;               // positions per line / local lines
      const long long nl = n / len0;
      RcExp rx;                                                  // fp32 model with fp64 recheck: unrefined entries carry a band
      memset(&rx, 0, sizeof(rx));
      if (c->rc_active) {
        rx.refined = (const uint8_t*)c->rc_refined.p;
        const double ys = std::max(1.0, c->mc.Y_std[cidx]);
        rx.dm = 1e-4 * ys;
        rx.dv = 1e-4 * ys * ys;
        rx.list = (long long*)((char*)c->rc_list.p + kRcList);
        rx.count = (unsigned long long*)((char*)c->rc_list.p + kRcCount2);
      }
      if (gb_of(c)) {                                            // guard band of an approximating fp64 posterior (fast path: no list)
        rx.gb_c = cidx;
        rx.gb_l = c->gb_slow ? -1 : lidx;                        // (the slow path has recomputed the Lipschitz keys exactly)
      }
      const dim3 dgrid((unsigned)((len0 + 255) / 256), (unsigned)std::min<long long>((nl + kDecideLines - 1) / kDecideLines, 65535));
#define SBO_DECIDE(LIST)                                                                                                            \
  hipLaunchKernelGGL((k_edt_decide<T, LIST>), dgrid, dim3(256), 0, c->stream, (const double*)din, nl, (int)len0, goff / len0, goff, \
                     d >= 2 ? stride : 1, last_cnt, last_h, d, xscale, mean_c, var_c, (T)o->b, (const uint8_t*)c->maskS.p,          \
                     (const unsigned long long*)c->Lmax.p, lidx, sc, G, (long long*)c->amb.p, cg, bmin, blk, slist, rx)
      // (grids whose lines are whole 8-byte mask words: eight candidates per lane, see k_edt_decide8)
      const bool wide = slist && len0 % 8 == 0 && d >= 2 && cg.enabled && ((uintptr_t)G & 7) == 0 &&
                        ((uintptr_t)c->maskS.p & 7) == 0;
      if (wide) {
        const dim3 g8((unsigned)((len0 / 8 + 255) / 256), (unsigned)std::min<long long>(nl, 65535));
        hipLaunchKernelGGL((k_edt_decide8<T>), g8, dim3(256), 0, c->stream, nl, (int)len0, goff / len0, goff, d, xscale, mean_c, var_c, (T)o->b,
                           (const uint8_t*)c->maskS.p, (const unsigned long long*)c->Lmax.p, lidx, sc, G, cg, slist, rx);
      } else if (slist) SBO_DECIDE(true);
      else SBO_DECIDE(false);
#undef SBO_DECIDE
      // (workgroups of the list scan: 2048 on config B's 4 M candidates, 4096 on H's 16 M -- -8 us there)
      const int scan_wgs = (int)std::min<long long>(4096, std::max<long long>(1024, n / 2048));
      if (slist) {
        // lanes per listed candidate: 16 by default (more candidates in flight beat shorter rounds: 53 k open candidates
        // of config B take 19 us with 16 lanes, 32 us with 32), never more than a wave, 64 for 64-step blocks on request
        const int gl = c->scan_waves == 8 || c->scan_waves == 32 || c->scan_waves == 64 ? c->scan_waves : 16;
#define SBO_SCAN_LIST(GL)                                                                                                       \
  if (u16)                                                                                                                      \
    hipLaunchKernelGGL((k_edt_scan_list<T, GL, DistU16>), dim3(scan_wgs), dim3(256), 0, c->stream,                              \
                       DistU16{reinterpret_cast<const unsigned short*>(din), c->cs.step[0]}, goff, stride, last_cnt,            \
                       last_h, d, xscale, mean_c, var_c, (T)o->b, (const unsigned long long*)c->Lmax.p, lidx, sc, G,            \
                       (long long*)c->amb.p, bmin, blk, (const long long*)slist, rx);                                           \
  else                                                                                                                          \
  hipLaunchKernelGGL((k_edt_scan_list<T, GL, DistF64>), dim3(scan_wgs), dim3(256), 0, c->stream, DistF64{(const double*)din, 0.0}, goff, stride, last_cnt, \
                     last_h, d, xscale, mean_c, var_c, (T)o->b, (const unsigned long long*)c->Lmax.p, lidx, sc, G,              \
                     (long long*)c->amb.p, bmin, blk, (const long long*)slist, rx)
        switch (gl) {
          case 8: SBO_SCAN_LIST(8); break;
          case 32: SBO_SCAN_LIST(32); break;
          case 64: SBO_SCAN_LIST(64); break;
          default: SBO_SCAN_LIST(16); break;
        }
#undef SBO_SCAN_LIST
      }
    }
  } else {
    // explicit candidate lists, and grid ranges that are not whole hyper-planes: exhaustive evaluation
    launch_minimizer<T>(c, o, mj);
    // (quadratic: every safe candidate against every U point, like the reference's vmap -- fine for the lists a campaign
    // uses, seconds at the cap)
    if (n > kListExpanderMax)
      return fail(SBO_E_UNSUPPORTED, "expander sets need a grid of whole hyper-planes, or at most 2097152 candidates (exhaustive)");
    if (multi_rank(c))
      return fail(SBO_E_UNSUPPORTED, "expander sets on explicit candidate lists are single-rank");
    hipLaunchKernelGGL(k_list_safe, dim3(nb), dim3(256), 0, c->stream, (const uint8_t*)c->maskS.p, n, sc, G,
                       (long long*)c->amb.p);
  }
  SBO_HIP(hipGetLastError());
  if (lazy_exact && c->cs.kind == 1 && plane_aligned) return SBO_OK;
  return launch_exact_d<T>(c, o, cidx, lidx, G);
}

static void coords_of(const sbo_ctx* c, long long gidx, double* x) {
  // host restatement of cand_coords for result decoding (grid) or a small D2H read (explicit list)
  for (int a = 0; a < SBO_MAX_D; ++a) x[a] = 0.0;
  if (gidx < 0) return;
  if (c->cs.kind == 1) {
    long long f = gidx;
    for (int a = 0; a < c->cs.d; ++a) {
      const long long cnt = c->cs.count[a];
      const long long i = f % cnt;
      f /= cnt;
      x[a] = (i == cnt - 1 && cnt > 1) ? c->cs.hi[a] : c->cs.lo[a] + (double)i * c->cs.step[a];
    }
  } else {
    const long long loc = gidx - c->cs.first;
    if (loc < 0 || loc >= c->cs.n_local) return;   // owned by another rank: filled by the caller's exchange
    if (c->cs.pts_dtype == SBO_F64) {
      (void)hipMemcpy(x, (const double*)c->pts.p + loc * c->cs.d, sizeof(double) * c->cs.d, hipMemcpyDeviceToHost);
    } else {
      float tmp[SBO_MAX_D];
      (void)hipMemcpy(tmp, (const float*)c->pts.p + loc * c->cs.d, sizeof(float) * c->cs.d, hipMemcpyDeviceToHost);
      for (int a = 0; a < c->cs.d; ++a) x[a] = tmp[a];
    }
  }
}

int sbo_posterior_enqueue_(sbo_ctx* c);

// (ranks > 1) C1: global u*, L and radius keys; C2: every rank's U mask as bit words (one all-gather carries both)
template <typename T>
static int sweep_exchange_front(sbo_ctx* c, const sbo_sweep_opts* o, bool need_U) {
  const int q = c->mc.q;
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  if (!multi_rank(c)) return SBO_OK;   // (the radius keys, max over S of ucb_c, come out of k_classify)
  int rc;
  if (!c->sharded && q > 1 && need_U)
    return fail(SBO_E_INVALID, "multi-rank sweeps with constraints need sbo_candidates_grid_sharded");
  if ((rc = ensure(c->xch, sizeof(double) * (size_t)(c->world * kC3Row + 64)))) return rc;
  unsigned long long* kb = (unsigned long long*)c->xch.p;
  if (need_U && q > 1) {
    // C1 + C2 in ONE all-gather: every rank sends [its keys (kC1Head words) | its U mask as bits (one ballot word per 64
    // candidates: 8x fewer bytes on the links than the byte mask)]; the keys are then max-reduced over the gathered heads by
    // a small kernel on every rank -- one collective latency per sweep less than an all-reduce followed by an all-gather.
    long long maxlocal = 0;
    for (int r = 0; r < c->world; ++r) maxlocal = std::max(maxlocal, c->first_of[r + 1] - c->first_of[r]);
    const long long words = (maxlocal + 63) / 64, stride = kC1Head + words;
    // (their own buffer: the words are read again per constraint, after GoOSE's weight exchanges have used `gather`)
    if ((rc = ensure(c->ubits, sizeof(unsigned long long) * (size_t)stride * (c->world + 1)))) return rc;
    c->gather_words = stride;
    unsigned long long* sendw = (unsigned long long*)c->ubits.p;
    unsigned long long* recvw = sendw + stride;
    hipLaunchKernelGGL(k_pack_c1, dim3(1), dim3(64), 0, c->stream, (const SweepScalars*)sc, (const unsigned long long*)c->Lmax.p, sendw);
    hipLaunchKernelGGL(k_pack_bits, dim3((unsigned)std::min<long long>((words + 3) / 4, 1 << 16)), dim3(256), 0, c->stream,
                       (const uint8_t*)c->maskU.p, c->cs.n_local, words, sendw + kC1Head);
    if ((rc = comm_allgather_bytes(c, sendw, recvw, sizeof(unsigned long long) * (size_t)stride))) return rc;
    hipLaunchKernelGGL(k_unpack_c1_gathered, dim3(1), dim3(64), 0, c->stream, sc, (unsigned long long*)c->Lmax.p,
                       (const unsigned long long*)recvw, stride, c->world, kb);
    // (the bits are expanded to bytes per constraint, window only: expander_set)
  } else {
    hipLaunchKernelGGL(k_pack_c1, dim3(1), dim3(64), 0, c->stream, (const SweepScalars*)sc, (const unsigned long long*)c->Lmax.p, kb);
    if ((rc = comm_allreduce_max_u64(c, kb, kC1Words))) return rc;
    hipLaunchKernelGGL(k_unpack_c1, dim3(1), dim3(64), 0, c->stream, sc, (unsigned long long*)c->Lmax.p, (const unsigned long long*)kb);
  }
  // the host needs the global L and radius keys to size the halo of the expander transform: the read-back goes to
  // pinned memory and is waited for only where the window is computed (sweep_exchange_wait), so the minimiser kernels are
  // already queued behind it and the GPU does not idle through the round trip
  SBO_HIP(hipMemcpyAsync(c->h_c1, kb, sizeof(unsigned long long) * (1 + 2 * kMaxQ), hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipEventRecord(c->ev[5], c->stream));
  c->c1_pending = true;
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

static int sweep_exchange_wait(sbo_ctx* c) {
  if (c->c1_pending) {
    SBO_HIP(hipEventSynchronize(c->ev[5]));
    c->c1_pending = false;
    ++c->host_syncs;
  }
  return SBO_OK;
}

// C3 + host merge: every rank's slots and counters -> global ones.  slot_is_max[i] selects arg-max / arg-min.
static int sweep_exchange_back(sbo_ctx* c, SweepScalars& h, const bool* slot_is_max, unsigned long long* Lk = nullptr,
                               hipEvent_t done_ev = nullptr, bool mirrored = false /* the last kernel wrote h_back and carries done_ev */) {
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  // (the Lipschitz keys ride in the same read-back: one synchronisation per sweep)
  // (pinned landing area: pageable destinations are staged by the runtime, ~20 us per copy)
  // (the Lipschitz keys live 3 KB into the same allocation: sbo_create)
  static_assert(sizeof(SweepScalars) <= 2048 && sizeof(unsigned long long) * kMaxQ <= 512, "read-back area");
  constexpr size_t kBack = 3072 + sizeof(unsigned long long) * kMaxQ;
  const unsigned long long* Lk_pinned = (const unsigned long long*)(c->h_back + 3072);
  if (!multi_rank(c)) {
    if (!mirrored) {
      SBO_HIP(hipMemcpyAsync(c->h_back, sc, kBack, hipMemcpyDeviceToHost, c->stream));
      if (done_ev) SBO_HIP(hipEventRecord(done_ev, c->stream));
    }
    // (a caller's invK whose reverse factor is still to be made: its ~20 launches are enqueued now, on their own stream,
    // while the host would otherwise only wait for the sweep -- model.hip: model_factor_enqueue)
    // (r04: no longer -- the chain's twenty launches run BESIDE the next model's upload, plan and sweep and cost them CUs: k_bpost
    // 206 -> 217 us, the plan's small kernels up to 4 x; the GEMM posteriors never need the factor, and whoever does --
    // the O(n^2) kernels, sbo_model_append -- enqueues it and waits in factor_sync)
    SBO_HIP(stream_wait(c, c->stream));
    ++c->host_syncs;
    memset(&h, 0, sizeof(h));
    memcpy(&h, c->h_back, kScalHost);
    if (Lk) memcpy(Lk, Lk_pinned, sizeof(unsigned long long) * kMaxQ);
    // decisions the guard band of an approximating posterior leaves open: the classification's and the verdict kernels' count
    // plus what the final reductions found in their slots
    if (getenv("SBO_DEBUG_GUARD") && c->gb_active) {
      fprintf(stderr, "[guard] device count %lld (classification %lld), slots:", h.n_guard, h.n_guard_cls);
      for (int t = 0; t < kArgSlots; ++t) fprintf(stderr, " %lld", h.guard_slot[t]);
      fprintf(stderr, "\n");
    }
    if (c->gb_active) for (int t = 0; t < kArgSlots; ++t) h.n_guard += h.guard_slot[t];
    else h.n_guard = 0;
    return SBO_OK;
  }
  double* buf = (double*)c->xch.p + 64;
  hipLaunchKernelGGL(k_pack_c3, dim3(1), dim3(256), 0, c->stream, (const SweepScalars*)sc, buf, c->world, c->rank);
  int rc;
  if ((rc = comm_allreduce_sum_f64(c, buf, c->world * kC3Row))) return rc;
  std::vector<double> rows((size_t)c->world * kC3Row);
  SBO_HIP(hipMemcpyAsync(c->h_back, sc, kBack, hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipMemcpyAsync(rows.data(), buf, sizeof(double) * rows.size(), hipMemcpyDeviceToHost, c->stream));
  if (done_ev) SBO_HIP(hipEventRecord(done_ev, c->stream));
  SBO_HIP(stream_wait(c, c->stream));
  ++c->host_syncs;
  memset(&h, 0, sizeof(h));
  memcpy(&h, c->h_back, kScalHost);
  if (Lk) memcpy(Lk, Lk_pinned, sizeof(unsigned long long) * kMaxQ);
  c->c1_pending = false;                       // (the whole stream has drained)
  h.count_S = h.count_U = h.count_M = h.n_amb_total = h.n_guard = 0;
  h.halo_short = 0;                            // GLOBAL: a rank that alone reran its set phase would leave the others in a collective
  for (int t = 0; t < kMaxQ; ++t) h.count_set[t] = 0;
  Best merged[kArgSlots];
  for (int t = 0; t < kArgSlots; ++t) merged[t] = slot_is_max[t] ? best_none<true>() : best_none<false>();
  for (int r = 0; r < c->world; ++r) {
    const double* row = &rows[(size_t)r * kC3Row];
    for (int t = 0; t < kArgSlots; ++t) {
      const long long idx = (long long)row[kArgSlots + t];
      if (idx < 0) continue;
      const Best rb{row[t], idx, row[2 * kArgSlots + t], row[3 * kArgSlots + t], row[4 * kArgSlots + t], (long long)row[5 * kArgSlots + t]};
      merged[t] = slot_is_max[t] ? best_merge<true>(merged[t], rb) : best_merge<false>(merged[t], rb);
    }
    h.count_S += (long long)row[kC3Counts + 0];
    h.count_U += (long long)row[kC3Counts + 1];
    h.count_M += (long long)row[kC3Counts + 2];
    h.n_amb_total += (long long)row[kC3Counts + 3];
    h.n_guard += (long long)row[kC3Counts + 4];
    for (int t = 0; t < kMaxQ; ++t) h.count_set[t] += (long long)row[kC3Counts + 5 + t];
    h.halo_short += (long long)row[kC3Halo];
  }
  static const bool dbg_guard = getenv("SBO_DEBUG_GUARD") != nullptr;
  if (dbg_guard && c->gb_active) {
    for (int r = 0; r < c->world; ++r) fprintf(stderr, "[guard] rank %d row: counted %g\n", r, rows[(size_t)r * kC3Row + kC3Counts + 4]);
  }
  for (int t = 0; t < kArgSlots; ++t) {
    h.arg_val[t] = merged[t].i >= 0 ? merged[t].v : 0.0;
    h.arg_idx[t] = merged[t].i;
    h.arg_d[t] = merged[t].d;
    const bool near = c->gb_active && (slot_is_max[t] ? arg_near<true>(merged[t]) : arg_near<false>(merged[t]));
    if (dbg_guard && near)
      fprintf(stderr, "[guard] slot %d near after the merge: v %.17g i %lld d %.3g e1 %.17g e2 %.17g ei %lld\n", t, merged[t].v, merged[t].i, merged[t].d,
              merged[t].e1, merged[t].e2, merged[t].ei);
    if (near) ++h.n_guard;
  }
  if (!c->gb_active) h.n_guard = 0;
  return SBO_OK;
}

static void sweep_times(sbo_ctx* c) {
  float te = 0;
  (void)hipEventElapsedTime(&te, c->ev[1], c->ev[4]);
  c->prof.set_phase_ms = te;
  c->prof.host_syncs = c->host_syncs;
  c->prof.comm_bytes = c->comm_bytes;
  c->prof.comm_calls = c->comm_calls;
  c->prof.halo_reruns = c->halo_reruns;
  double cms = c->comm_host_ms;
  for (int k = 0; k + 1 < c->comm_nev; k += 2) {
    float t = 0;
    if (hipEventElapsedTime(&t, c->comm_ev[k], c->comm_ev[k + 1]) == hipSuccess) cms += t;
  }
  c->prof.comm_ms = cms;
}
static void sweep_comm_reset(sbo_ctx* c) {
  if (c->in_halo_rerun) return;                // (the counters of the discarded pass stay in the sweep's record)
  c->halo_reruns = 0;
  c->host_syncs = 0;
  c->comm_bytes = 0;
  c->comm_calls = 0;
  c->comm_nev = 0;
  c->comm_host_ms = 0.0;
}

#include "sets_colpath.inc.hpp"

template <typename T>
static int sweep_safeopt_t(sbo_ctx* c, const sbo_sweep_opts* o, sbo_safeopt_result* res) {
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  int rc;
  sweep_comm_reset(c);
  SBO_HIP(hipEventRecord(c->ev[0], c->stream));
  const bool reuse = o->posterior_ready && c->posterior_valid;
  if ((rc = sweep_masks(c, o->b, !reuse))) return rc;
  // column path (sets_colpath.inc.hpp): a fresh posterior of a one-constraint fp64 model on one rank may deliver the classification
  // as column words (the GEMM posterior decides whether its launch qualifies: col_active)
  c->col_request = !reuse && std::is_same<T, double>::value && q == 2 && !multi_rank(c) && !c->rc_active && c->result_mirror && n > 0;
  const int lean = o->lean;
  c->col_lean = c->col_request ? (lean >= 2 ? 2 : (lean ? 1 : 0)) : 0;
  c->sweep_lean = (lean && q >= 2 && !reuse) ? 1 : 0;
  c->col_active = false;
  if (!reuse && (rc = sbo_posterior_enqueue_(c))) { c->col_request = false; return rc; }
  c->col_request = false;
  c->fuse_request = 0;
  c->lmax_defer = false;
  if (!c->k1_stop_attached) SBO_HIP(hipEventRecord(c->ev[1], c->stream));
  c->k1_stop_attached = false;
  if (!reuse && !c->rc_active && (rc = guard_audit_enqueue(c, (c->col_active && c->col_lean) ? 1 : 0))) return rc;
  SweepScalars h;
  unsigned long long Lk[kMaxQ];
  const bool colpath = c->col_active;
  if (colpath) {
    // (a lean sweep left the objective's mean / var unwritten where no later stage reads them: whoever wants the posterior re-runs K1)
    if (c->col_lean) c->posterior_valid = false;
    if ((rc = col_set_phase(c, o, h, Lk))) return rc;
  } else {
  MinimizerJob mj;
  const int nb = reduce_blocks(c);
  // one constraint on one rank: the exhaustive recheck of in-band candidates (almost never any) is launched only when the
  // result block says that something was listed -- one launch (~5 us) less on the common path
  long long plane_ = 1;
  for (int a = 0; a < c->cs.d - 1; ++a) plane_ *= c->cs.count[a];
  const bool lazy_exact = c->exact_lazy && q == 2 && !multi_rank(c) && !c->rc_active && c->result_mirror && n > 0 && c->cs.kind == 1 &&
                          c->cs.first % plane_ == 0 && n % plane_ == 0;      // (grids: lists decide every expander exhaustively)
  // partials of the q arg-max reductions side by side: merged by one launch at the end (k_safeopt_finals)
  const size_t pstride = partial_stride(nb);
  if ((rc = ensure(c->partial, pstride * (size_t)q))) return rc;
  unsigned char* pbase = (unsigned char*)c->partial.p;
  const bool lanes = lanes_on(c);
  {
    const bool defer = q >= 2 && !multi_rank(c) && c->set_fuse && n > 0 && !lanes;   // (lanes fork right behind the merge)
    if ((rc = sweep_common_front<T>(c, o, defer ? &mj.fin : nullptr))) return rc;
    if ((rc = sweep_exchange_front<T>(c, o, true))) return rc;
    // (single rank: the minimiser rides in the first constraint's k_set_mid; with ranks > 1 it is queued here, ahead of the
    // host's wait for the C1 keys)
    mj.pending = n > 0;
    mj.nb = nb;
    mj.partial = (Best*)pbase;
    if (q < 2 || multi_rank(c)) launch_minimizer<T>(c, o, &mj);
    if (c->phase_events) SBO_HIP(hipEventRecord(c->ev[2], c->stream));
    if (lanes && (rc = lanes_fork(c))) return rc;
    for (int cc = 1; cc < q; ++cc) {
      uint8_t* G = (uint8_t*)c->maskG.p + (size_t)(cc - 1) * n;
      LaneScope lane(c, lanes && ((cc - 1) & 1));              // constraints alternate between the two lanes
      if ((rc = expander_set<T>(c, o, cc, G, lane.on ? nullptr : &mj, lazy_exact))) return rc;
    }
    launch_minimizer<T>(c, o, &mj);
    if (lanes && (rc = lanes_join(c))) return rc;
  }
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  if (c->phase_events) SBO_HIP(hipEventRecord(c->ev[3], c->stream));
  if (n > 0 && q > 1) {
    hipLaunchKernelGGL((k_arg_masked_multi<T, true, ValArray<T>>), dim3((unsigned)nb, (unsigned)(q - 1)), dim3(256), 0, c->stream,
                       ValArray<T>{(const T*)c->var.p, 0.0}, (const uint8_t*)nullptr, (const uint8_t*)c->maskG.p, n, (long long)c->cs.first, pbase,
                       pstride, 1, gb_of(c));
  }
  const bool mirrored = !multi_rank(c) && c->result_mirror;
  hipExtLaunchKernelGGL(k_sweep_finals<true>, dim3((unsigned)q), dim3(256), 0, c->stream, nullptr, mirrored ? c->ev[4] : nullptr, 0,
                        (const unsigned char*)pbase, pstride, n > 0 ? nb : 0, sc,
                        lanes ? (const SweepScalars*)c->lane1.scal.p : (const SweepScalars*)nullptr,
                        mirrored ? c->h_back : (unsigned char*)nullptr, (const unsigned long long*)c->Lmax.p, gb_of(c) ? 1 : 0);
  SBO_HIP(hipGetLastError());
  bool is_max[kArgSlots];
  for (int t = 0; t < kArgSlots; ++t) is_max[t] = true;
  if ((rc = sweep_exchange_back(c, h, is_max, Lk, c->ev[4], mirrored))) return rc;
  if (lazy_exact && (h.n_amb > 0 || c->exact_lazy == 2)) {      // (2: always, the test of this path)
    // in-band candidates after all: their exhaustive recheck, then the expanders' arg-max and the finals once more
    const int lidx = o->reference_quirk_L_index ? q - 1 : 1;
    if ((rc = launch_exact_d<T>(c, o, 1, lidx, (uint8_t*)c->maskG.p))) return rc;
    hipLaunchKernelGGL((k_arg_masked_multi<T, true, ValArray<T>>), dim3((unsigned)nb, 1u), dim3(256), 0, c->stream,
                       ValArray<T>{(const T*)c->var.p, 0.0}, (const uint8_t*)nullptr, (const uint8_t*)c->maskG.p, n, (long long)c->cs.first, pbase,
                       pstride, 1, gb_of(c));
    hipLaunchKernelGGL(k_sweep_clear_slot, dim3(1), dim3(1), 0, c->stream, sc, 1);
    hipExtLaunchKernelGGL(k_sweep_finals<true>, dim3((unsigned)q), dim3(256), 0, c->stream, nullptr, c->ev[4], 0,
                          (const unsigned char*)pbase, pstride, nb, sc, (const SweepScalars*)nullptr, c->h_back,
                          (const unsigned long long*)c->Lmax.p, gb_of(c) ? 1 : 0);
    SBO_HIP(hipGetLastError());
    if ((rc = sweep_exchange_back(c, h, is_max, Lk, c->ev[4], true))) return rc;
  }
  if (multi_rank(c)) {
    if (h.halo_short) {
      // a speculative window was too narrow for this sweep's radii (every rank sees the same keys and the same guess): the set
      // phase again, this time waiting for the keys
      for (auto& g : c->halo_guess) g = -1;
      sbo_sweep_opts o2 = *o;
      o2.posterior_ready = 1;
      ++c->halo_reruns;
      c->in_halo_rerun = true;
      const int rr = sweep_safeopt_t<T>(c, &o2, res);
      c->in_halo_rerun = false;
      return rr;
    }
    halo_learn(c, h, Lk, o->reference_quirk_L_index);
  }
  }      // (byte-mask path)
  c->masks_valid = true;
  c->last_sweep = 1;
  if (getenv("SBO_DEBUG_SCAN")) fprintf(stderr, "[safebo] open candidates scanned (last constraint) %lld, exact rechecks %lld\n", h.n_scan, h.n_amb_total);

  float t01 = 0, t12 = 0, t23 = 0, t34 = 0, t04 = 0;
  SBO_HIP(hipEventElapsedTime(&t01, c->ev[0], c->ev[1]));
  if (c->phase_events) {
    SBO_HIP(hipEventElapsedTime(&t12, c->ev[1], c->ev[2]));
    SBO_HIP(hipEventElapsedTime(&t23, c->ev[2], c->ev[3]));
    SBO_HIP(hipEventElapsedTime(&t34, c->ev[3], c->ev[4]));
  }
  SBO_HIP(hipEventElapsedTime(&t04, c->ev[0], c->ev[4]));
  memset(&c->prof, 0, sizeof(c->prof));
  c->prof.posterior_ms = t01;
  c->prof.classify_ms = t12;
  c->prof.expander_ms = t23;
  c->prof.argreduce_ms = t34;
  c->prof.total_ms = t04;
  c->prof.candidates = n;
  c->prof.posterior_launches = (!reuse && n > 0) ? 1 : 0;
  const double nn = c->mc.n, dd = c->mc.d;
  c->prof.posterior_flops = reuse ? 0.0 : q * (nn * nn + (2 * dd + 10) * nn) * (double)n;
  sweep_times(c);
  c->prof.set_path = colpath ? 1 : 0;

  memset(res, 0, sizeof(*res));
  res->count_S = h.count_S;
  res->count_U = h.count_U;
  res->count_M = h.count_M;
  res->n_exact_rechecks = h.n_amb_total;
  res->guard_band = h.n_guard;
  c->guard_first = h.n_guard;
  for (int i = 0; i < q; ++i) memcpy(&res->L[i], &Lk[i], 8);
  // (a lean sweep does not report the objective's key: no sweep of the reference reads it -- models/SafeOpt.py:110, GoOSE.py:100 --
  // and K1t / K1i leave its gradient quantities out; zero whichever kernel ran)
  if (o->lean && q >= 2) res->L[0] = 0.0;
  res->minimizer_index = -1;
  res->expander_index = -1;
  for (int cc = 1; cc < q; ++cc) res->expander_index_c[cc - 1] = -1;
  if (h.count_S == 0) return fail(SBO_E_EMPTY_SAFE_SET, "safe set S_t is empty on this candidate set");
  res->u_star = ord_val(h.ustar_key);
  res->minimizer_index = h.arg_idx[0];
  res->minimizer_std = std::sqrt(h.arg_val[0]);                 // models/SafeOpt.py:66
  coords_of(c, res->minimizer_index, res->minimizer_x);
  double best_std = 0.0;
  int best_c = 0;
  for (int cc = 1; cc < q; ++cc) {
    res->count_G[cc - 1] = h.count_set[cc - 1];
    res->expander_index_c[cc - 1] = h.arg_idx[cc];
    res->expander_std_c[cc - 1] = h.arg_idx[cc] >= 0 ? std::sqrt(h.arg_val[cc]) : 0.0;
    // max(std_expanders) / .index(max_std): first maximum wins (models/SafeOpt.py:119-121)
    if (h.arg_idx[cc] >= 0 && (best_c == 0 || res->expander_std_c[cc - 1] > best_std)) {
      best_std = res->expander_std_c[cc - 1];
      best_c = cc;
    }
  }
  res->expander_best_c = best_c;
  if (best_c) {
    res->expander_index = res->expander_index_c[best_c - 1];
    res->expander_std = best_std;
    coords_of(c, res->expander_index, res->expander_x);
  }
  res->choose_minimizer = res->minimizer_std > res->expander_std;   // test/test_SafeOpt.py:153
  if (gb_of(c)) {
    // the choices among the reductions' winners are decisions too: which constraint's expander is kept (largest var_0), and
    // minimiser against expander (std_min > std_exp) -- var_0 is known to +- arg_d
    long long near = 0;
    // (one candidate that wins two reductions ties with itself under any posterior: those comparisons are settled)
    for (int cc = 1; cc < q && best_c; ++cc)
      if (cc != best_c && h.arg_idx[cc] >= 0 && h.arg_idx[cc] != h.arg_idx[best_c] &&
          !(h.arg_val[cc] + h.arg_d[cc] < h.arg_val[best_c] - h.arg_d[best_c])) ++near;
    if (best_c && h.arg_idx[0] >= 0 && h.arg_idx[0] != h.arg_idx[best_c] &&
        std::fabs(h.arg_val[0] - h.arg_val[best_c]) <= h.arg_d[0] + h.arg_d[best_c]) ++near;
    res->guard_band += near;
    c->guard_first += near;
  }
  return SBO_OK;
}

template <typename T>
static int sweep_goose_t(sbo_ctx* c, const sbo_sweep_opts* o, sbo_goose_result* res);
template <typename T>
static int sweep_tr_t(sbo_ctx* c, const sbo_sweep_opts* o, const double* x0, double r, sbo_tr_result* res);
#include "sets_recheck.inc.hpp"

template <typename T, int D>
static int goose_sets(sbo_ctx* c, const sbo_sweep_opts* o, int cidx, const uint8_t* src, uint8_t* O) {
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  const int lidx = o->reference_quirk_L_index ? q - 1 : cidx;   // models/GoOSE.py:100 (loop-leaked i)
  const T* mean_c = (const T*)c->mean.p + (size_t)cidx * n;
  const T* var_c = (const T*)c->var.p + (size_t)cidx * n;
  int rc;
  long long maxlocal = n;
  for (int r = 0; r < c->world && multi_rank(c); ++r) maxlocal = std::max(maxlocal, c->first_of[r + 1] - c->first_of[r]);
  if ((rc = ensure(c->gw, sizeof(T) * (size_t)std::max<long long>(maxlocal, 1)))) return rc;
  if (n > 0)
    hipLaunchKernelGGL((k_goose_weights<T>), dim3(reduce_blocks(c)), dim3(256), 0, c->stream, mean_c, var_c, n, (T)o->b, src,
                       (T*)c->gw.p, (SweepScalars*)c->scal.p);
  CandSpec css = c->cs;                 // the source candidates
  const T* W = (const T*)c->gw.p;
  long long run_lo = 0, run_hi = (n + kRun - 1) / kRun;
  long long win_p0 = 0, win_p1 = 0;     // ranks > 1: window of hyper-planes holding every source that can matter
  bool w_is_window = false;             // ranks > 1, slab exchange: W holds exactly the window [win_p0, win_p1)
  if (multi_rank(c)) {
    // Sources of the other ranks that can reach this shard lie within H hyper-planes of it (H from the largest source
    // radius, keys of collective C1 -- exact, as for the expanders).
    const int d = c->cs.d;
    long long plane = 1;
    for (int a = 0; a < d - 1; ++a) plane *= c->cs.count[a];
    const long long planes_total = c->cs.count[d - 1];
    long long p0 = c->cs.first / plane, p1 = (c->cs.first + n + plane - 1) / plane;
    const long long own0 = p0, own1 = p1;
    const double hl = c->cs.step[d - 1];
    long long H;
    if ((rc = halo_for(c, cidx, lidx, hl, planes_total, &H))) return rc;
    p0 = std::max(0ll, p0 - H);
    p1 = std::min(planes_total, p1 + H);
    win_p0 = p0;
    win_p1 = p1;
    long long min_planes = planes_total;
    for (int r = 0; r < c->world; ++r) min_planes = std::min(min_planes, (c->first_of[r + 1] - c->first_of[r]) / plane);
    if (getenv("SBO_DEBUG_COMM"))
      fprintf(stderr, "[rank %d] GoOSE sources c=%d: halo %lld planes, smallest shard %lld planes -> %s\n", c->rank, cidx, H, min_planes,
              (c->sharded && H <= min_planes) ? "slab exchange" : "full all-gather");
    if (c->sharded && H <= min_planes) {
      // the halo fits inside the neighbours: every rank contributes only its first and last H planes (2 H plane values
      // instead of its whole shard), and the window is assembled from the previous rank's top slab, the own shard and
      // the next rank's bottom slab
      const size_t slab = (size_t)H * plane;
      if ((rc = ensure(c->gather, sizeof(T) * slab * 2 * (c->world + 1)))) return rc;
      if ((rc = ensure(c->Wfull, sizeof(T) * (size_t)(p1 - p0) * plane))) return rc;
      T* send = (T*)c->gather.p;
      T* recv = send + 2 * slab;
      SBO_HIP(hipMemcpyAsync(send, c->gw.p, sizeof(T) * slab, hipMemcpyDeviceToDevice, c->stream));
      SBO_HIP(hipMemcpyAsync(send + slab, (const T*)c->gw.p + (size_t)n - slab, sizeof(T) * slab, hipMemcpyDeviceToDevice, c->stream));
      if ((rc = comm_allgather_bytes(c, send, recv, sizeof(T) * slab * 2))) return rc;
      T* win = (T*)c->Wfull.p;
      size_t at = 0;
      if (own0 > p0) {       // previous rank's top slab (p0 = own0 - H exactly, since H <= its planes)
        SBO_HIP(hipMemcpyAsync(win, recv + (size_t)(c->rank - 1) * 2 * slab + slab, sizeof(T) * slab, hipMemcpyDeviceToDevice, c->stream));
        at = slab;
      }
      SBO_HIP(hipMemcpyAsync(win + at, c->gw.p, sizeof(T) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
      at += (size_t)n;
      if (p1 > own1)
        SBO_HIP(hipMemcpyAsync(win + at, recv + (size_t)(c->rank + 1) * 2 * slab, sizeof(T) * slab, hipMemcpyDeviceToDevice, c->stream));
      css.first = p0 * plane;
      css.n_local = (p1 - p0) * plane;
      W = (const T*)win;
      w_is_window = true;
      run_lo = 0;
      run_hi = (css.n_local + kRun - 1) / kRun;
    } else {
      // wide halo: all-gather the whole weight shards
      if ((rc = ensure(c->gather, sizeof(T) * (size_t)maxlocal * c->world))) return rc;
      if ((rc = ensure(c->Wfull, sizeof(T) * (size_t)c->grid_total))) return rc;
      if ((rc = comm_allgather_bytes(c, c->gw.p, c->gather.p, sizeof(T) * (size_t)maxlocal))) return rc;
      hipLaunchKernelGGL(k_compact_shards<T>, dim3((unsigned)std::min<long long>((c->grid_total + 255) / 256, 1 << 16)), dim3(256), 0,
                         c->stream, (const T*)c->gather.p, maxlocal, c->world, (const long long*)c->shard_first.p, c->grid_total,
                         (T*)c->Wfull.p);
      css.first = 0;
      css.n_local = c->grid_total;
      W = (const T*)c->Wfull.p;
      run_lo = p0 * plane / kRun;
      run_hi = (p1 * plane + kRun - 1) / kRun;
    }
  }
  if (n == 0) return SBO_OK;           // (an empty shard still took part in the all-gather)
  long long plane1 = 1;
  for (int a = 0; a < c->cs.d - 1; ++a) plane1 *= c->cs.count[a];
  const bool plane_aligned = c->cs.kind == 1 && c->cs.first % plane1 == 0 && n % plane1 == 0;
  if (plane_aligned && !c->goose_pairs) {
    // grids: power-distance transform of the source weights over the window, verdict by sign, exact recheck in the band
    const int d = c->cs.d;
    SweepScalars* sc = (SweepScalars*)c->scal.p;
    const long long own0 = c->cs.first / plane1;
    const long long w0 = multi_rank(c) ? win_p0 : own0, w1 = multi_rank(c) ? win_p1 : own0 + n / plane1;
    const long long wplanes = w1 - w0, nt = wplanes * plane1, goff = (own0 - w0) * plane1;
    const T* Wwin = (multi_rank(c) && !w_is_window) ? W + w0 * plane1 : W;
    if ((rc = ensure(c->dist2, sizeof(double) * (size_t)nt))) return rc;
    if (d > 2 && (rc = ensure(c->dist2b, sizeof(double) * (size_t)nt))) return rc;
    if ((rc = ensure(c->amb, sizeof(long long) * (size_t)n))) return rc;
    double xscale = 0.0;
    for (int a = 0; a < d; ++a) xscale = std::max(xscale, std::max(std::fabs(c->cs.lo[a]), std::fabs(c->cs.hi[a])));
    const int count0 = d >= 2 ? (int)c->cs.count[0] : (int)nt;
    const unsigned gridn = (unsigned)std::min<long long>((nt + 255) / 256, 1 << 20);
    c->amb_clean = false;                          // (k_goose_weights cleared the counters)
    // coarse bounds of the window: decide most candidates (and skip their axis-0 scans) without touching the fine arrays
    CoarseGrid cg;
    memset(&cg, 0, sizeof(cg));
    cg.d = d;
    bool coarse_ok = d >= 2 && nt >= (1ll << 16);
    long long nc = 1;
    for (int a = 0; a < d; ++a) {
      cg.count[a] = a == d - 1 ? wplanes : c->cs.count[a];
      cg.ccount[a] = (cg.count[a] + kCoarse - 1) / kCoarse;
      nc *= cg.ccount[a];
      if (cg.count[a] < 4 * kCoarse) coarse_ok = false;
    }
    const double *pc_lo = nullptr, *pc_hi = nullptr;
    if (coarse_ok) {
      cg.enabled = 1;
      if ((rc = ensure(c->coarse, (size_t)nc * 5 * sizeof(double) + 64))) return rc;
      double* lo0 = (double*)c->coarse.p;
      double* lo1 = lo0 + nc;
      double* hi0 = lo1 + nc;
      double* hi1 = hi0 + nc;
      hipLaunchKernelGGL((k_pdt_cell_min<T>), dim3((unsigned)std::min<long long>((nc * 8 + 255) / 256, 1 << 18)), dim3(256), 0, c->stream, Wwin, cg, nc,
                         (const unsigned long long*)c->Lmax.p, lidx, lo0, hi0);
      long long cstride = 1;
      for (int a = 0; a < d; ++a) {
        hipLaunchKernelGGL(k_pdt_coarse_scan, dim3((unsigned)std::min<long long>((2 * nc * 8 + 255) / 256, 1 << 18)), dim3(256), 0, c->stream, (const double*)lo0, lo1, (const double*)hi0,
                           hi1, nc, cstride, (int)cg.ccount[a], c->cs.step[a], (const SweepScalars*)sc, cidx,
                           (const unsigned long long*)c->Lmax.p, lidx, d, xscale);
        std::swap(lo0, lo1);
        std::swap(hi0, hi1);
        cstride *= cg.ccount[a];
      }
      pc_lo = lo0;
      pc_hi = hi0;
    }
    const T* wbmax = nullptr;
    const int blk0 = 32;
    if (c->scan_blocks && count0 >= 16 * blk0 && count0 <= 8192 && nt / count0 >= 1024) {
      // one workgroup per line, the line in LDS (pays once there are enough lines to fill the chip: from 1024 lines on --
      // config C's 1024 x 1024 grid of the Williams-Otto plant: 33 -> 19 us per constraint against the thread-per-position kernel)
      const size_t lds = sizeof(double) * ((size_t)count0 + (count0 + kAnchor - 1) / kAnchor) + sizeof(int) * 2 * ((size_t)(count0 + kAnchor - 1) / kAnchor + 2);
      SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pdt_axis0_lds<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((k_pdt_axis0_lds<T>), dim3((unsigned)std::min<long long>(nt / count0, 1 << 16)), dim3(256), lds, c->stream, Wwin,
                         nt / count0, count0, c->cs.step[0], (const SweepScalars*)sc, cidx, (const unsigned long long*)c->Lmax.p, lidx, d,
                         xscale, cg, pc_lo, blk0, (double*)c->dist2.p);
    } else {
      if (c->scan_blocks && count0 >= 16 * blk0) {
        const long long nbw = (nt / count0) * ((count0 + blk0 - 1) / blk0);
        if ((rc = ensure(c->blockmax, sizeof(T) * (size_t)nbw))) return rc;
        hipLaunchKernelGGL((k_block_max_w<T>), dim3((unsigned)std::min<long long>((nbw + 255) / 256, 1 << 20)), dim3(256), 0, c->stream,
                           Wwin, nt, count0, blk0, (T*)c->blockmax.p);
        wbmax = (const T*)c->blockmax.p;
      }
      hipLaunchKernelGGL((k_pdt_axis0<T>), dim3(gridn), dim3(256), 0, c->stream, Wwin, nt, count0, c->cs.step[0],
                         (const SweepScalars*)sc, cidx, (const unsigned long long*)c->Lmax.p, lidx, d, xscale, cg, pc_lo, wbmax,
                         blk0, (double*)c->dist2.p);
    }
    double* pin = (double*)c->dist2.p;
    double* pout = (double*)c->dist2b.p;
    long long stride = count0;
    for (int a = 1; a < d - 1; ++a) {
      hipLaunchKernelGGL(k_pdt_scan, dim3(gridn), dim3(256), 0, c->stream, (const double*)pin, pout, nt, stride,
                         (int)c->cs.count[a], c->cs.step[a], (const SweepScalars*)sc, cidx,
                         (const unsigned long long*)c->Lmax.p, lidx, d, xscale);
      std::swap(pin, pout);
      stride *= c->cs.count[a];
    }
    const int last_cnt = d >= 2 ? (int)wplanes : 1;
    const double last_h = d >= 2 ? c->cs.step[d - 1] : 0.0;
    const double* bmin = nullptr;
    const int blk = last_cnt >= 8192 ? 64 : 32;
    if (d >= 2 && last_cnt >= 4 * blk && c->scan_blocks) {
      const long long nb_ = (long long)((last_cnt + blk - 1) / blk) * stride;
      if ((rc = ensure(c->blockmin, sizeof(double) * (size_t)nb_))) return rc;
      hipLaunchKernelGGL(k_block_min, dim3((unsigned)std::min<long long>((stride + 63) / 64 * ((last_cnt + blk - 1) / blk), 1 << 20)), dim3(256), 0, c->stream,
                         (const double*)pin, stride, last_cnt, blk, (double*)c->blockmin.p);
      bmin = (const double*)c->blockmin.p;
    }
    const long long len0 = d >= 2 ? count0 : n, nl = n / len0;    // positions per line / local lines
    long long* slist = nullptr;                    // open points of the verdict: listed and scanned by groups of lanes
    if (bmin && blk <= 64 && c->scan_waves) {
      if ((rc = ensure(c->scanlist, sizeof(long long) * (size_t)n))) return rc;
      slist = (long long*)c->scanlist.p;
    }
    hipLaunchKernelGGL(k_pdt_decide, dim3((unsigned)((len0 + 255) / 256), (unsigned)std::min<long long>((nl + kDecideLines - 1) / kDecideLines, 65535)),
                       dim3(256), 0, c->stream, (const double*)pin, nl, (int)len0, goff / len0, goff, d >= 2 ? stride : 1, last_cnt, last_h, d,
                       xscale, (const uint8_t*)c->maskU.p, (const unsigned long long*)c->Lmax.p, lidx, sc, cidx, O, (long long*)c->amb.p, cg,
                       pc_lo, pc_hi, bmin, blk, slist);
    if (slist) {
      if (c->scan_waves == 32 || c->scan_waves == 64)
        hipLaunchKernelGGL((k_pdt_scan_list<32>), dim3(2048), dim3(256), 0, c->stream, (const double*)pin, goff, stride, last_cnt, last_h, d,
                           xscale, (const unsigned long long*)c->Lmax.p, lidx, sc, cidx, O, (long long*)c->amb.p, bmin, blk,
                           (const long long*)slist);
      else
        hipLaunchKernelGGL((k_pdt_scan_list<16>), dim3(2048), dim3(256), 0, c->stream, (const double*)pin, goff, stride, last_cnt, last_h, d,
                           xscale, (const unsigned long long*)c->Lmax.p, lidx, sc, cidx, O, (long long*)c->amb.p, bmin, blk,
                           (const long long*)slist);
    }
    hipLaunchKernelGGL((k_goose_exact<T, D>), dim3(1024), dim3(256), 0, c->stream, c->cs, css, W,
                       (const unsigned long long*)c->Lmax.p, lidx, sc, cidx, (const long long*)c->amb.p, O,
                       (gb_of(c) && !c->rc_active && !c->gb_slow) ? 1 : 0);
    SBO_HIP(hipGetLastError());
    return SBO_OK;
  }
  // explicit lists and ragged grid ranges: pruned exact pair evaluation over runs of 256 candidates
  const long long nsrc_runs = run_hi - run_lo;
  if (nsrc_runs > 0x7fffffffll) return fail(SBO_E_UNSUPPORTED, "too many source runs");
  if ((rc = ensure(c->runmeta, sizeof(RunMeta) * (size_t)std::max<long long>(nsrc_runs, 1)))) return rc;
  hipLaunchKernelGGL((k_goose_run_meta<T, D>), dim3((unsigned)nsrc_runs), dim3(256), 0, c->stream, css, W,
                     (const unsigned long long*)c->Lmax.p, lidx, run_lo, (RunMeta*)c->runmeta.p);
  hipLaunchKernelGGL((k_goose_optimistic<T, D>), dim3((unsigned)((n + kRun - 1) / kRun)), dim3(256), 0, c->stream, c->cs, css, W,
                     (const uint8_t*)c->maskU.p, (const unsigned long long*)c->Lmax.p, lidx, (const RunMeta*)c->runmeta.p,
                     run_lo, (int)nsrc_runs, O);
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

template <typename T>
static int goose_sets_d(sbo_ctx* c, const sbo_sweep_opts* o, int cidx, const uint8_t* src, uint8_t* O) {
  switch (c->mc.dpad) {
    case 2: return goose_sets<T, 2>(c, o, cidx, src, O);
    case 4: return goose_sets<T, 4>(c, o, cidx, src, O);
    case 8: return goose_sets<T, 8>(c, o, cidx, src, O);
  }
  return fail(SBO_E_UNSUPPORTED, "unsupported padded dimension");
}

// arg-min over S_t of the distance to the target (explore_safeset, models/GoOSE.py:116-119) -> partials for k_arg_final
template <typename T>
static void launch_argmin_dist(sbo_ctx* c, const double* dev_target, int nb) {
  const long long n = c->cs.n_local;
  const uint8_t* S = (const uint8_t*)c->maskS.p;
  switch (c->mc.dpad) {
    case 2:
      hipLaunchKernelGGL((k_arg_masked<T, false, ValDist<T, 2>>), dim3(nb), dim3(256), 0, c->stream, ValDist<T, 2>{c->cs, dev_target}, S, n,
                         (long long)c->cs.first, (Best*)c->partial.p, (const GuardBand*)nullptr);
      break;
    case 4:
      hipLaunchKernelGGL((k_arg_masked<T, false, ValDist<T, 4>>), dim3(nb), dim3(256), 0, c->stream, ValDist<T, 4>{c->cs, dev_target}, S, n,
                         (long long)c->cs.first, (Best*)c->partial.p, (const GuardBand*)nullptr);
      break;
    default:
      hipLaunchKernelGGL((k_arg_masked<T, false, ValDist<T, 8>>), dim3(nb), dim3(256), 0, c->stream, ValDist<T, 8>{c->cs, dev_target}, S, n,
                         (long long)c->cs.first, (Best*)c->partial.p, (const GuardBand*)nullptr);
      break;
  }
}

// GoOSE iteration (models/GoOSE.py:63-119, test/test_GoOSE.py:151-162) on the resident candidates.  Ranks > 1: C1 + C2
// as for SafeOpt, one all-gather of the source weights per constraint, C3 for the arg-min slots and a second C3 for
// the explore step (every rank derives the same target from the merged slots).
template <typename T>
static int sweep_goose_t(sbo_ctx* c, const sbo_sweep_opts* o, sbo_goose_result* res) {
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  int rc;
  sweep_comm_reset(c);
  SBO_HIP(hipEventRecord(c->ev[0], c->stream));
  const bool reuse = o->posterior_ready && c->posterior_valid;
  if ((rc = sweep_masks(c, o->b, !reuse))) return rc;
  c->sweep_lean = 0;
  if (!reuse && (rc = sbo_posterior_enqueue_(c))) return rc;
  c->fuse_request = 0;
  c->lmax_defer = false;
  if (!c->k1_stop_attached) SBO_HIP(hipEventRecord(c->ev[1], c->stream));
  c->k1_stop_attached = false;
  if (!reuse && !c->rc_active && (rc = guard_audit_enqueue(c, 0))) return rc;
  const int nb = reduce_blocks(c);
  const bool lanes = lanes_on(c);
  {
    if ((rc = sweep_common_front<T>(c, o, nullptr))) return rc;
    if ((rc = sweep_exchange_front<T>(c, o, true))) return rc;
    if ((rc = ensure(c->maskO, (size_t)std::max<long long>(n, 1) * std::max(1, q - 1)))) return rc;
    if (c->phase_events) SBO_HIP(hipEventRecord(c->ev[2], c->stream));
    // Only expanders can cover an unsafe point: "g covers h" is the predicate that puts g into G_c.  So G_c is built
    // first (distance transform, cheap) and serves as the source set of the coverage search instead of all of S_t.
    if ((rc = ensure(c->maskG, (size_t)std::max<long long>(n, 1) * std::max(1, q - 1)))) return rc;
    if (lanes && (rc = lanes_fork(c))) return rc;
    for (int cc = 1; cc < q; ++cc) {
      uint8_t* G = (uint8_t*)c->maskG.p + (size_t)(cc - 1) * n;
      uint8_t* O = (uint8_t*)c->maskO.p + (size_t)(cc - 1) * n;
      LaneScope lane(c, lanes && ((cc - 1) & 1));              // constraints alternate between the two lanes
      // (a large explicit list has no transform to build G_c with: all of S_t stays the source set there)
      long long plane = 1;
      for (int a = 0; a < c->cs.d - 1; ++a) plane *= c->cs.count[a];
      const bool can_expand = n <= (1ll << 17) || (c->cs.kind == 1 && c->cs.first % plane == 0 && n % plane == 0);   // (larger lists: S_t is the source set)
      const uint8_t* src = (const uint8_t*)c->maskS.p;
      if (can_expand) {
        if ((rc = expander_set<T>(c, o, cc, G))) return rc;
        src = G;
      }
      if ((rc = goose_sets_d<T>(c, o, cc, src, O))) return rc;
    }
    if (lanes && (rc = lanes_join(c))) return rc;
  }
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  if (c->phase_events) SBO_HIP(hipEventRecord(c->ev[3], c->stream));
  // arg-min of lcb_0 over S_t and over every O_c: the bound is computed for the masked candidates only; one launch, the q
  // reductions side by side
  const ValLcb<T> lcb0{(const T*)c->mean.p, (const T*)c->var.p, (T)o->b, 0.0, 0.0};
  const size_t pstride = partial_stride(nb);     // q regions of partials, one merge launch
  if ((rc = ensure(c->partial, pstride * (size_t)q))) return rc;
  unsigned char* pbase = (unsigned char*)c->partial.p;
  if (n > 0)
    hipLaunchKernelGGL((k_arg_masked_multi<T, false, ValLcb<T>>), dim3((unsigned)nb, (unsigned)q), dim3(256), 0, c->stream, lcb0,
                       (const uint8_t*)c->maskS.p, (const uint8_t*)c->maskO.p, n, (long long)c->cs.first, pbase, pstride, 0, gb_of(c));
  c->lmax_pending = false;
  SweepScalars h;
  bool is_max[kArgSlots];
  for (int t = 0; t < kArgSlots; ++t) is_max[t] = false;
  // Single rank: the explore step (target choice, distances, arg-min over S) is enqueued before the read-back, one host
  // round trip per sweep; ranks > 1 need the merged target slots first and take a second one below.
  const bool fused_explore = !multi_rank(c) && q > 1;
  double* dev_t = (double*)c->scal.p + 256;
  // (one rank: the finals and the target choice in one launch, the last merge writes the host's block itself)
  const int gbon = gb_of(c) ? 1 : 0;
  const bool short_tail = fused_explore && c->result_mirror && (c->mc.dpad == 2 || c->mc.dpad == 4 || c->mc.dpad == 8);
  const SweepScalars* l1 = lanes ? (const SweepScalars*)c->lane1.scal.p : (const SweepScalars*)nullptr;
  if (short_tail) {
    switch (c->mc.dpad) {
      case 2: hipLaunchKernelGGL((k_goose_finals<2>), dim3((unsigned)q), dim3(256), 0, c->stream, (const unsigned char*)pbase, pstride, n > 0 ? nb : 0, q, sc, l1, c->cs, dev_t, gbon); break;
      case 4: hipLaunchKernelGGL((k_goose_finals<4>), dim3((unsigned)q), dim3(256), 0, c->stream, (const unsigned char*)pbase, pstride, n > 0 ? nb : 0, q, sc, l1, c->cs, dev_t, gbon); break;
      default: hipLaunchKernelGGL((k_goose_finals<8>), dim3((unsigned)q), dim3(256), 0, c->stream, (const unsigned char*)pbase, pstride, n > 0 ? nb : 0, q, sc, l1, c->cs, dev_t, gbon); break;
    }
    if (n > 0) launch_argmin_dist<T>(c, dev_t, nb);
    hipExtLaunchKernelGGL(k_arg_final_mirror, dim3(1), dim3(256), 0, c->stream, nullptr, c->ev[4], 0, (const Best*)c->partial.p, n > 0 ? nb : 0, sc,
                          kArgSlots - 1, c->h_back, (const unsigned long long*)c->Lmax.p, 0);
    SBO_HIP(hipGetLastError());
  } else {
    hipLaunchKernelGGL(k_sweep_finals<false>, dim3((unsigned)q), dim3(256), 0, c->stream, (const unsigned char*)pbase, pstride,
                       n > 0 ? nb : 0, sc, l1, (unsigned char*)nullptr, (const unsigned long long*)nullptr, gbon);
    SBO_HIP(hipGetLastError());
    if (fused_explore) {
      switch (c->mc.dpad) {
        case 2: hipLaunchKernelGGL((k_pick_target<2>), dim3(1), dim3(1), 0, c->stream, c->cs, (const SweepScalars*)sc, q, dev_t); break;
        case 4: hipLaunchKernelGGL((k_pick_target<4>), dim3(1), dim3(1), 0, c->stream, c->cs, (const SweepScalars*)sc, q, dev_t); break;
        default: hipLaunchKernelGGL((k_pick_target<8>), dim3(1), dim3(1), 0, c->stream, c->cs, (const SweepScalars*)sc, q, dev_t); break;
      }
      if (n > 0) {
        launch_argmin_dist<T>(c, dev_t, nb);
      }
      hipLaunchKernelGGL((k_arg_final<false>), dim3(1), dim3(256), 0, c->stream, (const Best*)c->partial.p, n > 0 ? nb : 0, sc,
                         kArgSlots - 1, (long long*)nullptr, 0);
    }
  }
  unsigned long long Lk[kMaxQ];
  if ((rc = sweep_exchange_back(c, h, is_max, Lk, short_tail ? c->ev[4] : nullptr, short_tail))) return rc;
  if (multi_rank(c)) {
    if (h.halo_short) {                     // (see sweep_safeopt_t)
      for (auto& g : c->halo_guess) g = -1;
      sbo_sweep_opts o2 = *o;
      o2.posterior_ready = 1;
      ++c->halo_reruns;
      c->in_halo_rerun = true;
      const int rr = sweep_goose_t<T>(c, &o2, res);
      c->in_halo_rerun = false;
      return rr;
    }
    halo_learn(c, h, Lk, o->reference_quirk_L_index);
  }
  c->masks_valid = true;
  c->last_sweep = 2;

  memset(res, 0, sizeof(*res));
  res->count_S = h.count_S;
  res->count_U = h.count_U;
  res->n_exact_rechecks = h.n_amb_total;
  res->guard_band = h.n_guard;
  c->guard_first = h.n_guard;
  for (int i = 0; i < q; ++i) memcpy(&res->L[i], &Lk[i], 8);
  res->safe_min_index = res->target_index = res->explore_index = -1;
  for (int cc = 1; cc < q; ++cc) res->target_index_c[cc - 1] = -1;
  if (h.count_S == 0) return fail(SBO_E_EMPTY_SAFE_SET, "safe set S_t is empty on this candidate set");
  res->safe_min_index = h.arg_idx[0];
  res->safe_min_lcb = h.arg_val[0];
  coords_of(c, res->safe_min_index, res->safe_min_x);
  int best_c = 0;
  double best_lcb = 0.0;
  for (int cc = 1; cc < q; ++cc) {
    res->count_O[cc - 1] = h.count_set[cc - 1];
    res->target_index_c[cc - 1] = h.arg_idx[cc];
    res->target_lcb_c[cc - 1] = h.arg_idx[cc] >= 0 ? h.arg_val[cc] : INFINITY;
    // min(lcb_target) / .index(min): first minimum wins (models/GoOSE.py:110-112)
    if (h.arg_idx[cc] >= 0 && (best_c == 0 || h.arg_val[cc] < best_lcb)) {
      best_c = cc;
      best_lcb = h.arg_val[cc];
    }
  }
  res->target_best_c = best_c;
  res->target_lcb = best_c ? best_lcb : INFINITY;
  res->choose_safe_min = best_c ? (res->safe_min_lcb <= res->target_lcb) : 1;   // test/test_GoOSE.py:158
  if (gb_of(c) && best_c) {
    // choices among the winners (the short tail judged the targets' one on the device): which constraint's target, and safe
    // minimum against target (min_safe_lcb <= target_lcb) -- lcb_0 is known to +- arg_d
    long long near = 0;
    for (int cc = 1; cc < q && !short_tail; ++cc)
      if (cc != best_c && h.arg_idx[cc] >= 0 && h.arg_idx[cc] != h.arg_idx[best_c] &&
          !(h.arg_val[cc] - h.arg_d[cc] > h.arg_val[best_c] + h.arg_d[best_c])) ++near;
    if (h.arg_idx[0] >= 0 && std::fabs(h.arg_val[0] - h.arg_val[best_c]) <= h.arg_d[0] + h.arg_d[best_c]) ++near;
    res->guard_band += near;
    c->guard_first += near;
  }
  if (best_c) {
    res->target_index = res->target_index_c[best_c - 1];
    coords_of(c, res->target_index, res->target_x);
    // explore_safeset(target): argmin_{S} ||x - target||_2 (models/GoOSE.py:116-119)
    if (fused_explore) {
      res->explore_index = h.arg_idx[kArgSlots - 1];
      coords_of(c, res->explore_index, res->explore_x);
    } else {
    SBO_HIP(hipMemcpyAsync(dev_t, res->target_x, sizeof(double) * SBO_MAX_D, hipMemcpyHostToDevice, c->stream));
    if (n > 0) {
      launch_argmin_dist<T>(c, dev_t, nb);
    }
    hipLaunchKernelGGL((k_arg_final<false>), dim3(1), dim3(256), 0, c->stream, (const Best*)c->partial.p, n > 0 ? nb : 0, sc,
                       kArgSlots - 1, (long long*)nullptr, 0);
    SweepScalars h2;
    if ((rc = sweep_exchange_back(c, h2, is_max))) return rc;
    res->explore_index = h2.arg_idx[kArgSlots - 1];
    coords_of(c, res->explore_index, res->explore_x);
    }
  }
  if (!short_tail) {                       // (short tail: the end event rode on the last merge and has been waited for)
    SBO_HIP(hipEventRecord(c->ev[4], c->stream));
    SBO_HIP(hipEventSynchronize(c->ev[4]));
  }
  float t01 = 0, t12 = 0, t23 = 0, t34 = 0, t04 = 0;
  SBO_HIP(hipEventElapsedTime(&t01, c->ev[0], c->ev[1]));
  if (c->phase_events) {
    SBO_HIP(hipEventElapsedTime(&t12, c->ev[1], c->ev[2]));
    SBO_HIP(hipEventElapsedTime(&t23, c->ev[2], c->ev[3]));
    SBO_HIP(hipEventElapsedTime(&t34, c->ev[3], c->ev[4]));
  }
  SBO_HIP(hipEventElapsedTime(&t04, c->ev[0], c->ev[4]));
  memset(&c->prof, 0, sizeof(c->prof));
  c->prof.posterior_ms = t01;
  c->prof.classify_ms = t12;
  c->prof.expander_ms = t23;
  c->prof.argreduce_ms = t34;
  c->prof.total_ms = t04;
  c->prof.candidates = n;
  c->prof.posterior_launches = (!reuse && n > 0) ? 1 : 0;
  const double nn = c->mc.n, dd = c->mc.d;
  c->prof.posterior_flops = reuse ? 0.0 : q * (nn * nn + (2 * dd + 10) * nn) * (double)n;
  sweep_times(c);
  return SBO_OK;
}

// Trust-region acquisition (models/GP_TR.py:43-51): argmin lcb_0 over S and the ball
template <typename T>
static int sweep_tr_t(sbo_ctx* c, const sbo_sweep_opts* o, const double* x0, double r, sbo_tr_result* res) {
  const long long n = c->cs.n_local;
  int rc;
  SBO_HIP(hipEventRecord(c->ev[0], c->stream));
  const bool reuse = o->posterior_ready && c->posterior_valid;
  if ((rc = sweep_masks(c, o->b, !reuse))) return rc;
  c->sweep_lean = 0;
  if (!reuse && (rc = sbo_posterior_enqueue_(c))) return rc;
  c->fuse_request = 0;
  c->lmax_defer = false;
  if (!c->k1_stop_attached) SBO_HIP(hipEventRecord(c->ev[1], c->stream));
  c->k1_stop_attached = false;
  if (!reuse && !c->rc_active && (rc = guard_audit_enqueue(c, 0))) return rc;
  if ((rc = sweep_common_front<T>(c, o))) return rc;
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  const int nb = reduce_blocks(c);
  double* dev_x0 = (double*)c->scal.p + 256;
  SBO_HIP(hipMemcpyAsync(dev_x0, x0, sizeof(double) * c->cs.d, hipMemcpyHostToDevice, c->stream));
  if (n > 0) {
    switch (c->mc.dpad) {
      case 2: hipLaunchKernelGGL((k_ball_mask<2>), dim3(nb), dim3(256), 0, c->stream, c->cs, n, (const uint8_t*)c->maskS.p, (const double*)dev_x0, r, (uint8_t*)c->maskM.p); break;
      case 4: hipLaunchKernelGGL((k_ball_mask<4>), dim3(nb), dim3(256), 0, c->stream, c->cs, n, (const uint8_t*)c->maskS.p, (const double*)dev_x0, r, (uint8_t*)c->maskM.p); break;
      default: hipLaunchKernelGGL((k_ball_mask<8>), dim3(nb), dim3(256), 0, c->stream, c->cs, n, (const uint8_t*)c->maskS.p, (const double*)dev_x0, r, (uint8_t*)c->maskM.p); break;
    }
    hipLaunchKernelGGL((k_arg_masked<T, false, ValLcb<T>>), dim3(nb), dim3(256), 0, c->stream,
                       ValLcb<T>{(const T*)c->mean.p, (const T*)c->var.p, (T)o->b, 0.0, 0.0}, (const uint8_t*)c->maskM.p, n,
                       (long long)c->cs.first, (Best*)c->partial.p, gb_of(c));
  }
  hipLaunchKernelGGL((k_arg_final<false>), dim3(1), dim3(256), 0, c->stream, (const Best*)c->partial.p, n > 0 ? nb : 0, sc, 0, &sc->count_M, gb_of(c) ? 1 : 0);
  SBO_HIP(hipGetLastError());
  SweepScalars h;
  bool is_max[kArgSlots];
  for (int t = 0; t < kArgSlots; ++t) is_max[t] = false;
  if (multi_rank(c)) {
    if ((rc = ensure(c->xch, sizeof(double) * (size_t)(c->world * kC3Row + 64)))) return rc;
  }
  if ((rc = sweep_exchange_back(c, h, is_max))) return rc;
  SBO_HIP(hipEventRecord(c->ev[2], c->stream));
  SBO_HIP(hipEventSynchronize(c->ev[2]));
  c->masks_valid = true;
  c->last_sweep = 3;
  float t01 = 0, t12 = 0, t02 = 0;
  SBO_HIP(hipEventElapsedTime(&t01, c->ev[0], c->ev[1]));
  SBO_HIP(hipEventElapsedTime(&t12, c->ev[1], c->ev[2]));
  SBO_HIP(hipEventElapsedTime(&t02, c->ev[0], c->ev[2]));
  memset(&c->prof, 0, sizeof(c->prof));
  c->prof.posterior_ms = t01;
  c->prof.classify_ms = t12;
  c->prof.total_ms = t02;
  c->prof.candidates = n;
  c->prof.posterior_launches = (!reuse && n > 0) ? 1 : 0;
  const double nn = c->mc.n, dd = c->mc.d;
  c->prof.posterior_flops = reuse ? 0.0 : c->mc.q * (nn * nn + (2 * dd + 10) * nn) * (double)n;
  memset(res, 0, sizeof(*res));
  res->guard_band = h.n_guard;
  c->guard_first = h.n_guard;
  res->count_S = h.count_S;
  res->count_T = h.count_M;
  res->index = h.arg_idx[0];
  res->lcb = h.arg_idx[0] >= 0 ? h.arg_val[0] : INFINITY;
  coords_of(c, res->index, res->x);
  return SBO_OK;
}

}  // namespace sbo

using namespace sbo;

extern "C" {

int sbo_posterior_enqueue(sbo_ctx* c);
}
namespace sbo {
int sbo_posterior_enqueue_(sbo_ctx* c) { return sbo_posterior_enqueue(c); }
}

extern "C" {

int sbo_sweep_safeopt(sbo_ctx* c, const sbo_sweep_opts* opts, sbo_safeopt_result* result) {
  if (!c || !opts || !result) return fail(SBO_E_INVALID, "NULL argument");
  if (!c->has_model) return fail(SBO_E_NO_MODEL, "sbo_model_set has not been called");
  if (!c->has_cand) return fail(SBO_E_NO_CANDIDATES, "no candidates resident");
  if (c->cs.d != c->mc.d) return fail(SBO_E_INVALID, "ERROR W and X_norm dimension should be same");
  SBO_HIP(hipSetDevice(c->device));
  if (!(opts->b >= 0.0) || !std::isfinite(opts->b)) return fail(SBO_E_INVALID, "confidence multiplier b must be finite and >= 0");
  int rc;
  c->guard_first = 0;
  if (c->dtype == SBO_F32 && c->fp64_recheck && c->shadow && c->shadow->has_model)
    rc = sweep_safeopt_recheck<float>(c, opts, result);
  else
    rc = c->dtype == SBO_F64 ? sweep_safeopt_t<double>(c, opts, result) : sweep_safeopt_t<float>(c, opts, result);
  // an approximating posterior (K1b / K1t) whose band left decisions of that pass open: re-evaluate exactly, decide again
  if ((rc == SBO_OK || rc == SBO_E_EMPTY_SAFE_SET) && c->dtype == SBO_F64 && gb_of(c) && (c->guard_first > 0 || c->guard_band == 2)) {
    const long long first = c->guard_first;
    rc = sweep_safeopt_recheck<double>(c, opts, result);
    if (rc == SBO_OK || rc == SBO_E_EMPTY_SAFE_SET) result->guard_band = first;
  }
  if (rc != SBO_OK && rc != SBO_E_EMPTY_SAFE_SET) drain_streams(c);   // (kernels of the failed call may still sit on the side streams)
  return rc;
}

int sbo_sweep_goose(sbo_ctx* c, const sbo_sweep_opts* opts, sbo_goose_result* result) {
  if (!c || !opts || !result) return fail(SBO_E_INVALID, "NULL argument");
  if (!c->has_model) return fail(SBO_E_NO_MODEL, "sbo_model_set has not been called");
  if (!c->has_cand) return fail(SBO_E_NO_CANDIDATES, "no candidates resident");
  if (c->cs.d != c->mc.d) return fail(SBO_E_INVALID, "ERROR W and X_norm dimension should be same");
  SBO_HIP(hipSetDevice(c->device));
  if (!(opts->b >= 0.0) || !std::isfinite(opts->b)) return fail(SBO_E_INVALID, "confidence multiplier b must be finite and >= 0");
  int rc;
  c->guard_first = 0;
  if (c->dtype == SBO_F32 && c->fp64_recheck && c->shadow && c->shadow->has_model)
    rc = sweep_goose_recheck<float>(c, opts, result);
  else
    rc = c->dtype == SBO_F64 ? sweep_goose_t<double>(c, opts, result) : sweep_goose_t<float>(c, opts, result);
  if ((rc == SBO_OK || rc == SBO_E_EMPTY_SAFE_SET) && c->dtype == SBO_F64 && gb_of(c) && (c->guard_first > 0 || c->guard_band == 2)) {
    const long long first = c->guard_first;
    rc = sweep_goose_recheck<double>(c, opts, result);
    if (rc == SBO_OK || rc == SBO_E_EMPTY_SAFE_SET) result->guard_band = first;
  }
  if (rc != SBO_OK && rc != SBO_E_EMPTY_SAFE_SET) drain_streams(c);
  return rc;
}

int sbo_sweep_tr(sbo_ctx* c, const sbo_sweep_opts* opts, const double* x_0, double r, sbo_tr_result* result) {
  if (!c || !opts || !x_0 || !result) return fail(SBO_E_INVALID, "NULL argument");
  if (!c->has_model) return fail(SBO_E_NO_MODEL, "sbo_model_set has not been called");
  if (!c->has_cand) return fail(SBO_E_NO_CANDIDATES, "no candidates resident");
  if (c->cs.d != c->mc.d) return fail(SBO_E_INVALID, "ERROR W and X_norm dimension should be same");
  if (!(r >= 0.0)) return fail(SBO_E_INVALID, "trust-region radius must be >= 0");
  if (!(opts->b >= 0.0) || !std::isfinite(opts->b)) return fail(SBO_E_INVALID, "confidence multiplier b must be finite and >= 0");
  SBO_HIP(hipSetDevice(c->device));
  c->guard_first = 0;
  if (c->dtype == SBO_F32 && c->fp64_recheck && c->shadow && c->shadow->has_model)
    return sweep_tr_recheck<float>(c, opts, x_0, r, result);
  int rc = c->dtype == SBO_F64 ? sweep_tr_t<double>(c, opts, x_0, r, result) : sweep_tr_t<float>(c, opts, x_0, r, result);
  if (rc == SBO_OK && c->dtype == SBO_F64 && gb_of(c) && (c->guard_first > 0 || c->guard_band == 2)) {
    const long long first = c->guard_first;
    rc = sweep_tr_recheck<double>(c, opts, x_0, r, result);
    if (rc == SBO_OK) result->guard_band = first;
  }
  return rc;
}

// explore_safeset(target) with a caller's own target (models/GoOSE.py:116-119): argmin over the safe set S_t of the last sweep of
// ||x - target||_2 -- the two kernels the GoOSE sweep runs for its own target (r05: the host class used to rebuild all N coordinates
// and distances in NumPy next to a 0.5 ms device sweep)
int sbo_explore_safeset(sbo_ctx* c, const double* target, int64_t* index_out, double* x_out) {
  if (!c || !target || !index_out) return fail(SBO_E_INVALID, "NULL argument");
  if (!c->masks_valid) return fail(SBO_E_INVALID, "no sweep has produced a safe set on these candidates");
  SBO_HIP(hipSetDevice(c->device));
  const long long n = c->cs.n_local;
  int rc;
  if ((rc = ensure(c->scal, sizeof(SweepScalars)))) return rc;
  const int nb = reduce_blocks(c);
  if ((rc = ensure(c->partial, partial_stride(nb)))) return rc;
  if (c->masks_bits && n > 0) col_expand(c, c->cbS, (uint8_t*)c->maskS.p);
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  double* dev_t = (double*)c->scal.p + 256;
  double t8[SBO_MAX_D] = {0};
  for (int a = 0; a < c->cs.d; ++a) t8[a] = target[a];
  SBO_HIP(hipMemcpyAsync(dev_t, t8, sizeof(t8), hipMemcpyHostToDevice, c->stream));
  if (n > 0) {
    if (c->dtype == SBO_F64) launch_argmin_dist<double>(c, dev_t, nb);
    else launch_argmin_dist<float>(c, dev_t, nb);
  }
  hipLaunchKernelGGL((k_arg_final<false>), dim3(1), dim3(256), 0, c->stream, (const Best*)c->partial.p, n > 0 ? nb : 0, sc, kArgSlots - 1,
                     (long long*)nullptr, 0);
  SBO_HIP(hipGetLastError());
  SweepScalars h;
  bool is_max[kArgSlots];
  for (int t = 0; t < kArgSlots; ++t) is_max[t] = false;
  if (multi_rank(c) && (rc = ensure(c->xch, sizeof(double) * (size_t)(c->world * kC3Row + 64)))) return rc;
  if ((rc = sweep_exchange_back(c, h, is_max))) return rc;
  *index_out = h.arg_idx[kArgSlots - 1];
  if (x_out) coords_of(c, h.arg_idx[kArgSlots - 1], x_out);
  return h.arg_idx[kArgSlots - 1] >= 0 ? SBO_OK : fail(SBO_E_EMPTY_SAFE_SET, "safe set S_t is empty on this candidate set");
}

int sbo_masks_get(sbo_ctx* c, int which, int cidx, uint8_t* out) {
  if (!c || !out) return fail(SBO_E_INVALID, "NULL argument");
  if (!c->masks_valid) return fail(SBO_E_INVALID, "no sweep has produced masks on these candidates");
  const long long n = c->cs.n_local;
  if (c->masks_bits && n > 0) {
    // the last sweep ran on column words (sets_colpath.inc.hpp): the byte form of the mask asked for is made here
    SBO_HIP(hipSetDevice(c->device));
    if (which == SBO_MASK_S) col_expand(c, c->cbS, (uint8_t*)c->maskS.p);
    else if (which == SBO_MASK_U) col_expand(c, c->cbU, (uint8_t*)c->maskU.p);
    else if (which == SBO_MASK_M) col_expand(c, c->cbM, (uint8_t*)c->maskM.p);
    else if (which == SBO_MASK_G && cidx == 1 && !c->col_G_bytes) col_expand(c, c->cbG, (uint8_t*)c->maskG.p);
    SBO_HIP(hipGetLastError());
  }
  const void* src = nullptr;
  switch (which) {
    case SBO_MASK_S: src = c->maskS.p; break;
    case SBO_MASK_U: src = c->maskU.p; break;
    case SBO_MASK_M:   // after a trust-region sweep this slot holds S intersected with the ball
      if (c->last_sweep != 1 && c->last_sweep != 3) return fail(SBO_E_INVALID, "M is produced by the SafeOpt sweep");
      src = c->maskM.p;
      break;
    case SBO_MASK_G:
    case SBO_MASK_O:
      if ((which == SBO_MASK_G) != (c->last_sweep == 1)) return fail(SBO_E_INVALID, "mask not produced by the last sweep");
      if (cidx < 1 || cidx >= c->mc.q) return fail(SBO_E_INVALID, "constraint index out of range");
      src = (const uint8_t*)(which == SBO_MASK_G ? c->maskG.p : c->maskO.p) + (size_t)(cidx - 1) * n;
      break;
    default: return fail(SBO_E_INVALID, "unknown mask");
  }
  if (n > 0) {
    SBO_HIP(hipMemcpyAsync(out, src, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    SBO_HIP(hipStreamSynchronize(c->stream));
  }
  return SBO_OK;
}

}  // extern "C"
