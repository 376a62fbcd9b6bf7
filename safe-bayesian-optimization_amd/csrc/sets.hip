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
#include "device_common.hpp"

namespace sbo {

int comm_allreduce_max_u64(sbo_ctx* c, unsigned long long* dev, int count);
int comm_allreduce_min_u64(sbo_ctx* c, unsigned long long* dev, int count);
int comm_allreduce_sum_f64(sbo_ctx* c, double* dev, int count);
int comm_allgather_bytes(sbo_ctx* c, const void* send, void* recv, size_t bytes_per_rank);

constexpr double kInfD = 1.0e300;
constexpr int kArgSlots = 16;

// small device-resident scalar block of one sweep
struct SweepScalars {
  unsigned long long ustar_key;            // min over S of ord_key(ucb_0)
  unsigned long long rmax_key[kMaxQ];      // max over S of ord_key(ucb_c)
  long long count_S, count_U, count_M;
  long long count_set[kMaxQ];              // |G_c| or |O_c|
  long long n_amb, n_amb_total;
  long long n_scan;                        // candidates handed from the coarse decision to the wave-per-candidate scan
  double arg_val[kArgSlots];
  long long arg_idx[kArgSlots];
};

struct Best {
  double v;
  long long i;   // global flat index, -1 = none
};

template <bool MAX>
__device__ __forceinline__ bool better(const Best& a, const Best& b) {
  if (a.i < 0) return false;
  if (b.i < 0) return true;
  if (MAX ? (a.v > b.v) : (a.v < b.v)) return true;
  return a.v == b.v && a.i < b.i;   // ties -> lowest flat index
}

template <bool MAX>
__device__ __forceinline__ Best wave_best(Best x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    Best y;
    y.v = __shfl_xor(x.v, o);
    y.i = __shfl_xor(x.i, o);
    if (better<MAX>(y, x)) x = y;
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
    Best y = lane < nw ? sh[lane] : Best{0.0, -1};
    y = wave_best<MAX>(y);
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

// ---- K3a: S / U masks, u* --------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_classify(const T* __restrict__ mean, const T* __restrict__ var, long long n,
                                                  int q, T b, uint8_t* __restrict__ S, uint8_t* __restrict__ U,
                                                  SweepScalars* sc) {
  __shared__ unsigned long long rmax_sh[kMaxQ];   // max over S of ucb_c: bounds the expander search radius
  if (threadIdx.x < kMaxQ) rmax_sh[threadIdx.x] = 0;
  __syncthreads();
  unsigned long long umin = ~0ull;
  long long cS = 0, cU = 0;
  auto one = [&](long long g, const T* mv /* [2 * q]: mean_c, var_c of this candidate */) {
    bool s = true, u = true;
    T ucbc[kMaxQ];                                  // kept for the radius keys: one sqrt per (candidate, constraint)
#pragma unroll
    for (int c = 1; c < kMaxQ; ++c) {
      if (c < q) {
        T lcb;
        lcb_ucb(mv[2 * c], mv[2 * c + 1], b, lcb, ucbc[c]);
        s = s && (lcb >= T(0));
        u = u && (lcb <= T(0));
      }
    }
    cS += s;
    cU += u;
    if (s) {
      T lcb, ucb;
      lcb_ucb(mv[0], mv[1], b, lcb, ucb);
      const unsigned long long k = ord_key((double)ucb);
      umin = k < umin ? k : umin;
#pragma unroll
      for (int c = 1; c < kMaxQ; ++c) {
        if (c < q) {
          const unsigned long long kc = ord_key((double)ucbc[c]);
          if (kc > rmax_sh[c]) atomicMax(&rmax_sh[c], kc);
        }
      }
    }
    return (unsigned)(s ? 1u : 0u) | (unsigned)(u ? 2u : 0u);
  };
  // two consecutive candidates per thread: the q mean / var streams are read with 16-byte (fp64) / 8-byte (fp32) loads
  typedef T T2 __attribute__((ext_vector_type(2)));
  const long long npair = n >> 1;
  const bool aligned = (n & 1) == 0;               // output c starts at c * n elements: pairs stay aligned only for even n
  if (aligned) {
    for (long long pi = (long long)blockIdx.x * blockDim.x + threadIdx.x; pi < npair; pi += (long long)gridDim.x * blockDim.x) {
      T mv0[2 * kMaxQ], mv1[2 * kMaxQ];
#pragma unroll
      for (int c = 0; c < kMaxQ; ++c) {
        if (c < q) {
          const T2 m2 = *reinterpret_cast<const T2*>(mean + (size_t)c * n + 2 * pi);
          const T2 v2 = *reinterpret_cast<const T2*>(var + (size_t)c * n + 2 * pi);
          mv0[2 * c] = m2[0]; mv0[2 * c + 1] = v2[0];
          mv1[2 * c] = m2[1]; mv1[2 * c + 1] = v2[1];
        }
      }
      const unsigned r0 = one(2 * pi, mv0), r1 = one(2 * pi + 1, mv1);
      *reinterpret_cast<unsigned short*>(S + 2 * pi) = (unsigned short)((r0 & 1u) | ((r1 & 1u) << 8));
      *reinterpret_cast<unsigned short*>(U + 2 * pi) = (unsigned short)(((r0 >> 1) & 1u) | (((r1 >> 1) & 1u) << 8));
    }
  } else {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
      T mv[2 * kMaxQ];
#pragma unroll
      for (int c = 0; c < kMaxQ; ++c)
        if (c < q) { mv[2 * c] = mean[(size_t)c * n + g]; mv[2 * c + 1] = var[(size_t)c * n + g]; }
      const unsigned r = one(g, mv);
      S[g] = (uint8_t)(r & 1u);
      U[g] = (uint8_t)((r >> 1) & 1u);
    }
  }
  umin = block_ext_u64<false>(umin);
  cS = block_sum_ll(cS);
  cU = block_sum_ll(cU);
  __syncthreads();
  if (threadIdx.x == 0) {
    if (umin != ~0ull) atomicMin(&sc->ustar_key, umin);
    if (cS) atomicAdd((unsigned long long*)&sc->count_S, (unsigned long long)cS);
    if (cU) atomicAdd((unsigned long long*)&sc->count_U, (unsigned long long)cU);
  }
  if (threadIdx.x >= 1 && threadIdx.x < q && rmax_sh[threadIdx.x]) atomicMax(&sc->rmax_key[threadIdx.x], rmax_sh[threadIdx.x]);
}

// ---- K3b + K5: M mask and arg-max of var_0 over M -----------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_minimizer(const T* __restrict__ mean0, const T* __restrict__ var0, long long n,
                                                   long long first, T b, const uint8_t* __restrict__ S,
                                                   uint8_t* __restrict__ M, SweepScalars* sc, Best* partial) {
  const T ustar = (T)ord_val(sc->ustar_key);
  Best best{0.0, -1};
  long long cM = 0;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    bool m = false;
    if (S[g]) {
      T lcb, ucb;
      lcb_ucb(mean0[g], var0[g], b, lcb, ucb);
      m = lcb <= ustar;                       // models/SafeOpt.py:62
    }
    M[g] = m;
    if (m) {
      ++cM;
      const Best cand{(double)var0[g], first + g};
      if (better<true>(cand, best)) best = cand;
    }
  }
  best = block_best<true>(best);
  cM = block_sum_ll(cM);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = best;
    if (cM) atomicAdd((unsigned long long*)&sc->count_M, (unsigned long long)cM);
  }
}

// generic masked arg-max / arg-min of a value array, plus the mask population
template <typename T, bool MAX>
__global__ __launch_bounds__(256) void k_arg_masked(const T* __restrict__ val, const uint8_t* __restrict__ mask, long long n,
                                                    long long first, long long* count, Best* partial) {
  Best best{0.0, -1};
  long long cnt = 0;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    if (mask[g]) {
      ++cnt;
      const Best cand{(double)val[g], first + g};
      if (better<MAX>(cand, best)) best = cand;
    }
  }
  best = block_best<MAX>(best);
  cnt = block_sum_ll(cnt);
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = best;
    if (cnt && count) atomicAdd((unsigned long long*)count, (unsigned long long)cnt);
  }
}

template <bool MAX>
__global__ __launch_bounds__(256) void k_arg_final(const Best* __restrict__ partial, int nparts, SweepScalars* sc, int slot) {
  Best best{0.0, -1};
  for (int i = threadIdx.x; i < nparts; i += blockDim.x)
    if (better<MAX>(partial[i], best)) best = partial[i];
  best = block_best<MAX>(best);
  if (threadIdx.x == 0) {
    sc->arg_val[slot] = best.v;
    sc->arg_idx[slot] = best.i;
  }
}

// ---- K4: exact Euclidean distance transform of the U mask on the grid ---------------------------------
// axis 0: one wave per grid line, nearest set bit on either side found with ballots (coalesced, exact)
__global__ __launch_bounds__(256) void k_edt_axis0(const uint8_t* __restrict__ U, long long nlines, int count0, double h0,
                                                   double* __restrict__ D) {
  const int lane = threadIdx.x & 63;
  const long long line = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (line >= nlines) return;
  const uint8_t* u = U + line * count0;
  double* d = D + line * count0;
  const int nch = (count0 + 63) >> 6;
  long long carry = -1;
  for (int ch = 0; ch < nch; ++ch) {
    const int i = ch * 64 + lane;
    const bool bit = i < count0 && u[i];
    const unsigned long long m = __ballot(bit);
    const unsigned long long lower = m & (lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull));
    const long long li = lower ? (long long)ch * 64 + (63 - __clzll((long long)lower)) : carry;
    if (i < count0) d[i] = (double)li;
    if (m) carry = (long long)ch * 64 + (63 - __clzll((long long)m));
  }
  carry = -1;
  for (int ch = nch - 1; ch >= 0; --ch) {
    const int i = ch * 64 + lane;
    const bool bit = i < count0 && u[i];
    const unsigned long long m = __ballot(bit);
    const unsigned long long upper = m >> lane;
    const long long ri = upper ? (long long)ch * 64 + lane + (__ffsll((long long)upper) - 1) : carry;
    if (i < count0) {
      const long long li = (long long)d[i];
      long long t = -1;
      if (li >= 0) t = i - li;
      if (ri >= 0 && (t < 0 || ri - i < t)) t = ri - i;
      double v = kInfD;
      if (t >= 0) {
        const double dt = h0 * (double)t;
        v = dt * dt;
      }
      d[i] = v;
    }
    if (m) carry = (long long)ch * 64 + (__ffsll((long long)m) - 1);
  }
}

// axes >= 1: D_out[g] = min_t D_in[g + t stride] + (h t)^2, searched outwards with the two exits
//   (h t)^2 >= best  (nothing further can improve)  and  h t > cap  (beyond any radius that matters).
// `accept2`: once best <= accept2 the caller's decision is already "within the radius" and a smaller minimum cannot
// change it, so the search stops (pass -1 to get the exact minimum).
__device__ __forceinline__ double edt_scan_point(const double* __restrict__ Din, long long g, long long stride, int cnt,
                                                 int ia, double h, double cap, double accept2 = -1.0) {
  double best = Din[g];
  for (int t = 1; t < cnt; ++t) {
    const double dt = h * (double)t;
    const double e = dt * dt;
    if (e >= best || dt > cap || best <= accept2) break;
    const bool lo_ok = ia - t >= 0, hi_ok = ia + t < cnt;
    if (!lo_ok && !hi_ok) break;
    const double c1 = lo_ok ? Din[g - (long long)t * stride] : kInfD;
    const double c2 = hi_ok ? Din[g + (long long)t * stride] : kInfD;
    const double c = (c1 < c2 ? c1 : c2) + e;
    best = c < best ? c : best;
  }
  return best;
}

// Last-axis scans with a one-level min-pyramid.  Bmin[b * stride + p] = min of the input over the kBlk (or `blk`) steps
// of block b at in-plane position p.  A block whose bound  Bmin + (h gap)^2  cannot beat the running minimum is skipped
// with one load instead of `blk`; the candidates examined inside a block and their arithmetic are those of the
// step-by-step scan, so the minimum (up to the same early exits) is identical.  This keeps the scan cost near
// O(sqrt(radius in steps)) when the grid is much finer along the last axis than the radius (weak-scaling grids).
__global__ __launch_bounds__(256) void k_block_min(const double* __restrict__ Din, long long stride, int cnt, int blk,
                                                   double* __restrict__ Bmin) {
  const int nblk = (cnt + blk - 1) / blk;
  const long long total = (long long)nblk * stride;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long p = i % stride;
    const int b = (int)(i / stride);
    const int j1 = (b + 1) * blk < cnt ? (b + 1) * blk : cnt;
    double m = kInfD;
    for (int j = b * blk; j < j1; ++j) {
      const double v = Din[(long long)j * stride + p];
      m = v < m ? v : m;
    }
    Bmin[i] = m;
  }
}

// Euclidean form (values >= 0, exits as edt_scan_point: (h t)^2 >= best, h t > cap, best <= accept2).
// Order of visits: the block with the smallest bound first (it almost always holds the minimiser, so `best` is near
// its final value after one block), then every block whose bound still beats `best`.
__device__ __forceinline__ double edt_scan_blocked(const double* __restrict__ Din, const double* __restrict__ Bmin, long long p,
                                                   long long stride, int cnt, int ia, double h, double cap, double accept2,
                                                   int blk) {
  // The few candidates that reach this scan decide the kernel's duration through their chain of dependent loads, so
  // loads are issued in independent batches (eight block values / eight bounds at a time) and only then examined.
  double best = Din[(long long)ia * stride + p];
  const int nblk = (cnt + blk - 1) / blk, b0 = ia / blk;
  auto scan_block = [&](int b) {
    const int j1 = (b + 1) * blk < cnt ? (b + 1) * blk : cnt;
    for (int j = b * blk; j < j1; j += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = j + u < j1 ? Din[(long long)(j + u) * stride + p] : kInfD;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int jj = j + u;
        const double dt = h * (double)(jj > ia ? jj - ia : ia - jj);
        const double cnd = v[u] + dt * dt;
        if (dt <= cap && cnd < best) best = cnd;
      }
    }
  };
  // gap (in steps) between ia and the nearest step of block b
  auto gap_of = [&](int b) { return b == b0 ? 0 : (b < b0 ? ia - (b * blk + blk - 1) : b * blk - ia); };
  auto bound_at = [&](int b) { return (b >= 0 && b < nblk) ? Bmin[(long long)b * stride + p] : kInfD; };
  // pass A: block with the smallest bound
  double lb_min = Bmin[(long long)b0 * stride + p];
  int b_min = b0;
  for (int k0 = 1; k0 < nblk; k0 += 4) {
    double lo[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { lo[u] = bound_at(b0 - k0 - u); hi[u] = bound_at(b0 + k0 + u); }
    bool any = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      any = false;
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        const int b = side ? b0 + k0 + u : b0 - k0 - u;
        if (b < 0 || b >= nblk) continue;
        const double dg = h * (double)gap_of(b);
        const double e = dg * dg;
        if (e >= best || e >= lb_min || dg > cap) continue;
        any = true;
        const double lb = (side ? hi[u] : lo[u]) + e;
        if (lb < lb_min) { lb_min = lb; b_min = b; }
      }
      if (!any) break;
    }
    if (!any) break;
  }
  if (lb_min < best) scan_block(b_min);
  if (best <= accept2) return best;
  // pass B: whatever can still improve
  if (b_min != b0 && Bmin[(long long)b0 * stride + p] < best) scan_block(b0);
  for (int k0 = 1; k0 < nblk && !(best <= accept2); k0 += 4) {
    double lo[4], hi[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { lo[u] = bound_at(b0 - k0 - u); hi[u] = bound_at(b0 + k0 + u); }
    bool any = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      any = false;
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        const int b = side ? b0 + k0 + u : b0 - k0 - u;
        if (b < 0 || b >= nblk) continue;
        const double dg = h * (double)gap_of(b);
        const double e = dg * dg;
        if (e >= best || dg > cap) continue;
        any = true;
        if (b != b_min && (side ? hi[u] : lo[u]) + e < best) scan_block(b);
      }
      if (!any) break;
    }
    if (!any) break;
  }
  return best;
}

__global__ __launch_bounds__(256) void k_edt_scan(const double* __restrict__ Din, double* __restrict__ Dout, long long n,
                                                  long long stride, int cnt, double h, const SweepScalars* sc, int c,
                                                  const unsigned long long* Lkeys, int lidx, int uncapped) {
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const double rmax = sc->rmax_key[c] ? ord_val(sc->rmax_key[c]) : 0.0;
  const double cap = (L > 0 && !uncapped) ? rmax / L * 1.000001 + 1e-6 : kInfD;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    const int ia = (int)((g / stride) % cnt);
    Dout[g] = edt_scan_point(Din, g, stride, cnt, ia, h, cap);
  }
}

// Coarse pre-decision for the expander query.  The U mask is OR-reduced over cells of kCoarse^d candidates and the
// exact transform of that small mask gives, for any candidate g in cell C, the sandwich
//     dC - delta <= dist(g, U) <= dC + delta,   delta = (kCoarse - 1) * sqrt(sum_a h_a^2)
// (dC = distance between cell origins to the nearest U-holding cell).  Candidates whose verdict is the same at both
// ends skip the per-candidate scan of the fine transform; only the shell around the boundary of G_c scans.
constexpr int kCoarse = 8;
struct CoarseGrid {
  int enabled;
  int d;
  long long count[kMaxD];    // fine counts
  long long ccount[kMaxD];   // coarse counts
  double delta;
  const double* Dc;          // squared coarse distances [prod ccount]
};

__global__ __launch_bounds__(256) void k_coarsen_mask(const uint8_t* __restrict__ U, long long n, const CoarseGrid cg,
                                                      uint8_t* __restrict__ Uc) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    if (!U[g]) continue;
    long long f = g, cell = 0, cs = 1;
    for (int a = 0; a < cg.d; ++a) {
      const long long i = f % cg.count[a];
      f /= cg.count[a];
      cell += (i / kCoarse) * cs;
      cs *= cg.ccount[a];
    }
    Uc[cell] = 1;   // idempotent store
  }
}

__device__ __forceinline__ double coarse_dist2(const CoarseGrid& cg, long long gg) {
  long long f = gg, cell = 0, cs = 1;
  for (int a = 0; a < cg.d; ++a) {
    const long long i = f % cg.count[a];
    f /= cg.count[a];
    cell += (i / kCoarse) * cs;
    cs *= cg.ccount[a];
  }
  return cg.Dc[cell];
}

// Reference expression for one (g, h) pair, unfused, in the oracle's order:
//   ucb - L * sqrt(sum_a (x_g[a] - x_h[a] + 1e-8)^2) >= 0            models/SafeOpt.py:85-88
template <int D>
__device__ __forceinline__ bool lipschitz_pair(const double (&xg)[D], const double (&xh)[D], int d, double ucb, double L) {
  double ss = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    if (a < d) {
      const double df = __dadd_rn(__dsub_rn(xg[a], xh[a]), 1e-8);
      ss = (a == 0) ? __dmul_rn(df, df) : __dadd_rn(ss, __dmul_rn(df, df));
    }
  }
  const double dist = __dsqrt_rn(ss);
  return __dsub_rn(ucb, __dmul_rn(L, dist)) >= 0.0;
}

// last axis + decision.  For g in S: nearest-U distance dm (unshifted) -> G bit, or the ambiguous list when
// ucb - L dm lies inside the band the "+1e-8" shift and rounding can move it across zero.
template <typename T>
__global__ __launch_bounds__(256) void k_edt_decide(const double* __restrict__ Din, long long n, long long goff,
                                                    long long stride, int cnt,
                                                    double h, int d, double xscale, const T* __restrict__ mean_c,
                                                    const T* __restrict__ var_c, T b, const uint8_t* __restrict__ S,
                                                    const unsigned long long* Lkeys, int lidx, SweepScalars* sc,
                                                    uint8_t* __restrict__ G, long long* __restrict__ amb,
                                                    const CoarseGrid cg, const double* __restrict__ Bmin, int blk,
                                                    long long* __restrict__ scanlist) {
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const bool anyU = sc->count_U > 0;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    uint8_t out = 0;
    if (S[g] && anyU) {
      T lcb, ucbT;
      lcb_ucb(mean_c[g], var_c[g], b, lcb, ucbT);
      const double ucb = (double)ucbT;
      if (!(L > 0)) {
        out = ucb >= 0.0;                         // radius unbounded: any U point is a witness
      } else {
        const double eps_abs = 1.01e-8 * sqrt((double)d) + 1e-14 * xscale + 1e-13;
        const double cap = ucb / L + 4.0 * eps_abs + 1e-9 * fabs(ucb / L);
        const long long gg = goff + g;   // index in the transform (whole grid when ranks share it)
        if (cg.enabled) {
          const double dC = sqrt(coarse_dist2(cg, gg));
          const double dhi = dC * (1.0 + 1e-9) + cg.delta, dlo = fmax(0.0, dC * (1.0 - 1e-9) - cg.delta);
          const double tolc = 1e-12 * (fabs(ucb) + L * dhi);
          if (ucb - L * (dhi + eps_abs + 1e-11 * dhi) > tolc) { G[g] = 1; continue; }     // within the radius for sure
          if (ucb - L * (dlo - eps_abs - 1e-11 * dlo) < -tolc) { G[g] = 0; continue; }    // beyond it for sure
        }
        if (scanlist && Bmin && cnt > 1) {
          // the few candidates the coarse bounds leave open go to k_edt_scan_list (one wave each): a lane scanning
          // here would hold its whole wave for a chain of ~100 dependent loads
          scanlist[atomicAdd((unsigned long long*)&sc->n_scan, 1ull)] = g;
          G[g] = 0;
          continue;
        }
        const int ia = (int)((gg / stride) % cnt);
        const double thr = ucb / L * (1.0 - 1e-10) - 2.0 * eps_abs - 1e-12;   // inside it the verdict is "sure true"
        const double acc2 = thr > 0 ? thr * thr : -1.0;
        const double best = cnt <= 1 ? Din[gg]
                            : Bmin  ? edt_scan_blocked(Din, Bmin, gg % stride, stride, cnt, ia, h, cap, acc2, blk)
                                    : edt_scan_point(Din, gg, stride, cnt, ia, h, cap, acc2);
        if (best < 0.5 * kInfD) {
          const double dm = sqrt(best);
          const double eps = eps_abs + 1e-11 * dm;
          const double tol = 1e-12 * (fabs(ucb) + L * dm);
          const double lo = ucb - L * (dm + eps), hi = ucb - L * (dm - eps);
          if (lo > tol) out = 1;
          else if (hi >= -tol) {
            const long long slot = (long long)atomicAdd((unsigned long long*)&sc->n_amb, 1ull);
            amb[slot] = g;
          }
        }
      }
    }
    G[g] = out;
  }
}

// Last-axis scan + verdict for the listed candidates, one group of GL lanes (half a wave or a wave, GL >= blk) per
// candidate: the lanes take GL blocks (bounds) or the steps of one block at a time and combine with group minima -- the
// same candidates and arithmetic as edt_scan_blocked, with the dependent-load chain cut from ~100 to ~5 per candidate.
// (The two halves of a wave follow their own trip counts; every cross-lane operation stays inside one half.)
template <typename T, int GL>
__global__ __launch_bounds__(256) void k_edt_scan_list(const double* __restrict__ Din, long long goff, long long stride, int cnt,
                                                       double h, int d, double xscale, const T* __restrict__ mean_c,
                                                       const T* __restrict__ var_c, T b, const unsigned long long* Lkeys, int lidx,
                                                       SweepScalars* sc, uint8_t* __restrict__ G, long long* __restrict__ amb,
                                                       const double* __restrict__ Bmin, int blk,
                                                       const long long* __restrict__ scanlist) {
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const long long nscan = sc->n_scan;
  const int lane = threadIdx.x & (GL - 1);
  const int sub = (threadIdx.x & 63) / GL;
  const long long ngroups = (long long)gridDim.x * (blockDim.x / GL);
  const int nblk = (cnt + blk - 1) / blk;
  auto group_min = [&](double v) {
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) { const double w = __shfl_xor(v, o); v = w < v ? w : v; }
    return v;
  };
  auto group_ballot = [&](bool pred) {
    const unsigned long long m = __ballot(pred);
    return GL == 64 ? m : ((m >> (32 * sub)) & 0xffffffffull);
  };
  for (long long qi = (long long)blockIdx.x * (blockDim.x / GL) + threadIdx.x / GL; qi < nscan; qi += ngroups) {
    const long long g = scanlist[qi];
    T lcb, ucbT;
    lcb_ucb(mean_c[g], var_c[g], b, lcb, ucbT);
    const double ucb = (double)ucbT;
    const double eps_abs = 1.01e-8 * sqrt((double)d) + 1e-14 * xscale + 1e-13;
    const double cap = ucb / L + 4.0 * eps_abs + 1e-9 * fabs(ucb / L);
    const double thr = ucb / L * (1.0 - 1e-10) - 2.0 * eps_abs - 1e-12;
    const double acc2 = thr > 0 ? thr * thr : -1.0;
    const long long gg = goff + g, p = gg % stride;
    const int ia = (int)((gg / stride) % cnt), b0 = ia / blk;
    double best = Din[(long long)ia * stride + p];
    // blocks within reach of the radius
    const int kmax = (int)fmin((double)nblk, floor(cap / (h * (double)blk)) + 2.0);
    const int blo = b0 - kmax > 0 ? b0 - kmax : 0, bhi = b0 + kmax < nblk - 1 ? b0 + kmax : nblk - 1;
    auto bound_of = [&](int bb) {       // bound of block bb for this lane (inf outside the reach / the axis)
      if (bb < blo || bb > bhi) return kInfD;
      const int gap = bb == b0 ? 0 : (bb < b0 ? ia - (bb * blk + blk - 1) : bb * blk - ia);
      const double dg = h * (double)gap;
      if (dg > cap) return kInfD;
      return Bmin[(long long)bb * stride + p] + dg * dg;
    };
    auto scan_block = [&](int bb) {     // the group: the (<= GL) steps of block bb
      const int jn = bb * blk + lane;
      double cnd = kInfD;
      if (lane < blk && jn < cnt) {
        const double dt = h * (double)(jn > ia ? jn - ia : ia - jn);
        if (dt <= cap) cnd = Din[(long long)jn * stride + p] + dt * dt;
      }
      cnd = group_min(cnd);
      best = cnd < best ? cnd : best;
    };
    // pass A: the block with the smallest bound
    double lb_min = kInfD;
    int b_min = -1;
    for (int base = blo; base <= bhi; base += GL) {
      const double lb = bound_of(base + lane);
      const double m = group_min(lb);
      if (m < lb_min) {
        lb_min = m;
        const unsigned long long who = group_ballot(lb == m);
        b_min = base + (int)(__ffsll((long long)who) - 1);
      }
    }
    if (b_min >= 0 && lb_min < best) scan_block(b_min);
    // pass B: every other block whose bound still beats the running minimum
    for (int base = blo; base <= bhi && !(best <= acc2); base += GL) {
      const double lb = bound_of(base + lane);
      unsigned long long todo = group_ballot(lb < best && base + lane != b_min);
      while (todo && !(best <= acc2)) {
        const int l = (int)(__ffsll((long long)todo) - 1);
        todo &= todo - 1;
        const double lbl = __shfl(lb, l + GL * sub);
        if (lbl < best) scan_block(base + l);
      }
    }
    if (lane == 0) {
      uint8_t out = 0;
      if (best < 0.5 * kInfD) {
        const double dm = sqrt(best);
        const double eps = eps_abs + 1e-11 * dm;
        const double tol = 1e-12 * (fabs(ucb) + L * dm);
        const double lo = ucb - L * (dm + eps), hi = ucb - L * (dm - eps);
        if (lo > tol) out = 1;
        else if (hi >= -tol) amb[atomicAdd((unsigned long long*)&sc->n_amb, 1ull)] = g;
      }
      G[g] = out;
    }
  }
}

// every S point goes to the exhaustive list (explicit candidate lists have no grid to transform)
__global__ __launch_bounds__(256) void k_list_safe(const uint8_t* __restrict__ S, long long n, SweepScalars* sc,
                                                   uint8_t* __restrict__ G, long long* __restrict__ amb) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    G[g] = 0;
    if (S[g]) {
      const long long slot = (long long)atomicAdd((unsigned long long*)&sc->n_amb, 1ull);
      amb[slot] = g;
    }
  }
}

// exhaustive evaluation of the reference predicate for the listed g: one workgroup per g, every U point that can matter
template <typename T, int D>
__global__ __launch_bounds__(256) void k_expander_exact(const CandSpec cs, const CandSpec csU, const T* __restrict__ mean_c,
                                                        const T* __restrict__ var_c, T b, const uint8_t* __restrict__ U,
                                                        const unsigned long long* Lkeys, int lidx, SweepScalars* sc,
                                                        const long long* __restrict__ amb, uint8_t* __restrict__ G) {
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const long long namb = sc->n_amb;
  constexpr int kParts = 64;   // a listed candidate's box can be as large as the grid: cut into slices, one workgroup each
  for (long long wi = blockIdx.x; wi < namb * kParts; wi += gridDim.x) {
    const long long qi = wi / kParts;
    const int part = (int)(wi % kParts);
    const long long g = amb[qi];
    T lcb, ucbT;
    lcb_ucb(mean_c[g], var_c[g], b, lcb, ucbT);
    const double ucb = (double)ucbT;
    double xg[D];
    cand_coords<D>(cs, g, xg);
    int found = 0;
    if (csU.kind == 1) {
      // grid: only witnesses inside the index box of half-width ceil(r / h_a) + 1 around g can satisfy the predicate
      long long lo[D], len[D], stridea[D];
      long long f = cs.first + g, total = 1, sa = 1;
      const double rg = L > 0 ? ucb / L : 1e300;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        lo[a] = 0; len[a] = 1; stridea[a] = 0;
        if (a < cs.d) {
          const long long cnt = cs.count[a];
          const long long ig = f % cnt;
          f /= cnt;
          long long R = cnt;
          if (L > 0 && cs.step[a] > 0) {
            const double rr = (rg * (1.0 + 1e-9) + 1e-7) / cs.step[a];
            R = rr < (double)cnt ? (long long)ceil(rr) + 1 : cnt;
            if (R < 0) R = 0;
          }
          const long long l0 = ig - R > 0 ? ig - R : 0, h0 = ig + R < cnt - 1 ? ig + R : cnt - 1;
          lo[a] = l0; len[a] = h0 - l0 + 1; stridea[a] = sa;
          total *= len[a];
          sa *= cnt;
        }
      }
      const long long chunk = (total + kParts - 1) / kParts;
      const long long t1 = (part + 1) * chunk < total ? (part + 1) * chunk : total;
      for (long long t = part * chunk + threadIdx.x; t < t1 && !found; t += blockDim.x) {
        long long u = t, hh = 0;
        double xh[D];
#pragma unroll
        for (int a = 0; a < D; ++a) {
          xh[a] = 0.0;
          if (a < cs.d) {
            const long long ia = lo[a] + u % len[a];
            u /= len[a];
            hh += ia * stridea[a];
            const long long cnt = cs.count[a];
            xh[a] = (ia == cnt - 1 && cnt > 1) ? cs.hi[a] : __dadd_rn(cs.lo[a], __dmul_rn((double)ia, cs.step[a]));
          }
        }
        const long long hl = hh - csU.first;
        if (hl >= 0 && hl < csU.n_local && U[hl] && lipschitz_pair<D>(xg, xh, cs.d, ucb, L)) found = 1;
      }
    } else {
      for (long long hh = (long long)part * blockDim.x + threadIdx.x; hh < csU.n_local && !found; hh += (long long)kParts * blockDim.x) {
        if (U[hh]) {
          double xh[D];
          cand_coords<D>(csU, hh, xh);
          if (lipschitz_pair<D>(xg, xh, cs.d, ucb, L)) found = 1;
        }
      }
    }
    found = __syncthreads_or(found);
    if (threadIdx.x == 0 && found) G[g] = 1;
    __syncthreads();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sc->n_amb_total += namb;
  }
}

__global__ void k_reset_amb(SweepScalars* sc) { sc->n_amb = 0; sc->n_scan = 0; }

// ---- multi-rank exchange -----------------------------------------------------------
// C1: one max all-reduce of [~u*_key, L keys, radius keys]
__global__ void k_pack_c1(const SweepScalars* sc, const unsigned long long* Lkeys, unsigned long long* buf) {
  const int t = threadIdx.x;
  if (t == 0) buf[0] = ~sc->ustar_key;
  if (t < kMaxQ) { buf[1 + t] = Lkeys[t]; buf[1 + kMaxQ + t] = sc->rmax_key[t]; }
}
__global__ void k_unpack_c1(SweepScalars* sc, unsigned long long* Lkeys, const unsigned long long* buf) {
  const int t = threadIdx.x;
  if (t == 0) sc->ustar_key = ~buf[0];
  if (t < kMaxQ) { Lkeys[t] = buf[1 + t]; sc->rmax_key[t] = buf[1 + kMaxQ + t]; }
}
// C2: all-gathered padded shards -> contiguous whole-grid mask
template <typename E>
__global__ __launch_bounds__(256) void k_compact_shards(const E* __restrict__ recv, long long maxlocal, int world,
                                                        const long long* __restrict__ first_of, long long total,
                                                        E* __restrict__ full) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    int r = 0;
    while (r + 1 < world && g >= first_of[r + 1]) ++r;
    full[g] = recv[(size_t)r * maxlocal + (g - first_of[r])];
  }
}
// C2 (bit form): own mask -> one word per 64 candidates (zero beyond n); gathered words -> whole-grid byte mask
__global__ __launch_bounds__(256) void k_pack_bits(const uint8_t* __restrict__ U, long long n, long long words,
                                                   unsigned long long* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long nwaves = (long long)gridDim.x * (blockDim.x >> 6);
  for (long long w = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); w < words; w += nwaves) {
    const long long g = w * 64 + lane;
    const unsigned long long m = __ballot(g < n && U[g]);
    if (lane == 0) out[w] = m;
  }
}
__global__ __launch_bounds__(256) void k_unpack_shards(const unsigned long long* __restrict__ recv, long long words, int world,
                                                       const long long* __restrict__ first_of, long long total,
                                                       uint8_t* __restrict__ full) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    int r = 0;
    while (r + 1 < world && g >= first_of[r + 1]) ++r;
    const long long l = g - first_of[r];
    full[g] = (uint8_t)((recv[(size_t)r * words + (l >> 6)] >> (l & 63)) & 1ull);
  }
}
// C3: each rank fills its own row of [world][kC3Row] doubles, one sum all-reduce delivers every row everywhere
constexpr int kC3Row = 2 * kArgSlots + 4 + kMaxQ;
__global__ void k_pack_c3(const SweepScalars* sc, double* buf, int world, int rank) {
  for (int i = threadIdx.x; i < world * kC3Row; i += blockDim.x) buf[i] = 0.0;
  __syncthreads();
  double* row = buf + (size_t)rank * kC3Row;
  const int t = threadIdx.x;
  if (t < kArgSlots) { row[t] = sc->arg_idx[t] >= 0 ? sc->arg_val[t] : 0.0; row[kArgSlots + t] = (double)sc->arg_idx[t]; }
  if (t == 0) {
    row[2 * kArgSlots + 0] = (double)sc->count_S;
    row[2 * kArgSlots + 1] = (double)sc->count_U;
    row[2 * kArgSlots + 2] = (double)sc->count_M;
    row[2 * kArgSlots + 3] = (double)sc->n_amb_total;
  }
  if (t < kMaxQ) row[2 * kArgSlots + 4 + t] = (double)sc->count_set[t];
}
__global__ void k_init_scalars(SweepScalars* sc) {
  unsigned long long* w = reinterpret_cast<unsigned long long*>(sc);       // (the struct is a multiple of 8 bytes)
  for (unsigned i = threadIdx.x; i < sizeof(SweepScalars) / 8; i += blockDim.x) w[i] = 0ull;
  __syncthreads();
  if (threadIdx.x == 0) sc->ustar_key = ~0ull;
  if (threadIdx.x < kArgSlots) sc->arg_idx[threadIdx.x] = -1;
}

// ---- GoOSE: optimistic set O_c (models/GoOSE.py:93-101) -----------------------------------------------------
//   O_c = {h in U : exists g in S, ucb_c(g) - L ||x_g - x_h + 1e-8|| >= 0}
// Here the radius ucb_c(g)/L belongs to the *source* g, so the query is a union-of-balls coverage test, not a
// nearest-neighbour one; it is answered exactly by evaluating the reference predicate on every pair of
// 256-candidate runs whose bounding boxes are closer than the largest radius in the S-run.
constexpr int kRun = 256;
struct RunMeta {
  double rmax;                    // max ucb_c/L over the S points of the run (< 0: no S point)
  double lo[kMaxD], hi[kMaxD];    // bounding box of the run's S points
};

// source weights: W[g] = ucb_c(g) on the source set, -inf elsewhere (every source is in S_t, so its ucb_c >= lcb_c >= 0)
template <typename T>
__global__ __launch_bounds__(256) void k_goose_weights(const T* __restrict__ mean_c, const T* __restrict__ var_c, long long n, T b,
                                                       const uint8_t* __restrict__ src, T* __restrict__ W) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    T lcb, ucb;
    lcb_ucb(mean_c[g], var_c[g], b, lcb, ucb);
    W[g] = src[g] ? ucb : (T)-INFINITY;
  }
}

// css / W describe the SOURCE candidates (this rank's, or with ranks > 1 the whole grid); block i handles run run_lo + i
template <typename T, int D>
__global__ __launch_bounds__(256) void k_goose_run_meta(const CandSpec css, const T* __restrict__ W,
                                                        const unsigned long long* Lkeys, int lidx, long long run_lo,
                                                        RunMeta* __restrict__ meta) {
  __shared__ double red[4][1 + 2 * D];
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const long long g = (run_lo + blockIdx.x) * kRun + threadIdx.x;
  double r = -1.0, lo[D], hi[D];
#pragma unroll
  for (int a = 0; a < D; ++a) { lo[a] = 1e300; hi[a] = -1e300; }
  if (g < css.n_local) {
    const double uc = (double)W[g];
    if (uc >= 0.0) {
      r = L > 0 ? uc / L : 1e300;
      double x[D];
      cand_coords<D>(css, g, x);
#pragma unroll
      for (int a = 0; a < D; ++a) { lo[a] = x[a]; hi[a] = x[a]; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    r = fmax(r, __shfl_xor(r, o));
#pragma unroll
    for (int a = 0; a < D; ++a) { lo[a] = fmin(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmax(hi[a], __shfl_xor(hi[a], o)); }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[wave][0] = r;
#pragma unroll
    for (int a = 0; a < D; ++a) { red[wave][1 + a] = lo[a]; red[wave][1 + D + a] = hi[a]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    RunMeta m;
    m.rmax = fmax(fmax(red[0][0], red[1][0]), fmax(red[2][0], red[3][0]));
    for (int a = 0; a < kMaxD; ++a) { m.lo[a] = 0; m.hi[a] = 0; }
#pragma unroll
    for (int a = 0; a < D; ++a) {
      m.lo[a] = fmin(fmin(red[0][1 + a], red[1][1 + a]), fmin(red[2][1 + a], red[3][1 + a]));
      m.hi[a] = fmax(fmax(red[0][1 + D + a], red[1][1 + D + a]), fmax(red[2][1 + D + a], red[3][1 + D + a]));
    }
    meta[blockIdx.x] = m;
  }
}

template <typename T, int D>
__global__ __launch_bounds__(256) void k_goose_optimistic(const CandSpec cs, const CandSpec css, const T* __restrict__ W,
                                                          const uint8_t* __restrict__ U, const unsigned long long* Lkeys,
                                                          int lidx, const RunMeta* __restrict__ meta, long long run_lo,
                                                          int nruns, uint8_t* __restrict__ O) {
  __shared__ double gx[kRun][D];
  __shared__ double gr[kRun], gucb[kRun];
  __shared__ int list[kRun];
  __shared__ int nlist, nopen;
  __shared__ double ubox[2 * D];
  __shared__ double red[4][2 * D];
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const long long h = (long long)blockIdx.x * kRun + threadIdx.x;
  const bool isU = h < cs.n_local && U[h];
  double xh[D];
  if (h < cs.n_local) cand_coords<D>(cs, h, xh);
  // bounding box of this run's U points
  double lo[D], hi[D];
#pragma unroll
  for (int a = 0; a < D; ++a) { lo[a] = isU ? xh[a] : 1e300; hi[a] = isU ? xh[a] : -1e300; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int a = 0; a < D; ++a) { lo[a] = fmin(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmax(hi[a], __shfl_xor(hi[a], o)); }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < D; ++a) { red[wave][a] = lo[a]; red[wave][D + a] = hi[a]; }
  }
  if (threadIdx.x == 0) nopen = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int a = 0; a < D; ++a) {
      ubox[a] = fmin(fmin(red[0][a], red[1][a]), fmin(red[2][a], red[3][a]));
      ubox[D + a] = fmax(fmax(red[0][D + a], red[1][D + a]), fmax(red[2][D + a], red[3][D + a]));
    }
  }
  if (isU) atomicAdd(&nopen, 1);
  __syncthreads();
  bool covered = false;
  if (nopen > 0) {
    for (int base = 0; base < nruns; base += kRun) {
      // which of the next 256 S-runs can reach this run's U points at all
      if (threadIdx.x == 0) nlist = 0;
      __syncthreads();
      const int t = base + threadIdx.x;
      if (t < nruns) {
        const RunMeta m = meta[t];
        if (m.rmax >= 0.0) {
          double d2 = 0.0;
#pragma unroll
          for (int a = 0; a < D; ++a) {
            const double gap = fmax(0.0, fmax(m.lo[a] - ubox[D + a], ubox[a] - m.hi[a]));
            d2 += gap * gap;
          }
          const double reach = m.rmax * (1.0 + 1e-9) + 1e-6;
          if (d2 <= reach * reach) list[atomicAdd(&nlist, 1)] = t;
        }
      }
      __syncthreads();
      const int nl = nlist;
      for (int li = 0; li < nl; ++li) {
        const long long g = (run_lo + list[li]) * kRun + threadIdx.x;
        double r = -1.0, uc = 0.0;
        if (g < css.n_local) {
          uc = (double)W[g];
          if (uc >= 0.0) {
            r = L > 0 ? uc / L : 1e300;
            double x[D];
            cand_coords<D>(css, g, x);
#pragma unroll
            for (int a = 0; a < D; ++a) gx[threadIdx.x][a] = x[a];
          }
        }
        gr[threadIdx.x] = r;
        gucb[threadIdx.x] = uc;
        __syncthreads();
        if (isU && !covered) {
          for (int k = 0; k < kRun; ++k) {
            const double r = gr[k];
            if (r < 0.0) continue;
            double ss = 0.0;
#pragma unroll
            for (int a = 0; a < D; ++a) {
              const double df = (gx[k][a] - xh[a]) + 1e-8;       // x_g - x_h + 1e-8, models/GoOSE.py:71
              ss = (a == 0) ? df * df : ss + df * df;
            }
            const double rhi = r * (1.0 + 1e-12), rlo = r * (1.0 - 1e-12);
            if (ss > rhi * rhi) continue;
            if (ss < rlo * rlo || gucb[k] - L * sqrt(ss) >= 0.0) { covered = true; break; }
          }
        }
        __syncthreads();
      }
    }
  }
  if (h < cs.n_local) O[h] = covered;
}

// ---- GoOSE coverage on grids: power-distance transform --------------------------------------------------------
// "h is covered" means min over sources g of  ||x_g - x_h||^2 - r_g^2  <= 0  (r_g = ucb_c(g) / L): a min-plus
// transform of the sampled function F(g) = -r_g^2 (sources) / +inf (others) with parabolas, which separates by axis
// exactly like the Euclidean transform:  P_a(x) = min_t P_{a-1}(x + t e_a) + (h_a t)^2.
// The transform is evaluated in index space without the reference's "+1e-8" shift; |P| <= band is the zone where the
// shift and rounding could move the reference predicate across zero -- those h go to the exact recheck
// (k_goose_exact), everything else is decided by the sign.  band = 3 rmax eps bounds |dist - r| > eps on both sides:
// (dist - r)(dist + r) = P and dist + r <= 3 rmax whenever dist <= 2 rmax.
// Values above `band` can never lead to a covered verdict on a later axis, so the outward scans stop at
//   (h t)^2 - rmax^2 > band   (no source that far can matter)   and   (h t)^2 - rmax^2 >= best   (cannot improve).
struct PdtParams {
  double invL, rmax2, band;    // 1/L, (max source radius)^2, ambiguity band on P
  int L_positive;
};
__device__ __forceinline__ PdtParams pdt_params(const SweepScalars* sc, int c, const unsigned long long* Lkeys, int lidx, int d,
                                                double xscale) {
  PdtParams p;
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  p.L_positive = L > 0;
  p.invL = p.L_positive ? 1.0 / L : 0.0;
  const double rm = (sc->rmax_key[c] ? fmax(0.0, ord_val(sc->rmax_key[c])) : 0.0) * p.invL * (1.0 + 1e-12);
  const double eps = 1.01e-8 * sqrt((double)d) + 1e-14 * xscale + 1e-13 + 1e-10 * rm;
  p.rmax2 = rm * rm;
  p.band = 3.03 * rm * eps + eps * eps + 1e-13 * (xscale * xscale + p.rmax2);
  return p;
}

// Coarse bounds for the power transform (cells of kCoarse^d candidates, Fmin = smallest F in the cell).  A source of
// cell J and a candidate of cell I are between lo_t = (t-1) kCoarse + 1 (0 when t = 0) and hi_t = (t+1) kCoarse - 1
// steps apart along an axis, t = |I - J|, hence
//     min_J Fmin(J) + sum_a (h_a lo_t)^2  <=  P(x)  <=  min_J Fmin(J) + sum_a (h_a hi_t)^2      for every x in cell I.
// Both sides are separable min-plus transforms of the small array Fmin.  Lower side > band: nothing in the cell can be
// covered and its axis-0 values cannot matter either (they are >= P); upper side < -band: every U point is covered.
// Fmin of every coarse cell: one thread per cell walks its kCoarse^d candidates (axis 0 innermost); cells without a
// source get +inf.  Written to both bound arrays (they start from the same values).
template <typename T>
__global__ __launch_bounds__(256) void k_pdt_cell_min(const T* __restrict__ W, const CoarseGrid cg, long long nc,
                                                      const unsigned long long* Lkeys, int lidx, double* __restrict__ lo,
                                                      double* __restrict__ hi) {
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const double invL = L > 0 ? 1.0 / L : 0.0;
  for (long long cell = (long long)blockIdx.x * blockDim.x + threadIdx.x; cell < nc; cell += (long long)gridDim.x * blockDim.x) {
    // fine index of the cell origin and the cell's extent per axis
    long long f = cell, stride = 1, origin = 0, len[kMaxD], fstride[kMaxD], total = 1;
    for (int a = 0; a < cg.d; ++a) {
      const long long ci = f % cg.ccount[a];
      f /= cg.ccount[a];
      const long long i0 = ci * kCoarse;
      len[a] = cg.count[a] - i0 < kCoarse ? cg.count[a] - i0 : kCoarse;
      fstride[a] = stride;
      origin += i0 * stride;
      stride *= cg.count[a];
      total *= len[a];
    }
    double wmax = -1.0;
    for (long long t = 0; t < total; ++t) {
      long long u = t, g = origin;
      for (int a = 0; a < cg.d; ++a) {
        g += (u % len[a]) * fstride[a];
        u /= len[a];
      }
      const double w = (double)W[g];
      wmax = w > wmax ? w : wmax;
    }
    double v = kInfD;
    if (wmax >= 0.0) { const double r = wmax * invL; v = -(r * r); }
    lo[cell] = v;
    hi[cell] = v;
  }
}
// one axis of both coarse bound transforms (lower-bound costs on the first array, upper-bound costs on the second)
__global__ __launch_bounds__(256) void k_pdt_coarse_scan(const double* __restrict__ LoIn, double* __restrict__ LoOut,
                                                         const double* __restrict__ HiIn, double* __restrict__ HiOut, long long nc,
                                                         long long stride, int cnt, double h, const SweepScalars* sc, int c,
                                                         const unsigned long long* Lkeys, int lidx, int d, double xscale) {
  const PdtParams pp = pdt_params(sc, c, Lkeys, lidx, d, xscale);
  for (long long g2 = (long long)blockIdx.x * blockDim.x + threadIdx.x; g2 < 2 * nc; g2 += (long long)gridDim.x * blockDim.x) {
    const int upper = g2 >= nc;                          // first half of the index space: lower bounds, second half: upper
    const long long g = upper ? g2 - nc : g2;
    const double* Pin = upper ? HiIn : LoIn;
    const int ia = (int)((g / stride) % cnt);
    double best = kInfD;
    for (int t = 0; t < cnt; ++t) {
      const double steps_lo = t == 0 ? 0.0 : (double)((t - 1) * kCoarse + 1);
      const double steps = upper ? (double)((t + 1) * kCoarse - 1) : steps_lo;
      const double dl = h * steps_lo, dd = h * steps;
      const double floor_ = dl * dl - pp.rmax2;          // no source that far can bring any candidate below the band
      if (floor_ > pp.band || dd * dd - pp.rmax2 >= best) break;
      const bool lo_ok = ia - t >= 0, hi_ok = ia + t < cnt;
      if (!lo_ok && !hi_ok) break;
      const double c1 = lo_ok ? Pin[g - (long long)t * stride] : kInfD;
      const double c2 = hi_ok ? Pin[g + (long long)t * stride] : kInfD;
      const double cnd = (c1 < c2 ? c1 : c2) + dd * dd;
      best = cnd < best ? cnd : best;
    }
    (upper ? HiOut : LoOut)[g] = best;
  }
}
__device__ __forceinline__ long long coarse_cell(const CoarseGrid& cg, long long gg) {
  long long f = gg, cell = 0, cs = 1;
  for (int a = 0; a < cg.d; ++a) {
    const long long i = f % cg.count[a];
    f /= cg.count[a];
    cell += (i / kCoarse) * cs;
    cs *= cg.ccount[a];
  }
  return cell;
}

// largest source weight of every block of `blk` consecutive axis-0 positions (-inf when the block holds no source)
template <typename T>
__global__ __launch_bounds__(256) void k_block_max_w(const T* __restrict__ W, long long nt, int count0, int blk,
                                                     T* __restrict__ Bmax) {
  const int nblk = (count0 + blk - 1) / blk;
  const long long total = (nt / count0) * nblk;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long line = i / nblk;
    const int b = (int)(i % nblk);
    const int j1 = (b + 1) * blk < count0 ? (b + 1) * blk : count0;
    T m = (T)-INFINITY;
    for (int j = b * blk; j < j1; ++j) {
      const T v = W[line * count0 + j];
      m = v > m ? v : m;
    }
    Bmax[i] = m;
  }
}

// blocked axis-0 scan of one position i of a line: Wl = the line's source weights, Bl = largest weight per block of
// `blk` positions (global or LDS pointers).  `best` comes in as the position's own value.
template <typename TW>
__device__ __forceinline__ double pdt_axis0_point(const TW* Wl, const TW* Bl, int count0, int nblk, int i, double h0,
                                                  const PdtParams& pp, int blk, double best) {
  const int b0 = i / blk;
  auto scan_block = [&](int b) {
    const int j1 = (b + 1) * blk < count0 ? (b + 1) * blk : count0;
    for (int j = b * blk; j < j1; ++j) {
      const double wj = (double)Wl[j];
      if (wj >= 0.0) {
        const double dt = h0 * (double)(j > i ? j - i : i - j), r = wj * pp.invL;
        const double cnd = dt * dt - r * r;
        best = cnd < best ? cnd : best;
      }
    }
  };
  auto gap_of = [&](int b) { return b == b0 ? 0 : (b < b0 ? i - (b * blk + blk - 1) : b * blk - i); };
  auto bound_of = [&](int b, double e) {
    const double wb = (double)Bl[b];
    if (!(wb >= 0.0)) return kInfD;
    const double r = wb * pp.invL;
    return e - r * r;
  };
  double lb_min = bound_of(b0, 0.0);
  int b_min = b0;
  for (int k = 1; k < nblk; ++k) {
    bool any = false;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int b = side ? b0 + k : b0 - k;
      if (b < 0 || b >= nblk) continue;
      const double dg = h0 * (double)gap_of(b);
      const double e = dg * dg, floor_ = e - pp.rmax2;
      if (floor_ > pp.band || floor_ >= best || floor_ >= lb_min) continue;
      any = true;
      const double lb = bound_of(b, e);
      if (lb < lb_min) { lb_min = lb; b_min = b; }
    }
    if (!any) break;
  }
  if (lb_min < best) scan_block(b_min);
  if (b_min != b0 && bound_of(b0, 0.0) < best) scan_block(b0);
  for (int k = 1; k < nblk; ++k) {
    bool any = false;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int b = side ? b0 + k : b0 - k;
      if (b < 0 || b >= nblk) continue;
      const double dg = h0 * (double)gap_of(b);
      const double e = dg * dg, floor_ = e - pp.rmax2;
      if (floor_ > pp.band || floor_ >= best) continue;
      any = true;
      if (b != b_min && bound_of(b, e) < best) scan_block(b);
    }
    if (!any) break;
  }
  return best;
}

// The same pass with one workgroup per grid line: the line's weights (count0 <= 8192 doubles) and its block maxima sit in
// LDS, so the dependent loads of a position's scan cost an LDS round trip instead of an L2 / HBM one.
template <typename T>
__global__ __launch_bounds__(256) void k_pdt_axis0_lds(const T* __restrict__ W, long long nlines, int count0, double h0,
                                                       const SweepScalars* sc, int c, const unsigned long long* Lkeys, int lidx,
                                                       int d, double xscale, const CoarseGrid cg, const double* __restrict__ PcLo,
                                                       int blk, double* __restrict__ P) {
  extern __shared__ double lds_w[];            // [count0] weights as double | [nblk] block maxima
  const PdtParams pp = pdt_params(sc, c, Lkeys, lidx, d, xscale);
  const int nblk = (count0 + blk - 1) / blk;
  double* Wl = lds_w;
  double* Bl = lds_w + count0;
  for (long long line = blockIdx.x; line < nlines; line += gridDim.x) {
    const long long g0 = line * count0;
    __syncthreads();                           // the previous line's scans are done with the buffers
    for (int i = threadIdx.x; i < count0; i += blockDim.x) Wl[i] = (double)W[g0 + i];
    __syncthreads();
    for (int bb = threadIdx.x; bb < nblk; bb += blockDim.x) {
      const int j1 = (bb + 1) * blk < count0 ? (bb + 1) * blk : count0;
      double m = -INFINITY;
      for (int j = bb * blk; j < j1; ++j) m = Wl[j] > m ? Wl[j] : m;
      Bl[bb] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < count0; i += blockDim.x) {
      const long long g = g0 + i;
      if (cg.enabled && PcLo[coarse_cell(cg, g)] > pp.band) { P[g] = kInfD; continue; }
      const double w = Wl[i];
      double best = kInfD;
      if (w >= 0.0) { const double r = w * pp.invL; best = -(r * r); }
      P[g] = pdt_axis0_point((const double*)Wl, (const double*)Bl, count0, nblk, i, h0, pp, blk, best);
    }
  }
}

// axis 0, reading the source weights directly.  With Bmax (per-block largest weight = smallest F) the scan is blocked
// like the last-axis ones: a block whose bound (h gap)^2 - r_block^2 cannot beat the running minimum costs one load,
// the block with the smallest bound is visited first.  Same candidates and arithmetic as the step-by-step scan.
template <typename T>
__global__ __launch_bounds__(256) void k_pdt_axis0(const T* __restrict__ W, long long nt, int count0, double h0,
                                                   const SweepScalars* sc, int c, const unsigned long long* Lkeys, int lidx,
                                                   int d, double xscale, const CoarseGrid cg, const double* __restrict__ PcLo,
                                                   const T* __restrict__ Bmax, int blk, double* __restrict__ P) {
  const PdtParams pp = pdt_params(sc, c, Lkeys, lidx, d, xscale);
  const int nblk = (count0 + blk - 1) / blk;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < nt; g += (long long)gridDim.x * blockDim.x) {
    if (cg.enabled && PcLo[coarse_cell(cg, g)] > pp.band) { P[g] = kInfD; continue; }
    const int i = (int)(g % count0);
    const double w = (double)W[g];
    double best = kInfD;
    if (w >= 0.0) { const double r = w * pp.invL; best = -(r * r); }
    if (Bmax == nullptr) {
      for (int t = 1; t < count0; ++t) {
        const double dt = h0 * (double)t;
        const double e = dt * dt;
        const double floor_ = e - pp.rmax2;
        if (floor_ > pp.band || floor_ >= best) break;
        const bool lo_ok = i - t >= 0, hi_ok = i + t < count0;
        if (!lo_ok && !hi_ok) break;
        const double w1 = lo_ok ? (double)W[g - t] : -1.0;
        const double w2 = hi_ok ? (double)W[g + t] : -1.0;
        const double wm = fmax(w1, w2);                 // the larger radius wins at equal distance
        if (wm >= 0.0) {
          const double r = wm * pp.invL;
          const double cnd = e - r * r;
          best = cnd < best ? cnd : best;
        }
      }
      P[g] = best;
      continue;
    }
    best = pdt_axis0_point(W + (g - i), Bmax + (g / count0) * nblk, count0, nblk, i, h0, pp, blk, best);
    P[g] = best;
  }
}

__device__ __forceinline__ double pdt_scan_point(const double* __restrict__ Pin, long long g, long long stride, int cnt, int ia,
                                                 double h, const PdtParams& pp, bool early_accept) {
  double best = Pin[g];
  for (int t = 1; t < cnt; ++t) {
    const double dt = h * (double)t;
    const double e = dt * dt;
    const double floor_ = e - pp.rmax2;
    if (floor_ > pp.band || floor_ >= best || (early_accept && best < -pp.band)) break;
    const bool lo_ok = ia - t >= 0, hi_ok = ia + t < cnt;
    if (!lo_ok && !hi_ok) break;
    const double c1 = lo_ok ? Pin[g - (long long)t * stride] : kInfD;
    const double c2 = hi_ok ? Pin[g + (long long)t * stride] : kInfD;
    const double cnd = (c1 < c2 ? c1 : c2) + e;
    best = cnd < best ? cnd : best;
  }
  return best;
}

// power-transform form of the blocked last-axis scan (values >= -rmax2; exits as pdt_scan_point), same visiting order
__device__ __forceinline__ double pdt_scan_blocked(const double* __restrict__ Pin, const double* __restrict__ Bmin, long long p,
                                                   long long stride, int cnt, int ia, double h, const PdtParams& pp, int blk) {
  double best = Pin[(long long)ia * stride + p];
  const int nblk = (cnt + blk - 1) / blk, b0 = ia / blk;
  auto scan_block = [&](int b) {
    const int j1 = (b + 1) * blk < cnt ? (b + 1) * blk : cnt;
    for (int j = b * blk; j < j1; ++j) {
      const double dt = h * (double)(j > ia ? j - ia : ia - j);
      const double cnd = Pin[(long long)j * stride + p] + dt * dt;
      best = cnd < best ? cnd : best;
    }
  };
  auto gap_of = [&](int b) { return b == b0 ? 0 : (b < b0 ? ia - (b * blk + blk - 1) : b * blk - ia); };
  double lb_min = Bmin[(long long)b0 * stride + p];
  int b_min = b0;
  for (int k = 1; k < nblk; ++k) {
    bool any = false;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int b = side ? b0 + k : b0 - k;
      if (b < 0 || b >= nblk) continue;
      const double dg = h * (double)gap_of(b);
      const double e = dg * dg, floor_ = e - pp.rmax2;
      if (floor_ > pp.band || floor_ >= best || floor_ >= lb_min) continue;
      any = true;
      const double lb = Bmin[(long long)b * stride + p] + e;
      if (lb < lb_min) { lb_min = lb; b_min = b; }
    }
    if (!any) break;
  }
  if (lb_min < best) scan_block(b_min);
  if (best < -pp.band) return best;
  if (b_min != b0 && Bmin[(long long)b0 * stride + p] < best) scan_block(b0);
  for (int k = 1; k < nblk && !(best < -pp.band); ++k) {
    bool any = false;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int b = side ? b0 + k : b0 - k;
      if (b < 0 || b >= nblk) continue;
      const double dg = h * (double)gap_of(b);
      const double e = dg * dg, floor_ = e - pp.rmax2;
      if (floor_ > pp.band || floor_ >= best) continue;
      any = true;
      if (b != b_min && Bmin[(long long)b * stride + p] + e < best) scan_block(b);
    }
    if (!any) break;
  }
  return best;
}

__global__ __launch_bounds__(256) void k_pdt_scan(const double* __restrict__ Pin, double* __restrict__ Pout, long long nt,
                                                  long long stride, int cnt, double h, const SweepScalars* sc, int c,
                                                  const unsigned long long* Lkeys, int lidx, int d, double xscale) {
  const PdtParams pp = pdt_params(sc, c, Lkeys, lidx, d, xscale);
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < nt; g += (long long)gridDim.x * blockDim.x) {
    const int ia = (int)((g / stride) % cnt);
    Pout[g] = pdt_scan_point(Pin, g, stride, cnt, ia, h, pp, false);
  }
}

// last axis + verdict for the own U points (window offset goff); ambiguous ones are listed for the exact recheck
__global__ __launch_bounds__(256) void k_pdt_decide(const double* __restrict__ Pin, long long n, long long goff, long long stride,
                                                    int cnt, double h, int d, double xscale, const uint8_t* __restrict__ U,
                                                    const unsigned long long* Lkeys, int lidx, SweepScalars* sc, int c,
                                                    uint8_t* __restrict__ O, long long* __restrict__ amb, const CoarseGrid cg,
                                                    const double* __restrict__ PcLo, const double* __restrict__ PcHi,
                                                    const double* __restrict__ Bmin, int blk) {
  const PdtParams pp = pdt_params(sc, c, Lkeys, lidx, d, xscale);
  const bool anyS = sc->count_S > 0;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    uint8_t out = 0;
    if (U[g]) {
      if (!pp.L_positive) {
        out = anyS;                                  // radius unbounded: any source covers (ucb_c >= 0 on every source)
      } else {
        const long long gg = goff + g;
        if (cg.enabled) {
          const long long cell = coarse_cell(cg, gg);
          if (PcHi[cell] < -pp.band) { O[g] = 1; continue; }     // covered wherever it sits in its cell
          if (PcLo[cell] > pp.band) { O[g] = 0; continue; }      // out of every source's reach
        }
        const int ia = (int)((gg / stride) % cnt);
        const double best = cnt <= 1 ? Pin[gg]
                            : Bmin  ? pdt_scan_blocked(Pin, Bmin, gg % stride, stride, cnt, ia, h, pp, blk)
                                    : pdt_scan_point(Pin, gg, stride, cnt, ia, h, pp, true);
        if (best < -pp.band) out = 1;
        else if (best <= pp.band) {
          const long long slot = (long long)atomicAdd((unsigned long long*)&sc->n_amb, 1ull);
          amb[slot] = g;
        }
      }
    }
    O[g] = out;
  }
}

// exact recheck of the listed U points: the reference predicate against every source inside the index box that the
// largest radius can reach.  cs: the own candidates (h); css / W: the source candidates (own range or whole grid)
template <typename T, int D>
__global__ __launch_bounds__(256) void k_goose_exact(const CandSpec cs, const CandSpec css, const T* __restrict__ W,
                                                     const unsigned long long* Lkeys, int lidx, SweepScalars* sc, int c,
                                                     const long long* __restrict__ amb, uint8_t* __restrict__ O) {
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const long long namb = sc->n_amb;
  const double rm = (L > 0 && sc->rmax_key[c]) ? fmax(0.0, ord_val(sc->rmax_key[c])) / L : 0.0;
  // one listed point can own a box as large as the grid: its box is cut into kParts slices, one workgroup each
  constexpr int kParts = 64;
  for (long long wi = blockIdx.x; wi < namb * kParts; wi += gridDim.x) {
    const long long qi = wi / kParts;
    const int part = (int)(wi % kParts);
    const long long hl = amb[qi];
    double xh[D];
    cand_coords<D>(cs, hl, xh);
    long long lo[D], len[D], stridea[D];
    long long f = cs.first + hl, total = 1, sa = 1;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      lo[a] = 0; len[a] = 1; stridea[a] = 0;
      if (a < cs.d) {
        const long long cnt = cs.count[a];
        const long long ih = f % cnt;
        f /= cnt;
        long long R = cnt;
        if (cs.step[a] > 0) {
          const double rr = (rm * (1.0 + 1e-9) + 1e-7) / cs.step[a];
          R = rr < (double)cnt ? (long long)ceil(rr) + 1 : cnt;
        }
        const long long l0 = ih - R > 0 ? ih - R : 0, h0 = ih + R < cnt - 1 ? ih + R : cnt - 1;
        lo[a] = l0; len[a] = h0 - l0 + 1; stridea[a] = sa;
        total *= len[a];
        sa *= cnt;
      }
    }
    int found = 0;
    const long long chunk = (total + kParts - 1) / kParts;
    const long long t1 = (part + 1) * chunk < total ? (part + 1) * chunk : total;
    for (long long t = part * chunk + threadIdx.x; t < t1 && !found; t += blockDim.x) {
      long long u = t, gg = 0;
      double xg[D];
#pragma unroll
      for (int a = 0; a < D; ++a) {
        xg[a] = 0.0;
        if (a < cs.d) {
          const long long ia = lo[a] + u % len[a];
          u /= len[a];
          gg += ia * stridea[a];
          const long long cnt = cs.count[a];
          xg[a] = (ia == cnt - 1 && cnt > 1) ? cs.hi[a] : __dadd_rn(cs.lo[a], __dmul_rn((double)ia, cs.step[a]));
        }
      }
      const long long gl = gg - css.first;
      if (gl >= 0 && gl < css.n_local) {
        const double w = (double)W[gl];
        if (w >= 0.0 && lipschitz_pair<D>(xg, xh, cs.d, w, L)) found = 1;
      }
    }
    found = __syncthreads_or(found);
    if (threadIdx.x == 0 && found) O[hl] = 1;
    __syncthreads();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) sc->n_amb_total += namb;
}

// trust-region mask: S and ||x - x_0||_2 <= r, the norm evaluated as sqrt(sum (x - x_0)^2) (models/GP_TR.py:49)
template <int D>
__global__ void k_ball_mask(const CandSpec cs, long long n, const uint8_t* __restrict__ S, const double* __restrict__ x0,
                            double r, uint8_t* __restrict__ out) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    uint8_t m = 0;
    if (S[g]) {
      double x[D];
      cand_coords<D>(cs, g, x);
      double ss = 0.0;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        if (a < cs.d) {
          const double df = x[a] - x0[a];
          ss = (a == 0) ? df * df : ss + df * df;
        }
      }
      m = sqrt(ss) <= r;
    }
    out[g] = m;
  }
}

// value arrays for the arg-reductions of the GoOSE sweep
template <typename T>
__global__ void k_lcb0(const T* __restrict__ mean0, const T* __restrict__ var0, long long n, T b, T* __restrict__ out) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    T lcb, ucb;
    lcb_ucb(mean0[g], var0[g], b, lcb, ucb);
    out[g] = lcb;
  }
}
// Euclidean distance to the target, as scipy.spatial.distance.cdist computes it (models/GoOSE.py:117)
template <typename T, int D>
__global__ void k_dist_to(const CandSpec cs, long long n, const double* __restrict__ target, T* __restrict__ out) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
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
    out[g] = (T)sqrt(ss);
  }
}

// ---- host orchestration -------------------------------------------------------------------------------
static int reduce_blocks(const sbo_ctx* c) {
  const long long n = c->cs.n_local;
  return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, (long long)c->n_cu * 4));   // (per-block reduction tails cost more than extra grid-stride turns: x4 measured best)
}

template <typename T>
static int sweep_common_front(sbo_ctx* c, const sbo_sweep_opts* o) {
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  int rc;
  long long npad_shard = n;                  // ranks > 1: the U mask is all-gathered with the largest shard's size
  for (size_t r = 0; r + 1 < c->first_of.size() && c->world > 1 && c->sharded; ++r)
    npad_shard = std::max(npad_shard, c->first_of[r + 1] - c->first_of[r]);
  if ((rc = ensure(c->maskS, (size_t)n))) return rc;
  if ((rc = ensure(c->maskU, (size_t)npad_shard))) return rc;
  if ((rc = ensure(c->maskM, (size_t)n))) return rc;
  if ((rc = ensure(c->maskG, (size_t)n * std::max(1, q - 1)))) return rc;
  if ((rc = ensure(c->scal, sizeof(SweepScalars)))) return rc;
  const int nb = reduce_blocks(c);
  if ((rc = ensure(c->partial, sizeof(Best) * (size_t)nb))) return rc;
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  hipLaunchKernelGGL(k_init_scalars, dim3(1), dim3(64), 0, c->stream, sc);
  if (n > 0)
    hipLaunchKernelGGL(k_classify<T>, dim3((unsigned)std::max(1, std::min(nb, c->n_cu * 2))), dim3(256), 0, c->stream,
                       (const T*)c->mean.p, (const T*)c->var.p, n, q,
                       (T)o->b, (uint8_t*)c->maskS.p, (uint8_t*)c->maskU.p, sc);
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

template <typename T, int D>
static int launch_exact(sbo_ctx* c, const sbo_sweep_opts* o, int cidx, int lidx, uint8_t* G) {
  const long long n = c->cs.n_local;
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  const T* mean_c = (const T*)c->mean.p + (size_t)cidx * n;
  const T* var_c = (const T*)c->var.p + (size_t)cidx * n;
  CandSpec csU = c->cs;          // the witness set: every candidate of every rank
  const uint8_t* Uall = (const uint8_t*)c->maskU.p;
  if (c->world > 1) {
    csU.first = 0;
    csU.n_local = c->grid_total;
    Uall = (const uint8_t*)c->Ufull.p;
  }
  hipLaunchKernelGGL((k_expander_exact<T, D>), dim3(1024), dim3(256), 0, c->stream, c->cs, csU, mean_c, var_c, (T)o->b,
                     Uall, (const unsigned long long*)c->Lmax.p, lidx, sc,
                     (const long long*)c->amb.p, G);
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

// G_c for constraint cidx (1..q-1) into G[n]
template <typename T>
static int expander_set(sbo_ctx* c, const sbo_sweep_opts* o, int cidx, uint8_t* G) {
  const long long n = c->cs.n_local;
  if (n == 0) return SBO_OK;
  const int q = c->mc.q;
  const int lidx = o->reference_quirk_L_index ? q - 1 : cidx;   // models/SafeOpt.py:110 (loop-leaked i)
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  const T* mean_c = (const T*)c->mean.p + (size_t)cidx * n;
  const T* var_c = (const T*)c->var.p + (size_t)cidx * n;
  const int nb = reduce_blocks(c);
  int rc;
  if ((rc = ensure(c->amb, sizeof(long long) * (size_t)n))) return rc;
  hipLaunchKernelGGL(k_reset_amb, dim3(1), dim3(1), 0, c->stream, sc);
  const int d_ = c->cs.d;
  long long plane = 1;                       // candidates per step of the slowest axis
  for (int a = 0; a < d_ - 1; ++a) plane *= c->cs.count[a];
  const bool plane_aligned = c->cs.kind == 1 && c->cs.first % plane == 0 && n % plane == 0;
  if (c->cs.kind == 1 && plane_aligned) {
    const int d = d_;
    // Window of the transform: this rank's hyper-planes of the slowest axis plus, with ranks > 1, a halo of
    // ceil(cap / h) planes on either side taken from the all-gathered U mask (cap = largest radius that can matter,
    // from the keys of collective C1).  Witnesses further away cannot change a verdict, so the window is exact.
    const long long planes_total = c->world > 1 ? c->cs.count[d - 1] : n / plane;
    long long p0 = c->world > 1 ? c->cs.first / plane : 0, p1 = p0 + n / plane;
    const long long own0 = p0;
    if (c->world > 1) {
      if ((rc = sweep_exchange_wait(c))) return rc;
      double L, rmax = 0.0;
      memcpy(&L, &c->h_c1[1 + lidx], 8);
      if (c->h_c1[1 + kMaxQ + cidx]) rmax = ord_val(c->h_c1[1 + kMaxQ + cidx]);
      const double hl = d >= 2 ? c->cs.step[d - 1] : c->cs.step[0];
      long long H = planes_total;
      if (L > 0 && hl > 0) {
        const double cap = rmax / L * 1.000001 + 1e-6;
        const double hp = std::ceil(cap / hl) + 2.0;
        if (hp < (double)planes_total) H = (long long)hp;
      }
      p0 = std::max(0ll, p0 - H);
      p1 = std::min(planes_total, p1 + H);
    }
    const long long wplanes = p1 - p0;
    const long long nt = wplanes * plane;
    const long long goff = (own0 - p0) * plane;                 // own candidates start here inside the window
    const uint8_t* Uall = c->world > 1 ? (const uint8_t*)c->Ufull.p + p0 * plane : (const uint8_t*)c->maskU.p;
    if ((rc = ensure(c->dist2, sizeof(double) * (size_t)nt))) return rc;
    if (d > 2 && (rc = ensure(c->dist2b, sizeof(double) * (size_t)nt))) return rc;
    const int count0 = d >= 2 ? (int)c->cs.count[0] : (int)nt;  // d == 1: the window is one line
    const long long nlines = nt / count0;
    hipLaunchKernelGGL(k_edt_axis0, dim3((unsigned)((nlines + 3) / 4)), dim3(256), 0, c->stream, Uall, nlines, count0,
                       c->cs.step[0], (double*)c->dist2.p);
    double* din = (double*)c->dist2.p;
    double* dout = (double*)c->dist2b.p;
    long long stride = count0;
    for (int a = 1; a < d - 1; ++a) {
      hipLaunchKernelGGL(k_edt_scan, dim3((unsigned)std::min<long long>((nt + 255) / 256, 1 << 20)), dim3(256), 0, c->stream,
                         (const double*)din, dout, nt, stride, (int)c->cs.count[a], c->cs.step[a],
                         (const SweepScalars*)sc, cidx, (const unsigned long long*)c->Lmax.p, lidx, 0);
      std::swap(din, dout);
      stride *= c->cs.count[a];
    }
    // coarse transform of the window (uncapped, tiny): lets most candidates decide without the per-candidate scan
    CoarseGrid cg;
    memset(&cg, 0, sizeof(cg));
    cg.d = d;
    bool coarse_ok = d >= 2 && nt >= (1ll << 16);
    long long nc = 1;
    double h2 = 0.0;
    for (int a = 0; a < d; ++a) {
      cg.count[a] = a == d - 1 ? wplanes : c->cs.count[a];
      cg.ccount[a] = (cg.count[a] + kCoarse - 1) / kCoarse;
      nc *= cg.ccount[a];
      h2 += c->cs.step[a] * c->cs.step[a];
      if (cg.count[a] < 4 * kCoarse) coarse_ok = false;
    }
    if (coarse_ok) {
      cg.enabled = 1;
      cg.delta = (kCoarse - 1) * std::sqrt(h2) * (1.0 + 1e-9);
      if ((rc = ensure(c->coarse, (size_t)nc * (1 + 2 * sizeof(double)) + 64))) return rc;
      double* dc0 = (double*)c->coarse.p;
      double* dc1 = dc0 + nc;
      uint8_t* Uc = (uint8_t*)(dc1 + nc);
      SBO_HIP(hipMemsetAsync(Uc, 0, (size_t)nc, c->stream));
      hipLaunchKernelGGL(k_coarsen_mask, dim3((unsigned)std::min<long long>((nt + 255) / 256, 1 << 16)), dim3(256), 0, c->stream,
                         Uall, nt, cg, Uc);
      const int cc0 = (int)cg.ccount[0];
      const long long clines = nc / cc0;
      hipLaunchKernelGGL(k_edt_axis0, dim3((unsigned)((clines + 3) / 4)), dim3(256), 0, c->stream, (const uint8_t*)Uc, clines, cc0,
                         c->cs.step[0] * kCoarse, dc0);
      long long cstride = cc0;
      for (int a = 1; a < d; ++a) {
        hipLaunchKernelGGL(k_edt_scan, dim3((unsigned)std::min<long long>((nc + 255) / 256, 1 << 16)), dim3(256), 0, c->stream,
                           (const double*)dc0, dc1, nc, cstride, (int)cg.ccount[a], c->cs.step[a] * kCoarse,
                           (const SweepScalars*)sc, cidx, (const unsigned long long*)c->Lmax.p, lidx, 1);
        std::swap(dc0, dc1);
        cstride *= cg.ccount[a];
      }
      cg.Dc = dc0;
    }
    double xscale = 0.0;
    for (int a = 0; a < d; ++a) xscale = std::max(xscale, std::max(std::fabs(c->cs.lo[a]), std::fabs(c->cs.hi[a])));
    const int last_cnt = d >= 2 ? (int)wplanes : 1;
    const double last_h = d >= 2 ? c->cs.step[d - 1] : 0.0;
    {
      const double* bmin = nullptr;
      const int blk = last_cnt >= 8192 ? 64 : 32;
      if (d >= 2 && last_cnt >= 4 * blk && c->scan_blocks) {
        const long long nb_ = (long long)((last_cnt + blk - 1) / blk) * stride;
        if ((rc = ensure(c->blockmin, sizeof(double) * (size_t)nb_))) return rc;
        hipLaunchKernelGGL(k_block_min, dim3((unsigned)std::min<long long>((nb_ + 255) / 256, 1 << 20)), dim3(256), 0, c->stream,
                           (const double*)din, stride, last_cnt, blk, (double*)c->blockmin.p);
        bmin = (const double*)c->blockmin.p;
      }
      long long* slist = nullptr;
      if (bmin && blk <= 64 && c->scan_waves) {
        if ((rc = ensure(c->scanlist, sizeof(long long) * (size_t)n))) return rc;
        slist = (long long*)c->scanlist.p;
      }
      hipLaunchKernelGGL((k_edt_decide<T>), dim3((unsigned)std::min<long long>((n + 255) / 256, 1 << 20)), dim3(256), 0,
                         c->stream, (const double*)din, n, goff, d >= 2 ? stride : 1, last_cnt, last_h, d, xscale, mean_c, var_c,
                         (T)o->b, (const uint8_t*)c->maskS.p, (const unsigned long long*)c->Lmax.p, lidx, sc, G,
                         (long long*)c->amb.p, cg, bmin, blk, slist);
      if (slist) {
        if (blk <= 32)
          hipLaunchKernelGGL((k_edt_scan_list<T, 32>), dim3(2048), dim3(256), 0, c->stream, (const double*)din, goff, stride, last_cnt,
                             last_h, d, xscale, mean_c, var_c, (T)o->b, (const unsigned long long*)c->Lmax.p, lidx, sc, G,
                             (long long*)c->amb.p, bmin, blk, (const long long*)slist);
        else
          hipLaunchKernelGGL((k_edt_scan_list<T, 64>), dim3(2048), dim3(256), 0, c->stream, (const double*)din, goff, stride, last_cnt,
                             last_h, d, xscale, mean_c, var_c, (T)o->b, (const unsigned long long*)c->Lmax.p, lidx, sc, G,
                             (long long*)c->amb.p, bmin, blk, (const long long*)slist);
      }
    }
  } else {
    // explicit candidate lists, and grid ranges that are not whole hyper-planes: exhaustive evaluation
    if (n > (1ll << 17))
      return fail(SBO_E_UNSUPPORTED, "expander sets need a grid of whole hyper-planes, or at most 131072 candidates (exhaustive)");
    if (c->world > 1)
      return fail(SBO_E_UNSUPPORTED, "expander sets on explicit candidate lists are single-rank");
    hipLaunchKernelGGL(k_list_safe, dim3(nb), dim3(256), 0, c->stream, (const uint8_t*)c->maskS.p, n, sc, G,
                       (long long*)c->amb.p);
  }
  SBO_HIP(hipGetLastError());
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

// (ranks > 1) C1: global u*, L and radius keys; C2: whole-grid U mask
template <typename T>
static int sweep_exchange_front(sbo_ctx* c, const sbo_sweep_opts* o, bool need_U) {
  const int q = c->mc.q;
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  if (c->world <= 1) return SBO_OK;   // (the radius keys, max over S of ucb_c, come out of k_classify)
  int rc;
  if (!c->sharded && q > 1 && need_U)
    return fail(SBO_E_INVALID, "multi-rank sweeps with constraints need sbo_candidates_grid_sharded");
  if ((rc = ensure(c->xch, sizeof(double) * (size_t)(c->world * kC3Row + 64)))) return rc;
  unsigned long long* kb = (unsigned long long*)c->xch.p;
  hipLaunchKernelGGL(k_pack_c1, dim3(1), dim3(64), 0, c->stream, (const SweepScalars*)sc, (const unsigned long long*)c->Lmax.p, kb);
  if ((rc = comm_allreduce_max_u64(c, kb, 1 + 2 * kMaxQ))) return rc;
  hipLaunchKernelGGL(k_unpack_c1, dim3(1), dim3(64), 0, c->stream, sc, (unsigned long long*)c->Lmax.p, (const unsigned long long*)kb);
  // the host needs the global L and radius keys to size the halo of the expander transform: the read-back goes to
  // pinned memory and is waited for only where the window is computed (sweep_exchange_wait), so the mask all-gather
  // and the minimiser kernels are already queued behind it and the GPU does not idle through the round trip
  SBO_HIP(hipMemcpyAsync(c->h_c1, kb, sizeof(unsigned long long) * (1 + 2 * kMaxQ), hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipEventRecord(c->ev[5], c->stream));
  c->c1_pending = true;
  if (need_U && q > 1) {
    // the mask travels as bits (one ballot word per 64 candidates): 8x fewer bytes on the links than the byte mask
    long long maxlocal = 0;
    for (int r = 0; r < c->world; ++r) maxlocal = std::max(maxlocal, c->first_of[r + 1] - c->first_of[r]);
    const long long words = (maxlocal + 63) / 64;
    if ((rc = ensure(c->gather, sizeof(unsigned long long) * (size_t)words * (c->world + 1)))) return rc;
    if ((rc = ensure(c->Ufull, (size_t)c->grid_total))) return rc;
    unsigned long long* sendw = (unsigned long long*)c->gather.p;
    unsigned long long* recvw = sendw + words;
    hipLaunchKernelGGL(k_pack_bits, dim3((unsigned)std::min<long long>((words + 3) / 4, 1 << 16)), dim3(256), 0, c->stream,
                       (const uint8_t*)c->maskU.p, c->cs.n_local, words, sendw);
    if ((rc = comm_allgather_bytes(c, sendw, recvw, sizeof(unsigned long long) * (size_t)words))) return rc;
    hipLaunchKernelGGL(k_unpack_shards, dim3((unsigned)std::min<long long>((c->grid_total + 255) / 256, 1 << 16)), dim3(256), 0,
                       c->stream, (const unsigned long long*)recvw, words, c->world, (const long long*)c->shard_first.p,
                       c->grid_total, (uint8_t*)c->Ufull.p);
  }
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

static int sweep_exchange_wait(sbo_ctx* c) {
  if (c->c1_pending) {
    SBO_HIP(hipEventSynchronize(c->ev[5]));
    c->c1_pending = false;
  }
  return SBO_OK;
}

// C3 + host merge: every rank's slots and counters -> global ones.  slot_is_max[i] selects arg-max / arg-min.
static int sweep_exchange_back(sbo_ctx* c, SweepScalars& h, const bool* slot_is_max, unsigned long long* Lk = nullptr,
                               hipEvent_t done_ev = nullptr) {
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  // (the Lipschitz keys ride in the same read-back: one synchronisation per sweep)
  if (Lk) SBO_HIP(hipMemcpyAsync(Lk, c->Lmax.p, sizeof(unsigned long long) * kMaxQ, hipMemcpyDeviceToHost, c->stream));
  if (c->world <= 1) {
    SBO_HIP(hipMemcpyAsync(&h, sc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    if (done_ev) SBO_HIP(hipEventRecord(done_ev, c->stream));
    SBO_HIP(hipStreamSynchronize(c->stream));
    return SBO_OK;
  }
  double* buf = (double*)c->xch.p + 64;
  hipLaunchKernelGGL(k_pack_c3, dim3(1), dim3(256), 0, c->stream, (const SweepScalars*)sc, buf, c->world, c->rank);
  int rc;
  if ((rc = comm_allreduce_sum_f64(c, buf, c->world * kC3Row))) return rc;
  std::vector<double> rows((size_t)c->world * kC3Row);
  SBO_HIP(hipMemcpyAsync(&h, sc, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipMemcpyAsync(rows.data(), buf, sizeof(double) * rows.size(), hipMemcpyDeviceToHost, c->stream));
  if (done_ev) SBO_HIP(hipEventRecord(done_ev, c->stream));
  SBO_HIP(hipStreamSynchronize(c->stream));
  c->c1_pending = false;                       // (the whole stream has drained)
  h.count_S = h.count_U = h.count_M = h.n_amb_total = 0;
  for (int t = 0; t < kMaxQ; ++t) h.count_set[t] = 0;
  for (int t = 0; t < kArgSlots; ++t) { h.arg_idx[t] = -1; h.arg_val[t] = 0.0; }
  for (int r = 0; r < c->world; ++r) {
    const double* row = &rows[(size_t)r * kC3Row];
    for (int t = 0; t < kArgSlots; ++t) {
      const long long idx = (long long)row[kArgSlots + t];
      if (idx < 0) continue;
      const double v = row[t];
      const bool mx = slot_is_max[t];
      const bool take = h.arg_idx[t] < 0 || (mx ? v > h.arg_val[t] : v < h.arg_val[t]) || (v == h.arg_val[t] && idx < h.arg_idx[t]);
      if (take) { h.arg_val[t] = v; h.arg_idx[t] = idx; }
    }
    h.count_S += (long long)row[2 * kArgSlots + 0];
    h.count_U += (long long)row[2 * kArgSlots + 1];
    h.count_M += (long long)row[2 * kArgSlots + 2];
    h.n_amb_total += (long long)row[2 * kArgSlots + 3];
    for (int t = 0; t < kMaxQ; ++t) h.count_set[t] += (long long)row[2 * kArgSlots + 4 + t];
  }
  return SBO_OK;
}

template <typename T>
static int sweep_safeopt_t(sbo_ctx* c, const sbo_sweep_opts* o, sbo_safeopt_result* res) {
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  int rc;
  SBO_HIP(hipEventRecord(c->ev[0], c->stream));
  const bool reuse = o->posterior_ready && c->posterior_valid;
  if (!reuse && (rc = sbo_posterior_enqueue_(c))) return rc;
  SBO_HIP(hipEventRecord(c->ev[1], c->stream));
  if ((rc = sweep_common_front<T>(c, o))) return rc;
  if ((rc = sweep_exchange_front<T>(c, o, true))) return rc;
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  const int nb = reduce_blocks(c);
  if (n > 0)
    hipLaunchKernelGGL((k_minimizer<T>), dim3(nb), dim3(256), 0, c->stream, (const T*)c->mean.p, (const T*)c->var.p, n,
                       (long long)c->cs.first, (T)o->b, (const uint8_t*)c->maskS.p, (uint8_t*)c->maskM.p, sc,
                       (Best*)c->partial.p);
  hipLaunchKernelGGL((k_arg_final<true>), dim3(1), dim3(256), 0, c->stream, (const Best*)c->partial.p, n > 0 ? nb : 0, sc, 0);
  SBO_HIP(hipEventRecord(c->ev[2], c->stream));
  for (int cc = 1; cc < q; ++cc) {
    uint8_t* G = (uint8_t*)c->maskG.p + (size_t)(cc - 1) * n;
    if ((rc = expander_set<T>(c, o, cc, G))) return rc;
  }
  SBO_HIP(hipEventRecord(c->ev[3], c->stream));
  for (int cc = 1; cc < q; ++cc) {
    const uint8_t* G = (const uint8_t*)c->maskG.p + (size_t)(cc - 1) * n;
    if (n > 0)
      hipLaunchKernelGGL((k_arg_masked<T, true>), dim3(nb), dim3(256), 0, c->stream, (const T*)c->var.p, G, n,
                         (long long)c->cs.first, &sc->count_set[cc - 1], (Best*)c->partial.p);
    hipLaunchKernelGGL((k_arg_final<true>), dim3(1), dim3(256), 0, c->stream, (const Best*)c->partial.p, n > 0 ? nb : 0,
                       sc, cc);
  }
  SBO_HIP(hipGetLastError());
  SweepScalars h;
  bool is_max[kArgSlots];
  for (int t = 0; t < kArgSlots; ++t) is_max[t] = true;
  unsigned long long Lk[kMaxQ];
  if ((rc = sweep_exchange_back(c, h, is_max, Lk, c->ev[4]))) return rc;
  c->masks_valid = true;
  c->last_sweep = 1;

  float t01 = 0, t12 = 0, t23 = 0, t34 = 0, t04 = 0;
  SBO_HIP(hipEventElapsedTime(&t01, c->ev[0], c->ev[1]));
  SBO_HIP(hipEventElapsedTime(&t12, c->ev[1], c->ev[2]));
  SBO_HIP(hipEventElapsedTime(&t23, c->ev[2], c->ev[3]));
  SBO_HIP(hipEventElapsedTime(&t34, c->ev[3], c->ev[4]));
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

  memset(res, 0, sizeof(*res));
  res->count_S = h.count_S;
  res->count_U = h.count_U;
  res->count_M = h.count_M;
  res->n_exact_rechecks = h.n_amb_total;
  for (int i = 0; i < q; ++i) memcpy(&res->L[i], &Lk[i], 8);
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
  return SBO_OK;
}

template <typename T, int D>
static int goose_sets(sbo_ctx* c, const sbo_sweep_opts* o, int cidx, const uint8_t* src, uint8_t* O) {
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  const int lidx = o->reference_quirk_L_index ? q - 1 : cidx;   // models/GoOSE.py:100 (loop-leaked i)
  const T* mean_c = (const T*)c->mean.p + (size_t)cidx * n;
  const T* var_c = (const T*)c->var.p + (size_t)cidx * n;
  int rc;
  long long maxlocal = n;
  for (int r = 0; r < c->world && c->world > 1; ++r) maxlocal = std::max(maxlocal, c->first_of[r + 1] - c->first_of[r]);
  if ((rc = ensure(c->gw, sizeof(T) * (size_t)std::max<long long>(maxlocal, 1)))) return rc;
  if (n > 0)
    hipLaunchKernelGGL((k_goose_weights<T>), dim3(reduce_blocks(c)), dim3(256), 0, c->stream, mean_c, var_c, n, (T)o->b, src,
                       (T*)c->gw.p);
  CandSpec css = c->cs;                 // the source candidates
  const T* W = (const T*)c->gw.p;
  long long run_lo = 0, run_hi = (n + kRun - 1) / kRun;
  long long win_p0 = 0, win_p1 = 0;     // ranks > 1: window of hyper-planes holding every source that can matter
  bool w_is_window = false;             // ranks > 1, slab exchange: W holds exactly the window [win_p0, win_p1)
  if (c->world > 1) {
    // Sources of the other ranks that can reach this shard lie within H hyper-planes of it (H from the largest source
    // radius, keys of collective C1 -- exact, as for the expanders).
    const int d = c->cs.d;
    long long plane = 1;
    for (int a = 0; a < d - 1; ++a) plane *= c->cs.count[a];
    const long long planes_total = c->cs.count[d - 1];
    long long p0 = c->cs.first / plane, p1 = (c->cs.first + n + plane - 1) / plane;
    const long long own0 = p0, own1 = p1;
    if ((rc = sweep_exchange_wait(c))) return rc;
    double L, rmax = 0.0;
    memcpy(&L, &c->h_c1[1 + lidx], 8);
    if (c->h_c1[1 + kMaxQ + cidx]) rmax = ord_val(c->h_c1[1 + kMaxQ + cidx]);
    const double hl = c->cs.step[d - 1];
    long long H = planes_total;
    if (L > 0 && hl > 0) {
      const double hp = std::ceil((rmax / L * 1.000001 + 1e-6) / hl) + 2.0;
      if (hp < (double)planes_total) H = (long long)hp;
    }
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
    const long long w0 = c->world > 1 ? win_p0 : own0, w1 = c->world > 1 ? win_p1 : own0 + n / plane1;
    const long long wplanes = w1 - w0, nt = wplanes * plane1, goff = (own0 - w0) * plane1;
    const T* Wwin = (c->world > 1 && !w_is_window) ? W + w0 * plane1 : W;
    if ((rc = ensure(c->dist2, sizeof(double) * (size_t)nt))) return rc;
    if (d > 2 && (rc = ensure(c->dist2b, sizeof(double) * (size_t)nt))) return rc;
    if ((rc = ensure(c->amb, sizeof(long long) * (size_t)n))) return rc;
    double xscale = 0.0;
    for (int a = 0; a < d; ++a) xscale = std::max(xscale, std::max(std::fabs(c->cs.lo[a]), std::fabs(c->cs.hi[a])));
    const int count0 = d >= 2 ? (int)c->cs.count[0] : (int)nt;
    const unsigned gridn = (unsigned)std::min<long long>((nt + 255) / 256, 1 << 20);
    hipLaunchKernelGGL(k_reset_amb, dim3(1), dim3(1), 0, c->stream, sc);
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
      const unsigned gridc = (unsigned)std::min<long long>((nc + 255) / 256, 1 << 16);
      hipLaunchKernelGGL((k_pdt_cell_min<T>), dim3(gridc), dim3(256), 0, c->stream, Wwin, cg, nc,
                         (const unsigned long long*)c->Lmax.p, lidx, lo0, hi0);
      long long cstride = 1;
      for (int a = 0; a < d; ++a) {
        hipLaunchKernelGGL(k_pdt_coarse_scan, dim3(2 * gridc), dim3(256), 0, c->stream, (const double*)lo0, lo1, (const double*)hi0,
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
    if (c->scan_blocks && count0 >= 16 * blk0 && count0 <= 8192 && nt / count0 >= 2048) {
      // one workgroup per line, the line in LDS (pays once there are enough lines to fill the chip: measured on 1024^2 it
      // loses to the thread-per-position kernel, on 2048^2 it wins)
      const size_t lds = sizeof(double) * ((size_t)count0 + (count0 + blk0 - 1) / blk0);
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
      hipLaunchKernelGGL(k_block_min, dim3((unsigned)std::min<long long>((nb_ + 255) / 256, 1 << 20)), dim3(256), 0, c->stream,
                         (const double*)pin, stride, last_cnt, blk, (double*)c->blockmin.p);
      bmin = (const double*)c->blockmin.p;
    }
    hipLaunchKernelGGL(k_pdt_decide, dim3((unsigned)std::min<long long>((n + 255) / 256, 1 << 20)), dim3(256), 0, c->stream,
                       (const double*)pin, n, goff, d >= 2 ? stride : 1, last_cnt, last_h, d, xscale, (const uint8_t*)c->maskU.p,
                       (const unsigned long long*)c->Lmax.p, lidx, sc, cidx, O, (long long*)c->amb.p, cg, pc_lo, pc_hi, bmin, blk);
    hipLaunchKernelGGL((k_goose_exact<T, D>), dim3(1024), dim3(256), 0, c->stream, c->cs, css, W,
                       (const unsigned long long*)c->Lmax.p, lidx, sc, cidx, (const long long*)c->amb.p, O);
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

template <typename T>
static int launch_dist_to(sbo_ctx* c, const double* dev_target, T* out) {
  const long long n = c->cs.n_local;
  const int nb = reduce_blocks(c);
  switch (c->mc.dpad) {
    case 2: hipLaunchKernelGGL((k_dist_to<T, 2>), dim3(nb), dim3(256), 0, c->stream, c->cs, n, dev_target, out); break;
    case 4: hipLaunchKernelGGL((k_dist_to<T, 4>), dim3(nb), dim3(256), 0, c->stream, c->cs, n, dev_target, out); break;
    default: hipLaunchKernelGGL((k_dist_to<T, 8>), dim3(nb), dim3(256), 0, c->stream, c->cs, n, dev_target, out); break;
  }
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

// GoOSE iteration (models/GoOSE.py:63-119, test/test_GoOSE.py:151-162) on the resident candidates.  Ranks > 1: C1 + C2
// as for SafeOpt, one all-gather of the source weights per constraint, C3 for the arg-min slots and a second C3 for
// the explore step (every rank derives the same target from the merged slots).
template <typename T>
static int sweep_goose_t(sbo_ctx* c, const sbo_sweep_opts* o, sbo_goose_result* res) {
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  int rc;
  SBO_HIP(hipEventRecord(c->ev[0], c->stream));
  const bool reuse = o->posterior_ready && c->posterior_valid;
  if (!reuse && (rc = sbo_posterior_enqueue_(c))) return rc;
  SBO_HIP(hipEventRecord(c->ev[1], c->stream));
  if ((rc = sweep_common_front<T>(c, o))) return rc;
  if ((rc = sweep_exchange_front<T>(c, o, true))) return rc;
  if ((rc = ensure(c->maskO, (size_t)std::max<long long>(n, 1) * std::max(1, q - 1)))) return rc;
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  const int nb = reduce_blocks(c);
  SBO_HIP(hipEventRecord(c->ev[2], c->stream));
  // Only expanders can cover an unsafe point: "g covers h" is the predicate that puts g into G_c.  So G_c is built
  // first (distance transform, cheap) and serves as the source set of the coverage search instead of all of S_t.
  if ((rc = ensure(c->maskG, (size_t)std::max<long long>(n, 1) * std::max(1, q - 1)))) return rc;
  for (int cc = 1; cc < q; ++cc) {
    uint8_t* G = (uint8_t*)c->maskG.p + (size_t)(cc - 1) * n;
    uint8_t* O = (uint8_t*)c->maskO.p + (size_t)(cc - 1) * n;
    // (a large explicit list has no transform to build G_c with: all of S_t stays the source set there)
    long long plane = 1;
    for (int a = 0; a < c->cs.d - 1; ++a) plane *= c->cs.count[a];
    const bool can_expand = n <= (1ll << 17) || (c->cs.kind == 1 && c->cs.first % plane == 0 && n % plane == 0);
    const uint8_t* src = (const uint8_t*)c->maskS.p;
    if (can_expand) {
      if ((rc = expander_set<T>(c, o, cc, G))) return rc;
      src = G;
    }
    if ((rc = goose_sets_d<T>(c, o, cc, src, O))) return rc;
  }
  SBO_HIP(hipEventRecord(c->ev[3], c->stream));
  // value array for the arg-min reductions (after the expander transform, which uses the same scratch buffer)
  if ((rc = ensure(c->dist2, sizeof(double) * (size_t)std::max<long long>(n, 1)))) return rc;
  T* lcb0 = (T*)c->dist2.p;
  if (n > 0) {
    hipLaunchKernelGGL((k_lcb0<T>), dim3(nb), dim3(256), 0, c->stream, (const T*)c->mean.p, (const T*)c->var.p, n, (T)o->b, lcb0);
    hipLaunchKernelGGL((k_arg_masked<T, false>), dim3(nb), dim3(256), 0, c->stream, (const T*)lcb0, (const uint8_t*)c->maskS.p, n,
                       (long long)c->cs.first, (long long*)nullptr, (Best*)c->partial.p);
  }
  hipLaunchKernelGGL((k_arg_final<false>), dim3(1), dim3(256), 0, c->stream, (const Best*)c->partial.p, n > 0 ? nb : 0, sc, 0);
  for (int cc = 1; cc < q; ++cc) {
    const uint8_t* O = (const uint8_t*)c->maskO.p + (size_t)(cc - 1) * n;
    if (n > 0)
      hipLaunchKernelGGL((k_arg_masked<T, false>), dim3(nb), dim3(256), 0, c->stream, (const T*)lcb0, O, n,
                         (long long)c->cs.first, &sc->count_set[cc - 1], (Best*)c->partial.p);
    hipLaunchKernelGGL((k_arg_final<false>), dim3(1), dim3(256), 0, c->stream, (const Best*)c->partial.p, n > 0 ? nb : 0, sc, cc);
  }
  SBO_HIP(hipGetLastError());
  SweepScalars h;
  bool is_max[kArgSlots];
  for (int t = 0; t < kArgSlots; ++t) is_max[t] = false;
  if ((rc = sweep_exchange_back(c, h, is_max))) return rc;
  unsigned long long Lk[kMaxQ];
  SBO_HIP(hipMemcpyAsync(Lk, c->Lmax.p, sizeof(Lk), hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipStreamSynchronize(c->stream));
  c->masks_valid = true;
  c->last_sweep = 2;

  memset(res, 0, sizeof(*res));
  res->count_S = h.count_S;
  res->count_U = h.count_U;
  res->n_exact_rechecks = h.n_amb_total;
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
  if (best_c) {
    res->target_index = res->target_index_c[best_c - 1];
    coords_of(c, res->target_index, res->target_x);
    // explore_safeset(target): argmin_{S} ||x - target||_2 (models/GoOSE.py:116-119)
    double* dev_t = (double*)c->scal.p + 256;
    SBO_HIP(hipMemcpyAsync(dev_t, res->target_x, sizeof(double) * SBO_MAX_D, hipMemcpyHostToDevice, c->stream));
    if (n > 0) {
      if ((rc = launch_dist_to<T>(c, dev_t, lcb0))) return rc;
      hipLaunchKernelGGL((k_arg_masked<T, false>), dim3(nb), dim3(256), 0, c->stream, (const T*)lcb0, (const uint8_t*)c->maskS.p, n,
                         (long long)c->cs.first, (long long*)nullptr, (Best*)c->partial.p);
    }
    hipLaunchKernelGGL((k_arg_final<false>), dim3(1), dim3(256), 0, c->stream, (const Best*)c->partial.p, n > 0 ? nb : 0, sc,
                       kArgSlots - 1);
    SweepScalars h2;
    if ((rc = sweep_exchange_back(c, h2, is_max))) return rc;
    res->explore_index = h2.arg_idx[kArgSlots - 1];
    coords_of(c, res->explore_index, res->explore_x);
  }
  SBO_HIP(hipEventRecord(c->ev[4], c->stream));
  SBO_HIP(hipEventSynchronize(c->ev[4]));
  float t01 = 0, t12 = 0, t23 = 0, t34 = 0, t04 = 0;
  SBO_HIP(hipEventElapsedTime(&t01, c->ev[0], c->ev[1]));
  SBO_HIP(hipEventElapsedTime(&t12, c->ev[1], c->ev[2]));
  SBO_HIP(hipEventElapsedTime(&t23, c->ev[2], c->ev[3]));
  SBO_HIP(hipEventElapsedTime(&t34, c->ev[3], c->ev[4]));
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
  return SBO_OK;
}

// Trust-region acquisition (models/GP_TR.py:43-51): argmin lcb_0 over S and the ball
template <typename T>
static int sweep_tr_t(sbo_ctx* c, const sbo_sweep_opts* o, const double* x0, double r, sbo_tr_result* res) {
  const long long n = c->cs.n_local;
  int rc;
  SBO_HIP(hipEventRecord(c->ev[0], c->stream));
  const bool reuse = o->posterior_ready && c->posterior_valid;
  if (!reuse && (rc = sbo_posterior_enqueue_(c))) return rc;
  SBO_HIP(hipEventRecord(c->ev[1], c->stream));
  if ((rc = sweep_common_front<T>(c, o))) return rc;
  if ((rc = ensure(c->dist2, sizeof(double) * (size_t)std::max<long long>(n, 1)))) return rc;
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  const int nb = reduce_blocks(c);
  T* lcb0 = (T*)c->dist2.p;
  double* dev_x0 = (double*)c->scal.p + 256;
  SBO_HIP(hipMemcpyAsync(dev_x0, x0, sizeof(double) * c->cs.d, hipMemcpyHostToDevice, c->stream));
  if (n > 0) {
    hipLaunchKernelGGL((k_lcb0<T>), dim3(nb), dim3(256), 0, c->stream, (const T*)c->mean.p, (const T*)c->var.p, n, (T)o->b, lcb0);
    switch (c->mc.dpad) {
      case 2: hipLaunchKernelGGL((k_ball_mask<2>), dim3(nb), dim3(256), 0, c->stream, c->cs, n, (const uint8_t*)c->maskS.p, (const double*)dev_x0, r, (uint8_t*)c->maskM.p); break;
      case 4: hipLaunchKernelGGL((k_ball_mask<4>), dim3(nb), dim3(256), 0, c->stream, c->cs, n, (const uint8_t*)c->maskS.p, (const double*)dev_x0, r, (uint8_t*)c->maskM.p); break;
      default: hipLaunchKernelGGL((k_ball_mask<8>), dim3(nb), dim3(256), 0, c->stream, c->cs, n, (const uint8_t*)c->maskS.p, (const double*)dev_x0, r, (uint8_t*)c->maskM.p); break;
    }
    hipLaunchKernelGGL((k_arg_masked<T, false>), dim3(nb), dim3(256), 0, c->stream, (const T*)lcb0, (const uint8_t*)c->maskM.p, n,
                       (long long)c->cs.first, &sc->count_M, (Best*)c->partial.p);
  }
  hipLaunchKernelGGL((k_arg_final<false>), dim3(1), dim3(256), 0, c->stream, (const Best*)c->partial.p, n > 0 ? nb : 0, sc, 0);
  SBO_HIP(hipGetLastError());
  SweepScalars h;
  bool is_max[kArgSlots];
  for (int t = 0; t < kArgSlots; ++t) is_max[t] = false;
  if (c->world > 1) {
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
  return c->dtype == SBO_F64 ? sweep_safeopt_t<double>(c, opts, result) : sweep_safeopt_t<float>(c, opts, result);
}

int sbo_sweep_goose(sbo_ctx* c, const sbo_sweep_opts* opts, sbo_goose_result* result) {
  if (!c || !opts || !result) return fail(SBO_E_INVALID, "NULL argument");
  if (!c->has_model) return fail(SBO_E_NO_MODEL, "sbo_model_set has not been called");
  if (!c->has_cand) return fail(SBO_E_NO_CANDIDATES, "no candidates resident");
  if (c->cs.d != c->mc.d) return fail(SBO_E_INVALID, "ERROR W and X_norm dimension should be same");
  SBO_HIP(hipSetDevice(c->device));
  return c->dtype == SBO_F64 ? sweep_goose_t<double>(c, opts, result) : sweep_goose_t<float>(c, opts, result);
}

int sbo_sweep_tr(sbo_ctx* c, const sbo_sweep_opts* opts, const double* x_0, double r, sbo_tr_result* result) {
  if (!c || !opts || !x_0 || !result) return fail(SBO_E_INVALID, "NULL argument");
  if (!c->has_model) return fail(SBO_E_NO_MODEL, "sbo_model_set has not been called");
  if (!c->has_cand) return fail(SBO_E_NO_CANDIDATES, "no candidates resident");
  if (c->cs.d != c->mc.d) return fail(SBO_E_INVALID, "ERROR W and X_norm dimension should be same");
  if (!(r >= 0.0)) return fail(SBO_E_INVALID, "trust-region radius must be >= 0");
  SBO_HIP(hipSetDevice(c->device));
  return c->dtype == SBO_F64 ? sweep_tr_t<double>(c, opts, x_0, r, result) : sweep_tr_t<float>(c, opts, x_0, r, result);
}

int sbo_masks_get(sbo_ctx* c, int which, int cidx, uint8_t* out) {
  if (!c || !out) return fail(SBO_E_INVALID, "NULL argument");
  if (!c->masks_valid) return fail(SBO_E_INVALID, "no sweep has produced masks on these candidates");
  const long long n = c->cs.n_local;
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
