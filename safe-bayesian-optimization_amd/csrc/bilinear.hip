// bilinear.hip -- K1b: GP posterior on 2-D tensor grids as two dense fp64 GEMMs (gfx950).
//
// Reference arithmetic replaced: GP.GP_inference, models/GP_Safe.py:326-347, evaluated at every point of a
// linspace x linspace grid (test/test_SafeOpt.py:324-345).  The math and the host-side construction of the bases are
// in bilinear_host.hpp; this file holds the plan (device tables) and the kernels:
//
//   stage 1 (k_bstage1) Bt[x1, k0]  = sum_k1 P1[k1, x1] T4qq[k0, k1]        (lines of the grid  x  pair index of axis 0)
//   stage 2 (k_bpost)  quad[x1,x0] = sum_k0 Bt[x1, k0] P0[k0, x0]   ->  var = max(0, sf2 - quad) Y_std^2
//                      m[x1, x0]   = sum_p  V0[p, x1] S0[p, x0]     (and the two gradient sums, same shape)
//
// Both GEMMs run on the matrix cores with the fragment conventions of device_common.hpp (four v_mfma_f64_4x4x4 per
// 16x16x4 step): A operands are stored as packed 16x16 block images, B operands in fragment order, so every operand
// load is one contiguous 512-byte (B) or 2-KiB (A) wave access.  Inner dimensions: K0 = r0 (r0 + 1) / 2 ~ 280 for the
// reference's length-scales, whatever n is.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>
#include <type_traits>
#include <hip/hip_ext.h>
#include "device_common.hpp"

namespace sbo {

// Plain GEMM on the matrix cores.  OUT[rows x cols] = A[rows x K] B[K x cols];  A: packed 16 x 16 block images, Bf:
// [ncs][KB * 4][64] fragments.  A wave owns 16 rows x (16 S) columns; the four waves of a workgroup take four
// consecutive row blocks.  Used for the per-model build of T4 (stage 1 of a posterior launch has its own kernel below).
//   TRI   0: A images [nrb][KB][256];  1: lower-triangular images [tri(rb, kb)][256] (the model's factor M as K1g keeps
//            it), only k-blocks <= rb exist and are run
//   OMODE 0: OUT as packed block images [nrb][ncs][256] (the A operand of a following GEMM with K = cols)
//         1: OUT both as B fragments [ncs][nrb * 4][64] (K = rows) and, transposed, as A images [ncs][nrb][256]
//         2: OUT row-major with leading dimension ld
template <int S, int TRI, int OMODE>
__global__ __launch_bounds__(256) void k_bgemm(const double* __restrict__ A, size_t a_stride_o, const double* __restrict__ Bf,
                                               size_t b_stride_o, int KB, int nrb, int ncs, double* __restrict__ out,
                                               size_t out_stride_o, double* __restrict__ out2, long long ld) {
  const int o = blockIdx.z;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rb = blockIdx.y * 4 + wave;
  if (rb >= nrb) return;                       // (no barriers in this kernel)
  const int cs0 = blockIdx.x * S;
  const double* Ablk = A + (size_t)o * a_stride_o + (TRI ? (size_t)rb * (rb + 1) / 2 : (size_t)rb * KB) * 256;
  const double* Bo = Bf + (size_t)o * b_stride_o;
  const int kend = TRI ? rb + 1 : KB;
  size_t boff[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int cs = cs0 + s < ncs ? cs0 + s : ncs - 1;
    boff[s] = (size_t)cs * KB * 4 * 64 + lane;
  }
  d4_t acc[S];
#pragma unroll
  for (int s = 0; s < S; ++s) acc[s] = d4_t{0.0, 0.0, 0.0, 0.0};
  // operands of k-block kb + 1 are loaded while kb is multiplied: with two to three waves per SIMD nothing else hides
  // the L2 latency of the fragment loads
  d4_t a[2][4];
  double b[2][4][S];
  auto load_kb = [&](int kb, d4_t (&aa)[4], double (&bb)[4][S]) {
    MM<double>::load_a4(Ablk + (size_t)kb * 256, lane, aa);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int s = 0; s < S; ++s) bb[kk][s] = Bo[boff[s] + (size_t)(kb * 4 + kk) * 64];
  };
  auto mul_kb = [&](const d4_t (&aa)[4], const double (&bb)[4][S]) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int s = 0; s < S; ++s) acc[s] = MM<double>::mfma(aa[kk], bb[kk][s], acc[s]);
  };
  load_kb(0, a[0], b[0]);
  for (int kb = 0; kb < kend; kb += 2) {
    if (kb + 1 < kend) load_kb(kb + 1, a[1], b[1]);
    mul_kb(a[0], b[0]);
    if (kb + 1 < kend) {
      if (kb + 2 < kend) load_kb(kb + 2, a[0], b[0]);
      mul_kb(a[1], b[1]);
    }
  }
  // accumulator element t of lane l: row 4 t + (l >> 4), column l & 15 of the 16 x 16 tile
  const int col_in = lane & 15, row_in = lane >> 4;
  out += (size_t)o * out_stride_o;
  if (OMODE == 1) out2 += (size_t)o * out_stride_o;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    if (cs0 + s >= ncs) continue;
    if (OMODE == 0) {
      double* blk = out + ((size_t)rb * ncs + (cs0 + s)) * 256;
#pragma unroll
      for (int t = 0; t < 4; ++t) blk[MM<double>::pack_pos(4 * t + row_in, col_in & 3, col_in >> 2)] = acc[s][t];
    } else if (OMODE == 1) {
      // as B fragments: k = row = rb * 16 + 4 t + row_in  ->  k-step rb * 4 + t, slot row_in: the lane's own position
      double* fr = out + ((size_t)(cs0 + s) * nrb * 4 + (size_t)rb * 4) * 64 + lane;
      // transposed as A images: row = column index, k = rb * 16 + 4 t + row_in
      double* img = out2 + ((size_t)(cs0 + s) * nrb + rb) * 256;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fr[(size_t)t * 64] = acc[s][t];
        img[MM<double>::pack_pos(col_in, row_in, t)] = acc[s][t];
      }
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        out[(size_t)(rb * 16 + 4 * t + row_in) * ld + (size_t)(cs0 + s) * 16 + col_in] = acc[s][t];
    }
  }
}

typedef double d2_t __attribute__((ext_vector_type(2)));

// Stage 1 of every posterior launch (OUT as packed block images, the OMODE 0 of k_bgemm): the four waves of a workgroup
// own four row blocks and share S column strips, so the strips' B fragments are staged once per workgroup through LDS
// (double-buffered, one barrier per k-block) instead of once per wave from L2 -- with K ~ 290 and only 16 x 16 S outputs
// per wave the plain kernel is bound by L2 bandwidth (~110 KB of operands per wave), not by the matrix cores.
template <int S>
__global__ __launch_bounds__(256) void k_bstage1(const double* __restrict__ A, size_t a_stride_o, const double* __restrict__ Bf,
                                                 size_t b_stride_o, int KB, int nrb, int ncs, double* __restrict__ out,
                                                 size_t out_stride_o, const int* __restrict__ eff /* nullptr, or the Chebyshev core's
                                                 per-output counts: [4 o + 1] strips to produce, [4 o + 2] k-blocks to run */) {
  __shared__ double Bs[2][4 * S * 64];          // [buffer][(kk S + s) 64 + lane]
  // A: a wave's 16 x 16 block image goes through LDS too.  Read straight from memory, a fragment load has the four
  // lanes of a quad fetch the same 32 bytes (8 KB of lane traffic for a 2 KB image per k-block, which is what the
  // texture path then limits); here a lane fetches 32 bytes once and the replicated reads are LDS broadcasts.  Layout per
  // k-step as in k_bpost: the first 16-byte halves of the 16 chunks, then the second halves (conflict-free b128 reads).
  __shared__ __attribute__((aligned(16))) double As[2][4][256];   // [buffer][wave][image]
  const int o = blockIdx.z, tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int rb = blockIdx.y * 4 + wave, rbl = rb < nrb ? rb : nrb - 1;
  const int cs0 = blockIdx.x * S;
  const int KBrun = eff ? eff[4 * o + 2] : KB;            // (KB stays the layout's stride)
  if (eff && cs0 >= eff[4 * o + 1]) return;               // strips behind the last degree that matters: never read
  const double* Ablk = A + (size_t)o * a_stride_o + (size_t)rbl * KB * 256 + lane * 4;   // this lane's 32 bytes of an image
  const double* Bo = Bf + (size_t)o * b_stride_o;
  // staging role: element e = tid + 256 j (j < S) of the k-block's [4][S][64] fragment set
  size_t goff[S];
#pragma unroll
  for (int j = 0; j < S; ++j) {
    const int e = tid + 256 * j, l = e & 63, ks = e >> 6, s_ = ks % S, kk = ks / S;
    const int cs = cs0 + s_ < ncs ? cs0 + s_ : ncs - 1;
    goff[j] = ((size_t)cs * KB * 4 + kk) * 64 + l;
  }
  const int a_st = (lane >> 4) * 64 + (lane & 15) * 2;                        // where this lane's 32 bytes go (two halves)
  const int a_rd = (((lane >> 4) << 2) + (lane & 3)) * 2;                     // the chunk its fragment reads
  d4_t acc[S];
#pragma unroll
  for (int s_ = 0; s_ < S; ++s_) acc[s_] = d4_t{0.0, 0.0, 0.0, 0.0};
  double breg[S];
  d4_t areg = *reinterpret_cast<const d4_t*>(Ablk);
#pragma unroll
  for (int j = 0; j < S; ++j) Bs[0][tid + 256 * j] = Bo[goff[j]];
  *reinterpret_cast<d2_t*>(&As[0][wave][a_st]) = d2_t{areg[0], areg[1]};
  *reinterpret_cast<d2_t*>(&As[0][wave][a_st + 32]) = d2_t{areg[2], areg[3]};
  __syncthreads();
  // one k-block: prefetch block kb + 1 (registers), multiply block kb out of LDS buffer CUR, park the prefetch in the
  // other buffer (its last readers passed the previous barrier).  Written twice so that the buffers are compile-time names.
  auto step = [&](int kb, const double* bs_cur, double* bs_nxt, const double* as_cur, double* as_nxt) {
    const bool more = kb + 1 < KBrun;
    if (more) {
#pragma unroll
      for (int j = 0; j < S; ++j) breg[j] = Bo[goff[j] + (size_t)(kb + 1) * 256];
      areg = *reinterpret_cast<const d4_t*>(Ablk + (size_t)(kb + 1) * 256);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const d2_t lo = *reinterpret_cast<const d2_t*>(as_cur + kk * 64 + a_rd), hi = *reinterpret_cast<const d2_t*>(as_cur + kk * 64 + 32 + a_rd);
      const d4_t af = d4_t{lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
      for (int s_ = 0; s_ < S; ++s_) acc[s_] = MM<double>::mfma(af, bs_cur[(kk * S + s_) * 64 + lane], acc[s_]);
    }
    if (more) {
#pragma unroll
      for (int j = 0; j < S; ++j) bs_nxt[tid + 256 * j] = breg[j];
      *reinterpret_cast<d2_t*>(as_nxt + a_st) = d2_t{areg[0], areg[1]};
      *reinterpret_cast<d2_t*>(as_nxt + a_st + 32) = d2_t{areg[2], areg[3]};
    }
    __syncthreads();
  };
  for (int kb = 0; kb < KBrun; kb += 2) {
    step(kb, Bs[0], Bs[1], As[0][wave], As[1][wave]);
    if (kb + 1 < KBrun) step(kb + 1, Bs[1], Bs[0], As[1][wave], As[0][wave]);
  }
  if (rb >= nrb) return;
  const int col_in = lane & 15, row_in = lane >> 4;
#pragma unroll
  for (int s_ = 0; s_ < S; ++s_) {
    if (cs0 + s_ >= ncs) continue;
    double* blk = out + (size_t)o * out_stride_o + ((size_t)rb * ncs + (cs0 + s_)) * 256;
#pragma unroll
    for (int t = 0; t < 4; ++t) blk[MM<double>::pack_pos(4 * t + row_in, col_in & 3, col_in >> 2)] = acc[s_][t];
  }
}

// ---- per-model plan build, all on the device ------------------------------------------------------------------------------
// A model change costs: the 2 q axis bases (k_bl_basis, one workgroup each), then a dozen launches batched over the outputs
// (blockIdx.y / .z = output) that turn the bases into the operand tables of the two GEMMs.  The host only reads the bases'
// ranks back (to size the GEMMs) -- no host numerics, no staging upload.
constexpr int kBlMaxR = 64;             // largest rank of an axis basis
constexpr int kBlMaxRc = 128;           // largest Chebyshev degree of an axis family
static inline int pair_count(int r) { return r * (r + 1) / 2; }

struct BlDims {                          // plan dimensions, by value
  int q, n, KBn;
  int r0[kMaxQ], r1[kMaxQ], rc0[kMaxQ], rc1[kMaxQ];
  int r0u, r1u, r0p, KB0, KB1, KBm, KBm2, ncs0, nrb, ncsR;
  int K0m, K1m;                          // largest pair counts over the outputs (Chebyshev core: rows of the pair-coefficient matrices)
  int D0m, D1m;                          // Chebyshev core: padded degrees (16 KB0, 16 KB1)
  long long cnt0, nlines;
  double a[2], b[2];                     // interval of each axis in normalised coordinates
  double sf2[kMaxQ];
};

// pair index k -> (p <= p'), pairs enumerated row by row : row p starts at p (2 r - p + 1) / 2
__device__ __forceinline__ void pair_of(int k, int r, int& p, int& pp) {
  const double t = 2.0 * r + 1.0;
  int g = (int)((t - sqrt(t * t - 8.0 * k)) * 0.5);
  g = g < 0 ? 0 : (g > r - 1 ? r - 1 : g);
  while (g > 0 && g * (2 * r - g + 1) / 2 > k) --g;
  while (g + 1 < r && (g + 1) * (2 * r - g) / 2 <= k) ++g;
  p = g;
  pp = g + (k - g * (2 * r - g + 1) / 2);
}

// normalised positions of the grid axes: xn0[cnt0] (all of axis 0), xn1[nlines] (the local lines of axis 1) -- the
// arithmetic of cand_coords and models/GP_Safe.py:326
__global__ __launch_bounds__(256) void k_bl_axes(const ModelConst mc, const CandSpec cs, long long cnt0, long long line0, long long nlines,
                                                 double* __restrict__ xn0, double* __restrict__ xn1) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < cnt0 + nlines; t += (long long)gridDim.x * blockDim.x) {
    const int a = t < cnt0 ? 0 : 1;
    const long long i = a == 0 ? t : line0 + (t - cnt0), tot = cs.count[a];
    const double x = (i == tot - 1 && tot > 1) ? cs.hi[a] : __dadd_rn(cs.lo[a], __dmul_rn((double)i, cs.step[a]));
    const double v = (x - mc.X_mean[a]) / mc.X_std[a];
    if (a == 0) xn0[t] = v; else xn1[t - cnt0] = v;
  }
}

// Basis of one axis family f_j(x) = exp(-1/2 (x vinv - A_j)^2), j < n, on the interval [a, b] (normalised coordinates),
// one workgroup per (output, axis):
//   1. Chebyshev interpolant of every f_j (degree raised 32 -> 128 until the last four coefficients are < 1e-14),
//      coefficient rows c_j in R^rc by the discrete cosine sum;
//   2. column-pivoted Gram-Schmidt on the rows (largest residual first, every direction re-orthogonalised twice against
//      the ones before): directions q_0 .. q_{r-1}, stopped when the largest residual row is below 2e-16 ||coef||_F --
//      the rank a singular value decomposition finds (or one less), at a fraction of its sequential depth (r steps instead
//      of several hundred rounds of rotations);
//   3. U_jp = c_j . q_p, taken from the loop itself (the residual of row j is c_j minus its components along the earlier,
//      orthogonal directions), so that f_j(x) = sum_p U_jp S_p(x), S_p = sum_c Q_pc T_c(xi(x)).
// Outputs: U [r][n], Vs = Q [r][rc], sig = 1, info = (ok, r, rc).  ok = 0: the interpolant did not converge or the rank
// exceeds kBlMaxR -- the caller keeps the separable-table kernel; ok = 2: the small form ran out of LDS (degree > 64), the
// caller runs the big one.
// The pivot loop is a chain of short dependent phases, so operand latency is its duration: small problems (n <= 128: NT =
// 256 threads, LDSRES) keep samples / residual rows in LDS, every problem keeps the directions there when rc <= 64.
struct BlJobs {
  int n, dpad;
  double vinv[2 * kMaxQ], a[2], b[2];
};
constexpr int kBasisQLds = 64 * 64;          // doubles of LDS for the directions (used when rc <= 64, row stride rc)
constexpr int kBasisResLds = 64 * 128;       // doubles of LDS for samples / residual rows of the small variant

template <int NT>
__device__ __forceinline__ double bl_block_sum(double v, double* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) s += red[w];
  return s;                                             // (same order in every thread: deterministic)
}

// the pivot loop; res [rc][n] and Q [..][rc] may live in LDS (pointers derived from the dynamic LDS block at the call site)
template <int NT, typename Stamp>
__device__ __forceinline__ int basis_pivot_loop(int n, int rc, double* res, double* Q, double* __restrict__ U, double tol2, double* nrm2,
                                                double* qv, double* cf, double* red, int* redi, bool& too_large, Stamp stamp, double& res2_out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rmax = rc < n ? rc : n;
  int r = 0;
  too_large = false;
  // pivot of the first step: the row with the largest norm (ties -> lowest row)
  double bv = -1.0;
  int bi = 0x7fffffff;
  for (int j = tid; j < n; j += NT)
    if (nrm2[j] > bv) { bv = nrm2[j]; bi = j; }
  for (;;) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double yv = __shfl_xor(bv, o);
      const int yi = __shfl_xor(bi, o);
      if (yv > bv || (yv == bv && yi < bi)) { bv = yv; bi = yi; }
    }
    if (lane == 0) { red[wave] = bv; redi[wave] = bi; }
    __syncthreads();
    bv = red[0];
    bi = redi[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w)
      if (red[w] > bv || (red[w] == bv && redi[w] < bi)) { bv = red[w]; bi = redi[w]; }
    if (!(bv > tol2) || r >= rmax) break;
    if (r >= kBlMaxR) { too_large = true; break; }
    const double inv0 = 1.0 / sqrt(bv);
    if (tid < rc) qv[tid] = res[(size_t)tid * n + bi] * inv0;
    __syncthreads();
    // twice: remove what is left along the previous directions (the residual rows are only orthogonal to them to the
    // rounding of the ORIGINAL rows, which is not small against a residual of 1e-12)
    double mine = 0.0, nq = 1.0;
    for (int pass = 0; pass < 2; ++pass) {
      for (int i = tid >> 4; i < r; i += NT / 16) {              // 16 lanes per previous direction
        double s_ = 0.0;
        for (int cidx = tid & 15; cidx < rc; cidx += 16) s_ += Q[(size_t)i * rc + cidx] * qv[cidx];
        s_ += __shfl_xor(s_, 8);
        s_ += __shfl_xor(s_, 4);
        s_ += __shfl_xor(s_, 2);
        s_ += __shfl_xor(s_, 1);
        if ((tid & 15) == 0) cf[i] = s_;
      }
      __syncthreads();
      if (tid < rc) {
        // (four partial sums: the chain of dependent multiply-adds, not their count, is what a step waits for)
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
        int i = 0;
        for (; i + 3 < r; i += 4) {
          p0 += cf[i] * Q[(size_t)i * rc + tid];
          p1 += cf[i + 1] * Q[(size_t)(i + 1) * rc + tid];
          p2 += cf[i + 2] * Q[(size_t)(i + 2) * rc + tid];
          p3 += cf[i + 3] * Q[(size_t)(i + 3) * rc + tid];
        }
        for (; i < r; ++i) p0 += cf[i] * Q[(size_t)i * rc + tid];
        mine = qv[tid] - ((p0 + p1) + (p2 + p3));
      }
      if (pass == 0) {
        if (tid < rc) qv[tid] = mine;
        __syncthreads();
      } else {
        nq = bl_block_sum<NT>(tid < rc ? mine * mine : 0.0, red);
      }
    }
    if (nq < 0.25) break;                                  // mostly rounding of rows already covered: nothing left to add
    if (tid < rc) {
      mine /= sqrt(nq);
      qv[tid] = mine;
      Q[(size_t)r * rc + tid] = mine;
    }
    __syncthreads();
    // residual rows: d_j = res_j . q (= U_j,r), res_j -= d_j q, norms recomputed from the updated rows, next pivot
    bv = -1.0;
    bi = 0x7fffffff;
    for (int j = tid; j < n; j += NT) {
      double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;         // (rc is a multiple of 16)
      for (int cidx = 0; cidx < rc; cidx += 4) {
        d0 += res[(size_t)cidx * n + j] * qv[cidx];
        d1 += res[(size_t)(cidx + 1) * n + j] * qv[cidx + 1];
        d2 += res[(size_t)(cidx + 2) * n + j] * qv[cidx + 2];
        d3 += res[(size_t)(cidx + 3) * n + j] * qv[cidx + 3];
      }
      const double d_ = (d0 + d1) + (d2 + d3);
      U[(size_t)r * n + j] = d_;
      double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
      for (int cidx = 0; cidx < rc; cidx += 4) {
        const double v0 = res[(size_t)cidx * n + j] - d_ * qv[cidx], v1 = res[(size_t)(cidx + 1) * n + j] - d_ * qv[cidx + 1];
        const double v2 = res[(size_t)(cidx + 2) * n + j] - d_ * qv[cidx + 2], v3 = res[(size_t)(cidx + 3) * n + j] - d_ * qv[cidx + 3];
        res[(size_t)cidx * n + j] = v0;
        res[(size_t)(cidx + 1) * n + j] = v1;
        res[(size_t)(cidx + 2) * n + j] = v2;
        res[(size_t)(cidx + 3) * n + j] = v3;
        n0 += v0 * v0; n1 += v1 * v1; n2 += v2 * v2; n3 += v3 * v3;
      }
      const double nn = (n0 + n1) + (n2 + n3);
      if (nn > bv) { bv = nn; bi = j; }
    }
    ++r;
    __syncthreads();                                       // (red / redi / qv are rewritten by the next step)
    stamp();
  }
  res2_out = bv > 0.0 ? bv : 0.0;          // (the block's largest residual row norm^2 at every way out of the loop)
  return r;
}

// The same loop with the residual rows in REGISTERS (r03): two threads per row, each holding half of its rc coefficients
// (rc = 32 / 48 / 64, n <= NT / 2).  A step of the loop above re-reads and re-writes every residual row through LDS (n <= 128)
// or global memory (n > 128: 13 us per step at n = 512, 22 steps); here the pivot row is published once to LDS and every
// thread updates its own registers -- the step is left with the barriers of the re-orthogonalisation.  Same algorithm and
// stopping rules; sums are grouped differently (per half row), which moves U by a few ulp.
template <int NT, int RCH>
__device__ __forceinline__ int basis_pivot_loop_reg(int n, const double* __restrict__ resG, double* Q, double* __restrict__ U, double* qv, double* cf,
                                                    double* red, int* redi, bool& too_large, double& fro2_out, double& res2_out) {
  constexpr int rc = 2 * RCH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = tid >> 1, half = tid & 1;
  const bool row = j < n;
  double rr[RCH];
#pragma unroll
  for (int k = 0; k < RCH; ++k) rr[k] = row ? resG[(size_t)(half * RCH + k) * n + j] : 0.0;
  auto row_norm = [&]() {
    double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
#pragma unroll
    for (int k = 0; k < RCH; k += 4) { n0 += rr[k] * rr[k]; n1 += rr[k + 1] * rr[k + 1]; n2 += rr[k + 2] * rr[k + 2]; n3 += rr[k + 3] * rr[k + 3]; }
    double nn = (n0 + n1) + (n2 + n3);
    nn += __shfl_xor(nn, 1);
    return nn;
  };
  double nn = row_norm();
  const double fro2 = bl_block_sum<NT>((row && half == 0) ? nn : 0.0, red);
  fro2_out = fro2;
  __syncthreads();
  const double tol2 = (2e-16 * 2e-16) * fro2;
  const int rmax = rc < n ? rc : n;
  int r = 0;
  too_large = false;
  double bv = (row && half == 0) ? nn : -1.0;
  int bi = (row && half == 0) ? j : 0x7fffffff;
  for (;;) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double yv = __shfl_xor(bv, o);
      const int yi = __shfl_xor(bi, o);
      if (yv > bv || (yv == bv && yi < bi)) { bv = yv; bi = yi; }
    }
    if (lane == 0) { red[wave] = bv; redi[wave] = bi; }
    __syncthreads();
    bv = red[0];
    bi = redi[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w)
      if (red[w] > bv || (red[w] == bv && redi[w] < bi)) { bv = red[w]; bi = redi[w]; }
    if (!(bv > tol2) || r >= rmax) break;
    if (r >= kBlMaxR) { too_large = true; break; }
    const double inv0 = 1.0 / sqrt(bv);
    if (j == bi) {
#pragma unroll
      for (int k = 0; k < RCH; ++k) qv[half * RCH + k] = rr[k] * inv0;
    }
    __syncthreads();
    double mine = 0.0, nq = 1.0;
    for (int pass = 0; pass < 2; ++pass) {
      for (int i = tid >> 4; i < r; i += NT / 16) {              // 16 lanes per previous direction
        double s_ = 0.0;
        for (int cidx = tid & 15; cidx < rc; cidx += 16) s_ += Q[(size_t)i * rc + cidx] * qv[cidx];
        s_ += __shfl_xor(s_, 8);
        s_ += __shfl_xor(s_, 4);
        s_ += __shfl_xor(s_, 2);
        s_ += __shfl_xor(s_, 1);
        if ((tid & 15) == 0) cf[i] = s_;
      }
      __syncthreads();
      if (tid < rc) {
        double p0 = 0.0, p1 = 0.0, p2 = 0.0, p3 = 0.0;
        int i = 0;
        for (; i + 3 < r; i += 4) {
          p0 += cf[i] * Q[(size_t)i * rc + tid];
          p1 += cf[i + 1] * Q[(size_t)(i + 1) * rc + tid];
          p2 += cf[i + 2] * Q[(size_t)(i + 2) * rc + tid];
          p3 += cf[i + 3] * Q[(size_t)(i + 3) * rc + tid];
        }
        for (; i < r; ++i) p0 += cf[i] * Q[(size_t)i * rc + tid];
        mine = qv[tid] - ((p0 + p1) + (p2 + p3));
      }
      if (pass == 0) {
        __syncthreads();                                         // (everyone has read qv / cf of this pass)
        if (tid < rc) qv[tid] = mine;
        __syncthreads();
      } else {
        nq = bl_block_sum<NT>(tid < rc ? mine * mine : 0.0, red);
      }
    }
    if (nq < 0.25) break;                                  // mostly rounding of rows already covered: nothing left to add
    if (tid < rc) {
      mine /= sqrt(nq);
      qv[tid] = mine;
      Q[(size_t)r * rc + tid] = mine;
    }
    __syncthreads();
    // residual rows in registers: d_j = res_j . q (= U_j,r), res_j -= d_j q, norms recomputed from the updated rows
    double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
    const double* qh = qv + half * RCH;
#pragma unroll
    for (int k = 0; k < RCH; k += 4) {
      d0 += rr[k] * qh[k];
      d1 += rr[k + 1] * qh[k + 1];
      d2 += rr[k + 2] * qh[k + 2];
      d3 += rr[k + 3] * qh[k + 3];
    }
    double d_ = (d0 + d1) + (d2 + d3);
    d_ += __shfl_xor(d_, 1);
    if (row && half == 0) U[(size_t)r * n + j] = d_;
#pragma unroll
    for (int k = 0; k < RCH; ++k) rr[k] -= d_ * qh[k];
    nn = row_norm();
    bv = (row && half == 0) ? nn : -1.0;
    bi = (row && half == 0) ? j : 0x7fffffff;
    ++r;
    __syncthreads();                                       // (red / redi / qv are rewritten by the next step)
  }
  res2_out = bv > 0.0 ? bv : 0.0;          // (the block's largest residual row norm^2 at every way out of the loop)
  return r;
}

// ---- front end of the bases, spread over the chip -------------------------------------------------------------------------
// k_bl_basis is one workgroup per (output, axis): its pivot loop is sequential by nature, but the two passes in front of it
// (Chebyshev samples + degree test, all coefficients) are not, and on a single CU they took 54 of config B's 182 us and 184
// of H's 520.  k_bl_degree finds the smallest degree of the ladder 32 / 48 / 64 / 96 / 128 whose last four coefficients are
// below 1e-14 for every row (a thread per row, the ladder walked per workgroup, one atomicMax per workgroup and job);
// k_bl_coef writes the coefficient rows of that degree straight into the job's residual-row workspace (a workgroup per 32
// rows: its samples in LDS, the sums in the order the single-workgroup pass uses -- the same values bit for bit).
constexpr int kCoefRows = 32;
__device__ __forceinline__ int bl_degree_of(int trial) { return trial == 0 ? 32 : (trial == 1 ? 48 : (trial == 2 ? 64 : (trial == 3 ? 96 : kBlMaxRc))); }

__global__ __launch_bounds__(256) void k_bl_degree(const BlJobs jb, const double* __restrict__ Xn, int* __restrict__ rc_job) {
  // four lanes per row (64 rows per workgroup): each takes every fourth sample -- a thread alone walks up to 80 exp() before
  // the ladder stops at degree 48
  __shared__ double ct[4 * kBlMaxRc];
  __shared__ double red[4];
  const int job = blockIdx.y, axis = job & 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = jb.n, j = blockIdx.x * 64 + (tid >> 2), sub = tid & 3;
  const double vinv = jb.vinv[job], a = jb.a[axis], b = jb.b[axis];
  const double xj = j < n ? Xn[(size_t)j * jb.dpad + axis] * vinv : 0.0;
  int pass = 0;                                                    // 0: no degree of the ladder is enough
  for (int trial = 0; trial < 5; ++trial) {
    const int rc = bl_degree_of(trial);
    __syncthreads();
    for (int m = tid; m < 4 * rc; m += 256) ct[m] = cospi((double)m / (2.0 * rc));
    __syncthreads();
    double s_[4] = {0.0, 0.0, 0.0, 0.0};
    if (j < n) {
      int m_[4], st_[4];                                           // angle index p (2 k + 1) mod 4 rc and its step for k += 4
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pq = rc - 4 + u;
        m_[u] = (pq * (2 * sub + 1)) % (4 * rc);
        st_[u] = (pq * 8) % (4 * rc);
      }
      for (int k = sub; k < rc; k += 4) {
        const double x = 0.5 * (ct[2 * k + 1] * (b - a) + (a + b));
        const double df = x * vinv - xj;
        const double f = exp(-0.5 * (df * df));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          s_[u] += f * ct[m_[u]];
          m_[u] += st_[u];
          if (m_[u] >= 4 * rc) m_[u] -= 4 * rc;
        }
      }
    }
    double tail = 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      double v = s_[u];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      const double t_ = fabs(v * (2.0 / rc));
      tail = t_ > tail ? t_ : tail;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_xor(tail, o); tail = y > tail ? y : tail; }
    __syncthreads();
    if (lane == 0) red[wave] = tail;
    __syncthreads();
    tail = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (tail <= 1e-14) { pass = rc; break; }                       // (uniform: every thread sees the same maximum)
  }
  if (tid == 0) atomicMax(&rc_job[job], pass ? pass : 0x7fffffff);
}

__global__ __launch_bounds__(256) void k_bl_coef(const BlJobs jb, const double* __restrict__ Xn, const int* __restrict__ rc_job,
                                                 double* __restrict__ work, size_t work_stride, double* __restrict__ axis_eps) {
  __shared__ double ct[4 * kBlMaxRc];
  __shared__ double fs[kBlMaxRc * kCoefRows];                       // samples [k][row of the tile]
  const int job = blockIdx.y, axis = job & 1, tid = threadIdx.x;
  const int n = jb.n, rc = rc_job[job];
  if (rc <= 0 || rc > kBlMaxRc) return;                             // no degree qualified: k_bl_basis reports it
  const int j0 = blockIdx.x * kCoefRows;
  const double vinv = jb.vinv[job], a = jb.a[axis], b = jb.b[axis];
  double* res = work + (size_t)job * work_stride + (size_t)n * kBlMaxRc;   // the job's residual rows (resG of k_bl_basis)
  for (int m = tid; m < 4 * rc; m += 256) ct[m] = cospi((double)m / (2.0 * rc));
  __syncthreads();
  for (int w = tid; w < rc * kCoefRows; w += 256) {
    const int k = w / kCoefRows, jl = w % kCoefRows, j = j0 + jl;
    double f = 0.0;
    if (j < n) {
      const double x = 0.5 * (ct[2 * k + 1] * (b - a) + (a + b));
      const double df = x * vinv - Xn[(size_t)j * jb.dpad + axis] * vinv;
      f = exp(-0.5 * (df * df));
    }
    fs[w] = f;
  }
  __syncthreads();
  double t4 = 0.0;
  for (int w = tid; w < rc * kCoefRows; w += 256) {
    const int p = w / kCoefRows, jl = w % kCoefRows, j = j0 + jl;
    if (j >= n) continue;
    double s_ = 0.0;
    int m = p;
    for (int k = 0; k < rc; ++k) {
      s_ += fs[k * kCoefRows + jl] * ct[m];
      m += 2 * p;
      if (m >= 4 * rc) m -= 4 * rc;
    }
    const double cv = s_ * ((p == 0 ? 1.0 : 2.0) / rc);
    res[(size_t)p * n + j] = cv;
    if (p >= rc - 4) t4 = fmax(t4, fabs(cv));
  }
  // the largest of the last four coefficients at the degree chosen: what the guard band extrapolates the dropped tail from
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t4 = fmax(t4, __shfl_xor(t4, o));
  if ((tid & 63) == 0 && t4 > 0.0 && axis_eps)
    atomicMax(reinterpret_cast<unsigned long long*>(axis_eps + 2 * job), (unsigned long long)__double_as_longlong(t4));
}

template <int NT, bool LDSRES>
__global__ __launch_bounds__(NT) void k_bl_basis(const BlJobs jb, const double* __restrict__ Xn, double* __restrict__ work,
                                                 size_t work_stride, double* __restrict__ Uout, double* __restrict__ Vsout,
                                                 double* __restrict__ sigout, int* __restrict__ info,
                                                 long long* __restrict__ dbg /* nullptr, or 80 time stamps of job 0 */,
                                                 const int* __restrict__ rc_pre /* nullptr, or the degree per job from k_bl_degree
                                                                                    (its coefficient rows are in the workspace) */,
                                                 int no_reg /* 1: never the register form of the pivot loop (A/B) */, double* __restrict__ axis_eps) {
  extern __shared__ double bl_dyn[];          // [kBasisQLds directions | kBasisResLds samples / residual rows (LDSRES)]
  __shared__ double ct[4 * kBlMaxRc];        // cos(pi m / (2 rc)), m < 4 rc
  __shared__ double qv[kBlMaxRc];            // the direction being built
  __shared__ double cf[kBlMaxR];             // its coefficients along the previous directions
  __shared__ double nrm2[SBO_MAX_N];         // squared norm of every row
  __shared__ double red[NT / 64];
  __shared__ int redi[NT / 64];
  const int job = blockIdx.x, axis = job & 1, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = jb.n;
  const double vinv = jb.vinv[job], a = jb.a[axis], b = jb.b[axis];
  double* wk = work + (size_t)job * work_stride;
  double* fnT = LDSRES ? bl_dyn + kBasisQLds : wk;                          // [k][j] samples
  double* resG = wk + (size_t)n * kBlMaxRc;                                // [p][j] residual rows (global form)
  double* QG = resG + (size_t)n * kBlMaxRc;                                // directions (global form, rc > 64)
  int* inf = info + 4 * job;
  int rc = 0;
  bool ok = false, out_of_lds = false, regpath = false;
  int ndbg = 0;
  auto stamp = [&]() { if (dbg && job == 0 && tid == 0 && ndbg < 80) dbg[ndbg++] = wall_clock64(); };
  stamp();
  if (rc_pre) {
    rc = rc_pre[job];
    ok = rc > 0 && rc <= kBlMaxRc;
    // rows in registers, two threads per row (the 1024-thread form has 128 registers per thread: degree 64 would spill)
    regpath = ok && n <= NT / 2 && (rc == 32 || rc == 48 || (rc == 64 && NT <= 256)) && !no_reg;
    if (ok && LDSRES && !regpath && rc * n > kBasisResLds) { ok = false; out_of_lds = true; }
    if (ok && LDSRES && !regpath) {                                // the rows move from the workspace into LDS
      for (int w = tid; w < rc * n; w += NT) (bl_dyn + kBasisQLds)[w] = resG[w];
    }
    __syncthreads();
  }
  for (int trial = 0; trial < 5 && !ok && !rc_pre; ++trial) {
    rc = trial == 0 ? 32 : (trial == 1 ? 48 : (trial == 2 ? 64 : (trial == 3 ? 96 : kBlMaxRc)));
    if (LDSRES && rc * n > kBasisResLds) { out_of_lds = true; break; }
    __syncthreads();
    for (int m = tid; m < 4 * rc; m += NT) ct[m] = cospi((double)m / (2.0 * rc));
    __syncthreads();
    // samples at the Chebyshev points of the first kind, theta_k = pi (k + 1/2) / rc
    for (int w = tid; w < rc * n; w += NT) {
      const int k = w / n, j = w - k * n;
      const double x = 0.5 * (ct[2 * k + 1] * (b - a) + (a + b));
      const double df = x * vinv - Xn[(size_t)j * jb.dpad + axis] * vinv;
      fnT[w] = exp(-0.5 * (df * df));
    }
    __syncthreads();
    // the last four coefficients decide whether this degree is enough
    double tail = 0.0;
    for (int w = tid; w < 4 * n; w += NT) {
      const int p = rc - 4 + w / n, j = w % n;
      double s_ = 0.0;
      int m = p;
      for (int k = 0; k < rc; ++k) {
        s_ += fnT[(size_t)k * n + j] * ct[m];
        m += 2 * p;
        if (m >= 4 * rc) m -= 4 * rc;
      }
      s_ = fabs(s_ * (2.0 / rc));
      tail = s_ > tail ? s_ : tail;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double y = __shfl_xor(tail, o); tail = y > tail ? y : tail; }
    __syncthreads();
    if (lane == 0) red[wave] = tail;
    __syncthreads();
    tail = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) tail = red[w] > tail ? red[w] : tail;
    ok = tail <= 1e-14;
  }
  if (!ok) {
    if (tid == 0) { inf[0] = out_of_lds ? 2 : 0; inf[1] = 0; inf[2] = rc; inf[3] = 0; }
    return;
  }
  stamp();
  // all coefficients: coef[j][p] = w_p / rc sum_k f_j(x_k) cos(p theta_k), straight into the residual rows.  The small
  // variant reuses the samples' LDS: every thread holds its coefficients in registers until all sums are done.
  double* res = LDSRES ? bl_dyn + kBasisQLds : resG;
  constexpr int kKeep = LDSRES ? kBasisResLds / NT : 1;
  double keep[kKeep];
  if (rc_pre) {
    // (the coefficient rows were written by k_bl_coef)
  } else if (LDSRES) {
#pragma unroll
    for (int it = 0; it < kKeep; ++it) {
      const int w = tid + it * NT;
      double s_ = 0.0;
      if (w < rc * n) {
        const int p = w / n, j = w - p * n;
        int m = p;
        for (int k = 0; k < rc; ++k) {
          s_ += fnT[(size_t)k * n + j] * ct[m];
          m += 2 * p;
          if (m >= 4 * rc) m -= 4 * rc;
        }
        s_ *= (p == 0 ? 1.0 : 2.0) / rc;
      }
      keep[it] = s_;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kKeep; ++it) {
      const int w = tid + it * NT;
      if (w < rc * n) res[w] = keep[it];
    }
  } else {
    for (int w = tid; w < rc * n; w += NT) {
      const int p = w / n, j = w - p * n;
      double s_ = 0.0;
      int m = p;
      for (int k = 0; k < rc; ++k) {
        s_ += fnT[(size_t)k * n + j] * ct[m];
        m += 2 * p;
        if (m >= 4 * rc) m -= 4 * rc;
      }
      res[w] = s_ * ((p == 0 ? 1.0 : 2.0) / rc);
    }
  }
  __syncthreads();
  double* U = Uout + (size_t)job * kBlMaxR * n;
  bool too_large = false;
  int r;
  double res2 = 0.0;
  if (regpath) {
    double fro2r = 0.0;
    if (rc == 32) r = basis_pivot_loop_reg<NT, 16>(n, resG, bl_dyn, U, qv, cf, red, redi, too_large, fro2r, res2);
    else if (rc == 48) r = basis_pivot_loop_reg<NT, 24>(n, resG, bl_dyn, U, qv, cf, red, redi, too_large, fro2r, res2);
    else if constexpr (NT <= 256) r = basis_pivot_loop_reg<NT, 32>(n, resG, bl_dyn, U, qv, cf, red, redi, too_large, fro2r, res2);
    else r = 0;
    stamp();
  } else {
  double fro2 = 0.0;
  for (int j = tid; j < n; j += NT) {
    double s_ = 0.0;
    for (int p = 0; p < rc; ++p) { const double v = res[(size_t)p * n + j]; s_ += v * v; }
    nrm2[j] = s_;
    fro2 += s_;
  }
  fro2 = bl_block_sum<NT>(fro2, red);
  __syncthreads();
  const double tol2 = (2e-16 * 2e-16) * fro2;
  stamp();
  if (rc <= 64)
    r = basis_pivot_loop<NT>(n, rc, res, bl_dyn, U, tol2, nrm2, qv, cf, red, redi, too_large, stamp, res2);
  else
    r = basis_pivot_loop<NT>(n, rc, res, QG, U, tol2, nrm2, qv, cf, red, redi, too_large, stamp, res2);
  }
  if (too_large || r < 1) {
    if (tid == 0) { inf[0] = 0; inf[1] = r; inf[2] = rc; inf[3] = 0; }
    return;
  }
  __syncthreads();
  const double* Q = rc <= 64 ? bl_dyn : QG;
  double* Vs = Vsout + (size_t)job * kBlMaxR * kBlMaxRc;
  for (int w = tid; w < r * rc; w += NT) Vs[w] = Q[w];
  if (tid < r) sigout[(size_t)job * kBlMaxR + tid] = 1.0;
  if (tid == 0) { inf[0] = 1; inf[1] = r; inf[2] = rc; inf[3] = 0; }
  // what the rank cut leaves of any factor, as a function on the axis: |sum_p R_jp T_p| <= ||R_j||_1 <= sqrt(rc) ||R_j||_2
  if (tid == 0 && axis_eps) axis_eps[2 * job + 1] = sqrt((double)rc * res2);
  __syncthreads();
  stamp();
  if (dbg && job == 0 && tid == 0) dbg[80] = ndbg;
}

// Axis tables S[p][i] = sig_p sum_c Vs[p][c] T_c(xi_i) from the Chebyshev series of the bases (three-term recurrence, sum
// in ascending c, one thread per entry); blockIdx.y = 2 o + axis
__global__ __launch_bounds__(256) void k_bl_stab(const BlDims dm, const double* __restrict__ Vsall, const double* __restrict__ sigall,
                                                 const double* __restrict__ xn0, const double* __restrict__ xn1, double* __restrict__ S0all,
                                                 double* __restrict__ S1all) {
  const int job = blockIdx.y, o = job >> 1, axis = job & 1;
  const int r = axis ? dm.r1[o] : dm.r0[o], rc = axis ? dm.rc1[o] : dm.rc0[o];
  const long long count = axis ? dm.nlines : dm.cnt0;
  const double a = dm.a[axis], b = dm.b[axis];
  const double* Vs = Vsall + (size_t)job * kBlMaxR * kBlMaxRc;
  const double* sig = sigall + (size_t)job * kBlMaxR;
  const double* xn = axis ? xn1 : xn0;
  double* S = axis ? S1all + (size_t)o * dm.r1u * dm.nlines : S0all + (size_t)o * dm.r0u * dm.cnt0;
  const long long total = (long long)r * count;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(idx / count);
    const long long i = idx % count;
    double xi = (2.0 * xn[i] - (a + b)) / (b - a);
    xi = xi < 1.0 ? xi : 1.0;
    xi = xi > -1.0 ? xi : -1.0;
    const double* v = Vs + (size_t)p * rc;
    double t0 = 1.0, t1 = xi, s_ = 0.0;
    s_ += v[0] * t0;
    if (rc > 1) s_ += v[1] * t1;
    for (int c2 = 2; c2 < rc; ++c2) {
      const double t = 2.0 * xi * t1 - t0;
      s_ += v[c2] * t;
      t0 = t1;
      t1 = t;
    }
    S[idx] = sig[p] * s_;
  }
}

// Z_j,(p,s) = U0_jp U1_js as B fragments [ncsR][KBn * 4][64] (k = observation j, column c = p r1 + s); blockIdx.y = o
__global__ __launch_bounds__(256) void k_bl_zf(const BlDims dm, const double* __restrict__ Uall, size_t nZf, double* __restrict__ Zfall) {
  const int o = blockIdx.y, n = dm.n, r0 = dm.r0[o], r1 = dm.r1[o], KBn = dm.KBn;
  const double* U0 = Uall + (size_t)(2 * o) * kBlMaxR * n;
  const double* U1 = Uall + (size_t)(2 * o + 1) * kBlMaxR * n;
  double* Zf = Zfall + (size_t)o * nZf;
  const long long total = (long long)dm.ncsR * KBn * 4 * 64;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    const long long fr = i >> 6;
    const int ks = (int)(fr % (KBn * 4)), cs = (int)(fr / (KBn * 4));
    const int j = (ks >> 2) * 16 + MM<double>::jslot(ks & 3, l >> 4);
    const int cidx = cs * 16 + (l & 15);
    double v = 0.0;
    if (j < n && cidx < r0 * r1) v = U0[(size_t)(cidx / r1) * n + j] * U1[(size_t)(cidx % r1) * n + j];
    Zf[i] = v;
  }
}
// T4qq^T as B fragments over the columns k0: [KB0][KB1 * 4][64], element (k1, k0) = sf2^2 / 2 (G[(p,s),(p',s')] + G[(p',s),(p,s')])
// (sym: G is symmetric only up to rounding -- the form with a caller's invK -- and both transposed entries are averaged in)
__global__ __launch_bounds__(256) void k_bl_t4f(const BlDims dm, const double* __restrict__ Gall, long long ldg, size_t sT4f,
                                                double* __restrict__ T4fall, int sym) {
  const int o = blockIdx.y, r0 = dm.r0[o], r1 = dm.r1[o], KB0 = dm.KB0, KB1 = dm.KB1;
  const int K0 = r0 * (r0 + 1) / 2, K1 = r1 * (r1 + 1) / 2;
  const double* G = Gall + (size_t)o * ldg * ldg;
  double* T4f = T4fall + (size_t)o * sT4f;
  const double scale = dm.sf2[o] * dm.sf2[o];
  const long long total = (long long)KB0 * KB1 * 4 * 64;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    const long long fr = i >> 6;
    const int ks = (int)(fr % (KB1 * 4)), cs = (int)(fr / (KB1 * 4));
    const int k1 = (ks >> 2) * 16 + MM<double>::jslot(ks & 3, l >> 4);
    const int k0 = cs * 16 + (l & 15);
    double v = 0.0;
    if (k0 < K0 && k1 < K1) {
      int p, pp, s1, ss;
      pair_of(k0, r0, p, pp);
      pair_of(k1, r1, s1, ss);
      const size_t a1 = (size_t)(p * r1 + s1), b1 = (size_t)(pp * r1 + ss), a2 = (size_t)(pp * r1 + s1), b2 = (size_t)(p * r1 + ss);
      v = sym ? scale * 0.25 * ((G[a1 * ldg + b1] + G[b1 * ldg + a1]) + (G[a2 * ldg + b2] + G[b2 * ldg + a2]))
              : scale * 0.5 * (G[a1 * ldg + b1] + G[a2 * ldg + b2]);
    }
    T4f[i] = v;
  }
}
// ---- mean-phase operands -------------------------------------------------------------------------------------------------
// Mb[o][b][p r1 + s] = sf2 sum_j beta_b[j] U0_jp U1_js, beta = (alpha, alpha Xn_0, alpha Xn_1): the bilinear forms of the
// mean and of its two gradient sums.  One wave per (b, p, s), lanes over the observations; blockIdx.y = o
__global__ __launch_bounds__(256) void k_bl_mb(const BlDims dm, const double* __restrict__ Uall, const double* __restrict__ alpha, int ald,
                                               const double* __restrict__ Xn, int dpad, double* __restrict__ Mball) {
  const int o = blockIdx.y, n = dm.n, r0 = dm.r0[o], r1 = dm.r1[o];
  const double* U0 = Uall + (size_t)(2 * o) * kBlMaxR * n;
  const double* U1 = Uall + (size_t)(2 * o + 1) * kBlMaxR * n;
  const double* al = alpha + (size_t)o * ald;
  double* Mb = Mball + (size_t)o * 3 * dm.r0u * dm.r1u;
  const int lane = threadIdx.x & 63, total = 3 * r0 * r1;
  for (int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < total; i += (gridDim.x * blockDim.x) >> 6) {
    const int s_ = i % r1, p = (i / r1) % r0, b = i / (r0 * r1);
    const double *u0 = U0 + (size_t)p * n, *u1 = U1 + (size_t)s_ * n;
    double acc = 0.0;
    for (int j = lane; j < n; j += 64) {
      const double be = b == 0 ? al[j] : al[j] * Xn[(size_t)j * dpad + (b - 1)];
      acc += be * u0[j] * u1[j];
    }
    acc = wave_sum(acc);
    if (lane == 0) Mb[i] = dm.sf2[o] * acc;
  }
}
// Vb[o][b][p][line] = sum_s Mb[b][p, s] S1[s][line]
__global__ __launch_bounds__(256) void k_bl_vb(const BlDims dm, const double* __restrict__ Mball, const double* __restrict__ S1all,
                                               double* __restrict__ Vball) {
  const int o = blockIdx.y, r0 = dm.r0[o], r1 = dm.r1[o], r0u = dm.r0u;
  const long long nlines = dm.nlines;
  const double* Mb = Mball + (size_t)o * 3 * dm.r0u * dm.r1u;
  const double* S1 = S1all + (size_t)o * dm.r1u * nlines;
  double* Vb = Vball + (size_t)o * 3 * r0u * nlines;
  const long long total = 3ll * r0 * nlines;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long l = i % nlines;
    const int p = (int)((i / nlines) % r0), b = (int)(i / (nlines * r0));
    double acc = 0.0;
    for (int s_ = 0; s_ < r1; ++s_) acc += Mb[((size_t)b * r0 + p) * r1 + s_] * S1[(size_t)s_ * nlines + l];
    Vb[((size_t)b * r0u + p) * nlines + l] = acc;
  }
}
// A images of the mean phases, three sets:  V0 (K = 16 KBm) | [V1; V0] (K = 16 KBm2, V0 from k = r0p) | V1x = V2 - xn1 V0
__global__ __launch_bounds__(256) void k_bl_va(const BlDims dm, const double* __restrict__ Vball, const double* __restrict__ xn1, size_t sVA,
                                               double* __restrict__ outall) {
  const int o = blockIdx.y, nrb = dm.nrb, KBm = dm.KBm, KBm2 = dm.KBm2, r0 = dm.r0[o], r0p = dm.r0p, r0u = dm.r0u;
  const long long nlines = dm.nlines;
  const double* Vb = Vball + (size_t)o * 3 * r0u * nlines;
  double* out = outall + (size_t)o * sVA;
  const long long set = (long long)nrb * KBm * 256, set2 = (long long)nrb * KBm2 * 256, total = 2 * set + set2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int which = i < set ? 0 : (i < set + set2 ? 1 : 2);
    const long long li = which == 0 ? i : (which == 1 ? i - set : i - set - set2);
    const int KBx = which == 1 ? KBm2 : KBm;
    // invert pack_pos(r, slot, kk) = kk * 64 + (slot * 4 + (r & 3)) * 4 + (r >> 2)
    const int e = (int)(li & 255);
    const long long blk = li >> 8;
    const int kk = e >> 6, rem = e & 63, slot = rem >> 4, r = ((rem >> 2) & 3) + 4 * (rem & 3);
    const int k = (int)(blk % KBx) * 16 + MM<double>::jslot(kk, slot);
    const long long line = (blk / KBx) * 16 + r;
    double v = 0.0;
    if (line < nlines) {
      if (which == 0) {
        if (k < r0) v = Vb[((size_t)0 * r0u + k) * nlines + line];
      } else if (which == 1) {
        if (k < r0) v = Vb[((size_t)1 * r0u + k) * nlines + line];
        else if (k >= r0p && k < r0p + r0) v = Vb[((size_t)0 * r0u + (k - r0p)) * nlines + line];
      } else if (k < r0) {
        v = Vb[((size_t)2 * r0u + k) * nlines + line] - xn1[line] * Vb[((size_t)0 * r0u + k) * nlines + line];
      }
    }
    out[i] = v;
  }
}
// B fragments of the mean phases, two sets:  S0 (K = 16 KBm) | [S0; -xn0 S0] (K = 16 KBm2, second copy from k = r0p)
__global__ __launch_bounds__(256) void k_bl_sbf(const BlDims dm, const double* __restrict__ S0all, const double* __restrict__ xn0, size_t sSBf,
                                                double* __restrict__ outall) {
  const int o = blockIdx.y, ncs0 = dm.ncs0, KBm = dm.KBm, KBm2 = dm.KBm2, r0 = dm.r0[o], r0p = dm.r0p;
  const long long cnt0 = dm.cnt0;
  const double* S0 = S0all + (size_t)o * dm.r0u * cnt0;
  double* out = outall + (size_t)o * sSBf;
  const long long fset = (long long)ncs0 * KBm * 256, total = fset + (long long)ncs0 * KBm2 * 256;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int which = i < fset ? 0 : 1;
    const long long li = which ? i - fset : i;
    const int KBx = which ? KBm2 : KBm;
    const int l = (int)(li & 63);
    const long long fr = li >> 6;
    const int ks = (int)(fr % (KBx * 4));
    const int k = (ks >> 2) * 16 + MM<double>::jslot(ks & 3, l >> 4);
    const long long x = (fr / (KBx * 4)) * 16 + (l & 15);
    double v = 0.0;
    if (x < cnt0) {
      if (k < r0) v = S0[(size_t)k * cnt0 + x];
      else if (which == 1 && k >= r0p && k < r0p + r0) v = -xn0[x] * S0[(size_t)(k - r0p) * cnt0 + x];
    }
    out[i] = v;
  }
}

// Fused stage 2: variance, mean and gradient keys of a 128 x 128 tile of the grid (8 row blocks x 8 column strips) per
// workgroup.  Four GEMM phases share the accumulators; operands are staged through LDS one 16-deep k-block at a time
// (double-buffered), so every fragment is fetched from L2 once per workgroup instead of once per wave:
//   phase 0  quad = Bt    . P0          (KS0 k-steps)  ->  var  = max(0, sf2 - quad) Y_std^2
//   phase 1  s1   = V[0]  . S0          (KSm)          ->  mean = (mp + s1) Y_std + Y_mean
//   phase 2  g0   = [V[1]; V[0]] . [S0; -xn0 S0]  (2 KSm)   gradient sum of axis 0 (the candidate's own xn0 sits in B)
//   phase 3  g1   = V1x   . S0          (KSm)              gradient sum of axis 1 (xn1 of the line is folded into V1x)

// = kClassifyRow of sets.hip: u* key, |S|, |U|, decisions inside the guard band, min-variance keys over S, radius keys
constexpr int kFuseRow = 4 + 2 * kMaxQ;
constexpr int kFuseVmin = 4, kFuseRmax = 4 + kMaxQ;
struct PostCtx {
  double* lds;
  int tid, lane, wave, rb0, cs0, nrb, ncs;
  int st_rb, st_cs, st_off, a_lo, b_st, a_rd;
  unsigned int ucnt0;
  long long nlines;
  bool full;          // the workgroup's 128 x 128 tile lies inside the grid: epilogues skip their bounds tests
  bool imode;         // K1i: every phase is a plain GEMM on its own Chebyshev coefficients (the gradient phases start from zero)
  // fused classification (one-constraint sweeps): the mean epilogue of the constraint's output reads back the variances this
  // thread stored in the variance phase and writes the S / U bytes; null = off
  const double* var_rd;
  uint8_t *S, *U;
  double bconf, bb;   // confidence multiplier and its square
  int cS, cU;         // this thread's counts
  double rmax;        // max ucb over its safe candidates (-1: none; ucb >= lcb >= 0 on S)
  // guard band of this posterior (guard.hip): sign tests the band of the constraint's output could move are counted, and the
  // smallest variance over the thread's safe candidates is kept (gdm < 0: no band in force)
  LcbBand lband;
  bool gb_on;
  double vminS;
  int cB;
  // column-word classification (r05): role 1 = the constraint's tile packs its S / U bits (bw: S in the low, U in the high 16
  // bits, bit 4 t + (lane >> 4) of the wave's 16 rows, one word per strip); role 2 = the objective's tile reads the S bits of its
  // candidates back (bw: its own four bits per strip at 0, 4, 8, 12) and keeps u* / min var_0 over them
  int role;
  unsigned int bw[8];
  unsigned int* lds_bits;   // [4 waves][128 columns] pieces of role 1
  double umin;              // role 2: min ucb_0 over the thread's safe candidates (+inf: none)
  double xmin;              // role 1: min of a lower bound of ucb_1 over its safe candidates; role 2: of lcb_0 (the tile's range for the set phase)
  bool skip_store;          // lean sweep, objective tile without a safe candidate: mean / var are not stored (role 0)
};

// One GEMM phase of k_bpost on the workgroup's 128 x 128 tile: A images / B fragments of `KB` k-blocks per row block /
// strip, `KS` k-steps run.  PH selects the epilogue: 0 variance, 1 mean, 2 / 3 gradient component of axis 0 / 1.
// The accumulators belong to the caller: phase 2 does not start from zero but from phase 1's sums scaled by -xn0 of the
// candidate's column, g0 = V1 . S0 - xn0 (V0 . S0) -- six k-steps instead of twelve for the stacked [V1; V0] operand.
// S / U bits of one candidate from its stored mean / var: the arithmetic of k_classify (models/SafeOpt.py:37-43, 57-59, 73-77)
// (the sign of lcb without the square root -- lcb_sign, device_common.hpp -- and the exact ucb, for the radius key, only when
// its cheap upper bound beats this thread's running maximum: the IEEE f64 square root is a dozen dependent instructions on the
// datapath the matrix cores use, which is what made this epilogue cost the kernel as much as the separate pass saved)
__device__ __forceinline__ void post_classify_mv(PostCtx& cx, size_t g, double m, double v) {
  const LcbSign sg = cx.gb_on ? lcb_sign_gb(m, v, cx.bconf, cx.bb, cx.lband, cx.cB) : lcb_sign(m, v, cx.bconf, cx.bb);
  cx.S[g] = sg.ge;
  cx.U[g] = sg.le;
  cx.cS += sg.ge;
  cx.cU += sg.le;
  if (sg.ge) {
    cx.vminS = v < cx.vminS ? v : cx.vminS;
    if (!(ucb_upper(m, v, cx.bconf) <= cx.rmax)) {
      const double ucb = add_rn(m, mul_rn(cx.bconf, sqrt_rn(v)));
      if (ucb > cx.rmax) cx.rmax = ucb;
    }
  }
}
__device__ __forceinline__ void post_classify(PostCtx& cx, size_t g, double m) { post_classify_mv(cx, g, m, cx.var_rd[g]); }
// the same decisions as bits of the strip's word (role 1): no byte stores
// The classification of one tile row (eight strips) of the constraint, branch-lean: the kernel is bound by instruction issue on
// the datapath the matrix cores share, and three quarters of a grid's tiles hold no safe candidate at all.  The sign of
// lcb = m - b sqrt(v) as lcb_sign decides it (device_common.hpp) -- from m^2 against b^2 v wherever that is safe, as plain predicates
// without branches; what they leave open (a bound within a few ulp of zero, NaN, out-of-range products: rare) takes the IEEE
// square root behind ONE wave-uniform test per row.  |S| and |U| are the populations of the bit words at the end (no counters
// here); everything that concerns safe candidates only -- the smallest variance, the tile's range of ucb_1 -- runs behind a second
// uniform test, so a tile without a safe candidate never enters it.  Decisions identical to post_classify_mv.
__device__ __forceinline__ void post_classify_row(PostCtx& cx, unsigned int pos, const double (&mv)[8], const double (&vr)[8]) {
  constexpr double c = 1.0 + 0x1p-48, tiny = 1e-250, huge = 1e300;
  unsigned int gem = 0u, lem = 0u, und = 0u;
  const bool bok = cx.bconf >= 0.0;
#pragma unroll
  for (int s2 = 0; s2 < 8; ++s2) {
    const double m = mv[s2], v = vr[s2];
    const double P = m * m, Q = cx.bb * v;
    if (cx.gb_on) {
      const double gap = P > Q ? P - Q : Q - P, am = m < 0 ? -m : m;
      cx.cB += !(gap > fma(am, cx.lband.c1, cx.lband.c0));               // (NaN operands count as near)
    }
    const bool ok = bok && v >= 0.0, rng = P < huge && Q < huge;
    const bool neg = ok && m < 0.0;
    const bool ge = ok && !neg && rng && P > tiny && P >= Q * c;
    const bool le = neg || (ok && rng && !ge && Q > tiny && P * c <= Q && m >= 0.0);
    gem |= ge ? (1u << s2) : 0u;
    lem |= le ? (1u << s2) : 0u;
    und |= (!ge && !le) ? (1u << s2) : 0u;
  }
  if (__ballot(und != 0u) != 0ull) {
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
      if ((und >> s2) & 1u) {
        const double sd = mul_rn(cx.bconf, sqrt_rn(vr[s2]));
        gem |= (mv[s2] >= sd) ? (1u << s2) : 0u;
        lem |= (mv[s2] <= sd) ? (1u << s2) : 0u;
      }
    }
  }
#pragma unroll
  for (int s2 = 0; s2 < 8; ++s2) cx.bw[s2] |= (((gem >> s2) & 1u) | (((lem >> s2) & 1u) << 16)) << pos;
  if (__ballot(gem != 0u) != 0ull) {
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
      if ((gem >> s2) & 1u) {
        const double m = mv[s2], v = vr[s2];
        cx.vminS = v < cx.vminS ? v : cx.vminS;
        // bounds of ucb_1 from one single-precision square root (ucb_upper / ucb_lower, device_common.hpp: the hardware's 1-ulp root
        // is well inside their 2^-20 slack): the exact bound only when it could raise the radius key; the lower bound feeds the
        // tile's range
        const float sf = __builtin_amdgcn_sqrtf((float)v);
        const double s_up = (double)sf * (1.0 + 0x1p-20) + 1e-18;
        double s_lo = (double)sf * (1.0 - 0x1p-20) - 1e-18;
        s_lo = s_lo > 0.0 ? s_lo : 0.0;
        const bool fin = sf < 3.0e38f;
        const double xu = m + cx.bconf * s_up, xl = m + cx.bconf * (fin ? s_lo : 0.0);
        const double up = fin ? xu + (xu < 0 ? -xu : xu) * 0x1p-50 : 1e300, lo = xl - (xl < 0 ? -xl : xl) * 0x1p-50;
        cx.xmin = lo < cx.xmin ? lo : cx.xmin;
        if (!(up <= cx.rmax)) {
          const double ucb = add_rn(m, mul_rn(cx.bconf, sqrt_rn(v)));
          if (ucb > cx.rmax) cx.rmax = ucb;
        }
      }
    }
  }
}
// role 2: a safe candidate of the objective -- u* = min over S of ucb_0 (models/SafeOpt.py:47-51), the exact bound only when the
// cheap lower bound could beat the thread's running minimum (as k_classify's objective pass), and the smallest var_0 over S
__device__ __forceinline__ void post_objective(PostCtx& cx, bool set, double m, double v) {
  if (set) {
    cx.vminS = v < cx.vminS ? v : cx.vminS;
    const float sf = __builtin_amdgcn_sqrtf((float)v);
    const double s_up = (double)sf * (1.0 + 0x1p-20) + 1e-18;
    double s_lo = (double)sf * (1.0 - 0x1p-20) - 1e-18;
    s_lo = s_lo > 0.0 ? s_lo : 0.0;
    const bool fin = sf < 3.0e38f;
    const double xl = m + cx.bconf * (fin ? s_lo : 0.0), yl = m - cx.bconf * s_up;
    const double ulo = xl - (xl < 0 ? -xl : xl) * 0x1p-50;                      // ucb_0 >= ulo
    const double llo = fin ? yl - (yl < 0 ? -yl : yl) * 0x1p-50 : -1e300;       // lcb_0 >= llo: the tile's range for the minimiser
    cx.xmin = llo < cx.xmin ? llo : cx.xmin;
    if (ulo < cx.umin) {
      const double ucb = add_rn(m, mul_rn(cx.bconf, sqrt_rn(v)));
      cx.umin = ucb < cx.umin ? ucb : cx.umin;
    }
  }
}

// End of a posterior kernel: the waves' Lipschitz maxima (already reduced over the lanes) and, with the fused classification,
// their counts and radius maxima, merged through LDS into ONE value / row per workgroup.  `sh`: NW x 4 doubles of LDS nobody
// reads any more (barrier first).
template <int NW>
__device__ __forceinline__ void post_partials(double* sh, int lane, int wave, double gmax, bool fuse, int cS_, int cU_, double rmax_,
                                              int cB_, double vmin_, double* __restrict__ lrow, unsigned long long* __restrict__ crow /* this
                                              workgroup's row of the field-major partials */, int pcap,
                                              bool objrow = false /* the row of an objective tile (column path): rmax_ carries -min ucb_0 over
                                              its safe candidates (so that the maximum below is the minimum), vmin_ their smallest var_0 */,
                                              unsigned long long* __restrict__ slots = nullptr /* column path: the scalars also join slot
                                              `slot` of every field by atomics (internal.hpp: ColSlotField) */, int slot = 0, int o = 0,
                                              int tile_row = -1, int tile_col = 0,
                                              double xmin_ = 1e300 /* column path: the tile's smallest lower bound of ucb_1 (constraint rows,
                                              field 0) / of lcb_0 (objective rows, field 1) over its safe candidates */,
                                              bool multi = false /* several constraints: this row is constraint o's (its plane's populations are
                                              not |S| / |U|; its variance / radius keys are taken over ITS safe candidates, a superset of S) */) {
  int cS = cS_, cU = cU_, cB = cB_;
  double rm = rmax_, vm = vmin_, xm = xmin_;
  if (fuse || objrow) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      cS += __shfl_xor(cS, off);
      cU += __shfl_xor(cU, off);
      cB += __shfl_xor(cB, off);
      const double other = __shfl_xor(rm, off);
      rm = other > rm ? other : rm;
      const double ov = __shfl_xor(vm, off);
      vm = ov < vm ? ov : vm;
      const double ox = __shfl_xor(xm, off);
      xm = ox < xm ? ox : xm;
    }
  }
  __syncthreads();
  if (lane == 0) {
    sh[wave * 6 + 0] = gmax;
    sh[wave * 6 + 1] = rm;
    reinterpret_cast<int*>(sh + wave * 6 + 2)[0] = cS;
    reinterpret_cast<int*>(sh + wave * 6 + 2)[1] = cU;
    reinterpret_cast<int*>(sh + wave * 6 + 3)[0] = cB;
    sh[wave * 6 + 4] = vm;
    sh[wave * 6 + 5] = xm;
  }
  __syncthreads();
  if (wave == 0) {
    double g = sh[0], r = sh[1], vmn = sh[4], xmn = sh[5];
    long long s_ = 0, u_ = 0, b_ = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      g = sh[w * 6] > g ? sh[w * 6] : g;
      r = sh[w * 6 + 1] > r ? sh[w * 6 + 1] : r;
      vmn = sh[w * 6 + 4] < vmn ? sh[w * 6 + 4] : vmn;
      xmn = sh[w * 6 + 5] < xmn ? sh[w * 6 + 5] : xmn;
      s_ += reinterpret_cast<const int*>(sh + w * 6 + 2)[0];
      u_ += reinterpret_cast<const int*>(sh + w * 6 + 2)[1];
      b_ += reinterpret_cast<const int*>(sh + w * 6 + 3)[0];
    }
    if (lane == 0) *lrow = g;
    if (slots) {
      // (no return values: the wave does not wait for them)
      if (lane == 0) atomicMax(&slots[(size_t)(o == 0 ? kSlotL0 : kSlotL1) * kColSlots + slot], (unsigned long long)__double_as_longlong(g));
      if (objrow) {
        if (lane == 1 && r > -1e300) atomicMin(&slots[(size_t)kSlotUmin * kColSlots + slot], ord_key(-r));
        if (lane == 2 && vmn < 1e300) atomicMin(&slots[(size_t)kSlotVmin0 * kColSlots + slot], ord_key(vmn));
      } else {
        if (lane == 1 && s_ != 0) atomicAdd(&slots[(size_t)kSlotS * kColSlots + slot], (unsigned long long)s_);
        if (lane == 2 && u_ != 0) atomicAdd(&slots[(size_t)kSlotU * kColSlots + slot], (unsigned long long)u_);
        if (lane == 3 && b_ != 0) atomicAdd(&slots[(size_t)kSlotB * kColSlots + slot], (unsigned long long)b_);
        if (lane == 4 && vmn < 1e300) atomicMin(&slots[(size_t)kSlotVmin1 * kColSlots + slot], ord_key(vmn));
        if (lane == 5 && r >= 0.0) atomicMax(&slots[(size_t)kSlotRmax1 * kColSlots + slot], ord_key(r));
        if (lane == 6 && s_ != 0 && tile_row >= 0) atomicOr(&slots[(size_t)kSlotRowMask * kColSlots + tile_row], 1ull << tile_col);
      }
    }
    if (objrow && lane < kFuseRow) {
      // [u* key, 0, 0, 0, min-variance key of output 0, none.., no radius keys]
      unsigned long long v = 0ull;
      if (lane == 0) v = r > -1e300 ? ord_key(-r) : ~0ull;
      else if (lane == 1) v = xmn < 1e300 ? ord_key(xmn) : ~0ull;                 // (the tile's range of lcb_0 over S: sets_colpath's minimiser)
      else if (lane >= kFuseVmin && lane < kFuseRmax) v = (lane == kFuseVmin && vmn < 1e300) ? ord_key(vmn) : ~0ull;
      crow[(size_t)lane * pcap] = v;
    } else
    if (fuse && lane < kFuseRow) {
      // partial row of the classification, merged by k_classify_final: [u* key (none here), |S|, |U|, decisions inside the guard
      // band, min-variance keys over S (output 1 only), radius keys (constraint 1 only)]
      unsigned long long v = 0ull;
      if (lane == 0) v = (slots && xmn < 1e300) ? ord_key(xmn) : ~0ull;           // (column path: the tile's lower end of ucb_1 over S)
      else if (lane == 1) v = multi ? 0ull : (unsigned long long)s_;
      else if (lane == 2) v = multi ? 0ull : (unsigned long long)u_;
      else if (lane == 3) v = (unsigned long long)b_;
      else if (lane >= kFuseVmin && lane < kFuseRmax) v = (lane == kFuseVmin + (o >= 1 ? o : 1) && vmn < 1e300) ? ord_key(vmn) : ~0ull;
      else if (lane == kFuseRmax + (o >= 1 ? o : 1)) v = r >= 0.0 ? ord_key(r) : 0ull;       // radius key of this constraint
      crow[(size_t)lane * pcap] = v;
    }
  }
}

// Epilogue of phase PH for the RB x 8 accumulator tiles of a wave (row blocks cx.rb0 + RB cx.wave + i, strips cx.cs0 + s2)
template <int PH, int RB, int ROLE>
__device__ __forceinline__ void post_epilogue(PostCtx& cx, double* __restrict__ outp, double c0, double c1, double c2, double& gmax_io,
                                              d4_t (&acc)[RB][8]) {
  // gradient phases: max |c0 v| = fl(|c0| max |v|) -- rounding is monotone --, so the tile keeps max |v| (one v_max_f64 with
  // |.| modifiers per element on the datapath the matrix cores share) and scales once
  double gmax = 0.0;
  struct Fold {
    double& io; const double& raw; double c0; bool on;
    __device__ ~Fold() {
      if (on) {
        double ga = c0 * raw;
        ga = ga < 0 ? -ga : ga;
        io = ga > io ? ga : io;
      }
    }
  } fold{gmax_io, gmax, c0, PH >= 2};
  // epilogue: accumulator element t of lane l is row 4 t + (l >> 4), column l & 15 of its 16 x 16 tile
  const unsigned int col_in = cx.lane & 15, row_in = cx.lane >> 4;
  if (PH == 1 && RB == 1 && ROLE != 0 && cx.role != 0) {
    // Column path (r05; its tiles are all interior).  Role 1, the constraint: the classification with its S / U decisions packed as
    // bits of the strip's word.  Role 2, the objective: u* and the range of lcb_0 over the tile's safe candidates (bits read back
    // from the constraint's launch).  Both need the variances this thread stored in the variance phase: the eight of row t + 1 are
    // requested BEFORE row t's means are stored (the stores may alias anything as far as the compiler knows, so it would not move
    // the loads across them itself) -- three of the four round trips to L2 run under the arithmetic of the row before.
    auto row_g0 = [&](int t) {
      const unsigned int line = (unsigned int)(cx.rb0 + cx.wave) * 16u + 4u * t + row_in;
      return (size_t)line * cx.ucnt0 + (unsigned int)cx.cs0 * 16u + col_in;
    };
    // (the objective's tiles: 2 us per launch on config H; the constraint's epilogue holds too much in registers for it -- with the
    // prefetch its allocation spilt 22 vector registers and the launch took 138 us against 132)
    constexpr bool kPrefetch = ROLE == 2;
    double vn[8];
    if (kPrefetch) {
      const size_t g0 = row_g0(0);
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) vn[s2] = cx.var_rd[g0 + s2 * 16];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const size_t g0 = row_g0(t), g1 = row_g0(t + 1 < 4 ? t + 1 : t);
      double* const rowp = outp + g0;
      if (ROLE == 1) {
        double vr[8], mv[8];
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) vr[s2] = cx.var_rd[g0 + s2 * 16];
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
          mv[s2] = (c0 + acc[0][s2][t]) * c1 + c2;
          rowp[s2 * 16] = mv[s2];
        }
        post_classify_row(cx, 4u * t + row_in, mv, vr);
      } else {
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
          const double v = vn[s2];
          if (t + 1 < 4) vn[s2] = cx.var_rd[g1 + s2 * 16];       // (the next row's variance: a row of arithmetic ahead of its use)
          const double m = (c0 + acc[0][s2][t]) * c1 + c2;
          rowp[s2 * 16] = m;
          post_objective(cx, ((cx.bw[s2] >> (4 * t)) & 1u) != 0u, m, v);
        }
      }
    }
    if (ROLE == 1) {
      // |S| / |U| of this thread: the populations of its words (S low half, U high half)
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) { cx.cS += __popc(cx.bw[s2] & 0xffffu); cx.cU += __popc(cx.bw[s2] >> 16); }
      // the wave's 16 rows of every column: the four lanes that hold a column OR their bits together, lanes 0..15 put the piece
      // (S low half, U high half) where the end of the kernel assembles the 64-bit words of the tile
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) {
        unsigned int w = cx.bw[s2];
        w |= (unsigned int)__shfl_xor((int)w, 16);
        w |= (unsigned int)__shfl_xor((int)w, 32);
        if (cx.lane < 16) cx.lds_bits[cx.wave * 128 + s2 * 16 + cx.lane] = w;
      }
    }
    return;
  }
  if (cx.full) {
    // interior tile: no bounds tests, one pointer per row, the eight strips at immediate offsets.  The matrix cores
    // share the f64 VALU datapath, so every instruction saved here is matrix time.
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const unsigned int line = (unsigned int)(cx.rb0 + RB * cx.wave + i) * 16u + 4u * t + row_in;
        const size_t g0 = (size_t)line * cx.ucnt0 + (unsigned int)cx.cs0 * 16u + col_in;
        double* const rowp = outp + g0;
        if (PH == 1 && ROLE == 0 && cx.S) {
          // fused classification: the eight variances of this row first (all loads in flight; the byte stores below may
          // alias anything as far as the compiler knows), then bounds, S / U bytes and the partial sums
          double vr[8], mv[8];
#pragma unroll
          for (int s2 = 0; s2 < 8; ++s2) vr[s2] = cx.var_rd[g0 + s2 * 16];
#pragma unroll
          for (int s2 = 0; s2 < 8; ++s2) {
            mv[s2] = (c0 + acc[i][s2][t]) * c1 + c2;
            rowp[s2 * 16] = mv[s2];
          }
#pragma unroll
          for (int s2 = 0; s2 < 8; ++s2) post_classify_mv(cx, g0 + s2 * 16, mv[s2], vr[s2]);
          continue;
        }
        if (PH <= 1 && RB == 1 && ROLE == 2 && cx.skip_store) continue;     // (lean sweep: nobody reads this tile's mean / var)
#pragma unroll
        for (int s2 = 0; s2 < 8; ++s2) {
          const double v = acc[i][s2][t];
          if (PH == 0) {
            double var = c0 - v;
            var = var > 0.0 ? var : 0.0;
            rowp[s2 * 16] = var * c1;
          } else if (PH == 1) {
            rowp[s2 * 16] = (c0 + v) * c1 + c2;
          } else {
            gmax = fmax(gmax, fabs(v));
          }
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int rb = cx.rb0 + RB * cx.wave + i;
    if (rb >= cx.nrb) continue;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned int line = (unsigned int)rb * 16u + 4u * t + row_in;
      if ((long long)line >= cx.nlines) continue;
      double* const rowp = outp + (size_t)line * cx.ucnt0;
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) {
        const unsigned int x0 = (unsigned int)(cx.cs0 + s2) * 16u + col_in;
        if (cx.cs0 + s2 >= cx.ncs || x0 >= cx.ucnt0) continue;
        const double v = acc[i][s2][t];
        if (PH == 0) {
          double var = c0 - v;                                              // models/GP_Safe.py:343, clipped at 0
          var = var > 0.0 ? var : 0.0;
          rowp[x0] = var * c1;                                              // :347
        } else if (PH == 1) {
          const double m = (c0 + v) * c1 + c2;                              // :342, :346
          rowp[x0] = m;
          if (ROLE == 0 && cx.S) post_classify(cx, (size_t)line * cx.ucnt0 + x0, m);
        } else {
          // component of the gradient of the un-normalised mean (analytic jax.grad(self.mean), SafeOpt.py:68-71)
          gmax = fmax(gmax, fabs(v));
        }
      }
    }
  }
}

template <int PH, int RB, int ROLE>
__device__ __forceinline__ void post_phase(PostCtx& cx, const double* __restrict__ A, const double* __restrict__ B, int KB,
                                           int KS, double* __restrict__ outp, double c0, double c1, double c2, double& gmax,
                                           d4_t (&acc)[RB][8], const double* __restrict__ xn0, d4_t (&pre)[4],
                                           const double* __restrict__ An, const double* __restrict__ Bn, int KBn) {
  const double* Ap = A + (size_t)cx.st_rb * KB * 256 + cx.st_off;
  const double* Bp = B + (size_t)cx.st_cs * KB * 256 + cx.st_off;
  const int nkb = (KS + 3) >> 2;
  double* const lds = cx.lds;
  auto stage = [&](double* buf, const d4_t& r0, const d4_t& r1, const d4_t& q0, const d4_t& q1) {
    if (RB == 2 || cx.a_lo >= 0) {
      *reinterpret_cast<d4_t*>(buf + cx.a_lo) = d4_t{r0[0], r0[1], r1[0], r1[1]};
      *reinterpret_cast<d4_t*>(buf + cx.a_lo + 32) = d4_t{r0[2], r0[3], r1[2], r1[3]};
    }
    *reinterpret_cast<d4_t*>(buf + cx.b_st) = q0;
    *reinterpret_cast<d4_t*>(buf + cx.b_st + 4) = q1;
  };
  if (PH == 2 && !cx.imode) {
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
      const unsigned int x = (unsigned int)(cx.cs0 + s2) * 16u + (cx.lane & 15);
      const double f = x < cx.ucnt0 ? -xn0[x] : 0.0;
#pragma unroll
      for (int i = 0; i < RB; ++i) acc[i][s2] = d4_t{acc[i][s2][0] * f, acc[i][s2][1] * f, acc[i][s2][2] * f, acc[i][s2][3] * f};
    }
  } else {
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) acc[i][s2] = d4_t{0.0, 0.0, 0.0, 0.0};
  }
  const bool stage_a = RB == 2 || cx.a_lo >= 0;
  // the first k-block's 64 + 64 bytes of this thread: requested by the previous phase before its epilogue (`pre`), so that
  // their latency runs under that epilogue instead of in front of this loop; phase 0 asks here
  // (128 x 128 tiles only: the 64 x 128 form runs three workgroups per CU on 168 registers and would spill the four)
  d4_t &ra0 = pre[0], &ra1 = pre[1], &rb0v = pre[2], &rb1v = pre[3];
  if (PH == 0 || RB == 1) {
    ra0 = ra1 = d4_t{0.0, 0.0, 0.0, 0.0};
    if (stage_a) { ra0 = *reinterpret_cast<const d4_t*>(Ap); ra1 = *reinterpret_cast<const d4_t*>(Ap + 4); }
    rb0v = *reinterpret_cast<const d4_t*>(Bp);
    rb1v = *reinterpret_cast<const d4_t*>(Bp + 4);
  }
  __syncthreads();                             // the previous phase has finished reading the buffers
  stage(lds, ra0, ra1, rb0v, rb1v);
  __syncthreads();
#pragma unroll 1
  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    if (kb + 1 < nkb) {
      if (stage_a) {
        ra0 = *reinterpret_cast<const d4_t*>(Ap + (size_t)(kb + 1) * 256);
        ra1 = *reinterpret_cast<const d4_t*>(Ap + (size_t)(kb + 1) * 256 + 4);
      }
      rb0v = *reinterpret_cast<const d4_t*>(Bp + (size_t)(kb + 1) * 256);
      rb1v = *reinterpret_cast<const d4_t*>(Bp + (size_t)(kb + 1) * 256 + 4);
    }
    constexpr int BUF = RB == 2 ? 4096 : 3072, BOFF = RB == 2 ? 2048 : 1024;   // doubles per buffer: 4 RB A images, 8 B strips
    const double* LA = lds + cur * BUF + (RB * cx.wave) * 256 + cx.a_rd;
    const double* LB = lds + cur * BUF + BOFF + cx.lane;
    const int kkn = KS - kb * 4 < 4 ? KS - kb * 4 : 4;
    // operands of k-step kk + 1 are requested before the products of kk are issued (two register sets, no copies): the LDS
    // latency of a k-step's ten reads otherwise sits in front of its 64 matrix instructions
    d4_t fa[2][RB];
    double fb[2][8];
    auto ld = [&](int kk, d4_t (&a)[RB], double (&b)[8]) {
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const d2_t l = *reinterpret_cast<const d2_t*>(LA + i * 256 + kk * 64), h = *reinterpret_cast<const d2_t*>(LA + i * 256 + kk * 64 + 32);
        a[i] = d4_t{l[0], l[1], h[0], h[1]};
      }
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) b[s2] = LB[(s2 * 4 + kk) * 64];
    };
    auto mm = [&](const d4_t (&a)[RB], const double (&b)[8]) {
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2)
#pragma unroll
        for (int i = 0; i < RB; ++i) acc[i][s2] = MM<double>::mfma(a[i], b[s2], acc[i][s2]);
    };
    if constexpr (RB == 2) {
      ld(0, fa[0], fb[0]);
      int kk = 0;
#pragma unroll 1
      for (; kk + 1 < kkn; kk += 2) {
        ld(kk + 1, fa[1], fb[1]);
        mm(fa[0], fb[0]);
        if (kk + 2 < kkn) ld(kk + 2, fa[0], fb[0]);
        mm(fa[1], fb[1]);
      }
      if (kk < kkn) mm(fa[0], fb[0]);
    } else {                               // (three workgroups per CU and 168 registers: no room for a second operand set)
#pragma unroll 1
      for (int kk = 0; kk < kkn; ++kk) {
        ld(kk, fa[0], fb[0]);
        mm(fa[0], fb[0]);
      }
    }
    if (kb + 1 < nkb) stage(lds + (cur ^ 1) * BUF, ra0, ra1, rb0v, rb1v);
    __syncthreads();
  }
  if (RB == 2 && An) {
    const double* Apn = An + (size_t)cx.st_rb * KBn * 256 + cx.st_off;
    const double* Bpn = Bn + (size_t)cx.st_cs * KBn * 256 + cx.st_off;
    if (stage_a) { ra0 = *reinterpret_cast<const d4_t*>(Apn); ra1 = *reinterpret_cast<const d4_t*>(Apn + 4); }
    rb0v = *reinterpret_cast<const d4_t*>(Bpn);
    rb1v = *reinterpret_cast<const d4_t*>(Bpn + 4);
  }
  post_epilogue<PH, RB, ROLE>(cx, outp, c0, c1, c2, gmax, acc);
}

#ifdef SBO_PHASE_CLOCKS
// Diagnostic build only (make phaseclk -> libsafebo_phaseclk.so, tools/dev_phase_clocks.py): the constant 100 MHz counter at the
// phase boundaries of every k_bpost workgroup, a row per workgroup (same-address atomics from 4096 workgroups would themselves
// take 0.1 ms); [4] = gradient phases the workgroup ran.  The host sums the rows.
constexpr int kPhaseClkRows = 1 << 14;
__device__ unsigned long long g_phase_clk[kPhaseClkRows][8];
#define SBO_CLK(i)                                                                                \
  do {                                                                                           \
    __syncthreads();                                                                             \
    if (threadIdx.x == 0) {                                                                      \
      const unsigned long long now_ = wall_clock64();                                            \
      g_phase_clk[clk_row_ & (kPhaseClkRows - 1)][i] += now_ - clk_;                             \
      clk_ = now_;                                                                               \
    }                                                                                            \
  } while (0)
#else
#define SBO_CLK(i) do { } while (0)
#endif

// RB: row blocks per wave.  2 = the 128 x 128 tile above; 1 = a 64 x 128 tile for grids whose 128 x 128 tiles would leave
// CUs without a workgroup (1024 x 1024 x 3 outputs: 192 tiles on 256 CUs) -- half the reuse of a B fragment, twice the
// workgroups.
template <int RB, int ROLE = 0 /* column path: 1 = the constraint's launch (S / U column words), 2 = the objective's (u* over S) */>
__global__ __launch_bounds__(256, (RB == 2 ? 2 : 3)) void k_bpost(const ModelConst mc, const CandSpec cs, const double* __restrict__ BtA, size_t sBtA,
                                                  const double* __restrict__ P0f, size_t sP0f, const double* __restrict__ VA,
                                                  size_t sVA, const double* __restrict__ SBf, size_t sSBf, int KB0, int KS0, int KBm,
                                                  int KSm, int KBm2, int nrb, int ncs, long long nlines, double* __restrict__ mean_out,
                                                  double* __restrict__ var_out, double* __restrict__ Lpart,
                                                  const double* __restrict__ xn0, uint8_t* __restrict__ Sfuse, uint8_t* __restrict__ Ufuse,
                                                  double bconf, unsigned long long* __restrict__ cpart /* [kFuseRow][pcap]: a row per workgroup of output 1 */,
                                                  int pcap,
                                                  const GuardBand* __restrict__ gb /* nullptr: no guard band */,
                                                  const int* __restrict__ eff /* nullptr, or the Chebyshev core's k-steps of the variance phase at [4 o] */,
                                                  const double* __restrict__ gtmax /* nullptr (every tile runs the gradient phases), or the plan's
                                                  largest gradient samples per tile [q][2][tiles], followed by the slacks [q][2] */,
                                                  const unsigned long long* __restrict__ gkey /* the grid's largest samples [q][2] */,
                                                  int imode /* K1i (interpolation from Chebyshev nodes): BtA / VA hold the stage-1 images of four
                                                  coefficient sets per output (quadratic form, mean sum, two gradient sums), SBf the Chebyshev table
                                                  P0f; the k-steps of every phase come from eff[4 (4 o + phase)] */,
                                                  const PostExtra px /* column path: one launch per output (o0), the S / U column words */) {
  extern __shared__ double lds[];               // [2][A: 8 x 256 | B: 8 x 256] (+ 2 KB of bit pieces behind them, column path)
  const int o = px.o0 + (int)blockIdx.z;
  PostCtx cx;
  cx.lds = lds;
  cx.tid = threadIdx.x; cx.lane = cx.tid & 63; cx.wave = cx.tid >> 6;
  cx.rb0 = blockIdx.y * (4 * RB); cx.cs0 = blockIdx.x * 8; cx.nrb = nrb; cx.ncs = ncs;
  cx.ucnt0 = (unsigned int)cs.count[0];
  cx.nlines = nlines;
  cx.full = (long long)(cx.rb0 + 4 * RB) * 16 <= nlines && (long long)(cx.cs0 + 8) * 16 <= cs.count[0];
  cx.imode = imode != 0;
  // staging role of this thread: 64 bytes of one A image and 64 bytes of one B strip per k-block.
  // LDS image of an A block: per k-step the 16 lane-chunks are split into their first and second 16 bytes
  // ([16 x 16 B][16 x 16 B]) so that both ds_read_b128 of a fragment load touch 256 contiguous bytes (no bank conflicts)
  const int st_i = cx.tid >> 5, st_j = cx.tid & 31;
  cx.st_off = st_j * 8;
  cx.st_rb = cx.rb0 + st_i < nrb ? cx.rb0 + st_i : nrb - 1;
  cx.st_cs = cx.cs0 + st_i < ncs ? cx.cs0 + st_i : ncs - 1;
  cx.a_lo = st_i < 4 * RB ? st_i * 256 + (st_j >> 3) * 64 + (st_j & 7) * 4 : -1;   // (RB = 1: four images, half the threads stage one)
  cx.b_st = (RB == 2 ? 2048 : 1024) + st_i * 256 + cx.st_off;
  cx.a_rd = (((cx.lane >> 4) << 2) + (cx.lane & 3)) * 2;
  // per-output operands; VA holds [V0 | V1;V0 | V1x] as three image sets, SBf holds [S0 | S0;-xn0 S0] as two fragment sets
  const double* VAo = VA + (size_t)o * sVA;
  const double* SBo = SBf + (size_t)o * sSBf;
  const double sf2 = mc.sf2[o], ystd = mc.Y_std[o];
  double* const vo = var_out + (size_t)o * cs.n_local;
  double* const mo = mean_out + (size_t)o * cs.n_local;
  // (one constraint: the masks themselves; several, r05: constraint o writes byte plane o - 1, AND-ed by k_classify_and)
  const bool fuse = Sfuse != nullptr && o >= 1;
  const bool fmulti = px.fstride != 0;
  cx.var_rd = vo;
  cx.S = fuse ? Sfuse + (size_t)(o - 1) * (size_t)px.fstride : nullptr;
  cx.U = fuse ? Ufuse + (size_t)(o - 1) * (size_t)px.fstride : Ufuse;
  cx.bconf = bconf;
  cx.bb = bconf * bconf;
  cx.cS = cx.cU = cx.cB = 0;
  cx.rmax = -1.0;
  // column path: a tile is 64 rows (one segment of the column words) x 128 columns
  const bool cbits = RB == 1 && ROLE == 1 && px.cb.Sw != nullptr && o == 1, obits = RB == 1 && ROLE == 2 && px.cb.Sw != nullptr && o == 0;
  const size_t ctile = (size_t)blockIdx.y * gridDim.x + blockIdx.x, ntile = (size_t)gridDim.x * gridDim.y;
  // (objective: how many safe candidates the constraint's launch counted in this tile -- field 1 of its partial row)
  const unsigned long long tile_nS = obits ? cpart[(size_t)1 * pcap + ctile] : 0ull;
  cx.role = cbits ? 1 : ((obits && tile_nS != 0ull) ? 2 : 0);
  cx.skip_store = obits && tile_nS == 0ull && px.lean != 0;
  cx.lds_bits = reinterpret_cast<unsigned int*>(lds + 2 * (RB == 2 ? 4096 : 3072));
  cx.umin = 1e300;
  cx.xmin = 1e300;
#pragma unroll
  for (int s2 = 0; s2 < 8; ++s2) cx.bw[s2] = 0u;
  cx.gb_on = (fuse || cbits) && gb != nullptr;
  cx.lband = cx.gb_on ? lcb_band(cx.bb, gb->dm[o >= 1 ? o : 1], gb->dv[o >= 1 ? o : 1]) : LcbBand{0.0, 0.0};
  cx.vminS = 1e300;
  double gmax = 0.0;
  // (Tried: odd outputs running the three short phases first and the variance phase last, so that the two workgroups of a
  // CU do not reach their phase changes together -- no gain on config B, 2.5 % slower on H; one order for all.)
  d4_t acc[RB][8];
  d4_t pre[4];
  const double* const A2 = VAo + (size_t)nrb * KBm * 256;
  const double* const B2 = imode ? SBo : SBo + (size_t)ncs * KBm * 256;
  const double* const A3 = VAo + (size_t)nrb * (KBm + KBm2) * 256;
  const int es = imode ? 4 : 1;
  const int KS1 = imode ? eff[4 * (4 * o + 1)] : KSm, KS2 = imode ? eff[4 * (4 * o + 2)] : KSm, KS3 = imode ? eff[4 * (4 * o + 3)] : KSm;
#ifdef SBO_PHASE_CLOCKS
  unsigned long long clk_ = wall_clock64();
  const unsigned int clk_row_ = ((unsigned int)o * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
#endif
  // The gradient phases (their maxima are the Lipschitz keys, models/SafeOpt.py:68-83) run on the tiles that can hold the grid's
  // maximum: the tile's largest coarse sample + the plan's bound on what lies between the samples reaches the grid's largest
  // sample (k_bl_gradbound / k_bl_gradcoarse above).  Every tile folds its own samples in (true grid values).  NaN: run.
  const double cg0 = ystd * mc.inv_ell[o][0] * mc.X_rstd[0], cg1 = ystd * mc.inv_ell[o][1] * mc.X_rstd[1];
  bool run2 = true, run3 = true;
  double gfold = 0.0;                       // (uniform: scalar registers -- folded in behind the phases)
  if (gtmax) {
    const size_t nt = (size_t)gridDim.x * gridDim.y, tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    const double* slack = gtmax + (size_t)px.q * 2 * nt;
    const double t0 = gtmax[((size_t)o * 2 + 0) * nt + tile], t1 = gtmax[((size_t)o * 2 + 1) * nt + tile];
    const double G0 = __longlong_as_double((long long)gkey[2 * o + 0]), G1 = __longlong_as_double((long long)gkey[2 * o + 1]);
    run2 = !(t0 + slack[2 * o + 0] < G0 * (1.0 - 1e-12));
    run3 = !(t1 + slack[2 * o + 1] < G1 * (1.0 - 1e-12));
    const double f0 = fabs(cg0 * t0), f1 = fabs(cg1 * t1);
    gfold = fmax(f0, f1);
  }
  if (px.nograd) { run2 = run3 = false; gfold = 0.0; }          // (the keys come from a launch of the gradient phases alone: k_bgrad)
  if (ROLE == 2 && px.lean) { run2 = run3 = false; gfold = 0.0; }   // (a lean sweep: nobody reads the objective's key, include/safebo.h)
  // lean sweeps, level 2: the objective's posterior of a tile without a safe candidate is not even evaluated -- u*, M and the
  // arg-max reductions read it on S only (models/SafeOpt.py:47-66); the tile still runs the gradient phases the gate asks for
  // (L_0 is a maximum over the whole grid), and with K1b's operands the mean phase those continue from
  const bool skip_tile = ROLE == 2 && obits && tile_nS == 0ull && px.lean >= 2;
  if (!skip_tile)
    post_phase<0, RB, ROLE>(cx, BtA + (size_t)o * sBtA, P0f + (size_t)o * sP0f, KB0, eff ? eff[4 * o * es] : KS0, vo, sf2, ystd * ystd, 0.0, gmax, acc, xn0, pre,
                            VAo, SBo, KBm);
  SBO_CLK(0);
  if (ROLE == 2 && cx.role == 2) {
    // the thread's own S bits: rows 16 wave + 4 t + (lane >> 4) of the segment, column (cs0 + s2) 16 + (lane & 15)
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
      const unsigned long long w = px.cb.Sw[(size_t)blockIdx.y * cx.ucnt0 + (unsigned int)(cx.cs0 + s2) * 16u + (cx.lane & 15)];
      cx.bw[s2] = (unsigned int)(w >> (16 * cx.wave + (cx.lane >> 4))) & 0x1111u;
    }
  }
  // (the mean phase requests the first operands of whichever phase follows it)
  if (!skip_tile || (run2 && !imode))
    post_phase<1, RB, ROLE>(cx, VAo, SBo, KBm, KS1, mo, mc.mp[o], ystd, mc.Y_mean[o], gmax, acc, xn0, pre, run2 ? A2 : A3, run2 ? B2 : SBo,
                            run2 ? KBm2 : KBm);
  SBO_CLK(1);
  // phase 2 continues on phase 1's sums: only the V1 half (the first KSm k-steps) of the stacked operands is run
  if (run2) post_phase<2, RB, ROLE>(cx, A2, B2, KBm2, KS2, nullptr, cg0, 0.0, 0.0, gmax, acc, xn0, pre, A3, SBo, KBm);
  if (run3) post_phase<3, RB, ROLE>(cx, A3, SBo, KBm, KS3, nullptr, cg1, 0.0, 0.0, gmax, acc, xn0, pre, nullptr, nullptr, 0);
  gmax = fmax(gmax, gfold);
  SBO_CLK(2);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double other = __shfl_xor(gmax, off);
    gmax = other > gmax ? other : gmax;
  }
  // one plain store per WORKGROUP, merged by k_lmax_reduce / the classification's final merge: every workgroup of this launch
  // is resident at once and ends at the same time, so atomics on the q keys would queue up in L2 as the kernel's tail -- and
  // a row per wave made that merge (one workgroup, 16384 rows of 88 bytes on config H) the longest job of the launch it shares
  // (column path: the constraint's rows first, the objective's rows behind them)
  post_partials<4>(cx.lds, cx.lane, cx.wave, gmax, fuse || cbits, cx.cS, cx.cU, obits ? -cx.umin : cx.rmax, cx.cB, cx.vminS,
                   Lpart + ((size_t)o * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x,
                   cpart + (obits ? ntile : (fmulti && o >= 1 ? (size_t)(o - 1) * ntile : (size_t)0)) + ctile, pcap, obits,
                   (cbits || obits) ? px.cb.slots : nullptr, (int)(ctile & (kColSlots - 1)), o, cbits ? (int)blockIdx.y : -1, (int)blockIdx.x, cx.xmin, fmulti);
  if (cbits && cx.tid < 128) {
    // the tile's words: column tid, the four waves' 16-row pieces (written before the barriers of post_partials)
    unsigned long long sw = 0ull, uw = 0ull;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned int pc = cx.lds_bits[w * 128 + cx.tid];
      sw |= (unsigned long long)(pc & 0xffffu) << (16 * w);
      uw |= (unsigned long long)(pc >> 16) << (16 * w);
    }
    const size_t col = (size_t)cx.cs0 * 16u + cx.tid;
    px.cb.Sw[(size_t)blockIdx.y * cx.ucnt0 + col] = sw;
    px.cb.Uw[(size_t)blockIdx.y * cx.ucnt0 + col] = uw;
    if (uw != 0ull) atomicOr(&px.cb.Usum[col], 1ull << blockIdx.y);
  }
  SBO_CLK(3);
#ifdef SBO_PHASE_CLOCKS
  if (threadIdx.x == 0) {
    g_phase_clk[clk_row_ & (kPhaseClkRows - 1)][4] += (unsigned long long)((run2 ? 1 : 0) + (run3 ? 1 : 0));
    g_phase_clk[clk_row_ & (kPhaseClkRows - 1)][5] += 1ull;
  }
#endif
}
#ifdef SBO_PHASE_CLOCKS
extern "C" int sbo_debug_phase_clocks(unsigned long long* out /* [16]: sums over the rows below `split` | from `split` on */, int reset, int split) {
  std::vector<unsigned long long> h((size_t)kPhaseClkRows * 8);
  if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_phase_clk), sizeof(unsigned long long) * h.size()) != hipSuccess) return 1;
  for (int k = 0; k < 16; ++k) out[k] = 0;
  for (size_t r = 0; r < (size_t)kPhaseClkRows; ++r)
    for (int k = 0; k < 8; ++k) out[(r < (size_t)split ? 0 : 8) + k] += h[r * 8 + k];
  if (reset) {
    std::fill(h.begin(), h.end(), 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_clk), h.data(), sizeof(unsigned long long) * h.size()) != hipSuccess) return 1;
  }
  return 0;
}
#endif


// K1i's deferred gradient launch (r05): the two gradient series of every output on the tiles the gate names, nothing else -- on
// stream3 behind the gate's kernels, beside the posterior launches, which then carry no gradient phases (PostExtra::nograd).  A
// resident-sized grid walks the (output, tile) table: a tile the gate excludes costs two loads, and its row of the Lipschitz partials
// holds its largest coarse sample as before.  Rows [q][tiles] as k_bpost writes them; column path: the workgroup's maximum per output
// also joins the slot block.
__global__ __launch_bounds__(256, 3) void k_bgrad(const ModelConst mc, const CandSpec cs, const double* __restrict__ VA, size_t sVA,
                                                  const double* __restrict__ P0f, int KB, int nrb, int ncs, long long nlines, int gx, int gy,
                                                  const int* __restrict__ eff, const double* __restrict__ gtmax,
                                                  const unsigned long long* __restrict__ gkey, const double* __restrict__ xn0, int q,
                                                  double* __restrict__ Lpart, unsigned long long* __restrict__ slots,
                                                  int o_first /* 1: a lean sweep -- nobody reads the objective's key, its rows stay zero */) {
  extern __shared__ double lds[];
  PostCtx cx;
  cx.lds = lds;
  cx.tid = threadIdx.x; cx.lane = cx.tid & 63; cx.wave = cx.tid >> 6;
  cx.nrb = nrb; cx.ncs = ncs;
  cx.ucnt0 = (unsigned int)cs.count[0];
  cx.nlines = nlines;
  cx.imode = true;
  const int st_i = cx.tid >> 5, st_j = cx.tid & 31;
  cx.st_off = st_j * 8;
  cx.a_lo = st_i < 4 ? st_i * 256 + (st_j >> 3) * 64 + (st_j & 7) * 4 : -1;
  cx.b_st = 1024 + st_i * 256 + cx.st_off;
  cx.a_rd = (((cx.lane >> 4) << 2) + (cx.lane & 3)) * 2;
  cx.var_rd = nullptr; cx.S = nullptr; cx.U = nullptr;
  cx.bconf = 0.0; cx.bb = 0.0; cx.cS = cx.cU = cx.cB = 0; cx.rmax = -1.0;
  cx.lband = LcbBand{0.0, 0.0}; cx.gb_on = false; cx.vminS = 1e300;
  cx.role = 0; cx.lds_bits = nullptr; cx.umin = 1e300; cx.xmin = 1e300; cx.skip_store = false;
#pragma unroll
  for (int s2 = 0; s2 < 8; ++s2) cx.bw[s2] = 0u;
  const size_t nt = (size_t)gx * gy;
  const double* slack = gtmax ? gtmax + (size_t)q * 2 * nt : nullptr;       // (no gate: every tile runs both phases)
  d4_t acc[1][8];
  d4_t pre[4];
  for (size_t tile = blockIdx.x; o_first > 0 && tile < nt; tile += gridDim.x)
    if (cx.tid == 0) Lpart[tile] = 0.0;
  for (int o = o_first; o < q; ++o) {
    const double ystd = mc.Y_std[o];
    const double cg0 = ystd * mc.inv_ell[o][0] * mc.X_rstd[0], cg1 = ystd * mc.inv_ell[o][1] * mc.X_rstd[1];
    const double G0 = gtmax ? __longlong_as_double((long long)gkey[2 * o + 0]) : 0.0, G1 = gtmax ? __longlong_as_double((long long)gkey[2 * o + 1]) : 0.0;
    const double s0 = gtmax ? slack[2 * o + 0] : 0.0, s1 = gtmax ? slack[2 * o + 1] : 0.0;
    const int KS2 = eff[4 * (4 * o + 2)], KS3 = eff[4 * (4 * o + 3)];
    const double* VAo = VA + (size_t)o * sVA;
    const double* A2 = VAo + (size_t)nrb * KB * 256;
    const double* A3 = VAo + (size_t)nrb * (2 * KB) * 256;
    double wg_max = 0.0;
    for (size_t tile = blockIdx.x; tile < nt; tile += gridDim.x) {
      const double t0 = gtmax ? gtmax[((size_t)o * 2 + 0) * nt + tile] : 0.0, t1 = gtmax ? gtmax[((size_t)o * 2 + 1) * nt + tile] : 0.0;
      const bool run2 = !(t0 + s0 < G0 * (1.0 - 1e-12)), run3 = !(t1 + s1 < G1 * (1.0 - 1e-12));          // (NaN: run)
      double g = fmax(fabs(cg0 * t0), fabs(cg1 * t1));
      if (run2 || run3) {
        const int bx = (int)(tile % (size_t)gx), by = (int)(tile / (size_t)gx);
        cx.rb0 = by * 4; cx.cs0 = bx * 8;
        cx.full = (long long)(cx.rb0 + 4) * 16 <= nlines && (long long)(cx.cs0 + 8) * 16 <= cs.count[0];
        cx.st_rb = cx.rb0 + st_i < nrb ? cx.rb0 + st_i : nrb - 1;
        cx.st_cs = cx.cs0 + st_i < ncs ? cx.cs0 + st_i : ncs - 1;
        double gmax = 0.0;
        if (run2) post_phase<2, 1, 0>(cx, A2, P0f, KB, KS2, nullptr, cg0, 0.0, 0.0, gmax, acc, xn0, pre, nullptr, nullptr, 0);
        if (run3) post_phase<3, 1, 0>(cx, A3, P0f, KB, KS3, nullptr, cg1, 0.0, 0.0, gmax, acc, xn0, pre, nullptr, nullptr, 0);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          const double other = __shfl_xor(gmax, off);
          gmax = other > gmax ? other : gmax;
        }
        __syncthreads();
        if (cx.lane == 0) lds[cx.wave] = gmax;
        __syncthreads();
        gmax = fmax(fmax(lds[0], lds[1]), fmax(lds[2], lds[3]));
        __syncthreads();
        g = fmax(g, gmax);
      }
      if (cx.tid == 0) Lpart[(size_t)o * nt + tile] = g;
      wg_max = fmax(wg_max, g);
    }
    if (slots && cx.tid == 0)
      atomicMax(&slots[(size_t)(o == 0 ? kSlotL0 : kSlotL1) * kColSlots + (blockIdx.x & (kColSlots - 1))], (unsigned long long)__double_as_longlong(wg_max));
  }
}

// Lipschitz keys of a K1b launch: Lmax[o] = max of the per-wave partials (values >= 0, so the bit pattern orders them)
__global__ __launch_bounds__(256) void k_lmax_reduce(const double* __restrict__ Lpart, int per_out, unsigned long long* __restrict__ Lmax) {
  __shared__ double sh[4];
  lmax_reduce_body((int)blockIdx.x, sh, Lpart, per_out, Lmax);
}

// ---- Chebyshev core of the variance phase (r03) ---------------------------------------------------------------------------
// The pair form above contracts  quad = sum_{k0, k1} T4[k0][k1] P0_k0(x0) P1_k1(x1)  over K = r (r + 1) / 2 ~ 276 pair products
// per axis.  But the pair products are not independent functions: S_p is a Chebyshev series of degree < rc on the axis, so
// every P_k = c S_p S_p' is a polynomial of degree <= 2 rc - 2, and quad is a polynomial of that degree in each variable --
//     quad(x0, x1) = sum_{a < D0, b < D1} Chat[a][b] T_a(xi0) T_b(xi1),      Chat = PC0^T T4 PC1,
// with PC[k][m] the Chebyshev coefficients of P_k (product formula T_a T_b = (T_{a+b} + T_|a-b|) / 2: exact, no sampling).
// The same two GEMMs then run with the Chebyshev polynomials of the grid positions as operand tables (no dependence on the
// data at all) and the inner dimensions D <= 2 rc - 1 = 95 instead of 276 -- and Chat decays: the variance surface of the
// BASELINE hyper-parameters needs degree ~50 for 1e-15, so the kernels stop at the degree behind which every coefficient is
// below 4e-15 of the largest (k_cheb_trunc writes the k-step / k-block counts per output; stage 1 and k_bpost read them from
// memory: no host round trip).  Where the coefficients do not decay that far -- ill-conditioned models, whose Chat carries the
// rounding of the reference formula itself -- nothing is dropped.
// PC[job][k][m] = Chebyshev coefficient m of P_k, job = 2 o + axis; rows k >= K and columns m >= 2 rc - 1 are zero.
// One workgroup per pair: the two coefficient rows go to LDS once, a thread per degree m sums its two convolutions from
// there (four independent partial sums each: read straight from memory, a thread waited out ~150 dependent loads, 21 us).
__global__ __launch_bounds__(256) void k_cheb_pairs(const BlDims dm, const double* __restrict__ Vsall, const double* __restrict__ sigall,
                                                    size_t sPC0, size_t sPC1, double* __restrict__ PC0all, double* __restrict__ PC1all) {
  __shared__ double va[kBlMaxRc], vb[kBlMaxRc];
  const int job = blockIdx.y, o = job >> 1, axis = job & 1, k = blockIdx.x, tid = threadIdx.x;
  const int r = axis ? dm.r1[o] : dm.r0[o], rc = axis ? dm.rc1[o] : dm.rc0[o];
  const int K = r * (r + 1) / 2, Km = axis ? dm.K1m : dm.K0m, Dm = axis ? dm.D1m : dm.D0m;
  if (k >= (Km + 15) / 16 * 16) return;
  // Output in the operand layouts of the two MFMA GEMMs that form Chat^T = (PC1^T T4^T) PC0:
  //   axis 1: A images of PC1^T  [D1m / 16][KBp][256]   (row = degree m, inner index = pair k)
  //   axis 0: B fragments of PC0 [D0m / 16][KBp * 4][64] (inner index = pair k, column = degree m)
  double* PC = axis ? PC1all + (size_t)o * sPC1 : PC0all + (size_t)o * sPC0;
  const int KBp = (Km + 15) / 16;
  auto put = [&](int m, double v) {
    const int kb = k >> 4, j = k & 15, kk = j >> 2, slot = j & 3;          // MM<double>::jslot(kk, slot) = 4 kk + slot
    if (axis) PC[(((size_t)(m >> 4) * KBp + kb) << 8) + (size_t)MM<double>::pack_pos(m & 15, slot, kk)] = v;
    else PC[(((size_t)(m >> 4) * (KBp * 4) + (size_t)(kb * 4 + kk)) << 6) + (size_t)(slot * 16 + (m & 15))] = v;
  };
  if (k >= K) {
    for (int m = tid; m < Dm; m += blockDim.x) put(m, 0.0);
    return;
  }
  const double* Vs = Vsall + (size_t)job * kBlMaxR * kBlMaxRc;
  const double* sig = sigall + (size_t)job * kBlMaxR;
  int p, pp;
  pair_of(k, r, p, pp);
  for (int a = tid; a < rc; a += blockDim.x) {
    va[a] = Vs[(size_t)p * rc + a];
    vb[a] = Vs[(size_t)pp * rc + a];
  }
  __syncthreads();
  const double scale = (p == pp ? 0.5 : 1.0) * (sig[p] * sig[pp]);       // (1 | 2) * 1/2
  for (int m = tid; m < Dm; m += blockDim.x) {
    double v = 0.0;
    if (m <= 2 * rc - 2) {
      // T_a T_b = (T_{a+b} + T_|a-b|) / 2
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      const int a0 = m - rc + 1 > 0 ? m - rc + 1 : 0, a1 = m < rc - 1 ? m : rc - 1;      // a + b = m, both below rc
      int a = a0;
      for (; a + 3 <= a1; a += 4) {
        s0 += va[a] * vb[m - a];
        s1 += va[a + 1] * vb[m - a - 1];
        s2 += va[a + 2] * vb[m - a - 2];
        s3 += va[a + 3] * vb[m - a - 3];
      }
      for (; a <= a1; ++a) s0 += va[a] * vb[m - a];
      if (m == 0) {
        for (a = 0; a + 3 < rc; a += 4) {
          s0 += va[a] * vb[a];
          s1 += va[a + 1] * vb[a + 1];
          s2 += va[a + 2] * vb[a + 2];
          s3 += va[a + 3] * vb[a + 3];
        }
        for (; a < rc; ++a) s0 += va[a] * vb[a];
      } else {
        const int lim = rc - m;                                        // |a - b| = m
        for (a = 0; a + 1 < lim; a += 2) {
          s0 += va[a + m] * vb[a];
          s1 += va[a] * vb[a + m];
          s2 += va[a + 1 + m] * vb[a + 1];
          s3 += va[a + 1] * vb[a + 1 + m];
        }
        for (; a < lim; ++a) { s0 += va[a + m] * vb[a]; s1 += va[a] * vb[a + m]; }
      }
      v = scale * ((s0 + s1) + (s2 + s3));
    }
    put(m, v);
  }
}
// T4 as a plain matrix [K0m][K1m] (the entries k_bl_t4f gathers, same formula)
__global__ __launch_bounds__(256) void k_cheb_t4(const BlDims dm, const double* __restrict__ Gall, long long ldg, double* __restrict__ T4all,
                                                 int sym) {
  const int o = blockIdx.y, r0 = dm.r0[o], r1 = dm.r1[o];
  const int K0 = r0 * (r0 + 1) / 2, K1 = r1 * (r1 + 1) / 2;
  const double* G = Gall + (size_t)o * ldg * ldg;
  double* T4 = T4all + (size_t)o * dm.K0m * dm.K1m;
  const double scale = dm.sf2[o] * dm.sf2[o];
  const long long total = (long long)dm.K0m * dm.K1m;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k0 = (int)(i / dm.K1m), k1 = (int)(i % dm.K1m);
    double v = 0.0;
    if (k0 < K0 && k1 < K1) {
      int p, pp, s1, ss;
      pair_of(k0, r0, p, pp);
      pair_of(k1, r1, s1, ss);
      const size_t a1 = (size_t)(p * r1 + s1), b1 = (size_t)(pp * r1 + ss), a2 = (size_t)(pp * r1 + s1), b2 = (size_t)(p * r1 + ss);
      v = sym ? scale * 0.25 * ((G[a1 * ldg + b1] + G[b1 * ldg + a1]) + (G[a2 * ldg + b2] + G[b2 * ldg + a2]))
              : scale * 0.5 * (G[a1 * ldg + b1] + G[a2 * ldg + b2]);
    }
    T4[i] = v;
  }
}
// C = A^T B for small matrices, A [K][M], B [K][N] row-major (k-major); 32 x 32 outputs per workgroup, 64-deep k tiles (the
// 16-deep form waited out 18 load -> barrier round trips for K = 276: 40 us), plain fp64 sums in ascending k.  TRANS: C stored
// transposed ([N][M]).  blockIdx.z = output.  rowmax / colmax (nullptr: none): max |C| of every row and column as bit
// patterns (non-negative doubles order like their bits), for k_cheb_trunc.
template <bool TRANS>
__global__ __launch_bounds__(256) void k_small_tn(const double* __restrict__ Aall, size_t sA, const double* __restrict__ Ball, size_t sB, int K,
                                                  int M, int N, double* __restrict__ Call, size_t sC, unsigned long long* __restrict__ rowmax,
                                                  unsigned long long* __restrict__ colmax) {
  constexpr int KT = 64;
  __shared__ double As[KT][33], Bs[KT][33];
  const int o = blockIdx.z, tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const double* A = Aall + (size_t)o * sA;
  const double* B = Ball + (size_t)o * sB;
  double* C = Call + (size_t)o * sC;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < K; k0 += KT) {
    for (int e = tid; e < KT * 32; e += 256) {
      const int kk = e >> 5, c = e & 31;
      As[kk][c] = (k0 + kk < K && m0 + c < M) ? A[(size_t)(k0 + kk) * M + m0 + c] : 0.0;
      Bs[kk][c] = (k0 + kk < K && n0 + c < N) ? B[(size_t)(k0 + kk) * N + n0 + c] : 0.0;
    }
    __syncthreads();
#pragma unroll 16
    for (int kk = 0; kk < KT; ++kk) {
      const double b = Bs[kk][tx];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] += As[kk][ty + 8 * u] * b;
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int m = m0 + ty + 8 * u, n = n0 + tx;
    if (m < M && n < N) {
      C[TRANS ? (size_t)n * M + m : (size_t)m * N + n] = acc[u];
      if (rowmax) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(fabs(acc[u]));
        atomicMax(&rowmax[(size_t)o * M + m], key);
        atomicMax(&colmax[(size_t)o * N + n], key);
      }
    }
  }
}
// degrees the kernels run to, per output: behind them every |Chat| is below thr of the largest.  ChatT [D1m][D0m] row-major
// (the GEMMs deliver the transpose).  eff[4 o + 0] = k-steps of the variance phase (axis 0, four degrees each), [1] = its
// 16-blocks (strips of stage 1), [2] = 16-blocks of axis 1 (inner dimension of stage 1), [3] = 0.  One workgroup of 1024.
// tail[o] (behind the q x 4 counts): the sum of |Chat| over the entries the kernels do NOT run -- a bound on what the truncation
// moves the quadratic form by (|T_a T_b| <= 1), part of K1b's guard band (guard.hip).
__global__ __launch_bounds__(1024) void k_cheb_trunc(const BlDims dm, const double* __restrict__ ChatT_all, double thr, int* __restrict__ eff) {
  __shared__ double red[16];
  __shared__ int redi[16][2];
  const int o = blockIdx.x, tid = threadIdx.x, D0 = dm.D0m, D1 = dm.D1m;
  const double* Ch = ChatT_all + (size_t)o * D0 * D1;
  double mx = 0.0;
  for (int i = tid; i < D0 * D1; i += 1024) { const double v = fabs(Ch[i]); mx = v > mx ? v : mx; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const double y = __shfl_xor(mx, off); mx = y > mx ? y : mx; }
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = 0.0;
#pragma unroll
  for (int w = 0; w < 16; ++w) mx = red[w] > mx ? red[w] : mx;
  const double cut = thr * mx;
  int a_hi = 0, b_hi = 0;                          // 1 + largest index holding an entry above the cut
  for (int i = tid; i < D0 * D1; i += 1024) {
    if (fabs(Ch[i]) > cut) {
      const int b = i / D0, a = i % D0;            // ChatT[b][a]
      a_hi = a + 1 > a_hi ? a + 1 : a_hi;
      b_hi = b + 1 > b_hi ? b + 1 : b_hi;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int ya = __shfl_xor(a_hi, off), yb = __shfl_xor(b_hi, off);
    a_hi = ya > a_hi ? ya : a_hi;
    b_hi = yb > b_hi ? yb : b_hi;
  }
  if ((tid & 63) == 0) { redi[tid >> 6][0] = a_hi; redi[tid >> 6][1] = b_hi; }
  __syncthreads();
  __shared__ int run_ab[2];
  if (tid == 0) {
    for (int w = 1; w < 16; ++w) { a_hi = redi[w][0] > a_hi ? redi[w][0] : a_hi; b_hi = redi[w][1] > b_hi ? redi[w][1] : b_hi; }
    a_hi = a_hi < 1 ? 1 : a_hi;
    b_hi = b_hi < 1 ? 1 : b_hi;
    eff[4 * o + 0] = (a_hi + 3) / 4;
    eff[4 * o + 1] = (a_hi + 15) / 16;
    eff[4 * o + 2] = (b_hi + 15) / 16;
    eff[4 * o + 3] = 0;
    run_ab[0] = (a_hi + 3) / 4 * 4;              // degrees the two GEMMs run: axis 0 in k-steps of four, axis 1 in blocks of 16
    run_ab[1] = (b_hi + 15) / 16 * 16;
  }
  __syncthreads();
  double ts = 0.0, fs = 0.0;
  for (int i = tid; i < D0 * D1; i += 1024) {
    const int b = i / D0, a = i % D0;
    if (a >= run_ab[0] || b >= run_ab[1]) ts += fabs(Ch[i]);
    if (a >= D0 - 4 || b >= D1 - 4) fs += fabs(Ch[i]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { ts += __shfl_xor(ts, off); fs += __shfl_xor(fs, off); }
  __syncthreads();
  __shared__ double redf[16];
  if ((tid & 63) == 0) { red[tid >> 6] = ts; redf[tid >> 6] = fs; }
  __syncthreads();
  if (tid == 0) {
    double s = 0.0, f = 0.0;
    for (int w = 0; w < 16; ++w) { s += red[w]; f += redf[w]; }
    reinterpret_cast<double*>(eff + 4 * dm.q)[o] = s;
    // frame[o] (behind the tails): the sum of |Chat| over the last four degrees of either axis -- for the series of an INTERPOLANT
    // (K1i: the coefficients beyond the node count are not zero, they alias into these) the figure its band extrapolates the
    // aliasing error from; K1b's core is the exact product of degree-(rc - 1) polynomials and holds zeros there
    reinterpret_cast<double*>(eff + 4 * dm.q)[gridDim.x + o] = f;
  }
}
// ChatT [D1m][D0m] -> the B fragments of stage 1 (T4f layout: k = axis-1 degree, column = axis-0 degree)
__global__ __launch_bounds__(256) void k_cheb_t4f(const BlDims dm, const double* __restrict__ Chat_all, size_t sT4f, double* __restrict__ T4fall) {
  const int o = blockIdx.y, KB0 = dm.KB0, KB1 = dm.KB1;
  const double* Ch = Chat_all + (size_t)o * dm.D0m * dm.D1m;
  double* T4f = T4fall + (size_t)o * sT4f;
  const long long total = (long long)KB0 * KB1 * 4 * 64;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    const long long fr = i >> 6;
    const int ks = (int)(fr % (KB1 * 4)), cs = (int)(fr / (KB1 * 4));
    const int k1 = (ks >> 2) * 16 + MM<double>::jslot(ks & 3, l >> 4);
    const int k0 = cs * 16 + (l & 15);
    T4f[i] = Ch[(size_t)k1 * dm.D0m + k0];
  }
}
// Chebyshev polynomials of the grid positions in the operand layouts of the two GEMMs (k_bl_pairs' layouts): FRAG 1: B
// fragments [ncs0][KB0 * 4][64] of axis 0; FRAG 0: A images [nrb][KB1][256] of axis 1.  A thread per position walks the
// three-term recurrence once (the argument exactly as k_bl_stab forms it); the same table serves every output.
template <int FRAG>
__global__ __launch_bounds__(256) void k_cheb_tab(const BlDims dm, const double* __restrict__ xn, double* __restrict__ out) {
  const int KB = FRAG ? dm.KB0 : dm.KB1, nblk = FRAG ? dm.ncs0 : dm.nrb;
  const long long count = FRAG ? dm.cnt0 : dm.nlines;
  const double a = dm.a[FRAG ? 0 : 1], b = dm.b[FRAG ? 0 : 1];
  for (long long x = (long long)blockIdx.x * blockDim.x + threadIdx.x; x < (long long)nblk * 16; x += (long long)gridDim.x * blockDim.x) {
    double xi = 0.0;
    if (x < count) {
      xi = (2.0 * xn[x] - (a + b)) / (b - a);
      xi = xi < 1.0 ? xi : 1.0;
      xi = xi > -1.0 ? xi : -1.0;
    }
    double t0 = 1.0, t1 = xi;
    for (int k = 0; k < KB * 16; ++k) {
      double v = k == 0 ? t0 : t1;
      if (k >= 2) {
        v = 2.0 * xi * t1 - t0;
        t0 = t1;
        t1 = v;
      }
      if (x >= count) v = 0.0;
      size_t idx;
      if (FRAG) {
        // fragment (strip = x / 16, k-step ks = k / 4 ... ): element of k-block kb = k / 16 with jslot(kk, slot) = k % 16
        const int kb = k >> 4, j = k & 15, kk = j >> 2, slot = j & 3;      // MM<double>::jslot(kk, slot) = 4 kk + slot
        idx = (((size_t)(x >> 4) * (KB * 4) + (size_t)(kb * 4 + kk)) << 6) + (size_t)(slot * 16 + (x & 15));
      } else {
        const int kb = k >> 4, j = k & 15, kk = j >> 2, slot = j & 3;
        idx = (((size_t)(x >> 4) * KB + kb) << 8) + (size_t)MM<double>::pack_pos((int)(x & 15), slot, kk);
      }
      out[idx] = v;
    }
  }
}

// K1b's own representation at the probe points of its guard band (guard.hip), from the plan's tables -- the sums the two GEMMs
// of a posterior launch form, term by term in another order (the difference is their rounding, which the band's floor covers):
//   quad = sum_{a < A, b < B} ChatT[b][a] T_a(xi0) T_b(xi1)   over the degrees the kernels run (eff), T by the tables' recurrence
//   mean = mp + sum_p Vb0[p][line] S0[p][x0]
// A wave per (probe, output): lane l takes the rows b = l, l + 64, .. of ChatT (its T_b by the recurrence up to b, then the row's
// sum over a with T_a walked along), the lanes' shares meet in a wave sum.  (A thread per probe walked 2304 dependent steps: 105 us.)
__global__ __launch_bounds__(256) void k_gb_probe_k1b(const ModelConst mc, const CandSpec cs, const BlDims dm, const double* __restrict__ ChatT_all,
                                                      const int* __restrict__ eff, const double* __restrict__ xn0, const double* __restrict__ xn1,
                                                      const double* __restrict__ S0all, const double* __restrict__ Vball,
                                                      double* __restrict__ pm, double* __restrict__ pv) {
  const int o = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, D0 = dm.D0m;
  const int p = blockIdx.x * 4 + wave;
  if (p >= kGbProbes) return;                      // (no barriers in this kernel)
  const int A = eff[4 * o] * 4, B = eff[4 * o + 2] * 16;
  const double* Ch = ChatT_all + (size_t)o * D0 * dm.D1m;
  long long x0, x1;
  gb_probe_xy(cs, dm.nlines, p, x0, x1);
  auto xi_of = [&](double xn, int axis) {
    double xi = (2.0 * xn - (dm.a[axis] + dm.b[axis])) / (dm.b[axis] - dm.a[axis]);
    xi = xi < 1.0 ? xi : 1.0;
    return xi > -1.0 ? xi : -1.0;
  };
  const double xi0 = xi_of(xn0[x0], 0), xi1 = xi_of(xn1[x1], 1);
  double quad = 0.0;
  for (int b = lane; b < B; b += 64) {
    double tb0 = 1.0, tb1 = xi1, tb = b == 0 ? 1.0 : xi1;
    for (int k = 2; k <= b; ++k) { tb = 2.0 * xi1 * tb1 - tb0; tb0 = tb1; tb1 = tb; }
    const double* rowp = Ch + (size_t)b * D0;
    double row = 0.0, ta0 = 1.0, ta1 = xi0;
    for (int a = 0; a < A; ++a) {
      double ta = a == 0 ? 1.0 : xi0;
      if (a >= 2) { ta = 2.0 * xi0 * ta1 - ta0; ta0 = ta1; ta1 = ta; }
      row = fma(rowp[a], ta, row);
    }
    quad = fma(row, tb, quad);
  }
  quad = wave_sum(quad);
  const int r0 = dm.r0[o];
  const double* S0 = S0all + (size_t)o * dm.r0u * dm.cnt0;
  const double* Vb = Vball + (size_t)o * 3 * dm.r0u * dm.nlines;
  double ms = 0.0;
  for (int pp = lane; pp < r0; pp += 64) ms = fma(Vb[(size_t)pp * dm.nlines + x1], S0[(size_t)pp * dm.cnt0 + x0], ms);
  ms = wave_sum(ms);
  if (lane == 0) {
    double var = mc.sf2[o] - quad;
    var = var > 0.0 ? var : 0.0;
    const double ys = mc.Y_std[o];
    pm[(size_t)o * kGbProbes + p] = (mc.mp[o] + ms) * ys + mc.Y_mean[o];
    pv[(size_t)o * kGbProbes + p] = var * (ys * ys);
  }
}

// ---- Lipschitz keys without the gradient phases on every tile (r04) ------------------------------------------------------------
// L_i = max over the grid of ||grad MEAN_i||_inf (models/SafeOpt.py:68-83) took two of k_bpost's four GEMM phases on EVERY tile
// -- ~90 of its 257 us on config H -- for two scalars per output.  The gradient sums are polynomials of the axes' Chebyshev
// variables (degree <= rc per axis, the degree of the bases), so:
//   (1) k_bl_gradw / k_bl_gradrow / k_bl_gradslack, per plan: the Chebyshev coefficients C[a][b] of g_0 = f_1 - xn0 f_0 and g_1 = f_2 - xn1 f_0 (f_b the
//       three bilinear forms of the mean phases) and from them sum |C| a^2, sum |C| b^2 >= max |dg / dxi| (|T_a'| <= a^2) -> a
//       bound `slack` on how far |g| can move over half a sampling cell;
//   (2) k_bl_gradcoarse, per plan: g at the centres of kGradStep x kGradStep cells of the grid (direct sums over the rank,
//       1 / 64 of the candidates) -> the largest sample of every k_bpost tile and of the whole grid;
//   (3) k_bpost runs a gradient phase only on the tiles whose largest sample + slack reaches the grid's largest sample: no
//       other tile can hold the maximum.  The key is still the maximum of the FINE-grid values of phases 2 / 3 (on those tiles),
//       i.e. the same number as before, to the last bit.  Config H: ~1-7 % of the tiles qualify.
constexpr int kGradStep = 8;
// coefficient bounds in three small launches (one workgroup per output took 0.21 ms of the plan build):
//   k_bl_gradw      W_b[p][c1] = sum_s Mb_b[p][s] sig1_s Vs1[s][c1]                       grid (3 r0u, q), a thread per c1
//   k_bl_gradrow    row a of the coefficient matrices C_b[a][c1] = sum_p sig0_p Vs0[p][a] W_b[p][c1] (C_0 also for a - 1, a + 1),
//                   the products with xn = mid + half xi by  xi T_0 = T_1, xi T_k = (T_{k+1} + T_{k-1}) / 2,  and the row's part
//                   of sum |C| a^2, sum |C| c1^2 for both gradient components                 grid (rc0m + 1, q)
//   k_bl_gradslack  the rows' parts summed in a fixed order -> slack[o][component]            grid (q)
__global__ __launch_bounds__(128) void k_bl_gradw(const BlDims dm, const double* __restrict__ Vsall, const double* __restrict__ sigall,
                                                  const double* __restrict__ Mball, double* __restrict__ Wscr /* [q][3][kBlMaxR][kBlMaxRc] */) {
  const int o = blockIdx.y, b = blockIdx.x / dm.r0u, p = blockIdx.x % dm.r0u, c1 = threadIdx.x;
  const int r0 = dm.r0[o], r1 = dm.r1[o], rc1 = dm.rc1[o];
  if (p >= r0 || c1 >= rc1) return;
  const double* Vs1 = Vsall + (size_t)(2 * o + 1) * kBlMaxR * kBlMaxRc;
  const double* sig1 = sigall + (size_t)(2 * o + 1) * kBlMaxR;
  const double* Mb = Mball + (size_t)o * 3 * dm.r0u * dm.r1u + ((size_t)b * r0 + p) * r1;
  double s = 0.0;
  for (int s_ = 0; s_ < r1; ++s_) s += Mb[s_] * (sig1[s_] * Vs1[(size_t)s_ * rc1 + c1]);
  Wscr[(((size_t)o * 3 + b) * kBlMaxR + p) * kBlMaxRc + c1] = s;
}
__global__ __launch_bounds__(256) void k_bl_gradrow(const BlDims dm, const double* __restrict__ Vsall, const double* __restrict__ sigall,
                                                    const double* __restrict__ Wscr, double* __restrict__ part /* [q][gridDim.x][4] */) {
  __shared__ double c0row[kBlMaxRc + 3];                 // C_0[a][-1 .. rc1 + 1] (zero outside the degrees)
  __shared__ double red[4][4];
  const int o = blockIdx.y, a = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c1 = tid;
  const int r0 = dm.r0[o], rc0 = dm.rc0[o], rc1 = dm.rc1[o];
  double b00 = 0.0, b01 = 0.0, b10 = 0.0, b11 = 0.0;       // [component][axis]: |coefficient| degree^2
  const bool row = a <= rc0;                               // (uniform: the products with xi reach one degree further)
  if (row) {
    const double* Vs0 = Vsall + (size_t)(2 * o) * kBlMaxR * kBlMaxRc;
    const double* sig0 = sigall + (size_t)(2 * o) * kBlMaxR;
    const double* W = Wscr + (size_t)o * 3 * kBlMaxR * kBlMaxRc;
    // C_0 at rows a - 1, a, a + 1, C_1 and C_2 at row a, column c1 of this thread
    double c0m = 0.0, c0 = 0.0, c0p = 0.0, c1v = 0.0, c2v = 0.0;
    if (c1 < rc1)
      for (int p = 0; p < r0; ++p) {
        const double w0 = W[((size_t)0 * kBlMaxR + p) * kBlMaxRc + c1], w1 = W[((size_t)1 * kBlMaxR + p) * kBlMaxRc + c1],
                     w2 = W[((size_t)2 * kBlMaxR + p) * kBlMaxRc + c1];
        const double* v = Vs0 + (size_t)p * rc0;
        const double sg = sig0[p];
        const double am = a >= 1 ? sg * v[a - 1] : 0.0, aa = a < rc0 ? sg * v[a] : 0.0, ap = a + 1 < rc0 ? sg * v[a + 1] : 0.0;
        c0m += am * w0;
        c0 += aa * w0;
        c0p += ap * w0;
        c1v += aa * w1;
        c2v += aa * w2;
      }
    if (c1 <= rc1 + 2) c0row[c1] = 0.0;
    __syncthreads();
    if (c1 < rc1) c0row[c1 + 1] = c0;
    __syncthreads();
    if (c1 <= rc1) {
      const double mid0 = 0.5 * (dm.a[0] + dm.b[0]), half0 = 0.5 * (dm.b[0] - dm.a[0]);
      const double mid1 = 0.5 * (dm.a[1] + dm.b[1]), half1 = 0.5 * (dm.b[1] - dm.a[1]);
      // (xi F)_a along the first index: D_0 = C_1 / 2, D_1 = C_0 + C_2 / 2, D_a = (C_{a-1} + C_{a+1}) / 2
      const double sh0 = a == 0 ? 0.5 * c0p : (a == 1 ? c0m + 0.5 * c0p : 0.5 * (c0m + c0p));
      const double lo = c0row[c1], hi = c0row[c1 + 2];     // C_0[a][c1 - 1], C_0[a][c1 + 1]
      const double sh1 = c1 == 0 ? 0.5 * hi : (c1 == 1 ? lo + 0.5 * hi : 0.5 * (lo + hi));
      const double g0c = c1v - mid0 * c0 - half0 * sh0;    // coefficient (a, c1) of g_0 = f_1 - xn0 f_0
      const double g1c = c2v - mid1 * c0 - half1 * sh1;    //                     of g_1 = f_2 - xn1 f_0
      const double a2 = (double)a * a, c2 = (double)c1 * c1;
      b00 = fabs(g0c) * a2;
      b01 = fabs(g0c) * c2;
      b10 = fabs(g1c) * a2;
      b11 = fabs(g1c) * c2;
    }
  }
  b00 = wave_sum(b00); b01 = wave_sum(b01); b10 = wave_sum(b10); b11 = wave_sum(b11);
  if (lane == 0) { red[wave][0] = b00; red[wave][1] = b01; red[wave][2] = b10; red[wave][3] = b11; }
  __syncthreads();
  if (tid < 4) part[((size_t)o * gridDim.x + a) * 4 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}
__global__ __launch_bounds__(64) void k_bl_gradslack(const double* __restrict__ part, int rows, double dxi0, double dxi1 /* half a sampling cell
                                                     in xi units */, double* __restrict__ slack /* [q][2] */) {
  const int o = blockIdx.x, k = threadIdx.x;
  if (k >= 4) return;
  double s = 0.0;
  for (int a = 0; a < rows; ++a) s += part[((size_t)o * rows + a) * 4 + k];
  // |g| moves by at most max |dg/dxi0| dxi0 + max |dg/dxi1| dxi1 between a candidate and the sample of its cell
  const double t = s * ((k & 1) ? dxi1 : dxi0);
  const double other = __shfl_xor(t, 1);
  if ((k & 1) == 0) slack[2 * o + (k >> 1)] = (t + other) * (1.0 + 1e-9);
}
// the gradient sums at the cell centres of one k_bpost tile (TL lines x 128 positions): a workgroup per (tile, output); the
// tile's columns of S0 and lines of Vb go through LDS (read straight from memory, strided by the cell: 0.11 ms on config H).
// tmax[(o 2 + comp) ntiles + tile] = the tile's largest |g_comp| sample, gkey[o 2 + comp] = the grid's (bit pattern: values >= 0)
constexpr int kGradTL = 64, kGradNpx = 128 / kGradStep, kGradNpl = kGradTL / kGradStep;
__global__ __launch_bounds__(128) void k_bl_gradcoarse(const BlDims dm, const double* __restrict__ S0all, const double* __restrict__ Vball,
                                                       const double* __restrict__ xn0, const double* __restrict__ xn1, int ntx,
                                                       double* __restrict__ tmax, unsigned long long* __restrict__ gkey) {
  __shared__ double sS[kBlMaxR][kGradNpx];
  __shared__ double sV[3][kBlMaxR][kGradNpl];
  __shared__ double red[2][2];
  const int o = blockIdx.y, tile = blockIdx.x, bx = tile % ntx, by = tile / ntx;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = dm.r0[o];
  const double* S0 = S0all + (size_t)o * dm.r0u * dm.cnt0;
  const double* Vb = Vball + (size_t)o * 3 * dm.r0u * dm.nlines;
  // the centre of the cell, or the last candidate of a cell the grid ends in: every candidate within kGradStep / 2 of a sample
  auto xs = [&](int i) { const long long x = (long long)bx * 128 + i * kGradStep; return x >= dm.cnt0 ? -1ll : (x + kGradStep / 2 < dm.cnt0 ? x + kGradStep / 2 : dm.cnt0 - 1); };
  auto ls = [&](int j) { const long long l = (long long)by * kGradTL + j * kGradStep; return l >= dm.nlines ? -1ll : (l + kGradStep / 2 < dm.nlines ? l + kGradStep / 2 : dm.nlines - 1); };
  for (int e = tid; e < r0 * kGradNpx; e += blockDim.x) {
    const int p = e / kGradNpx, i = e % kGradNpx;
    const long long x = xs(i);
    sS[p][i] = x >= 0 ? S0[(size_t)p * dm.cnt0 + x] : 0.0;
  }
  for (int e = tid; e < 3 * r0 * kGradNpl; e += blockDim.x) {
    const int j = e % kGradNpl, p = (e / kGradNpl) % r0, b = e / (kGradNpl * r0);
    const long long l = ls(j);
    sV[b][p][j] = l >= 0 ? Vb[((size_t)b * dm.r0u + p) * dm.nlines + l] : 0.0;
  }
  __syncthreads();
  double m0 = 0.0, m1 = 0.0;
  {
    const int i = tid % kGradNpx, j = tid / kGradNpx;      // 16 x 8 samples: one per thread
    const long long x = xs(i), l = ls(j);
    if (x >= 0 && l >= 0) {
      const double x0n = xn0[x], x1n = xn1[l];
      double g0 = 0.0, g1 = 0.0;
      for (int p = 0; p < r0; ++p) {
        const double s0 = sS[p][i], v0 = sV[0][p][j];
        g0 = fma(sV[1][p][j] - x0n * v0, s0, g0);
        g1 = fma(sV[2][p][j] - x1n * v0, s0, g1);
      }
      m0 = fabs(g0);
      m1 = fabs(g1);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    m0 = fmax(m0, __shfl_xor(m0, off));
    m1 = fmax(m1, __shfl_xor(m1, off));
  }
  if (lane == 0) { red[wave][0] = m0; red[wave][1] = m1; }
  __syncthreads();
  if (tid == 0) {
    m0 = fmax(m0, red[1][0]);
    m1 = fmax(m1, red[1][1]);
    const size_t nt = (size_t)gridDim.x;
    tmax[((size_t)o * 2 + 0) * nt + tile] = m0;
    tmax[((size_t)o * 2 + 1) * nt + tile] = m1;
    // (the grid's largest samples: k_bl_gradmax -- 4096 workgroups on four addresses queue their atomics up in L2)
  }
}
__global__ __launch_bounds__(256) void k_bl_gradmax(const double* __restrict__ tmax, int nt, unsigned long long* __restrict__ gkey) {
  __shared__ double red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  double m = 0.0;
  for (int i = tid; i < nt; i += blockDim.x) m = fmax(m, tmax[(size_t)row * nt + i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) gkey[row] = (unsigned long long)__double_as_longlong(fmax(fmax(red[0], red[1]), fmax(red[2], red[3])));
}

// ---- plan ------------------------------------------------------------------------------------------------
static double axis_position(const sbo_ctx* c, int a, long long i) {
  const CandSpec& cs = c->cs;
  const long long tot = cs.count[a];
  const double x = (i == tot - 1 && tot > 1) ? cs.hi[a] : cs.lo[a] + (double)i * cs.step[a];
  return (x - c->mc.X_mean[a]) / c->mc.X_std[a];                                                  // GP_Safe.py:326
}

bool bilinear_applicable(const sbo_ctx* c) {
  const CandSpec& cs = c->cs;
  if (!c->bilinear || c->dtype != SBO_F64 || !c->has_cand || cs.kind != 1 || cs.d != 2 || c->mc.d != 2) return false;
  const long long cnt0 = cs.count[0];
  if ((cs.n_local <= 0 && !(c->sharded && c->world > 1)) || cs.first % cnt0 != 0 || cs.n_local % cnt0 != 0) return false;
  // the bases pay off (and the interpolation interval is meaningful) only on real grids
  // (ranks > 1: ONE decision for all of them -- the recheck behind an approximating posterior contains collectives, so a rank that
  // took the exact kernel K1g for its 15-line shard would return while its 16-line neighbours wait for it there.  The shard sizes
  // are known to every rank: the smallest one decides.)
  long long min_lines = cs.n_local / cnt0;
  if (c->sharded && c->first_of.size() == (size_t)c->world + 1)
    for (int r = 0; r < c->world; ++r) min_lines = std::min(min_lines, (c->first_of[r + 1] - c->first_of[r]) / cnt0);
  // (a caller's matrix whose factor is deferred: the band's reference runs on the matrix itself and must fit its LDS budget)
  if (c->mc.factor == SBO_FACTOR_INVK && c->chol_async && c->mc.npad > kGuardRefMaxNpad) return false;
  return cnt0 >= 64 && cs.count[1] >= 64 && min_lines >= 16 && c->alpha64.p != nullptr && c->f_cap >= c->mc.n;
}

// layout of bl_basis (doubles): U [2q][kBlMaxR][n] | Vs [2q][kBlMaxR][kBlMaxRc] | sig [2q][kBlMaxR] | info (ints, 2q x 4, in
// 4 q doubles) | rcjob | eps [2q][2] | work [2q][3 n kBlMaxRc + kBlMaxR kBlMaxRc]
struct BasisLayout {
  size_t U, Vs, sig, info, rcjob, eps, work, work_stride, total;
};
static BasisLayout basis_layout(int n, int q) {
  BasisLayout L;
  L.U = 0;
  L.Vs = L.U + (size_t)2 * q * kBlMaxR * n;
  L.sig = L.Vs + (size_t)2 * q * kBlMaxR * kBlMaxRc;
  L.info = L.sig + (size_t)2 * q * kBlMaxR;
  L.rcjob = L.info + (size_t)4 * q;                 // ints, 2 q (in q doubles): the degree k_bl_degree found per job
  L.eps = L.rcjob + (size_t)q;                      // [2 q][2]: truncation figures of the axis factors for the guard band (GbAnalytic)
  L.work = L.eps + (size_t)4 * q;
  L.work_stride = (size_t)3 * n * kBlMaxRc + (size_t)kBlMaxR * kBlMaxRc;
  L.total = L.work + (size_t)2 * q * L.work_stride;
  return L;
}

// The interval of each grid axis in normalised coordinates.  Axis 1 always spans the WHOLE axis, so that every rank of a
// sharded grid builds the same functions (shard results are bit-identical to the whole grid's).
static bool basis_intervals(const sbo_ctx* c, double (&ab)[4]) {
  for (int a = 0; a < 2; ++a) {
    const double x0 = axis_position(c, a, 0), x1 = axis_position(c, a, c->cs.count[a] - 1);
    ab[2 * a] = std::min(x0, x1);
    ab[2 * a + 1] = std::max(x0, x1);
    if (!(ab[2 * a + 1] > ab[2 * a])) return false;
  }
  return true;
}

// Enqueues the 2 q basis workgroups and the read-back of their (ok, r, rc) records on `st`; the caller synchronises.
// Used by bilinear_setup, and ahead of time by sbo_model_set (next to the factorisation, on a second stream) when a grid
// is already resident.
int bilinear_basis_enqueue(sbo_ctx* c, hipStream_t st, bool force_big) {
  const ModelConst& mc = c->mc;
  const int n = mc.n, q = mc.q;
  double ab[4];
  c->bl_basis_ok = false;
  if (!basis_intervals(c, ab)) return SBO_OK;
  const BasisLayout L = basis_layout(n, q);
  int rc;
  if ((rc = ensure(c->bl_basis, sizeof(double) * L.total))) return rc;
  double* base = (double*)c->bl_basis.p;
  {
    BlJobs jb;
    memset(&jb, 0, sizeof(jb));
    jb.n = n;
    jb.dpad = mc.dpad;
    for (int o = 0; o < q; ++o)
      for (int a = 0; a < 2; ++a) jb.vinv[2 * o + a] = mc.vinv[o][a];
    for (int a = 0; a < 2; ++a) { jb.a[a] = ab[2 * a]; jb.b[a] = ab[2 * a + 1]; }
    // n <= 128: 256 threads, samples / residual rows in LDS (degree <= 64 assumed: a larger one fails over to the big form)
    const bool small = !force_big && n <= kBasisResLds / 64;
    const size_t dyn = sizeof(double) * (kBasisQLds + (small ? kBasisResLds : 0));
    long long* dbg = nullptr;
    const bool phases = getenv("SBO_BL_TIMING") != nullptr;
    if (phases && hipMalloc(&dbg, sizeof(long long) * 81) != hipSuccess) dbg = nullptr;
    // front end on the whole chip: degree per job, then its coefficient rows into the jobs' workspaces
    int* rcjob = (int*)(base + L.rcjob);
    SBO_HIP(hipMemsetAsync(rcjob, 0, sizeof(double) * 5 * q, st));          // (the degrees and, behind them, the truncation figures)
    hipLaunchKernelGGL(k_bl_degree, dim3((unsigned)((n + 63) / 64), (unsigned)(2 * q)), dim3(256), 0, st, jb, (const double*)c->Xn.p, rcjob);
    hipLaunchKernelGGL(k_bl_coef, dim3((unsigned)((n + kCoefRows - 1) / kCoefRows), (unsigned)(2 * q)), dim3(256), 0, st, jb,
                       (const double*)c->Xn.p, (const int*)rcjob, base + L.work, L.work_stride, base + L.eps);
    for (int rep = 0; rep < (dbg ? 2 : 1); ++rep) {       // (with SBO_BL_TIMING a second run records the phase stamps of job 0)
      long long* dg = rep ? dbg : nullptr;
      if (rep)      // (the pivot loop consumed the rows in place)
        hipLaunchKernelGGL(k_bl_coef, dim3((unsigned)((n + kCoefRows - 1) / kCoefRows), (unsigned)(2 * q)), dim3(256), 0, st, jb,
                           (const double*)c->Xn.p, (const int*)rcjob, base + L.work, L.work_stride, base + L.eps);
      if (small) {
        SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bl_basis<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL((k_bl_basis<256, true>), dim3((unsigned)(2 * q)), dim3(256), dyn, st, jb, (const double*)c->Xn.p, base + L.work,
                           L.work_stride, base + L.U, base + L.Vs, base + L.sig, (int*)(base + L.info), dg, (const int*)rcjob, 0, base + L.eps);
      } else {
        SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bl_basis<1024, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        hipLaunchKernelGGL((k_bl_basis<1024, false>), dim3((unsigned)(2 * q)), dim3(1024), dyn, st, jb, (const double*)c->Xn.p, base + L.work,
                           L.work_stride, base + L.U, base + L.Vs, base + L.sig, (int*)(base + L.info), dg, (const int*)rcjob, 0, base + L.eps);
      }
    }
    SBO_HIP(hipGetLastError());
    if (dbg) {
      long long h[81];
      if (hipMemcpyAsync(h, dbg, sizeof(h), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess) {
        fprintf(stderr, "[k_bl_basis] job 0 phases (us):");
        for (int i = 1; i < (int)h[80]; ++i) fprintf(stderr, " %.1f", (double)(h[i] - h[i - 1]) / 100.0);
        fprintf(stderr, "\n");
      }
      (void)hipFree(dbg);
    }
  }
  SBO_HIP(hipMemcpyAsync(c->h_back + 4096, base + L.info, sizeof(int) * 8 * q, hipMemcpyDeviceToHost, st));
  for (int t = 0; t < 4; ++t) c->bl_basis_ab[t] = ab[t];
  c->bl_basis_serial = c->model_serial;
  c->bl_basis_ok = true;       // (enqueued: valid once `st` has drained)
  return SBO_OK;
}

static bool basis_current(const sbo_ctx* c) {
  double ab[4];
  if (!c->bl_basis_ok || c->bl_basis_serial != c->model_serial || !basis_intervals(c, ab)) return false;
  for (int t = 0; t < 4; ++t)
    if (ab[t] != c->bl_basis_ab[t]) return false;
  return true;
}

// Builds the device tables for the current (model, candidates).  Returns SBO_OK with plan.usable = false when the bases do
// not qualify (rank / interpolation limits): the caller then keeps the separable-table kernel.
int bilinear_setup(sbo_ctx* c) {
  BilinearPlan& pl = c->bl;
  pl.valid = true;
  pl.usable = false;
  pl.band_ready = false;
  pl.setup_ms = 0.0;
  const auto t_begin = std::chrono::steady_clock::now();
  const bool timing = getenv("SBO_BL_TIMING") != nullptr;
  auto lap = [&](const char* what) {
    if (timing) fprintf(stderr, "[K1b setup] %-12s %8.2f ms\n", what,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  const ModelConst& mc = c->mc;
  const CandSpec& cs = c->cs;
  const int n = mc.n, q = mc.q;
  const long long cnt0 = cs.count[0], nlines = cs.n_local / cnt0, line0 = cs.first / cnt0;
  int rc;
  if (!basis_current(c)) {
    if ((rc = bilinear_basis_enqueue(c, c->stream, false))) return rc;
    if (!c->bl_basis_ok) return SBO_OK;
    SBO_HIP(hipStreamSynchronize(c->stream));
  }
  const int* inf = (const int*)(c->h_back + 4096);
  for (int t = 0; t < 2 * q; ++t)
    if (inf[4 * t] == 2) {                 // the small form of the basis kernel needs more LDS than it has: the big form
      if ((rc = bilinear_basis_enqueue(c, c->stream, true))) return rc;
      SBO_HIP(hipStreamSynchronize(c->stream));
      break;
    }
  lap("bases");
  BlDims dm;
  memset(&dm, 0, sizeof(dm));
  int r0u = 0, r1u = 0, K0 = 0, K1 = 0, Rmax = 0, rc0m = 0, rc1m = 0;
  for (int o = 0; o < q; ++o) {
    if (inf[8 * o] != 1 || inf[8 * o + 4] != 1) return SBO_OK;    // a basis did not qualify
    dm.r0[o] = inf[8 * o + 1]; dm.rc0[o] = inf[8 * o + 2];
    dm.r1[o] = inf[8 * o + 5]; dm.rc1[o] = inf[8 * o + 6];
    if (dm.r0[o] < 1 || dm.r0[o] > kBlMaxR || dm.r1[o] < 1 || dm.r1[o] > kBlMaxR) return SBO_OK;
    r0u = std::max(r0u, dm.r0[o]);
    r1u = std::max(r1u, dm.r1[o]);
    K0 = std::max(K0, pair_count(dm.r0[o]));
    K1 = std::max(K1, pair_count(dm.r1[o]));
    Rmax = std::max(Rmax, dm.r0[o] * dm.r1[o]);
    rc0m = std::max(rc0m, dm.rc0[o]);
    rc1m = std::max(rc1m, dm.rc1[o]);
    dm.sf2[o] = mc.sf2[o];
    pl.r0[o] = dm.r0[o];
    pl.r1[o] = dm.r1[o];
    if (timing) fprintf(stderr, "[K1b setup] output %d: r0 %d (degree %d), r1 %d (degree %d)\n", o, dm.r0[o], dm.rc0[o], dm.r1[o], dm.rc1[o]);
  }
  // inner dimensions of the two GEMMs of the variance phase: the degrees of quad as a polynomial of the axis (D = 2 rc - 1, cut
  // on the device where its coefficients have decayed); K0m / K1m: the pair counts r (r + 1) / 2 the core is contracted from
  const int K0m = K0, K1m = K1;
  K0 = 2 * rc0m - 1;
  K1 = 2 * rc1m - 1;
  const int KB0 = (K0 + 15) / 16, KB1 = (K1 + 15) / 16;
  const int ncs0 = (int)((cnt0 + 15) / 16), nrb = (int)((nlines + 15) / 16);
  {
    // worth it only while the two GEMMs issue clearly fewer flops than the triangular contraction of K1g
    const double gemm = (double)nlines * KB0 * 16.0 * KB1 * 16.0 + (double)cs.n_local * (KB0 * 16.0 + 3.0 * r0u);
    const double tri = 0.5 * (double)mc.npad * (mc.npad + 16.0) * (double)cs.n_local;
    if (gemm > 0.7 * tri) return SBO_OK;
  }
  const long long nlines_pad = (long long)nrb * 16;
  pl.KB0 = KB0; pl.KB1 = KB1; pl.r0u = r0u; pl.ncs0 = ncs0; pl.nrb = nrb; pl.nlines_pad = nlines_pad;
  pl.sP0f = 0;                                              // (one table of Chebyshev polynomials for every output)
  pl.sP1A = 0;
  const size_t nP0f = (size_t)ncs0 * KB0 * 4 * 64, nP1A = (size_t)nrb * KB1 * 256;
  pl.sT4f = (size_t)KB0 * KB1 * 4 * 64;
  pl.sBtA = (size_t)nrb * KB0 * 256;
  // mean phases: K = r0p (basis size rounded to whole k-steps); the axis-0 gradient phase concatenates two such operands
  const int r0p = (r0u + 3) / 4 * 4;
  const int KBm = (r0p + 15) / 16, KBm2 = (2 * r0p + 15) / 16;
  pl.KBm = KBm;
  pl.KBm2 = KBm2;
  pl.KSm = r0p / 4;
  pl.KS0 = KB0 * 4;
  pl.sVA = (size_t)nrb * (2 * KBm + KBm2) * 256;     // image sets  V0 | [V1; V0] | V1x
  pl.sSBf = (size_t)ncs0 * (KBm + KBm2) * 256;      // fragment sets  S0 | [S0; -xn0 S0]
  if ((rc = ensure(c->bl_P0f, sizeof(double) * nP0f))) return rc;
  if ((rc = ensure(c->bl_P1A, sizeof(double) * nP1A))) return rc;
  // Chebyshev core scratch: PC0 | PC1 | T4 plain | Y^T | Chat | eff (ints)
  const size_t D0m = (size_t)KB0 * 16, D1m = (size_t)KB1 * 16;
  // (PC0 as B fragments [D0m / 16][KBp0 * 4][64], PC1^T as A images [D1m / 16][KBp1][256], T4^T as B fragments [KBp0][KBp1 * 4][64],
  // Y' = PC1^T T4^T as A images [D1m / 16][KBp0][256], Chat^T [D1m][D0m] row-major)
  const int KBp0 = (K0m + 15) / 16, KBp1 = (K1m + 15) / 16;
  const size_t nPC0 = D0m * (size_t)KBp0 * 16, nPC1 = D1m * (size_t)KBp1 * 16, nT4p = (size_t)KBp0 * KBp1 * 256, nYt = D1m * (size_t)KBp0 * 16,
               nCh = D0m * D1m;
  if ((rc = ensure(c->bl_cheb, sizeof(double) * (size_t)q * (nPC0 + nPC1 + nT4p + nYt + nCh) + 256))) return rc;
  if ((rc = ensure(c->bl_T4f, sizeof(double) * pl.sT4f * q))) return rc;
  if ((rc = ensure(c->bl_SBf, sizeof(double) * pl.sSBf * q))) return rc;     // mean-phase B fragments
  if ((rc = ensure(c->bl_VA, sizeof(double) * pl.sVA * q))) return rc;      // mean-phase A images
  if ((rc = ensure(c->bl_BtA, sizeof(double) * pl.sBtA * q))) return rc;
  const int KBn = mc.npad / 16, ncsR = (Rmax + 15) / 16;
  const size_t nZf = (size_t)ncsR * KBn * 256;                // fragments of Z, of C, images of C^T: same size
  const size_t ldg = (size_t)ncsR * 16;
  if ((rc = ensure(c->bl_work, sizeof(double) * (size_t)q * (3 * nZf + ldg * ldg)))) return rc;
  // bl_small: xn0 | xn1 (local lines) | S0 | S1 | Mb | Vb  -- everything made on the device
  const size_t oS0 = (size_t)cnt0 + (size_t)nlines, oS1 = oS0 + (size_t)q * r0u * cnt0, oMb = oS1 + (size_t)q * r1u * nlines,
               oVb = oMb + (size_t)q * 3 * r0u * r1u, oEnd = oVb + (size_t)q * 3 * r0u * nlines;
  if ((rc = ensure(c->bl_small, sizeof(double) * oEnd))) return rc;
  dm.K0m = K0m; dm.K1m = K1m; dm.D0m = (int)D0m; dm.D1m = (int)D1m;
  dm.q = q; dm.n = n; dm.KBn = KBn;
  dm.r0u = r0u; dm.r1u = r1u; dm.r0p = r0p; dm.KB0 = KB0; dm.KB1 = KB1; dm.KBm = KBm; dm.KBm2 = KBm2;
  dm.ncs0 = ncs0; dm.nrb = nrb; dm.ncsR = ncsR; dm.cnt0 = cnt0; dm.nlines = nlines;
  for (int a = 0; a < 2; ++a) { dm.a[a] = c->bl_basis_ab[2 * a]; dm.b[a] = c->bl_basis_ab[2 * a + 1]; }
  const BasisLayout L = basis_layout(n, q);
  const double* bb = (const double*)c->bl_basis.p;
  const double *dU = bb + L.U, *dVs = bb + L.Vs, *dsig = bb + L.sig;
  double* dsm = (double*)c->bl_small.p;
  double *dxn0 = dsm, *dxn1 = dsm + cnt0, *dS0 = dsm + oS0, *dS1 = dsm + oS1, *dMb = dsm + oMb, *dVb = dsm + oVb;
  double* Zf = (double*)c->bl_work.p;
  double* Cf = Zf + (size_t)q * nZf;
  double* CtA = Cf + (size_t)q * nZf;
  double* G = CtA + (size_t)q * nZf;
  auto blocks = [&](size_t total, unsigned y) { return dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 16), y); };
  const unsigned uq = (unsigned)q;
  // Two independent chains (r03): X = the form T4 of the variance phase (Z fragments -> two GEMMs -> gather: the long one,
  // ~0.11 ms on config H), Y = everything made from the axis tables (normalised axes, S0 / S1, their pair tables, the mean
  // phases' operands: eight small launches, ~0.09 ms).  Y runs on the second stream beside X and joins before stage 1.
  hipStream_t xs = c->stream, ys = (c->stream2 && !c->is_shadow) ? c->stream2 : c->stream;
  // (r04) Z = the guard band's reference values at the probe points (guard.hip: the reference formula itself, 0.12 ms on config
  // H and independent of everything here): a third stream, or Y -- now also carrying the gradient gate -- becomes the long chain
  hipStream_t zs = (ys != xs && c->stream3) ? c->stream3 : ys;
  if (ys != xs) {
    SBO_HIP(hipEventRecord(c->ev[7], xs));                 // (alpha, Xn and the bases are in place at this point of the main stream)
    SBO_HIP(hipStreamWaitEvent(ys, c->ev[7], 0));
    if (zs != ys) SBO_HIP(hipStreamWaitEvent(zs, c->ev[7], 0));
  }
  // (enqueue order: with a caller's invK the head of the long chain X goes first -- twelve short launches of Y ahead of it cost
  // X ~60 us of host time; with the library's own factor X starts with a host wait for the factorisation, and Y goes first)
  hipLaunchKernelGGL(k_bl_zf, blocks(nZf, uq), dim3(256), 0, xs, dm, dU, nZf, Zf);
  // G = Z^T invK Z.  With the library's own Cholesky factor (M = L^-1): C = M Z from the packed triangular images, written as
  // fragments (k = observation) and as images of C^T, then G = C^T C.  With a CALLER's invK (sbo_ctx::invk_img, r03): W =
  // invK Z with the matrix as given -- the contraction the reference itself performs, models/GP_Safe.py:341-343, no
  // factorisation of an ill-conditioned inverse in between --, then G^T = W^T Z; k_bl_t4f symmetrises what rounding leaves.
  if (mc.factor == SBO_FACTOR_INVK && c->chol_async && !c->invk_img_valid && c->invk_w_valid && (rc = model_pack_invk(c))) return rc;
  const bool direct = mc.factor == SBO_FACTOR_INVK && c->chol_async && c->invk_img_valid;
  auto x_head = [&]() -> int {
  if (!direct && (rc = factor_sync(c))) return rc;
  if (direct) {
    hipLaunchKernelGGL((k_bgemm<4, 0, 1>), dim3((unsigned)((ncsR + 3) / 4), (unsigned)((KBn + 3) / 4), uq), dim3(256), 0, c->stream,
                       (const double*)c->invk_img.p, (size_t)mc.npad * mc.npad, (const double*)Zf, nZf, KBn, KBn, ncsR, Cf, nZf, CtA, 0ll);
    hipLaunchKernelGGL((k_bgemm<4, 0, 2>), dim3((unsigned)((ncsR + 3) / 4), (unsigned)((ncsR + 3) / 4), uq), dim3(256), 0, c->stream,
                       (const double*)CtA, nZf, (const double*)Zf, nZf, KBn, ncsR, ncsR, G, ldg * ldg, (double*)nullptr, (long long)ldg);
  } else {
    hipLaunchKernelGGL((k_bgemm<4, 1, 1>), dim3((unsigned)((ncsR + 3) / 4), (unsigned)((KBn + 3) / 4), uq), dim3(256), 0, c->stream,
                       (const double*)c->Fpk.p, c->fpk_stride, (const double*)Zf, nZf, KBn, KBn, ncsR, Cf, nZf, CtA, 0ll);
    // G = C^T C  (R x R, row-major)
    hipLaunchKernelGGL((k_bgemm<4, 0, 2>), dim3((unsigned)((ncsR + 3) / 4), (unsigned)((ncsR + 3) / 4), uq), dim3(256), 0, c->stream,
                       (const double*)CtA, nZf, (const double*)Cf, nZf, KBn, ncsR, ncsR, G, ldg * ldg, (double*)nullptr, (long long)ldg);
  }
  return SBO_OK;
  };
  if (direct && (rc = x_head())) return rc;
  hipLaunchKernelGGL(k_bl_axes, dim3((unsigned)std::min<long long>((cnt0 + nlines + 255) / 256, 4096)), dim3(256), 0, ys, mc, cs,
                     cnt0, line0, nlines, dxn0, dxn1);
  hipLaunchKernelGGL(k_bl_stab, blocks((size_t)std::max(r0u * cnt0, r1u * nlines), 2 * uq), dim3(256), 0, ys, dm, dVs, dsig,
                     (const double*)dxn0, (const double*)dxn1, dS0, dS1);
  hipLaunchKernelGGL((k_cheb_tab<1>), dim3((unsigned)((ncs0 * 16 + 255) / 256)), dim3(256), 0, ys, dm, (const double*)dxn0, (double*)c->bl_P0f.p);
  hipLaunchKernelGGL((k_cheb_tab<0>), dim3((unsigned)((nrb * 16 + 255) / 256)), dim3(256), 0, ys, dm, (const double*)dxn1, (double*)c->bl_P1A.p);
  // mean phases: Mb (forms of alpha, alpha Xn_0, alpha Xn_1) -> Vb = Mb S1 -> A images [V0 | V1;V0 | V1x], B fragments
  // [S0 | S0;-xn0 S0]
  hipLaunchKernelGGL(k_bl_mb, blocks((size_t)3 * r0u * r1u * 64, uq), dim3(256), 0, ys, dm, dU, (const double*)c->alpha64.p,
                     c->a_ld, (const double*)c->Xn.p, mc.dpad, dMb);
  hipLaunchKernelGGL(k_bl_vb, blocks((size_t)3 * r0u * nlines, uq), dim3(256), 0, ys, dm, (const double*)dMb, (const double*)dS1, dVb);
  hipLaunchKernelGGL(k_bl_va, blocks(pl.sVA, uq), dim3(256), 0, ys, dm, (const double*)dVb, (const double*)dxn1, pl.sVA,
                     (double*)c->bl_VA.p);
  hipLaunchKernelGGL(k_bl_sbf, blocks(pl.sSBf, uq), dim3(256), 0, ys, dm, (const double*)dS0, (const double*)dxn0, pl.sSBf,
                     (double*)c->bl_SBf.p);
  // where the gradient phases of k_bpost have to run (k_bl_gradbound / k_bl_gradcoarse): [q][2][tiles] largest samples | [q][2]
  // slacks | [q][2] keys of the grid's largest samples | scratch of the coefficient kernel
  pl.gtmax = nullptr;
  pl.gkey = nullptr;
  {
    static_assert(kGradNpx * kGradNpl == 128, "k_bl_gradcoarse: one sample per thread");
    const int ntx = (ncs0 + 7) / 8, nty = (nrb + 3) / 4;
    const size_t nt = (size_t)ntx * nty, head = (size_t)q * 2 * nt + 4 * (size_t)q;
    const int rows = std::max(rc0m, 1) + 1;                  // coefficient rows 0 .. largest degree + 1
    if ((rc = ensure(c->bl_grad, sizeof(double) * (head + (size_t)q * (3 * kBlMaxR * kBlMaxRc + 4 * (size_t)rows))))) return rc;
    double* gt = (double*)c->bl_grad.p;
    double* slack = gt + (size_t)q * 2 * nt;
    unsigned long long* gkey = (unsigned long long*)(slack + 2 * q);
    // half a sampling cell in the Chebyshev variable of each axis: kGradStep / 2 grid steps
    const double half0 = 0.5 * (dm.b[0] - dm.a[0]), half1 = 0.5 * (dm.b[1] - dm.a[1]);
    const double dxi0 = cs.count[0] > 1 ? 0.5 * kGradStep * std::fabs(cs.step[0] / mc.X_std[0]) / half0 : 0.0;
    const double dxi1 = cs.count[1] > 1 ? 0.5 * kGradStep * std::fabs(cs.step[1] / mc.X_std[1]) / half1 : 0.0;
    SBO_HIP(hipMemsetAsync(gkey, 0, sizeof(unsigned long long) * 2 * q, ys));
    double* Wscr = (double*)(gkey + 2 * q);
    double* part = Wscr + (size_t)q * 3 * kBlMaxR * kBlMaxRc;
    hipLaunchKernelGGL(k_bl_gradw, dim3((unsigned)(3 * r0u), uq), dim3(128), 0, ys, dm, dVs, dsig, (const double*)dMb, Wscr);
    hipLaunchKernelGGL(k_bl_gradrow, dim3((unsigned)rows, uq), dim3(256), 0, ys, dm, dVs, dsig, (const double*)Wscr, part);
    hipLaunchKernelGGL(k_bl_gradslack, dim3(uq), dim3(64), 0, ys, (const double*)part, rows, dxi0, dxi1, slack);
    hipLaunchKernelGGL(k_bl_gradcoarse, dim3((unsigned)nt, uq), dim3(128), 0, ys, dm, (const double*)dS0, (const double*)dVb, (const double*)dxn0,
                       (const double*)dxn1, ntx, gt, gkey);
    hipLaunchKernelGGL(k_bl_gradmax, dim3(2 * uq), dim3(256), 0, ys, (const double*)gt, (int)nt, gkey);
    if (std::isfinite(dxi0) && std::isfinite(dxi1)) {
      pl.gtmax = gt;
      pl.gkey = gkey;
    }
  }
  // guard band of this plan (guard.hip): the exact evaluator at the probe points runs here, beside the core's GEMM chain
  const bool band = c->guard_band && !c->is_shadow;
  double *gref_m = nullptr, *gref_v = nullptr;
  if (band && (rc = guard_probe_reference(c, zs, &gref_m, &gref_v))) return rc;
  // (and the exact gradient components there: the scale the analytic band of the Lipschitz keys is taken relative to)
  double* gref_g = band ? gref_m + (4 * (size_t)q + 2) * kGbProbes : nullptr;
  if (band && (rc = guard_probe_gradients(c, zs, gref_m + 4 * (size_t)q * kGbProbes, gref_g, nullptr))) return rc;
  if (band && zs != ys) SBO_HIP(hipEventRecord(c->ev_join[6], zs));
  if (ys != xs) SBO_HIP(hipEventRecord(c->ev_join[3], ys));
  if (!direct && (rc = x_head())) return rc;
  {
    double* PC0 = (double*)c->bl_cheb.p;
    double* PC1 = PC0 + (size_t)q * nPC0;
    double* T4p = PC1 + (size_t)q * nPC1;
    double* Yt = T4p + (size_t)q * nT4p;
    double* Chat = Yt + (size_t)q * nYt;
    int* eff = (int*)(Chat + (size_t)q * nCh);
    // Chat^T = (PC1^T T4^T) PC0 on the matrix cores: PC1^T as A images and PC0 as B fragments straight out of k_cheb_pairs, T4^T
    // as B fragments -- which is what k_bl_t4f writes for the pair form --; Y' = PC1^T T4^T comes out as A images (k_bgemm OMODE 0)
    hipLaunchKernelGGL(k_cheb_pairs, dim3((unsigned)(std::max(KBp0, KBp1) * 16), 2 * uq), dim3(128), 0, xs, dm, dVs, dsig, nPC0, nPC1, PC0, PC1);
    {
      BlDims dp = dm;                    // (the pair form's block counts for the gather)
      dp.KB0 = KBp0;
      dp.KB1 = KBp1;
      hipLaunchKernelGGL(k_bl_t4f, blocks(nT4p, uq), dim3(256), 0, xs, dp, (const double*)G, (long long)ldg, nT4p, T4p, direct ? 1 : 0);
    }
    const int nrbD1 = (int)(D1m / 16), ncsD0 = (int)(D0m / 16);
    hipLaunchKernelGGL((k_bgemm<4, 0, 0>), dim3((unsigned)((KBp0 + 3) / 4), (unsigned)((nrbD1 + 3) / 4), uq), dim3(256), 0, xs, (const double*)PC1,
                       nPC1, (const double*)T4p, nT4p, KBp1, nrbD1, KBp0, Yt, nYt, (double*)nullptr, 0ll);
    hipLaunchKernelGGL((k_bgemm<4, 0, 2>), dim3((unsigned)((ncsD0 + 3) / 4), (unsigned)((nrbD1 + 3) / 4), uq), dim3(256), 0, xs, (const double*)Yt,
                       nYt, (const double*)PC0, nPC0, KBp0, nrbD1, ncsD0, Chat, nCh, (double*)nullptr, (long long)D0m);
    hipLaunchKernelGGL(k_cheb_trunc, dim3(uq), dim3(1024), 0, xs, dm, (const double*)Chat, c->cheb_tol, eff);
    hipLaunchKernelGGL(k_cheb_t4f, blocks(pl.sT4f, uq), dim3(256), 0, xs, dm, (const double*)Chat, pl.sT4f, (double*)c->bl_T4f.p);
    // (the counts also travel to the host, unwaited: the profile's flop count reads them after the next sweep's own sync)
    SBO_HIP(hipMemcpyAsync(c->h_back + 5376, eff, sizeof(int) * 4 * q, hipMemcpyDeviceToHost, xs));
    pl.eff = eff;
  }
  if (ys != xs) SBO_HIP(hipStreamWaitEvent(xs, c->ev_join[3], 0));
  if (band && zs != ys) SBO_HIP(hipStreamWaitEvent(xs, c->ev_join[6], 0));
  if (band) {
    // ... K1b's own values at the probes from the tables just made, and the band from the deviations: in place before the
    // plan's first posterior launch, whose fused classification reads it
    double* pm = gref_m + 2 * (size_t)q * kGbProbes;
    double* pv = pm + (size_t)q * kGbProbes;
    const double* Chat = (const double*)c->bl_cheb.p + (size_t)q * (nPC0 + nPC1 + nT4p + nYt);
    hipLaunchKernelGGL(k_gb_probe_k1b, dim3((unsigned)((kGbProbes + 3) / 4), uq), dim3(256), 0, xs, mc, cs, dm, Chat, (const int*)pl.eff,
                       (const double*)dxn0, (const double*)dxn1, (const double*)dS0, (const double*)dVb, pm, pv);
    GbAnalytic an;
    memset(&an, 0, sizeof(an));
    an.axis_eps = bb + basis_layout(n, q).eps;
    an.alpha = (const double*)c->alpha64.p;
    an.a_ld = c->a_ld;
    an.n = n;
    for (int o = 0; o < q; ++o) { an.rc[o][0] = dm.rc0[o]; an.rc[o][1] = dm.rc1[o]; }
    an.Xn = (const double*)c->Xn.p;
    an.dpad = mc.dpad;
    for (int t = 0; t < 4; ++t) an.ab[t] = c->bl_basis_ab[t];
    an.ref_g = gref_g;
    if ((rc = guard_band_from_probes(c, pm, pv, gref_m, gref_v, reinterpret_cast<const double*>(pl.eff + 4 * q), an))) return rc;
    pl.band_ready = true;
  }
  SBO_HIP(hipGetLastError());
  lap("enqueue");
  if (timing && pl.gtmax) {
    // (diagnostic: how many tiles keep their gradient phases)
    const int ntx = (ncs0 + 7) / 8, nty = (nrb + 3) / 4;
    const size_t nt = (size_t)ntx * nty;
    std::vector<double> h((size_t)q * 2 * nt + 4 * (size_t)q);
    SBO_HIP(hipStreamSynchronize(xs));
    SBO_HIP(hipMemcpy(h.data(), pl.gtmax, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    const double* sl = h.data() + (size_t)q * 2 * nt;
    for (int o = 0; o < q; ++o)
      for (int k = 0; k < 2; ++k) {
        double G;
        memcpy(&G, sl + 2 * q + 2 * o + k, 8);
        size_t run = 0;
        for (size_t i = 0; i < nt; ++i) run += !(h[((size_t)o * 2 + k) * nt + i] + sl[2 * o + k] < G * (1.0 - 1e-12));
        fprintf(stderr, "[K1b setup] gradient phase %d of output %d: %zu of %zu tiles (largest sample %.6e, slack %.3e)\n", k, o, run, nt, G, sl[2 * o + k]);
      }
  }
  pl.usable = true;       // (nothing to wait for: the tables are made in stream order ahead of the posterior kernels)
  pl.setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return SBO_OK;
}


// ---- K1i: the first sweep of a model by interpolation from Chebyshev nodes (r04) -----------------------------------------------
// The reference refits its models after every sample (models/GP_Safe.py:283-304) and sweeps each of them ONCE
// (test/test_SafeOpt.py:144-179), so what an iteration pays for K1b is its plan: axis bases by pivoted Gram-Schmidt (0.16 ms the
// host has to wait for -- their ranks size everything after them), the core's contraction over rank^2 columns, guard probes:
// ~0.45 ms of a 1.07 ms iteration on config H.  K1b's own evaluation stage does not care where its Chebyshev coefficients come
// from.  So, for the first sweep of a model with a caller's invK:
//   1. the posterior's two scalar fields per output -- quad = k*^T invK k* and s1 = k*^T alpha -- EXACTLY (the reference formula,
//      models/GP_Safe.py:341-343, with the matrix as given) at the Dn x Dn tensor grid of Chebyshev nodes of the first kind on the
//      grid's box: K*^T as B fragments from two n x Dn tables of axis factors (the kernel is separable), C = invK K*^T on the matrix
//      cores (k_bgemm on the packed images of invK), then column dots quad_c = Z_c . C_c, s1_c = Z_c . alpha;
//   2. a 2-D discrete cosine transform of each field -> its Chebyshev coefficients, and those of the two gradient sums of the mean
//      by the derivative recurrence (d_{m-1} = d_{m+1} + 2 m c_m);
//   3. the coefficients through k_cheb_trunc / k_cheb_t4f / k_bstage1 / k_bpost as K1b's core goes: four coefficient sets per output.
// Nothing of this needs a number from the device on the host: the plan is enqueued by sbo_model_set behind the upload and the
// first sweep follows in stream order; the bases and K1b's own plan are built when the same model is swept a second time.
// Accuracy: Dn from the length scales as K1t chooses it (32 / 48 / 64); measured on the BASELINE models 1e-13 (mean) and 1e-12
// (variance: the rounding of the reference formula itself) -- and measured again for every plan at the guard band's probe points
// (values and gradient), so the sweep's decisions stay those of the exact kernel whatever the interpolation error is.
constexpr int kIMaxDn = 64;
constexpr double kGbAliasFactor = 4.0;   // aliasing estimate of an interpolant's band, in units of the last four degrees' coefficient sum
constexpr double kGbInf = 1.0e300;          // a probe that is not finite: everything is "inside the band" (guard.hip)
struct InterpDims {
  int Dn, q, n, npad, dpad;
  double mid[2], half[2];                  // node interval of each axis, normalised coordinates
};
// Everything of a plan that changes with the MODEL (hyper-parameters, normalisation, the box in normalised coordinates) reaches the
// plan's kernels through this block in device memory -- launch arguments then depend on the grid and on n only, and the plan of the
// next model of the same shape is the same HIP graph with another block (interp_setup).
struct InterpParams {
  ModelConst mc;
  InterpDims id;
  BlDims dm;
  double dxi0, dxi1;                       // half a sampling cell of the gradient gate in xi units
};
// E[(2 o + axis)][p][j] = (axis == 0 ? sf2 : 1) exp(-1/2 (As_j,axis - xn_p vinv)^2)   (k_bl_zf multiplies the two axes)
__global__ __launch_bounds__(256) void k_i_etab(const InterpParams* __restrict__ P, const double* __restrict__ As, double* __restrict__ E) {
  const ModelConst& mc = P->mc;
  const InterpDims& id = P->id;
  const int job = blockIdx.y, o = job >> 1, axis = job & 1;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < id.Dn * id.n; e += gridDim.x * blockDim.x) {
    const int p = e / id.n, j = e % id.n;
    const double xn = id.mid[axis] + id.half[axis] * cospi(((double)p + 0.5) / (double)id.Dn);
    const double dlt = As[((size_t)o * id.npad + j) * id.dpad + axis] - xn * mc.vinv[o][axis];
    E[((size_t)job * kBlMaxR + p) * id.n + j] = (axis == 0 ? mc.sf2[o] : 1.0) * exp(-0.5 * dlt * dlt);
  }
}
// node values from the fragments of Z = K*^T and C = invK Z ([ncsR][KBn * 4][64], k = observation): V[o][0][c] = sum_j Z_jc C_jc,
// V[o][1][c] = sum_j Z_jc alpha_j.  A wave per strip of 16 columns.
__global__ __launch_bounds__(64) void k_i_nodevals(int KBn, int n, const double* __restrict__ Zfall, const double* __restrict__ Cfall, size_t nZf,
                                                   const double* __restrict__ alpha, int ald, int ncols, double* __restrict__ V) {
  const int o = blockIdx.y, cs = blockIdx.x, l = threadIdx.x;
  const double* Zf = Zfall + (size_t)o * nZf + (size_t)cs * KBn * 256;
  const double* Cf = Cfall + (size_t)o * nZf + (size_t)cs * KBn * 256;
  double aq[4] = {0.0, 0.0, 0.0, 0.0}, am[4] = {0.0, 0.0, 0.0, 0.0};
  for (int ks = 0; ks < KBn * 4; ks += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = ks + u, j = (k >> 2) * 16 + MM<double>::jslot(k & 3, l >> 4);
      const double z = Zf[(size_t)k * 64 + l];
      aq[u] = fma(z, Cf[(size_t)k * 64 + l], aq[u]);
      am[u] = fma(z, j < n ? alpha[(size_t)o * ald + j] : 0.0, am[u]);
    }
  }
  double sq = (aq[0] + aq[1]) + (aq[2] + aq[3]), sm = (am[0] + am[1]) + (am[2] + am[3]);
  sq += __shfl_xor(sq, 16); sm += __shfl_xor(sm, 16);
  sq += __shfl_xor(sq, 32); sm += __shfl_xor(sm, 32);
  const int c = cs * 16 + l;
  if (l < 16 && c < ncols) {
    V[((size_t)o * 2 + 0) * ncols + c] = sq;
    V[((size_t)o * 2 + 1) * ncols + c] = sm;
  }
}
// Chebyshev coefficients of the node fields, ChatT[4 o + f][b][a] (b: degree along axis 1, a: along axis 0; the layout k_cheb_trunc /
// k_cheb_t4f read): f = 0 quad, 1 s1, 2 / 3 the gradient sums g_a = (d s1 / d xn_a) / inv_ell_a of the two axes (k_bpost scales by
// Y_std inv_ell X_rstd).  One workgroup per output; node c = p Dn + s (p: axis 0).  T[m][k] = w_m cos(m pi (k + 1/2) / Dn) / Dn.
__global__ __launch_bounds__(1024) void k_i_dct(const InterpParams* __restrict__ P_, const double* __restrict__ V, double* __restrict__ ChatT_all) {
  const ModelConst& mc = P_->mc;
  const InterpDims& id = P_->id;
  extern __shared__ double sh[];                 // T [Dn][Dn + 1] | field [Dn][Dn + 1] | tmp [Dn][Dn + 1]  (rows padded: column walks)
  const int Dn = id.Dn, o = blockIdx.x >> 1, f = blockIdx.x & 1, tid = threadIdx.x, N2 = Dn * Dn, P = Dn + 1;
  double *T = sh, *F = sh + Dn * P, *W = sh + 2 * Dn * P;
  for (int e = tid; e < N2; e += blockDim.x) {
    const int m = e / Dn, k = e % Dn;
    T[m * P + k] = (m == 0 ? 1.0 : 2.0) / (double)Dn * cospi((double)m * ((double)k + 0.5) / (double)Dn);
    F[m * P + k] = V[((size_t)o * 2 + f) * N2 + e];                                         // F[p][s]
  }
  __syncthreads();
  for (int e = tid; e < N2; e += blockDim.x) {                                                // W[a][s] = sum_p T[a][p] F[p][s]
    const int a = e / Dn, s_ = e % Dn;
    double a0 = 0.0, a1 = 0.0;
    for (int p = 0; p + 1 < Dn; p += 2) {
      a0 = fma(T[a * P + p], F[p * P + s_], a0);
      a1 = fma(T[a * P + p + 1], F[(p + 1) * P + s_], a1);
    }
    W[a * P + s_] = a0 + a1;
  }
  __syncthreads();
  double* out = ChatT_all + (size_t)(4 * o + f) * N2;
  for (int e = tid; e < N2; e += blockDim.x) {                                                // C[a][b] = sum_s W[a][s] T[b][s]
    const int b = e / Dn, a = e % Dn;
    double a0 = 0.0, a1 = 0.0;
    for (int s_ = 0; s_ + 1 < Dn; s_ += 2) {
      a0 = fma(W[a * P + s_], T[b * P + s_], a0);
      a1 = fma(W[a * P + s_ + 1], T[b * P + s_ + 1], a1);
    }
    out[e] = a0 + a1;                                                                         // ChatT[b][a]
    F[b * P + a] = a0 + a1;
  }
  if (f == 0) return;                            // (uniform: the workgroup of the mean sum goes on to its derivatives)
  __syncthreads();
  // derivative series of s1: along axis 0 (index a) for g_0, along axis 1 (index b) for g_1; d xi / d xn = 1 / half
  double* g0 = ChatT_all + (size_t)(4 * o + 2) * N2;
  double* g1 = ChatT_all + (size_t)(4 * o + 3) * N2;
  const double sc0 = 1.0 / (id.half[0] * mc.inv_ell[o][0]), sc1 = 1.0 / (id.half[1] * mc.inv_ell[o][1]);
  for (int r = tid; r < 2 * Dn; r += blockDim.x) {
    const int line = r % Dn;
    const bool along0 = r < Dn;
    // coefficients c_m of this line: along axis 0 the line is a row b of F (stride 1), along axis 1 a column a (stride P)
    const int fb = along0 ? line * P : line, fs = along0 ? 1 : P;
    const int ob = along0 ? line * Dn : line, os = along0 ? 1 : Dn;
    double* dst = along0 ? g0 : g1;
    const double sc = along0 ? sc0 : sc1;
    double d2 = 0.0, d1 = 0.0;                  // d_{m+1}, d_m while walking m = Dn - 1 .. 1
    dst[ob + (Dn - 1) * os] = 0.0;
    for (int m = Dn - 1; m >= 1; --m) {
      const double dm1 = d2 + 2.0 * (double)m * F[fb + m * fs];            // d_{m-1}
      dst[ob + (m - 1) * os] = (m == 1 ? 0.5 * dm1 : dm1) * sc;
      d2 = d1;
      d1 = dm1;
    }
  }
}
// ---- K1i: where the gradient phases have to run (as k_bl_gradcoarse does it for K1b) -------------------------------------------
// The coarse kernel of K1b takes a rank-r0 bilinear form sum_p Vb[p][line] S0[p][x0]; a Chebyshev series is one with
// S0[a][x0] = T_a(xi0(x0)) and Vb[comp][a][line] = sum_b ChatT_comp[b][a] T_b(xi1(line)).
// K1i's tables of the grid positions in ONE launch (the plan is bound by the host's enqueue rate on the smaller grids: every launch
// less is ~7 us): normalised positions xn0 / xn1 (k_bl_axes), Chebyshev polynomials as B fragments of axis 0 / A images of axis 1
// (k_cheb_tab<1> / <0>) and the plain table of axis 0 for the gradient gate.  A thread per position.
__global__ __launch_bounds__(256) void k_i_tabs(const InterpParams* __restrict__ P_, const CandSpec cs, long long line0, double* __restrict__ xn0,
                                                double* __restrict__ xn1, double* __restrict__ P0f, double* __restrict__ P1A,
                                                double* __restrict__ S0all) {
  const ModelConst& mc = P_->mc;
  const BlDims& dm = P_->dm;
  const int KB = dm.KB0, Dn = dm.D0m;
  const long long n0 = (long long)dm.ncs0 * 16, n1 = (long long)dm.nrb * 16;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n0 + n1; t += (long long)gridDim.x * blockDim.x) {
    const int axis = t < n0 ? 0 : 1;
    const long long x = axis == 0 ? t : t - n0, count = axis == 0 ? dm.cnt0 : dm.nlines;
    double xi = 0.0;
    if (x < count) {
      const long long i = axis == 0 ? x : line0 + x, tot = cs.count[axis];
      const double xr = (i == tot - 1 && tot > 1) ? cs.hi[axis] : __dadd_rn(cs.lo[axis], __dmul_rn((double)i, cs.step[axis]));
      const double xn = (xr - mc.X_mean[axis]) / mc.X_std[axis];
      if (axis == 0) xn0[x] = xn; else xn1[x] = xn;
      xi = (2.0 * xn - (dm.a[axis] + dm.b[axis])) / (dm.b[axis] - dm.a[axis]);
      xi = xi < 1.0 ? xi : 1.0;
      xi = xi > -1.0 ? xi : -1.0;
    }
    double t0 = 1.0, t1 = xi;
    for (int k = 0; k < KB * 16; ++k) {
      double v = k == 0 ? t0 : t1;
      if (k >= 2) { v = 2.0 * xi * t1 - t0; t0 = t1; t1 = v; }
      if (x >= count) v = 0.0;
      const int kb = k >> 4, j = k & 15, kk = j >> 2, slot = j & 3;
      if (axis == 0) {
        P0f[(((size_t)(x >> 4) * (KB * 4) + (size_t)(kb * 4 + kk)) << 6) + (size_t)(slot * 16 + (x & 15))] = v;
        if (x < count && k < Dn)
          for (int o = 0; o < dm.q / 4; ++o) S0all[((size_t)o * Dn + k) * dm.cnt0 + x] = v;
      } else {
        P1A[(((size_t)(x >> 4) * KB + kb) << 8) + (size_t)MM<double>::pack_pos((int)(x & 15), slot, kk)] = v;
      }
    }
  }
}
// Vb[o][comp + 1][a][line] for the two gradient sums (comp 0: zero -- the slot K1b's form multiplies by xn); blockIdx.y = o.
// A thread per (line, eight degrees a -- blockIdx.z): its T_b(xi1) in registers, the coefficients through LDS.
template <int DM>
__global__ __launch_bounds__(256) void k_i_rtab(const InterpParams* __restrict__ P_, const double* __restrict__ ChatT_all, const int* __restrict__ eff,
                                                const double* __restrict__ xn1, double* __restrict__ Vball) {
  const BlDims& dm = P_->dm;
  __shared__ double Cs[DM * DM];
  const int o = blockIdx.y, Dn = dm.D0m;
  const long long line = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  double tb[DM];
  {
    double xi = 0.0;
    if (line < dm.nlines) {
      xi = (2.0 * xn1[line] - (dm.a[1] + dm.b[1])) / (dm.b[1] - dm.a[1]);
      xi = xi < 1.0 ? xi : 1.0;
      xi = xi > -1.0 ? xi : -1.0;
    }
    tb[0] = 1.0;
    if (DM > 1) tb[1] = xi;
#pragma unroll
    for (int b = 2; b < DM; ++b) tb[b] = 2.0 * xi * tb[b - 1] - tb[b - 2];
  }
  double* Vb = Vball + (size_t)o * 3 * Dn * dm.nlines;
  for (int comp = 0; comp < 2; ++comp) {
    const double* Ch = ChatT_all + (size_t)(4 * o + 2 + comp) * Dn * Dn;
    const int B = eff[4 * (4 * o + 2 + comp) + 2] * 16;                    // degrees of axis 1 the kernels run
    __syncthreads();
    for (int e = threadIdx.x; e < Dn * Dn; e += blockDim.x) Cs[e] = Ch[e];
    __syncthreads();
    if (line < dm.nlines) {
      for (int a = blockIdx.z * 8; a < (int)blockIdx.z * 8 + 8 && a < Dn; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < DM; ++b)
          if (b < B) s = fma(Cs[b * Dn + a], tb[b], s);
        Vb[((size_t)(comp + 1) * Dn + a) * dm.nlines + line] = s;
        if (comp == 0) Vb[(size_t)a * dm.nlines + line] = 0.0;
      }
    }
  }
}
// bound on what a gradient sum moves by over half a sampling cell, from its coefficients: sum |C| a^2 dxi0 + sum |C| b^2 dxi1
__global__ __launch_bounds__(256) void k_i_gradslack(const InterpParams* __restrict__ P_, const double* __restrict__ ChatT_all,
                                                     double* __restrict__ slack /* [q][2] */) {
  const BlDims& dm = P_->dm;
  const double dxi0 = P_->dxi0, dxi1 = P_->dxi1;
  __shared__ double red[4][2];
  const int oc = blockIdx.x, o = oc >> 1, comp = oc & 1, Dn = dm.D0m, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double* Ch = ChatT_all + (size_t)(4 * o + 2 + comp) * Dn * Dn;
  double sa = 0.0, sb = 0.0;
  for (int e = tid; e < Dn * Dn; e += blockDim.x) {
    const double v = fabs(Ch[e]), a = (double)(e % Dn), b = (double)(e / Dn);
    sa = fma(v, a * a, sa);
    sb = fma(v, b * b, sb);
  }
  sa = wave_sum(sa);
  sb = wave_sum(sb);
  if (lane == 0) { red[wave][0] = sa; red[wave][1] = sb; }
  __syncthreads();
  if (tid == 0) {
    sa = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    sb = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    slack[oc] = (sa * dxi0 + sb * dxi1) * (1.0 + 1e-9);
  }
}
// the plan's own values at the guard band's probe points: raw[f][p] = sum over the degrees the kernels run of ChatT T_a(xi0) T_b(xi1),
// a wave per (probe, coefficient set)
__global__ __launch_bounds__(256) void k_gb_probe_series(const CandSpec cs, const InterpParams* __restrict__ P_, const double* __restrict__ ChatT_all,
                                                         const int* __restrict__ eff, const double* __restrict__ xn0, const double* __restrict__ xn1,
                                                         double* __restrict__ raw) {
  const BlDims& dm = P_->dm;
  const int f = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, D0 = dm.D0m;
  const int p = blockIdx.x * 4 + wave;
  if (p >= kGbProbes) return;
  const int A = eff[4 * f] * 4, B = eff[4 * f + 2] * 16;
  const double* Ch = ChatT_all + (size_t)f * D0 * dm.D1m;
  long long x0, x1;
  gb_probe_xy(cs, dm.nlines, p, x0, x1);
  auto xi_of = [&](double xn, int axis) {
    double xi = (2.0 * xn - (dm.a[axis] + dm.b[axis])) / (dm.b[axis] - dm.a[axis]);
    xi = xi < 1.0 ? xi : 1.0;
    return xi > -1.0 ? xi : -1.0;
  };
  const double xi0 = xi_of(xn0[x0], 0), xi1 = xi_of(xn1[x1], 1);
  double sum = 0.0;
  for (int b = lane; b < B; b += 64) {
    double tb0 = 1.0, tb1 = xi1, tb = b == 0 ? 1.0 : xi1;
    for (int k = 2; k <= b; ++k) { tb = 2.0 * xi1 * tb1 - tb0; tb0 = tb1; tb1 = tb; }
    const double* rowp = Ch + (size_t)b * D0;
    double row = 0.0, ta0 = 1.0, ta1 = xi0;
    for (int a = 0; a < A; ++a) {
      double ta = a == 0 ? 1.0 : xi0;
      if (a >= 2) { ta = 2.0 * xi0 * ta1 - ta0; ta0 = ta1; ta1 = ta; }
      row = fma(rowp[a], ta, row);
    }
    sum = fma(row, tb, sum);
  }
  sum = wave_sum(sum);
  if (lane == 0) raw[(size_t)f * kGbProbes + p] = sum;
}
// K1i's band from its probes (as guard.hip's k_gb_band, with the mean's truncation tail and a MEASURED band of the Lipschitz
// keys: the gradient sums are derivatives of an interpolant).  raw [4 q][P]; ref_g [q][2][P] the exact gradient components.
__global__ __launch_bounds__(256) void k_gb_band_i(const InterpParams* __restrict__ P_, const double* __restrict__ raw, const double* __restrict__ ref_m,
                                                   const double* __restrict__ ref_v, const double* __restrict__ ref_g,
                                                   const double* __restrict__ tail /* [4 q] tails | [4 q] frames */, const double* __restrict__ alpha,
                                                   int a_ld, GuardBand* gb, GuardBand* gb_mirror /* pinned host copy (sbo_profile_get) */) {
  const ModelConst& mc = P_->mc;
  __shared__ double sh[4][6];
  __shared__ double sha[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int o = 0; o < mc.q; ++o) {
    const double ys = mc.Y_std[o];
    double e[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};          // |dm|, |dv|, |m|, |v|, |dg|, |g|
    bool bad = false;
    for (int p = tid; p < kGbProbes; p += blockDim.x) {
      double var = mc.sf2[o] - raw[(size_t)(4 * o) * kGbProbes + p];
      var = (var > 0.0 ? var : 0.0) * (ys * ys);
      const double m = (mc.mp[o] + raw[(size_t)(4 * o + 1) * kGbProbes + p]) * ys + mc.Y_mean[o];
      const double rm = ref_m[(size_t)o * kGbProbes + p], rv = ref_v[(size_t)o * kGbProbes + p];
      const double dm_ = fabs(m - rm), dv_ = fabs(var - rv);
      bad = bad || !(dm_ < kGbInf) || !(dv_ < kGbInf);
      e[0] = fmax(e[0], dm_); e[1] = fmax(e[1], dv_); e[2] = fmax(e[2], fabs(rm)); e[3] = fmax(e[3], fabs(rv));
      for (int a = 0; a < 2; ++a) {
        const double g = ys * mc.inv_ell[o][a] * mc.X_rstd[a] * raw[(size_t)(4 * o + 2 + a) * kGbProbes + p];
        const double rg = ref_g[((size_t)o * 2 + a) * kGbProbes + p];
        bad = bad || !(fabs(g - rg) < kGbInf);
        e[4] = fmax(e[4], fabs(g - rg));
        e[5] = fmax(e[5], fabs(rg));
      }
    }
    // (a probe whose deviation is not finite: fmax drops a NaN, so the flag joins the reduction itself -- every thread holds at most
    // one of the 144 probes, and only lane 1's flag used to be published)
    if (bad) e[0] = kGbInf;
    double a1p = 0.0;                                         // ||alpha_o||_1 (the worst-case rounding of the mean's sum: the check below)
    for (int j = tid; j < mc.n; j += blockDim.x) a1p += fabs(alpha[(size_t)o * a_ld + j]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a1p += __shfl_xor(a1p, off);
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) e[k] = fmax(e[k], __shfl_xor(e[k], off));
    __syncthreads();
    if (lane == 0) {
      for (int k = 0; k < 6; ++k) sh[wave][k] = e[k];
      sha[wave] = a1p;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w)
        for (int k = 0; k < 6; ++k) e[k] = fmax(e[k], sh[w][k]);
      e[0] = fmax(e[0], sh[0][0]);
      const double eps = 2.220446049250313e-16;
      const bool inf = !(e[0] < kGbInf);
      // analytic part (r05): the dropped coefficients of the series that is run (|T_a T_b| <= 1), and the interpolation error of the node
      // fields -- at most twice the sum of the TRUE coefficients beyond the node count, which is extrapolated from the last four degrees
      // held (kGbAliasFactor x their sum: twice a geometric continuation at a ratio <= 0.8 per degree; the posterior of an RBF kernel is
      // entire, its coefficients decay faster than any such ratio once they decay at all)
      const double* frame = tail + 4 * mc.q;
      const double an_m = (tail[4 * o + 1] + kGbAliasFactor * frame[4 * o + 1]) * ys;
      const double an_v = (tail[4 * o] + kGbAliasFactor * frame[4 * o]) * ys * ys;
      const double fl_m = 64.0 * eps * fmax(e[2], fabs(mc.Y_mean[o]) + ys), fl_v = 64.0 * eps * fmax(e[3], mc.sf2[o] * ys * ys);
      gb->an_m[o] = an_m; gb->an_v[o] = an_v; gb->pr_m[o] = e[0]; gb->pr_v[o] = e[1];
      // ... plus the measured rounding level of the plan's sums; the check: a probe deviation that truncation + the worst-case rounding
      // of the reference formula do not explain (GuardBand, device_common.hpp)
      const double a1 = (sha[0] + sha[1]) + (sha[2] + sha[3]);
      const bool distrust = e[0] > an_m + gb_round_mean(mc.n, mc.sf2[o], a1, ys) || e[1] > an_v + gb_round_var(mc.n, mc.sf2[o], mc.sn2[o], ys);
      gb->dm[o] = (inf || distrust) ? kGbInf : an_m + kGbSafety * e[0] + fl_m;
      gb->dv[o] = (inf || distrust) ? kGbInf : an_v + kGbSafety * e[1] + fl_v;
      gb->rl[o] = (e[5] > 0.0 && !inf) ? 16.0 * e[4] / e[5] + 1e-9 : 1e-3;
      if (gb_mirror) {
        gb_mirror->dm[o] = gb->dm[o]; gb_mirror->dv[o] = gb->dv[o]; gb_mirror->rl[o] = gb->rl[o];
        gb_mirror->an_m[o] = an_m; gb_mirror->an_v[o] = an_v; gb_mirror->pr_m[o] = e[0]; gb_mirror->pr_v[o] = e[1];
      }
    }
    __syncthreads();
  }
}

bool interp_applicable(const sbo_ctx* c) {
  if (c->bilinear != 1 || c->is_shadow || !bilinear_applicable(c)) return false;
  // (the node values come from the reference formula on the caller's matrix: the packed images sbo_model_set made of it)
  // (a grid that arrived after the model: the images are packed on demand from the upload that still sits in the build workspace)
  return c->mc.factor == SBO_FACTOR_INVK && c->chol_async && (c->invk_img_valid || c->invk_w_valid) && c->mc.npad % 16 == 0 &&
         c->dtype == SBO_F64;
}

// Enqueues the plan for the current (model, grid) on the context's streams; nothing waits for the device.
int interp_setup(sbo_ctx* c) {
  InterpPlan& ip = c->bi;
  ip.valid = true;
  ip.usable = false;
  ip.used = false;
  ip.serial = c->model_serial;
  const ModelConst& mc = c->mc;
  const CandSpec& cs = c->cs;
  const int n = mc.n, q = mc.q;
  const long long cnt0 = cs.count[0], nlines = cs.n_local / cnt0, line0 = cs.first / cnt0;
  double ab[4];
  if (!basis_intervals(c, ab)) return SBO_OK;
  // nodes per axis from the shortest length scale (tensor.hip's rule for its first two axes), one count for both
  int Dn = 32;
  for (int a = 0; a < 2; ++a) {
    double tmax = 0.0;
    for (int o = 0; o < q; ++o) tmax = std::max(tmax, (ab[2 * a + 1] - ab[2 * a]) * std::sqrt(mc.inv_ell[o][a]));
    const double want = 7.6 * tmax;
    const int need = want <= 32 ? 32 : (want <= 48 ? 48 : (want <= 64 ? 64 : 1 << 20));
    Dn = std::max(Dn, need);
  }
  if (Dn > kIMaxDn || 2 * Dn > cnt0 || 2 * Dn > cs.count[1]) return SBO_OK;      // (not worth it / not resolvable: K1b's plan takes over)
  const int QP = 4 * q;
  if (QP > 4 * kMaxQ) return SBO_OK;
  int rc;
  if (!c->invk_img_valid && (rc = model_pack_invk(c))) return rc;
  BlDims dm;
  memset(&dm, 0, sizeof(dm));
  const int KB = Dn / 16, KBn = mc.npad / 16, ncols = Dn * Dn, ncsR = ncols / 16;
  const int ncs0 = (int)((cnt0 + 15) / 16), nrb = (int)((nlines + 15) / 16);
  dm.q = QP; dm.n = n; dm.KBn = KBn; dm.ncsR = ncsR;
  for (int o = 0; o < kMaxQ; ++o) { dm.r0[o] = Dn; dm.r1[o] = Dn; dm.rc0[o] = Dn; dm.rc1[o] = Dn; }
  dm.r0u = dm.r1u = Dn; dm.KB0 = dm.KB1 = KB; dm.D0m = dm.D1m = Dn; dm.ncs0 = ncs0; dm.nrb = nrb; dm.cnt0 = cnt0; dm.nlines = nlines;
  for (int a = 0; a < 2; ++a) { dm.a[a] = ab[2 * a]; dm.b[a] = ab[2 * a + 1]; }
  InterpDims id;
  id.Dn = Dn; id.q = q; id.n = n; id.npad = mc.npad; id.dpad = mc.dpad;
  for (int a = 0; a < 2; ++a) { id.mid[a] = 0.5 * (ab[2 * a] + ab[2 * a + 1]); id.half[a] = 0.5 * (ab[2 * a + 1] - ab[2 * a]); }
  ip.Dn = Dn; ip.KB = KB; ip.ncs0 = ncs0; ip.nrb = nrb;
  ip.sT4f = (size_t)KB * KB * 256;
  ip.sBtA = (size_t)nrb * KB * 256;
  const size_t nP0f = (size_t)ncs0 * KB * 256, nP1A = (size_t)nrb * KB * 256, nZf = (size_t)ncsR * KBn * 256;
  if ((rc = ensure(c->bl_P0f, sizeof(double) * nP0f))) return rc;
  if ((rc = ensure(c->bl_P1A, sizeof(double) * nP1A))) return rc;
  if ((rc = ensure(c->bl_T4f, sizeof(double) * ip.sT4f * QP))) return rc;
  if ((rc = ensure(c->bl_BtA, sizeof(double) * ip.sBtA * QP))) return rc;
  if ((rc = ensure(c->bl_small, sizeof(double) * ((size_t)cnt0 + (size_t)nlines)))) return rc;
  // bl_work: E tables [2 q][kBlMaxR][n] | Zf | Cf | CtA (3 x q x nZf) | node fields [q][2][Dn^2]
  const size_t nE = (size_t)2 * q * kBlMaxR * n;
  if ((rc = ensure(c->bl_work, sizeof(double) * (nE + 3 * (size_t)q * nZf + (size_t)q * 2 * ncols)))) return rc;
  // bl_cheb: ChatT [4 q][Dn^2] | eff (4 QP ints) + tails (QP doubles)
  if ((rc = ensure(c->bl_cheb, sizeof(double) * ((size_t)QP * ncols + 4 * (size_t)QP + 16)))) return rc;
  double* E = (double*)c->bl_work.p;
  double* Zf = E + nE;
  double* Cf = Zf + (size_t)q * nZf;
  double* CtA = Cf + (size_t)q * nZf;
  double* V = CtA + (size_t)q * nZf;
  double* Chat = (double*)c->bl_cheb.p;
  int* eff = (int*)(Chat + (size_t)QP * ncols);
  double* dxn0 = (double*)c->bl_small.p;
  double* dxn1 = dxn0 + cnt0;
  const unsigned uq = (unsigned)q;
  auto blocks = [&](size_t total, unsigned y) { return dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 16), y); };
  hipStream_t xs = c->stream, ys = c->stream2 ? c->stream2 : c->stream, zs = (ys != xs && c->stream3) ? c->stream3 : ys;
  const bool band = c->guard_band != 0;
  // buffers of the gate and of the probes (before the plan's signature is taken: it holds their addresses)
  const int ntx = (ncs0 + 7) / 8, nty = (nrb + 3) / 4;
  const size_t nt = (size_t)ntx * nty, head = (size_t)q * 2 * nt + 4 * (size_t)q;
  const size_t nS0 = (size_t)q * Dn * cnt0, nVb = (size_t)q * 3 * Dn * nlines;
  if ((rc = ensure(c->bl_grad, sizeof(double) * (head + nS0 + nVb)))) return rc;
  if ((rc = ensure(c->gb_pts, sizeof(double) * ((size_t)QP * kGbProbes + 2 * (size_t)kGbProbes + 2 * (size_t)q * kGbProbes)))) return rc;
  if ((rc = ensure(c->bi_params, sizeof(InterpParams)))) return rc;
  if (!c->h_bi_params && hipHostMalloc(&c->h_bi_params, sizeof(InterpParams), hipHostMallocDefault) != hipSuccess)
    return fail(SBO_E_NOMEM, "pinned staging of the plan's parameters");
  double* gt = (double*)c->bl_grad.p;
  double* slack = gt + (size_t)q * 2 * nt;
  unsigned long long* gkey = (unsigned long long*)(slack + 2 * q);
  double* S0i = (double*)(gkey + 2 * q);
  double* Vbi = S0i + nS0;
  double* raw = (double*)c->gb_pts.p;
  double* ppts = raw + (size_t)QP * kGbProbes;
  double* pgrad = ppts + 2 * (size_t)kGbProbes;
  // ---- what changes with the model: one block, read by the plan's kernels from device memory
  InterpParams hp;
  memset(&hp, 0, sizeof(hp));
  hp.mc = mc;
  hp.id = id;
  hp.dm = dm;
  hp.dxi0 = cs.count[0] > 1 ? 0.5 * kGradStep * std::fabs(cs.step[0] / mc.X_std[0]) / id.half[0] : 0.0;
  hp.dxi1 = cs.count[1] > 1 ? 0.5 * kGradStep * std::fabs(cs.step[1] / mc.X_std[1]) / id.half[1] : 0.0;
  const InterpParams* dP = (const InterpParams*)c->bi_params.p;
  const bool gate = std::isfinite(hp.dxi0) && std::isfinite(hp.dxi1);
  ip.grad_S0 = S0i; ip.grad_Vb = Vbi; ip.grad_gt = gt; ip.grad_key = gkey;
  ip.gtmax = gate ? gt : nullptr;
  ip.gkey = gate ? gkey : nullptr;
  // ---- the plan as a HIP graph.  Its ~20 launches on three streams take the host ~0.15 ms to enqueue -- more than the device needs for
  // them on config B.  Launch arguments depend on the grid, on n and on addresses only (signature below); a model whose signature
  // equals the previous one's replays the captured graph with its own parameter block: one hipGraphLaunch.  (The reference's loop grows
  // n by one per iteration: there every plan is enqueued the plain way, at no extra cost -- a graph is captured only when a
  // signature REPEATS.)
  InterpSig sig;
  memset(&sig, 0, sizeof(sig));
  sig.cs = cs;
  sig.Dn = Dn; sig.q = q; sig.n = n; sig.npad = mc.npad; sig.dpad = mc.dpad; sig.guard = c->guard_band; sig.a_ld = c->a_ld; sig.gate = gate ? 1 : 0;
  sig.cheb_tol = c->cheb_tol;
  {
    const void* ptrs[] = {c->As.p, c->sqA.p, c->alpha.p, c->alpha64.p, c->Xn.p, c->invk_img.p, c->invk_plain, c->bl_P0f.p, c->bl_P1A.p, c->bl_T4f.p,
                          c->bl_BtA.p, c->bl_small.p, c->bl_work.p, c->bl_cheb.p, c->bl_grad.p, c->gb_pts.p, c->gb_probe.p, c->gb_part.p, c->gb.p,
                          c->bi_params.p, c->h_bi_params, (const void*)xs, (const void*)ys, (const void*)zs};
    static_assert(sizeof(ptrs) / sizeof(ptrs[0]) <= sizeof(sig.ptr) / sizeof(sig.ptr[0]), "signature slots");
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); ++i) sig.ptr[i] = ptrs[i];
  }
  auto finish = [&]() {
    if (band) c->gb_host_valid = false;
    ip.eff = eff;
    ip.band_ready = band;
    ip.usable = true;
    return SBO_OK;
  };
  const bool repeat = ip.sig_valid && !memcmp(&sig, &ip.sig, sizeof(sig));
  // (the previous plan's copy of the block has normally run long ago -- a sweep has synchronised since --, but two model changes in
  // a row must not let the first plan's copy read the second model's block)
  if (c->ev_bi_params) SBO_HIP(hipEventSynchronize((hipEvent_t)c->ev_bi_params));
  else {
    hipEvent_t ev;
    SBO_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    c->ev_bi_params = ev;
  }
  memcpy(c->h_bi_params, &hp, sizeof(hp));
  if (repeat && ip.exec) {
    SBO_HIP(hipGraphLaunch((hipGraphExec_t)ip.exec, xs));
    SBO_HIP(hipEventRecord((hipEvent_t)c->ev_bi_params, xs));
    return finish();
  }
  if (!repeat) {
    if (ip.exec) { (void)hipGraphExecDestroy((hipGraphExec_t)ip.exec); ip.exec = nullptr; }
    // (measured, ROCm 7.2: the replayed graph is SLOWER than the plain launches -- config B iteration 0.48 -> 0.91 ms, H 0.97 -> 1.38:
    // the runtime walks the three branches as one chain with ~15 us between nodes.  Off unless SBO_PLAN_GRAPH=1.)
    static const bool want_graph = getenv("SBO_PLAN_GRAPH") != nullptr;
    ip.graph_ok = want_graph && ys != xs;
  }
  // (a captured plan joins all its branches at its end: the deferred gate is for the plain launches only)
  const bool defer = gate && c->grad_defer && !ip.graph_ok && !(repeat && ip.exec) && zs != xs && zs != ys;
  ip.grad_deferred = defer;
  // (no gate at all -- the deferred launch runs both gradient phases on every tile -- where the grid is small enough for the gate's
  // three launches to cost more than the phases they save: A/B r05, config B 0.419 -> 0.398 ms per iteration, config H 0.870 -> 0.920.
  // grad_defer = 2: always; 3: never)
  const bool nogate = defer && (c->grad_defer == 2 || (c->grad_defer == 1 && (long long)ntx * nty * q <= 4ll * c->n_cu));
  if (nogate) { ip.gtmax = nullptr; ip.gkey = nullptr; }
  auto enqueue = [&]() -> int {
    SBO_HIP(hipMemcpyAsync(c->bi_params.p, c->h_bi_params, sizeof(InterpParams), hipMemcpyHostToDevice, xs));
    if (ys != xs) {
      SBO_HIP(hipEventRecord(c->ev[7], xs));               // (the model's arrays and the block are in place at this point of the main stream)
      SBO_HIP(hipStreamWaitEvent(ys, c->ev[7], 0));
      if (zs != ys) SBO_HIP(hipStreamWaitEvent(zs, c->ev[7], 0));
    }
    // (enqueue order: the plan is host-bound on the smaller grids -- Z's one long kernel first, then the head of X, then Y's one launch)
    double *gref_m = nullptr, *gref_v = nullptr;
    int rc2;
    if (band) {
      // Z: the guard band's references at the probe points -- the reference formula (guard.hip) and the exact gradient
      if ((rc2 = guard_probe_reference(c, zs, &gref_m, &gref_v, &dP->mc))) return rc2;
      if ((rc2 = guard_probe_gradients(c, zs, ppts, pgrad, &dP->mc))) return rc2;
      if (zs != ys) SBO_HIP(hipEventRecord(c->ev_join[6], zs));
    }
    // X: node fields and their coefficients
    hipLaunchKernelGGL(k_i_etab, blocks((size_t)Dn * n, 2 * uq), dim3(256), 0, xs, dP, (const double*)c->As.p, E);
    hipLaunchKernelGGL(k_bl_zf, blocks(nZf, uq), dim3(256), 0, xs, dm, (const double*)E, nZf, Zf);
    // (A/B r05: nine column strips per wave -- exactly one wave per SIMD on config H instead of 2.25 -- took 120 us against 93: the
    // fragment loads of a lone wave are not hidden by anything)
    hipLaunchKernelGGL((k_bgemm<4, 0, 1>), dim3((unsigned)((ncsR + 3) / 4), (unsigned)((KBn + 3) / 4), uq), dim3(256), 0, xs,
                       (const double*)c->invk_img.p, (size_t)mc.npad * mc.npad, (const double*)Zf, nZf, KBn, KBn, ncsR, Cf, nZf, CtA, 0ll);
    hipLaunchKernelGGL(k_i_nodevals, dim3((unsigned)ncsR, uq), dim3(64), 0, xs, KBn, n, (const double*)Zf, (const double*)Cf, nZf,
                       (const double*)c->alpha64.p, c->a_ld, ncols, V);
    {
      const size_t lds = sizeof(double) * 3 * (size_t)Dn * (Dn + 1);
      SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_i_dct), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_i_dct, dim3(2 * uq), dim3(1024), lds, xs, dP, (const double*)V, Chat);
    }
    // (deferred tail, r05: the fork of the gate's and the band's side chains rides on this kernel as its stop event -- a record of its own
    // would be a bubble in the chain)
    if (defer) hipExtLaunchKernelGGL(k_cheb_trunc, dim3((unsigned)QP), dim3(1024), 0, xs, nullptr, c->ev_grad[0], 0, dm, (const double*)Chat, c->cheb_tol, eff);
    else hipLaunchKernelGGL(k_cheb_trunc, dim3((unsigned)QP), dim3(1024), 0, xs, dm, (const double*)Chat, c->cheb_tol, eff);
    hipLaunchKernelGGL(k_cheb_t4f, blocks(ip.sT4f, (unsigned)QP), dim3(256), 0, xs, dm, (const double*)Chat, ip.sT4f, (double*)c->bl_T4f.p);
    // ... and which tiles of k_bpost can hold the largest gradient component (the gate of K1b's gradient phases, fed from the series.
    // A/B r04: without the gate -- 80 us of plan kernels against 45 us of gradient phases on every tile -- the iteration times are the
    // same within the spread)
    // (deferred gate, r05: these kernels are 80 us of small launches whose result only the Lipschitz keys need.  They run on stream3 --
    // behind the guard reference there, beside the plan's tail and the posterior launches --, followed by a launch of the gradient
    // phases alone on the tiles they name (launch_posterior_interp); the posterior launches carry none)
    hipStream_t gs = defer ? zs : xs;
    if (defer) SBO_HIP(hipStreamWaitEvent(gs, c->ev_grad[0], 0));
    if (!nogate) hipLaunchKernelGGL(k_i_gradslack, dim3(2 * uq), dim3(256), 0, gs, dP, (const double*)Chat, slack);
    // Y: the tables of the grid positions (one launch)
    hipLaunchKernelGGL(k_i_tabs, dim3((unsigned)std::min<long long>(((long long)(ncs0 + nrb) * 16 + 255) / 256, 4096)), dim3(256), 0, ys, dP, cs, line0,
                       dxn0, dxn1, (double*)c->bl_P0f.p, (double*)c->bl_P1A.p, S0i);
    if (ys != xs) SBO_HIP(hipEventRecord(c->ev_join[3], ys));
    if (ys != xs) SBO_HIP(hipStreamWaitEvent(xs, c->ev_join[3], 0));
    // (X again, with the tables of Y: the lines' sums of the gradient series and the sums at the cell centres of every tile)
    if (defer && ys != gs) SBO_HIP(hipStreamWaitEvent(gs, c->ev_join[3], 0));
    if (gate && !nogate) {
      switch (Dn) {
        case 32: hipLaunchKernelGGL((k_i_rtab<32>), dim3((unsigned)((nlines + 255) / 256), uq, (unsigned)(Dn / 8)), dim3(256), 0, gs, dP, (const double*)Chat, (const int*)eff, (const double*)dxn1, Vbi); break;
        case 48: hipLaunchKernelGGL((k_i_rtab<48>), dim3((unsigned)((nlines + 255) / 256), uq, (unsigned)(Dn / 8)), dim3(256), 0, gs, dP, (const double*)Chat, (const int*)eff, (const double*)dxn1, Vbi); break;
        default: hipLaunchKernelGGL((k_i_rtab<64>), dim3((unsigned)((nlines + 255) / 256), uq, (unsigned)(Dn / 8)), dim3(256), 0, gs, dP, (const double*)Chat, (const int*)eff, (const double*)dxn1, Vbi); break;
      }
      hipLaunchKernelGGL(k_bl_gradcoarse, dim3((unsigned)(ntx * nty), uq), dim3(128), 0, gs, dm, (const double*)S0i, (const double*)Vbi,
                         (const double*)dxn0, (const double*)dxn1, ntx, gt, gkey);
      hipLaunchKernelGGL(k_bl_gradmax, dim3(2 * uq), dim3(256), 0, gs, (const double*)gt, ntx * nty, gkey);
    }
    if (band) {
      // (deferred: the plan's own values at the probes and the band from them on Y -- idle since its tables -- beside the series' fragments
      // and stage 1; the posterior launch, whose classification reads the band, waits for ev_grad[3]: launch_posterior_interp)
      hipStream_t bs = defer ? ys : xs;
      if (defer) SBO_HIP(hipStreamWaitEvent(bs, c->ev_grad[0], 0));
      if (zs != ys) SBO_HIP(hipStreamWaitEvent(bs, c->ev_join[6], 0));
      hipLaunchKernelGGL(k_gb_probe_series, dim3((unsigned)((kGbProbes + 3) / 4), (unsigned)QP), dim3(256), 0, bs, cs, dP, (const double*)Chat,
                         (const int*)eff, (const double*)dxn0, (const double*)dxn1, raw);
      hipLaunchKernelGGL(k_gb_band_i, dim3(1), dim3(256), 0, bs, dP, (const double*)raw, (const double*)gref_m, (const double*)gref_v,
                         (const double*)pgrad, reinterpret_cast<const double*>(eff + 4 * QP), (const double*)c->alpha64.p, c->a_ld, (GuardBand*)c->gb.p,
                         (GuardBand*)(c->h_back + kGbMirrorOffset));
      c->gb_mirrored = true;
      if (defer) SBO_HIP(hipEventRecord(c->ev_grad[3], bs));
    }
    SBO_HIP(hipGetLastError());
    return SBO_OK;
  };
  // the second model of this signature: capture what the first one enqueued the plain way (every buffer is allocated by now), then
  // replay.  Anything the runtime refuses in a capture switches the graph off for this signature, and the plan goes out the plain way.
  const bool reference_ok = !band || guard_reference_is_direct(c);
  if (repeat && ip.graph_ok && reference_ok && c->invk_img_valid) {
    hipGraph_t g = nullptr;
    bool ok = hipStreamBeginCapture(xs, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (ok) {
      const int rce = enqueue();
      const hipError_t ee = hipStreamEndCapture(xs, &g);
      ok = rce == SBO_OK && ee == hipSuccess && g != nullptr;
    }
    hipGraphExec_t ex = nullptr;
    if (ok) ok = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess && ex != nullptr;
    if (g) (void)hipGraphDestroy(g);
    if (ok) {
      ip.exec = ex;
      SBO_HIP(hipGraphLaunch(ex, xs));
      SBO_HIP(hipEventRecord((hipEvent_t)c->ev_bi_params, xs));
      return finish();
    }
    (void)hipGetLastError();
    ip.graph_ok = false;
  }
  ip.sig = sig;
  ip.sig_valid = true;
  if ((rc = enqueue())) return rc;
  SBO_HIP(hipEventRecord((hipEvent_t)c->ev_bi_params, xs));
  return finish();
}

// Column path (r05): can this launch deliver the classification as column words?  One constraint, fp64 grid of whole 64 x 128
// tiles (every workgroup's tile inside the grid), at most 64 segments (Usum is one word per column); the sweep asked for it.
static bool col_words_ok(const sbo_ctx* c, long long cnt0, long long nlines) {
  if (!c->col_request || !c->col_path) return false;
  const bool shape = c->mc.q == 2 && c->cs.kind == 1 && c->cs.d == 2 && c->cs.first == 0 && cnt0 % 128 == 0 && nlines % 64 == 0 &&
                     nlines / 64 <= 64 && nlines >= 64 && cnt0 >= 128 && cnt0 <= 4096 && c->world == 1 && !c->comm_selftest;
  // (auto: from four tiles per CU and output on -- config H.  Below that the two launches per sweep cost more than the column
  // kernels save: config B, 512 tiles per output, 0.171 ms against 0.162 with the byte masks.  Option col_path = 2: whenever the shape fits.)
  const long long tiles = (cnt0 / 128) * (nlines / 64);
  return shape && (c->col_path == 2 || tiles >= 4ll * c->n_cu);
}
static int col_words_prepare(sbo_ctx* c, long long cnt0, long long nlines, ColBits* cb) {
  const size_t words = (size_t)(nlines / 64) * (size_t)cnt0;
  int rc;
  if ((rc = ensure(c->cbS, 8 * words)) || (rc = ensure(c->cbU, 8 * words)) || (rc = ensure(c->cbM, 8 * words)) || (rc = ensure(c->cbG, 8 * words))) return rc;
  const bool fresh = c->cbUsum.bytes < 8 * (size_t)cnt0;
  if ((rc = ensure(c->cbUsum, 8 * (size_t)cnt0))) return rc;
  if (fresh || c->usum_dirty) SBO_HIP(hipMemsetAsync(c->cbUsum.p, 0, c->cbUsum.bytes, c->stream));
  c->usum_dirty = true;       // (until the column path's second kernel has cleared it again)
  const size_t sbytes = sizeof(unsigned long long) * kColSlotFields * kColSlots;
  if (c->col_slots.bytes < sbytes) c->slots_clean = false;
  if ((rc = ensure(c->col_slots, sbytes))) return rc;
  if (!c->slots_clean) {
    unsigned long long init[kColSlotFields * kColSlots];
    for (int f = 0; f < kColSlotFields; ++f)
      for (int k = 0; k < kColSlots; ++k) init[f * kColSlots + k] = col_slot_is_min(f) ? ~0ull : 0ull;
    SBO_HIP(hipMemcpyAsync(c->col_slots.p, init, sbytes, hipMemcpyHostToDevice, c->stream));
    SBO_HIP(hipStreamSynchronize(c->stream));           // (first use, or after a failed sweep: `init` is on this stack)
  }
  c->slots_clean = false;     // (until the sweep's finals have reset the block)
  cb->Sw = (unsigned long long*)c->cbS.p;
  cb->Uw = (unsigned long long*)c->cbU.p;
  cb->Usum = (unsigned long long*)c->cbUsum.p;
  cb->slots = (unsigned long long*)c->col_slots.p;
  return SBO_OK;
}

int launch_posterior_interp(sbo_ctx* c) {
  InterpPlan& ip = c->bi;
  const ModelConst& mc = c->mc;
  const CandSpec& cs = c->cs;
  const int q = mc.q, QP = 4 * q, KB = ip.KB;
  const long long cnt0 = cs.count[0], nlines = cs.n_local / cnt0;
  ip.used = true;
  constexpr int S1 = 3;
  const unsigned gx = (unsigned)((ip.ncs0 + 7) / 8), gy = (unsigned)((ip.nrb + 3) / 4);
  const unsigned rows_out = gx * gy;
  const size_t lds = sizeof(double) * 2 * 3072 + 2048;
  int rc;
  // (deferred gate: the posterior launches write their -- empty -- Lipschitz rows behind the real ones, which the gradient launch fills)
  const bool defer = ip.grad_deferred && c->stream3 != nullptr;
  if ((rc = ensure(c->bl_lpart, sizeof(double) * (size_t)rows_out * q * 2))) return rc;
  const bool fuse_wanted = c->fuse_request == 1 || (c->fuse_request == 2 && ((long long)gx * gy * q >= 4ll * c->n_cu || q > 2));   // (several constraints: the separate pass costs more than one constraint's)
  bool fuse = fuse_wanted && q >= 2 && c->maskS.p && c->maskU.p && c->maskS.bytes >= (size_t)cs.n_local && c->maskU.bytes >= (size_t)cs.n_local &&
              (q == 2 || (c->fuseS.bytes >= (size_t)cs.n_local * (q - 1) && c->fuseU.bytes >= (size_t)cs.n_local * (q - 1)));
  const bool colw = fuse && col_words_ok(c, cnt0, nlines);
  PostExtra px;
  memset(&px, 0, sizeof(px));
  px.q = q;
  c->col_active = false;
  c->fuse_rows = 0;
  if (fuse) {
    c->fuse_rows = (int)rows_out * (colw ? 2 : (q - 1));
    px.fstride = q > 2 ? (long long)cs.n_local : 0ll;
    if ((rc = ensure(c->cpart, sizeof(unsigned long long) * kFuseRow * ((size_t)c->fuse_rows + 4 * (size_t)c->n_cu + 64)))) return rc;
    c->cpart_cap = (int)(c->cpart.bytes / (sizeof(unsigned long long) * kFuseRow));
  }
  if (colw) {
    if ((rc = col_words_prepare(c, cnt0, nlines, &px.cb))) return rc;
    px.lean = c->col_lean;
    c->col_active = true;
    c->col_forked = true;
    fuse = false;                      // (no byte masks: the words are the classification)
  }
  const GuardBand* gb_fused = (c->guard_band && ip.band_ready && c->gb.p) ? (const GuardBand*)c->gb.p : nullptr;
  // stage 1 (behind col_words_prepare: the gradient launch, which follows this kernel on another stream, merges into the slot block)
  if (defer)
    hipExtLaunchKernelGGL((k_bstage1<S1>), dim3((unsigned)((KB + S1 - 1) / S1), (unsigned)((ip.nrb + 3) / 4), (unsigned)QP), dim3(256), 0, c->stream, nullptr,
                          c->ev_grad[1], 0, (const double*)c->bl_P1A.p, (size_t)0, (const double*)c->bl_T4f.p, ip.sT4f, KB, ip.nrb, KB,
                          (double*)c->bl_BtA.p, ip.sBtA, (const int*)ip.eff);
  else
    hipLaunchKernelGGL((k_bstage1<S1>), dim3((unsigned)((KB + S1 - 1) / S1), (unsigned)((ip.nrb + 3) / 4), (unsigned)QP), dim3(256), 0, c->stream,
                       (const double*)c->bl_P1A.p, (size_t)0, (const double*)c->bl_T4f.p, ip.sT4f, KB, ip.nrb, KB, (double*)c->bl_BtA.p, ip.sBtA,
                       (const int*)ip.eff);
  if (defer && ip.band_ready) SBO_HIP(hipStreamWaitEvent(c->stream, c->ev_grad[3], 0));     // (the band: written on Y beside stage 1)
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bpost<1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bpost<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bpost<1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const double* BtA = (const double*)c->bl_BtA.p;
  double* const lrows = (double*)c->bl_lpart.p;
  if (defer) {
    // the gradient phases alone, behind the gate on stream3 and behind stage 1 (its images): one launch for all outputs
    hipStream_t gs = c->stream3;
    SBO_HIP(hipStreamWaitEvent(gs, c->ev_grad[1], 0));
    SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bgrad), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_bgrad, dim3((unsigned)std::min<long long>((long long)rows_out, 2ll * c->n_cu)), dim3(256), lds, gs, mc, cs, BtA + ip.sBtA,
                       4 * ip.sBtA, (const double*)c->bl_P0f.p, KB, ip.nrb, ip.ncs0, nlines, (int)gx, (int)gy, (const int*)ip.eff, ip.gtmax, ip.gkey,
                       (const double*)c->bl_small.p, q, lrows, colw ? px.cb.slots : (unsigned long long*)nullptr, (c->sweep_lean && q >= 2) ? 1 : 0);
    SBO_HIP(hipEventRecord(c->ev_grad[2], gs));
    c->grad_pending = true;
    px.nograd = 1;
  }
  // (column path: the constraint's launch first -- the objective's tiles read its words and its counts of safe candidates per tile)
  for (int part = 0; part < (colw ? 2 : 1); ++part) {
    px.o0 = colw ? 1 - part : 0;
    const bool last = !colw || part == 1;
    auto kpost = !colw ? k_bpost<1, 0> : (part == 0 ? k_bpost<1, 1> : k_bpost<1, 2>);
    // (the constraint's launch carries the fork event of the overlapped sweep: the expander chain starts behind it on stream3
    // while the objective's launch runs here)
    hipExtLaunchKernelGGL(kpost, dim3(gx, gy, (unsigned)(colw ? 1 : q)), dim3(256), lds, c->stream, nullptr,
                          (c->lmax_defer && last) ? c->ev[1] : ((colw && part == 0) ? c->ev_col[0] : nullptr), 0, mc, cs, BtA,
                          4 * ip.sBtA, (const double*)c->bl_P0f.p, (size_t)0, BtA + ip.sBtA, 4 * ip.sBtA, (const double*)c->bl_P0f.p, (size_t)0, KB,
                          KB * 4, KB, KB * 4, KB, ip.nrb, ip.ncs0, nlines, (double*)c->mean.p, (double*)c->var.p,
                          defer ? lrows + (size_t)rows_out * q : lrows,
                          (const double*)c->bl_small.p /* xn0 */, fuse ? (uint8_t*)(q > 2 ? c->fuseS.p : c->maskS.p) : (uint8_t*)nullptr,
                          fuse ? (uint8_t*)(q > 2 ? c->fuseU.p : c->maskU.p) : (uint8_t*)nullptr, c->fuse_b, (unsigned long long*)c->cpart.p, c->cpart_cap, gb_fused,
                          (const int*)ip.eff, ip.gtmax, ip.gkey, 1, px);
  }
  // (whoever merges the Lipschitz rows on this stream -- the reduction below, a sweep's first small kernel -- waits for the gradient launch;
  // the column path reads the keys from the slot block on stream3 itself, in stream order behind that launch: sets_colpath.inc.hpp)
  if (defer && !colw) {
    SBO_HIP(hipStreamWaitEvent(c->stream, c->ev_grad[2], 0));
    c->grad_pending = false;
  }
  if (c->lmax_defer) {
    c->lmax_pending = true;
    c->lmax_per_out = (int)rows_out;
  } else {
    hipExtLaunchKernelGGL(k_lmax_reduce, dim3((unsigned)q), dim3(256), 0, c->stream, nullptr, c->ev[1], 0, (const double*)c->bl_lpart.p,
                          (int)rows_out, (unsigned long long*)c->Lmax.p);
  }
  c->k1_stop_attached = true;
  c->gb_active = c->guard_band && ip.band_ready;
  // flops issued: stage 1 of four coefficient sets per output + four full phases of stage 2 (an upper bound: the counts the kernels
  // run to stay on the device -- a read-back per plan is a launch the host-bound plan does not need)
  const double tiles2 = (double)ip.nrb * ip.ncs0;
  c->last_k1_flops = (double)q * 2.0 * 1024.0 * 4.0 * (4.0 * (double)ip.nrb * KB * KB + tiles2 * KB * 4);
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

int launch_posterior_bilinear(sbo_ctx* c) {
  const BilinearPlan& pl = c->bl;
  const ModelConst& mc = c->mc;
  const CandSpec& cs = c->cs;
  const int q = mc.q;
  const long long cnt0 = cs.count[0], nlines = cs.n_local / cnt0, line0 = cs.first / cnt0;
  // stage 1: Bt = P1^T T4qq^T, written as the packed A operand of stage 2 (three strips per workgroup measured best on
  // config B: 27.7 us against 28.8 with two and 31.3 with four; the per-wave k_bgemm<2, 0, 0> took 32.6)
  constexpr int S1 = 3;
  hipLaunchKernelGGL((k_bstage1<S1>), dim3((unsigned)((pl.KB0 + S1 - 1) / S1), (unsigned)((pl.nrb + 3) / 4), (unsigned)q), dim3(256), 0,
                     c->stream, (const double*)c->bl_P1A.p, pl.sP1A, (const double*)c->bl_T4f.p, pl.sT4f, pl.KB1, pl.nrb, pl.KB0,
                     (double*)c->bl_BtA.p, pl.sBtA, (const int*)pl.eff);
  // stage 2 (fused): variance, mean, Lipschitz keys.  64 x 128 tiles (k_bpost<1>, three workgroups per CU): with the Chebyshev
  // core the variance phase is ~12 k-steps and no longer dominates, and the third workgroup per CU is worth more than the
  // B-fragment reuse of a 128 x 128 tile (r03: config B 0.204 -> 0.189 ms per sweep, H 0.547 -> 0.543)
  const unsigned gx = (unsigned)((pl.ncs0 + 7) / 8);
  constexpr int rbw = 1;
  const size_t lds = sizeof(double) * 2 * 3072 + 2048;
  const unsigned gy = (unsigned)((pl.nrb + 4 * rbw - 1) / (4 * rbw));
  const unsigned rows_out = gx * gy;                      // partial rows per output: one per workgroup
  int rc;
  if ((rc = ensure(c->bl_lpart, sizeof(double) * (size_t)rows_out * q))) return rc;
  // a sweep may ask for the S / U bytes, |S|, |U| and the radius key straight from the mean epilogue of the constraint
  // (one-constraint models; the masks are allocated by the sweep before it enqueues the posterior)
  // (r03: with the sqrt-free sign tests the fused epilogue saves the separate pass 76 us on config H and costs the GEMM 36;
  // on config B, two workgroups per CU, the two cancel -- "auto" asks for at least four workgroups per CU)
  const bool fuse_wanted = c->fuse_request == 1 || (c->fuse_request == 2 && ((long long)gx * gy * q >= 4ll * c->n_cu || q > 2));   // (several constraints: the separate pass costs more than one constraint's)
  bool fuse = fuse_wanted && q >= 2 && c->maskS.p && c->maskU.p && c->maskS.bytes >= (size_t)cs.n_local &&
              c->maskU.bytes >= (size_t)cs.n_local &&
              (q == 2 || (c->fuseS.bytes >= (size_t)cs.n_local * (q - 1) && c->fuseU.bytes >= (size_t)cs.n_local * (q - 1)));
  const bool colw = fuse && col_words_ok(c, cnt0, nlines);
  PostExtra px;
  memset(&px, 0, sizeof(px));
  px.q = q;
  c->col_active = false;
  c->fuse_rows = 0;
  if (fuse) {
    c->fuse_rows = (int)rows_out * (colw ? 2 : (q - 1));
    px.fstride = q > 2 ? (long long)cs.n_local : 0ll;
    // (room behind the rows for the partials of the objective pass, see sweep_common_front)
    if ((rc = ensure(c->cpart, sizeof(unsigned long long) * kFuseRow * ((size_t)c->fuse_rows + 4 * (size_t)c->n_cu + 64)))) return rc;
    c->cpart_cap = (int)(c->cpart.bytes / (sizeof(unsigned long long) * kFuseRow));
  }
  if (colw) {
    if ((rc = col_words_prepare(c, cnt0, nlines, &px.cb))) return rc;
    px.lean = c->col_lean;
    c->col_active = true;
    c->col_forked = true;
    fuse = false;                      // (no byte masks: the words are the classification)
  }
  // (the fused classification counts its sign tests inside the plan's guard band)
  const GuardBand* gb_fused = (c->guard_band && !c->is_shadow && c->bl.band_ready && c->gb.p) ? (const GuardBand*)c->gb.p : nullptr;
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bpost<rbw, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bpost<rbw, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bpost<rbw, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // the K1 stop event rides on the last launch (hipExtLaunchKernel): a separate hipEventRecord behind it is a barrier packet
  // the next kernel waits ~6 us for.  A sweep merges the Lipschitz partials in its own first small kernel (lmax_defer).
  // (column path: the constraint's launch first -- the objective's tiles read its words and its counts of safe candidates per tile)
  for (int part = 0; part < (colw ? 2 : 1); ++part) {
    px.o0 = colw ? 1 - part : 0;
    const bool last = !colw || part == 1;
    auto kpost = !colw ? k_bpost<rbw, 0> : (part == 0 ? k_bpost<rbw, 1> : k_bpost<rbw, 2>);
    hipExtLaunchKernelGGL(kpost, dim3(gx, gy, (unsigned)(colw ? 1 : q)), dim3(256), lds, c->stream, nullptr,
                          (c->lmax_defer && last) ? c->ev[1] : ((colw && part == 0) ? c->ev_col[0] : nullptr), 0,
                          mc, cs, (const double*)c->bl_BtA.p, pl.sBtA, (const double*)c->bl_P0f.p, pl.sP0f, (const double*)c->bl_VA.p,
                          pl.sVA, (const double*)c->bl_SBf.p, pl.sSBf, pl.KB0, pl.KS0, pl.KBm, pl.KSm, pl.KBm2, pl.nrb, pl.ncs0, nlines,
                          (double*)c->mean.p, (double*)c->var.p, (double*)c->bl_lpart.p, (const double*)c->bl_small.p /* xn0 */,
                          fuse ? (uint8_t*)(q > 2 ? c->fuseS.p : c->maskS.p) : (uint8_t*)nullptr, fuse ? (uint8_t*)(q > 2 ? c->fuseU.p : c->maskU.p) : (uint8_t*)nullptr, c->fuse_b,
                          (unsigned long long*)c->cpart.p, c->cpart_cap, gb_fused, (const int*)pl.eff, pl.gtmax, pl.gkey, 0, px);
  }
  if (c->lmax_defer) {
    c->lmax_pending = true;
    c->lmax_per_out = (int)rows_out;
  } else {
    hipExtLaunchKernelGGL(k_lmax_reduce, dim3((unsigned)q), dim3(256), 0, c->stream, nullptr, c->ev[1], 0, (const double*)c->bl_lpart.p,
                          (int)rows_out, (unsigned long long*)c->Lmax.p);
  }
  c->k1_stop_attached = true;
  (void)line0;
  c->gb_active = c->guard_band && !c->is_shadow && c->bl.band_ready;     // (the band came with the plan: bilinear_setup)
  // flops issued on the matrix cores: stage 1 + the four phases of stage 2 (KS0 + 3 KSm k-steps: the axis-0 gradient phase
  // runs on the mean phase's sums; 16 x 16 x 4 steps, 2 flops per multiply-add)
  const double tiles2 = (double)pl.nrb * pl.ncs0, tiles1 = (double)pl.nrb * pl.KB0;
  c->last_k1_flops = (double)q * 2.0 * 1024.0 * (4.0 * tiles1 * pl.KB1 + tiles2 * (pl.KS0 + 3 * pl.KSm));
  if (pl.eff) {
    // Chebyshev core: the counts the kernels actually run to (k_cheb_trunc; copied to the pinned block when the plan was built --
    // they have arrived long before a sweep's result is read: a plan build is followed by the sweep's own synchronisation
    // before anyone asks for the profile).  Until then the upper bound above stands.
    const int* he = (const int*)(c->h_back + 5376);
    double f = 0.0;
    bool ok = true;
    for (int o = 0; o < q; ++o) {
      const int ks = he[4 * o], kb0 = he[4 * o + 1], kb1 = he[4 * o + 2];
      if (ks < 1 || ks > pl.KS0 || kb0 < 1 || kb0 > pl.KB0 || kb1 < 1 || kb1 > pl.KB1) { ok = false; break; }
      // (the gradient phases run on the few tiles that can hold the maximum: not counted)
      f += 2.0 * 1024.0 * (4.0 * (double)pl.nrb * kb0 * kb1 + tiles2 * (ks + (pl.gtmax ? 1 : 3) * pl.KSm));
    }
    if (ok) c->last_k1_flops = f;
  }
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

}  // namespace sbo
