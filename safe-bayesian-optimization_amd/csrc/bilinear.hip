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
#include <hip/hip_ext.h>
#include "bilinear_host.hpp"
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
#pragma unroll
  for (int s = 0; s < S; ++s) {
    if (cs0 + s >= ncs) continue;
    if (OMODE == 0) {
      double* blk = out + (size_t)o * out_stride_o + ((size_t)rb * ncs + (cs0 + s)) * 256;
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
                                                 size_t out_stride_o) {
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
    const bool more = kb + 1 < KB;
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
  for (int kb = 0; kb < KB; kb += 2) {
    step(kb, Bs[0], Bs[1], As[0][wave], As[1][wave]);
    if (kb + 1 < KB) step(kb + 1, Bs[1], Bs[0], As[1][wave], As[0][wave]);
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

// ---- per-model table builders (device side of bilinear_setup) -------------------------------------------------
// Z_j,(p,s) = U0_jp U1_js as B fragments [ncsR][KBn * 4][64] (k = observation j, column c = p r1 + s)
__global__ __launch_bounds__(256) void k_bl_zf(const double* __restrict__ U0, const double* __restrict__ U1, int n, int KBn, int r0,
                                               int r1, int ncsR, double* __restrict__ Zf) {
  const long long total = (long long)ncsR * KBn * 4 * 64;
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
// T4qq^T as B fragments over the columns k0: [KB0][KB1 * 4][64], element (k1, k0) = scale/2 (G[(p,s),(p',s')] + G[(p',s),(p,s')])
__global__ __launch_bounds__(256) void k_bl_t4f(const double* __restrict__ G, long long ldg, int r1, const int* __restrict__ map0,
                                                int K0, const int* __restrict__ map1, int K1, double scale, int KB0, int KB1,
                                                double* __restrict__ T4f) {
  const long long total = (long long)KB0 * KB1 * 4 * 64;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int l = (int)(i & 63);
    const long long fr = i >> 6;
    const int ks = (int)(fr % (KB1 * 4)), cs = (int)(fr / (KB1 * 4));
    const int k1 = (ks >> 2) * 16 + MM<double>::jslot(ks & 3, l >> 4);
    const int k0 = cs * 16 + (l & 15);
    double v = 0.0;
    if (k0 < K0 && k1 < K1) {
      const int p = map0[2 * k0], pp = map0[2 * k0 + 1], s1 = map1[2 * k1], ss = map1[2 * k1 + 1];
      v = scale * 0.5 * (G[(size_t)(p * r1 + s1) * ldg + (pp * r1 + ss)] + G[(size_t)(pp * r1 + s1) * ldg + (p * r1 + ss)]);
    }
    T4f[i] = v;
  }
}
// pair products of an axis table S [r][count]: P[(p <= p')][x] = w S_p S_p' (w = 1 diagonal, 2 off it)
//   FRAG 1: as B fragments [ncs][KB * 4][64] (columns = positions);  FRAG 0: transposed as A images [nrb][KB][256] (rows = positions)
template <int FRAG>
__global__ __launch_bounds__(256) void k_bl_pairs(const double* __restrict__ Stab, long long count, const int* __restrict__ map,
                                                  int K, int KB, int nblk, double* __restrict__ out) {
  const long long total = (long long)nblk * KB * 256;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int k;
    long long x;
    if (FRAG) {
      const int l = (int)(i & 63);
      const long long fr = i >> 6;
      const int ks = (int)(fr % (KB * 4));
      k = (ks >> 2) * 16 + MM<double>::jslot(ks & 3, l >> 4);
      x = (fr / (KB * 4)) * 16 + (l & 15);
    } else {
      // invert pack_pos(r, slot, kk) = kk * 64 + (slot * 4 + (r & 3)) * 4 + (r >> 2)
      const int e = (int)(i & 255);
      const long long blk = i >> 8;
      const int kk = e >> 6, rem = e & 63, slot = rem >> 4, r = ((rem >> 2) & 3) + 4 * (rem & 3);
      k = (int)(blk % KB) * 16 + MM<double>::jslot(kk, slot);
      x = (blk / KB) * 16 + r;
    }
    double v = 0.0;
    if (k < K && x < count) {
      const int p = map[2 * k], pp = map[2 * k + 1];
      v = (p == pp ? 1.0 : 2.0) * Stab[(size_t)p * count + x] * Stab[(size_t)pp * count + x];
    }
    out[i] = v;
  }
}

// Axis table S[p][i] = sig_p sum_c Vs[p][c] T_c(xi_i) from the Chebyshev series of the basis (the arithmetic of
// bl::axis_basis' own tabulation: three-term recurrence, sum in ascending c, one thread per entry)
__global__ __launch_bounds__(256) void k_bl_stab(const double* __restrict__ Vs, const double* __restrict__ sig,
                                                 const double* __restrict__ xn, double a, double b, int rc, int r, long long count,
                                                 double* __restrict__ S) {
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

// ---- mean-phase operands on the device (the host only supplies the bases and beta) ---------------------------------
// Mb[b][p r1 + s] = scale sum_j beta_b[j] U0_jp U1_js  (bilinear forms of the mean and of its two gradient sums)
__global__ __launch_bounds__(256) void k_bl_mb(const double* __restrict__ U0, const double* __restrict__ U1,
                                               const double* __restrict__ beta, int n, int r0, int r1, int nbeta, double scale,
                                               double* __restrict__ Mb) {
  const int total = nbeta * r0 * r1;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int s_ = i % r1, p = (i / r1) % r0, b = i / (r0 * r1);
    const double *u0 = U0 + (size_t)p * n, *u1 = U1 + (size_t)s_ * n, *be = beta + (size_t)b * n;
    double acc = 0.0;
    for (int j = 0; j < n; ++j) acc += be[j] * u0[j] * u1[j];
    Mb[i] = scale * acc;
  }
}
// Vb[b][p][line] = sum_s Mb[b][p, s] S1[s][line]
__global__ __launch_bounds__(256) void k_bl_vb(const double* __restrict__ Mb, const double* __restrict__ S1, int r0, int r1, int r0u,
                                               long long nlines, int nbeta, double* __restrict__ Vb) {
  const long long total = (long long)nbeta * r0 * nlines;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long l = i % nlines;
    const int p = (int)((i / nlines) % r0), b = (int)(i / (nlines * r0));
    double acc = 0.0;
    for (int s_ = 0; s_ < r1; ++s_) acc += Mb[((size_t)b * r0 + p) * r1 + s_] * S1[(size_t)s_ * nlines + l];
    Vb[((size_t)b * r0u + p) * nlines + l] = acc;
  }
}
// A images of the mean phases, three sets:  V0 (K = 16 KBm) | [V1; V0] (K = 16 KBm2, V0 from k = r0p) | V1x = V2 - xn1 V0
__global__ __launch_bounds__(256) void k_bl_va(const double* __restrict__ Vb, const double* __restrict__ xn1, int nrb, int KBm,
                                               int KBm2, int r0, int r0p, int r0u, long long nlines, double* __restrict__ out) {
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
__global__ __launch_bounds__(256) void k_bl_sbf(const double* __restrict__ S0, const double* __restrict__ xn0, int ncs0, int KBm,
                                                int KBm2, int r0, int r0p, long long cnt0, double* __restrict__ out) {
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

struct PostCtx {
  double* lds;
  int tid, lane, wave, rb0, cs0, nrb, ncs;
  int st_rb, st_cs, st_off, a_lo, b_st, a_rd;
  unsigned int ucnt0;
  long long nlines;
  bool full;          // the workgroup's 128 x 128 tile lies inside the grid: epilogues skip their bounds tests
};

// One GEMM phase of k_bpost on the workgroup's 128 x 128 tile: A images / B fragments of `KB` k-blocks per row block /
// strip, `KS` k-steps run.  PH selects the epilogue: 0 variance, 1 mean, 2 / 3 gradient component of axis 0 / 1.
// The accumulators belong to the caller: phase 2 does not start from zero but from phase 1's sums scaled by -xn0 of the
// candidate's column, g0 = V1 . S0 - xn0 (V0 . S0) -- six k-steps instead of twelve for the stacked [V1; V0] operand.
template <int PH>
__device__ __forceinline__ void post_phase(const PostCtx& cx, const double* __restrict__ A, const double* __restrict__ B, int KB,
                                           int KS, double* __restrict__ outp, double c0, double c1, double c2, double& gmax,
                                           d4_t (&acc)[2][8], const double* __restrict__ xn0) {
  const double* Ap = A + (size_t)cx.st_rb * KB * 256 + cx.st_off;
  const double* Bp = B + (size_t)cx.st_cs * KB * 256 + cx.st_off;
  const int nkb = (KS + 3) >> 2;
  double* const lds = cx.lds;
  auto stage = [&](double* buf, const d4_t& r0, const d4_t& r1, const d4_t& q0, const d4_t& q1) {
    *reinterpret_cast<d4_t*>(buf + cx.a_lo) = d4_t{r0[0], r0[1], r1[0], r1[1]};
    *reinterpret_cast<d4_t*>(buf + cx.a_lo + 32) = d4_t{r0[2], r0[3], r1[2], r1[3]};
    *reinterpret_cast<d4_t*>(buf + cx.b_st) = q0;
    *reinterpret_cast<d4_t*>(buf + cx.b_st + 4) = q1;
  };
  if (PH == 2) {
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
      const unsigned int x = (unsigned int)(cx.cs0 + s2) * 16u + (cx.lane & 15);
      const double f = x < cx.ucnt0 ? -xn0[x] : 0.0;
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i][s2] = d4_t{acc[i][s2][0] * f, acc[i][s2][1] * f, acc[i][s2][2] * f, acc[i][s2][3] * f};
    }
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) acc[i][s2] = d4_t{0.0, 0.0, 0.0, 0.0};
  }
  d4_t ra0 = *reinterpret_cast<const d4_t*>(Ap), ra1 = *reinterpret_cast<const d4_t*>(Ap + 4);
  d4_t rb0v = *reinterpret_cast<const d4_t*>(Bp), rb1v = *reinterpret_cast<const d4_t*>(Bp + 4);
  __syncthreads();                             // the previous phase has finished reading the buffers
  stage(lds, ra0, ra1, rb0v, rb1v);
  __syncthreads();
#pragma unroll 1
  for (int kb = 0; kb < nkb; ++kb) {
    const int cur = kb & 1;
    if (kb + 1 < nkb) {
      ra0 = *reinterpret_cast<const d4_t*>(Ap + (size_t)(kb + 1) * 256);
      ra1 = *reinterpret_cast<const d4_t*>(Ap + (size_t)(kb + 1) * 256 + 4);
      rb0v = *reinterpret_cast<const d4_t*>(Bp + (size_t)(kb + 1) * 256);
      rb1v = *reinterpret_cast<const d4_t*>(Bp + (size_t)(kb + 1) * 256 + 4);
    }
    const double* LA = lds + cur * 4096 + (2 * cx.wave) * 256 + cx.a_rd;
    const double* LB = lds + cur * 4096 + 2048 + cx.lane;
    const int kkn = KS - kb * 4 < 4 ? KS - kb * 4 : 4;
#pragma unroll 1
    for (int kk = 0; kk < kkn; ++kk) {
      const d2_t l0 = *reinterpret_cast<const d2_t*>(LA + kk * 64), h0 = *reinterpret_cast<const d2_t*>(LA + kk * 64 + 32);
      const d2_t l1 = *reinterpret_cast<const d2_t*>(LA + 256 + kk * 64), h1 = *reinterpret_cast<const d2_t*>(LA + 256 + kk * 64 + 32);
      const d4_t a0 = d4_t{l0[0], l0[1], h0[0], h0[1]}, a1 = d4_t{l1[0], l1[1], h1[0], h1[1]};
#pragma unroll
      for (int s2 = 0; s2 < 8; ++s2) {
        const double b = LB[(s2 * 4 + kk) * 64];
        acc[0][s2] = MM<double>::mfma(a0, b, acc[0][s2]);
        acc[1][s2] = MM<double>::mfma(a1, b, acc[1][s2]);
      }
    }
    if (kb + 1 < nkb) stage(lds + (cur ^ 1) * 4096, ra0, ra1, rb0v, rb1v);
    __syncthreads();
  }
  // epilogue: accumulator element t of lane l is row 4 t + (l >> 4), column l & 15 of its 16 x 16 tile
  const unsigned int col_in = cx.lane & 15, row_in = cx.lane >> 4;
  if (cx.full) {
    // interior tile: no bounds tests, one pointer per row, the eight strips at immediate offsets.  The matrix cores
    // share the f64 VALU datapath, so every instruction saved here is matrix time.
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const unsigned int line = (unsigned int)(cx.rb0 + 2 * cx.wave + i) * 16u + 4u * t + row_in;
        double* const rowp = outp + ((size_t)line * cx.ucnt0 + (unsigned int)cx.cs0 * 16u + col_in);
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
            double ga = c0 * v;
            ga = ga < 0 ? -ga : ga;
            gmax = ga > gmax ? ga : gmax;
          }
        }
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rb = cx.rb0 + 2 * cx.wave + i;
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
          rowp[x0] = (c0 + v) * c1 + c2;                                    // :342, :346
        } else {
          // component of the gradient of the un-normalised mean (analytic jax.grad(self.mean), SafeOpt.py:68-71)
          double ga = c0 * v;
          ga = ga < 0 ? -ga : ga;
          gmax = ga > gmax ? ga : gmax;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256, 2) void k_bpost(const ModelConst mc, const CandSpec cs, const double* __restrict__ BtA, size_t sBtA,
                                                  const double* __restrict__ P0f, size_t sP0f, const double* __restrict__ VA,
                                                  size_t sVA, const double* __restrict__ SBf, size_t sSBf, int KB0, int KS0, int KBm,
                                                  int KSm, int KBm2, int nrb, int ncs, long long nlines, double* __restrict__ mean_out,
                                                  double* __restrict__ var_out, double* __restrict__ Lpart,
                                                  const double* __restrict__ xn0) {
  extern __shared__ double lds[];               // [2][A: 8 x 256 | B: 8 x 256]
  const int o = blockIdx.z;
  PostCtx cx;
  cx.lds = lds;
  cx.tid = threadIdx.x; cx.lane = cx.tid & 63; cx.wave = cx.tid >> 6;
  cx.rb0 = blockIdx.y * 8; cx.cs0 = blockIdx.x * 8; cx.nrb = nrb; cx.ncs = ncs;
  cx.ucnt0 = (unsigned int)cs.count[0];
  cx.nlines = nlines;
  cx.full = (long long)(cx.rb0 + 8) * 16 <= nlines && (long long)(cx.cs0 + 8) * 16 <= cs.count[0];
  // staging role of this thread: 64 bytes of one A image and 64 bytes of one B strip per k-block.
  // LDS image of an A block: per k-step the 16 lane-chunks are split into their first and second 16 bytes
  // ([16 x 16 B][16 x 16 B]) so that both ds_read_b128 of a fragment load touch 256 contiguous bytes (no bank conflicts)
  const int st_i = cx.tid >> 5, st_j = cx.tid & 31;
  cx.st_off = st_j * 8;
  cx.st_rb = cx.rb0 + st_i < nrb ? cx.rb0 + st_i : nrb - 1;
  cx.st_cs = cx.cs0 + st_i < ncs ? cx.cs0 + st_i : ncs - 1;
  cx.a_lo = st_i * 256 + (st_j >> 3) * 64 + (st_j & 7) * 4;
  cx.b_st = 2048 + st_i * 256 + cx.st_off;
  cx.a_rd = (((cx.lane >> 4) << 2) + (cx.lane & 3)) * 2;
  // per-output operands; VA holds [V0 | V1;V0 | V1x] as three image sets, SBf holds [S0 | S0;-xn0 S0] as two fragment sets
  const double* VAo = VA + (size_t)o * sVA;
  const double* SBo = SBf + (size_t)o * sSBf;
  const double sf2 = mc.sf2[o], ystd = mc.Y_std[o];
  double* const vo = var_out + (size_t)o * cs.n_local;
  double* const mo = mean_out + (size_t)o * cs.n_local;
  double gmax = 0.0;
  d4_t acc[2][8];
  post_phase<0>(cx, BtA + (size_t)o * sBtA, P0f + (size_t)o * sP0f, KB0, KS0, vo, sf2, ystd * ystd, 0.0, gmax, acc, xn0);
  post_phase<1>(cx, VAo, SBo, KBm, KSm, mo, mc.mp[o], ystd, mc.Y_mean[o], gmax, acc, xn0);
  // phase 2 continues on phase 1's sums: only the V1 half (the first KSm k-steps) of the stacked operands is run
  post_phase<2>(cx, VAo + (size_t)nrb * KBm * 256, SBo + (size_t)ncs * KBm * 256, KBm2, KSm, nullptr,
                ystd * mc.inv_ell[o][0] * mc.X_rstd[0], 0.0, 0.0, gmax, acc, xn0);
  post_phase<3>(cx, VAo + (size_t)nrb * (KBm + KBm2) * 256, SBo, KBm, KSm, nullptr, ystd * mc.inv_ell[o][1] * mc.X_rstd[1], 0.0, 0.0,
                gmax, acc, xn0);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double other = __shfl_xor(gmax, off);
    gmax = other > gmax ? other : gmax;
  }
  // one plain store per wave, merged by k_lmax_reduce: every workgroup of this launch is resident at once and ends at
  // the same time, so atomics on the q keys would queue up in L2 as the kernel's tail
  if (cx.lane == 0)
    Lpart[(((size_t)o * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + cx.wave] = gmax;
}

// Lipschitz keys of a K1b launch: Lmax[o] = max of the per-wave partials (values >= 0, so the bit pattern orders them)
__global__ __launch_bounds__(256) void k_lmax_reduce(const double* __restrict__ Lpart, int per_out, unsigned long long* __restrict__ Lmax) {
  __shared__ double sh[4];
  const int o = blockIdx.x;
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
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = g;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) g = sh[w] > g ? sh[w] : g;
    Lmax[o] = (unsigned long long)__double_as_longlong(g);
  }
}

// ---- plan ------------------------------------------------------------------------------------------------
static void grid_axis_positions(const sbo_ctx* c, int a, long long i0, long long cnt, std::vector<double>& xn) {
  const CandSpec& cs = c->cs;
  xn.resize((size_t)cnt);
  for (long long k = 0; k < cnt; ++k) {
    const long long i = i0 + k, tot = cs.count[a];
    const double x = (i == tot - 1 && tot > 1) ? cs.hi[a] : cs.lo[a] + (double)i * cs.step[a];
    xn[(size_t)k] = (x - c->mc.X_mean[a]) / c->mc.X_std[a];                                       // GP_Safe.py:326
  }
}

bool bilinear_applicable(const sbo_ctx* c) {
  const CandSpec& cs = c->cs;
  if (!c->bilinear || c->dtype != SBO_F64 || cs.kind != 1 || cs.d != 2 || c->mc.d != 2) return false;
  const long long cnt0 = cs.count[0];
  if (cs.n_local <= 0 || cs.first % cnt0 != 0 || cs.n_local % cnt0 != 0) return false;
  // the bases pay off (and the interpolation interval is meaningful) only on real grids
  return cnt0 >= 64 && cs.count[1] >= 64 && cs.n_local / cnt0 >= 16 && c->h_alpha.size() == (size_t)c->mc.q * c->mc.npad;
}

// Builds the device tables for the current (model, candidates).  Returns SBO_OK with plan.usable = false when the bases do
// not qualify (rank / interpolation limits): the caller then keeps the separable-table kernel.
int bilinear_setup(sbo_ctx* c) {
  BilinearPlan& pl = c->bl;
  pl.valid = true;
  pl.usable = false;
  pl.setup_ms = 0.0;
  const auto t_begin = std::chrono::steady_clock::now();
  const bool timing = getenv("SBO_BL_TIMING") != nullptr;
  auto lap = [&](const char* what) {
    if (timing) fprintf(stderr, "[K1b setup] %-12s %8.2f ms\n", what,
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  };
  const ModelConst& mc = c->mc;
  const CandSpec& cs = c->cs;
  const int n = mc.n, q = mc.q, d = 2, NB = 1 + d;
  const long long cnt0 = cs.count[0], nlines = cs.n_local / cnt0, line0 = cs.first / cnt0;
  std::vector<double> xn0, xn1_all, xn1;
  grid_axis_positions(c, 0, 0, cnt0, xn0);
  // the axis-1 basis is built on the interval of the WHOLE axis, so that every rank of a sharded grid uses the same
  // functions; only the local lines are tabulated
  grid_axis_positions(c, 1, 0, cs.count[1], xn1_all);
  std::vector<bl::AxisBasis> b0(q), b1(q);
  {
    // the 2 q bases are independent: one host thread each
    std::vector<char> ok(2 * q, 0);
    bl::parallel_ranges(2 * q, [&](int lo_, int hi_) {
      std::vector<double> col(n);
      for (int t = lo_; t < hi_; ++t) {
        const int o = t / 2, a = t % 2;
        for (int j = 0; j < n; ++j) col[j] = c->h_Xnorm[(size_t)j * mc.d + a] * mc.vinv[o][a];   // GP_Safe.py:115
        bl::AxisBasis& b = a == 0 ? b0[o] : b1[o];
        const std::vector<double>& xs = a == 0 ? xn0 : xn1_all;
        ok[t] = bl::axis_basis(n, col.data(), mc.vinv[o][a], xs.data(), (int)xs.size(), b, /*tabulate=*/false) ? 1 : 0;
      }
    });
    for (int t = 0; t < 2 * q; ++t)
      if (!ok[t]) return SBO_OK;
  }
  lap("bases");
  int r0u = 0, K0 = 0, K1 = 0;
  for (int o = 0; o < q; ++o) {
    r0u = std::max(r0u, b0[o].r);
    K0 = std::max(K0, bl::pair_count(b0[o].r));
    K1 = std::max(K1, bl::pair_count(b1[o].r));
  }
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
  pl.sP0f = (size_t)ncs0 * KB0 * 4 * 64;
  pl.sP1A = (size_t)nrb * KB1 * 256;
  pl.sT4f = (size_t)KB0 * KB1 * 4 * 64;
  pl.sBtA = (size_t)nrb * KB0 * 256;
  // mean phases: K = r0p (basis size rounded to whole k-steps); the axis-0 gradient phase concatenates two such operands
  const int r0p = (r0u + 3) / 4 * 4;
  const int KBm = (r0p + 15) / 16, KBm2 = (2 * r0p + 15) / 16;
  pl.KBm = KBm;
  pl.KBm2 = KBm2;
  pl.KSm = r0p / 4;
  pl.KS0 = (K0 + 3) / 4;
  pl.sVA = (size_t)nrb * (2 * KBm + KBm2) * 256;     // image sets  V0 | [V1; V0] | V1x
  pl.sSBf = (size_t)ncs0 * (KBm + KBm2) * 256;      // fragment sets  S0 | [S0; -xn0 S0]
  // Everything below the bases runs on the device, all outputs back to back on the stream and one synchronisation at
  // the end: per output the bases / tables / beta go up (a few hundred KB), then Z -> C = M Z -> G = C^T C -> T4, the
  // pair tables of both axes, and the mean-phase operands Mb -> Vb -> A images / B fragments.
  int rc;
  if ((rc = ensure(c->bl_P0f, sizeof(double) * pl.sP0f * q))) return rc;
  if ((rc = ensure(c->bl_P1A, sizeof(double) * pl.sP1A * q))) return rc;
  if ((rc = ensure(c->bl_T4f, sizeof(double) * pl.sT4f * q))) return rc;
  if ((rc = ensure(c->bl_SBf, sizeof(double) * pl.sSBf * q))) return rc;     // mean-phase B fragments
  if ((rc = ensure(c->bl_VA, sizeof(double) * pl.sVA * q))) return rc;      // mean-phase A images
  if ((rc = ensure(c->bl_BtA, sizeof(double) * pl.sBtA * q))) return rc;
  const int KBn = mc.npad / 16;
  // host staging that must outlive the asynchronous uploads: one set per output
  // Layout of bl_small: [uploaded: xn0 | xn1 (local lines) | per output U0 U1 beta Vs0 sg0 Vs1 sg1 | pair maps (ints)]
  // [device-made: per output S0 S1 Mb Vb].  The uploaded part is assembled in one pinned staging buffer and goes up
  // as a single copy (a dozen small copies from pageable memory cost ~30 us each).
  std::vector<std::vector<int>> maps0(q), maps1(q);
  struct Region { size_t U0, U1, beta, Vs0, sg0, Vs1, sg1, map0, map1, S0, S1, Mb, Vb; };
  std::vector<Region> reg(q);
  size_t ndbl = (size_t)cnt0 + (size_t)nlines, nint = 0, work_max = 0;
  for (int o = 0; o < q; ++o) {
    const int r0 = b0[o].r, r1 = b1[o].r;
    Region& g = reg[o];
    g.U0 = ndbl;
    g.U1 = g.U0 + (size_t)n * r0;
    g.beta = g.U1 + (size_t)n * r1;
    g.Vs0 = g.beta + (size_t)NB * n;
    g.sg0 = g.Vs0 + b0[o].Vs.size();
    g.Vs1 = g.sg0 + (size_t)r0;
    g.sg1 = g.Vs1 + b1[o].Vs.size();
    ndbl = g.sg1 + (size_t)r1;
    bl::pair_map(r0, maps0[o]);
    bl::pair_map(r1, maps1[o]);
    g.map0 = nint;
    g.map1 = nint + maps0[o].size();
    nint = g.map1 + maps1[o].size();
    const size_t ncsR = ((size_t)r0 * r1 + 15) / 16;
    work_max = std::max(work_max, 3 * ncsR * KBn * 256 + (ncsR * 16) * (ncsR * 16));
  }
  const size_t nint_pad = (nint + 1) / 2 * 2;               // keep the doubles behind the ints 8-byte aligned
  const size_t up_bytes = sizeof(double) * ndbl + sizeof(int) * nint_pad;
  size_t ndev = 0;                                          // device-made part, in doubles behind the uploaded bytes
  for (int o = 0; o < q; ++o) {
    const int r0 = b0[o].r, r1 = b1[o].r;
    Region& g = reg[o];
    g.S0 = ndev;
    g.S1 = g.S0 + (size_t)r0 * cnt0;
    g.Mb = g.S1 + (size_t)r1 * nlines;
    g.Vb = g.Mb + (size_t)NB * r0 * r1;
    ndev = g.Vb + (size_t)NB * r0u * nlines;
  }
  if ((rc = ensure(c->bl_small, up_bytes + sizeof(double) * ndev))) return rc;
  if ((rc = ensure(c->bl_work, sizeof(double) * work_max))) return rc;
  if (c->h_stage_bytes < up_bytes) {
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr;
    c->h_stage_bytes = 0;
    const size_t want = up_bytes + up_bytes / 2 + 4096;
    if (hipHostMalloc(&c->h_stage, want, hipHostMallocDefault) != hipSuccess) return fail(SBO_E_HIP, "hipHostMalloc (K1b staging)");
    c->h_stage_bytes = want;
  }
  double* hsm = (double*)c->h_stage;
  int* hints = (int*)(hsm + ndbl);
  double* dsm = (double*)c->bl_small.p;
  const int* dints = (const int*)(dsm + ndbl);
  double* ddev = (double*)((char*)c->bl_small.p + up_bytes);
  memcpy(hsm, xn0.data(), sizeof(double) * cnt0);
  memcpy(hsm + cnt0, &xn1_all[(size_t)line0], sizeof(double) * nlines);
  for (int o = 0; o < q; ++o) {
    const Region& g = reg[o];
    memcpy(hsm + g.U0, b0[o].U.data(), sizeof(double) * b0[o].U.size());
    memcpy(hsm + g.U1, b1[o].U.data(), sizeof(double) * b1[o].U.size());
    double* beta = hsm + g.beta;
    for (int j = 0; j < n; ++j) {
      const double al = c->h_alpha[(size_t)o * mc.npad + j];
      beta[j] = al;
      beta[(size_t)n + j] = al * c->h_Xnorm[(size_t)j * mc.d + 0];
      beta[(size_t)2 * n + j] = al * c->h_Xnorm[(size_t)j * mc.d + 1];
    }
    memcpy(hsm + g.Vs0, b0[o].Vs.data(), sizeof(double) * b0[o].Vs.size());
    memcpy(hsm + g.sg0, b0[o].sig.data(), sizeof(double) * b0[o].sig.size());
    memcpy(hsm + g.Vs1, b1[o].Vs.data(), sizeof(double) * b1[o].Vs.size());
    memcpy(hsm + g.sg1, b1[o].sig.data(), sizeof(double) * b1[o].sig.size());
    memcpy(hints + g.map0, maps0[o].data(), sizeof(int) * maps0[o].size());
    memcpy(hints + g.map1, maps1[o].data(), sizeof(int) * maps1[o].size());
  }
  SBO_HIP(hipMemcpyAsync(dsm, hsm, up_bytes, hipMemcpyHostToDevice, c->stream));
  auto blocks = [](size_t total) { return dim3((unsigned)std::min<size_t>((total + 255) / 256, 1u << 16)); };
  for (int o = 0; o < q; ++o) {
    const int r0 = b0[o].r, r1 = b1[o].r, k0n = bl::pair_count(r0), k1n = bl::pair_count(r1);
    const double sf2 = mc.sf2[o];
    const Region& g = reg[o];
    const int R = r0 * r1, ncsR = (R + 15) / 16;
    const size_t nZf = (size_t)ncsR * KBn * 256;              // fragments of Z, of C, images of C^T: same size
    const size_t ldg = (size_t)ncsR * 16;
    double *dU0 = dsm + g.U0, *dU1 = dsm + g.U1, *dbeta = dsm + g.beta;
    double *dS0 = ddev + g.S0, *dS1 = ddev + g.S1, *dMb = ddev + g.Mb, *dVb = ddev + g.Vb;
    const int *dmap0 = dints + g.map0, *dmap1 = dints + g.map1;
    double* Zf = (double*)c->bl_work.p;
    double* Cf = Zf + nZf;
    double* CtA = Cf + nZf;
    double* G = CtA + nZf;
    // axis tables from the bases' Chebyshev series: all positions of axis 0, the local lines of axis 1
    hipLaunchKernelGGL(k_bl_stab, blocks((size_t)r0 * cnt0), dim3(256), 0, c->stream, (const double*)(dsm + g.Vs0),
                       (const double*)(dsm + g.sg0), (const double*)dsm, b0[o].a, b0[o].b, b0[o].rc, r0, cnt0, dS0);
    hipLaunchKernelGGL(k_bl_stab, blocks((size_t)r1 * nlines), dim3(256), 0, c->stream, (const double*)(dsm + g.Vs1),
                       (const double*)(dsm + g.sg1), (const double*)(dsm + cnt0), b1[o].a, b1[o].b, b1[o].rc, r1, nlines, dS1);
    hipLaunchKernelGGL(k_bl_zf, blocks(nZf), dim3(256), 0, c->stream, (const double*)dU0, (const double*)dU1, n, KBn, r0, r1, ncsR, Zf);
    // C = M Z with the model's packed triangular factor; written as fragments (k = observation) and as images of C^T
    hipLaunchKernelGGL((k_bgemm<4, 1, 1>), dim3((unsigned)((ncsR + 3) / 4), (unsigned)((KBn + 3) / 4), 1), dim3(256), 0, c->stream,
                       (const double*)c->Fpk.p + (size_t)o * c->fpk_stride, (size_t)0, (const double*)Zf, (size_t)0, KBn, KBn, ncsR,
                       Cf, (size_t)0, CtA, 0ll);
    // G = C^T C  (R x R, row-major)
    hipLaunchKernelGGL((k_bgemm<4, 0, 2>), dim3((unsigned)((ncsR + 3) / 4), (unsigned)((ncsR + 3) / 4), 1), dim3(256), 0, c->stream,
                       (const double*)CtA, (size_t)0, (const double*)Cf, (size_t)0, KBn, ncsR, ncsR, G, (size_t)0, (double*)nullptr,
                       (long long)ldg);
    hipLaunchKernelGGL(k_bl_t4f, blocks(pl.sT4f), dim3(256), 0, c->stream, (const double*)G, (long long)ldg, r1, (const int*)dmap0, k0n,
                       (const int*)dmap1, k1n, sf2 * sf2, KB0, KB1, (double*)c->bl_T4f.p + pl.sT4f * o);
    hipLaunchKernelGGL((k_bl_pairs<1>), blocks(pl.sP0f), dim3(256), 0, c->stream, (const double*)dS0, cnt0, (const int*)dmap0, k0n, KB0,
                       ncs0, (double*)c->bl_P0f.p + pl.sP0f * o);
    hipLaunchKernelGGL((k_bl_pairs<0>), blocks(pl.sP1A), dim3(256), 0, c->stream, (const double*)dS1, nlines, (const int*)dmap1, k1n, KB1,
                       nrb, (double*)c->bl_P1A.p + pl.sP1A * o);
    // mean phases: Mb (forms of alpha, alpha Xn_0, alpha Xn_1) -> Vb = Mb S1 -> A images [V0 | V1;V0 | V1x], B fragments
    // [S0 | S0;-xn0 S0]
    hipLaunchKernelGGL(k_bl_mb, blocks((size_t)NB * R), dim3(256), 0, c->stream, (const double*)dU0, (const double*)dU1,
                       (const double*)dbeta, n, r0, r1, NB, sf2, dMb);
    hipLaunchKernelGGL(k_bl_vb, blocks((size_t)NB * r0 * nlines), dim3(256), 0, c->stream, (const double*)dMb, (const double*)dS1, r0,
                       r1, r0u, nlines, NB, dVb);
    hipLaunchKernelGGL(k_bl_va, blocks(pl.sVA), dim3(256), 0, c->stream, (const double*)dVb, (const double*)(dsm + cnt0), nrb, KBm, KBm2,
                       r0, r0p, r0u, nlines, (double*)c->bl_VA.p + pl.sVA * o);
    hipLaunchKernelGGL(k_bl_sbf, blocks(pl.sSBf), dim3(256), 0, c->stream, (const double*)dS0, (const double*)dsm, ncs0, KBm, KBm2, r0,
                       r0p, cnt0, (double*)c->bl_SBf.p + pl.sSBf * o);
    SBO_HIP(hipGetLastError());
    pl.r0[o] = r0;
    pl.r1[o] = r1;
  }
  lap("enqueue");
  SBO_HIP(hipStreamSynchronize(c->stream));      // the host staging vectors go out of scope
  lap("upload");
  pl.usable = true;
  pl.setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
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
                     (double*)c->bl_BtA.p, pl.sBtA);
  // stage 2 (fused): variance, mean, Lipschitz keys
  const size_t lds = sizeof(double) * 2 * 4096;
  const unsigned gx = (unsigned)((pl.ncs0 + 7) / 8), gy = (unsigned)((pl.nrb + 7) / 8);
  int rc;
  if ((rc = ensure(c->bl_lpart, sizeof(double) * 4 * (size_t)gx * gy * q))) return rc;
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bpost), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_bpost, dim3(gx, gy, (unsigned)q), dim3(256), lds, c->stream,
                     mc, cs, (const double*)c->bl_BtA.p, pl.sBtA, (const double*)c->bl_P0f.p, pl.sP0f, (const double*)c->bl_VA.p,
                     pl.sVA, (const double*)c->bl_SBf.p, pl.sSBf, pl.KB0, pl.KS0, pl.KBm, pl.KSm, pl.KBm2, pl.nrb, pl.ncs0, nlines,
                     (double*)c->mean.p, (double*)c->var.p, (double*)c->bl_lpart.p, (const double*)c->bl_small.p /* xn0 */);
  // the K1 stop event rides on this launch (hipExtLaunchKernel): a separate hipEventRecord behind it is a barrier packet
  // the next kernel waits ~6 us for
  hipExtLaunchKernelGGL(k_lmax_reduce, dim3((unsigned)q), dim3(256), 0, c->stream, nullptr, c->ev[1], 0, (const double*)c->bl_lpart.p,
                        (int)(4 * gx * gy), (unsigned long long*)c->Lmax.p);
  c->k1_stop_attached = true;
  (void)line0;
  // flops issued on the matrix cores: stage 1 + the four phases of stage 2 (KS0 + 3 KSm k-steps: the axis-0 gradient phase
  // runs on the mean phase's sums; 16 x 16 x 4 steps, 2 flops per multiply-add)
  const double tiles2 = (double)pl.nrb * pl.ncs0, tiles1 = (double)pl.nrb * pl.KB0;
  c->last_k1_flops = (double)q * 2.0 * 1024.0 * (4.0 * tiles1 * pl.KB1 + tiles2 * (pl.KS0 + 3 * pl.KSm));
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

}  // namespace sbo
