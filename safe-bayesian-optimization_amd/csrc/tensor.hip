// tensor.hip -- K1t: the posterior on fp64 tensor grids of three or four axes by Chebyshev interpolation (r03).
//
// The O(n^2)-per-candidate kernel K1g evaluates GP_inference (models/GP_Safe.py:310-352) at every grid point: 216 ms for the
// 268 M candidates of BASELINE.json configs[3] (128^4, n = 128).  But with the separable RBF-ARD kernel the posterior mean, the
// quadratic form of the variance and the mean's gradient are analytic functions of the candidate whose variation along an axis
// is set by the length scale, not by the grid: on [lo_a, hi_a] each of them is, to rounding, a polynomial of degree < Dn_a
// (Dn = 40 .. 48 for the BASELINE hyper-parameters -- the degree the 2-D path's Chebyshev core runs to).  So:
//   1. K1g itself on the Dn_0 x .. x Dn_{d-1} tensor grid of Chebyshev nodes of the first kind (explicit axis positions,
//      launch_posterior_on_axes): mean, variance and the signed gradient components of the mean at 48 x 48 x 40 x 40 = 3.7 M
//      points for config D -- 1.4 % of the grid;
//   2. interpolation to the grid, one axis at a time: g(.., x_a, ..) = sum_k W_a[x_a][k] g(.., node k, ..) with
//      W_a[x][k] = (1 / Dn) sum_m w_m T_m(xi_x) T_m(xi_k)  (the discrete Chebyshev transform and the series evaluation in one
//      matrix).  Axes d-1 .. 2 are contracted by k_t_mode on the small tensors, the last two -- where the data grow to grid
//      size -- by k_t_final on the matrix cores, a (quantity, plane) unit at a time: out = W_0 M W_1^T with the plane's Dn_0 x
//      Dn_1 core M, written in grid order (axis 0 fastest) or, for the gradient components, reduced to max |.| on the fly (the
//      Lipschitz keys).
// Accuracy is checked, not assumed: when a plan is built for a (model, grid) pair, 2048 grid points are also evaluated by the
// generic exact kernel and compared with the interpolated values; a plan that misses 2e-11 (normalised units) retries one step
// up the degree ladder and then declines (K1g runs on the whole grid).  Values differ from K1g's by the rounding of the
// interpolation sums (~1e-13 normalised).  Config D: 216.6 -> 13.8 ms (nodes 3.2, small contractions 1.9, planes 8.7 = 0.62 of
// the FP64 rate).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include "internal.hpp"
#include "device_common.hpp"

namespace sbo {

typedef double d2_t __attribute__((ext_vector_type(2)));
constexpr int kTMaxD = 4;
constexpr int kTProbes = 2048;

struct TensorDims {                  // by value
  int d, q;
  int Dn[kTMaxD];                    // nodes per axis
  long long cnt[kTMaxD];             // grid counts per axis (last: local hyper-planes)
  long long first_last;              // first local index along the last axis
  double mid[kTMaxD], half[kTMaxD];  // axis intervals in raw coordinates
};

// Chebyshev nodes of the first kind per axis, concatenated (axis a at offset sum_{a' < a} Dn): x = mid + half cos(pi (k + 1/2) / Dn)
__global__ void k_t_axes(const TensorDims td, double* __restrict__ axc) {
  int off = 0;
  for (int a = 0; a < td.d; ++a) {
    for (int k = threadIdx.x; k < td.Dn[a]; k += blockDim.x) axc[off + k] = td.mid[a] + td.half[a] * cospi(((double)k + 0.5) / (double)td.Dn[a]);
    off += td.Dn[a];
  }
}

// Analytic part of K1t's guard band (r05): how fast the node tensors' Chebyshev coefficients have decayed at the node count of each
// axis.  A fiber of the node tensor along axis a (nodes of the first kind) has the coefficients c_p = 2 / Dn sum_k v_k cos(pi p (k + 1/2) / Dn);
// the sum of the last four |c_p|, largest over kTFibers pseudo-random fibers per (quantity, axis), is what the band extrapolates the
// interpolation error of that axis from (tensor.hip: the plan's band).  A workgroup per (fiber, axis, quantity).
constexpr int kTFibers = 64;
__global__ __launch_bounds__(64) void k_t_fiber_tail(const TensorDims td, const double* __restrict__ vals /* [nq][Nn] */, long long Nn,
                                                    unsigned long long seed, unsigned long long* __restrict__ out /* [nq][d] */) {
  const int f = blockIdx.x, a = blockIdx.y, Q = blockIdx.z, lane = threadIdx.x, Dn = td.Dn[a];
  // the fiber's base: a pseudo-random multi-index of the other axes (the first fibers pinned to the corners of the node box)
  unsigned long long sd = seed ^ (0x9e3779b97f4a7c15ull * (unsigned long long)(1 + f + 131 * a));
  long long base = 0, mul = 1, stride = 1;
  for (int b = 0; b < td.d; ++b) {
    sd = sd * 6364136223846793005ull + 1442695040888963407ull;
    long long k = (long long)((sd >> 11) % (unsigned long long)td.Dn[b]);
    if (f < (1 << td.d)) k = ((f >> b) & 1) ? td.Dn[b] - 1 : 0;
    if (b == a) { stride = mul; k = 0; }
    base += k * mul;
    mul *= td.Dn[b];
  }
  const double* v = vals + (size_t)Q * Nn + base;
  double s4 = 0.0;
  for (int p = Dn - 4 > 0 ? Dn - 4 : 0; p < Dn; ++p) {
    double acc = 0.0;
    for (int k = lane; k < Dn; k += 64) acc += v[(size_t)k * stride] * cospi((double)p * ((double)k + 0.5) / (double)Dn);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    s4 += fabs(acc) * (2.0 / Dn);
  }
  if (lane == 0 && s4 > 0.0) atomicMax(&out[(size_t)Q * td.d + a], (unsigned long long)__double_as_longlong(s4));
}

// ranks > 1: the nodes are shared out along the LAST axis -- rank r evaluates node planes [r per, (r + 1) per) (the last
// ranks repeat plane Dn - 1 where the count does not divide) --: the axes' positions with the last axis cut to that slab ...
__global__ void k_t_axes_slab(const TensorDims td, const double* __restrict__ axc, int k_first, int per, double* __restrict__ out) {
  int off = 0;
  for (int a = 0; a < td.d - 1; ++a) {
    for (int k = threadIdx.x; k < td.Dn[a]; k += blockDim.x) out[off + k] = axc[off + k];
    off += td.Dn[a];
  }
  const int last = td.Dn[td.d - 1];
  for (int j = threadIdx.x; j < per; j += blockDim.x) out[off + j] = axc[off + (k_first + j < last ? k_first + j : last - 1)];
}
// ... and the gathered slabs [rank][quantity][per planes of `pre` nodes] back into the stacked node tensors [quantity][Nn]
__global__ __launch_bounds__(256) void k_t_unslab(const double* __restrict__ G, int nq, long long pre, int per, int Dn_last, double* __restrict__ out) {
  const long long Nn = pre * Dn_last, Nl = pre * per, total = Nn * nq;
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long qi = e / Nn, rem = e % Nn, k = rem / pre, p = rem % pre;
    out[e] = G[((k / per) * nq + qi) * Nl + (k % per) * pre + p];
  }
}

// interpolation matrix of axis a: W[x][k], x over the (local) grid positions of the axis; transposed copy Wt[k][x] for axes 0 and 1
__global__ __launch_bounds__(256) void k_t_wmat(const TensorDims td, const CandSpec cs, int a, long long nx, long long x_first,
                                                double* __restrict__ W, double* __restrict__ Wt) {
  const int Dn = td.Dn[a];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nx * Dn; i += (long long)gridDim.x * blockDim.x) {
    const long long xl = i / Dn;
    const int k = (int)(i % Dn);
    const long long ix = x_first + xl, tot = cs.count[a];
    const double x = (ix == tot - 1 && tot > 1) ? cs.hi[a] : __dadd_rn(cs.lo[a], __dmul_rn((double)ix, cs.step[a]));   // cand_coords
    double xi = (x - td.mid[a]) / td.half[a];
    xi = xi < 1.0 ? xi : 1.0;
    xi = xi > -1.0 ? xi : -1.0;
    // sum_m w_m T_m(xi) T_m(xi_k), T_m(xi_k) = cos(m pi (k + 1/2) / Dn); T_m(xi) by the three-term recurrence
    double t0 = 1.0, t1 = xi, s_ = 1.0;
    const double th = ((double)k + 0.5) / (double)Dn;
    if (Dn > 1) s_ += 2.0 * t1 * cospi(th);
    for (int m = 2; m < Dn; ++m) {
      const double t = 2.0 * xi * t1 - t0;
      s_ += 2.0 * t * cospi((double)m * th);
      t0 = t1;
      t1 = t;
    }
    const double v = s_ / (double)Dn;
    W[xl * Dn + k] = v;
    if (Wt) Wt[(size_t)k * nx + xl] = v;
  }
}

// position p of a plane's dmc x dmc core in the packed-image order k_t_final reads its A operand in (M^T: rows b, inner index a;
// images [b block][a block][256]) -> (a, b)
__device__ __forceinline__ void core_pos(int p, int dmc, int& a, int& b) {
  const int kb = dmc >> 4, blk = p >> 8, bb = blk / kb, ab = blk % kb;
  int r, k, kk;
  MM<double>::unpack_pos(p & 255, r, k, kk);
  b = bb * 16 + r;
  a = ab * 16 + kk * 4 + k;
}

// contraction of one axis that is not the fastest: out[p + pre (x + nx t)] = sum_m W[x][m] in[p + pre (m + DM t)], for nq
// stacked quantities (strides sin / sout).  A thread per (p, t): its DM inputs in registers, W through LDS in chunks of rows.
// dmc > 0 (the first contraction): the (k0, k1) digits of the output go into packed-image order -- the node tensors arrive
// in grid order (k0 fastest) --, i.e. the thread of output position p reads input position a + dmc b, (a, b) = core_pos(p).
template <int DM>
__global__ __launch_bounds__(256) void k_t_mode(const double* __restrict__ in, size_t sin, double* __restrict__ out, size_t sout,
                                                const double* __restrict__ W, long long pre, int Dm, long long nx, long long post, int dmc,
                                                int q2 /* 2 q */, int gskip /* lean sweeps: the gradient tensors of output 0 (quantities 2 q .. 2 q + d - 1)
                                                are left out -- no sweep reads the objective's Lipschitz key (models/SafeOpt.py:110, GoOSE.py:100) */) {
  constexpr int kRows = 32;
  __shared__ double Ws[kRows][DM];
  const int qi = (int)blockIdx.y < q2 ? (int)blockIdx.y : (int)blockIdx.y + gskip;
  const double* I = in + (size_t)qi * sin;
  double* O = out + (size_t)qi * sout;
  const long long total = pre * post;
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = e < total;
  const long long p = live ? e % pre : 0, t = live ? e / pre : 0;
  long long pin = p;
  if (dmc > 0) {
    const int cc = dmc * dmc;
    int a, b;
    core_pos((int)(p % cc), dmc, a, b);
    pin = (p / cc) * cc + a + dmc * b;
  }
  double v[DM];
#pragma unroll
  for (int m = 0; m < DM; ++m) v[m] = (live && m < Dm) ? I[pin + pre * ((long long)m + (long long)Dm * t)] : 0.0;
  for (long long x0 = 0; x0 < nx; x0 += kRows) {
    __syncthreads();
    for (int i = threadIdx.x; i < kRows * DM; i += blockDim.x) {
      const int r = i / DM, m = i % DM;
      Ws[r][m] = (x0 + r < nx && m < Dm) ? W[(x0 + r) * Dm + m] : 0.0;
    }
    __syncthreads();
    const int rows = nx - x0 < kRows ? (int)(nx - x0) : kRows;
    for (int r = 0; r < rows; ++r) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int m = 0; m < DM; m += 4) {
        s0 = fma(Ws[r][m], v[m], s0);
        s1 = fma(Ws[r][m + 1], v[m + 1], s1);
        s2 = fma(Ws[r][m + 2], v[m + 2], s2);
        s3 = fma(Ws[r][m + 3], v[m + 3], s3);
      }
      if (live) O[p + pre * ((x0 + r) + nx * t)] = (s0 + s1) + (s2 + s3);
    }
  }
}

// The last two axes, where the data grow to grid size: out[x0][x1] = sum_{a,b} W0[x0][a] M[a][b] W1[x1][b] per (x_2, .., x_{d-1})
// plane, all quantities in one launch, on the matrix cores (fragment conventions of device_common.hpp: four v_mfma_f64_4x4x4
// per 16 x 16 x 4 step).  Plain multiply-adds run at the same FP64 rate on gfx950, but an 8 x 8 register tile per thread needs
// 128 B / clk / CU of LDS operands -- all the LDS delivers (measured: 0.34 of the rate); a matrix-core step shares its operands
// across the lanes in hardware.
// A persistent workgroup of four waves; wave w owns the 32 positions x0 = 128 t0 + 32 w .. + 31 (two strips of 16) and keeps
// W0 of them as B fragments in registers (KS x 2 doubles per lane); W1 of a chunk of 16 CS positions x1 sits in LDS as B
// fragments for all four waves.  Units (quantity, plane):
//   step 1 T^T[b][x0] = sum_a M^T[b][a] W0^T[a][x0]: the A images come straight from memory -- the node tensors are stored
//          with the (k0, k1) digits in packed-image order (core_pos) for exactly this, 2 KB contiguous per image;
//          the accumulators go to the wave's own LDS area as the A images of step 2 (rows x0, inner index b)
//   step 2 out[x0][x1] = sum_b T[x0][b] W1[x1][b]: 2 x CS accumulator tiles per wave
//   epilogue: gradient components reduce max |.| -> Lmax[o] (the bit pattern orders non-negative doubles); mean / variance
//          tiles turn through a 16 x 32 LDS patch so that a lane stores 64 contiguous bytes of a grid row (axis 0 fastest).
// Nothing is shared between the waves but the read-only W1 fragments: no barriers in the unit loop when n1 fits one chunk.
// Quantity qi: mean of output qi (qi < q), variance (qi < 2 q, clipped at zero), gradient component (o, a) = (qi - 2 q) / d, % d.
template <int DM, int CS>
__global__ __launch_bounds__(256, 1) void k_t_final(const double* __restrict__ in, size_t stride_q, const double* __restrict__ W0t,
                                                    const double* __restrict__ W1t, long long n0, long long n1, long long planes, int q, int d,
                                                    int nq, double* __restrict__ mean, double* __restrict__ var, size_t n_local,
                                                    unsigned long long* __restrict__ Lmax, int gskip /* lean sweeps: output 0's d gradient
                                                    quantities are not enumerated (see k_t_mode) */) {
  constexpr int KS = DM / 4, KB = DM / 16, SP = 36;   // k-steps, k-blocks; row stride of the store patch
  extern __shared__ __attribute__((aligned(32))) double sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* W1f = sm;                                                       // [KS][CS][64]
  double* Timg = sm + KS * CS * 64 + (size_t)wave * (2 * KB * 256);       // [strip 2][kb][256]
  double* stg = sm + KS * CS * 64 + 4 * (2 * KB * 256) + (size_t)wave * (16 * SP);   // [x1 16][SP]
  // the plane's core for all four waves, double-buffered: the next unit's 18 KB are fetched into registers during step 2 and
  // parked here behind the unit's only barrier, so step 1 never waits for memory (DM <= 48; the LDS has no room at 64)
  constexpr bool PF = DM <= 48;
  constexpr int PFN = DM * DM / 256;
  double* Mimg = sm + KS * CS * 64 + 4 * (2 * KB * 256) + 4 * (16 * SP);              // [2][DM * DM]
  const int kq = lane >> 4, col = lane & 15;
  const long long tiles0 = (n0 + 127) / 128, chunks1 = (n1 + 16 * CS - 1) / (16 * CS);
  const int nqe = nq - gskip;                                             // quantities enumerated
  const long long per_tile = planes * nqe, units = per_tile * tiles0;     // t0 slowest: the W0 fragments change once per tile
  // every wave runs the same number of iterations (the chunk loads of W1 are workgroup-wide); a wave without a unit idles
  const long long iters = (units + gridDim.x - 1) / gridDim.x;
  long long cur_t0 = -1, cur_c1 = -1;
  double bw0[KS][2];
  double pf[PFN];
  auto core_of = [&](long long it_) {
    const long long u_ = blockIdx.x + it_ * gridDim.x, uu_ = u_ < units ? u_ : units - 1, rest_ = uu_ % per_tile;
    const int qe_ = (int)(rest_ % nqe);
    return in + (size_t)(qe_ < 2 * q ? qe_ : qe_ + gskip) * stride_q + (size_t)(rest_ / nqe) * DM * DM;
  };
  if (PF) {
    const double* I0 = core_of(0);
#pragma unroll
    for (int j = 0; j < PFN; ++j) Mimg[tid + 256 * j] = I0[tid + 256 * j];
    __syncthreads();
  }
  for (long long it = 0; it < iters; ++it) {
    const long long u = blockIdx.x + it * gridDim.x;
    const bool have = u < units;
    const long long uu = have ? u : units - 1;
    const long long t0 = uu / per_tile, rest = uu % per_tile;
    const int qe = (int)(rest % nqe), qi = qe < 2 * q ? qe : qe + gskip;
    const long long plane = rest / nqe;
    const long long x0w = t0 * 128 + wave * 32;                          // first position of this wave
    if (t0 != cur_t0) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
          const long long x0 = x0w + s_ * 16 + col;
          bw0[ks][s_] = x0 < n0 ? W0t[(size_t)(ks * 4 + kq) * n0 + x0] : 0.0;
        }
      cur_t0 = t0;
    }
    // step 1
    const double* I = PF ? Mimg + (size_t)(it & 1) * (DM * DM) : in + (size_t)qi * stride_q + (size_t)plane * DM * DM;
    {
      d4_t acc1[KB][2];
#pragma unroll
      for (int bb = 0; bb < KB; ++bb) acc1[bb][0] = acc1[bb][1] = d4_t{0.0, 0.0, 0.0, 0.0};
      // operands of k-step ks + 1 are requested before the products of ks are issued (one wave per SIMD: nothing else hides the
      // LDS latency)
      d4_t am[2][KB];
#pragma unroll
      for (int bb = 0; bb < KB; ++bb) am[0][bb] = MM<double>::load_a(I + (size_t)(bb * KB) * 256, lane, 0);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
#pragma unroll
          for (int bb = 0; bb < KB; ++bb) am[(ks + 1) & 1][bb] = MM<double>::load_a(I + (size_t)(bb * KB + ((ks + 1) >> 2)) * 256, lane, (ks + 1) & 3);
        }
#pragma unroll
        for (int bb = 0; bb < KB; ++bb) {
          acc1[bb][0] = MM<double>::mfma(am[ks & 1][bb], bw0[ks][0], acc1[bb][0]);
          acc1[bb][1] = MM<double>::mfma(am[ks & 1][bb], bw0[ks][1], acc1[bb][1]);
        }
      }
      __builtin_amdgcn_wave_barrier();     // (the previous unit's step 2 reads of T are issued)
      // element t of acc1[bb][s] at lane l: b = 16 bb + 4 t + kq, x0 = 16 s + col  ->  image (strip s, k-block bb),
      // pack_pos(col, kq, t) = 64 t + (4 kq + (col & 3)) 4 + (col >> 2)
      const int tp = (kq * 4 + (col & 3)) * 4 + (col >> 2);
#pragma unroll
      for (int bb = 0; bb < KB; ++bb)
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
          for (int t = 0; t < 4; ++t) Timg[(size_t)(s_ * KB + bb) * 256 + t * 64 + tp] = acc1[bb][s_][t];
      __builtin_amdgcn_wave_barrier();
    }
    if (PF && it + 1 < iters) {
      const double* In = core_of(it + 1);
#pragma unroll
      for (int j = 0; j < PFN; ++j) pf[j] = In[tid + 256 * j];
    }
    const int kind = qi < q ? 0 : (qi < 2 * q ? 1 : 2);
    double* O = kind == 0 ? mean + (size_t)qi * n_local : (kind == 1 ? var + (size_t)(qi - q) * n_local : nullptr);
    if (O) O += (size_t)plane * n0 * n1;
    double gmax = 0.0;
    for (long long c1 = 0; c1 < chunks1; ++c1) {
      if (c1 != cur_c1) {                  // (workgroup-uniform: every wave sees the same chunk sequence)
        if (cur_c1 >= 0) __syncthreads();
        for (int e = tid; e < KS * CS * 64; e += 256) {
          const int l = e & 63, cs = (e >> 6) % CS, ks = (e >> 6) / CS;
          const long long x1 = c1 * (16 * CS) + cs * 16 + (l & 15);
          W1f[e] = x1 < n1 ? W1t[(size_t)(ks * 4 + (l >> 4)) * n1 + x1] : 0.0;
        }
        cur_c1 = c1;
        __syncthreads();
      }
      d4_t acc[2][CS];
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) acc[s_][cs] = d4_t{0.0, 0.0, 0.0, 0.0};
      d4_t at[2][2];
      double wv[2][CS];
      at[0][0] = MM<double>::load_a(Timg, lane, 0);
      at[0][1] = MM<double>::load_a(Timg + (size_t)KB * 256, lane, 0);
#pragma unroll
      for (int cs = 0; cs < CS; ++cs) wv[0][cs] = W1f[cs * 64 + lane];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
          at[(ks + 1) & 1][0] = MM<double>::load_a(Timg + (size_t)((ks + 1) >> 2) * 256, lane, (ks + 1) & 3);
          at[(ks + 1) & 1][1] = MM<double>::load_a(Timg + (size_t)(KB + ((ks + 1) >> 2)) * 256, lane, (ks + 1) & 3);
#pragma unroll
          for (int cs = 0; cs < CS; ++cs) wv[(ks + 1) & 1][cs] = W1f[((ks + 1) * CS + cs) * 64 + lane];
        }
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
          acc[0][cs] = MM<double>::mfma(at[ks & 1][0], wv[ks & 1][cs], acc[0][cs]);
          acc[1][cs] = MM<double>::mfma(at[ks & 1][1], wv[ks & 1][cs], acc[1][cs]);
        }
      }
      // park the prefetched core before this unit's stores are issued: loads and stores return through one in-order counter,
      // so a wait placed behind the stores would wait for them to reach memory
      if (PF && c1 == 0 && it + 1 < iters) {
#pragma unroll
        for (int j = 0; j < PFN; ++j) Mimg[(size_t)((it + 1) & 1) * (DM * DM) + tid + 256 * j] = pf[j];
        __syncthreads();
      }
      // element t of acc[s][cs] at lane l: x0 = x0w + 16 s + 4 t + kq, x1 = c1 16 CS + 16 cs + col
      if (!have) continue;
      if (kind == 2) {
        const bool full = x0w + 32 <= n0 && (c1 + 1) * (16 * CS) <= n1;       // (wave-uniform)
        if (full) {
          double g0 = gmax, g1 = 0.0, g2 = 0.0, g3 = 0.0;                      // one v_max_f64 with |.| per element, four chains
#pragma unroll
          for (int cs = 0; cs < CS; ++cs)
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) {
              g0 = fmax(g0, fabs(acc[s_][cs][0]));
              g1 = fmax(g1, fabs(acc[s_][cs][1]));
              g2 = fmax(g2, fabs(acc[s_][cs][2]));
              g3 = fmax(g3, fabs(acc[s_][cs][3]));
            }
          gmax = fmax(fmax(g0, g1), fmax(g2, g3));
        } else {
#pragma unroll
          for (int cs = 0; cs < CS; ++cs) {
            const bool live1 = c1 * (16 * CS) + cs * 16 + col < n1;
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const double v = acc[s_][cs][t], av = v < 0 ? -v : v;
                gmax = (live1 && x0w + s_ * 16 + 4 * t + kq < n0 && av > gmax) ? av : gmax;
              }
          }
        }
      } else {
        const bool vec = (n0 & 1) == 0;
#pragma unroll
        for (int cs = 0; cs < CS; ++cs) {
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              const double v = acc[s_][cs][t];
              stg[col * SP + s_ * 16 + 4 * t + kq] = kind == 1 ? (v > 0.0 ? v : 0.0) : v;
            }
          __builtin_amdgcn_wave_barrier();
          // read back as rows: lane (l >> 4, l & 15) takes two x0 of patch rows l >> 4, 4 + (l >> 4), ..: one store instruction
          // writes four whole 256-byte row pieces
          const int xc = col * 2;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int xr = r * 4 + kq;
            const d2_t v = *reinterpret_cast<const d2_t*>(stg + xr * SP + xc);
            const long long x1 = c1 * (16 * CS) + cs * 16 + xr, x0 = x0w + xc;
            if (x1 < n1) {
              double* row = O + (size_t)x1 * n0;
              if (vec) {
                if (x0 < n0) *reinterpret_cast<d2_t*>(row + x0) = v;
              } else {
                if (x0 < n0) row[x0] = v[0];
                if (x0 + 1 < n0) row[x0 + 1] = v[1];
              }
            }
          }
        }
      }
    }
    if (kind == 2 && have) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double y = __shfl_xor(gmax, off);
        gmax = y > gmax ? y : gmax;
      }
      if (lane == 0 && gmax > 0.0) atomicMax(Lmax + (qi - 2 * q) / d, (unsigned long long)__double_as_longlong(gmax));
    }
  }
}

// values of stacked quantities at listed grid points (the plan's accuracy probe): out[qi np + i] = arr[qi stride + idx[i]]
__global__ void k_t_pick(const double* __restrict__ arr, size_t stride, const long long* __restrict__ idx, int np, double* __restrict__ out) {
  const int qi = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < np; i += gridDim.x * blockDim.x) out[(size_t)qi * np + i] = arr[(size_t)qi * stride + idx[i]];
}
template <int D>
__global__ void k_t_probe_pts(const CandSpec cs, const long long* __restrict__ idx, int np, double* __restrict__ pts) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < np; i += gridDim.x * blockDim.x) {
    double x[D];
    cand_coords<D>(cs, idx[i], x);
    for (int a = 0; a < cs.d; ++a) pts[(size_t)i * cs.d + a] = x[a];
  }
}

// the interpolant of quantity qi at listed grid points, from the tensor the small contractions leave (cores [planes][DM x DM] in
// packed-image order per quantity): out[(qi - q_first) np + i] = W0[x0] . M . W1[x1].  A wave per (point, quantity).  Used by the
// plan's probe for the gradient components, which k_t_final reduces to their maximum without ever storing them.
__global__ __launch_bounds__(256) void k_t_probe_core(const double* __restrict__ cur, size_t stride_q, int DM, const double* __restrict__ W0,
                                                      const double* __restrict__ W1, long long n0, long long n1,
                                                      const long long* __restrict__ idx, int np, int q_first, int nq, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long w = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, total = (long long)np * nq;
  if (w >= total) return;
  const int i = (int)(w % np), qi = q_first + (int)(w / np);
  const long long g = idx[i], x0 = g % n0, x1 = (g / n0) % n1, plane = g / (n0 * n1);
  const double* M = cur + (size_t)qi * stride_q + (size_t)plane * DM * DM;
  double s = 0.0;
  for (int p = lane; p < DM * DM; p += 64) {
    int a, b;
    core_pos(p, DM, a, b);
    s = fma(W0[x0 * DM + a] * W1[x1 * DM + b], M[p], s);
  }
  s = wave_sum(s);
  if (lane == 0) out[(size_t)(qi - q_first) * np + i] = s;
}

// ---- host ----------------------------------------------------------------------------------------------------------------
int comm_allreduce_min_u64(sbo_ctx* c, unsigned long long* dev, int count);
int comm_allgather_bytes(sbo_ctx* c, const void* send, void* recv, size_t bytes_per_rank);
// node counts: the first two axes in matrix-core k-blocks (k_t_final), the further ones in steps of eight (k_t_mode takes any)
static const int kLadder01[4] = {32, 48, 64, 96};
static const int kLadderN[7] = {32, 40, 48, 56, 64, 80, 96};
static int ladder_size(int a) { return a < 2 ? 4 : 7; }
static int ladder_at(int a, int lv) { return a < 2 ? kLadder01[std::min(lv, 3)] : kLadderN[std::min(lv, 6)]; }

bool tensor_applicable(const sbo_ctx* c) {
  if (!c->tensor_cheb || c->is_shadow || c->dtype != SBO_F64 || c->cs.kind != 1) return false;
  const int d = c->cs.d;
  if (d < 3 || d > kTMaxD) return false;
  long long plane = 1;
  for (int a = 0; a < d - 1; ++a) plane *= c->cs.count[a];
  if (c->cs.n_local <= 0 || c->cs.first % plane != 0 || c->cs.n_local % plane != 0) return false;
  for (int a = 0; a < d; ++a)
    if (c->cs.count[a] < 64) return false;
  return c->grid_total >= (1ll << 22);
}

template <int DM>
static void launch_mode(hipStream_t st, const double* in, size_t sin, double* out, size_t sout, const double* W, long long pre, int Dm,
                        long long nx, long long post, int nq, int dmc, int q2, int gskip) {
  const long long total = pre * post;
  hipLaunchKernelGGL((k_t_mode<DM>), dim3((unsigned)((total + 255) / 256), (unsigned)(nq - gskip)), dim3(256), 0, st, in, sin, out, sout, W, pre, Dm, nx,
                     post, dmc, q2, gskip);
}
static int mode_dispatch(hipStream_t st, int DM, const double* in, size_t sin, double* out, size_t sout, const double* W, long long pre, int Dm,
                         long long nx, long long post, int nq, int dmc, int q2, int gskip) {
  switch (DM) {
    case 32: launch_mode<32>(st, in, sin, out, sout, W, pre, Dm, nx, post, nq, dmc, q2, gskip); break;
    case 48: launch_mode<48>(st, in, sin, out, sout, W, pre, Dm, nx, post, nq, dmc, q2, gskip); break;
    case 64: launch_mode<64>(st, in, sin, out, sout, W, pre, Dm, nx, post, nq, dmc, q2, gskip); break;
    case 96: launch_mode<96>(st, in, sin, out, sout, W, pre, Dm, nx, post, nq, dmc, q2, gskip); break;
    default: return fail(SBO_E_UNSUPPORTED, "internal: interpolation degree");
  }
  return SBO_OK;
}
template <int DM, int CS>
static int launch_final(sbo_ctx* c, const double* in, size_t stride_q, const TensorDims& td, long long planes, int nq, int gskip) {
  const size_t lds = sizeof(double) * ((size_t)(DM / 4) * CS * 64 + 4 * (size_t)(2 * (DM / 16) * 256) + 4 * 16 * 36 + (DM <= 48 ? 2 * DM * DM : 0));
  auto kern = k_t_final<DM, CS>;
  SBO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const long long units = planes * (nq - gskip) * ((td.cnt[0] + 127) / 128);
  const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>(units, (long long)c->n_cu));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, c->stream, in, stride_q, (const double*)c->tn_W0t.p, (const double*)c->tn_W1t.p, td.cnt[0],
                     td.cnt[1], planes, td.q, td.d, nq, (double*)c->mean.p, (double*)c->var.p, (size_t)c->cs.n_local, (unsigned long long*)c->Lmax.p, gskip);
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

// all `nq` stacked node tensors (stride Nn: q means, q variances, q d gradient components) to the grid
static int interpolate(sbo_ctx* c, const TensorDims& td, const double* nodes, long long Nn, int nq, const double** cur_out, size_t* stride_out) {
  const int d = td.d;
  int rc;
  const int gskip = (c->sweep_lean && td.q >= 2) ? d : 0;      // (a lean sweep: nobody reads L_0)
  // axes d-1 .. 2 on the small tensors (ping-pong in tn_work), then the planes
  const double* cur = nodes;
  size_t cur_stride = (size_t)Nn;
  long long pre = 1;
  for (int a = 0; a < d - 1; ++a) pre *= td.Dn[a];          // product of the node counts below the axis being contracted
  long long post = 1;
  double* bufA = (double*)c->tn_work.p;
  double* bufB = bufA + c->tn_work_half;
  bool toA = true;
  for (int a = d - 1; a >= 2; --a) {
    const long long nx = td.cnt[a];
    double* dst = toA ? bufA : bufB;
    const size_t dst_stride = (size_t)pre * nx * post;
    if ((rc = mode_dispatch(c->stream, td.Dn[a] <= 32 ? 32 : (td.Dn[a] <= 48 ? 48 : (td.Dn[a] <= 64 ? 64 : 96)), cur, cur_stride, dst, dst_stride,
                            (const double*)c->tn_W[a].p, pre, td.Dn[a], nx, post, nq, a == d - 1 ? td.Dn[0] : 0, 2 * td.q, gskip)))
      return rc;
    c->tn_flops += 2.0 * (double)pre * td.Dn[a] * (double)nx * (double)post * (nq - gskip);
    cur = dst;
    cur_stride = dst_stride;
    post *= nx;
    pre /= td.Dn[a - 1];
    toA = !toA;
  }
  // now cur = [DM][DM][planes = post] per quantity
  *cur_out = cur;
  *stride_out = cur_stride;
  c->tn_flops += 2.0 * (double)post * (nq - gskip) * ((double)td.cnt[0] * td.Dn[0] * td.Dn[1] + (double)td.cnt[0] * td.cnt[1] * td.Dn[1]);
  SBO_HIP(hipMemsetAsync(c->Lmax.p, 0, sizeof(unsigned long long) * kMaxQ, c->stream));
  switch (td.Dn[0]) {
    case 32: return launch_final<32, 8>(c, cur, cur_stride, td, post, nq, gskip);
    case 48: return launch_final<48, 8>(c, cur, cur_stride, td, post, nq, gskip);
    case 64: return launch_final<64, 8>(c, cur, cur_stride, td, post, nq, gskip);
  }
  return fail(SBO_E_UNSUPPORTED, "internal: interpolation degree");
}

// SBO_OK with *declined = true: the plan does not qualify (K1g runs)
int launch_posterior_tensor(sbo_ctx* c, bool* declined) {
  *declined = true;
  const ModelConst& mc = c->mc;
  const CandSpec& cs = c->cs;
  const int d = cs.d, q = mc.q;
  int rc;
  long long plane = 1;
  for (int a = 0; a < d - 1; ++a) plane *= cs.count[a];
  TensorDims td;
  memset(&td, 0, sizeof(td));
  td.d = d;
  td.q = q;
  for (int a = 0; a < d; ++a) {
    td.cnt[a] = a == d - 1 ? cs.n_local / plane : cs.count[a];
    td.mid[a] = 0.5 * (cs.lo[a] + cs.hi[a]);
    td.half[a] = 0.5 * (cs.hi[a] - cs.lo[a]);
    if (!(td.half[a] > 0.0)) return SBO_OK;
  }
  td.first_last = cs.first / plane;
  // plan: degrees per axis, decided once per (model, grid) with an accuracy probe
  const bool same_grid = c->tn_valid && c->tn_model == c->model_serial && c->tn_first == cs.first && c->tn_nlocal == cs.n_local &&
                         !memcmp(c->tn_count, cs.count, sizeof(long long) * kTMaxD) && !memcmp(c->tn_lo, cs.lo, sizeof(double) * kTMaxD) &&
                         !memcmp(c->tn_hi, cs.hi, sizeof(double) * kTMaxD);
  // same candidates as the last plan's, whatever the model
  const bool same_box = c->tn_valid && c->tn_first == cs.first && c->tn_nlocal == cs.n_local && !memcmp(c->tn_count, cs.count, sizeof(long long) * kTMaxD) &&
                        !memcmp(c->tn_lo, cs.lo, sizeof(double) * kTMaxD) && !memcmp(c->tn_hi, cs.hi, sizeof(double) * kTMaxD);
  if (!same_box) c->tn_bump = 0;
  int level0[kTMaxD], base0[kTMaxD] = {0, 0, 0, 0};
  if (same_grid) {
    if (!c->tn_usable) return SBO_OK;
    for (int a = 0; a < d; ++a) level0[a] = c->tn_level[a];
  } else {
    // first guess from the shortest length scale of the axis: degree ~ 7.6 x (normalised interval / ell); the probe below corrects a short guess
    for (int a = 0; a < d; ++a) {
      double tmax = 0.0;
      for (int o = 0; o < q; ++o) tmax = std::max(tmax, 2.0 * td.half[a] / mc.X_std[a] * std::sqrt(mc.inv_ell[o][a]));
      // (the further axes' ladder is finer and their guess tighter: 40 nodes measured 1e-13 where this rule gives 39.4)
      const double want = (a < 2 ? 7.6 : 6.9) * tmax * (0.01 * c->tensor_guess_pct);     // (option: a test hook for the second attempt)
      int lv = 0;
      while (lv < ladder_size(a) - 1 && ladder_at(a, lv) < want) ++lv;
      base0[a] = lv;
      level0[a] = lv + (same_box ? c->tn_bump : 0);     // (the previous model on this grid needed the second attempt: start there)
    }
  }
  for (int attempt = 0; attempt < (same_grid ? 1 : 2); ++attempt) {
    long long Nn = 1;
    bool ok_dims = true;
    for (int a = 0; a < d; ++a) {
      td.Dn[a] = same_grid ? c->tn_dn[a] : ladder_at(a, level0[a] + attempt);
      // (steps beyond the first guess: the first two axes stay within what k_t_final takes -- the error may well sit on the others)
      if (!same_grid && a < 2 && td.Dn[a] > 64 && ladder_at(a, base0[a]) <= 64) td.Dn[a] = 64;
    }
    td.Dn[0] = td.Dn[1] = std::max(td.Dn[0], td.Dn[1]);      // (k_t_final: one square core per plane)
    for (int a = 0; a < d; ++a) {
      if (td.Dn[a] > 64 && a < 2) ok_dims = false;          // (k_t_final's LDS: cores up to 64 x 64)
      if (td.Dn[a] * 4 > 3 * cs.count[a]) ok_dims = false;   // not worth it: the grid is hardly finer than the nodes
      Nn *= td.Dn[a];
    }
    if (!ok_dims || Nn > (1ll << 24)) break;
    // buffers: node list, node values (mean, var: q each; gradient: q d), interpolation matrices, ping-pong work
    const int nqg = q * d;
    if ((rc = ensure(c->tn_pts, sizeof(double) * 8 * 128))) return rc;       // the node positions of the axes (+ a rank's slab of them)
    // (the two large buffers: running out of memory for them is a reason to decline -- K1g needs neither --, not to fail the sweep.
    // With ranks > 1 it IS an error: the plan is one decision of all ranks, and a rank that left for K1g alone would leave the others
    // in the all-gather of the node slabs -- ADVICE r04)
    if ((rc = ensure(c->tn_vals, sizeof(double) * (size_t)Nn * (2 * q + nqg)))) { if (rc == SBO_E_NOMEM && !multi_rank(c)) break; return rc; }
    size_t half_elems = 0;
    {
      long long pre = 1, post = 1;
      for (int a = 0; a < d - 1; ++a) pre *= td.Dn[a];
      for (int a = d - 1; a >= 2; --a) {
        half_elems = std::max(half_elems, (size_t)(pre * td.cnt[a] * post) * (size_t)(2 * q + nqg));
        post *= td.cnt[a];
        pre /= td.Dn[a - 1];
      }
    }
    c->tn_work_half = half_elems;
    if ((rc = ensure(c->tn_work, sizeof(double) * 2 * std::max<size_t>(half_elems, 16)))) { if (rc == SBO_E_NOMEM && !multi_rank(c)) break; return rc; }
    for (int a = 0; a < d; ++a)
      if ((rc = ensure(c->tn_W[a], sizeof(double) * (size_t)td.cnt[a] * td.Dn[a]))) return rc;
    if ((rc = ensure(c->tn_W0t, sizeof(double) * (size_t)td.cnt[0] * td.Dn[0]))) return rc;
    if ((rc = ensure(c->tn_W1t, sizeof(double) * (size_t)td.cnt[1] * td.Dn[1]))) return rc;
    if ((rc = ensure(c->tn_scr, 512))) return rc;
    hipLaunchKernelGGL(k_t_axes, dim3(1), dim3(128), 0, c->stream, td, (double*)c->tn_pts.p);
    for (int a = 0; a < d; ++a)
      hipLaunchKernelGGL(k_t_wmat, dim3((unsigned)std::min<long long>((td.cnt[a] * td.Dn[a] + 255) / 256, 4096)), dim3(256), 0, c->stream, td, cs, a,
                         td.cnt[a], a == d - 1 ? td.first_last : 0ll, (double*)c->tn_W[a].p, a == 0 ? (double*)c->tn_W0t.p : (a == 1 ? (double*)c->tn_W1t.p : (double*)nullptr));
    // exact values at the nodes
    double* nmean = (double*)c->tn_vals.p;
    double* nvar = nmean + (size_t)q * Nn;
    double* ngrad = nvar + (size_t)q * Nn;
    const int W = c->world;
    if (W > 1 && td.Dn[d - 1] >= W) {
      // (r04) every rank used to evaluate ALL the nodes -- 3.2 of config D's 13.8 ms that no number of GPUs made shorter.  A slab
      // of the last axis' node planes per rank, one all-gather of the slabs (12 quantities x 3.7 M nodes x 8 B = 354 MB in all
      // for config D: ~1 ms over xGMI at 8 ranks), and the stacked tensors are put together again on every rank.
      const int nq_all = 2 * q + nqg, per = (td.Dn[d - 1] + W - 1) / W;
      long long pre_all = 1;
      for (int a = 0; a < d - 1; ++a) pre_all *= td.Dn[a];
      const long long Nl = pre_all * per;
      if ((rc = ensure(c->tn_gather, sizeof(double) * (size_t)nq_all * (size_t)Nl * (size_t)(W + 1)))) { if (rc == SBO_E_NOMEM && !multi_rank(c)) break; return rc; }
      double* slab = (double*)c->tn_gather.p;
      double* gathered = slab + (size_t)nq_all * Nl;
      double* axl = (double*)c->tn_pts.p + 4 * 128;
      hipLaunchKernelGGL(k_t_axes_slab, dim3(1), dim3(128), 0, c->stream, td, (const double*)c->tn_pts.p, c->rank * per, per, axl);
      long long cnt[kTMaxD];
      for (int a = 0; a < d; ++a) cnt[a] = a == d - 1 ? per : td.Dn[a];
      if ((rc = launch_posterior_on_axes(c, d, cnt, (const double*)axl, slab, slab + (size_t)q * Nl, slab + (size_t)2 * q * Nl,
                                         (unsigned long long*)c->tn_scr.p)))
        return rc;
      if ((rc = comm_allgather_bytes(c, slab, gathered, sizeof(double) * (size_t)nq_all * Nl))) return rc;
      hipLaunchKernelGGL(k_t_unslab, dim3((unsigned)std::min<long long>((Nn * nq_all + 255) / 256, 1 << 16)), dim3(256), 0, c->stream,
                         (const double*)gathered, nq_all, pre_all, per, td.Dn[d - 1], nmean);
    } else {
      long long cnt[kTMaxD];
      for (int a = 0; a < d; ++a) cnt[a] = td.Dn[a];
      if ((rc = launch_posterior_on_axes(c, d, cnt, (const double*)c->tn_pts.p, nmean, nvar, ngrad, (unsigned long long*)c->tn_scr.p))) return rc;
    }
    // (the coefficient tails of the node tensors of mean and variance along every axis: the analytic part of the plan's band, below)
    if (!same_grid) {
      if ((rc = ensure(c->tn_tail, sizeof(unsigned long long) * 2 * kMaxQ * kTMaxD))) return rc;
      SBO_HIP(hipMemsetAsync(c->tn_tail.p, 0, sizeof(unsigned long long) * 2 * kMaxQ * kTMaxD, c->stream));
      hipLaunchKernelGGL(k_t_fiber_tail, dim3(kTFibers, (unsigned)d, (unsigned)(2 * q)), dim3(64), 0, c->stream, td, (const double*)nmean, Nn,
                         0x9e3779b97f4a7c15ull ^ (unsigned long long)c->model_serial, (unsigned long long*)c->tn_tail.p);
    }
    // interpolation: mean, variance (clipped at zero), gradient components -> Lipschitz keys max_a max_x |d MEAN_o / d x_a|
    c->tn_flops = 0.0;
    const double* cur = nullptr;
    size_t cur_stride = 0;
    if ((rc = interpolate(c, td, nmean, Nn, 2 * q + nqg, &cur, &cur_stride))) return rc;
    if (!same_grid) {
      // accuracy probe: 2048 grid points, exact against interpolated
      std::vector<long long> idx(kTProbes);
      unsigned long long sd = 0x9e3779b97f4a7c15ull ^ (unsigned long long)c->model_serial;
      // pseudo-random local candidates; the first ones pinned to the ends of the axes (corners of the local box, then points on
      // its faces), where a polynomial interpolant errs most
      for (int i = 0; i < kTProbes; ++i) {
        long long f = 0, mul = 1;
        for (int a = 0; a < d; ++a) {
          sd = sd * 6364136223846793005ull + 1442695040888963407ull;
          const long long cnt = td.cnt[a];
          long long k = (long long)((sd >> 11) % (unsigned long long)cnt);
          if (i < (1 << d)) k = ((i >> a) & 1) ? cnt - 1 : 0;
          else if (i < 512 && a == i % d) k = (i & 64) ? cnt - 1 : 0;
          f += k * mul;
          mul *= cnt;
        }
        idx[i] = f;
      }
      // probe buffers: indices | points | exact mean, var | interpolated mean, var | exact gradient, interpolated gradient [q d] | keys
      const size_t pbytes = sizeof(long long) * kTProbes + sizeof(double) * ((size_t)kTProbes * (d + 4 * q + 2 * nqg) + kMaxQ);
      if ((rc = ensure(c->tn_probe, pbytes))) return rc;
      long long* didx = (long long*)c->tn_probe.p;
      double* ppts = (double*)(didx + kTProbes);
      double* pexm = ppts + (size_t)kTProbes * d;          // exact mean [q][np], exact var, interpolated mean, interpolated var
      double* pexv = pexm + (size_t)q * kTProbes;
      double* pinm = pexv + (size_t)q * kTProbes;
      double* pinv = pinm + (size_t)q * kTProbes;
      double* pexg = pinv + (size_t)q * kTProbes;          // exact / interpolated signed gradient components [q d][np]
      double* ping = pexg + (size_t)nqg * kTProbes;
      double* pkeys = ping + (size_t)nqg * kTProbes;       // the Lipschitz keys k_t_final has just reduced
      SBO_HIP(hipMemcpyAsync(didx, idx.data(), sizeof(long long) * kTProbes, hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL((k_t_probe_pts<4>), dim3(8), dim3(256), 0, c->stream, cs, (const long long*)didx, kTProbes, ppts);
      if ((rc = launch_posterior_on_list(c, ppts, kTProbes, pexm, pexv))) return rc;
      if ((rc = guard_exact_grad_list(c, ppts, kTProbes, pexg))) return rc;
      hipLaunchKernelGGL(k_t_pick, dim3(8, q), dim3(256), 0, c->stream, (const double*)c->mean.p, (size_t)cs.n_local, (const long long*)didx, kTProbes, pinm);
      hipLaunchKernelGGL(k_t_pick, dim3(8, q), dim3(256), 0, c->stream, (const double*)c->var.p, (size_t)cs.n_local, (const long long*)didx, kTProbes, pinv);
      hipLaunchKernelGGL(k_t_probe_core, dim3((unsigned)(((long long)kTProbes * nqg * 64 + 255) / 256)), dim3(256), 0, c->stream, cur, cur_stride,
                         td.Dn[0], (const double*)c->tn_W[0].p, (const double*)c->tn_W[1].p, td.cnt[0], td.cnt[1], (const long long*)didx, kTProbes,
                         2 * q, nqg, ping);
      SBO_HIP(hipMemcpyAsync(pkeys, c->Lmax.p, sizeof(double) * kMaxQ, hipMemcpyDeviceToDevice, c->stream));
      std::vector<double> h((size_t)(4 * q + 2 * nqg) * kTProbes + kMaxQ);
      double tails[2 * kMaxQ * kTMaxD];
      SBO_HIP(hipMemcpyAsync(h.data(), pexm, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream));
      SBO_HIP(hipMemcpyAsync(tails, c->tn_tail.p, sizeof(tails), hipMemcpyDeviceToHost, c->stream));
      SBO_HIP(hipStreamSynchronize(c->stream));
      // deviations per output (un-normalised: the band's units) and the probe error in normalised units (the plan's gate);
      // a value that is not finite fails the gate (NaN compares false)
      double err = 0.0, em[kMaxQ], ev[kMaxQ], eg[kMaxQ], am[kMaxQ], av[kMaxQ];
      bool finite = true;
      for (int o = 0; o < q; ++o) {
        const double ys = std::max(1.0, mc.Y_std[o]);
        em[o] = ev[o] = eg[o] = am[o] = av[o] = 0.0;
        for (int i = 0; i < kTProbes; ++i) {
          const double xm = h[(size_t)o * kTProbes + i], xv = h[((size_t)q + o) * kTProbes + i];
          const double dmv = std::fabs(h[((size_t)2 * q + o) * kTProbes + i] - xm), dvv = std::fabs(h[((size_t)3 * q + o) * kTProbes + i] - xv);
          finite = finite && std::isfinite(dmv) && std::isfinite(dvv);
          em[o] = std::max(em[o], dmv);
          ev[o] = std::max(ev[o], dvv);
          am[o] = std::max(am[o], std::fabs(xm));
          av[o] = std::max(av[o], std::fabs(xv));
        }
        for (int a = 0; a < d; ++a)
          for (int i = 0; i < kTProbes && !(o == 0 && c->sweep_lean && q >= 2); ++i) {      // (lean: output 0's gradient cores were not made)
            const double dg = std::fabs(h[((size_t)4 * q + nqg + (size_t)o * d + a) * kTProbes + i] - h[((size_t)4 * q + (size_t)o * d + a) * kTProbes + i]);
            finite = finite && std::isfinite(dg);
            eg[o] = std::max(eg[o], dg);
          }
        err = std::max(err, std::max(em[o] / ys, ev[o] / (ys * ys)));
      }
      if (getenv("SBO_DEBUG_TENSOR"))
        fprintf(stderr, "[K1t] nodes %d x %d x %d x %d = %lld, probe error %.2e (gradient %.2e / %.2e) (attempt %d)\n", td.Dn[0], td.Dn[1], td.Dn[2],
                d > 3 ? td.Dn[3] : 1, Nn, err, eg[0], q > 1 ? eg[1] : 0.0, attempt);
      // ranks > 1: the plan is ONE decision -- a rank whose shard misses the gate takes every rank to the next attempt / to K1g
      // (results must not depend on the world size)
      unsigned long long ok_local = (finite && err <= 2e-11) ? 1ull : 0ull, ok_all = ok_local;
      {
        unsigned long long* dkey = reinterpret_cast<unsigned long long*>(pkeys) + kMaxQ - 1;    // (a spare word of the probe block)
        SBO_HIP(hipMemcpyAsync(dkey, &ok_local, 8, hipMemcpyHostToDevice, c->stream));
        if ((rc = comm_allreduce_min_u64(c, dkey, 1))) return rc;
        SBO_HIP(hipMemcpyAsync(&ok_all, dkey, 8, hipMemcpyDeviceToHost, c->stream));
        SBO_HIP(hipStreamSynchronize(c->stream));
      }
      // the plan's guard band (guard.hip).  Analytic ESTIMATE (r05): interpolating axis by axis, the error is at most
      // sum_a (prod_{b != a} Lambda_b) E_a with Lambda_b <= 2 / pi ln Dn_b + 1 the Lebesgue constant of Chebyshev nodes of the first kind and
      // E_a the interpolation error along axis a -- at most twice the coefficients beyond Dn_a, extrapolated as kAlias x the sum of the
      // last four coefficients held, largest over 64 sampled fibers of the node tensor (an estimate: sampled fibers, extrapolated tail).
      // Measured part: 16 x the largest probe deviation (the rounding of the interpolation sums) + a floor; the Lipschitz keys'
      // relative band from the gradient components' deviation.  The probes gate the plan as before (2e-11).
      const double eps = 2.220446049250313e-16, safety = 16.0, kAlias = 4.0;
      for (int o = 0; o < q; ++o) {
        const double Lo = h[(size_t)(4 * q + 2 * nqg) * kTProbes + o];
        double an[2] = {0.0, 0.0};
        for (int w = 0; w < 2; ++w)
          for (int a = 0; a < d; ++a) {
            double lam = 1.0;
            for (int b = 0; b < d; ++b)
              if (b != a) lam *= 2.0 / 3.141592653589793 * std::log((double)td.Dn[b]) + 1.0;
            double t4;
            memcpy(&t4, &tails[((size_t)(w * q + o)) * d + a], 8);
            an[w] += lam * kAlias * t4;
          }
        // (REPORTED, not added: with the Lebesgue products of a tensor interpolant the estimate is ~1e3 x the deviations seen at the 2048
        // probes and on whole-grid comparisons with K1g -- 7e-9 against 1e-12 in the variance of a 64^3 grid --, and as a band it would send
        // about a fifth of config D's sweeps into a second pass of their 5.7 ms set phase.  K1t's band stays MEASURED: sbo_profile shows both)
        c->tn_band[o] = safety * em[o] + 64.0 * eps * std::max(am[o], std::fabs(mc.Y_mean[o]) + mc.Y_std[o]);
        c->tn_band[kMaxQ + o] = safety * ev[o] + 64.0 * eps * std::max(av[o], mc.sf2[o] * mc.Y_std[o] * mc.Y_std[o]);
        c->tn_band[2 * kMaxQ + o] = (Lo > 0.0 ? safety * eg[o] / Lo : 0.0) + 1e-13;
        c->tn_band[3 * kMaxQ + o] = an[0];
        c->tn_band[4 * kMaxQ + o] = an[1];
        c->tn_band[5 * kMaxQ + o] = em[o];
        c->tn_band[6 * kMaxQ + o] = ev[o];
      }
      c->tn_valid = true;
      c->tn_model = c->model_serial;
      c->tn_first = cs.first;
      c->tn_nlocal = cs.n_local;
      memcpy(c->tn_count, cs.count, sizeof(long long) * kTMaxD);
      memcpy(c->tn_lo, cs.lo, sizeof(double) * kTMaxD);
      memcpy(c->tn_hi, cs.hi, sizeof(double) * kTMaxD);
      for (int a = 0; a < d; ++a) {
        c->tn_level[a] = std::min(ladder_size(a) - 1, level0[a] + attempt);
        c->tn_dn[a] = td.Dn[a];
      }
      c->tn_usable = ok_all != 0ull;
      if (!c->tn_usable) continue;           // one step up the ladder, or give up
      if (attempt > 0) c->tn_bump = std::min(2, c->tn_bump + 1);
    }
    *declined = false;
    c->last_k1 = 5;
    if (c->guard_band) {
      if ((rc = guard_band_host(c, c->tn_band, c->tn_band + kMaxQ, c->tn_band + 2 * kMaxQ, c->tn_band + 3 * kMaxQ))) return rc;   // (whoever used the block last)
      c->gb_active = true;
    }
    // flops issued: node posterior (block-triangular contraction) + interpolation sums
    c->last_k1_flops = (double)q * mc.npad * (mc.npad + 16.0) * (double)Nn / (double)((c->world > 1 && td.Dn[d - 1] >= c->world) ? c->world : 1) +
                       c->tn_flops;
    return SBO_OK;
  }
  c->tn_valid = true;
  c->tn_usable = false;
  c->tn_model = c->model_serial;
  c->tn_first = cs.first;
  c->tn_nlocal = cs.n_local;
  memcpy(c->tn_count, cs.count, sizeof(long long) * kTMaxD);
  memcpy(c->tn_lo, cs.lo, sizeof(double) * kTMaxD);
  memcpy(c->tn_hi, cs.hi, sizeof(double) * kTMaxD);
  return SBO_OK;
}

}  // namespace sbo
