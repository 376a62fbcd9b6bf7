// sets_goose.inc.hpp: GoOSE optimistic sets (pair evaluation and power-distance transform), trust-region helpers -- part of the sets.hip translation unit (included inside namespace sbo; not a standalone header).
#pragma once

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
                                                       const uint8_t* __restrict__ src, T* __restrict__ W, SweepScalars* sc) {
  // (also clears the recheck / scan counters for the coverage kernels that follow: the expander's are done with them)
  if (blockIdx.x == 0 && threadIdx.x == 0) { sc->n_amb = 0; sc->n_scan = 0; }
  // mean / var are read for the sources only (a tenth of config B's grid): wave tiles of 512 candidates as in
  // k_arg_masked, every lane reads eight mask bytes as one word
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  const long long ntiles = (((uintptr_t)src) & 7) == 0 ? n / 512 : 0;
  for (long long t = wave; t < ntiles; t += nwaves) {
    const long long base = t * 512;
    const unsigned long long w = ((const unsigned long long*)(src + base))[lane];
    const bool any = __ballot(w != 0ull) != 0ull;
    T mu[8], va[8];
    bool set[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      set[k] = any && tile_byte(w, k, lane);
      const long long g = base + k * 64 + lane;
      mu[k] = set[k] ? mean_c[g] : (T)0;
      va[k] = set[k] ? var_c[g] : (T)0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      T lcb, ucb;
      lcb_ucb(mu[k], va[k], b, lcb, ucb);
      W[base + k * 64 + lane] = set[k] ? ucb : (T)-INFINITY;
    }
  }
  for (long long g = ntiles * 512 + (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
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
  double gband;                // the part of `band` that comes from the guard band of an approximating posterior (0: none)
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
  // guard band: a source's radius r = ucb_c / L is known to dr = du / L + rl r, so r^2 -- and with it P -- to 2 r dr + dr^2; the
  // band is widened by that much, and a verdict inside the widened band is counted (SweepScalars::n_guard)
  const double dr = sc->gb_du[c] * p.invL + sc->gb_rl[lidx] * rm;
  p.gband = dr > 0.0 ? (2.0 * rm * dr + dr * dr) * (1.0 + 1e-9) : 0.0;
  p.band += p.gband;
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
// (r03: eight lanes per cell, each walking every eighth line of the cell -- a thread per cell ran its kCoarse^(d-1) lines of
// eight loads one after the other, 12 us of pure load latency on config C's 16384 cells; the maximum does not depend on the order)
template <typename T>
__global__ __launch_bounds__(256) void k_pdt_cell_min(const T* __restrict__ W, const CoarseGrid cg, long long nc,
                                                      const unsigned long long* Lkeys, int lidx, double* __restrict__ lo,
                                                      double* __restrict__ hi) {
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const double invL = L > 0 ? 1.0 / L : 0.0;
  const int sub = threadIdx.x & 7;
  const long long ncr = (nc + 31) / 32 * 32;               // whole groups of eight lanes up to the end of the last wave's cells
  for (long long cell = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3; cell < ncr; cell += ((long long)gridDim.x * blockDim.x) >> 3) {
    const bool on = cell < nc;
    // fine index of the cell origin and the cell's extent per axis
    long long f = on ? cell : 0, stride = 1, origin = 0, len[kMaxD], fstride[kMaxD], total = 1;
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
    const long long nsub = on ? total / len[0] : 0;        // lines of <= kCoarse consecutive candidates along axis 0
    for (long long sline = sub; sline < nsub; sline += 8) {
      long long u = sline, g = origin;
      for (int a = 1; a < cg.d; ++a) {
        g += (u % len[a]) * fstride[a];
        u /= len[a];
      }
      double w[kCoarse];
#pragma unroll
      for (int k = 0; k < kCoarse; ++k) w[k] = k < (int)len[0] ? (double)W[g + k] : -1.0;
#pragma unroll
      for (int k = 0; k < kCoarse; ++k) wmax = w[k] > wmax ? w[k] : wmax;
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      const double y = __shfl_xor(wmax, o);
      wmax = y > wmax ? y : wmax;
    }
    if (on && sub == 0) {
      double v = kInfD;
      if (wmax >= 0.0) { const double r = wmax * invL; v = -(r * r); }
      lo[cell] = v;
      hi[cell] = v;
    }
  }
}
// one axis of both coarse bound transforms (lower-bound costs on the first array, upper-bound costs on the second)
// (r03: eight lanes per cell and bound, lane j takes the offsets t = j, j + 8, .. -- a thread per cell walked up to `cnt` offsets
// with two dependent loads each, 13 us on config C's 128 x 128 cells; the minimum over the offsets inside the band is the same)
__global__ __launch_bounds__(256) void k_pdt_coarse_scan(const double* __restrict__ LoIn, double* __restrict__ LoOut,
                                                         const double* __restrict__ HiIn, double* __restrict__ HiOut, long long nc,
                                                         long long stride, int cnt, double h, const SweepScalars* sc, int c,
                                                         const unsigned long long* Lkeys, int lidx, int d, double xscale) {
  const PdtParams pp = pdt_params(sc, c, Lkeys, lidx, d, xscale);
  const int sub = threadIdx.x & 7;
  for (long long g2 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 3; g2 < 2 * nc; g2 += ((long long)gridDim.x * blockDim.x) >> 3) {
    const int upper = g2 >= nc;                          // first half of the index space: lower bounds, second half: upper
    const long long g = upper ? g2 - nc : g2;
    const double* Pin = upper ? HiIn : LoIn;
    const int ia = (int)((g / stride) % cnt);
    double best = kInfD;
    for (int t = sub; t < cnt; t += 8) {
      const double steps_lo = t == 0 ? 0.0 : (double)((t - 1) * kCoarse + 1);
      const double steps = upper ? (double)((t + 1) * kCoarse - 1) : steps_lo;
      const double dl = h * steps_lo, dd = h * steps;
      if (dl * dl - pp.rmax2 > pp.band) break;           // no source that far can bring any candidate below the band
      if (dd * dd - pp.rmax2 >= best) break;             // nor improve this lane's minimum (the values are >= -rmax2)
      if (ia - t < 0 && ia + t >= cnt) break;
      const double c1 = ia - t >= 0 ? Pin[g - (long long)t * stride] : kInfD;
      const double c2 = ia + t < cnt ? Pin[g + (long long)t * stride] : kInfD;
      const double cnd = (c1 < c2 ? c1 : c2) + dd * dd;
      if (cnd < best) best = cnd;
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      const double y = __shfl_xor(best, o);
      best = y < best ? y : best;
    }
    if (sub == 0) (upper ? HiOut : LoOut)[g] = best;
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

// The same pass with one workgroup per grid line, the line's weights (count0 <= 8192 doubles) in LDS, and the search
// cut down by the structure of the problem: every parabola (h0 (i - j))^2 - r_j^2 has the same curvature, so the
// (leftmost) minimising source j*(i) never moves left as i moves right -- also inside the sliding window of the global
// reach, see DESIGN.md.  Phase A finds j* exactly for an anchor every 32 positions (one wave per anchor, the window
// spread over its lanes); phase B gives every position between two anchors the range [j*(left), j*(right)], on average
// a few dozen sources instead of the ~1000 inside the reach.  Values agree with the exhaustive search up to rounding
// between near-ties (1e-16, far inside the band that sends a verdict to the exact recheck).
constexpr int kAnchor = 32;
template <typename T>
__global__ __launch_bounds__(256) void k_pdt_axis0_lds(const T* __restrict__ W, long long nlines, int count0, double h0,
                                                       const SweepScalars* sc, int c, const unsigned long long* Lkeys, int lidx,
                                                       int d, double xscale, const CoarseGrid cg, const double* __restrict__ PcLo,
                                                       int blk, double* __restrict__ P) {
  extern __shared__ double lds_w[];            // [count0] weights as double | [ngap] block maxima | [nanch] int argmins | [nanch] int gap flags
  const PdtParams pp = pdt_params(sc, c, Lkeys, lidx, d, xscale);
  const int ngap = (count0 + kAnchor - 1) / kAnchor, nanch = ngap + 1;      // anchor m sits at min(32 m, count0 - 1)
  double* Wl = lds_w;
  double* Bl = lds_w + count0;                 // largest weight of every block of 32 positions (< 0: no source in it)
  int* jstar = reinterpret_cast<int*>(Bl + ngap);
  int* gap_on = jstar + nanch;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  // global reach in steps: sources further away have (h0 t)^2 - rmax2 > band and cannot bring a value below the band
  const double rsteps = sqrt(pp.rmax2 + pp.band) / h0 + 2.0;
  const int Rg = rsteps < (double)count0 ? (int)rsteps : count0;
  auto anchor_pos = [&](int m) { return m * kAnchor < count0 - 1 ? m * kAnchor : count0 - 1; };
  for (long long line = blockIdx.x; line < nlines; line += gridDim.x) {
    const long long g0 = line * count0;
    __syncthreads();                           // the previous line's scans are done with the buffers
    for (int i = threadIdx.x; i < count0; i += blockDim.x) Wl[i] = (double)W[g0 + i];
    long long cell0 = 0;                         // coarse cell of the line's first position (uniform per workgroup)
    if (cg.enabled) {
      long long f = line, ccs = cg.ccount[0];
      for (int a = 1; a < cg.d; ++a) {
        const long long ix = a == cg.d - 1 ? f : f % cg.count[a];
        f = a == cg.d - 1 ? 0 : f / cg.count[a];
        cell0 += (ix / kCoarse) * ccs;
        ccs *= cg.ccount[a];
      }
    }
    __syncthreads();
    for (int bb = threadIdx.x; bb < ngap; bb += blockDim.x) {
      const int j1 = (bb + 1) * kAnchor < count0 ? (bb + 1) * kAnchor : count0;
      double mx = -1.0;
      for (int j = bb * kAnchor; j < j1; ++j) mx = Wl[j] > mx ? Wl[j] : mx;
      Bl[bb] = mx;
    }
    for (int m = threadIdx.x; m < ngap; m += blockDim.x) {       // does gap m = [32 m, 32 m + 32) hold an active position?
      int on = !cg.enabled;
      for (int cc = 0; cc < kAnchor / kCoarse && !on; ++cc) {
        const int i = m * kAnchor + cc * kCoarse;
        if (i < count0 && !(PcLo[cell0 + i / kCoarse] > pp.band)) on = 1;
      }
      gap_on[m] = on;
    }
    __syncthreads();
    // phase A: exact leftmost minimiser of every anchor next to an active gap, in two levels -- every eighth anchor (and
    // the last one) searches its whole window, the anchors between two of those only between their minimisers
    auto scan_anchor = [&](int m, int lo, int hi) {       // one wave: the range spread over its lanes
      const int a = anchor_pos(m);
      double best = kInfD;
      int bj = -1;
      for (int j0 = (lo / 64) * 64; j0 <= hi; j0 += 64) {          // two blocks per round, empty pairs skipped (uniform)
        const int b0 = j0 / kAnchor;
        if (!(Bl[b0] >= 0.0) && !(b0 + 1 < ngap && Bl[b0 + 1] >= 0.0)) continue;
        const int j = j0 + lane;
        if (j < lo || j > hi) continue;
        const double wj = Wl[j];
        if (wj >= 0.0) {
          const double dt = h0 * (double)(j > a ? j - a : a - j), r = wj * pp.invL;
          const double cnd = dt * dt - r * r;
          if (cnd < best) { best = cnd; bj = j; }
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const int oj = __shfl_xor(bj, off);
        if (oj >= 0 && (bj < 0 || ob < best || (ob == best && oj < bj))) { best = ob; bj = oj; }
      }
      if (lane == 0) jstar[m] = bj;
    };
    int any_on = 0;
    for (int m = threadIdx.x; m < ngap; m += blockDim.x) any_on |= gap_on[m];
    if (!__syncthreads_or(any_on)) {           // nothing on this line can be covered
      for (int i = threadIdx.x; i < count0; i += blockDim.x) P[g0 + i] = kInfD;
      continue;
    }
    constexpr int kTop = 8;
    const int ntop = (nanch - 1 + kTop - 1) / kTop + 1;
    for (int k = wave; k < ntop; k += nwave) {
      const int m = k * kTop < nanch - 1 ? k * kTop : nanch - 1;
      const int a = anchor_pos(m);
      scan_anchor(m, a - Rg > 0 ? a - Rg : 0, a + Rg < count0 - 1 ? a + Rg : count0 - 1);
    }
    __syncthreads();
    for (int m = wave; m < nanch - 1; m += nwave) {
      if (m % kTop == 0) continue;
      if (!(gap_on[m] || gap_on[m - 1])) continue;                 // (uniform per wave)
      const int m0 = (m / kTop) * kTop, m1 = m0 + kTop < nanch - 1 ? m0 + kTop : nanch - 1;
      const int a = anchor_pos(m);
      const int jl = jstar[m0] >= 0 ? jstar[m0] : anchor_pos(m0) - Rg, jr = jstar[m1] >= 0 ? jstar[m1] : anchor_pos(m1) + Rg;
      int lo = a - Rg > jl ? a - Rg : jl, hi = a + Rg < jr ? a + Rg : jr;
      lo = lo > 0 ? lo : 0;
      hi = hi < count0 - 1 ? hi : count0 - 1;
      scan_anchor(m, lo, hi);
    }
    __syncthreads();
    // phase B: every active position searches between the minimisers of its two anchors
    for (int i = threadIdx.x; i < count0; i += blockDim.x) {
      const long long g = g0 + i;
      if (cg.enabled && PcLo[cell0 + i / kCoarse] > pp.band) { P[g] = kInfD; continue; }
      const int m = i / kAnchor;
      const int a0 = anchor_pos(m), a1 = anchor_pos(m + 1);
      const int jl = jstar[m] >= 0 ? jstar[m] : a0 - Rg, jr = jstar[m + 1] >= 0 ? jstar[m + 1] : a1 + Rg;
      int lo = i - Rg > jl ? i - Rg : jl, hi = i + Rg < jr ? i + Rg : jr;
      lo = lo > 0 ? lo : 0;
      hi = hi < count0 - 1 ? hi : count0 - 1;
      double best = kInfD;
      for (int b = lo / kAnchor; b <= hi / kAnchor; ++b) {
        if (!(Bl[b] >= 0.0)) continue;             // no source in this block
        const int j0 = b * kAnchor > lo ? b * kAnchor : lo, j1 = b * kAnchor + kAnchor - 1 < hi ? b * kAnchor + kAnchor - 1 : hi;
        for (int j = j0; j <= j1; j += 4) {        // four reads in flight, branch-free
          double cnd[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int jj = j + u <= j1 ? j + u : j1;
            const double wj = Wl[jj];
            const double dt = h0 * (double)(jj > i ? jj - i : i - jj), r = wj * pp.invL;
            cnd[u] = wj >= 0.0 ? dt * dt - r * r : kInfD;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) best = cnd[u] < best ? cnd[u] : best;
        }
      }
      P[g] = best;
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
  auto scan_block = [&](int b) {       // (eight loads in flight at a time: the chain of loads is what a thread waits for)
    const int j1 = (b + 1) * blk < cnt ? (b + 1) * blk : cnt;
    for (int j = b * blk; j < j1; j += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = j + u < j1 ? Pin[(long long)(j + u) * stride + p] : kInfD;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int jj = j + u;
        const double dt = h * (double)(jj > ia ? jj - ia : ia - jj);
        const double cnd = v[u] + dt * dt;
        best = cnd < best ? cnd : best;
      }
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

// last axis + verdict for the own U points (window offset goff); ambiguous ones are listed for the exact recheck.
// With a scan list (grids with block minima) the points the coarse bounds leave open are collected in LDS, appended
// with one atomic per workgroup and scanned by k_pdt_scan_list; otherwise by their own thread.
__global__ __launch_bounds__(256) void k_pdt_decide(const double* __restrict__ Pin, long long nl, int len0, long long line0,
                                                    long long goff, long long stride,
                                                    int cnt, double h, int d, double xscale, const uint8_t* __restrict__ U,
                                                    const unsigned long long* Lkeys, int lidx, SweepScalars* sc, int c,
                                                    uint8_t* __restrict__ O, long long* __restrict__ amb, const CoarseGrid cg,
                                                    const double* __restrict__ PcLo, const double* __restrict__ PcHi,
                                                    const double* __restrict__ Bmin, int blk, long long* __restrict__ scanlist) {
  __shared__ long long sl[256 * kDecideLines];
  __shared__ int scnt;
  __shared__ long long sbase;
  const PdtParams pp = pdt_params(sc, c, Lkeys, lidx, d, xscale);
  const bool anyS = sc->count_S > 0;
  const bool listing = scanlist != nullptr && Bmin != nullptr && cnt > 1;
  // launch as k_edt_decide: x over the positions of a grid line, y over blocks of kDecideLines local lines
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i0 < len0;
  const long long nlb = (nl + kDecideLines - 1) / kDecideLines;
  for (long long lb = blockIdx.y; lb < nlb; lb += gridDim.y) {
    if (threadIdx.x == 0) scnt = 0;
    __syncthreads();
    for (long long ln = lb * kDecideLines; active && ln < nl && ln < (lb + 1) * kDecideLines; ++ln) {
      const long long g = ln * len0 + i0;
      uint8_t out = 0;
      if (U[g]) {
        if (!pp.L_positive) {
          out = anyS;                                  // radius unbounded: any source covers (ucb_c >= 0 on every source)
        } else {
          const long long gg = goff + g;
          long long f = line0 + ln, cell = i0 / kCoarse, ccs = cg.ccount[0];
          int ia = 0;                                  // index along the last axis (inside the window)
          for (int a = 1; a < d; ++a) {
            const long long ix = a == d - 1 ? f : f % cg.count[a];
            f = a == d - 1 ? 0 : f / cg.count[a];
            cell += (ix / kCoarse) * ccs;
            ccs *= cg.ccount[a];
            ia = (int)ix;
          }
          if (cg.enabled) {
            if (PcHi[cell] < -pp.band) { O[g] = 1; continue; }     // covered wherever it sits in its cell
            if (PcLo[cell] > pp.band) { O[g] = 0; continue; }      // out of every source's reach
          }
          if (listing) {
            sl[atomicAdd(&scnt, 1)] = g;
            O[g] = 0;
            continue;
          }
          const double best = cnt <= 1 ? Pin[gg]
                              : Bmin  ? pdt_scan_blocked(Pin, Bmin, gg - (long long)ia * stride, stride, cnt, ia, h, pp, blk)
                                      : pdt_scan_point(Pin, gg, stride, cnt, ia, h, pp, true);
          if (best < -pp.band) out = 1;
          else if (best <= pp.band) {
            // (guard band: k_goose_exact judges whether the band could move the listed point's verdict -- nearly every point listed
            // here sits inside the reference's 1e-8 shift, not inside the band, and a counted decision re-evaluates ALL of S)
            const long long slot = (long long)atomicAdd((unsigned long long*)&sc->n_amb, 1ull);
            amb[slot] = g;
          }
        }
      }
      O[g] = out;
    }
    __syncthreads();
    const int cntl = scnt;
    if (cntl > 0) {
      if (threadIdx.x == 0) sbase = (long long)atomicAdd((unsigned long long*)&sc->n_scan, (unsigned long long)cntl);
      __syncthreads();
      for (int k = threadIdx.x; k < cntl; k += blockDim.x) scanlist[sbase + k] = sl[k];
    }
    __syncthreads();
  }
}

// Last-axis scan + verdict of the listed U points, one group of GL lanes per point (the power-transform counterpart of
// k_edt_scan_list): the lanes take GL block bounds, or GL steps of a block, at a time and combine with group minima --
// the candidates and arithmetic of pdt_scan_blocked, its chain of dependent loads cut to a handful.
template <int GL>
__global__ __launch_bounds__(256) void k_pdt_scan_list(const double* __restrict__ Pin, long long goff, long long stride, int cnt,
                                                       double h, int d, double xscale, const unsigned long long* Lkeys, int lidx,
                                                       SweepScalars* sc, int c, uint8_t* __restrict__ O, long long* __restrict__ amb,
                                                       const double* __restrict__ Bmin, int blk,
                                                       const long long* __restrict__ scanlist) {
  const PdtParams pp = pdt_params(sc, c, Lkeys, lidx, d, xscale);
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
    if (GL == 64) return m;
    return (m >> ((GL & 63) * sub)) & ((1ull << (GL & 63)) - 1ull);
  };
  // blocks further than this cannot hold a source that brings a value below the band
  const double reach_steps = sqrt(pp.rmax2 + pp.band) / h;
  const int kmax = (int)fmin((double)nblk, floor(reach_steps / (double)blk) + 2.0);
  for (long long qi = (long long)blockIdx.x * (blockDim.x / GL) + threadIdx.x / GL; qi < nscan; qi += ngroups) {
    const long long g = scanlist[qi];
    const long long gg = goff + g, p = gg % stride;
    const int ia = (int)((gg / stride) % cnt), b0 = ia / blk;
    double best = Pin[(long long)ia * stride + p];
    const int blo = b0 - kmax > 0 ? b0 - kmax : 0, bhi = b0 + kmax < nblk - 1 ? b0 + kmax : nblk - 1;
    auto bound_of = [&](int bb) {       // bound of block bb for this lane (inf outside the reach / the axis)
      if (bb < blo || bb > bhi) return kInfD;
      const int gap = bb == b0 ? 0 : (bb < b0 ? ia - (bb * blk + blk - 1) : bb * blk - ia);
      const double dg = h * (double)gap;
      const double e = dg * dg;
      if (e - pp.rmax2 > pp.band) return kInfD;
      return Bmin[(long long)bb * stride + p] + e;
    };
    auto scan_block = [&](int bb) {     // the group: the steps of block bb, GL at a time
      double cnd = kInfD;
#pragma unroll 4
      for (int s0 = 0; s0 < blk; s0 += GL) {
        const int jn = bb * blk + s0 + lane;
        if (s0 + lane < blk && jn < cnt) {
          const double dt = h * (double)(jn > ia ? jn - ia : ia - jn);
          const double v = Pin[(long long)jn * stride + p] + dt * dt;
          cnd = v < cnd ? v : cnd;
        }
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
    // pass B: every other block whose bound still beats the running minimum, until the verdict is sure
    for (int base = blo; base <= bhi && !(best < -pp.band); base += GL) {
      const double lb = bound_of(base + lane);
      unsigned long long todo = group_ballot(lb < best && base + lane != b_min);
      while (todo && !(best < -pp.band)) {
        const int l = (int)(__ffsll((long long)todo) - 1);
        todo &= todo - 1;
        const double lbl = __shfl(lb, l + GL * sub);
        if (lbl < best) scan_block(base + l);
      }
    }
    if (lane == 0) {
      uint8_t out = 0;
      if (best < -pp.band) out = 1;
      else if (best <= pp.band) {
        amb[atomicAdd((unsigned long long*)&sc->n_amb, 1ull)] = g;     // (guard band: judged by k_goose_exact)
      }
      O[g] = out;
    }
  }
}

// exact recheck of the listed U points: the reference predicate against every source inside the index box that the
// largest radius can reach.  cs: the own candidates (h); css / W: the source candidates (own range or whole grid)
template <typename T, int D>
__global__ __launch_bounds__(256) void k_goose_exact(const CandSpec cs, const CandSpec css, const T* __restrict__ W,
                                                     const unsigned long long* Lkeys, int lidx, SweepScalars* sc, int c,
                                                     const long long* __restrict__ amb, uint8_t* __restrict__ O, int gband) {
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const long long namb = sc->n_amb;
  // guard band (fast path of an approximating posterior): a source's weight is known to dw = du_c + rl |w|; a listed point's verdict
  // is in the band iff it has a witness with w + dw and none with w - dw
  const double gdu = gband ? sc->gb_du[c] : 0.0, grl = gband ? sc->gb_rl[lidx] : 0.0;
  const double rm = (L > 0 && sc->rmax_key[c]) ? fmax(0.0, ord_val(sc->rmax_key[c])) / L : 0.0;
  // one listed point can own a box as large as the grid: its box is cut into kParts slices, one workgroup each
  constexpr int kParts = 256;
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
    int found = 0, found_hi = 0, found_lo = 0;
    const long long chunk = (total + kParts - 1) / kParts;
    const long long t1 = (part + 1) * chunk < total ? (part + 1) * chunk : total;
    for (long long t = part * chunk + threadIdx.x; t < t1 && !(gband ? found_lo : found); t += blockDim.x) {   // (band: until a SURE witness)
      // position in the box (32-bit divisions when the box allows), source weight first: most positions hold no source
      long long gg = 0, ia[D];
      if (total < (1ll << 31)) {
        unsigned int u = (unsigned int)t;
#pragma unroll
        for (int a = 0; a < D; ++a)
          if (a < cs.d) { const unsigned int la = (unsigned int)len[a]; ia[a] = lo[a] + u % la; u /= la; }
      } else {
        long long u = t;
#pragma unroll
        for (int a = 0; a < D; ++a)
          if (a < cs.d) { ia[a] = lo[a] + u % len[a]; u /= len[a]; }
      }
#pragma unroll
      for (int a = 0; a < D; ++a)
        if (a < cs.d) gg += ia[a] * stridea[a];
      const long long gl = gg - css.first;
      if (gl >= 0 && gl < css.n_local) {
        const double w = (double)W[gl];
        if (w >= 0.0) {
          double xg[D];
#pragma unroll
          for (int a = 0; a < D; ++a) {
            xg[a] = 0.0;
            if (a < cs.d) {
              const long long cnt = cs.count[a];
              xg[a] = (ia[a] == cnt - 1 && cnt > 1) ? cs.hi[a] : __dadd_rn(cs.lo[a], __dmul_rn((double)ia[a], cs.step[a]));
            }
          }
          if (lipschitz_pair<D>(xg, xh, cs.d, w, L)) found = 1;
          if (gband) {
            const double dw = gdu + grl * w;
            if (lipschitz_pair<D>(xg, xh, cs.d, w + dw, L)) found_hi = 1;
            if (lipschitz_pair<D>(xg, xh, cs.d, w - dw, L)) found_lo = 1;
          }
        }
      }
    }
    found = __syncthreads_or(found);
    if (gband) {
      found_hi = __syncthreads_or(found_hi);
      found_lo = __syncthreads_or(found_lo);
      if (threadIdx.x == 0 && found_hi && !found_lo) atomicAdd((unsigned long long*)&sc->n_guard, 1ull);
    }
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

// Single rank: the target of the explore step chosen on the device (min over the constraints' targets, the first minimum
// wins -- models/GoOSE.py:110-112) and its coordinates parked for k_dist_to, so the sweep needs one host round trip.
template <int D>
__global__ void k_pick_target(const CandSpec cs, const SweepScalars* sc, int q, double* __restrict__ target) {
  if (blockIdx.x || threadIdx.x) return;
  int best_c = 0;
  double best = 0.0;
  for (int cc = 1; cc < q; ++cc)
    if (sc->arg_idx[cc] >= 0 && (best_c == 0 || sc->arg_val[cc] < best)) { best_c = cc; best = sc->arg_val[cc]; }
  double x[D];
#pragma unroll
  for (int a = 0; a < D; ++a) x[a] = 0.0;
  if (best_c) cand_coords<D>(cs, sc->arg_idx[best_c] - cs.first, x);
#pragma unroll
  for (int a = 0; a < D; ++a) target[a] = x[a];
}

