// sets_colpath.inc.hpp: the set phase of a one-constraint SafeOpt sweep on COLUMN WORDS (r05) -- part of the sets.hip translation
// unit (included inside namespace sbo; not a standalone header).
//
// Why: on config H (4096^2) the byte-mask pipeline took 0.24 ms in seven dependent launches for 0.26 GB of traffic -- every kernel
// near its latency floor, none near the HBM roof.  Two things made the passes long: 16.8 MB byte masks read and written by
// every kernel, and an exact distance transform whose per-candidate search runs ACROSS memory (its first pass goes along the
// contiguous axis 0, so the search for the nearest fully-unsafe point walks axis 1: ~130 scattered cache lines per open
// candidate).  Here
//   * the masks are 64-bit column words (internal.hpp: ColBits), written by the posterior kernel's own epilogue: 2 MB each;
//   * the first pass of the transform runs ALONG axis 1 (down the columns -- a word holds exactly 64 rows of one column, so a
//     lane owns a column and walks its word bit by bit), the image is t(i, j) = steps along column i from row j to the nearest
//     U point, and the per-candidate search then runs along axis 0: 32 consecutive 16-bit entries of one row are one 64-byte
//     line, the block minima of a whole row two lines;
//   * u* comes out of the objective's posterior tiles, the arg-max partials of G out of the verdict kernels themselves.
// Launches behind K1: k_col_a (merge of the classification's rows | coarse axis-0 pass | column pass: image + block minima),
// k_col_mid (coarse scan | minimiser), k_col_decide, k_col_scan, k_col_finals.  The decisions are those of the byte-mask path bit for
// bit: same predicates, same conservative pre-tests, the same exhaustive recheck inside the reference's "+1e-8" band.
#pragma once

struct ColGeom {
  int W, H, NS, NB;          // columns (axis 0), rows (axis 1), 64-row segments, 32-column blocks per row
  int NBp;                   // row stride of the block minima (NB rounded up to 8: a lane's eight blocks are one aligned 16-byte load)
  int CW, CH;                // coarse cells per axis (8 x 8 candidates)
  int CNB, CNBp, CWp;        // 32-cell blocks per coarse row, row strides of the coarse block minima / the coarse image (padded)
  int gxt;                   // k_bpost tiles per tile row (W / 128)
  double h0, h1;
  double delta;              // sandwich of the coarse transform: dC - delta <= dist(g, U) <= dC + delta (sets_expander.inc.hpp)
};
constexpr int kColBig = 1 << 20;       // "no U point on this side of the column" (far above 65535 through 64 increments)

// Column pass of one (segment, 64 columns) item by one wave: lane = column.  The carries above / below the segment come from Usum
// (which segments of the column hold a U point) and one more word each; the lane then walks its word as edt_axis0_wave_seq walks
// a row word -- steps since the last U point going down the rows, steps to the next one going up, three 32-bit instructions per
// step -- and drops t = min of the two (capped at 0xffff = none) into the wave's 64 x 64 LDS tile, from which the image rows
// leave as 16-byte stores (eight rows of 128 bytes per instruction) and the two 32-column block minima of every row are taken.
typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void col_fine_item(long long item, const ColGeom gm, const ColBits cb, unsigned short* __restrict__ tile,
                                              unsigned short* __restrict__ img, unsigned short* __restrict__ bmin,
                                              unsigned long long* __restrict__ Mw, unsigned long long* __restrict__ Gw) {
  const int lane = threadIdx.x & 63;
  const int ncg = gm.W >> 6;
  const int s = (int)(item / ncg), cgp = (int)(item % ncg);
  const int i = cgp * 64 + lane;
  // (the M / G words start from zero: the kernels that set their bits walk the tiles with a safe candidate only)
  Mw[(size_t)s * gm.W + i] = 0ull;
  Gw[(size_t)s * gm.W + i] = 0ull;
  const unsigned long long m = cb.Uw[(size_t)s * gm.W + i];
  const unsigned long long us = cb.Usum[i];
  const unsigned long long below = us & ((1ull << s) - 1ull);
  const unsigned long long above = s < 63 ? (us >> (s + 1)) : 0ull;
  int f = kColBig, dn = kColBig;
  if (below) {
    const int sp = 63 - __clzll((long long)below);
    const unsigned long long w = cb.Uw[(size_t)sp * gm.W + i];
    if (w) f = 64 * s - 1 - (64 * sp + 63 - __clzll((long long)w));
  }
  if (above) {
    const int sn = s + __ffsll((long long)above);
    const unsigned long long w = cb.Uw[(size_t)sn * gm.W + i];
    if (w) dn = (64 * sn + __ffsll((long long)w) - 1) - (64 * s + 64);
  }
  const unsigned int nlo = ~(unsigned int)m, nhi = ~(unsigned int)(m >> 32);     // bit clear -> 1
  int F[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const unsigned int h = r < 32 ? nlo : nhi;
    const int keep = (int)(h << (31 - (r & 31))) >> 31;        // -1 when row r of the segment holds no U point in this column
    f = (f + 1) & keep;
    F[r] = f;
  }
#pragma unroll
  for (int r = 63; r >= 0; --r) {
    const unsigned int h = r < 32 ? nlo : nhi;
    const int keep = (int)(h << (31 - (r & 31))) >> 31;
    dn = (dn + 1) & keep;
    int t = F[r] < dn ? F[r] : dn;
    t = t > 0xffff ? 0xffff : t;
    tile[r * 64 + lane] = (unsigned short)t;
  }
  __builtin_amdgcn_wave_barrier();
  // image rows: store k moves rows 8 k .. 8 k + 7, lane l the 16 bytes (8 columns) l & 7 of row 8 k + (l >> 3)
  {
    const int rsub = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = 8 * k + rsub;
      *reinterpret_cast<uint4*>(img + ((size_t)(64 * s + r)) * gm.W + (size_t)cgp * 64 + ch * 8) = *reinterpret_cast<const uint4*>(tile + r * 64 + ch * 8);
    }
  }
  // block minima: 128 (row, half) pairs, two per lane; lanes 2 m and 2 m + 1 hold the two halves of one row -> one 4-byte store
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int p = lane + 64 * k, r = p >> 1, half = p & 1;
    const uint4* src = reinterpret_cast<const uint4*>(tile + r * 64 + half * 32);
    us2_t mn = us2_t{0xffff, 0xffff};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint4 q4 = src[u];
      const unsigned int ws[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        us2_t v;
        v[0] = (unsigned short)(ws[e] & 0xffffu);
        v[1] = (unsigned short)(ws[e] >> 16);
        mn = __builtin_elementwise_min(mn, v);
      }
    }
    const unsigned int mine = mn[0] < mn[1] ? mn[0] : mn[1];
    const unsigned int other = (unsigned int)__shfl_down((int)mine, 1);
    if (half == 0) *reinterpret_cast<unsigned int*>(bmin + ((size_t)(64 * s + r)) * gm.NBp + (size_t)cgp * 2) = mine | (other << 16);
  }
  __builtin_amdgcn_wave_barrier();                       // (the wave's next item overwrites the tile)
}

// The same transform on the 8 x 8 cells (the coarse sandwich of the verdicts): a workgroup takes 32 coarse columns, thread
// (column, 64-row coarse segment) ORs the bits of its 8 x 64 cells out of the fine column words, finds its carries in the other
// segments' words (LDS), walks its word, and the 32 columns of the workgroup are exactly one block of the coarse rows' block minima.
// The per-cell distances themselves are searched where they are needed (k_col_decide: col_row_search<2>): the separate coarse scan of
// the byte-mask path -- 2 x 190 steps per far cell, 26 us on config H -- is gone.
__device__ __forceinline__ void col_coarse_job(int blk, const ColGeom gm, const unsigned long long* __restrict__ Uw, unsigned long long* lds_words,
                                               unsigned short* __restrict__ cimg, unsigned short* __restrict__ cbmin) {
  const int cl = threadIdx.x & 31, Sg = threadIdx.x >> 5;
  const int ci = blk * 32 + cl;
  const int CNS = (gm.CH + 63) >> 6;
  unsigned long long word = 0ull;
  if (ci < gm.CW && Sg < CNS) {
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int s = 8 * Sg + f;
      if (s < gm.NS) {
        const unsigned long long* w = Uw + (size_t)s * gm.W + (size_t)ci * 8;
        unsigned long long x = 0ull;
#pragma unroll
        for (int k = 0; k < 8; ++k) x |= w[k];
        x |= x >> 4;
        x |= x >> 2;
        x |= x >> 1;
        x &= 0x0101010101010101ull;
        word |= ((x * 0x0102040810204080ull) >> 56) << (8 * f);
      }
    }
  }
  lds_words[Sg * 32 + cl] = word;
  __syncthreads();
  if (Sg >= CNS) return;
  int f = kColBig, dn = kColBig;
  for (int sp = Sg - 1; sp >= 0; --sp) {
    const unsigned long long w = lds_words[sp * 32 + cl];
    if (w) { f = 64 * Sg - 1 - (64 * sp + 63 - __clzll((long long)w)); break; }
  }
  for (int sn = Sg + 1; sn < CNS; ++sn) {
    const unsigned long long w = lds_words[sn * 32 + cl];
    if (w) { dn = (64 * sn + __ffsll((long long)w) - 1) - (64 * Sg + 64); break; }
  }
  const unsigned int nlo = ~(unsigned int)word, nhi = ~(unsigned int)(word >> 32);
  int F[64];
#pragma unroll
  for (int r = 0; r < 64; ++r) {
    const unsigned int h = r < 32 ? nlo : nhi;
    const int keep = (int)(h << (31 - (r & 31))) >> 31;
    f = (f + 1) & keep;
    F[r] = f;
  }
  unsigned int pk[32];
#pragma unroll
  for (int r = 63; r >= 0; --r) {
    const unsigned int h = r < 32 ? nlo : nhi;
    const int keep = (int)(h << (31 - (r & 31))) >> 31;
    dn = (dn + 1) & keep;
    int t = F[r] < dn ? F[r] : dn;
    t = t > 0xffff ? 0xffff : t;
    cimg[(size_t)(64 * Sg + r) * gm.CWp + ci] = (unsigned short)t;
    if (r & 1) pk[r >> 1] = (unsigned int)t << 16;
    else pk[r >> 1] |= (unsigned int)t;
  }
  // the block minimum of every row over the workgroup's 32 columns (the lanes of one half of a wave), two rows per word
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    us2_t v;
    v[0] = (unsigned short)(pk[k] & 0xffffu);
    v[1] = (unsigned short)(pk[k] >> 16);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const unsigned int y = (unsigned int)__shfl_xor((int)((unsigned int)v[0] | ((unsigned int)v[1] << 16)), o);
      us2_t w2;
      w2[0] = (unsigned short)(y & 0xffffu);
      w2[1] = (unsigned short)(y >> 16);
      v = __builtin_elementwise_min(v, w2);
    }
    if (cl == k) {
      cbmin[(size_t)(64 * Sg + 2 * k) * gm.CNBp + blk] = v[0];
      cbmin[(size_t)(64 * Sg + 2 * k + 1) * gm.CNBp + blk] = v[1];
    }
  }
}

// The merge of the classification's partial rows, in two halves (the sweep runs as two chains: sets_colpath's host part).
// Half 1, behind the CONSTRAINT's posterior launch: the scalar block cleared, |S|, |U|, the radius key and the smallest variance of the
// constraint over S, the sign tests inside the guard band, the constraint's Lipschitz key -- everything the expander chain reads.
// Rows [0, nrows): one per constraint tile, field-major (sets.hip: kClassifyRow).
// one wave reduces a field of the slot block (internal.hpp: ColSlotField): a load per lane, six shuffle steps; valid in every lane
__device__ __forceinline__ unsigned long long col_slot_reduce(const unsigned long long* __restrict__ slots, int field) {
  unsigned long long v = slots[(size_t)field * kColSlots + (threadIdx.x & 63)];
  const bool is_min = col_slot_is_min(field), is_sum = field == kSlotS || field == kSlotU || field == kSlotB;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long y = __shfl_xor(v, o);
    v = is_sum ? v + y : (is_min ? (y < v ? y : v) : (y > v ? y : v));
  }
  return v;
}
__device__ __forceinline__ void col_merge1_body(const unsigned long long* __restrict__ slots, SweepScalars* sc, unsigned long long* Lmax,
                                                const GuardBand* gb, double b) {
  unsigned long long* w = reinterpret_cast<unsigned long long*>(sc);
  for (unsigned i = threadIdx.x; i < sizeof(SweepScalars) / 8; i += blockDim.x) w[i] = 0ull;
  __syncthreads();
  if (threadIdx.x < 64) {
    const unsigned long long cS = col_slot_reduce(slots, kSlotS), cU = col_slot_reduce(slots, kSlotU), cB = col_slot_reduce(slots, kSlotB);
    const unsigned long long vmin = col_slot_reduce(slots, kSlotVmin1), rmax = col_slot_reduce(slots, kSlotRmax1), l1 = col_slot_reduce(slots, kSlotL1);
    if (threadIdx.x == 0) {
      Lmax[1] = l1;
      sc->ustar_key = ~0ull;                       // (the objective's half: k_col_min, into its own block)
      sc->count_S = (long long)cS;
      sc->count_U = (long long)cU;
      sc->rmax_key[1] = rmax;
      if (gb) {
        sc->n_guard = (long long)cB;
        sc->n_guard_cls = (long long)cB;
        sc->vmin_key[1] = vmin;
        const bool any = vmin != ~0ull;
        const double vm = any ? fmax(0.0, ord_val(vmin)) : 0.0;
        sc->gb_du[1] = gb->dm[1] + (any ? b * gb_dsqrt(vm, gb->dv[1]) : 0.0);
        sc->gb_rl[0] = gb->rl[0];
        sc->gb_rl[1] = gb->rl[1];
      }
    }
  }
  if (threadIdx.x >= 64 && threadIdx.x < 64 + kArgSlots) sc->arg_idx[threadIdx.x - 64] = -1;
}
// Half 2, behind the OBJECTIVE's launch (rows [0, nrows) of ITS tiles): u* = min over S of ucb_0 and the smallest var_0 over S.  Every
// workgroup of the minimiser merges them for itself (16 KB of keys, two coalesced loads per thread: no launch in between); the values
// are valid in every thread.
struct ColScal2 {            // the objective's scalars, in their own block (the expander chain clears and fills SweepScalars concurrently)
  unsigned long long ustar_key, vmin0_key;
  double gb_du0;
};
__device__ __forceinline__ void col_merge2_body(const unsigned long long* __restrict__ slots, unsigned long long& ustar_key,
                                                unsigned long long& vmin0_key) {
  ustar_key = col_slot_reduce(slots, kSlotUmin);
  vmin0_key = col_slot_reduce(slots, kSlotVmin0);
}

struct ColMergeJob {
  const unsigned long long* slots;
  SweepScalars* sc;
  unsigned long long* Lmax;
  const GuardBand* gb;
  double b;
};
// first launch of the expander chain: [1: merge, half 1 | ncoarse: coarse axis-0 pass | the rest: column pass]
__global__ __launch_bounds__(256) void k_col_a(const ColGeom gm, const ColBits cb, const ColMergeJob mg, int ncoarse, unsigned short* __restrict__ cimg,
                                               unsigned short* __restrict__ cbmin, unsigned short* __restrict__ img, unsigned short* __restrict__ bmin,
                                               unsigned long long* __restrict__ Mw, unsigned long long* __restrict__ Gw) {
  constexpr size_t kTileBytes = 4 * 64 * 64 * sizeof(unsigned short);
  __shared__ __attribute__((aligned(16))) unsigned char mem[kTileBytes];
  const int bid = (int)blockIdx.x;
  if (bid == 0) {
    col_merge1_body(mg.slots, mg.sc, mg.Lmax, mg.gb, mg.b);
  } else if (bid < 1 + ncoarse) {
    col_coarse_job(bid - 1, gm, cb.Uw, reinterpret_cast<unsigned long long*>(mem), cimg, cbmin);
  } else {
    const int nfine = (int)gridDim.x - 1 - ncoarse;
    unsigned short* tile = reinterpret_cast<unsigned short*>(mem) + (threadIdx.x >> 6) * 64 * 64;
    const long long nitems = (long long)gm.NS * (gm.W >> 6);
    for (long long item = (long long)(bid - 1 - ncoarse) * 4 + (threadIdx.x >> 6); item < nitems; item += (long long)nfine * 4)
      col_fine_item(item, gm, cb, tile, img, bmin, Mw, Gw);
  }
}
// bounds of lcb = fl(m - fl(b fl(sqrt v))) from a single-precision square root (as ucb_lower / ucb_upper, device_common.hpp)
__device__ __forceinline__ void lcb_bounds(double m, double v, double b, double& lo, double& hi) {
  const float sf = __fsqrt_rn((float)v);
  const double s_up = (double)sf * (1.0 + 0x1p-20) + 1e-18;
  double s_lo = (double)sf * (1.0 - 0x1p-20) - 1e-18;
  s_lo = s_lo > 0.0 ? s_lo : 0.0;
  const double x = m - b * s_up, y = m - b * s_lo;
  lo = x - (x < 0 ? -x : x) * 0x1p-50;
  hi = y + (y < 0 ? -y : y) * 0x1p-50;
  if (!(sf < 3.0e38f)) { lo = -kInfD; hi = kInfD; }              // (inf / NaN: no information -- the exact bound decides)
}

// Single-precision pre-test of a bound m +- b sqrt(v) (the decisions of the verdict kernels and the minimiser for the nine
// candidates in ten that are nowhere near their threshold; ~18 two-cycle instructions where the fp64 forms of lcb_bounds / ucb_lower
// cost ~150 four-cycle ones -- these kernels are instruction-issue bound).  uf = mf + bf sf from operands rounded to float and the
// hardware's 1-ulp square root is within 4.8e-7 (|m| + b sqrt v) of the exact bound; `del` is twice that.  ok = false: no
// information (operands outside the comfortable float range) -- the fp64 path decides.
struct F32Bound {
  float u, amax, del;
  bool ok;
};
__device__ __forceinline__ F32Bound f32_bound(double m, double v, float bf /* +b: upper bound; -b: lower bound */) {
  const float mf = (float)m, sf = __builtin_amdgcn_sqrtf((float)v);
  F32Bound r;
  r.amax = fmaf(fabsf(bf), sf, fabsf(mf));
  r.u = fmaf(bf, sf, mf);
  r.del = r.amax * 1.0e-6f;
  r.ok = r.amax > 1.0e-25f && r.amax < 1.0e25f;
  return r;
}
// arg-max with ties to the lowest index (32-bit: a grid of the column path has at most 2^24 candidates), runner-up value in e2 when a
// guard band wants it; `on`: this candidate takes part
template <bool E2>
__device__ __forceinline__ void take_max(bool on, double v, unsigned int g, double& bv, unsigned int& bg, double& e2) {
  const bool c = on && (v > bv || (v == bv && g < bg));
  if (E2) {
    const double loser = c ? bv : (on ? v : -kInfD);
    e2 = loser > e2 ? loser : e2;
  }
  bv = c ? v : bv;
  bg = c ? g : bg;
}
__device__ __forceinline__ Best best_from(double bv, unsigned int bg, double e2) {
  Best b = best_none<true>();
  if (bg != 0xffffffffu) { b.v = bv; b.i = (long long)bg; b.e2 = e2; }
  return b;
}

// ---- the S-driven kernels walk UNITS: (tile of the list with a safe candidate, one of its eight 8-row octets) ----------------------
// A wave takes a unit: 8 rows x 128 columns, two adjacent columns per lane -- every load instruction of mean / var moves 1 KB of one
// row, and the lanes' 8 x 2 loads of a unit are all in flight at once.  (64 columns x 16 rows per wave, 512 bytes per row and 32
// rows of two arrays in flight, waited 5 us per batch for its loads on config H.)  The grid is sized to be resident at once and the
// waves loop: what a wave pays before its first unit -- kernel arguments, the sweep's scalars, a dependent chain of ~4 us -- it pays
// once (32768 short-lived waves paid it in eight rounds).
// The units in the order (tile row, octet, tile column): consecutive waves read the SAME eight rows of adjacent tiles -- their 1 KB
// pieces line up into long runs of the same DRAM pages (units of tiles in the order their workgroups happened to finish read
// 54 MB of mean / var at 2.4 TB/s).  No list: row s of the slot block's kSlotRowMask holds the bits of its tiles with a safe
// candidate; a wave keeps the rows' unit counts and their prefix sums in its lanes (lane = tile row) and finds a unit's place with a
// ballot and three broadcasts.
struct ColUnitMap {
  unsigned int mask;        // lane s: tiles of tile row s with a safe candidate
  int cnt, pre;             // 8 popc(mask); units of the rows before s
  long long total;
};
__device__ __forceinline__ ColUnitMap col_unit_map(const unsigned long long* __restrict__ slots, int NS) {
  const int lane = threadIdx.x & 63;
  ColUnitMap um;
  um.mask = lane < NS ? (unsigned int)slots[(size_t)kSlotRowMask * kColSlots + lane] : 0u;
  um.cnt = 8 * __popc(um.mask);
  int inc = um.cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
  um.pre = inc - um.cnt;
  um.total = __shfl(inc, 63);
  return um;
}
struct ColUnit {
  int s, c, oct;            // segment, the lane's first column (even), octet of the segment
  int tile;                 // row-major index of the posterior tile (a row of the classification's partials)
  size_t w0;                // index of the column word of c (c + 1: the next word)
  size_t g0;                // candidate (first row of the octet, column c)
};
__device__ __forceinline__ ColUnit col_unit(const ColGeom& gm, const ColUnitMap& um, long long u, int lane) {
  // tile row: the last one whose prefix does not exceed u (rows without units share their successor's prefix: skipped by cnt > 0)
  const unsigned long long at = __ballot(um.cnt > 0 && (long long)um.pre <= u);
  const int s = 63 - __clzll((long long)at);
  const unsigned int mask = (unsigned int)__shfl((int)um.mask, s);
  const int pc = __shfl(um.cnt, s) >> 3, local = (int)(u - __shfl(um.pre, s));
  const int oct = local / pc, t = local - oct * pc;
  // tile column: the t-th set bit of the row's mask
  const unsigned long long hit = __ballot(lane < 32 && ((mask >> lane) & 1u) && __popc(mask & ((1u << lane) - 1u)) == t);
  const int tc = (int)__ffsll((long long)hit) - 1;
  ColUnit x;
  x.oct = oct;
  x.s = s;
  x.tile = s * gm.gxt + tc;
  x.c = tc * 128 + 2 * lane;
  x.w0 = (size_t)x.s * gm.W + x.c;
  x.g0 = ((size_t)(64 * x.s + 8 * x.oct)) * gm.W + x.c;
  return x;
}
typedef double d2c_t __attribute__((ext_vector_type(2)));
typedef unsigned long long u2c_t __attribute__((ext_vector_type(2)));

// ---- minimiser: M = {g in S : lcb_0 <= u*} (models/SafeOpt.py:62), arg-max var_0 over M, |M|; and, from the finished G words,
// |G_1| and arg-max var_0 over G_1 (G_1 is a subset of S: its var_0 is already in registers) ------------------------------------------
// lcb_0 against u* is decided from the single-precision bounds where they settle it -- the exact bound (IEEE square root) only inside
// their margin.
struct ColMinJob {
  ColGeom gm;
  const unsigned long long* slots;
  unsigned long long* Lmax;
  ColScal2* sc2;
  const unsigned long long* olmin;        // per objective tile: key of the smallest lower bound of lcb_0 over its safe candidates
  const unsigned long long* Sw;
  const double* mean0;
  const double* var0;
  double b;
  unsigned long long* Mw;
  const unsigned long long* Gw;
  Best* partial;            // rows of the minimiser
  Best* gpartial;           // rows of the arg-max over G_1
  const GuardBand* gb;
};
__global__ __launch_bounds__(256) void k_col_min(const ColMinJob j) {
  unsigned long long ukey, vkey;
  col_merge2_body(j.slots, ukey, vkey);
  const ColUnitMap um = col_unit_map(j.slots, j.gm.NS);
  const long long nunits = um.total;
  double du0 = 0.0;
  if (j.gb) {
    const bool any = vkey != ~0ull;
    const double vm = any ? fmax(0.0, ord_val(vkey)) : 0.0;
    du0 = j.gb->dm[0] + (any ? j.b * gb_dsqrt(vm, j.gb->dv[0]) : 0.0);
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
    const unsigned long long l0 = col_slot_reduce(j.slots, kSlotL0);
    if (threadIdx.x == 0) { j.Lmax[0] = l0; j.sc2->ustar_key = ukey; j.sc2->vmin0_key = vkey; j.sc2->gb_du0 = du0; }
  }
  const ColGeom& gm = j.gm;
  const double ustar = ord_val(ukey), b = j.b;
  const bool anyS = ukey != ~0ull;
  const GuardBand* gb = j.gb;
  const bool e2on = gb != nullptr;
  double bvM = -kInfD, e2M = -kInfD, bvG = -kInfD, e2G = -kInfD;
  unsigned int bgM = 0xffffffffu, bgG = 0xffffffffu;
  long long cM = 0, cB = 0, cG = 0;
  const double dM = gb ? 2.0 * du0 * (1.0 + 0x1p-40) : -1.0, dv0 = gb ? gb->dv[0] : 0.0;
  const double mg = dM > 0.0 ? dM : 0.0;
  // thresholds of the pre-test, rounded away from u*
  const float bf = -(float)b, ulo_f = __double2float_rd(ustar - mg), uhi_f = __double2float_ru(ustar + mg);
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  uint8_t* Mb = reinterpret_cast<uint8_t*>(j.Mw);
  for (long long u = wave; u < nunits && anyS; u += nwaves) {
    const ColUnit x = col_unit(gm, um, u, lane);
    const u2c_t sw = *reinterpret_cast<const u2c_t*>(j.Sw + x.w0), gw = *reinterpret_cast<const u2c_t*>(j.Gw + x.w0);
    const unsigned int sb[2] = {(unsigned int)(sw[0] >> (8 * x.oct)) & 0xffu, (unsigned int)(sw[1] >> (8 * x.oct)) & 0xffu};
    const unsigned int gbt[2] = {(unsigned int)(gw[0] >> (8 * x.oct)) & 0xffu, (unsigned int)(gw[1] >> (8 * x.oct)) & 0xffu};
    unsigned int mb[2] = {0u, 0u}, und[2] = {0u, 0u};
    // The tile's range first: lcb_0 >= lmin on all of its safe candidates, so a tile with lmin > u* (+ the guard's margin) holds no
    // member of M and nothing near it -- M is a thin set (80 of the 515 tiles with a safe candidate on config H) --, and such a
    // tile only takes part through its members of G_1: var_0 of those rows alone.
    const unsigned long long lk = j.olmin[x.tile];
    const bool mposs = lk != ~0ull && !(ord_val(lk) > ustar + mg);
    const unsigned int any8 = mposs ? (sb[0] | sb[1]) : (gbt[0] | gbt[1]);
    if (__ballot(any8 != 0u) != 0ull) {
      d2c_t mu[8], va[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool on = (any8 >> k) & 1u;
        const size_t g = x.g0 + (size_t)k * gm.W;
        mu[k] = (on && mposs) ? *reinterpret_cast<const d2c_t*>(j.mean0 + g) : d2c_t{0.0, 0.0};
        va[k] = on ? *reinterpret_cast<const d2c_t*>(j.var0 + g) : d2c_t{0.0, 0.0};
      }
      const unsigned int g32 = (unsigned int)x.g0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const bool on = mposs && ((sb[e] >> k) & 1u);
          const double v = va[k][e];
          const F32Bound fb = f32_bound(mu[k][e], v, bf);                 // lcb_0 within fb.del of fb.u
          const bool sureM = fb.ok && fb.u + fb.del < ulo_f, sureN = fb.ok && fb.u - fb.del > uhi_f;
          const bool mm = on && sureM;
          und[e] |= (on && !sureM && !sureN) ? (1u << k) : 0u;
          mb[e] |= mm ? (1u << k) : 0u;
          const unsigned int g = g32 + (unsigned int)(k * gm.W + e);
          if (__ballot(mm) != 0ull) {                                    // (M is a thin set: most steps have no member in the wave)
            cM += mm;
            if (e2on) take_max<true>(mm, v, g, bvM, bgM, e2M);
            else take_max<false>(mm, v, g, bvM, bgM, e2M);
          }
          const bool gg = (gbt[e] >> k) & 1u;
          if (__ballot(gg) != 0ull) {
            cG += gg;
            if (e2on) take_max<true>(gg, v, g, bvG, bgG, e2G);
            else take_max<false>(gg, v, g, bvG, bgG, e2G);
          }
        }
      }
      // what the single-precision test left open (candidates within ~1e-6 of u*, or inside the guard band): the exact bound
      while (__ballot((und[0] | und[1]) != 0u) != 0ull) {
        const bool act = (und[0] | und[1]) != 0u;
        const int e = und[0] ? 0 : 1;
        const int k = act ? (int)(__ffs((int)und[e]) - 1) : 0;
        if (act) und[e] &= und[e] - 1u;
        const size_t g = x.g0 + (size_t)k * gm.W + e;
        const double m = act ? j.mean0[g] : 0.0, v = act ? j.var0[g] : 0.0;
        bool mm = false;
        if (act) {
          double lcb, ucb;
          lcb_ucb(m, v, b, lcb, ucb);
          mm = lcb <= ustar;                       // models/SafeOpt.py:62
          const double gap = lcb - ustar;
          cB += (gap < 0 ? -gap : gap) <= dM;
        }
        mb[e] |= mm ? (1u << k) : 0u;
        cM += mm;
        if (e2on) take_max<true>(mm, v, (unsigned int)g, bvM, bgM, e2M);
        else take_max<false>(mm, v, (unsigned int)g, bvM, bgM, e2M);
      }
    }
    Mb[8 * x.w0 + x.oct] = (uint8_t)mb[0];
    Mb[8 * (x.w0 + 1) + x.oct] = (uint8_t)mb[1];
  }
  Best best = best_from(bvM, bgM, e2M), bestG = best_from(bvG, bgG, e2G);
  best = block_best_uni<true>(best, dv0);
  cM = block_sum_ll(cM);
  cB = block_sum_ll(cB);
  if (threadIdx.x == 0) {
    j.partial[blockIdx.x] = best;
    ((long long*)(j.partial + gridDim.x))[blockIdx.x] = cM;
    ((long long*)(j.partial + gridDim.x))[gridDim.x + blockIdx.x] = cB;
  }
  __syncthreads();
  bestG = block_best_uni<true>(bestG, dv0);
  cG = block_sum_ll(cG);
  if (threadIdx.x == 0) {
    j.gpartial[blockIdx.x] = bestG;
    ((long long*)(j.gpartial + gridDim.x))[blockIdx.x] = cG;
    ((long long*)(j.gpartial + gridDim.x))[gridDim.x + blockIdx.x] = 0;
  }
}

// ---- exact nearest-U search along a row of a column image ---------------------------------------------------------------------
// Position i of row `rowi` (16-bit steps down the columns, 0xffff: no U point in that column) with the row's block minima `rowb`
// (32 positions per block): min over i' of (h1 t(i'))^2 + (h0 (i' - i))^2, EIGHT lanes per search (eight searches per wave, all
// lanes of the wave take part in the shuffles).  A lane bounds BPL blocks in SINGLE precision, rounded down (the bounds only
// prune); the block with the smallest bound is searched first -- 32 entries, four per lane, exact fp64 --, then every block whose
// bound still beats the running minimum (rounded up to float: "<" never skips a block the fp64 comparison would search).  `cap`:
// entries further along the row are not examined; `acc2`: a minimum at or below it ends the search (-1: none).  BPL = 16 covers the
// 128 blocks of a fine row of 4096 columns, BPL = 2 the 16 blocks of a coarse row (the same transform on the 8 x 8 cells).
template <int BPL>
__device__ __forceinline__ double col_row_search(const unsigned short* __restrict__ rowi, const unsigned short* __restrict__ rowb, int NB, int i,
                                                 double h0, double h1, double inv_h0, double cap, double acc2, bool live, int lane, int sub) {
  constexpr int GL = 8;
  const float h0f = (float)h0, h1f = (float)h1;
  unsigned int bm[BPL];
  {
    const int bb0 = BPL * lane;
    if (BPL == 16) {
      uint4 qa = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu), qb = qa;
      if (live && bb0 < NB) qa = *reinterpret_cast<const uint4*>(rowb + bb0);
      if (live && bb0 + 8 < NB) qb = *reinterpret_cast<const uint4*>(rowb + bb0 + 8);
      const unsigned int ws[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
#pragma unroll
      for (int e = 0; e < BPL / 2; ++e) { bm[2 * e] = ws[e] & 0xffffu; bm[2 * e + 1] = ws[e] >> 16; }
    } else {
      unsigned int w = 0xffffffffu;
      if (live && bb0 < NB) w = *reinterpret_cast<const unsigned int*>(rowb + bb0);
      bm[0] = w & 0xffffu;
      bm[BPL - 1] = w >> 16;
    }
  }
  double best_d = kInfD;
  if (live) {
    const unsigned int t = rowi[i];
    const double dt = h1 * (double)t;
    best_d = t == 0xffffu ? kInfD : dt * dt;
  }
  // steps along the row beyond which h0 gap > cap for sure (a block kept although the cap drops it holds no entry within the cap:
  // its search changes nothing)
  const double gq = cap * inv_h0 * (1.0 + 1e-12) + 1.0;
  const int gapmax = gq < 1.0e9 ? (int)gq : 1000000000;
  // bounds: (h1 bm)^2 + (h0 gap)^2 carries <= 6 roundings of 2^-24, the factor takes 1e-6 off (a block without a U point holds
  // 0xffff: its bound is beyond every distance of the grid)
  float lb[BPL];
  float lbl = 3.0e38f;
  int bl = -1;
#pragma unroll
  for (int e = 0; e < BPL; ++e) {
    const int bb = BPL * lane + e;
    const int ga = i - (bb * 32 + 31), gb_ = bb * 32 - i;
    int gap = ga > gb_ ? ga : gb_;
    gap = gap > 0 ? gap : 0;
    const float x = h1f * (float)bm[e], y = h0f * (float)gap;
    float v = (x * x + y * y) * (1.0f - 1.0e-6f);
    if (!live || gap > gapmax || bb >= NB) v = 3.0e38f;
    lb[e] = v;
    if (v < lbl) { lbl = v; bl = bb; }
  }
  int b_min = bl;
  {
    float m = lbl;
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) {
      const float wm = __shfl_xor(m, o);
      const int wb = __shfl_xor(b_min, o);
      if (wm < m || (wm == m && wb >= 0 && (b_min < 0 || wb < b_min))) { m = wm; b_min = wb; }
    }
    lbl = m;                                            // (the group's smallest bound and its block: the same in every lane)
  }
  auto scan_block = [&](int bb, bool act) {             // the group: the 32 entries of block bb, four per lane
    double cnd = kInfD;
    if (act) {
      const uint2 w2 = *reinterpret_cast<const uint2*>(rowi + bb * 32 + 4 * lane);
      const unsigned int ts[4] = {w2.x & 0xffffu, w2.x >> 16, w2.y & 0xffffu, w2.y >> 16};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ii = bb * 32 + 4 * lane + e;
        const double dt = h0 * (double)(ii > i ? ii - i : i - ii);
        if (dt <= cap && ts[e] != 0xffffu) {
          const double dv = h1 * (double)ts[e];
          const double v = dv * dv + dt * dt;
          cnd = v < cnd ? v : cnd;
        }
      }
    }
#pragma unroll
    for (int o = GL / 2; o > 0; o >>= 1) { const double w = __shfl_xor(cnd, o); cnd = w < cnd ? w : cnd; }
    best_d = cnd < best_d ? cnd : best_d;
  };
  float best_f = __double2float_ru(best_d);
  scan_block(b_min, live && b_min >= 0 && lbl < best_f);
  best_f = __double2float_ru(best_d);
  unsigned int todo = 0u;
#pragma unroll
  for (int e = 0; e < BPL; ++e) todo |= (lb[e] < best_f && BPL * lane + e != b_min) ? (1u << e) : 0u;
  if (best_d <= acc2) todo = 0u;
  while (__ballot(todo != 0u) != 0ull) {
    const unsigned int gm8 = (unsigned int)((__ballot(todo != 0u) >> (GL * sub)) & 0xffull);
    const bool act = gm8 != 0u;
    const int leader = act ? (int)(__ffs((int)gm8) - 1) : 0;
    const int mine = BPL * lane + (todo ? (int)(__ffs((int)todo) - 1) : 0);
    const int bb = __shfl(mine, leader + GL * sub);
    if (act && lane == leader) todo &= todo - 1u;
    scan_block(bb, act);
    if (act) {
      best_f = __double2float_ru(best_d);
      unsigned int keep = 0u;
#pragma unroll
      for (int e = 0; e < BPL; ++e) keep |= (lb[e] < best_f) ? (1u << e) : 0u;
      todo &= keep;
      if (best_d <= acc2) todo = 0u;
    }
  }
  return best_d;
}

// ---- verdicts: G = {g in S : exists h in U, ucb_1(g) - L ||x_g - x_h + 1e-8|| >= 0} (models/SafeOpt.py:85-88, 111) ------------------
// The coarse sandwich of k_edt_decide8 per candidate, first on the single-precision bounds of ucb (no IEEE square root for the
// nine candidates in ten it settles), then on the exact bound; what stays open goes to the list.  A workgroup takes one
// (segment, 64 columns, 32-row half), its four waves one coarse row (eight rows) each: one batch of loads per wave, the open
// candidates meet in a queue in LDS and leave with ONE atomic on the list's counter per workgroup (a wave that waited for its own
// atomic after every ~200 entries lived 60 us on config H).
struct ColVerdict {
  const double* mean_c;
  const double* var_c;
  const double* var0;
  double b;
  const unsigned long long* Lkeys;
  int lidx;
  double xscale;
  RcExp rx;
};
constexpr int kColQueue = 512;                  // open candidates a WAVE collects in LDS before it takes list slots with one atomic
__global__ __launch_bounds__(256) void k_col_decide(const ColGeom gm, const unsigned long long* __restrict__ Sw, const unsigned long long* __restrict__ slots,
                                                    const unsigned long long* __restrict__ tumin /* per constraint tile: key of the lower end
                                                    of ucb_1 over its safe candidates */, const unsigned long long* __restrict__ tumax /* of the
                                                    upper end (the tile's radius key) */, const unsigned short* __restrict__ cimg, const unsigned short* __restrict__ cbmin,
                                                    double cap_extra, unsigned long long* __restrict__ Usum,
                                                    const unsigned short* __restrict__ img /* the fine column image */, const ColVerdict cv,
                                                    SweepScalars* sc, unsigned long long* __restrict__ Gw, long long* __restrict__ scanlist) {
  __shared__ long long qg[4][kColQueue];
  __shared__ double qu[4][kColQueue];
  const double L = __longlong_as_double((long long)cv.Lkeys[cv.lidx]);
  const bool on_all = sc->count_U > 0 && sc->count_S > 0;
  const ColUnitMap um = col_unit_map(slots, gm.NS);
  const long long nunits = on_all ? um.total : 0;
  const RcBandK bk = rc_band(cv.rx, sc);
  const double eps_abs = 1.01e-8 * sqrt(2.0) + 1e-14 * cv.xscale + 1e-13;
  const double delta = gm.delta;
  const double b = cv.b;
  const float bfp = (float)b;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  uint8_t* Gb = reinterpret_cast<uint8_t*>(Gw);
  // (Usum cleared for the next sweep's posterior: its readers ran in the launch before)
  for (int i = (int)blockIdx.x * 256 + threadIdx.x; i < gm.W; i += (int)gridDim.x * 256) Usum[i] = 0ull;
  // the coarse transform's distances: searched per cell, capped beyond every radius that can matter (as the byte-mask path's coarse
  // scan: a cell beyond the cap is "beyond every radius" whether its value is the true minimum or a larger one)
  const double rmax = sc->rmax_key[1] ? ord_val(sc->rmax_key[1]) : 0.0;
  const double capC = L > 0 ? rmax / L * 1.000001 + 1e-6 + cap_extra : kInfD;
  const double h0c = gm.h0 * kCoarse, h1c = gm.h1 * kCoarse, inv_h0c = 1.0 / h0c;
  int qcnt = 0;                                   // (uniform over the wave)
  auto flush = [&]() {
    if (qcnt == 0) return;
    __builtin_amdgcn_wave_barrier();
    long long base = 0;
    if (lane == 0) base = (long long)atomicAdd((unsigned long long*)&sc->n_scan, (unsigned long long)qcnt);
    base = __shfl(base, 0);
    for (int k = lane; k < qcnt; k += 64) {
      scanlist[2 * (base + k)] = qg[wv][k];
      reinterpret_cast<double*>(scanlist)[2 * (base + k) + 1] = qu[wv][k];
    }
    __builtin_amdgcn_wave_barrier();
    qcnt = 0;
  };
  for (long long u = wave; u < nunits; u += nwaves) {
    const ColUnit x = col_unit(gm, um, u, lane);
    const u2c_t sw = *reinterpret_cast<const u2c_t*>(Sw + x.w0);
    const unsigned int sb[2] = {(unsigned int)(sw[0] >> (8 * x.oct)) & 0xffu, (unsigned int)(sw[1] >> (8 * x.oct)) & 0xffu};
    unsigned int gbits[2] = {0u, 0u};
    const unsigned int any8 = sb[0] | sb[1];
    if (__ballot(any8 != 0u) != 0ull) {
      const int row0 = 64 * x.s + 8 * x.oct;
      // The coarse distances of the unit's sixteen cells (the octet is one coarse row, the lane's two columns lie in cell lane >> 2):
      // eight lanes search a cell, two rounds; a lane then picks up its own cell's value.
      double dc2;
      {
        const int cj = row0 >> 3, ci0 = (x.c - 2 * lane) >> 3, l8 = lane & 7, sub = lane >> 3;
        const unsigned short* rowi = cimg + (size_t)cj * gm.CWp;
        const unsigned short* rowb = cbmin + (size_t)cj * gm.CNBp;
        const double d0 = col_row_search<2>(rowi, rowb, gm.CNB, ci0 + sub, h0c, h1c, inv_h0c, capC, -1.0, true, l8, sub);
        const double d1 = col_row_search<2>(rowi, rowb, gm.CNB, ci0 + 8 + sub, h0c, h1c, inv_h0c, capC, -1.0, true, l8, sub);
        const int m = lane >> 2;
        const double v0 = __shfl(d0, 8 * (m & 7)), v1 = __shfl(d1, 8 * (m & 7));
        dc2 = (m >> 3) ? v1 : v0;
      }
      // The whole unit at once where that settles it: ucb_1 of every safe candidate of the tile lies in [ulo, uhi] (the posterior's
      // epilogue kept both ends), the coarse distances of the unit's sixteen cells in [dmin, dmax] -- if the candidate with the
      // SMALLEST bound at the LARGEST distance is within its radius for sure, all are (G = S on the unit, no candidate is read), and
      // likewise none is when the largest bound at the smallest distance is beyond it.  Only the units the boundary of G_1 crosses
      // read mean / var (a third of the 515 tiles with a safe candidate on config H).
      if (L > 0) {
        double dmx = dc2, dmn = dc2;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const double a = __shfl_xor(dmx, o), c2 = __shfl_xor(dmn, o);
          dmx = a > dmx ? a : dmx;
          dmn = c2 < dmn ? c2 : dmn;
        }
        const unsigned long long klo = tumin[x.tile], khi = tumax[x.tile];
        if (klo != ~0ull && khi != 0ull) {
          const double ulo = ord_val(klo), uhi = ord_val(khi);
          const double amax = fmax(fabs(ulo), fabs(uhi));
          const double du = cv.rx.gb_c > 0 ? bk.du + bk.rl * amax : 0.0;
          const double dCx = sqrt(dmx), dCn = sqrt(dmn);
          const double dhi = dCx * (1.0 + 1e-9) + delta, dlo = fmax(0.0, dCn * (1.0 - 1e-9) - delta);
          const double tolu = 1e-12 * (amax + du + L * dhi);
          if (ulo - du - L * (dhi + eps_abs + 1e-11 * dhi) > tolu) {             // every safe candidate of the unit is in G_1
            Gb[8 * x.w0 + x.oct] = (uint8_t)sb[0];
            Gb[8 * (x.w0 + 1) + x.oct] = (uint8_t)sb[1];
            continue;
          }
          if (uhi + du - L * (dlo - eps_abs - 1e-11 * dlo) < -tolu) {           // none is
            Gb[8 * x.w0 + x.oct] = 0;
            Gb[8 * (x.w0 + 1) + x.oct] = 0;
            continue;
          }
        }
      }
      d2c_t mu[8], va[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool on = (any8 >> k) & 1u;
        const size_t g = x.g0 + (size_t)k * gm.W;
        mu[k] = on ? *reinterpret_cast<const d2c_t*>(cv.mean_c + g) : d2c_t{0.0, 0.0};
        va[k] = on ? *reinterpret_cast<const d2c_t*>(cv.var_c + g) : d2c_t{0.0, 0.0};
      }
      const double dC = sqrt(dc2);
      const double dhi = dC * (1.0 + 1e-9) + delta, dlo = fmax(0.0, dC * (1.0 - 1e-9) - delta);
      const double Lhi = L * (dhi + eps_abs + 1e-11 * dhi), Llo = L * (dlo - eps_abs - 1e-11 * dlo), Ldhi = L * dhi;
      // Pre-test in single precision (f32_bound): ucb within fb.del of fb.u, the thresholds rounded away from the candidate's side,
      // the band of the guard (du + rl |ucb|) taken at the largest |ucb| the interval allows, and a tolerance of 1e-6 of the terms'
      // sizes -- six orders above the fp64 test's 1e-12 and above the float roundings of the difference itself.  What it does not
      // settle (`und`) takes the fp64 path below: the coarse sandwich on the exact bound, as k_edt_decide8.
      const float Lhi_f = __double2float_ru(Lhi), Llo_f = __double2float_rd(Llo);
      const float du_f = cv.rx.gb_c > 0 ? __double2float_ru(bk.du) : 0.0f, rl_f = cv.rx.gb_c > 0 ? __double2float_ru(bk.rl) : 0.0f;
      unsigned int und[2] = {0u, 0u};
      if (L > 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const bool on = (sb[e] >> k) & 1u;
            const F32Bound fb = f32_bound(mu[k][e], va[k][e], bfp);
            const float dumax = fmaf(rl_f, fb.amax, du_f) * 1.000001f;
            const float tol = 1.0e-6f * (fb.amax + dumax + Lhi_f);
            const bool in = fb.ok && (fb.u - fb.del - dumax - Lhi_f) > tol, out = fb.ok && (fb.u + fb.del + dumax - Llo_f) < -tol;
            gbits[e] |= (on && in) ? (1u << k) : 0u;
            und[e] |= (on && !in && !out) ? (1u << k) : 0u;
          }
        }
      } else {
        und[0] = sb[0];
        und[1] = sb[1];
      }
      while (__ballot((und[0] | und[1]) != 0u) != 0ull) {
        const bool act = (und[0] | und[1]) != 0u;
        const int e = und[0] ? 0 : 1;
        const int k = act ? (int)(__ffs((int)und[e]) - 1) : 0;
        if (act) und[e] &= und[e] - 1u;
        const long long g = (long long)(x.g0 + (size_t)k * gm.W + e);
        const double m = act ? cv.mean_c[g] : 0.0, v = act ? cv.var_c[g] : 0.0;     // (second read of a value this wave has just loaded)
        bool open = false, in = false;
        double ucb = 0.0;
        if (act) {
          double lcb;
          lcb_ucb(m, v, b, lcb, ucb);
          const double du = rc_du(cv.rx, bk, g, v, b, ucb);
          if (!(L > 0)) {
            in = ucb >= 0.0;                                  // radius unbounded: any U point is a witness
            if (du > 0.0 && fabs(ucb) <= du) rc_defer(cv.rx, sc, g);
          } else {
            const double tolc = 1e-12 * (fabs(ucb) + du + Ldhi);
            if (ucb - du - Lhi > tolc) in = true;                      // within the radius for sure
            else if (ucb + du - Llo < -tolc) in = false;               // beyond it for sure
            else {
              // the candidate's own column: the U point straight above or below it bounds the distance from above -- within the
              // radius for sure when that already is (the list search would find a minimum no larger and say the same)
              const unsigned int t = img[(size_t)g];
              const double dup = gm.h1 * (double)t;
              if (t != 0xffffu && ucb - du - L * (dup + eps_abs + 1e-11 * dup) > 1e-12 * (fabs(ucb) + du + L * dup)) in = true;
              else open = true;
            }
          }
        }
        gbits[e] |= in ? (1u << k) : 0u;
        const unsigned long long om = __ballot(open);
        if (om) {
          if (qcnt > kColQueue - 64) flush();
          if (open) {
            const int slot = qcnt + __popcll(om & ((1ull << lane) - 1ull));
            qg[wv][slot] = ((long long)(row0 + k) << 32) | (long long)(x.c + e);          // (row, column): the list kernel needs no division
            qu[wv][slot] = ucb;
          }
          qcnt += __popcll(om);
        }
      }
    }
    Gb[8 * x.w0 + x.oct] = (uint8_t)gbits[0];
    Gb[8 * (x.w0 + 1) + x.oct] = (uint8_t)gbits[1];
  }
  flush();
}

// Search + verdict for the listed candidates, EIGHT lanes each (eight candidates per wave).  Candidate (i, j): the nearest U point
// minimises (h1 t(i', j))^2 + (h0 (i' - i))^2 over the columns i' of ROW j -- contiguous 16-bit entries.  A lane takes the block
// minima of sixteen 32-column blocks (two 16-byte loads; a row of 4096 columns is 128 blocks) and bounds them in SINGLE precision,
// rounded down (they only prune); the block with the smallest bound is searched first -- 32 entries, four per lane, exact fp64 --
// then every block whose bound still beats the running minimum.  Same candidates, exits and verdict arithmetic as k_edt_scan_list
// (the two squares of a witness are the same doubles in either orientation): 1/6 of the instructions of the 16-lane fp64 form.
__global__ __launch_bounds__(256) void k_col_scan(const ColGeom gm, const unsigned short* __restrict__ img, const unsigned short* __restrict__ bmin,
                                                  const ColVerdict cv, SweepScalars* sc, unsigned long long* __restrict__ Gw,
                                                  long long* __restrict__ amb, const long long* __restrict__ scanlist) {
  constexpr int GL = 8;
  const double L = __longlong_as_double((long long)cv.Lkeys[cv.lidx]);
  const long long nscan = sc->n_scan;
  const RcBandK bk = rc_band(cv.rx, sc);
  const double eps_abs = 1.01e-8 * sqrt(2.0) + 1e-14 * cv.xscale + 1e-13;
  const int lane = threadIdx.x & (GL - 1);
  const int sub = (threadIdx.x & 63) / GL;
  const long long ngroups = (long long)gridDim.x * (blockDim.x / GL);
  const double h0 = gm.h0, h1 = gm.h1;
  const double invL = 1.0 / L, inv_h0 = 1.0 / h0;
  unsigned int* Gh = reinterpret_cast<unsigned int*>(Gw);
  const long long nrounds = (nscan + ngroups - 1) / ngroups;
  const long long grp = (long long)blockIdx.x * (blockDim.x / GL) + threadIdx.x / GL;
  for (long long rd = 0; rd < nrounds; ++rd) {
    const long long qi = grp + rd * ngroups;
    const bool live = qi < nscan;                         // (dead groups keep the wave's shuffles company)
    const long long rc = live ? scanlist[2 * qi] : 0;
    const int j = (int)(rc >> 32), i = (int)(rc & 0xffffffffll);
    const long long g = (long long)j * gm.W + i;
    const double ucb = live ? reinterpret_cast<const double*>(scanlist)[2 * qi + 1] : 0.0;
    const double du = cv.rx.gb_c > 0 ? rc_du(cv.rx, bk, g, 0.0, cv.b, ucb) : 0.0;
    // (radius by the reciprocal of L: the cap only has to stay above the radius -- it does by 4e-8 + 1e-9 r --, and the early exit
    // below it -- by 2e-8 + 1e-10 r; a quotient rounded differently from the byte-mask path's moves neither verdict)
    const double rhi = (ucb + du) * invL, rlo = (ucb - du) * invL;
    const double cap = rhi * (1.0 + 1e-15) + 4.0 * eps_abs + 1e-9 * fabs(rhi);
    const double thr = rlo * (1.0 - 1e-10) - 2.0 * eps_abs - 1e-12;
    const double acc2 = thr > 0 ? thr * thr : -1.0;
    const double best_d = col_row_search<16>(img + (size_t)j * gm.W, bmin + (size_t)j * gm.NBp, gm.NB, i, h0, h1, inv_h0, cap, acc2, live, lane, sub);
    if (live && lane == 0) {
      bool out = false;
      if (best_d < 0.5 * kInfD) {
        const double dm = sqrt(best_d);
        const double eps = eps_abs + 1e-11 * dm;
        const double tol = 1e-12 * (fabs(ucb) + du + L * dm);
        const double lo = ucb - du - L * (dm + eps), hi = ucb + du - L * (dm - eps);
        if (lo > tol) out = true;
        else if (hi >= -tol) amb[atomicAdd((unsigned long long*)&sc->n_amb, 1ull)] = g;     // (judged by k_expander_exact, guard band included)
      }
      if (out) atomicOr(&Gh[2 * ((size_t)(j >> 6) * gm.W + i) + ((j >> 5) & 1)], 1u << (j & 31));
    }
  }
}

// The final reductions, two levels in one launch: workgroup (slot, p) merges its share of the slot's region -- slot 0 the minimiser's
// rows -> |M| and arg-max var_0 over M, slot 1 the verdict kernels' rows -> |G_1| and arg-max var_0 over G_1 --, and the slot's last
// workgroup to finish merges the kColFinParts intermediate results and writes them into the host's pinned block (as k_sweep_finals<true>
// with a mirror; one workgroup took 23 us for the 6144 rows of config H).  `fin`: [2][kColFinParts] intermediate rows + 2 tickets.
constexpr int kColFinParts = 8;
struct ColFinRow { Best b; long long cnt, nb; };
__global__ __launch_bounds__(256) void k_col_finals(const Best* __restrict__ reg0, int n0, const Best* __restrict__ reg1, int n1, SweepScalars* sc,
                                                    const ColScal2* sc2, ColFinRow* fin, unsigned long long* tickets, unsigned char* mirror,
                                                    const unsigned long long* Lkeys, int gb_on, unsigned long long* slots) {
  const int slot = blockIdx.y, p = blockIdx.x;
  const Best* reg = slot == 0 ? reg0 : reg1;
  const int n = slot == 0 ? n0 : n1;
  const int i0 = (int)((long long)n * p / (int)gridDim.x), i1 = (int)((long long)n * (p + 1) / (int)gridDim.x);
  Best best = best_none<true>();
  long long cnt = 0, nb = 0;
  {
    const long long* pc = (const long long*)(reg + n);
#pragma unroll 4
    for (int i = i0 + (int)threadIdx.x; i < i1; i += blockDim.x) {
      best = best_merge<true>(best, reg[i]);
      cnt += pc[i];
      nb += pc[n + i];
    }
    best = block_best<true>(best);
    cnt = block_sum_ll(cnt);
    nb = block_sum_ll(nb);
  }
  // (one workgroup per slot -- gridDim.x == 1, the launch of today's 2 x 1024 rows -- needs no second level: no ticket, no fences)
  const int parts = (int)gridDim.x;
  __shared__ int last;
  if (parts > 1) {
    if (threadIdx.x == 0) {
      fin[slot * kColFinParts + p] = ColFinRow{best, cnt, nb};
      __threadfence();
      const unsigned long long t = atomicAdd(&tickets[slot], 1ull);
      last = t == (unsigned long long)(parts - 1);
      if (last) tickets[slot] = 0ull;                     // (for the next sweep)
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
  }
  // (the slot block back to its neutral elements: every reader of this sweep ran in an earlier launch)
  if (slot == 1)
    for (int i = threadIdx.x; i < kColSlotFields * kColSlots; i += blockDim.x) slots[i] = col_slot_is_min(i / kColSlots) ? ~0ull : 0ull;
  if (threadIdx.x == 0) {
    if (parts > 1) {
    best = best_none<true>();
    cnt = nb = 0;
    for (int k = 0; k < parts; ++k) {
      const ColFinRow r{Best{__hip_atomic_load(&fin[slot * kColFinParts + k].b.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                             __hip_atomic_load(&fin[slot * kColFinParts + k].b.i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                             __hip_atomic_load(&fin[slot * kColFinParts + k].b.d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                             __hip_atomic_load(&fin[slot * kColFinParts + k].b.e1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                             __hip_atomic_load(&fin[slot * kColFinParts + k].b.e2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                             __hip_atomic_load(&fin[slot * kColFinParts + k].b.ei, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)},
                        __hip_atomic_load(&fin[slot * kColFinParts + k].cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                        __hip_atomic_load(&fin[slot * kColFinParts + k].nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
      best = best_merge<true>(best, r.b);
      cnt += r.cnt;
      nb += r.nb;
    }
    }
    store_slot<true>(sc, slot, best, nb, gb_on != 0);
    if (slot == 0) sc->count_M += cnt;
    else sc->count_set[slot - 1] += cnt;
    SweepScalars* hm = reinterpret_cast<SweepScalars*>(mirror);
    hm->arg_val[slot] = best.v;
    hm->arg_idx[slot] = best.i;
    hm->arg_d[slot] = best.d;
    hm->guard_slot[slot] = sc->guard_slot[slot];
    if (slot == 0) {
      hm->count_M = sc->count_M;
      sc->ustar_key = sc2->ustar_key;
      hm->ustar_key = sc2->ustar_key;
      for (int t = 2; t < kArgSlots; ++t) hm->guard_slot[t] = 0;
    } else {
      // (the expander chain has finished before this launch: its counters are final)
      hm->count_set[slot - 1] = sc->count_set[slot - 1];
      hm->count_S = sc->count_S;
      hm->count_U = sc->count_U;
      hm->n_amb = sc->n_amb;
      hm->n_amb_total = sc->n_amb_total;
      hm->n_scan = sc->n_scan;
      hm->n_guard = sc->n_guard;
      for (int t = 0; t < kMaxQ; ++t) {
        hm->rmax_key[t] = sc->rmax_key[t];
        reinterpret_cast<unsigned long long*>(mirror + 3072)[t] = Lkeys[t];
      }
    }
  }
}

// column words -> the byte mask the C ABI hands out (sbo_masks_get) and the exhaustive recheck reads
__global__ __launch_bounds__(256) void k_col_expand(const unsigned long long* __restrict__ Wd, int W, long long n, uint8_t* __restrict__ out) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    const long long j = g / W;
    const int i = (int)(g - j * W);
    out[g] = (uint8_t)((Wd[(size_t)(j >> 6) * W + i] >> (j & 63)) & 1ull);
  }
}
static void col_expand(sbo_ctx* c, const DevBuf& words, uint8_t* out) {
  const long long n = c->cs.n_local;
  hipLaunchKernelGGL(k_col_expand, dim3((unsigned)std::min<long long>((n + 255) / 256, 1 << 16)), dim3(256), 0, c->stream,
                     (const unsigned long long*)words.p, (int)c->cs.count[0], n, out);
}

// ---- host: the set phase behind a posterior launch that delivered column words (sbo_ctx::col_active) -------------------------------
// Two chains (option "col_overlap", default on).  The EXPANDER chain needs the constraint's posterior only -- S / U words, the radius
// and Lipschitz keys of the constraint --, so it runs on the high-priority stream3 behind the constraint's k_bpost launch (fork event
// carried by that launch) WHILE the objective's k_bpost launch runs on the main stream: k_col_a, k_col_cs, k_col_decide, k_col_scan are
// latency-bound kernels whose waiting the matrix kernel fills.  The OBJECTIVE chain follows its own launch on the main stream: u*
// merged by every workgroup of k_col_min, M and its arg-max; the main stream then waits for the chain's join event and k_col_finals
// merges both.  What the set phase adds to K1 is the minimiser, the join and the finals.
static int col_set_phase(sbo_ctx* c, const sbo_sweep_opts* o, SweepScalars& h, unsigned long long* Lk) {
  const long long n = c->cs.n_local;
  const int q = c->mc.q;                 // == 2
  int rc;
  ColGeom gm;
  gm.W = (int)c->cs.count[0];
  gm.H = (int)(n / gm.W);
  gm.NS = gm.H / 64;
  gm.NB = gm.W / 32;
  gm.NBp = (gm.NB + 7) & ~7;
  gm.CW = gm.W / kCoarse;
  gm.CH = gm.H / kCoarse;
  gm.gxt = gm.W / 128;
  gm.CNB = (gm.CW + 31) / 32;
  gm.CNBp = 16;
  gm.CWp = gm.CNB * 32;
  gm.h0 = c->cs.step[0];
  gm.h1 = c->cs.step[1];
  double h2 = 0.0, hmax = 0.0;
  for (int a = 0; a < 2; ++a) { h2 += c->cs.step[a] * c->cs.step[a]; hmax = std::max(hmax, c->cs.step[a]); }
  gm.delta = (kCoarse - 1) * std::sqrt(h2) * (1.0 + 1e-9);
  const int lidx = o->reference_quirk_L_index ? q - 1 : 1;     // models/SafeOpt.py:110 (loop-leaked i)
  if ((rc = ensure(c->scal, sizeof(SweepScalars)))) return rc;
  SweepScalars* sc = (SweepScalars*)c->scal.p;
  // (their own small block: the objective's scalars, the finals' tickets and intermediate rows)
  static_assert(128 + 2 * kColFinParts * sizeof(ColFinRow) <= 4096, "column path scalar block");
  const bool fresh_fin = c->col_fin.bytes < 4096;
  if ((rc = ensure(c->col_fin, 4096))) return rc;
  ColScal2* sc2 = (ColScal2*)c->col_fin.p;
  unsigned long long* tickets = (unsigned long long*)((char*)c->col_fin.p + 64);
  ColFinRow* fin = (ColFinRow*)((char*)c->col_fin.p + 128);
  const bool overlap = c->col_overlap && c->stream3 && c->col_forked;
  hipStream_t xs = c->stream, es = overlap ? c->stream3 : c->stream;      // objective chain / expander chain
  if (fresh_fin) SBO_HIP(hipMemsetAsync(c->col_fin.p, 0, 4096, es));       // (tickets: the last workgroup of a slot resets its own)
  const int nb = reduce_blocks(c);
  // partial regions: slot 0 (the minimiser, nb rows) in the layout of the byte-mask path -- its late exhaustive recheck merges
  // slots 0 / 1 from there --, the rows of the arg-max over G_1 (nb) behind both
  const size_t pstride = partial_stride(nb);
  // (the list kernel: eight lanes per candidate, 32 candidates per workgroup and round)
  const int nsc = (int)std::min<long long>(2048, std::max<long long>(256, n / 4096));
  if ((rc = ensure(c->partial, pstride * 3))) return rc;
  unsigned char* pbase = (unsigned char*)c->partial.p;
  Best* reg0 = (Best*)pbase;
  Best* reg1 = (Best*)(pbase + 2 * pstride);
  if ((rc = ensure(c->amb, sizeof(long long) * (size_t)n))) return rc;
  if ((rc = ensure(c->scanlist, 2 * sizeof(long long) * (size_t)n))) return rc;
  if ((rc = ensure(c->col_img, sizeof(unsigned short) * (size_t)n + 64))) return rc;
  if ((rc = ensure(c->col_bmin, sizeof(unsigned short) * (size_t)gm.H * gm.NBp + 64))) return rc;
  // coarse column image / block minima (rows padded to whole 64-row coarse segments; block-minimum entries beyond the row stay 0xffff)
  const int crows = ((gm.CH + 63) / 64) * 64;
  const size_t cimg_bytes = sizeof(unsigned short) * (size_t)crows * gm.CWp + 64, cbmin_bytes = sizeof(unsigned short) * (size_t)crows * gm.CNBp + 64;
  const bool fresh_c = c->col_cbmin.bytes < cbmin_bytes || c->col_ckey != ((long long)gm.W << 32 | gm.H);
  if ((rc = ensure(c->col_cimg, cimg_bytes)) || (rc = ensure(c->col_cbmin, cbmin_bytes))) return rc;
  if (fresh_c) {
    SBO_HIP(hipMemsetAsync(c->col_cbmin.p, 0xff, c->col_cbmin.bytes, es));
    c->col_ckey = (long long)gm.W << 32 | gm.H;
  }
  ColBits cb{(unsigned long long*)c->cbS.p, (unsigned long long*)c->cbU.p, (unsigned long long*)c->cbUsum.p, (unsigned long long*)c->col_slots.p};
  c->lmax_pending = false;        // (the Lipschitz keys come out of the slot block)
  c->col_forked = false;

  // ---- expander chain
  if (overlap) SBO_HIP(hipStreamWaitEvent(es, c->ev_col[0], 0));
  // (K1i's deferred gradient launch merges the Lipschitz keys into the slot block on stream3: in stream order ahead of this chain when
  // the chain runs there, an event otherwise)
  if (c->grad_pending && es != c->stream3) SBO_HIP(hipStreamWaitEvent(es, c->ev_grad[2], 0));
  c->grad_pending = false;
  ColMergeJob mg;
  mg.slots = cb.slots;
  mg.sc = sc;
  mg.Lmax = (unsigned long long*)c->Lmax.p;
  mg.gb = gb_of(c);
  mg.b = o->b;
  const int ncoarse = gm.CNB;
  const int nfine = (int)std::max<long long>(1, std::min<long long>(((long long)gm.NS * (gm.W / 64) + 3) / 4, (long long)c->n_cu * 5));
  hipLaunchKernelGGL(k_col_a, dim3((unsigned)(1 + ncoarse + nfine)), dim3(256), 0, es, gm, cb, mg, ncoarse, (unsigned short*)c->col_cimg.p,
                     (unsigned short*)c->col_cbmin.p, (unsigned short*)c->col_img.p, (unsigned short*)c->col_bmin.p, (unsigned long long*)c->cbM.p, (unsigned long long*)c->cbG.p);
  c->usum_dirty = false;
  ColVerdict cv;
  memset(&cv, 0, sizeof(cv));
  cv.mean_c = (const double*)c->mean.p + (size_t)n;
  cv.var_c = (const double*)c->var.p + (size_t)n;
  cv.var0 = (const double*)c->var.p;
  cv.b = o->b;
  cv.Lkeys = (const unsigned long long*)c->Lmax.p;
  cv.lidx = lidx;
  double xscale = 0.0;
  for (int a = 0; a < 2; ++a) xscale = std::max(xscale, std::max(std::fabs(c->cs.lo[a]), std::fabs(c->cs.hi[a])));
  cv.xscale = xscale;
  if (gb_of(c)) {
    cv.rx.gb_c = 1;
    cv.rx.gb_l = c->gb_slow ? -1 : lidx;
  }
  // (the verdict kernels do not read var_0 -- overlapped, the objective's launch may still be writing it: the arg-max over G_1 is the
  // second job of k_col_min)
  // (grids sized to be resident at once: the waves loop over the units of the tiles with a safe candidate)
  const int ndw = std::max(1, c->n_cu * 4);
  const unsigned long long* rows = (const unsigned long long*)c->cpart.p;        // the posterior's partial rows: constraint tiles, then objective tiles
  const int ntiles = c->fuse_rows / 2;
  hipLaunchKernelGGL(k_col_decide, dim3((unsigned)ndw), dim3(256), 0, es, gm, (const unsigned long long*)cb.Sw, (const unsigned long long*)cb.slots,
                     rows, rows + (size_t)(kRowRmax + 1) * c->cpart_cap, (const unsigned short*)c->col_cimg.p, (const unsigned short*)c->col_cbmin.p,
                     2.0 * gm.delta + 2.0 * kCoarse * hmax, cb.Usum, (const unsigned short*)c->col_img.p, cv, sc, (unsigned long long*)c->cbG.p, (long long*)c->scanlist.p);
  hipLaunchKernelGGL(k_col_scan, dim3((unsigned)nsc), dim3(256), 0, es, gm, (const unsigned short*)c->col_img.p, (const unsigned short*)c->col_bmin.p,
                     cv, sc, (unsigned long long*)c->cbG.p, (long long*)c->amb.p, (const long long*)c->scanlist.p);
  c->amb_clean = false;
  if (overlap) {
    SBO_HIP(hipEventRecord(c->ev_col[1], es));
    SBO_HIP(hipStreamWaitEvent(xs, c->ev_col[1], 0));
  }

  // ---- objective chain (overlapped: its kernel takes the arg-max over G_1 along, so it follows the join)
  ColMinJob j;
  memset(&j, 0, sizeof(j));
  j.gm = gm;
  j.slots = cb.slots;
  j.Lmax = (unsigned long long*)c->Lmax.p;
  j.sc2 = sc2;
  j.olmin = rows + (size_t)1 * c->cpart_cap + ntiles;
  j.Sw = cb.Sw;
  j.mean0 = (const double*)c->mean.p;
  j.var0 = (const double*)c->var.p;
  j.b = o->b;
  j.Mw = (unsigned long long*)c->cbM.p;
  j.Gw = (const unsigned long long*)c->cbG.p;
  j.partial = reg0;
  j.gpartial = reg1;
  j.gb = gb_of(c);
  const int nbg = nb;
  hipLaunchKernelGGL(k_col_min, dim3((unsigned)nb), dim3(256), 0, xs, j);
  hipExtLaunchKernelGGL(k_col_finals, dim3(nb > 2048 ? kColFinParts : 1, 2), dim3(256), 0, xs, nullptr, c->ev[4], 0, (const Best*)reg0, nb, (const Best*)reg1,
                        nbg, sc, (const ColScal2*)sc2, fin, tickets, c->h_back, (const unsigned long long*)c->Lmax.p, gb_of(c) ? 1 : 0, cb.slots);
  c->slots_clean = true;
  SBO_HIP(hipGetLastError());
  bool is_max[kArgSlots];
  for (int t = 0; t < kArgSlots; ++t) is_max[t] = true;
  if ((rc = sweep_exchange_back(c, h, is_max, Lk, c->ev[4], true))) return rc;
  c->masks_bits = true;
  c->col_G_bytes = false;
  if (h.n_amb > 0 || c->exact_lazy == 2) {       // (2: always, the test of this path)
    // verdicts inside the reference's "+1e-8" band after all: the exhaustive recheck of the byte-mask path on the expanded U / G
    // masks, then the expanders' arg-max and the finals once more (slot 0 is where that merge expects it)
    col_expand(c, c->cbU, (uint8_t*)c->maskU.p);
    col_expand(c, c->cbG, (uint8_t*)c->maskG.p);
    if ((rc = launch_exact_d<double>(c, o, 1, lidx, (uint8_t*)c->maskG.p))) return rc;
    hipLaunchKernelGGL((k_arg_masked_multi<double, true, ValArray<double>>), dim3((unsigned)nb, 1u), dim3(256), 0, c->stream,
                       ValArray<double>{(const double*)c->var.p, 0.0}, (const uint8_t*)nullptr, (const uint8_t*)c->maskG.p, n, (long long)c->cs.first, pbase,
                       pstride, 1, gb_of(c));
    hipLaunchKernelGGL(k_sweep_clear_slot, dim3(1), dim3(1), 0, c->stream, sc, 1);
    hipExtLaunchKernelGGL(k_sweep_finals<true>, dim3(2), dim3(256), 0, c->stream, nullptr, c->ev[4], 0, (const unsigned char*)pbase, pstride, nb, sc,
                          (const SweepScalars*)nullptr, c->h_back, (const unsigned long long*)c->Lmax.p, gb_of(c) ? 1 : 0);
    SBO_HIP(hipGetLastError());
    if ((rc = sweep_exchange_back(c, h, is_max, Lk, c->ev[4], true))) return rc;
    c->col_G_bytes = true;
  }
  return SBO_OK;
}
