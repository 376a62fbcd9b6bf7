// sets_expander.inc.hpp: K4 kernels, expander sets by exact distance transform + band recheck -- part of the sets.hip translation unit (included inside namespace sbo; not a standalone header).
#pragma once

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

// The image of the fine axis-0 pass: squared distances as doubles, or -- on the shared-launch path of 2-D grids -- the step
// counts themselves as 16-bit integers (0xffff: no U point on the line), a quarter of the bytes written by the pass and read
// by the block minima and the list scan; the readers rebuild (h0 t)^2 exactly as the pass computes it.
struct DistF64 {
  const double* p;
  double h0;
  __device__ __forceinline__ double operator()(long long i) const { return p[i]; }
};
struct DistU16 {
  const unsigned short* p;
  double h0;
  __device__ __forceinline__ double operator()(long long i) const {
    const unsigned t = p[i];
    const double dt = h0 * (double)t;
    return t == 0xffffu ? kInfD : dt * dt;
  }
};
constexpr int kCoarse = 8;
constexpr int kDecideLines = 4;   // lines per workgroup of k_edt_decide / k_pdt_decide (2: 21.8 us, 4: 19.5, 8: 20.8, 16: 27.1 on config B)
struct CoarseGrid {
  int enabled;
  int d;
  long long count[kMaxD];    // fine counts
  long long ccount[kMaxD];   // coarse counts
  double delta;
  const double* Dc;          // squared coarse distances [prod ccount]
};

// OR of the U bytes of the kCoarse^d candidates of one coarse cell (lines of <= kCoarse bytes along axis 0)
__device__ __forceinline__ bool coarse_cell_any(const uint8_t* __restrict__ U, const CoarseGrid& cg, long long cell) {
  long long f = cell, i0[kMaxD], len[kMaxD], fstride[kMaxD], nsub = 1, fs = 1;
  for (int a = 0; a < cg.d; ++a) {
    const long long ca = f % cg.ccount[a];
    f /= cg.ccount[a];
    i0[a] = ca * kCoarse;
    len[a] = cg.count[a] - i0[a] < kCoarse ? cg.count[a] - i0[a] : kCoarse;
    fstride[a] = fs;
    fs *= cg.count[a];
    if (a > 0) nsub *= len[a];
  }
  bool any = false;
  if (cg.d == 2 && len[0] == 8 && len[1] == 8 && ((((uintptr_t)U) + i0[0] + i0[1] * fstride[1]) & 7) == 0 && (fstride[1] & 7) == 0) {
    // a whole 8 x 8 cell of a 2-D grid: its eight words at once (independent loads; the early exit below makes each line wait
    // for the one before)
    const unsigned long long* w = reinterpret_cast<const unsigned long long*>(U + i0[0] + i0[1] * fstride[1]);
    const long long ws = fstride[1] >> 3;
    unsigned long long acc = 0ull;
#pragma unroll
    for (int r = 0; r < 8; ++r) acc |= w[r * ws];
    return acc != 0ull;
  }
  for (long long sline = 0; sline < nsub && !any; ++sline) {
    long long r = sline, base = i0[0];
    for (int a = 1; a < cg.d; ++a) {
      base += (i0[a] + r % len[a]) * fstride[a];
      r /= len[a];
    }
    const uint8_t* u = U + base;
    if (len[0] == 8 && ((uintptr_t)u & 7) == 0) {
      any = *(const unsigned long long*)u != 0ull;
    } else {
      for (int k = 0; k < (int)len[0]; ++k) any = any || u[k];
    }
  }
  return any;
}

// axis 0, one workgroup per grid line (count0 <= kAxis0Max): the line's bits go to LDS as 64-bit words (one ballot per
// wave and chunk), waves 0 and 1 scan the words for the nearest set bit before / after every word, and every element
// then finds its neighbours from its own word and the two carries -- no serial chain along the line, the mask is read
// once and the squared distance written once (same arithmetic as k_edt_axis0: (h0 t)^2 with t the step count).
// COARSE: the lines are lines of coarse cells and the bit of a cell is formed here from the fine mask (no coarse mask in
// memory, one launch less).
constexpr int kAxis0Max = 65536;
struct Axis0Lds {
  unsigned long long words[kAxis0Max / 64];
  int lastw[kAxis0Max / 64];    // index of the last set bit in words 0..w (-1: none)
  int firstw[kAxis0Max / 64];   // index of the first set bit in words w.. (INT_MAX: none)
};
// (bid / nblk: this workgroup's index and the number of workgroups on this job -- the job may be one range of a launch)
template <bool COARSE, bool U16 = false>
__device__ __forceinline__ void edt_axis0_wg_body(int bid, int nblk, Axis0Lds& lds, const uint8_t* __restrict__ U, long long nlines,
                                                  int count0, double h0, double* __restrict__ D, const CoarseGrid& cg) {
  unsigned long long* words = lds.words;
  int* lastw = lds.lastw;
  int* firstw = lds.firstw;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nwords = (count0 + 63) >> 6;
  constexpr int kNone = 0x7fffffff;
  for (long long line = bid; line < nlines; line += nblk) {
    const uint8_t* u = U + line * count0;
    double* d = D + line * count0;
    unsigned short* d16 = reinterpret_cast<unsigned short*>(D) + line * count0;     // (U16: the image holds step counts)
    if (!COARSE && (count0 & 63) == 0 && (((uintptr_t)u) & 7) == 0) {
      // fine lines of whole words (r03): a lane reads eight mask bytes at once and drops their eight bits into LDS as one byte
      // of the word array (bit i of word w = position 64 w + i: little-endian bytes are exactly that order) -- 2 loads per lane
      // for a 4096-long line instead of 16 single-byte loads and ballots
      uint8_t* wb = reinterpret_cast<uint8_t*>(words);
      for (int t = threadIdx.x; t < (count0 >> 3); t += blockDim.x) {
        unsigned long long x = reinterpret_cast<const unsigned long long*>(u)[t];
        x |= x >> 4;                         // any bit of a byte -> its bit 0 (the masks hold 0 / 1, but nothing relies on it)
        x |= x >> 2;
        x |= x >> 1;
        x &= 0x0101010101010101ull;
        wb[t] = (uint8_t)((x * 0x0102040810204080ull) >> 56);
      }
    } else {
      for (int w = wave; w < nwords; w += 4) {
        const int i = w * 64 + lane;
        const unsigned long long m = __ballot(i < count0 && (COARSE ? coarse_cell_any(U, cg, line * count0 + i) : u[i] != 0));
        if (lane == 0) words[w] = m;
      }
    }
    __syncthreads();
    if (wave == 0) {
      int carry = -1;
      for (int base = 0; base < nwords; base += 64) {
        const int w = base + lane;
        const unsigned long long m = w < nwords ? words[w] : 0ull;
        int v = m ? w * 64 + (63 - __clzll((long long)m)) : -1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o && t > v) v = t; }
        v = v > carry ? v : carry;
        if (w < nwords) lastw[w] = v;
        carry = __shfl(v, 63);
      }
    } else if (wave == 1) {
      int carry = kNone;
      for (int base = 0; base < nwords; base += 64) {
        const int w = nwords - 1 - (base + lane);
        const unsigned long long m = w >= 0 ? words[w] : 0ull;
        int v = m ? w * 64 + (__ffsll((long long)m) - 1) : kNone;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o && t < v) v = t; }
        v = v < carry ? v : carry;
        if (w >= 0) firstw[w] = v;
        carry = __shfl(v, 63);
      }
    }
    __syncthreads();
    for (int w = wave; w < nwords; w += 4) {
      const int i = w * 64 + lane;
      const unsigned long long m = words[w];
      const unsigned long long lower = m & (lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull));
      const unsigned long long upper = m >> lane;
      const int li = lower ? w * 64 + (63 - __clzll((long long)lower)) : (w > 0 ? lastw[w - 1] : -1);
      const int ri = upper ? i + (__ffsll((long long)upper) - 1) : (w + 1 < nwords ? firstw[w + 1] : kNone);
      if (i < count0) {
        int t = -1;
        if (li >= 0) t = i - li;
        if (ri != kNone && (t < 0 || ri - i < t)) t = ri - i;
        if (U16) {
          d16[i] = t >= 0 ? (unsigned short)t : (unsigned short)0xffffu;
        } else {
          double v = kInfD;
          if (t >= 0) {
            const double dt = h0 * (double)t;
            v = dt * dt;
          }
          d[i] = v;
        }
      }
    }
    __syncthreads();
  }
}
// axis 0 of a fine line by ONE wave, no barriers (r03): lines of whole 64-bit words, at most 4096 positions (<= 64 words: a
// word per lane).  The eight-byte loads drop their bits into the wave's own 512 bytes of LDS as above, lane w then holds word
// w; the nearest set bit before / after every word is a wave scan (six shuffle steps each way); the output loop walks the
// words -- word k and the two carries around it broadcast from lanes k, k - 1, k + 1 -- and lane l writes position 64 k + l:
// coalesced stores, the same values as the workgroup form.  Four independent lines per workgroup instead of one line behind
// three barriers.
template <bool U16>
__device__ __forceinline__ void edt_axis0_wave_body(long long line, unsigned long long* wlds /* this wave's 64 words */,
                                                    const uint8_t* __restrict__ U, int count0, double h0, double* __restrict__ D) {
  const int lane = threadIdx.x & 63;
  const int nwords = count0 >> 6;
  constexpr int kNone = 0x7fffffff;
  const uint8_t* u = U + line * count0;
  uint8_t* wb = reinterpret_cast<uint8_t*>(wlds);
  for (int t = lane; t < (count0 >> 3); t += 64) {
    unsigned long long x = reinterpret_cast<const unsigned long long*>(u)[t];
    x |= x >> 4;
    x |= x >> 2;
    x |= x >> 1;
    x &= 0x0101010101010101ull;
    wb[t] = (uint8_t)((x * 0x0102040810204080ull) >> 56);
  }
  __builtin_amdgcn_wave_barrier();
  const unsigned long long m = lane < nwords ? wlds[lane] : 0ull;
  __builtin_amdgcn_wave_barrier();                      // (the next line of this wave overwrites the words)
  int last = m ? lane * 64 + (63 - __clzll((long long)m)) : -1;            // -> last set bit in words 0 .. lane
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(last, o); if (lane >= o && t > last) last = t; }
  int first = m ? lane * 64 + (__ffsll((long long)m) - 1) : kNone;        // -> first set bit in words lane ..
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_down(first, o); if (lane + o < 64 && t < first) first = t; }
  const unsigned int mlo = (unsigned int)m, mhi = (unsigned int)(m >> 32);
  double* d = D + line * count0;
  unsigned short* d16 = reinterpret_cast<unsigned short*>(D) + line * count0;
  const unsigned long long le_mask = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
  for (int k = 0; k < nwords; ++k) {                      // (k is uniform: the broadcasts are v_readlane)
    const unsigned long long mk = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)mhi, k) << 32) |
                                  (unsigned int)__builtin_amdgcn_readlane((int)mlo, k);
    const int lprev = k > 0 ? __builtin_amdgcn_readlane(last, k - 1) : -1;
    const int fnext = k + 1 < nwords ? __builtin_amdgcn_readlane(first, k + 1) : kNone;
    const int i = k * 64 + lane;
    const unsigned long long lower = mk & le_mask, upper = mk >> lane;
    const int li = lower ? k * 64 + (63 - __clzll((long long)lower)) : lprev;
    const int ri = upper ? i + (__ffsll((long long)upper) - 1) : fnext;
    int t = -1;
    if (li >= 0) t = i - li;
    if (ri != kNone && (t < 0 || ri - i < t)) t = ri - i;
    if (U16) {
      d16[i] = t >= 0 ? (unsigned short)t : (unsigned short)0xffffu;
    } else {
      double v = kInfD;
      if (t >= 0) {
        const double dt = h0 * (double)t;
        v = dt * dt;
      }
      d[i] = v;
    }
  }
}

// The 16-bit image of a fine line by one wave, a word per lane walked bit by bit (r03): the broadcast form above spends ~40
// instructions per element, half of them 64-bit shifts and bit searches (35 us on config H); here a lane carries two counters
// over its own 64 positions -- steps since the last set bit going up, steps to the next one going down: F(i) = bit_i ? 0 : F(i - 1) + 1,
// three 32-bit instructions per step with the bit as a sign-extended field --, seeded by the wave scans' carries, and the 32 packed
// words of a lane turn through the wave's LDS patch so that the stores are whole 1 KB runs.  Same step counts.
constexpr int kA0Big = 1 << 20;       // "no set bit on this side" (stays far above 65535 through 64 increments)
constexpr int kA0Row = 36;            // words per lane row of the patch (16-byte aligned rows)
__device__ __forceinline__ void edt_axis0_wave_seq(long long line, unsigned long long* wlds, unsigned int* patch /* [64][kA0Row] */,
                                                   const uint8_t* __restrict__ U, int count0, double* __restrict__ D) {
  const int lane = threadIdx.x & 63;
  const int nwords = count0 >> 6;
  constexpr int kNone = 0x7fffffff;
  const uint8_t* u = U + line * count0;
  uint8_t* wb = reinterpret_cast<uint8_t*>(wlds);
  for (int t = lane; t < (count0 >> 3); t += 64) {
    unsigned long long x = reinterpret_cast<const unsigned long long*>(u)[t];
    x |= x >> 4;
    x |= x >> 2;
    x |= x >> 1;
    x &= 0x0101010101010101ull;
    wb[t] = (uint8_t)((x * 0x0102040810204080ull) >> 56);
  }
  __builtin_amdgcn_wave_barrier();
  const unsigned long long m = lane < nwords ? wlds[lane] : 0ull;
  __builtin_amdgcn_wave_barrier();
  int last = m ? lane * 64 + (63 - __clzll((long long)m)) : -1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(last, o); if (lane >= o && t > last) last = t; }
  int first = m ? lane * 64 + (__ffsll((long long)m) - 1) : kNone;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_down(first, o); if (lane + o < 64 && t < first) first = t; }
  const int lprev = __shfl_up(last, 1), fnext = __shfl_down(first, 1);
  // F(-1): steps from the last position of the previous word to the last set bit before this word; D(64) likewise upwards
  int f = (lane > 0 && lprev >= 0) ? lane * 64 - 1 - lprev : kA0Big;
  int dn = (lane + 1 < nwords && fnext != kNone) ? fnext - (lane * 64 + 64) : kA0Big;
  const unsigned int nlo = ~(unsigned int)m, nhi = ~(unsigned int)(m >> 32);     // bit clear -> 1
  int F[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const unsigned int h = i < 32 ? nlo : nhi;
    const int keep = (int)(h << (31 - (i & 31))) >> 31;        // -1 when position i holds no U point
    f = (f + 1) & keep;
    F[i] = f;
  }
  unsigned int packed[32];
#pragma unroll
  for (int i = 63; i >= 0; --i) {
    const unsigned int h = i < 32 ? nlo : nhi;
    const int keep = (int)(h << (31 - (i & 31))) >> 31;
    dn = (dn + 1) & keep;
    int t = F[i] < dn ? F[i] : dn;
    t = t > 0xffff ? 0xffff : t;
    if (i & 1) packed[i >> 1] = (unsigned int)t << 16;
    else packed[i >> 1] |= (unsigned int)t;
  }
  if (lane < nwords) {
    unsigned int* row = patch + lane * kA0Row;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      *reinterpret_cast<uint4*>(row + 4 * j) = make_uint4(packed[4 * j], packed[4 * j + 1], packed[4 * j + 2], packed[4 * j + 3]);
  }
  __builtin_amdgcn_wave_barrier();
  // lane l of store j writes the four words 256 j + 4 l .. + 3 of the line: row 8 j + (l >> 3), words 4 (l & 7) .. + 3
  uint4* out = reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(D) + line * count0);
  const int nst = count0 >> 9;                                   // stores of 512 positions; a tail of 64 .. 448 positions below
  for (int j = 0; j < nst; ++j)
    out[j * 64 + lane] = *reinterpret_cast<const uint4*>(patch + (8 * j + (lane >> 3)) * kA0Row + 4 * (lane & 7));
  const int rem_rows = nwords - 8 * nst;                           // whole words left (count0 is a multiple of 64)
  if (rem_rows > 0 && (lane >> 3) < rem_rows)
    out[nst * 64 + lane] = *reinterpret_cast<const uint4*>(patch + (8 * nst + (lane >> 3)) * kA0Row + 4 * (lane & 7));
  __builtin_amdgcn_wave_barrier();
}

template <bool COARSE>
__global__ __launch_bounds__(256) void k_edt_axis0_wg(const uint8_t* __restrict__ U, long long nlines, int count0, double h0,
                                                      double* __restrict__ D, const CoarseGrid cg) {
  __shared__ Axis0Lds lds;
  edt_axis0_wg_body<COARSE>((int)blockIdx.x, (int)gridDim.x, lds, U, nlines, count0, h0, D, cg);
}
// axis 0 as squared distances (grids of three and more axes, and 2-D grids outside the paired launch), a wave per line: lines of
// whole 64-bit words up to 4096 positions.  Config D has 2.1 M lines of 128 positions: a workgroup per line spent three barriers
// on two words of mask (1.6 ms); a wave needs a load, two scans and two rounds of stores.
__global__ __launch_bounds__(256) void k_edt_axis0_waves(const uint8_t* __restrict__ U, long long nlines, int count0, double h0,
                                                         double* __restrict__ D) {
  __shared__ unsigned long long wl[4][64];
  const int wave = threadIdx.x >> 6;
  for (long long line = (long long)blockIdx.x * 4 + wave; line < nlines; line += (long long)gridDim.x * 4)
    edt_axis0_wave_body<false>(line, wl[wave], U, count0, h0, D);
}
// fine and coarse axis-0 passes of a 2-D grid in one launch (both read the U mask only): workgroups [0, nfine) take the
// fine lines, the rest the lines of coarse cells
// (+ one workgroup for the merge of the classification's partials when `fin` is pending: nothing here reads the scalars)
template <bool U16>
__global__ __launch_bounds__(256) void k_edt_axis0_pair(const uint8_t* __restrict__ U, long long nlines, int count0, double h0,
                                                        double* __restrict__ D, int nfine, int ncoarse, long long clines, int cc0,
                                                        double h0c, double* __restrict__ Dc, const CoarseGrid cg, const FinalJob fin,
                                                        int wave_lines /* 1: a fine workgroup is four waves with a line each */) {
  // one block of LDS for the roles a workgroup can have: the workgroup-per-line form's words and carries (16 KB), or four
  // waves' 64 words + store patches (2 + 36 KB)
  constexpr size_t kWaveBytes = 4 * 512 + (U16 ? 4 * 64 * kA0Row * 4 : 0);
  __shared__ __attribute__((aligned(16))) unsigned char a0mem[sizeof(Axis0Lds) > kWaveBytes ? sizeof(Axis0Lds) : kWaveBytes];
  Axis0Lds& lds = *reinterpret_cast<Axis0Lds*>(a0mem);
  // (order of the ranges: the merge and the coarse lines first -- few workgroups with longer chains that should start with the
  // launch, not in the slots the fine lines leave at its end)
  const int nfin = (int)gridDim.x - nfine - ncoarse, bid = (int)blockIdx.x;
  if (bid < nfin) classify_final_body(fin.part, fin.nparts, fin.pcap, fin.q, fin.sc, fin.Lpart, fin.per_out, fin.Lmax, fin.sc_copy, fin.gb, fin.b);
  else if (bid < nfin + ncoarse) edt_axis0_wg_body<true>(bid - nfin, ncoarse, lds, U, clines, cc0, h0c, Dc, cg);
  else if (wave_lines) {
    // (the waves of the fine workgroups stride over the lines: the host sizes the launch to what is resident at once)
    unsigned long long* wwords = reinterpret_cast<unsigned long long*>(a0mem) + (threadIdx.x >> 6) * 64;
    for (long long line = (long long)(bid - nfin - ncoarse) * 4 + (threadIdx.x >> 6); line < nlines; line += (long long)nfine * 4) {
      if constexpr (U16) edt_axis0_wave_seq(line, wwords, reinterpret_cast<unsigned int*>(a0mem + 4 * 512) + (threadIdx.x >> 6) * 64 * kA0Row, U, count0, D);
      else edt_axis0_wave_body<false>(line, wwords, U, count0, h0, D);
    }
  } else edt_axis0_wg_body<false, U16>(bid - nfin - ncoarse, nfine, lds, U, nlines, count0, h0, D, cg);
}

// axes >= 1: D_out[g] = min_t D_in[g + t stride] + (h t)^2, searched outwards with the two exits
//   (h t)^2 >= best  (nothing further can improve)  and  h t > cap  (beyond any radius that matters).
// `accept2`: once best <= accept2 the caller's decision is already "within the radius" and a smaller minimum cannot
// change it, so the search stops (pass -1 to get the exact minimum).
__device__ __forceinline__ double edt_scan_point(const double* __restrict__ Din, long long g, long long stride, int cnt,
                                                 int ia, double h, double cap, double accept2 = -1.0) {
  // Four steps (eight loads) are issued per round of exit tests: the serial chain of the search is what a thread
  // waits for, and a step examined beyond an exit cannot lower the minimum (its candidate is >= (h t)^2 >= best; steps
  // beyond the cap are masked as before).
  double best = Din[g];
  for (int t = 1; t < cnt; t += 4) {
    const double dt = h * (double)t;
    if (dt * dt >= best || dt > cap || best <= accept2) break;
    if (ia - t < 0 && ia + t >= cnt) break;
    double c1[4], c2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tt = t + u;
      c1[u] = ia - tt >= 0 ? Din[g - (long long)tt * stride] : kInfD;
      c2[u] = ia + tt < cnt ? Din[g + (long long)tt * stride] : kInfD;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double du = h * (double)(t + u);
      const double c = (c1[u] < c2[u] ? c1[u] : c2[u]) + du * du;
      if (du <= cap && c < best) best = c;
    }
  }
  return best;
}

// Last-axis scans with a one-level min-pyramid.  Bmin[b * stride + p] = min of the input over the kBlk (or `blk`) steps
// of block b at in-plane position p.  A block whose bound  Bmin + (h gap)^2  cannot beat the running minimum is skipped
// with one load instead of `blk`; the candidates examined inside a block and their arithmetic are those of the
// step-by-step scan, so the minimum (up to the same early exits) is identical.  This keeps the scan cost near
// O(sqrt(radius in steps)) when the grid is much finer along the last axis than the radius (weak-scaling grids).
template <typename DS>
__device__ __forceinline__ void block_min_body(int bid, int nwg, double (&part)[4][64], const DS Din, long long stride,
                                               int cnt, int blk, double* __restrict__ Bmin) {
  // 64 in-plane positions x 4 quarters of a block per workgroup: a thread takes blk / 4 steps (all loads in flight),
  // the quarters meet in LDS
  const int nblk = (cnt + blk - 1) / blk;
  const int px = threadIdx.x & 63, sub = threadIdx.x >> 6, per = (blk + 3) / 4;
  const long long ptiles = (stride + 63) / 64, total = ptiles * nblk;
  for (long long w = bid; w < total; w += nwg) {
    const long long p = (w % ptiles) * 64 + px;
    const int b = (int)(w / ptiles);
    const int j0 = b * blk + sub * per;
    const int jend = (b + 1) * blk < cnt ? (b + 1) * blk : cnt;
    const int j1 = j0 + per < jend ? j0 + per : jend;
    double m = kInfD;
    if (p < stride) {
#pragma unroll 8
      for (int j = j0; j < j1; ++j) {
        const double v = Din((long long)j * stride + p);
        m = v < m ? v : m;
      }
    }
    part[sub][px] = m;
    __syncthreads();
    if (sub == 0 && p < stride) {
      const double a = part[0][px] < part[1][px] ? part[0][px] : part[1][px];
      const double c = part[2][px] < part[3][px] ? part[2][px] : part[3][px];
      Bmin[(long long)b * stride + p] = a < c ? a : c;
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void k_block_min(const double* __restrict__ Din, long long stride, int cnt, int blk,
                                                   double* __restrict__ Bmin) {
  __shared__ double part[4][64];
  block_min_body((int)blockIdx.x, (int)gridDim.x, part, DistF64{Din, 0.0}, stride, cnt, blk, Bmin);
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

__device__ __forceinline__ void edt_scan_body(int bid, int nwg, const double* __restrict__ Din, double* __restrict__ Dout, long long n,
                                              long long stride, int cnt, double h, const SweepScalars* sc, int c,
                                              const unsigned long long* Lkeys, int lidx, int uncapped, double cap_extra) {
  // cap: offsets beyond the largest radius that can matter are not examined.  The coarse transform passes
  // cap_extra = 2 delta + two coarse steps: a cell whose true distance lies beyond that cap is "beyond every radius" by
  // the sandwich dC -+ delta whether its stored value is the true minimum or a larger one.
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const double rmax = sc->rmax_key[c] ? ord_val(sc->rmax_key[c]) : 0.0;
  const double cap = (L > 0 && !uncapped) ? rmax / L * 1.000001 + 1e-6 + cap_extra : kInfD;
  for (long long g = (long long)bid * blockDim.x + threadIdx.x; g < n; g += (long long)nwg * blockDim.x) {
    const int ia = (int)((g / stride) % cnt);
    Dout[g] = edt_scan_point(Din, g, stride, cnt, ia, h, cap);
  }
}
__global__ __launch_bounds__(256) void k_edt_scan(const double* __restrict__ Din, double* __restrict__ Dout, long long n,
                                                  long long stride, int cnt, double h, const SweepScalars* sc, int c,
                                                  const unsigned long long* Lkeys, int lidx, int uncapped, double cap_extra) {
  edt_scan_body((int)blockIdx.x, (int)gridDim.x, Din, Dout, n, stride, cnt, h, sc, c, Lkeys, lidx, uncapped, cap_extra);
}

// Coarse pre-decision for the expander query.  The U mask is OR-reduced over cells of kCoarse^d candidates and the
// exact transform of that small mask gives, for any candidate g in cell C, the sandwich
//     dC - delta <= dist(g, U) <= dC + delta,   delta = (kCoarse - 1) * sqrt(sum_a h_a^2)
// (dC = distance between cell origins to the nearest U-holding cell).  Candidates whose verdict is the same at both
// ends skip the per-candidate scan of the fine transform; only the shell around the boundary of G_c scans.
// one thread per coarse cell (fallback when the coarse axis-0 pass cannot take the fine mask itself)
__global__ __launch_bounds__(256) void k_coarsen_mask(const uint8_t* __restrict__ U, const CoarseGrid cg, long long ncells,
                                                      uint8_t* __restrict__ Uc) {
  for (long long cell = (long long)blockIdx.x * blockDim.x + threadIdx.x; cell < ncells; cell += (long long)gridDim.x * blockDim.x)
    Uc[cell] = coarse_cell_any(U, cg, cell) ? 1 : 0;
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

// fp32 models with fp64 recheck (sets_recheck.inc.hpp): the verdict kernels see a posterior whose UNREFINED entries are
// fp32 values, good to +- (dm, dv).  For those the ucb is an interval of half-width du; a verdict the interval cannot
// settle sends the candidate to the refinement list instead of deciding it.  refined == nullptr: every value is exact.
constexpr size_t kRcCount2 = 32, kRcGKeys = 64, kRcList = 256;   // layout of sbo_ctx::rc_list: counters, G keys, the list
// The guard band of an approximating fp64 posterior (device_common.hpp: GuardBand) rides on the same mechanism: gb_c >= 1 names
// the constraint whose band SweepScalars::gb_du[gb_c] (one bound for every candidate of S) and relative Lipschitz band gb_rl
// apply to entries that are not `refined`; on the fast path there is no list and an unsettled verdict is only counted
// (SweepScalars::n_guard).
struct RcExp {
  const uint8_t* refined;
  double dm, dv;
  long long* list;                 // candidates to re-evaluate exactly (nullptr: count only)
  unsigned long long* count;
  int gb_c, gb_l;                  // guard band: constraint / Lipschitz index (0: no guard band)
};
struct RcBandK {                   // per-kernel constants of the guard band (loaded once from the scalar block)
  double du, rl;
};
__device__ __forceinline__ RcBandK rc_band(const RcExp& rx, const SweepScalars* sc) {
  RcBandK k{0.0, 0.0};
  if (rx.gb_c > 0) { k.du = sc->gb_du[rx.gb_c]; k.rl = rx.gb_l >= 0 ? sc->gb_rl[rx.gb_l] : 0.0; }
  return k;
}
__device__ __forceinline__ double rc_du(const RcExp& rx, const RcBandK& bk, long long g, double var, double b, double ucb) {
  if (rx.refined && rx.refined[g]) return 0.0;
  if (rx.gb_c > 0) return bk.du + bk.rl * (ucb < 0 ? -ucb : ucb);      // (a relative band of L moves L dist by at most rl ucb at the boundary)
  if (!rx.refined) return 0.0;
  return rx.dm + b * (sqrt(var + rx.dv) - sqrt(fmax(0.0, var - rx.dv)));
}
__device__ __forceinline__ void rc_defer(const RcExp& rx, SweepScalars* sc, long long g) {
  if (rx.list) rx.list[atomicAdd(rx.count, 1ull)] = g;
  else atomicAdd((unsigned long long*)&sc->n_guard, 1ull);
}

// last axis + decision.  For g in S: nearest-U distance dm (unshifted) -> G bit, or the ambiguous list when
// ucb - L dm lies inside the band the "+1e-8" shift and rounding can move it across zero.
// LIST: open candidates go to the scan list (the usual grid path); otherwise they are scanned here by their own thread.
template <typename T, bool LIST>
__global__ __launch_bounds__(256) void k_edt_decide(const double* __restrict__ Din, long long nl, int len0, long long line0,
                                                    long long goff, long long stride, int cnt,
                                                    double h, int d, double xscale, const T* __restrict__ mean_c,
                                                    const T* __restrict__ var_c, T b, const uint8_t* __restrict__ S,
                                                    const unsigned long long* Lkeys, int lidx, SweepScalars* sc,
                                                    uint8_t* __restrict__ G, long long* __restrict__ amb,
                                                    const CoarseGrid cg, const double* __restrict__ Bmin, int blk,
                                                    long long* __restrict__ scanlist, const RcExp rx) {
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const bool anyU = sc->count_U > 0;
  const RcBandK bk = rc_band(rx, sc);
  // launch: x over the positions of a grid line (len0 = count0; the whole range when d == 1), y over blocks of
  // kDecideLines local lines -- the line index is uniform per workgroup and a 2-D grid needs no division to split a
  // candidate index.  Open candidates are collected in LDS and appended to the scan list with one atomic per
  // workgroup (one per wave on a single counter serialises in L2).
  __shared__ long long sl[256 * kDecideLines];
  __shared__ double su[256 * kDecideLines];
  __shared__ int scnt;
  __shared__ long long sbase;
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i0 < len0;
  const long long nlb = (nl + kDecideLines - 1) / kDecideLines;
  for (long long lb = blockIdx.y; lb < nlb; lb += gridDim.y) {
  if (threadIdx.x == 0) scnt = 0;
  __syncthreads();
  // the thread's kDecideLines candidates: mask bytes, then mean / var of the safe ones, then their coarse distances are
  // loaded for all lines before the first verdict (a line at a time would wait out three dependent loads per line)
  uint8_t sfl[kDecideLines];
  T mu[kDecideLines], va[kDecideLines];
  double dcv[kDecideLines];
  long long cellv[kDecideLines];
  int iav[kDecideLines];
#pragma unroll
  for (int u = 0; u < kDecideLines; ++u) {
    const long long ln = lb * kDecideLines + u;
    sfl[u] = (active && ln < nl) ? S[ln * len0 + i0] : 0;
    long long f = line0 + ln, cell = i0 / kCoarse, ccs = cg.ccount[0];
    int ia = 0;                          // index along the last axis (inside the window)
    for (int a = 1; a < d; ++a) {
      const long long ix = a == d - 1 ? f : f % cg.count[a];
      f = a == d - 1 ? 0 : f / cg.count[a];
      cell += (ix / kCoarse) * ccs;
      ccs *= cg.ccount[a];
      ia = (int)ix;
    }
    cellv[u] = cell;
    iav[u] = ia;
  }
#pragma unroll
  for (int u = 0; u < kDecideLines; ++u) {
    const bool on = sfl[u] && anyU;
    const long long g = (lb * kDecideLines + u) * len0 + i0;
    mu[u] = on ? mean_c[g] : (T)0;
    va[u] = on ? var_c[g] : (T)0;
    dcv[u] = (on && cg.enabled) ? cg.Dc[cellv[u]] : 0.0;
  }
  auto decide = [&](int u) {
    const long long ln = lb * kDecideLines + u;
    if (!(active && ln < nl)) return;
    const long long g = ln * len0 + i0;
    uint8_t out = 0;
    if (sfl[u] && anyU) {
      T lcb, ucbT;
      lcb_ucb(mu[u], va[u], b, lcb, ucbT);
      const double ucb = (double)ucbT;
      const double du = rc_du(rx, bk, g, (double)va[u], (double)b, ucb);   // 0 unless this entry is an unrefined fp32 value / carries a guard band
      if (!(L > 0)) {
        out = ucb >= 0.0;                         // radius unbounded: any U point is a witness
        if (du > 0.0 && fabs(ucb) <= du) rc_defer(rx, sc, g);
      } else {
        const double eps_abs = 1.01e-8 * sqrt((double)d) + 1e-14 * xscale + 1e-13;
        const double cap = (ucb + du) / L + 4.0 * eps_abs + 1e-9 * fabs((ucb + du) / L);
        const long long gg = goff + g;   // index in the transform (whole grid when ranks share it)
        const int ia = iav[u];
        if (cg.enabled) {
          const double dC = sqrt(dcv[u]);
          const double dhi = dC * (1.0 + 1e-9) + cg.delta, dlo = fmax(0.0, dC * (1.0 - 1e-9) - cg.delta);
          const double tolc = 1e-12 * (fabs(ucb) + du + L * dhi);
          if (ucb - du - L * (dhi + eps_abs + 1e-11 * dhi) > tolc) { G[g] = 1; return; }     // within the radius for sure
          if (ucb + du - L * (dlo - eps_abs - 1e-11 * dlo) < -tolc) { G[g] = 0; return; }    // beyond it for sure
        }
        if (LIST) {
          // the few candidates the coarse bounds leave open go to k_edt_scan_list (a group of lanes each): a lane
          // scanning here would hold its whole wave for a chain of ~100 dependent loads
          const int slot = atomicAdd(&scnt, 1);
          sl[slot] = g;
          su[slot] = ucb;                          // (rides along: the list kernel need not read mean / var again)
          G[g] = 0;
          return;
        }
        const double thr = (ucb - du) / L * (1.0 - 1e-10) - 2.0 * eps_abs - 1e-12;   // inside it the verdict is "sure true"
        const double acc2 = thr > 0 ? thr * thr : -1.0;
        const double best = cnt <= 1 ? Din[gg]
                            : Bmin  ? edt_scan_blocked(Din, Bmin, gg - (long long)ia * stride, stride, cnt, ia, h, cap, acc2, blk)
                                    : edt_scan_point(Din, gg, stride, cnt, ia, h, cap, acc2);
        if (best < 0.5 * kInfD) {
          const double dm = sqrt(best);
          const double eps = eps_abs + 1e-11 * dm;
          const double tol = 1e-12 * (fabs(ucb) + du + L * dm);
          const double lo = ucb - du - L * (dm + eps), hi = ucb + du - L * (dm - eps);
          if (lo > tol) out = 1;
          else if (hi >= -tol) {
            if (du > 0.0 && rx.list) {
              rc_defer(rx, sc, g);                 // an fp32 value cannot settle it: re-evaluate, decide in the next pass
              out = ucb - L * dm >= 0.0;
            } else {
              // (guard band, fast path: the exhaustive recheck decides it AND judges whether the band could move its verdict --
              // counting here charged every candidate inside the reference's 1e-8 shift to the band: 119 false alarms per
              // sweep of config H on four ranks, each a second pass)
              const long long slot = (long long)atomicAdd((unsigned long long*)&sc->n_amb, 1ull);
              amb[slot] = g;
            }
          }
        }
      }
    }
    G[g] = out;
  };
#pragma unroll
  for (int u = 0; u < kDecideLines; ++u) decide(u);
  __syncthreads();
  const int cntl = scnt;
  if (cntl > 0) {
    if (threadIdx.x == 0) sbase = (long long)atomicAdd((unsigned long long*)&sc->n_scan, (unsigned long long)cntl);
    __syncthreads();
    for (int k = threadIdx.x; k < cntl; k += blockDim.x) {      // entries: (candidate, ucb) pairs of 16 bytes
      scanlist[2 * (sbase + k)] = sl[k];
      reinterpret_cast<double*>(scanlist)[2 * (sbase + k) + 1] = su[k];
    }
  }
  __syncthreads();
  }
}

// The same verdicts with EIGHT consecutive candidates of a grid line per lane (r03; LIST form only, line length a multiple of
// 8).  Four fifths of a grid are not safe: a lane of k_edt_decide then loads one zero mask byte and stores one zero byte per
// candidate, and that path was most of the kernel (76 us on config H).  Here a lane reads the eight S bytes as one word and
// stores the eight G bytes as one word; a zero word costs two instructions.  Eight positions along axis 0 are exactly one coarse
// cell (kCoarse = 8), so the coarse distance is one load per lane; the mean / var loads of the set bytes are issued together
// before the first verdict.  Same arithmetic per candidate, identical masks and lists (as sets: the list order differs).
template <typename T>
__global__ __launch_bounds__(256) void k_edt_decide8(long long nl, int len0, long long line0, long long goff, int d, double xscale,
                                                     const T* __restrict__ mean_c, const T* __restrict__ var_c, T b,
                                                     const uint8_t* __restrict__ S, const unsigned long long* Lkeys, int lidx,
                                                     SweepScalars* sc, uint8_t* __restrict__ G, const CoarseGrid cg,
                                                     long long* __restrict__ scanlist, const RcExp rx) {
  static_assert(kCoarse == 8, "a lane's eight candidates are one coarse cell");
  constexpr int kCap = 1024;                     // open candidates a workgroup collects in LDS (beyond: straight to the list)
  __shared__ long long sl[kCap];
  __shared__ double su[kCap];
  __shared__ int scnt;
  __shared__ long long sbase;
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const bool anyU = sc->count_U > 0;
  const RcBandK bk = rc_band(rx, sc);
  const int t8 = blockIdx.x * blockDim.x + threadIdx.x;          // this lane's cell along axis 0
  const bool active = (long long)t8 * 8 < len0;
  const double eps_abs = 1.01e-8 * sqrt((double)d) + 1e-14 * xscale + 1e-13;
  for (long long ln = blockIdx.y; ln < nl; ln += gridDim.y) {
    if (threadIdx.x == 0) scnt = 0;
    __syncthreads();
    if (active) {
      const long long g0 = ln * len0 + (long long)t8 * 8;
      const unsigned long long w = anyU ? *reinterpret_cast<const unsigned long long*>(S + g0) : 0ull;
      unsigned long long gw = 0ull;
      if (w != 0ull) {
        // coarse cell of this lane (axes >= 1 from the line index, as k_edt_decide)
        long long f = line0 + ln, cell = t8, ccs = cg.ccount[0];
        for (int a = 1; a < d; ++a) {
          const long long ix = a == d - 1 ? f : f % cg.count[a];
          f = a == d - 1 ? 0 : f / cg.count[a];
          cell += (ix / kCoarse) * ccs;
          ccs *= cg.ccount[a];
        }
        const double dC = cg.enabled ? sqrt(cg.Dc[cell]) : 0.0;
        const double dhi = dC * (1.0 + 1e-9) + cg.delta, dlo = fmax(0.0, dC * (1.0 - 1e-9) - cg.delta);
        T mu[8], va[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const bool on = ((w >> (8 * k)) & 0xffull) != 0ull;
          mu[k] = on ? mean_c[g0 + k] : (T)0;
          va[k] = on ? var_c[g0 + k] : (T)0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (((w >> (8 * k)) & 0xffull) == 0ull) continue;
          const long long g = g0 + k;
          T lcb, ucbT;
          lcb_ucb(mu[k], va[k], b, lcb, ucbT);
          const double ucb = (double)ucbT;
          const double du = rc_du(rx, bk, g, (double)va[k], (double)b, ucb);   // 0 unless this entry is an unrefined fp32 value / carries a guard band
          if (!(L > 0)) {
            if (ucb >= 0.0) gw |= 1ull << (8 * k);                      // radius unbounded: any U point is a witness
            if (du > 0.0 && fabs(ucb) <= du) rc_defer(rx, sc, g);
            continue;
          }
          if (cg.enabled) {
            const double tolc = 1e-12 * (fabs(ucb) + du + L * dhi);
            if (ucb - du - L * (dhi + eps_abs + 1e-11 * dhi) > tolc) { gw |= 1ull << (8 * k); continue; }   // within the radius for sure
            if (ucb + du - L * (dlo - eps_abs - 1e-11 * dlo) < -tolc) continue;                              // beyond it for sure
          }
          const int slot = atomicAdd(&scnt, 1);
          if (slot < kCap) {
            sl[slot] = g;
            su[slot] = ucb;
          } else {                                                       // (a workgroup with more than kCap open candidates)
            const long long gs = (long long)atomicAdd((unsigned long long*)&sc->n_scan, 1ull);
            scanlist[2 * gs] = g;
            reinterpret_cast<double*>(scanlist)[2 * gs + 1] = ucb;
          }
        }
      }
      *reinterpret_cast<unsigned long long*>(G + g0) = gw;
    }
    __syncthreads();
    const int cntl = scnt < kCap ? scnt : kCap;
    if (cntl > 0) {
      if (threadIdx.x == 0) sbase = (long long)atomicAdd((unsigned long long*)&sc->n_scan, (unsigned long long)cntl);
      __syncthreads();
      for (int k = threadIdx.x; k < cntl; k += blockDim.x) {      // entries: (candidate, ucb) pairs of 16 bytes
        scanlist[2 * (sbase + k)] = sl[k];
        reinterpret_cast<double*>(scanlist)[2 * (sbase + k) + 1] = su[k];
      }
    }
    __syncthreads();
  }
}

// Last-axis scan + verdict for the listed candidates, one group of GL lanes (half a wave or a wave, GL >= blk) per
// candidate: the lanes take GL blocks (bounds) or the steps of one block at a time and combine with group minima -- the
// same candidates and arithmetic as edt_scan_blocked, with the dependent-load chain cut from ~100 to ~5 per candidate.
// (The two halves of a wave follow their own trip counts; every cross-lane operation stays inside one half.)
template <typename T, int GL, typename DS>
__global__ __launch_bounds__(256) void k_edt_scan_list(const DS Din, long long goff, long long stride, int cnt,
                                                       double h, int d, double xscale, const T* __restrict__ mean_c,
                                                       const T* __restrict__ var_c, T b, const unsigned long long* Lkeys, int lidx,
                                                       SweepScalars* sc, uint8_t* __restrict__ G, long long* __restrict__ amb,
                                                       const double* __restrict__ Bmin, int blk,
                                                       const long long* __restrict__ scanlist, const RcExp rx) {
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const long long nscan = sc->n_scan;
  const RcBandK bk = rc_band(rx, sc);
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
    return GL == 64 ? m : ((m >> (GL * sub)) & ((1ull << GL) - 1ull));
  };
  for (long long qi = (long long)blockIdx.x * (blockDim.x / GL) + threadIdx.x / GL; qi < nscan; qi += ngroups) {
    const long long g = scanlist[2 * qi];
    const double ucb = reinterpret_cast<const double*>(scanlist)[2 * qi + 1];
    const double du = (rx.refined || rx.gb_c > 0) ? rc_du(rx, bk, g, (rx.refined && !(rx.gb_c > 0)) ? (double)var_c[g] : 0.0, (double)b, ucb) : 0.0;
    const double eps_abs = 1.01e-8 * sqrt((double)d) + 1e-14 * xscale + 1e-13;
    const double cap = (ucb + du) / L + 4.0 * eps_abs + 1e-9 * fabs((ucb + du) / L);
    const double thr = (ucb - du) / L * (1.0 - 1e-10) - 2.0 * eps_abs - 1e-12;
    const double acc2 = thr > 0 ? thr * thr : -1.0;
    const long long gg = goff + g, p = gg % stride;
    const int ia = (int)((gg / stride) % cnt), b0 = ia / blk;
    double best = Din((long long)ia * stride + p);
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
    auto scan_block = [&](int bb) {     // the group: the steps of block bb, GL at a time (loads of all rounds in flight)
      double cnd = kInfD;
#pragma unroll 4
      for (int s0 = 0; s0 < blk; s0 += GL) {
        const int jn = bb * blk + s0 + lane;
        if (s0 + lane < blk && jn < cnt) {
          const double dt = h * (double)(jn > ia ? jn - ia : ia - jn);
          if (dt <= cap) {
            const double v = Din((long long)jn * stride + p) + dt * dt;
            cnd = v < cnd ? v : cnd;
          }
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
        const double tol = 1e-12 * (fabs(ucb) + du + L * dm);
        const double lo = ucb - du - L * (dm + eps), hi = ucb + du - L * (dm - eps);
        if (lo > tol) out = 1;
        else if (hi >= -tol) {
          if (du > 0.0 && rx.list) {
            rc_defer(rx, sc, g);
            out = ucb - L * dm >= 0.0;
          } else {
            amb[atomicAdd((unsigned long long*)&sc->n_amb, 1ull)] = g;     // (guard band: judged by k_expander_exact)
          }
        }
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
                                                        const long long* __restrict__ amb, uint8_t* __restrict__ G, const RcExp rx) {
  const double L = __longlong_as_double((long long)Lkeys[lidx]);
  const long long namb = sc->n_amb;
  // guard band of an approximating posterior (fast path): a listed candidate's verdict is in the band iff the predicate has a
  // witness with ucb + du but none with ucb - du (du carries the relative band of L as well: rc_du)
  const bool gband = rx.gb_c > 0 && !rx.list;
  const RcBandK bk = rc_band(rx, sc);
  constexpr int kParts = 64;   // a listed candidate's box can be as large as the grid: cut into slices, one workgroup each
  for (long long wi = blockIdx.x; wi < namb * kParts; wi += gridDim.x) {
    const long long qi = wi / kParts;
    const int part = (int)(wi % kParts);
    const long long g = amb[qi];
    T lcb, ucbT;
    lcb_ucb(mean_c[g], var_c[g], b, lcb, ucbT);
    const double ucb = (double)ucbT;
    const double du = gband ? rc_du(rx, bk, g, 0.0, (double)b, ucb) : 0.0;
    double xg[D];
    cand_coords<D>(cs, g, xg);
    int found = 0, found_hi = 0, found_lo = 0;
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
      for (long long t = part * chunk + threadIdx.x; t < t1 && !(gband ? found_lo : found); t += blockDim.x) {   // (band: until a SURE witness)
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
        if (hl >= 0 && hl < csU.n_local && U[hl]) {
          if (lipschitz_pair<D>(xg, xh, cs.d, ucb, L)) found = 1;
          if (gband) {
            if (lipschitz_pair<D>(xg, xh, cs.d, ucb + du, L)) found_hi = 1;
            if (lipschitz_pair<D>(xg, xh, cs.d, ucb - du, L)) found_lo = 1;
          }
        }
      }
    } else {
      for (long long hh = (long long)part * blockDim.x + threadIdx.x; hh < csU.n_local && !(gband ? found_lo : found); hh += (long long)kParts * blockDim.x) {
        if (U[hh]) {
          double xh[D];
          cand_coords<D>(csU, hh, xh);
          if (lipschitz_pair<D>(xg, xh, cs.d, ucb, L)) found = 1;
          if (gband) {
            if (lipschitz_pair<D>(xg, xh, cs.d, ucb + du, L)) found_hi = 1;
            if (lipschitz_pair<D>(xg, xh, cs.d, ucb - du, L)) found_lo = 1;
          }
        }
      }
    }
    found = __syncthreads_or(found);
    if (gband) {
      // (per slice: a witness under the wide bound only.  A sure witness in another slice makes this a false alarm -- rare, and safe)
      found_hi = __syncthreads_or(found_hi);
      found_lo = __syncthreads_or(found_lo);
      if (threadIdx.x == 0 && found_hi && !found_lo) atomicAdd((unsigned long long*)&sc->n_guard, 1ull);
    }
    if (threadIdx.x == 0 && found) G[g] = 1;
    __syncthreads();
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    sc->n_amb_total += namb;
  }
}

__global__ void k_reset_amb(SweepScalars* sc) { sc->n_amb = 0; sc->n_scan = 0; }

