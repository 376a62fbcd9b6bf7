// sets_exchange.inc.hpp: pack / unpack kernels of the multi-rank collectives C1-C3 -- part of the sets.hip translation unit (included inside namespace sbo; not a standalone header).
#pragma once

// ---- multi-rank exchange -----------------------------------------------------------
// C1: one max all-reduce of [~u*_key, L keys, radius keys, guard-band widths of the bounds on S (non-negative doubles: their bit
// patterns order like the values -- u* is a minimum over all ranks, so its band is the widest rank's)]
constexpr int kC1Words = 1 + 3 * kMaxQ;
__global__ void k_pack_c1(const SweepScalars* sc, const unsigned long long* Lkeys, unsigned long long* buf) {
  const int t = threadIdx.x;
  if (t == 0) buf[0] = ~sc->ustar_key;
  if (t < kMaxQ) {
    buf[1 + t] = Lkeys[t];
    buf[1 + kMaxQ + t] = sc->rmax_key[t];
    buf[1 + 2 * kMaxQ + t] = (unsigned long long)__double_as_longlong(sc->gb_du[t]);
  }
}
__global__ void k_unpack_c1(SweepScalars* sc, unsigned long long* Lkeys, const unsigned long long* buf) {
  const int t = threadIdx.x;
  if (t == 0) sc->ustar_key = ~buf[0];
  if (t < kMaxQ) {
    Lkeys[t] = buf[1 + t];
    sc->rmax_key[t] = buf[1 + kMaxQ + t];
    sc->gb_du[t] = __longlong_as_double((long long)buf[1 + 2 * kMaxQ + t]);
  }
}
// C1 riding at the head of the C2 all-gather: block r of `recv` (stride words) starts with rank r's kC1Head key words;
// their maxima go to the scalar block, the Lipschitz keys and (contiguously) to kb for the host's read-back
constexpr int kC1Head = 32;      // >= 1 + 2 kMaxQ, keeps the bit words 256-byte aligned
static_assert(kC1Head >= kC1Words, "C1 head");
__global__ void k_unpack_c1_gathered(SweepScalars* sc, unsigned long long* Lkeys, const unsigned long long* __restrict__ recv,
                                     long long stride, int world, unsigned long long* __restrict__ kb) {
  const int t = threadIdx.x;
  if (t >= kC1Words) return;
  unsigned long long m = 0ull;
  for (int r = 0; r < world; ++r) {
    const unsigned long long v = recv[(size_t)r * stride + t];
    m = v > m ? v : m;
  }
  if (t < 1 + 2 * kMaxQ) kb[t] = m;
  if (t == 0) sc->ustar_key = ~m;
  else if (t < 1 + kMaxQ) Lkeys[t - 1] = m;
  else if (t < 1 + 2 * kMaxQ) sc->rmax_key[t - 1 - kMaxQ] = m;
  else sc->gb_du[t - 1 - 2 * kMaxQ] = __longlong_as_double((long long)m);
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
// C2 (bit form): own mask -> one word per 64 candidates (zero beyond n); gathered words -> byte mask of a window
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
// (only the flat range [g0, g0 + total) a rank's transform window covers is expanded to bytes: out[g - g0])
__global__ __launch_bounds__(256) void k_unpack_shards(const unsigned long long* __restrict__ recv, long long words, int world,
                                                       const long long* __restrict__ first_of, long long g0, long long total,
                                                       uint8_t* __restrict__ out) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long g = g0 + t;
    int r = 0;
    while (r + 1 < world && g >= first_of[r + 1]) ++r;
    const long long l = g - first_of[r];
    out[t] = (uint8_t)((recv[(size_t)r * words + (l >> 6)] >> (l & 63)) & 1ull);
  }
}
// C3: each rank fills its own row of [world][kC3Row] doubles, one sum all-reduce delivers every row everywhere.  Row: values,
// indices, then per slot the winner's band and the two extreme far ends with the first one's candidate (guard band of an
// approximating posterior: the host merges them like the values), counts, the guard-band count, |G_c| / |O_c|
constexpr int kC3Counts = 6 * kArgSlots;
constexpr int kC3Row = kC3Counts + 6 + kMaxQ;
constexpr int kC3Halo = kC3Counts + 5 + kMaxQ;             // some rank's speculative halo window was too narrow
__global__ void k_pack_c3(const SweepScalars* sc, double* buf, int world, int rank) {
  for (int i = threadIdx.x; i < world * kC3Row; i += blockDim.x) buf[i] = 0.0;
  __syncthreads();
  double* row = buf + (size_t)rank * kC3Row;
  const int t = threadIdx.x;
  if (t < kArgSlots) {
    row[t] = sc->arg_idx[t] >= 0 ? sc->arg_val[t] : 0.0;
    row[kArgSlots + t] = (double)sc->arg_idx[t];
    row[2 * kArgSlots + t] = sc->arg_d[t];
    row[3 * kArgSlots + t] = sc->arg_e1[t];
    row[4 * kArgSlots + t] = sc->arg_e2[t];
    row[5 * kArgSlots + t] = (double)sc->arg_ei[t];
  }
  if (t == 0) {
    row[kC3Counts + 0] = (double)sc->count_S;
    row[kC3Counts + 1] = (double)sc->count_U;
    row[kC3Counts + 2] = (double)sc->count_M;
    row[kC3Counts + 3] = (double)sc->n_amb_total;
    // (the arg-reductions are judged again after the merge of the ranks' rows: only the counted decisions travel here)
    row[kC3Counts + 4] = (double)(sc->n_guard + sc->guard_nb0);
  }
  if (t < kMaxQ) row[kC3Counts + 5 + t] = (double)sc->count_set[t];
  if (t == 0) row[kC3Halo] = (double)sc->halo_short;
}
