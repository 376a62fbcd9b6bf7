// sets_recheck.inc.hpp: fp32 sweeps whose decisions equal the fp64 result -- part of the sets.hip translation unit (included
// inside namespace sbo; not a standalone header).
//
// SURVEY.md section 7, hard part 2: `lcb >= 0` flips for candidates whose bound lies inside the rounding band of the fp32
// posterior.  The fp32 kernels keep mean / var within 1e-4 (normalised units) of the fp64 values -- their contract, checked by
// the parity tests at <= 1e-5 --, so every decision is first taken with INTERVALS: a candidate's bounds are known to
// +- (dm, dv), u* to [u_lo, u_hi], the largest variance over M from below.  Candidates whose intervals cannot decide
//   (iv) the sign of a constraint's lcb                          (S, U)
//   (i)  whether they attain u* = min_S ucb_0                    (u*)
//   (ii) lcb_0 <= u*                                             (M)
//   (iii) whether they hold the largest var_0 over M             (Minimizer's arg-max)
// are listed, their posterior is re-evaluated in fp64 by the model's fp64 twin (generic K1 kernel on the compacted list),
// and the whole set phase then runs in fp64 arithmetic on the widened fp32 posterior with the listed entries replaced.
// S, U, u*, M and the minimiser are then those of an fp64 sweep.  The expander sets need two more things: the Lipschitz
// constant in fp64 (k_rc_grad64: the mean gradient is O(n d) per candidate, no contraction) and verdict kernels that
// carry the band of an unrefined ucb (RcExp in sets_expander.inc.hpp): a verdict the band cannot settle defers the
// candidate, and so does an unrefined member of G_c that could hold its largest variance.  Deferred candidates are
// re-evaluated and the set phase runs again, until nothing is deferred (each candidate is refined at most once).
//
// r04: the same machinery serves the GUARD BAND of an approximating fp64 posterior (K1b / K1t, device_common.hpp: GuardBand): a
// sweep whose fast pass counted decisions inside the band (SweepScalars::n_guard) comes here with TP = double -- the intervals
// are +- (dm, dv) of the plan's band, the listed candidates are re-evaluated by the exact kernel (guard_exact_list) and their
// values replace the approximate ones IN PLACE, the Lipschitz keys are recomputed exactly, and the set phase runs again.
#pragma once

struct RcBand {
  double dm[kMaxQ], dv[kMaxQ];
};
struct RcScal {                         // head of rc_list (256 bytes): this struct, the deferral counter at byte 32, G keys at 64
  unsigned long long ulo_key, uhi_key, vmax_key;
  long long count;
};

__device__ __forceinline__ void rc_interval(double m, double v, double b, double dm, double dv, double& lcb_lo, double& lcb_hi, double& ucb_lo,
                                            double& ucb_hi) {
  const double md = (double)m, vd = (double)v;
  const double s_hi = b * sqrt(vd + dv), s_lo = b * sqrt(fmax(0.0, vd - dv));
  lcb_lo = (md - dm) - s_hi;
  lcb_hi = (md + dm) - s_lo;
  ucb_lo = (md - dm) + s_lo;
  ucb_hi = (md + dm) + s_hi;
}

// possibly / surely safe from the constraints' intervals; `undecided`: some constraint's lcb interval contains zero
template <typename TP>
__device__ __forceinline__ void rc_safety(const TP* __restrict__ mean, const TP* __restrict__ var, long long n, long long g, int q,
                                          double b, const RcBand& bd, bool& possibly, bool& surely, bool& undecided) {
  possibly = surely = true;
  undecided = false;
  for (int c = 1; c < q; ++c) {
    double ll, lh, ul, uh;
    rc_interval((double)mean[(size_t)c * n + g], (double)var[(size_t)c * n + g], b, bd.dm[c], bd.dv[c], ll, lh, ul, uh);
    possibly = possibly && lh >= 0.0;
    surely = surely && ll >= 0.0;
    undecided = undecided || (ll <= 0.0 && lh >= 0.0);
  }
}

// pass 1: u_lo = min over possibly-safe of ucb0_lo, u_hi = min over surely-safe of ucb0_hi
template <typename TP>
__global__ __launch_bounds__(256) void k_rc_ustar(const TP* __restrict__ mean, const TP* __restrict__ var, long long n, int q, double b,
                                                  const RcBand bd, RcScal* sc) {
  unsigned long long klo = ~0ull, khi = ~0ull;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    bool ps, ss, un;
    rc_safety(mean, var, n, g, q, b, bd, ps, ss, un);
    if (!ps) continue;
    double ll, lh, ul, uh;
    rc_interval((double)mean[g], (double)var[g], b, bd.dm[0], bd.dv[0], ll, lh, ul, uh);
    const unsigned long long a = ord_key(ul), h = ord_key(uh);
    klo = a < klo ? a : klo;
    if (ss) khi = h < khi ? h : khi;
  }
  klo = block_ext_u64<false>(klo);
  khi = block_ext_u64<false>(khi);
  if (threadIdx.x == 0) {
    atomicMin(&sc->ulo_key, klo);
    atomicMin(&sc->uhi_key, khi);
  }
}
// pass 2: vmax_lo = max over surely-in-M (surely safe, lcb0_hi <= u_lo) of max(0, var0 - dv)
template <typename TP>
__global__ __launch_bounds__(256) void k_rc_vmax(const TP* __restrict__ mean, const TP* __restrict__ var, long long n, int q, double b,
                                                 const RcBand bd, RcScal* sc) {
  const double u_lo = ord_val(sc->ulo_key);
  unsigned long long kv = 0ull;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    bool ps, ss, un;
    rc_safety(mean, var, n, g, q, b, bd, ps, ss, un);
    if (!ss) continue;
    double ll, lh, ul, uh;
    rc_interval((double)mean[g], (double)var[g], b, bd.dm[0], bd.dv[0], ll, lh, ul, uh);
    if (lh <= u_lo) {
      const unsigned long long k = ord_key(fmax(0.0, (double)var[g] - bd.dv[0]));
      kv = k > kv ? k : kv;
    }
  }
  kv = block_ext_u64<true>(kv);
  if (threadIdx.x == 0) atomicMax(&sc->vmax_key, kv);
}
// pass 3: the list of candidates the intervals cannot decide
template <typename TP>
__global__ __launch_bounds__(256) void k_rc_flag(const TP* __restrict__ mean, const TP* __restrict__ var, long long n, int q, double b,
                                                 const RcBand bd, RcScal* sc, long long* __restrict__ list, int all_possibly_safe) {
  __shared__ int wcount[4];
  __shared__ long long base;
  const bool have_hi = sc->uhi_key != ~0ull;
  const double u_lo = sc->ulo_key != ~0ull ? ord_val(sc->ulo_key) : -kInfD;
  const double u_hi = have_hi ? ord_val(sc->uhi_key) : kInfD;
  const double vmax_lo = sc->vmax_key ? ord_val(sc->vmax_key) : -1.0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long span = (long long)gridDim.x * blockDim.x;
  for (long long g0 = (long long)blockIdx.x * blockDim.x; g0 < n; g0 += span) {
    const long long g = g0 + threadIdx.x;
    bool flag = false;
    if (g < n) {
      bool ps, ss, un;
      rc_safety(mean, var, n, g, q, b, bd, ps, ss, un);
      flag = un || (all_possibly_safe && ps && q > 1);
      if (ps && all_possibly_safe != 2) {
        double ll, lh, ul, uh;
        rc_interval((double)mean[g], (double)var[g], b, bd.dm[0], bd.dv[0], ll, lh, ul, uh);
        const bool maybe_m = ll <= u_hi;
        flag = flag || ul <= u_hi;                                     // may attain u*
        flag = flag || (maybe_m && lh >= u_lo);                        // lcb_0 <= u* undecided
        flag = flag || (maybe_m && (double)var[g] + bd.dv[0] >= vmax_lo);   // may hold the largest variance over M
      }
    }
    const unsigned long long m = __ballot(flag);
    if (lane == 0) wcount[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
      const int tot = wcount[0] + wcount[1] + wcount[2] + wcount[3];
      base = tot ? (long long)atomicAdd((unsigned long long*)&sc->count, (unsigned long long)tot) : 0;
    }
    __syncthreads();
    if (flag) {
      long long off = base;
      for (int w = 0; w < wave; ++w) off += wcount[w];
      off += __popcll(m & ((1ull << lane) - 1ull));
      list[off] = g;
    }
    __syncthreads();
  }
}

template <int D>
__global__ __launch_bounds__(256) void k_rc_gather(const CandSpec cs, const long long* __restrict__ list, long long nf, double* __restrict__ pts) {
  for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nf; k += (long long)gridDim.x * blockDim.x) {
    double x[D];
    cand_coords<D>(cs, list[k], x);
    for (int a = 0; a < cs.d; ++a) pts[k * cs.d + a] = x[a];
  }
}
__global__ __launch_bounds__(256) void k_rc_widen(const float* __restrict__ m32, const float* __restrict__ v32, long long total,
                                                  double* __restrict__ m64, double* __restrict__ v64) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    m64[i] = (double)m32[i];
    v64[i] = (double)v32[i];
  }
}
__global__ __launch_bounds__(256) void k_rc_scatter(const long long* __restrict__ list, long long nf, int q, long long n,
                                                    const double* __restrict__ ms, const double* __restrict__ vs, double* __restrict__ m64,
                                                    double* __restrict__ v64, uint8_t* __restrict__ refined) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nf * q; i += (long long)gridDim.x * blockDim.x) {
    const long long k = i % nf;
    const int o = (int)(i / nf);
    const long long g = list[k];
    m64[(size_t)o * n + g] = ms[(size_t)o * nf + k];
    v64[(size_t)o * n + g] = vs[(size_t)o * nf + k];
    if (o == 0) refined[g] = 1;
  }
}

// Lipschitz keys in fp64: max over the candidates of |d MEAN_i / d x_a| for every output (the analytic gradient of
// models/SafeOpt.py:68-71 on the twin's double arrays; one thread per candidate)
template <int D>
__global__ __launch_bounds__(256) void k_rc_grad64(const ModelConst mc, const CandSpec cs, const double* __restrict__ As,
                                                   const double* __restrict__ sqA, const double* __restrict__ alpha,
                                                   const double* __restrict__ Xn, unsigned long long* __restrict__ Lmax) {
  double gm[kMaxQ];
#pragma unroll
  for (int o = 0; o < kMaxQ; ++o) gm[o] = 0.0;
  const int n = mc.n, npad = mc.npad;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < cs.n_local; g += (long long)gridDim.x * blockDim.x) {
    double x[D], xn[D];
    cand_coords<D>(cs, g, x);
#pragma unroll
    for (int a = 0; a < D; ++a) xn[a] = a < mc.d ? (x[a] - mc.X_mean[a]) / mc.X_std[a] : 0.0;
    for (int o = 0; o < mc.q; ++o) {
      double bq[D], sqb = 0.0;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        bq[a] = a < mc.d ? xn[a] * mc.vinv[o][a] : 0.0;
        sqb += bq[a] * bq[a];
      }
      const double* Ao = As + (size_t)o * npad * D;
      const double* so = sqA + (size_t)o * npad;
      const double* al = alpha + (size_t)o * npad;
      double s0 = 0.0, sa[D];
#pragma unroll
      for (int a = 0; a < D; ++a) sa[a] = 0.0;
      for (int j = 0; j < n; ++j) {
        double dot = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) dot += Ao[(size_t)j * D + a] * bq[a];
        const double w = al[j] * (mc.sf2[o] * exp(-0.5 * ((-2.0 * dot + so[j]) + sqb)));
        s0 += w;
#pragma unroll
        for (int a = 0; a < D; ++a) sa[a] += w * Xn[(size_t)j * D + a];
      }
      double gn = 0.0;
#pragma unroll
      for (int a = 0; a < D; ++a) {
        if (a < mc.d) {
          double ga = mc.Y_std[o] * (sa[a] - xn[a] * s0) * mc.inv_ell[o][a] * mc.X_rstd[a];
          ga = ga < 0 ? -ga : ga;
          gn = ga > gn ? ga : gn;
        }
      }
      gm[o] = gn > gm[o] ? gn : gm[o];
    }
  }
  for (int o = 0; o < mc.q; ++o) {
    const unsigned long long k = block_ext_u64<true>((unsigned long long)__double_as_longlong(gm[o]));   // (values >= 0: bit order)
    if (threadIdx.x == 0) atomicMax(&Lmax[o], k);
    __syncthreads();
  }
}

// guard of Expander's arg-max: gkeys[c] = max over G_c of the lower end of var_0, then every unrefined member of G_c whose
// upper end reaches it is deferred
__global__ __launch_bounds__(256) void k_rc_gmax(const uint8_t* __restrict__ G, const double* __restrict__ var0, const uint8_t* __restrict__ refined,
                                                 long long n, double dv0, unsigned long long* __restrict__ gkeys) {
  const int c = blockIdx.y;
  const uint8_t* Gc = G + (size_t)c * n;
  unsigned long long k = 0ull;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x)
    if (Gc[g]) {
      const unsigned long long kk = ord_key(refined[g] ? var0[g] : fmax(0.0, var0[g] - dv0));
      k = kk > k ? kk : k;
    }
  k = block_ext_u64<true>(k);
  if (threadIdx.x == 0 && k) atomicMax(&gkeys[c], k);
}
__global__ __launch_bounds__(256) void k_rc_gdefer(const uint8_t* __restrict__ G, const double* __restrict__ var0, const uint8_t* __restrict__ refined,
                                                   long long n, double dv0, const unsigned long long* __restrict__ gkeys,
                                                   long long* __restrict__ list, unsigned long long* __restrict__ count) {
  const int c = blockIdx.y;
  const uint8_t* Gc = G + (size_t)c * n;
  if (!gkeys[c]) return;
  const double vlo = ord_val(gkeys[c]);
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x)
    if (Gc[g] && !refined[g] && var0[g] + dv0 >= vlo) list[atomicAdd(count, 1ull)] = g;
}

// re-evaluate list[0..nf) exactly and put the values in place.  fp32 models: the fp64 twin's generic kernel, into the widened
// copy rc_mean / rc_var; guard band of an approximating fp64 posterior (`guard`): the exact evaluator of the same model
// (guard.hip: guard_exact_list), into the posterior buffers themselves.
static int rc_refine(sbo_ctx* c, const long long* list, long long nf, bool guard = false) {
  sbo_ctx* s = guard ? c : c->shadow;
  const long long n = c->cs.n_local;
  const int q = c->mc.q, d = c->cs.d;
  int rc;
  if (nf <= 0) return SBO_OK;
  DevBuf& pbuf = guard ? c->gb_pts : s->pts;
  if ((rc = ensure(pbuf, sizeof(double) * (size_t)nf * d))) return rc;
  const unsigned nbf = (unsigned)std::max<long long>(1, std::min<long long>((nf + 255) / 256, 4096));
  switch (c->mc.dpad) {
    case 2: hipLaunchKernelGGL(k_rc_gather<2>, dim3(nbf), dim3(256), 0, c->stream, c->cs, list, nf, (double*)pbuf.p); break;
    case 4: hipLaunchKernelGGL(k_rc_gather<4>, dim3(nbf), dim3(256), 0, c->stream, c->cs, list, nf, (double*)pbuf.p); break;
    default: hipLaunchKernelGGL(k_rc_gather<8>, dim3(nbf), dim3(256), 0, c->stream, c->cs, list, nf, (double*)pbuf.p); break;
  }
  if (guard) {
    if ((rc = ensure(c->gb_vals, sizeof(double) * 2 * (size_t)nf * q))) return rc;
    double* em = (double*)c->gb_vals.p;
    double* ev = em + (size_t)nf * q;
    if ((rc = guard_exact_list(c, (const double*)pbuf.p, nf, em, ev))) return rc;
    hipLaunchKernelGGL(k_rc_scatter, dim3(nbf), dim3(256), 0, c->stream, list, nf, q, n, (const double*)em, (const double*)ev,
                       (double*)c->mean.p, (double*)c->var.p, (uint8_t*)c->rc_refined.p);
    SBO_HIP(hipGetLastError());
    return SBO_OK;
  }
  memset(&s->cs, 0, sizeof(s->cs));
  s->cs.kind = 0;
  s->cs.d = d;
  s->cs.pts_dtype = SBO_F64;
  s->cs.pts = s->pts.p;
  s->cs.n_local = nf;
  s->cs.first = 0;
  s->grid_total = nf;
  s->has_cand = true;
  s->posterior_path = 0;
  if ((rc = ensure(s->mean, sizeof(double) * (size_t)nf * q))) return rc;
  if ((rc = ensure(s->var, sizeof(double) * (size_t)nf * q))) return rc;
  if ((rc = launch_posterior(s))) return rc;
  hipLaunchKernelGGL(k_rc_scatter, dim3(nbf), dim3(256), 0, c->stream, list, nf, q, n, (const double*)s->mean.p, (const double*)s->var.p,
                     (double*)c->rc_mean.p, (double*)c->rc_var.p, (uint8_t*)c->rc_refined.p);
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

// Lipschitz keys in fp64 from the double arrays of `s` (the twin of an fp32 model, or the fp64 model itself) over the
// candidates of `c`
static int rc_lipschitz64(sbo_ctx* c, const sbo_ctx* s) {
  const long long n = c->cs.n_local;
  SBO_HIP(hipMemsetAsync(c->Lmax.p, 0, sizeof(unsigned long long) * kMaxQ, c->stream));
  const unsigned nbg = (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, (long long)c->n_cu * 8));
#define SBO_GRAD(DD)                                                                                                              \
  hipLaunchKernelGGL(k_rc_grad64<DD>, dim3(nbg), dim3(256), 0, c->stream, s->mc, c->cs, (const double*)s->As.p, (const double*)s->sqA.p, \
                     (const double*)s->alpha.p, (const double*)s->Xn.p, (unsigned long long*)c->Lmax.p)
  switch (c->mc.dpad) {
    case 2: SBO_GRAD(2); break;
    case 4: SBO_GRAD(4); break;
    default: SBO_GRAD(8); break;
  }
#undef SBO_GRAD
  SBO_HIP(hipGetLastError());
  return SBO_OK;
}

// the bands of the intervals: the fp32 contract (1e-4 normalised), or the plan's guard band read back from the device
static int rc_bands(sbo_ctx* c, bool guard, RcBand& bd) {
  memset(&bd, 0, sizeof(bd));
  const int q = c->mc.q;
  if (guard) {
    GuardBand hb;
    SBO_HIP(hipMemcpyAsync(&hb, c->gb.p, sizeof(hb), hipMemcpyDeviceToHost, c->stream));
    SBO_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < q; ++i) {
      if (!(hb.dm[i] >= 0.0) || !(hb.dv[i] >= 0.0)) return fail(SBO_E_HIP, "internal: guard band is not finite");
      bd.dm[i] = hb.dm[i];
      bd.dv[i] = hb.dv[i];
    }
    return SBO_OK;
  }
  for (int i = 0; i < q; ++i) {
    const double ys = std::max(1.0, c->mc.Y_std[i]);
    bd.dm[i] = 1e-4 * ys;
    bd.dv[i] = 1e-4 * ys * ys;
  }
  return SBO_OK;
}

// TP = float: an fp32 model (posterior in fp32, the fp64 twin re-evaluates); TP = double: the guard band of an approximating fp64
// posterior (the posterior is resident and valid; its listed entries are replaced in place by exact values)
template <typename TP>
static int sweep_safeopt_recheck(sbo_ctx* c, const sbo_sweep_opts* o, sbo_safeopt_result* res) {
  constexpr bool kGuard = std::is_same<TP, double>::value;
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  int rc;
  if (!kGuard && n == 0 && !multi_rank(c)) return sweep_safeopt_t<float>(c, o, res);
  SBO_HIP(hipEventRecord(c->ev_join[0], c->stream));
  // (guard: the posterior is resident -- unless the first pass was a lean sweep, which left part of it unwritten: K1 once more, in full)
  const bool reuse = (kGuard && c->posterior_valid) || (o->posterior_ready && c->posterior_valid);
  if (!reuse && (rc = sbo_posterior_enqueue_(c))) return rc;
  c->k1_stop_attached = false;
  SBO_HIP(hipEventRecord(c->ev_join[1], c->stream));
  RcBand bd;
  if ((rc = rc_bands(c, kGuard, bd))) return rc;
  // (one pass can list a candidate once per constraint in the verdict kernels and once more per constraint in k_rc_gdefer;
  // the first list -- k_rc_flag -- holds every candidate at most once)
  const size_t list_cap = (size_t)std::max<long long>(n, 1) * (size_t)(2 * std::max(1, q - 1) + 1);
  if ((rc = ensure(c->rc_list, kRcList + sizeof(long long) * list_cap))) return rc;
  if (!kGuard) {
    if ((rc = ensure(c->rc_mean, sizeof(double) * (size_t)q * std::max<long long>(n, 1)))) return rc;
    if ((rc = ensure(c->rc_var, sizeof(double) * (size_t)q * std::max<long long>(n, 1)))) return rc;
  }
  if ((rc = ensure(c->rc_refined, (size_t)std::max<long long>(n, 1)))) return rc;
  RcScal* sc = (RcScal*)c->rc_list.p;
  unsigned long long* count2 = (unsigned long long*)((char*)c->rc_list.p + kRcCount2);
  unsigned long long* gkeys = (unsigned long long*)((char*)c->rc_list.p + kRcGKeys);
  long long* list = (long long*)((char*)c->rc_list.p + kRcList);
  SBO_HIP(hipMemsetAsync(c->rc_list.p, 0, kRcList, c->stream));
  const RcScal init{~0ull, ~0ull, 0ull, 0};
  SBO_HIP(hipMemcpyAsync(sc, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  SBO_HIP(hipMemsetAsync(c->rc_refined.p, 0, (size_t)std::max<long long>(n, 1), c->stream));
  const TP* m32 = (const TP*)c->mean.p;
  const TP* v32 = (const TP*)c->var.p;
  const unsigned nbk = (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, (long long)c->n_cu * 4));
  // candidate sets without a grid to transform decide their expanders by exhaustive pair evaluation on the ucb of every
  // safe candidate: there all possibly-safe candidates are re-evaluated
  long long plane = 1;
  for (int a = 0; a < c->cs.d - 1; ++a) plane *= c->cs.count[a];
  const bool grid_expander = c->cs.kind == 1 && c->cs.first % plane == 0 && n % plane == 0;
  hipLaunchKernelGGL(k_rc_ustar<TP>, dim3(nbk), dim3(256), 0, c->stream, m32, v32, n, q, o->b, bd, sc);
  // (ranks > 1: the interval of u* and the variance guard are global quantities -- three keys through the collectives)
  if ((rc = comm_allreduce_min_u64(c, &sc->ulo_key, 2))) return rc;
  hipLaunchKernelGGL(k_rc_vmax<TP>, dim3(nbk), dim3(256), 0, c->stream, m32, v32, n, q, o->b, bd, sc);
  if ((rc = comm_allreduce_max_u64(c, &sc->vmax_key, 1))) return rc;
  hipLaunchKernelGGL(k_rc_flag<TP>, dim3(nbk), dim3(256), 0, c->stream, m32, v32, n, q, o->b, bd, sc, list, grid_expander ? 0 : 1);
  if (!kGuard)
    hipLaunchKernelGGL(k_rc_widen, dim3(nbk), dim3(256), 0, c->stream, (const float*)c->mean.p, (const float*)c->var.p, (long long)q * n,
                       (double*)c->rc_mean.p, (double*)c->rc_var.p);
  SBO_HIP(hipGetLastError());
  unsigned char* hb = c->h_back + 5120;                             // pinned landing area of the list lengths
  SBO_HIP(hipMemcpyAsync(hb, sc, 64, hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipStreamSynchronize(c->stream));
  long long total = ((const RcScal*)hb)->count;
  if ((rc = rc_refine(c, list, total, kGuard))) return rc;
  // Lipschitz keys in fp64 (the fp32 posterior kernel left fp32-accurate ones, an approximating one its own band)
  if (q > 1 && (rc = rc_lipschitz64(c, kGuard ? c : c->shadow))) return rc;
  SBO_HIP(hipEventRecord(c->ev_join[2], c->stream));
  // the set phase in fp64 arithmetic on the refined posterior (fp32: the widened copy -- the fp32 arrays stay what
  // sbo_posterior_get returns); repeated while verdicts are deferred
  const DevBuf keep_m = c->mean, keep_v = c->var;
  const int keep_dtype = c->dtype;
  const bool keep_valid = c->posterior_valid;
  sbo_sweep_opts o2 = *o;
  o2.posterior_ready = 1;
  float set_extra = 0.0f;
  int passes = 0;
  for (;; ++passes) {
    SBO_HIP(hipMemsetAsync(count2, 0, 8 + sizeof(unsigned long long) * kMaxQ + 24, c->stream));     // deferral counter + G keys
    if (!kGuard) {
      c->mean = c->rc_mean;
      c->var = c->rc_var;
    }
    c->dtype = SBO_F64;
    c->posterior_valid = true;
    c->rc_active = q > 1;
    c->gb_slow = kGuard;                   // (guard: the band stays in force for unrefined entries; L is exact now)
    rc = sweep_safeopt_t<double>(c, &o2, res);
    c->rc_active = false;
    c->gb_slow = false;
    if (!kGuard) {
      c->rc_mean = c->mean;
      c->rc_var = c->var;
      c->mean = keep_m;
      c->var = keep_v;
    }
    c->dtype = keep_dtype;
    c->posterior_valid = keep_valid || !reuse;
    if (rc != SBO_OK || q == 1) break;
    if (passes > 0) set_extra += (float)c->prof.total_ms;
    // Expander's arg-max over every G_c must not hinge on an unrefined variance
    const uint8_t* G = (const uint8_t*)c->maskG.p;
    const double* var0 = kGuard ? (const double*)c->var.p : (const double*)c->rc_var.p;
    hipLaunchKernelGGL(k_rc_gmax, dim3(nbk, (unsigned)(q - 1)), dim3(256), 0, c->stream, G, var0, (const uint8_t*)c->rc_refined.p, n, bd.dv[0],
                       gkeys);
    if ((rc = comm_allreduce_max_u64(c, gkeys, q - 1))) return rc;        // (the expanders' arg-max is over all ranks)
    hipLaunchKernelGGL(k_rc_gdefer, dim3(nbk, (unsigned)(q - 1)), dim3(256), 0, c->stream, G, var0, (const uint8_t*)c->rc_refined.p, n,
                       bd.dv[0], (const unsigned long long*)gkeys, list, count2);
    SBO_HIP(hipGetLastError());
    SBO_HIP(hipMemcpyAsync(hb, c->rc_list.p, 64, hipMemcpyDeviceToHost, c->stream));
    SBO_HIP(hipStreamSynchronize(c->stream));
    const long long nd = (long long)*(const unsigned long long*)(hb + kRcCount2);
    // (ranks > 1: every rank runs the set phase the same number of times -- its collectives are inside --, so "nothing deferred"
    // is a global statement: the largest count over the ranks)
    unsigned long long* dcount = (unsigned long long*)((char*)c->rc_list.p + kRcCount2 + 8);   // scratch word behind the counter
    long long nd_all = nd;
    if (multi_rank(c)) {
      SBO_HIP(hipMemcpyAsync(dcount, hb + kRcCount2, 8, hipMemcpyHostToDevice, c->stream));
      if ((rc = comm_allreduce_max_u64(c, dcount, 1))) return rc;
      SBO_HIP(hipMemcpyAsync(hb + 56, dcount, 8, hipMemcpyDeviceToHost, c->stream));
      SBO_HIP(hipStreamSynchronize(c->stream));
      nd_all = (long long)*(const unsigned long long*)(hb + 56);
    }
    if (nd_all == 0) break;
    if ((size_t)nd > list_cap) return fail(SBO_E_HIP, "internal: refinement list overflow");
    // (every candidate is refined at most once, so the passes end; six without an end would be a defect, not a slow case)
    if (passes >= 6) return fail(SBO_E_UNSUPPORTED, "recheck: verdicts still deferred after 7 passes of the set phase");
    if ((rc = rc_refine(c, list, nd, kGuard))) return rc;
    total += nd;
  }
  float t01 = 0, t12 = 0, t04 = 0;
  (void)hipEventElapsedTime(&t01, c->ev_join[0], c->ev_join[1]);
  (void)hipEventElapsedTime(&t12, c->ev_join[1], c->ev_join[2]);
  (void)hipEventElapsedTime(&t04, c->ev_join[0], c->ev[4]);
  if (kGuard) {
    c->prof.guard_ms = t04;
    res->guard_rechecks = total;
    res->guard_passes = passes + 1;
  } else {
    c->prof.posterior_ms = t01;
    c->prof.recheck_ms = t12;
    c->prof.total_ms = t04;
    c->prof.fp64_rechecks = total;
    c->prof.posterior_launches = reuse ? 0 : 1;
    const double nn = c->mc.n, dd = c->mc.d;
    c->prof.posterior_flops = reuse ? 0.0 : q * (nn * nn + (2 * dd + 10) * nn) * (double)n;
  }
  (void)set_extra;
  return rc;
}

// ---- GoOSE and trust-region sweeps of fp32 models (r03) -------------------------------------------------------------------------
// Coarser than the SafeOpt scheme above, and sufficient: with constraints, EVERY possibly-safe candidate is re-evaluated in
// fp64 (the sources of the optimistic sets are the expanders, whose radii ucb_c / L must be exact, and arg-min lcb_0 over S_t
// wants every safe candidate's bound) together with the candidates whose S / U membership the fp32 bounds cannot decide; the
// set phase then runs in fp64 arithmetic on exact values wherever a value enters a decision -- except the target, arg-min
// lcb_0 over the optimistic sets O_c (unsafe candidates, not re-evaluated): its contenders are found from the intervals
// afterwards, re-evaluated, and the set phase runs once more if there were any.  Without constraints (q = 1) only the
// contenders of the one arg-min (over all candidates, or over the trust region's ball) are re-evaluated.
// min over the (unrefined: widened) members of `mask` of the upper end of lcb_0 -> key;  refined entries count exactly
__global__ __launch_bounds__(256) void k_rc_lmin(const double* __restrict__ m0, const double* __restrict__ v0, const uint8_t* __restrict__ refined,
                                                 const uint8_t* __restrict__ mask /* nullptr: all */, int nmask, long long n, double b, double dm,
                                                 double dv, unsigned long long* key) {
  unsigned long long k = ~0ull;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    bool in = mask == nullptr;
    for (int t = 0; t < nmask && !in; ++t) in = mask[(size_t)t * n + g] != 0;
    if (!in) continue;
    const double md = m0[g], vd = v0[g];
    const double hi = refined[g] ? md - b * sqrt(vd) : (md + dm) - b * sqrt(fmax(0.0, vd - dv));
    const unsigned long long kk = ord_key(hi);
    k = kk < k ? kk : k;
  }
  k = block_ext_u64<false>(k);
  if (threadIdx.x == 0) atomicMin(key, k);
}
// ... and the unrefined members whose lower end reaches it: they may hold the minimum
__global__ __launch_bounds__(256) void k_rc_lflag(const double* __restrict__ m0, const double* __restrict__ v0, const uint8_t* __restrict__ refined,
                                                  const uint8_t* __restrict__ mask, int nmask, long long n, double b, double dm, double dv,
                                                  const unsigned long long* key, long long* __restrict__ list, unsigned long long* count) {
  if (*key == ~0ull) return;
  const double best_hi = ord_val(*key);
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    if (refined[g]) continue;
    bool in = mask == nullptr;
    for (int t = 0; t < nmask && !in; ++t) in = mask[(size_t)t * n + g] != 0;
    if (!in) continue;
    const double lo = (m0[g] - dm) - b * sqrt(v0[g] + dv);
    if (lo <= best_hi) list[atomicAdd(count, 1ull)] = g;
  }
}
// trust region without constraints: the ball as a byte mask (geometry: exact)
template <int D>
__global__ void k_rc_ball(const CandSpec cs, long long n, const double* __restrict__ x0, double r, uint8_t* __restrict__ out) {
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    double x[D];
    cand_coords<D>(cs, g, x);
    double ss = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a)
      if (a < cs.d) { const double df = x[a] - x0[a]; ss = (a == 0) ? df * df : ss + df * df; }
    out[g] = sqrt(ss) <= r;
  }
}

struct RcView {                        // the context looks at the refined fp64 posterior while the scope lives: the widened copy
  sbo_ctx* c;                          // of an fp32 posterior, or (guard) the fp64 posterior itself, refined in place
  bool guard;
  DevBuf keep_m, keep_v;
  int keep_dtype;
  bool keep_valid;
  RcView(sbo_ctx* c_, bool guard_) : c(c_), guard(guard_), keep_m(c_->mean), keep_v(c_->var), keep_dtype(c_->dtype), keep_valid(c_->posterior_valid) {
    if (!guard) {
      c->mean = c->rc_mean;
      c->var = c->rc_var;
    }
    c->dtype = SBO_F64;
    c->posterior_valid = true;
    c->gb_off = true;                  // (every value that enters a decision is exact by now: no band)
  }
  ~RcView() {
    if (!guard) {
      c->rc_mean = c->mean;
      c->rc_var = c->var;
      c->mean = keep_m;
      c->var = keep_v;
    }
    c->dtype = keep_dtype;
    c->posterior_valid = keep_valid;
    c->gb_off = false;
  }
};

// front end shared by the GoOSE / trust-region rechecks: posterior, (fp32) widened copy, first refinement list (q > 1: every
// possibly-safe or undecided candidate), fp64 Lipschitz keys.  Leaves the list buffers ready for further rounds.
template <typename TP>
static int rc_front_all(sbo_ctx* c, const sbo_sweep_opts* o, RcBand& bd, long long* total) {
  constexpr bool kGuard = std::is_same<TP, double>::value;
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  int rc;
  const bool reuse = kGuard || (o->posterior_ready && c->posterior_valid);
  if (!reuse && (rc = sbo_posterior_enqueue_(c))) return rc;
  c->k1_stop_attached = false;
  if ((rc = rc_bands(c, kGuard, bd))) return rc;
  if ((rc = ensure(c->rc_list, kRcList + sizeof(long long) * (size_t)std::max<long long>(n, 1) * 2))) return rc;
  if (!kGuard) {
    if ((rc = ensure(c->rc_mean, sizeof(double) * (size_t)q * std::max<long long>(n, 1)))) return rc;
    if ((rc = ensure(c->rc_var, sizeof(double) * (size_t)q * std::max<long long>(n, 1)))) return rc;
  }
  if ((rc = ensure(c->rc_refined, (size_t)std::max<long long>(n, 1)))) return rc;
  RcScal* sc = (RcScal*)c->rc_list.p;
  long long* list = (long long*)((char*)c->rc_list.p + kRcList);
  SBO_HIP(hipMemsetAsync(c->rc_list.p, 0, kRcList, c->stream));
  const RcScal init{~0ull, ~0ull, 0ull, 0};
  SBO_HIP(hipMemcpyAsync(sc, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  SBO_HIP(hipMemsetAsync(c->rc_refined.p, 0, (size_t)std::max<long long>(n, 1), c->stream));
  const TP* m32 = (const TP*)c->mean.p;
  const TP* v32 = (const TP*)c->var.p;
  const unsigned nbk = (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, (long long)c->n_cu * 4));
  *total = 0;
  if (q > 1) {
    // (ulo / uhi / vmax stay at their initial values: only the "undecided or possibly safe" rule of the flag pass fires)
    hipLaunchKernelGGL(k_rc_flag<TP>, dim3(nbk), dim3(256), 0, c->stream, m32, v32, n, q, o->b, bd, sc, list, 2);
  }
  if (!kGuard)
    hipLaunchKernelGGL(k_rc_widen, dim3(nbk), dim3(256), 0, c->stream, (const float*)c->mean.p, (const float*)c->var.p, (long long)q * n,
                       (double*)c->rc_mean.p, (double*)c->rc_var.p);
  SBO_HIP(hipGetLastError());
  unsigned char* hb = c->h_back + 5120;
  SBO_HIP(hipMemcpyAsync(hb, sc, 64, hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipStreamSynchronize(c->stream));
  *total = ((const RcScal*)hb)->count;
  if ((rc = rc_refine(c, list, *total, kGuard))) return rc;
  if (q > 1 && (rc = rc_lipschitz64(c, kGuard ? c : c->shadow))) return rc;
  return SBO_OK;
}

// contenders of arg-min lcb_0 over `mask` (nmask stacked byte masks, nullptr: every candidate): re-evaluated exactly; returns
// how many there were (0: the arg-min already rests on exact values)
static int rc_argmin_contenders(sbo_ctx* c, const sbo_sweep_opts* o, const RcBand& bd, const uint8_t* mask, int nmask, long long* found,
                                bool guard) {
  const long long n = c->cs.n_local;
  int rc;
  unsigned long long* key = (unsigned long long*)((char*)c->rc_list.p + kRcCount2 + 8);
  unsigned long long* count2 = (unsigned long long*)((char*)c->rc_list.p + kRcCount2);
  long long* list = (long long*)((char*)c->rc_list.p + kRcList);
  const unsigned long long init[2] = {0ull, ~0ull};
  SBO_HIP(hipMemcpyAsync(count2, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  const unsigned nbk = (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, (long long)c->n_cu * 4));
  const double* m0 = guard ? (const double*)c->mean.p : (const double*)c->rc_mean.p;
  const double* v0 = guard ? (const double*)c->var.p : (const double*)c->rc_var.p;
  hipLaunchKernelGGL(k_rc_lmin, dim3(nbk), dim3(256), 0, c->stream, m0, v0, (const uint8_t*)c->rc_refined.p, mask, nmask, n, o->b, bd.dm[0], bd.dv[0], key);
  if ((rc = comm_allreduce_min_u64(c, key, 1))) return rc;             // (ranks > 1: the arg-min is over all ranks)
  hipLaunchKernelGGL(k_rc_lflag, dim3(nbk), dim3(256), 0, c->stream, m0, v0, (const uint8_t*)c->rc_refined.p, mask, nmask, n, o->b, bd.dm[0], bd.dv[0],
                     (const unsigned long long*)key, list, count2);
  SBO_HIP(hipGetLastError());
  unsigned char* hb = c->h_back + 5120;
  SBO_HIP(hipMemcpyAsync(hb, c->rc_list.p, 64, hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipStreamSynchronize(c->stream));
  const long long mine = (long long)*(const unsigned long long*)(hb + kRcCount2);
  *found = mine;
  if (multi_rank(c)) {                   // every rank goes round the same number of times: the largest count decides
    unsigned long long* dcount = (unsigned long long*)((char*)c->rc_list.p + kRcCount2 + 16);
    SBO_HIP(hipMemcpyAsync(dcount, hb + kRcCount2, 8, hipMemcpyHostToDevice, c->stream));
    if ((rc = comm_allreduce_max_u64(c, dcount, 1))) return rc;
    SBO_HIP(hipMemcpyAsync(hb + 56, dcount, 8, hipMemcpyDeviceToHost, c->stream));
    SBO_HIP(hipStreamSynchronize(c->stream));
    *found = (long long)*(const unsigned long long*)(hb + 56);
  }
  if ((rc = rc_refine(c, list, mine, guard))) return rc;
  return SBO_OK;
}

template <typename TP>
static int sweep_goose_recheck(sbo_ctx* c, const sbo_sweep_opts* o, sbo_goose_result* res) {
  constexpr bool kGuard = std::is_same<TP, double>::value;
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  if (n == 0) return kGuard ? sweep_goose_t<double>(c, o, res) : sweep_goose_t<float>(c, o, res);
  int rc;
  SBO_HIP(hipEventRecord(c->ev_join[0], c->stream));
  const bool reuse = kGuard || (o->posterior_ready && c->posterior_valid);
  RcBand bd;
  long long total = 0, more = 0;
  if ((rc = rc_front_all<TP>(c, o, bd, &total))) return rc;
  SBO_HIP(hipEventRecord(c->ev_join[1], c->stream));
  sbo_sweep_opts o2 = *o;
  o2.posterior_ready = 1;
  o2.want_masks = 1;
  int passes = 0;
  if (q == 1) {
    // arg-min lcb_0 over every candidate: its contenders, then one fp64 set phase
    if ((rc = rc_argmin_contenders(c, o, bd, nullptr, 0, &more, kGuard))) return rc;
    total += more;
    RcView view(c, kGuard);
    rc = sweep_goose_t<double>(c, &o2, res);
    passes = 1;
  } else {
    for (int pass = 0; pass < 3; ++pass) {
      {
        RcView view(c, kGuard);
        rc = sweep_goose_t<double>(c, &o2, res);
      }
      ++passes;
      if (rc != SBO_OK) break;
      // the target: arg-min lcb_0 over the union of the optimistic sets (members are unsafe candidates, so far unrefined values)
      if ((rc = rc_argmin_contenders(c, o, bd, (const uint8_t*)c->maskO.p, q - 1, &more, kGuard))) return rc;
      total += more;
      if (more == 0) break;
    }
  }
  float t01 = 0;
  (void)hipEventElapsedTime(&t01, c->ev_join[0], c->ev_join[1]);
  if (kGuard) {
    res->guard_rechecks = total;
    res->guard_passes = passes;
    c->prof.guard_ms = t01;
  } else {
    c->posterior_valid = c->posterior_valid || !reuse;
    c->prof.fp64_rechecks = total;
    c->prof.posterior_launches = reuse ? 0 : 1;
    c->prof.recheck_ms = t01;
  }
  return rc;
}

template <typename TP>
static int sweep_tr_recheck(sbo_ctx* c, const sbo_sweep_opts* o, const double* x0, double r, sbo_tr_result* res) {
  constexpr bool kGuard = std::is_same<TP, double>::value;
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  if (n == 0) return kGuard ? sweep_tr_t<double>(c, o, x0, r, res) : sweep_tr_t<float>(c, o, x0, r, res);
  int rc;
  const bool reuse = kGuard || (o->posterior_ready && c->posterior_valid);
  RcBand bd;
  long long total = 0, more = 0;
  if ((rc = rc_front_all<TP>(c, o, bd, &total))) return rc;
  sbo_sweep_opts o2 = *o;
  o2.posterior_ready = 1;
  if (q == 1) {
    // arg-min lcb_0 over the ball: the ball as a mask (exact geometry), then the contenders inside it
    if ((rc = ensure(c->maskM, (size_t)n))) return rc;
    double* dev_x0 = (double*)c->scal.p + 256;
    SBO_HIP(hipMemcpyAsync(dev_x0, x0, sizeof(double) * c->cs.d, hipMemcpyHostToDevice, c->stream));
    const unsigned nbk = (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, (long long)c->n_cu * 4));
    switch (c->mc.dpad) {
      case 2: hipLaunchKernelGGL((k_rc_ball<2>), dim3(nbk), dim3(256), 0, c->stream, c->cs, n, (const double*)dev_x0, r, (uint8_t*)c->maskM.p); break;
      case 4: hipLaunchKernelGGL((k_rc_ball<4>), dim3(nbk), dim3(256), 0, c->stream, c->cs, n, (const double*)dev_x0, r, (uint8_t*)c->maskM.p); break;
      default: hipLaunchKernelGGL((k_rc_ball<8>), dim3(nbk), dim3(256), 0, c->stream, c->cs, n, (const double*)dev_x0, r, (uint8_t*)c->maskM.p); break;
    }
    if ((rc = rc_argmin_contenders(c, o, bd, (const uint8_t*)c->maskM.p, 1, &more, kGuard))) return rc;
    total += more;
  }
  {
    RcView view(c, kGuard);
    rc = sweep_tr_t<double>(c, &o2, x0, r, res);
  }
  if (kGuard) {
    res->guard_rechecks = total;
    res->guard_passes = 1;
  } else {
    c->posterior_valid = c->posterior_valid || !reuse;
    c->prof.fp64_rechecks = total;
  }
  return rc;
}
