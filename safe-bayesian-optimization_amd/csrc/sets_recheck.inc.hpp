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
// S, U, u*, M and the minimiser are then those of an fp64 sweep; the expander sets still see the fp32 Lipschitz constant
// and the fp32 ucb of unlisted candidates (their verdicts are fp32-accurate, not bit-exact).
#pragma once

struct RcBand {
  double dm[kMaxQ], dv[kMaxQ];
};
struct RcScal {                         // head of rc_list
  unsigned long long ulo_key, uhi_key, vmax_key;
  long long count;
};

__device__ __forceinline__ void rc_interval(float m, float v, double b, double dm, double dv, double& lcb_lo, double& lcb_hi, double& ucb_lo,
                                            double& ucb_hi) {
  const double md = (double)m, vd = (double)v;
  const double s_hi = b * sqrt(vd + dv), s_lo = b * sqrt(fmax(0.0, vd - dv));
  lcb_lo = (md - dm) - s_hi;
  lcb_hi = (md + dm) - s_lo;
  ucb_lo = (md - dm) + s_lo;
  ucb_hi = (md + dm) + s_hi;
}

// possibly / surely safe from the constraints' intervals; `undecided`: some constraint's lcb interval contains zero
__device__ __forceinline__ void rc_safety(const float* __restrict__ mean, const float* __restrict__ var, long long n, long long g, int q,
                                          double b, const RcBand& bd, bool& possibly, bool& surely, bool& undecided) {
  possibly = surely = true;
  undecided = false;
  for (int c = 1; c < q; ++c) {
    double ll, lh, ul, uh;
    rc_interval(mean[(size_t)c * n + g], var[(size_t)c * n + g], b, bd.dm[c], bd.dv[c], ll, lh, ul, uh);
    possibly = possibly && lh >= 0.0;
    surely = surely && ll >= 0.0;
    undecided = undecided || (ll <= 0.0 && lh >= 0.0);
  }
}

// pass 1: u_lo = min over possibly-safe of ucb0_lo, u_hi = min over surely-safe of ucb0_hi
__global__ __launch_bounds__(256) void k_rc_ustar(const float* __restrict__ mean, const float* __restrict__ var, long long n, int q, double b,
                                                  const RcBand bd, RcScal* sc) {
  unsigned long long klo = ~0ull, khi = ~0ull;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    bool ps, ss, un;
    rc_safety(mean, var, n, g, q, b, bd, ps, ss, un);
    if (!ps) continue;
    double ll, lh, ul, uh;
    rc_interval(mean[g], var[g], b, bd.dm[0], bd.dv[0], ll, lh, ul, uh);
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
__global__ __launch_bounds__(256) void k_rc_vmax(const float* __restrict__ mean, const float* __restrict__ var, long long n, int q, double b,
                                                 const RcBand bd, RcScal* sc) {
  const double u_lo = ord_val(sc->ulo_key);
  unsigned long long kv = 0ull;
  for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (long long)gridDim.x * blockDim.x) {
    bool ps, ss, un;
    rc_safety(mean, var, n, g, q, b, bd, ps, ss, un);
    if (!ss) continue;
    double ll, lh, ul, uh;
    rc_interval(mean[g], var[g], b, bd.dm[0], bd.dv[0], ll, lh, ul, uh);
    if (lh <= u_lo) {
      const unsigned long long k = ord_key(fmax(0.0, (double)var[g] - bd.dv[0]));
      kv = k > kv ? k : kv;
    }
  }
  kv = block_ext_u64<true>(kv);
  if (threadIdx.x == 0) atomicMax(&sc->vmax_key, kv);
}
// pass 3: the list of candidates the intervals cannot decide
__global__ __launch_bounds__(256) void k_rc_flag(const float* __restrict__ mean, const float* __restrict__ var, long long n, int q, double b,
                                                 const RcBand bd, RcScal* sc, long long* __restrict__ list) {
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
      flag = un;
      if (ps) {
        double ll, lh, ul, uh;
        rc_interval(mean[g], var[g], b, bd.dm[0], bd.dv[0], ll, lh, ul, uh);
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
                                                    double* __restrict__ v64) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nf * q; i += (long long)gridDim.x * blockDim.x) {
    const long long k = i % nf;
    const int o = (int)(i / nf);
    const long long g = list[k];
    m64[(size_t)o * n + g] = ms[(size_t)o * nf + k];
    v64[(size_t)o * n + g] = vs[(size_t)o * nf + k];
  }
}

static int sweep_safeopt_f32_recheck(sbo_ctx* c, const sbo_sweep_opts* o, sbo_safeopt_result* res) {
  sbo_ctx* s = c->shadow;
  const long long n = c->cs.n_local;
  const int q = c->mc.q;
  int rc;
  if (n == 0) return sweep_safeopt_t<float>(c, o, res);
  SBO_HIP(hipEventRecord(c->ev_join[0], c->stream));
  const bool reuse = o->posterior_ready && c->posterior_valid;
  if (!reuse && (rc = sbo_posterior_enqueue_(c))) return rc;
  if (c->k1_stop_attached) c->k1_stop_attached = false;            // (K1b never runs in fp32; kept for symmetry)
  SBO_HIP(hipEventRecord(c->ev_join[1], c->stream));
  // the fp32 contract as absolute bands per output
  RcBand bd;
  memset(&bd, 0, sizeof(bd));
  for (int i = 0; i < q; ++i) {
    const double ys = std::max(1.0, c->mc.Y_std[i]);
    bd.dm[i] = 1e-4 * ys;
    bd.dv[i] = 1e-4 * ys * ys;
  }
  if ((rc = ensure(c->rc_list, sizeof(RcScal) + sizeof(long long) * (size_t)n))) return rc;
  if ((rc = ensure(c->rc_mean, sizeof(double) * (size_t)q * n))) return rc;
  if ((rc = ensure(c->rc_var, sizeof(double) * (size_t)q * n))) return rc;
  RcScal* sc = (RcScal*)c->rc_list.p;
  long long* list = (long long*)((char*)c->rc_list.p + sizeof(RcScal));
  const RcScal init{~0ull, ~0ull, 0ull, 0};
  SBO_HIP(hipMemcpyAsync(sc, &init, sizeof(init), hipMemcpyHostToDevice, c->stream));
  const float* m32 = (const float*)c->mean.p;
  const float* v32 = (const float*)c->var.p;
  const unsigned nbk = (unsigned)std::max<long long>(1, std::min<long long>((n + 255) / 256, (long long)c->n_cu * 4));
  hipLaunchKernelGGL(k_rc_ustar, dim3(nbk), dim3(256), 0, c->stream, m32, v32, n, q, o->b, bd, sc);
  hipLaunchKernelGGL(k_rc_vmax, dim3(nbk), dim3(256), 0, c->stream, m32, v32, n, q, o->b, bd, sc);
  hipLaunchKernelGGL(k_rc_flag, dim3(nbk), dim3(256), 0, c->stream, m32, v32, n, q, o->b, bd, sc, list);
  hipLaunchKernelGGL(k_rc_widen, dim3(nbk), dim3(256), 0, c->stream, m32, v32, (long long)q * n, (double*)c->rc_mean.p, (double*)c->rc_var.p);
  SBO_HIP(hipGetLastError());
  RcScal* hsc = (RcScal*)(c->h_back + 5120);
  SBO_HIP(hipMemcpyAsync(hsc, sc, sizeof(RcScal), hipMemcpyDeviceToHost, c->stream));
  SBO_HIP(hipStreamSynchronize(c->stream));
  const long long nf = hsc->count;
  if (nf > 0) {
    // the listed candidates as an explicit fp64 list of the twin, its generic posterior kernel, and the values back in place
    const int d = c->cs.d;
    if ((rc = ensure(s->pts, sizeof(double) * (size_t)nf * d))) return rc;
    const unsigned nbf = (unsigned)std::max<long long>(1, std::min<long long>((nf + 255) / 256, 4096));
    switch (c->mc.dpad) {
      case 2: hipLaunchKernelGGL(k_rc_gather<2>, dim3(nbf), dim3(256), 0, c->stream, c->cs, (const long long*)list, nf, (double*)s->pts.p); break;
      case 4: hipLaunchKernelGGL(k_rc_gather<4>, dim3(nbf), dim3(256), 0, c->stream, c->cs, (const long long*)list, nf, (double*)s->pts.p); break;
      default: hipLaunchKernelGGL(k_rc_gather<8>, dim3(nbf), dim3(256), 0, c->stream, c->cs, (const long long*)list, nf, (double*)s->pts.p); break;
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
    hipLaunchKernelGGL(k_rc_scatter, dim3(nbf), dim3(256), 0, c->stream, (const long long*)list, nf, q, n, (const double*)s->mean.p,
                       (const double*)s->var.p, (double*)c->rc_mean.p, (double*)c->rc_var.p);
    SBO_HIP(hipGetLastError());
  }
  SBO_HIP(hipEventRecord(c->ev_join[2], c->stream));
  // the set phase in fp64 arithmetic on the widened + refined posterior (the fp32 arrays stay what sbo_posterior_get returns)
  const DevBuf keep_m = c->mean, keep_v = c->var;
  const int keep_dtype = c->dtype;
  const bool keep_valid = c->posterior_valid;
  c->mean = c->rc_mean;
  c->var = c->rc_var;
  c->dtype = SBO_F64;
  c->posterior_valid = true;
  sbo_sweep_opts o2 = *o;
  o2.posterior_ready = 1;
  rc = sweep_safeopt_t<double>(c, &o2, res);
  c->rc_mean = c->mean;          // (ensure() inside cannot have touched them, but keep the DevBufs in step)
  c->rc_var = c->var;
  c->mean = keep_m;
  c->var = keep_v;
  c->dtype = keep_dtype;
  c->posterior_valid = keep_valid || !reuse;
  float t01 = 0, t12 = 0, t04 = 0;
  (void)hipEventElapsedTime(&t01, c->ev_join[0], c->ev_join[1]);
  (void)hipEventElapsedTime(&t12, c->ev_join[1], c->ev_join[2]);
  (void)hipEventElapsedTime(&t04, c->ev_join[0], c->ev[4]);
  c->prof.posterior_ms = t01;
  c->prof.recheck_ms = t12;
  c->prof.total_ms = t04;
  c->prof.fp64_rechecks = nf;
  c->prof.posterior_launches = reuse ? 0 : 1;
  const double nn = c->mc.n, dd = c->mc.d;
  c->prof.posterior_flops = reuse ? 0.0 : q * (nn * nn + (2 * dd + 10) * nn) * (double)n;
  return rc;
}
