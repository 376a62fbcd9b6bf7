// api.hip -- C-ABI entry points of libsafebo.so: context, model upload, candidates, posterior, bounds.
// (sweeps live in sets.hip, collectives in comm.hip).  See include/safebo.h for the contract.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <chrono>
#include "internal.hpp"
#include "device_common.hpp"

namespace sbo {

static thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
int hip_fail(hipError_t e, const char* what) {
  g_err = std::string("HIP: ") + hipGetErrorString(e) + " in " + what;
  return SBO_E_HIP;
}
int ensure(DevBuf& b, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (b.bytes >= bytes) return SBO_OK;
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
  hipError_t e = hipMalloc(&b.p, bytes);
  if (e != hipSuccess) {
    g_err = std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? SBO_E_NOMEM : SBO_E_HIP;
  }
  b.bytes = bytes;
  return SBO_OK;
}
// The runtime's blocking wait parks the thread and wakes it through an interrupt (~15-25 us after the stream drained); a
// sweep is 0.3 ms, so the host polls for up to 5 ms first and only then sleeps.
hipError_t stream_wait(const sbo_ctx* c, hipStream_t st) {
  {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(st);
      if (e != hipErrorNotReady) return e;
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
    }
  }
  return hipStreamSynchronize(st);
}

// after a failed call: nothing of it may still be running on the side streams when the caller comes back (a later call
// synchronises the main stream only and would race with orphaned kernels on the shared scratch)
void drain_streams(sbo_ctx* c) {
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  if (c->stream3) (void)hipStreamSynchronize(c->stream3);
  if (c->stream4) (void)hipStreamSynchronize(c->stream4);
}

// The reverse Cholesky factor of a caller's invK may still be in the making (sbo_ctx::factor_pending): every consumer of
// Fpk / Fplain waits here first.  The positive-definiteness verdict arrives with it.
int factor_sync(sbo_ctx* c) {
  if (c->factor_todo) {                   // (not even enqueued: a model change that failed half-way)
    const int rc = model_factor_enqueue(c);
    if (rc) return rc;
  }
  if (!c->factor_pending) return SBO_OK;
  SBO_HIP(hipEventSynchronize(c->ev_factor));
  c->factor_pending = false;
  const int* hbad = (const int*)(c->h_back + 4864);
  for (int o = 0; o < c->mc.q; ++o)
    if (hbad[o]) {
      c->has_model = false;
      return fail(SBO_E_INVALID, "invK is not positive definite (its factor was needed by this call)");
    }
  return SBO_OK;
}

void release(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
}

int launch_soa_to_aos(sbo_ctx* c, const void* soa, void* aos);

}  // namespace sbo

using namespace sbo;

extern "C" {

int sbo_version(void) { return SBO_ABI_VERSION; }
const char* sbo_last_error(void) { return g_err.c_str(); }

int sbo_device_count(int* count) {
  if (!count) return fail(SBO_E_INVALID, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return hip_fail(e, "hipGetDeviceCount"); }
  *count = n;
  return SBO_OK;
}

int sbo_init(int device_id, sbo_ctx** out) {
  if (!out) return fail(SBO_E_INVALID, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(SBO_E_HIP, "no HIP device available: libsafebo has no CPU fallback");
  if (device_id < 0 || device_id >= n) return fail(SBO_E_INVALID, "device_id out of range");
  SBO_HIP(hipSetDevice(device_id));
  sbo_ctx* c = new sbo_ctx();
  c->device = device_id;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->n_cu = prop.multiProcessorCount;
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
  if (e == hipSuccess) {
    // the chain stream of an overlapped sweep carries many short kernels next to one long GEMM launch: highest priority,
    // so that its workgroups are placed ahead of the GEMM's when both queues have work
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
    e = hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, greatest);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->stream_audit, hipStreamNonBlocking, least);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream4, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_factor, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_w, hipEventDisableTiming);
    for (auto& ev : c->ev_grad)
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (auto& ev : c->ev_col)
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    for (auto& ev : c->ev_audit)
      if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  }
  if (e != hipSuccess) { delete c; return hip_fail(e, "hipStreamCreate"); }
  for (auto& ev : c->ev) {
    e = hipEventCreate(&ev);
    if (e != hipSuccess) { delete c; return hip_fail(e, "hipEventCreate"); }
  }
  for (auto& ev : c->ev_join) {
    e = hipEventCreate(&ev);
    if (e != hipSuccess) { delete c; return hip_fail(e, "hipEventCreate"); }
  }
  if (hipHostMalloc((void**)&c->h_c1, sizeof(unsigned long long) * (1 + 2 * kMaxQ), hipHostMallocDefault) != hipSuccess) {
    delete c;
    return fail(SBO_E_HIP, "hipHostMalloc");
  }
  if (hipHostMalloc((void**)&c->h_back, 8192, hipHostMallocDefault) != hipSuccess) {
    delete c;
    return fail(SBO_E_HIP, "hipHostMalloc");
  }
  // the sweep's scalar block and, 3 KB further, the Lipschitz keys: one allocation, so one read-back covers both
  // (layout of the 4 KB: SweepScalars at 0, the explore target / trust-region centre at 2048, the keys at 3072, a
  // collective's scratch word at 4000)
  int rc = ensure(c->scal, 4096);
  if (rc) { delete c; return rc; }
  c->Lmax.p = (char*)c->scal.p + 3072;   // (a view: not in the release list)
  c->Lmax.bytes = 512;
  *out = c;
  return SBO_OK;
}

int sbo_comm_destroy_internal(sbo_ctx* ctx);

// fp64 twin of an fp32 model: a context of its own (model arrays, candidate list, posterior buffers) on the owner's streams
static int shadow_ensure(sbo_ctx* c) {
  if (c->shadow) return SBO_OK;
  sbo_ctx* s = new sbo_ctx();
  s->is_shadow = true;
  s->device = c->device;
  s->n_cu = c->n_cu;
  s->stream = c->stream;
  s->stream2 = c->stream2;
  s->stream3 = c->stream3;
  s->stream4 = c->stream4;
  s->ev_factor = c->ev_factor;
  s->ev_w = c->ev_w;
  s->chol_async = 0;
  for (int i = 0; i < 8; ++i) s->ev[i] = c->ev[i];
  s->h_back = c->h_back;
  s->fp64_recheck = 0;
  s->bilinear = 0;
  int rc = ensure(s->scal, 4096);
  if (rc) { delete s; return rc; }
  s->Lmax.p = (char*)s->scal.p + 3072;
  s->Lmax.bytes = 512;
  c->shadow = s;
  return SBO_OK;
}

int sbo_shutdown(sbo_ctx* c) {
  if (!c) return SBO_OK;
  (void)hipSetDevice(c->device);
  guard_audit_harvest(c, true);
  (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  if (c->stream3) (void)hipStreamSynchronize(c->stream3);
  if (c->stream4) (void)hipStreamSynchronize(c->stream4);
  if (c->shadow) {
    sbo_ctx* s = c->shadow;
    for (DevBuf* b : {&s->Fpk, &s->As, &s->sqA, &s->alpha, &s->Xn, &s->pts, &s->mean, &s->var, &s->scal, &s->mwork, &s->Fplain, &s->alpha64})
      release(*b);
    if (s->h_stage) (void)hipHostFree(s->h_stage);
    delete s;
    c->shadow = nullptr;
  }
  sbo_comm_destroy_internal(c);
  for (DevBuf* b : {&c->Fpk, &c->As, &c->sqA, &c->alpha, &c->Xn, &c->pts, &c->mean, &c->var, &c->maskS,
                    &c->maskU, &c->maskM, &c->maskG, &c->maskO, &c->dist2, &c->dist2b, &c->coarse, &c->fitbuf, &c->fitwork, &c->scal, &c->partial, &c->amb, &c->runmeta, &c->bl_P0f, &c->bl_P1A, &c->bl_T4f, &c->bl_BtA, &c->bl_SBf, &c->bl_VA, &c->bl_small, &c->bl_work, &c->bl_cheb, &c->bl_basis, &c->mwork, &c->rc_mean, &c->rc_var, &c->rc_list, &c->rc_refined, &c->Fplain, &c->alpha64, &c->blockmin, &c->blockmax, &c->cpart, &c->invk_img, &c->bl_lpart, &c->bl_grad, &c->scanlist, &c->gw, &c->Wfull, &c->Uwin, &c->ubits, &c->lane1.dist2, &c->lane1.dist2b, &c->lane1.coarse, &c->lane1.blockmin, &c->lane1.blockmax, &c->lane1.scanlist, &c->lane1.amb, &c->lane1.gw, &c->lane1.runmeta, &c->lane1.scal, &c->gather, &c->xch, &c->shard_first, &c->E0f, &c->Er, &c->AXg, &c->tn_pts, &c->tn_vals, &c->tn_work, &c->tn_W0t, &c->tn_W1t, &c->tn_probe, &c->tn_scr, &c->tn_tail, &c->fuseS, &c->fuseU, &c->tn_gather, &c->bi_params, &c->gb, &c->gb_pts, &c->gb_vals, &c->gb_probe, &c->gb_part, &c->list_scr, &c->cbS, &c->cbU, &c->cbM, &c->cbG, &c->cbUsum, &c->col_img, &c->col_bmin, &c->col_fin, &c->col_slots, &c->col_cimg, &c->col_cbmin, &c->audit_pts, &c->audit_val, &c->audit_part, &c->audit_cnt})
    release(*b);
  for (auto& b : c->tn_W) release(b);
  for (auto& ev : c->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : c->ev_join)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : c->ev_col)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : c->ev_grad)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& ev : c->ev_audit)
    if (ev) (void)hipEventDestroy(ev);
  if (c->h_c1) (void)hipHostFree(c->h_c1);
  if (c->h_back) (void)hipHostFree(c->h_back);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->ev_factor) (void)hipEventDestroy(c->ev_factor);
  if (c->ev_w) (void)hipEventDestroy(c->ev_w);
  if (c->bi.exec) (void)hipGraphExecDestroy((hipGraphExec_t)c->bi.exec);
  if (c->h_bi_params) (void)hipHostFree(c->h_bi_params);
  if (c->ev_bi_params) (void)hipEventDestroy((hipEvent_t)c->ev_bi_params);
  if (c->stream4) (void)hipStreamDestroy(c->stream4);
  if (c->stream_audit) (void)hipStreamDestroy(c->stream_audit);
  if (c->stream3) (void)hipStreamDestroy(c->stream3);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return SBO_OK;
}

int sbo_synchronize(sbo_ctx* c) {
  if (!c) return fail(SBO_E_INVALID, "ctx is NULL");
  SBO_HIP(hipStreamSynchronize(c->stream));
  // (everything the library may still have in flight for this context: a deferred factorisation, the bases of a model)
  if (c->stream2) SBO_HIP(hipStreamSynchronize(c->stream2));
  if (c->stream3) SBO_HIP(hipStreamSynchronize(c->stream3));
  if (c->stream4) SBO_HIP(hipStreamSynchronize(c->stream4));
  if (c->stream_audit) SBO_HIP(hipStreamSynchronize(c->stream_audit));
  return SBO_OK;
}

int sbo_set_option(sbo_ctx* c, const char* key, int64_t value) {
  if (!c || !key) return fail(SBO_E_INVALID, "ctx/key is NULL");
  if (!strcmp(key, "halo_spec")) {
    c->halo_spec = value ? 1 : 0;
    for (auto& g : c->halo_guess) g = -1;
    return SBO_OK;
  }
  if (!strcmp(key, "comm_events")) {
    c->comm_events = value ? 1 : 0;
    return SBO_OK;
  }
  if (!strcmp(key, "cheb_tol_e17")) {
    c->cheb_tol = (double)value * 1e-17;
    c->bl.valid = false;
    c->bi.valid = false;
    c->posterior_valid = false;
    return SBO_OK;
  }
  if (!strcmp(key, "tensor_guess_pct")) {
    if (value < 10 || value > 400) return fail(SBO_E_INVALID, "tensor_guess_pct must be within 10 .. 400");
    c->tensor_guess_pct = (int)value;
    c->tn_valid = false;
    c->tn_bump = 0;
    c->posterior_valid = false;
    return SBO_OK;
  }
  if (!strcmp(key, "tensor_cheb")) {
    c->tensor_cheb = value ? 1 : 0;
    c->tn_valid = false;
    c->posterior_valid = false;
    return SBO_OK;
  }
  if (!strcmp(key, "exact_lazy")) {
    c->exact_lazy = value < 0 || value > 2 ? 1 : (int)value;      // 2: the late-recheck path runs on every sweep (test)
    return SBO_OK;
  }
  if (!strcmp(key, "chol_async")) {
    c->chol_async = value ? 1 : 0;
    return SBO_OK;
  }
  if (!strcmp(key, "bilinear")) {
    if (value < 0 || value > 2) return fail(SBO_E_INVALID, "bilinear must be 0 (off), 1 (on; a model's first sweep by node interpolation) or 2 (on, K1b's plan from the first sweep)");
    c->bilinear = (int)value;
    c->bl.valid = false;
    c->bi.valid = false;
    c->posterior_valid = false;
    return SBO_OK;
  }
  if (!strcmp(key, "phase_events")) {
    c->phase_events = value ? 1 : 0;
    return SBO_OK;
  }
  if (!strcmp(key, "scan_waves")) {
    c->scan_waves = (int)value;   // 0: off; 8 / 16 / 32 / 64: lanes per listed candidate (anything else: the default, 16)
    return SBO_OK;
  }
  if (!strcmp(key, "set_lanes")) {
    c->set_lanes = value ? 1 : 0;
    return SBO_OK;
  }
  if (!strcmp(key, "result_mirror")) {
    c->result_mirror = value ? 1 : 0;
    return SBO_OK;
  }
  if (!strcmp(key, "set_fuse")) {
    c->set_fuse = value ? 1 : 0;
    return SBO_OK;
  }
  if (!strcmp(key, "col_path")) {
    if (value < 0 || value > 2) return fail(SBO_E_INVALID, "col_path must be 0 (never), 1 (auto) or 2 (whenever the grid's shape allows)");
    c->col_path = (int)value;
    return SBO_OK;
  }
  if (!strcmp(key, "guard_audit")) {
    if (value < 0 || value > (1 << 20)) return fail(SBO_E_INVALID, "guard_audit: samples per sweep, 0 (off) .. 1048576");
    guard_audit_harvest(c, true);
    c->guard_audit = (int)value;
    return SBO_OK;
  }
  if (!strcmp(key, "guard_audit_scale_ppm")) {
    // (test hook: the audit compares against the band times value / 1e6 -- with a band a thousand times too narrow it MUST count
    // violations, which is how the suite shows that it compares real values; setting it clears the counts)
    if (value < 1 || value > 1000000000) return fail(SBO_E_INVALID, "guard_audit_scale_ppm: 1 .. 1e9");
    guard_audit_harvest(c, true);
    c->audit_scale = (double)value * 1e-6;
    c->audit_samples = c->audit_violations = 0;
    c->audit_worst = 0.0;
    return SBO_OK;
  }
  if (!strcmp(key, "guard_audit_every")) {
    if (value < 1 || value > (1 << 20)) return fail(SBO_E_INVALID, "guard_audit_every: 1 .. 1048576 sweeps");
    c->guard_audit_every = (int)value;
    c->audit_tick = 0;
    return SBO_OK;
  }
  if (!strcmp(key, "grad_defer")) {
    if (value < 0 || value > 3) return fail(SBO_E_INVALID, "grad_defer must be 0 .. 3");
    c->grad_defer = (int)value;
    c->bi.valid = false;
    return SBO_OK;
  }
  if (!strcmp(key, "col_overlap")) {
    c->col_overlap = value ? 1 : 0;
    return SBO_OK;
  }
  if (!strcmp(key, "scan_blocks")) {
    c->scan_blocks = value ? 1 : 0;
    return SBO_OK;
  }
  if (!strcmp(key, "fuse_classify")) {
    if (value < -1 || value > 1) return fail(SBO_E_INVALID, "fuse_classify must be -1 (auto), 0 or 1");
    c->fuse_classify = (int)value;
    return SBO_OK;
  }
  if (!strcmp(key, "goose_pairs")) {
    c->goose_pairs = value ? 1 : 0;
    return SBO_OK;
  }
  if (!strcmp(key, "guard_band")) {
    if (value < 0 || value > 2) return fail(SBO_E_INVALID, "guard_band must be 0 (off), 1 (on) or 2 (re-evaluate on every sweep)");
    c->guard_band = (int)value;
    c->bl.valid = false;
    c->bi.valid = false;                    // (plans measure their band when they are built)
    c->tn_valid = false;
    c->posterior_valid = false;
    return SBO_OK;
  }
  if (!strcmp(key, "fp64_recheck")) {
    c->fp64_recheck = value ? 1 : 0;       // (takes effect at the next sbo_model_set: the fp64 twin is built there)
    return SBO_OK;
  }
  if (!strcmp(key, "comm_selftest")) {
    if (value && !c->comm && !c->relay_allreduce)
      return fail(SBO_E_INVALID, "comm_selftest needs a communicator: call sbo_comm_init(ctx, 1, 0, id) with a unique id first");
    c->comm_selftest = value ? 1 : 0;
    return SBO_OK;
  }
  if (!strcmp(key, "posterior_path")) {
    if (value < 0 || value > 2) return fail(SBO_E_INVALID, "posterior_path must be 0 (auto), 1 (generic) or 2 (generic, chunked)");
    c->posterior_path = (int)value;
    c->posterior_valid = false;
    return SBO_OK;
  }
  return fail(SBO_E_INVALID, std::string("unknown option ") + key);
}

static int model_set_impl(sbo_ctx* c, int dtype, const char* kernel, int n, int d, int q, const double* X_mean,
                          const double* X_std, const double* Y_mean, const double* Y_std, const double* X_norm,
                          const double* Y_norm, const double* hypopt, const double* const* invK) {
  // (a standing audit of the last sweep reads the outgoing model's matrix out of the build workspace this call reuses)
  if (c) guard_audit_harvest(c, true);
  if (!c) return fail(SBO_E_INVALID, "ctx is NULL");
  if (!kernel || strcmp(kernel, "RBF") != 0)      // models/GP_Safe.py:159-162
    return fail(SBO_E_INVALID, std::string("ERROR no kernel with name ") + (kernel ? kernel : "(null)"));
  if (dtype != SBO_F64 && dtype != SBO_F32) return fail(SBO_E_INVALID, "dtype must be SBO_F64 or SBO_F32");
  if (n < 1 || n > SBO_MAX_N) return fail(SBO_E_INVALID, "n out of range [1, SBO_MAX_N]");
  if (d < 1 || d > SBO_MAX_D) return fail(SBO_E_INVALID, "d out of range [1, SBO_MAX_D]");
  if (q < 1 || q > SBO_MAX_Q) return fail(SBO_E_INVALID, "q out of range [1, SBO_MAX_Q]");
  if (!X_mean || !X_std || !Y_mean || !Y_std || !X_norm || !Y_norm || !hypopt)
    return fail(SBO_E_INVALID, "NULL model array");
  SBO_HIP(hipSetDevice(c->device));
  c->has_model = false;
  c->posterior_valid = false;
  c->masks_valid = false;
  ModelConst& mc = c->mc;
  memset(&mc, 0, sizeof(mc));
  mc.n = n; mc.d = d; mc.q = q;
  mc.npad = (n + 15) / 16 * 16;
  mc.dpad = d <= 2 ? 2 : (d <= 4 ? 4 : 8);
  mc.factor = invK ? SBO_FACTOR_INVK : SBO_FACTOR_CHOL;
  for (int a = 0; a < kMaxD; ++a) { mc.X_mean[a] = a < d ? X_mean[a] : 0.0; mc.X_std[a] = a < d ? X_std[a] : 1.0; mc.X_rstd[a] = 1.0 / mc.X_std[a]; }
  c->h_Xnorm.assign(X_norm, X_norm + (size_t)n * d);
  const double f32eps = (double)std::numeric_limits<float>::epsilon();
  for (int o = 0; o < q; ++o) {
    mc.Y_mean[o] = Y_mean[o];
    mc.Y_std[o] = Y_std[o];
    mc.mp[o] = (o == 0) ? 0.0 : (-2.0 * Y_mean[o]) / Y_std[o];          // GP_Safe.py:331-332
    mc.sf2[o] = std::exp(2.0 * hypopt[(size_t)d * q + o]);               // GP_Safe.py:338
    mc.sn2[o] = std::exp(2.0 * hypopt[(size_t)(d + 1) * q + o]) + f32eps;   // GP_Safe.py:229
    for (int a = 0; a < d; ++a) {
      const double ell = std::exp(2.0 * hypopt[(size_t)a * q + o]);
      mc.vinv[o][a] = std::pow(ell, -0.5);                               // GP_Safe.py:112
      mc.inv_ell[o][a] = 1.0 / ell;
    }
  }
  c->dtype = dtype;
  c->bl.valid = false;
    c->bi.valid = false;
  ++c->model_serial;
  // derived arrays (As, sqA, Xn, rhs), factorisation, alpha and the fragment images of the factor: on the device (model.hip)
  int rc = model_build(c, invK, X_norm, Y_norm);
  if (rc) return rc;
  // A grid is resident and qualifies for the GEMM posterior: its per-(model, grid) tables are enqueued now, so that they
  // run while the caller is on its way from this call to the sweep (the bases' ranks came back with the build's own
  // synchronisation; nothing here waits).  A grid change before the next sweep simply drops the plan.
  if (!c->is_shadow && interp_applicable(c)) {
    if ((rc = interp_setup(c))) return rc;
  } else if (!c->is_shadow && bilinear_applicable(c) && (rc = bilinear_setup(c))) return rc;
  if (dtype == SBO_F32 && c->fp64_recheck && !c->is_shadow) {
    // the fp64 twin: same constants, double arrays and factor images (built from the same inputs)
    if ((rc = shadow_ensure(c))) return rc;
    sbo_ctx* s = c->shadow;
    s->mc = c->mc;
    s->dtype = SBO_F64;
    s->has_model = false;
    s->posterior_valid = false;
    s->h_Xnorm = c->h_Xnorm;
    ++s->model_serial;
    if ((rc = model_build(s, invK, X_norm, Y_norm))) return rc;
    s->has_model = true;
  }
  c->has_model = true;
  return SBO_OK;
}

int sbo_model_set(sbo_ctx* c, int dtype, const char* kernel, int n, int d, int q, const double* X_mean,
                  const double* X_std, const double* Y_mean, const double* Y_std, const double* X_norm,
                  const double* Y_norm, const double* hypopt, const double* invK) {
  const double* parts[SBO_MAX_Q];
  if (invK && q >= 1 && q <= SBO_MAX_Q && n >= 1)
    for (int o = 0; o < q; ++o) parts[o] = invK + (size_t)o * n * n;
  return model_set_impl(c, dtype, kernel, n, d, q, X_mean, X_std, Y_mean, Y_std, X_norm, Y_norm, hypopt, invK ? parts : nullptr);
}

int sbo_model_set_list(sbo_ctx* c, int dtype, const char* kernel, int n, int d, int q, const double* X_mean,
                       const double* X_std, const double* Y_mean, const double* Y_std, const double* X_norm,
                       const double* Y_norm, const double* hypopt, const double* const* invK_list) {
  if (invK_list && q >= 1 && q <= SBO_MAX_Q)
    for (int o = 0; o < q; ++o)
      if (!invK_list[o]) return fail(SBO_E_INVALID, "invK_list holds a NULL matrix");
  return model_set_impl(c, dtype, kernel, n, d, q, X_mean, X_std, Y_mean, Y_std, X_norm, Y_norm, hypopt, invK_list);
}

// one more observation under frozen hyper-parameters and normalisation, O(n^2) on the
// device instead of a refit (the reference always refits and renormalises, models/GP_Safe.py:283-304 -- this is an
// opt-in fast path, not its behaviour).
int sbo_model_append(sbo_ctx* c, const double* x_norm_new, const double* y_norm_new) {
  if (c) guard_audit_harvest(c, true);
  if (!c || !x_norm_new || !y_norm_new) return fail(SBO_E_INVALID, "NULL argument");
  if (!c->has_model || !c->Fplain.p) return fail(SBO_E_NO_MODEL, "sbo_model_set has not been called");
  ModelConst& mc = c->mc;
  const int n = mc.n, d = mc.d, q = mc.q;
  if (n + 1 > SBO_MAX_N) return fail(SBO_E_UNSUPPORTED, "model is at its capacity: rebuild it with sbo_model_set");
  SBO_HIP(hipSetDevice(c->device));
  { const int rcf = factor_sync(c); if (rcf) return rcf; }     // (the update works on the resident factor)
  c->invk_img_valid = false;                                    // (the images of the caller's invK do not follow an append)
  c->invk_w_valid = false;
  // cross-covariances of the new point with the expanded distance of the reference (GP_Safe.py:115-119, 166)
  std::vector<double> kvec((size_t)q * n);
  double kappa[kMaxQ], rho[kMaxQ];
  for (int o = 0; o < q; ++o) {
    double bsq = 0.0, bvec[kMaxD];
    for (int a = 0; a < d; ++a) { bvec[a] = x_norm_new[a] * mc.vinv[o][a]; bsq += bvec[a] * bvec[a]; }
    for (int j = 0; j < n; ++j) {
      double dot = 0.0, asq = 0.0;
      for (int a = 0; a < d; ++a) {
        const double v = c->h_Xnorm[(size_t)j * d + a] * mc.vinv[o][a];
        dot += v * bvec[a];
        asq += v * v;
      }
      kvec[(size_t)o * n + j] = mc.sf2[o] * std::exp(-0.5 * ((-2.0 * dot + asq) + bsq));
    }
    kappa[o] = mc.sf2[o] + mc.sn2[o];
    rho[o] = y_norm_new[o] - mc.mp[o];
  }
  int rc = model_append(c, kvec, kappa, rho);
  if (rc) return rc;
  // the derived arrays with the new row (device), then the re-pack of the factor images
  c->h_Xnorm.insert(c->h_Xnorm.end(), x_norm_new, x_norm_new + d);
  mc.n = n + 1;
  mc.npad = (mc.n + 15) / 16 * 16;
  if ((rc = model_prep(c, c->h_Xnorm.data()))) return rc;
  if ((rc = model_repack(c))) return rc;
  ++c->model_serial;
  if (c->shadow && c->shadow->has_model && (rc = sbo_model_append(c->shadow, x_norm_new, y_norm_new))) return rc;
  c->posterior_valid = false;
  c->masks_valid = false;
  c->bl.valid = false;
    c->bi.valid = false;
  return SBO_OK;
}

static int alloc_workspace(sbo_ctx* c) {
  const size_t es = c->dtype == SBO_F64 ? 8 : 4;
  const size_t n = (size_t)std::max<long long>(c->cs.n_local, 1);
  int rc;
  if ((rc = ensure(c->mean, es * n * c->mc.q))) return rc;
  if ((rc = ensure(c->var, es * n * c->mc.q))) return rc;
  return SBO_OK;
}

int sbo_candidates_points(sbo_ctx* c, const void* points, int points_dtype, int64_t n_local, int d, int64_t first) {
  if (c) guard_audit_harvest(c, true);
  if (!c) return fail(SBO_E_INVALID, "ctx is NULL");
  if (n_local < 0 || (n_local > 0 && !points)) return fail(SBO_E_INVALID, "bad points / n_local");
  if (d < 1 || d > SBO_MAX_D) return fail(SBO_E_INVALID, "d out of range");
  if (points_dtype != SBO_F64 && points_dtype != SBO_F32) return fail(SBO_E_INVALID, "points_dtype");
  SBO_HIP(hipSetDevice(c->device));
  const size_t es = points_dtype == SBO_F64 ? 8 : 4;
  int rc = ensure(c->pts, es * (size_t)std::max<int64_t>(n_local, 1) * d);
  if (rc) return rc;
  if (n_local) {
    SBO_HIP(hipMemcpyAsync(c->pts.p, points, es * (size_t)n_local * d, hipMemcpyHostToDevice, c->stream));
    SBO_HIP(hipStreamSynchronize(c->stream));
  }
  memset(&c->cs, 0, sizeof(c->cs));
  c->cs.kind = 0; c->cs.d = d; c->cs.pts_dtype = points_dtype; c->cs.pts = c->pts.p;
  c->cs.n_local = n_local; c->cs.first = first;
  c->grid_total = n_local;
  c->sharded = false;
  c->has_cand = true;
  c->bl.valid = false;
    c->bi.valid = false;
  c->posterior_valid = false;
  c->masks_valid = false;
  return SBO_OK;
}

int sbo_candidates_grid(sbo_ctx* c, int d, const double* lo, const double* hi, const int64_t* count, int64_t first,
                        int64_t n_local) {
  if (c) guard_audit_harvest(c, true);
  if (!c) return fail(SBO_E_INVALID, "ctx is NULL");
  if (d < 1 || d > SBO_MAX_D || !lo || !hi || !count) return fail(SBO_E_INVALID, "bad grid description");
  long double total = 1;
  for (int a = 0; a < d; ++a) {
    if (count[a] < 1) return fail(SBO_E_INVALID, "grid count must be >= 1");
    total *= (long double)count[a];
  }
  if (total > 9.0e18L) return fail(SBO_E_INVALID, "grid too large");
  if (first < 0 || n_local < 0 || (long double)first + (long double)n_local > total)
    return fail(SBO_E_INVALID, "shard range outside the grid");
  memset(&c->cs, 0, sizeof(c->cs));
  c->cs.kind = 1; c->cs.d = d; c->cs.n_local = n_local; c->cs.first = first;
  for (int a = 0; a < d; ++a) {
    c->cs.lo[a] = lo[a]; c->cs.hi[a] = hi[a]; c->cs.count[a] = count[a];
    c->cs.step[a] = count[a] > 1 ? (hi[a] - lo[a]) / (double)(count[a] - 1) : 0.0;
  }
  for (int a = d; a < kMaxD; ++a) c->cs.count[a] = 1;
  c->grid_total = (long long)total;
  for (auto& g : c->halo_guess) g = -1;
  c->sharded = false;
  c->has_cand = true;
  c->bl.valid = false;
    c->bi.valid = false;
  c->posterior_valid = false;
  c->masks_valid = false;
  return SBO_OK;
}

int sbo_candidates_grid_sharded(sbo_ctx* c, int d, const double* lo, const double* hi, const int64_t* count,
                                int64_t* first_out, int64_t* n_local_out) {
  if (!c) return fail(SBO_E_INVALID, "ctx is NULL");
  if (d < 1 || d > SBO_MAX_D || !lo || !hi || !count) return fail(SBO_E_INVALID, "bad grid description");
  long long stride = 1;
  for (int a = 0; a < d - 1; ++a) {
    if (count[a] < 1) return fail(SBO_E_INVALID, "grid count must be >= 1");
    stride *= count[a];
  }
  if (count[d - 1] < 1) return fail(SBO_E_INVALID, "grid count must be >= 1");
  const long long planes = count[d - 1];
  const int W = c->world, r = c->rank;
  std::vector<long long> first_of(W + 1);
  for (int i = 0; i <= W; ++i) first_of[i] = (planes * i / W) * stride;
  int rc = sbo_candidates_grid(c, d, lo, hi, count, first_of[r], first_of[r + 1] - first_of[r]);
  if (rc) return rc;
  c->first_of = first_of;
  c->sharded = true;
  if ((rc = ensure(c->shard_first, sizeof(long long) * (W + 1)))) return rc;
  SBO_HIP(hipMemcpy(c->shard_first.p, first_of.data(), sizeof(long long) * (W + 1), hipMemcpyHostToDevice));
  if (first_out) *first_out = first_of[r];
  if (n_local_out) *n_local_out = first_of[r + 1] - first_of[r];
  return SBO_OK;
}

static int check_ready(sbo_ctx* c) {
  if (!c) return fail(SBO_E_INVALID, "ctx is NULL");
  if (!c->has_model) return fail(SBO_E_NO_MODEL, "sbo_model_set has not been called");
  if (!c->has_cand) return fail(SBO_E_NO_CANDIDATES, "no candidates resident");
  if (c->cs.d != c->mc.d) return fail(SBO_E_INVALID, "ERROR W and X_norm dimension should be same");  // GP_Safe.py:159
  return SBO_OK;
}

// K1 without synchronisation or timing (used inside the sweeps)
int sbo_posterior_enqueue(sbo_ctx* c) {
  int rc = check_ready(c);
  if (rc) return rc;
  SBO_HIP(hipSetDevice(c->device));
  guard_audit_harvest(c, false);
  if ((rc = alloc_workspace(c))) return rc;
  // (a standing audit of the last sweep may not have taken its sample of mean / var yet: it is a few microseconds of work on its stream)
  if (c->audit_pending) SBO_HIP(hipStreamWaitEvent(c->stream, c->ev_audit[0], 0));
  if (c->cs.n_local > 0 && (rc = launch_posterior(c))) return rc;
  c->posterior_valid = true;
  return SBO_OK;
}

double sbo_algorithmic_flops(const sbo_ctx* c) {
  const double n = c->mc.n, d = c->mc.d, q = c->mc.q;
  return q * (n * n + (2 * d + 10) * n) * (double)c->cs.n_local;
}

int sbo_posterior_run(sbo_ctx* c) {
  int rc = check_ready(c);
  if (rc) return rc;
  SBO_HIP(hipSetDevice(c->device));
  SBO_HIP(hipEventRecord(c->ev[0], c->stream));
  if ((rc = sbo_posterior_enqueue(c))) return rc;
  if (!c->k1_stop_attached) SBO_HIP(hipEventRecord(c->ev[1], c->stream));
  c->k1_stop_attached = false;
  SBO_HIP(hipEventSynchronize(c->ev[1]));
  float ms = 0;
  SBO_HIP(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
  memset(&c->prof, 0, sizeof(c->prof));
  c->prof.posterior_ms = ms;
  c->prof.total_ms = ms;
  c->prof.posterior_flops = sbo_algorithmic_flops(c);
  c->prof.candidates = c->cs.n_local;
  c->prof.posterior_launches = c->cs.n_local > 0 ? 1 : 0;
  return SBO_OK;
}

int sbo_posterior_get(sbo_ctx* c, void* mean_out, void* var_out) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (!c->posterior_valid && (rc = sbo_posterior_run(c))) return rc;
  const size_t es = c->dtype == SBO_F64 ? 8 : 4;
  const size_t bytes = es * (size_t)c->cs.n_local * c->mc.q;
  if (bytes == 0) return SBO_OK;
  DevBuf tmp;
  if ((rc = ensure(tmp, bytes))) return rc;
  for (int which = 0; which < 2; ++which) {
    void* dst = which ? var_out : mean_out;
    if (!dst) continue;
    rc = launch_soa_to_aos(c, which ? c->var.p : c->mean.p, tmp.p);
    if (rc) { release(tmp); return rc; }
    hipError_t e = hipMemcpyAsync(dst, tmp.p, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { release(tmp); return hip_fail(e, "copy posterior to host"); }
  }
  release(tmp);
  return SBO_OK;
}

int sbo_bounds(sbo_ctx* c, double b, int index, int kind, void* out) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (index < 0 || index >= c->mc.q) return fail(SBO_E_INVALID, "output index out of range");
  if (kind < SBO_MEAN || kind > SBO_VAR) return fail(SBO_E_INVALID, "bad bound kind");
  if (!out) return fail(SBO_E_INVALID, "out is NULL");
  if (!c->posterior_valid && (rc = sbo_posterior_run(c))) return rc;
  const size_t es = c->dtype == SBO_F64 ? 8 : 4;
  const size_t bytes = es * (size_t)c->cs.n_local;
  if (bytes == 0) return SBO_OK;
  DevBuf tmp;
  if ((rc = ensure(tmp, bytes))) return rc;
  rc = launch_bound(c, b, index, kind, tmp.p);
  if (!rc) {
    hipError_t e = hipMemcpyAsync(out, tmp.p, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = hip_fail(e, "copy bounds to host");
  }
  release(tmp);
  return rc;
}

int sbo_profile_get(sbo_ctx* c, sbo_profile* out) {
  if (!c || !out) return fail(SBO_E_INVALID, "NULL argument");
  *out = c->prof;
  out->posterior_kernel = c->last_k1;
  out->posterior_executed_flops = c->prof.posterior_launches ? c->last_k1_flops : 0.0;
  out->posterior_setup_ms = c->bl.setup_ms;
  guard_audit_harvest(c, false);
  out->guard_audit_samples = c->audit_samples;
  out->guard_audit_violations = c->audit_violations;
  out->guard_audit_worst = c->audit_worst;
  // the guard band of the posterior that is resident (K1b measures it on the device: a small read-back, off the hot path)
  for (int o = 0; o < SBO_MAX_Q; ++o)
    out->guard_dm[o] = out->guard_dv[o] = out->guard_rl[o] = out->guard_analytic_dm[o] = out->guard_analytic_dv[o] = out->guard_probe_dm[o] = out->guard_probe_dv[o] = 0.0;
  if (c->gb_active && c->guard_band && c->gb.p && c->posterior_valid) {
    if (!c->gb_host_valid) {               // (once per plan: the profile is read after every sweep of a timing loop)
      GuardBand hb;
      if (c->gb_mirrored) {
        // (the plan's band kernel wrote a copy into the pinned block; the sweep whose posterior is valid has synchronised since)
        memcpy(&hb, c->h_back + kGbMirrorOffset, sizeof(hb));
      } else {
        SBO_HIP(hipSetDevice(c->device));
        SBO_HIP(hipMemcpyAsync(&hb, c->gb.p, sizeof(hb), hipMemcpyDeviceToHost, c->stream));
        SBO_HIP(hipStreamSynchronize(c->stream));
      }
      for (int o = 0; o < SBO_MAX_Q; ++o) {
        c->gb_host[o] = hb.dm[o]; c->gb_host[SBO_MAX_Q + o] = hb.dv[o]; c->gb_host[2 * SBO_MAX_Q + o] = hb.rl[o];
        c->gb_host[3 * SBO_MAX_Q + o] = hb.an_m[o]; c->gb_host[4 * SBO_MAX_Q + o] = hb.an_v[o];
        c->gb_host[5 * SBO_MAX_Q + o] = hb.pr_m[o]; c->gb_host[6 * SBO_MAX_Q + o] = hb.pr_v[o];
      }
      c->gb_host_valid = true;
    }
    for (int o = 0; o < c->mc.q; ++o) {
      out->guard_dm[o] = c->gb_host[o];
      out->guard_dv[o] = c->gb_host[SBO_MAX_Q + o];
      out->guard_rl[o] = c->gb_host[2 * SBO_MAX_Q + o];
      out->guard_analytic_dm[o] = c->gb_host[3 * SBO_MAX_Q + o];
      out->guard_analytic_dv[o] = c->gb_host[4 * SBO_MAX_Q + o];
      out->guard_probe_dm[o] = c->gb_host[5 * SBO_MAX_Q + o];
      out->guard_probe_dv[o] = c->gb_host[6 * SBO_MAX_Q + o];
    }
  }
  return SBO_OK;
}

}  // extern "C"
