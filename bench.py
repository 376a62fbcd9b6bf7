#!/usr/bin/env python3
"""bench.py -- candidate-points/sec of the SafeOpt posterior + safe-set sweep on MI355X.

Contract (see the task statement): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 the
driver launches one rank per GPU with ``torch.distributed.run``.  One *step* is one full SafeOpt sweep
(K1 posterior, S/U/M masks, u*, Lipschitz bound, expander sets, masked arg-max, result to host) over the
candidates resident in HBM.

Workload at every N: the configuration the north-star target is quoted on -- config H, Benoit 2-D, 4096 x 4096 implicit
grid, n = 512 observations, q = 2 outputs, fp64 -- STRONG-scaled: the grid is sharded over the ranks by whole rows of its
slowest axis; the ranks exchange u* / L / radius keys and the fully-unsafe mask (one all-gather) and the arg-max candidates
(one sum all-reduce).  BASELINE.json configs[1] (config B, 2048 x 2048, n = 128) and the other configs are measured in
``extra`` of the same process at N = 1; at N > 1 ``extra`` holds config D (128^4, BASELINE.json configs[3]) strong-scaled.

What the JSON line reports (N = 1 adds the last five):
  value / ms_per_step   sweeps of ONE RESIDENT MODEL (the metric of BASELINE.json: device time of the whole sweep with the
                        inputs resident in HBM), K timed steps between barriers.
  roofline              matrix-core side: the posterior kernels (K1).  ``achieved`` = min(ALGORITHMIC flops of SURVEY.md
                        8(d), flops ISSUED on the matrix cores) / K1 time -- a utilisation <= 1; ``algorithmic`` is the
                        literal SURVEY figure (above the peak with K1b, whose GEMMs have an inner dimension ~r(r+1)/2 that
                        does not grow with n); ``executed`` the issued flops.
  roofline.hbm          HBM side: the set phase K3-K5 (classification, minimiser, expander transform, arg-max) --
                        SURVEY.md 8(d) bytes (2 q s + 4 per candidate) / set-phase device time, against 8.0 TB/s
                        (datasheet) and 6.29 TB/s (measured copy rate of the guide).
  comm                  (N > 1) collectives per sweep: calls, bytes a rank sends, event-timed microseconds per call (from a few
                        extra sweeps with option comm_events, outside the timed region).
  iteration             what ONE SafeOpt ITERATION costs: the reference refits and sweeps once per model
                        (test/test_SafeOpt.py:144-179), so here two data sets alternate and every timed step is
                        set_model (upload + alpha on the device; the reverse factor of the caller's invK is never made) + the
                        model's first sweep, which runs on node-interpolated coefficients (K1i) -- K1b's plan belongs to a
                        model that is swept again, as in the resident-model figure above.
  table_kernel          the same resident-model sweep with the exact O(n^2)-per-candidate kernel K1g (here on H; B and D in
                        `extra`), with `agreement`: every count and index of the two posteriors' results compared.
  config.result         counts, indices and the guard-band bookkeeping (`guard_band` = decisions the approximating posterior
                        could not make unconditionally on the fast path; 0 on every BASELINE config).
  extra                 the other BASELINE.json configs on the one GPU: B (2048^2, n = 128; with its iteration cost and K1g figure),
                        C (Williams-Otto, 1024^2, n = 256, q = 3: GoOSE and SafeOpt, with iteration costs), D (128^4, the whole
                        grid of the 8-GPU config), E (10^7 scattered 6-D points, n = 2048, fp32 with the fp64 recheck).
  cpu_baseline          the C + OpenMP restatement of the same sweep on the box's host cores (`numpy`: the NumPy oracle on a
                        bounded prefix) on the first 2^21 candidates of config H's own grid (n = 512); `config_B`: config B's whole grid.

No torch anywhere: the ranks of the launcher meet through safebo_amd.distributed.TcpGroup (stdlib sockets: unique-id
broadcast, barriers, max over ranks); device memory, streams and the collectives on the data path belong to libsafebo.so.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6      # AMD MI355X datasheet, FP64 matrix == FP64 vector (guide has no f64 row)
FP64_MFMA_MEASURED_TFLOPS = {"v_mfma_f64_4x4x4_4b_f64": 75.6, "v_mfma_f64_16x16x4_f64": 47.9}   # profiles/r01_mfma_probe.txt
FP32_MATRIX_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md, f32-input MFMA
HBM_PEAK_TBS, HBM_MEASURED_TBS = 8.0, 6.29    # MI355X_MICROARCH.md: datasheet / measured float4 copy
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # HBM bytes per K1 launch from rocprofv3 --pmc
K1_NAMES = {1: "k_posterior", 2: "k_posterior_chunked", 3: "k_posterior_grid", 4: "k_bpost (+ k_bstage1, stage 1)",
            5: "k_t_final (+ k_posterior_grid on the Chebyshev nodes, k_t_mode)",
            6: "k_bpost on node-interpolated coefficients (K1i: + k_bgemm invK K*^T at the Chebyshev nodes, k_i_dct, k_bstage1)"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="H", help="synthetic config (H = 4096^2, n = 512: the north-star target's; B = BASELINE.json configs[1])")
    ap.add_argument("--n", type=int, default=None, help="override the number of observations")
    ap.add_argument("--cpu-sample", type=int, default=1 << 21, help="candidates timed by the CPU baseline (0 = skip)")
    ap.add_argument("--points", type=int, default=10_000_000, help="candidates per rank for the scattered config E")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong: fixed grid, sharded over the ranks (default); weak: the grid's slowest axis grows with the ranks")
    ap.add_argument("--posterior", choices=["auto", "table"], default="auto",
                    help="auto: fp64 2-D grids use the bilinear GEMM posterior (K1b) when its bases qualify; "
                         "table: force the separable-table kernel (K1g), the O(n^2)-per-candidate contraction")
    ap.add_argument("--sweep", choices=["safeopt", "goose"], default="safeopt")
    ap.add_argument("--no-extra", action="store_true", help="skip the iteration / table-kernel / H / C records")
    ap.add_argument("--lean", type=int, choices=[0, 1, 2], default=2,
                    help="sbo_sweep_opts.lean of the SafeOpt sweeps: 2 (default) the caller wants the sweep's RESULT -- sets, indices, counts, u*, "
                         "the constraints' Lipschitz keys --: the objective's mean / var are neither stored nor evaluated on posterior tiles "
                         "without a safe candidate (no stage of the sweep reads them) and its Lipschitz key, which no sweep reads, is left out; "
                         "1 evaluated but not stored; 0 the whole posterior stays resident.  The other levels are reported beside the main line "
                         "(config.full_posterior, config.lean_1) with result_identical")
    return ap.parse_args()


def cpu_baseline(cfg, count, sample, prefix=None):
    """The CPU restatement of the same sweep on the box's host cores (rank 0, N = 1 only).  ``prefix``: the C + OpenMP column runs
    on the first `prefix` candidates of the grid instead of all of it (config H: a whole sweep would be minutes of CPU work).

    Main figure: oracle/c/sweep_omp.c -- the reference formulation (explicit invK, k^T invK per candidate, models/GP_Safe.py:
    310-352; bounds, S / U / M masks, u*, arg-max, models/SafeOpt.py:34-66) in C + OpenMP over the WHOLE grid, best of the
    thread counts tried (a box exposes more cores than its CPU share).  Beside it the NumPy oracle on a bounded prefix
    (`numpy`).  The quadratic brute-force expander of the oracle is left out of both (that favours the CPU figures)."""
    import oracle
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    total = int(np.prod(count))
    visible = len(os.sched_getaffinity(0))
    out = None
    try:
        from oracle import omp
        omp.safeopt_sweep(lo, hi, count, cfg["ds"], cfg["b"], n=min(total, 1 << 16))        # build + warm the thread pool
        best = None
        done = total if prefix is None else min(int(prefix), total)
        for th in sorted({min(visible, t) for t in (16, 32, 64, 128)}):
            omp.set_threads(th)
            t0 = time.perf_counter()
            r = omp.safeopt_sweep(lo, hi, count, cfg["ds"], cfg["b"], n=done) if done < total else omp.safeopt_sweep(lo, hi, count, cfg["ds"], cfg["b"])
            dt = time.perf_counter() - t0
            if best is None or done / dt > best[0]:
                best = (done / dt, th, dt, r)
        rate, th, dt, r = best
        out = {"value": rate, "unit": "candidates/s", "cores": th, "kind": "port",
               "sample": (f"all {total}" if done == total else f"the first {done} of the {total}") + " candidates of the same grid: posterior + bounds + S/U/M masks + u* + arg-max in C + OpenMP "
                         f"(oracle/c/sweep_omp.c, {th} threads -- the best of 16/32/64/128 --, {visible} cores visible), {dt:.2f} s; "
                         f"|S| = {r['count_S']}, minimiser {r['minimizer_index']}; the quadratic brute-force expander is excluded"}
    except Exception as e:           # (no compiler on the box and no prebuilt library: the NumPy column alone)
        print(f"[bench] C + OpenMP CPU column unavailable ({type(e).__name__}: {e}); reporting the NumPy oracle only", file=sys.stderr)
    sample = min(sample, total)
    if sample <= 0 and out is not None:
        return out
    pts = oracle.grid_points(lo, hi, count, first=0, n=sample)
    oracle.gp_inference(pts[:4096], cfg["ds"])           # warm the BLAS threads
    t0 = time.perf_counter()
    mean, var = oracle.gp_inference(pts, cfg["ds"])
    lcb, ucb = oracle.bounds(mean, var, cfg["b"])
    S = np.all(lcb[:, 1:] >= 0, axis=1)
    U = np.all(lcb[:, 1:] <= 0, axis=1)
    if S.any():
        u_star = np.min(ucb[S, 0])
        M = S & (lcb[:, 0] <= u_star)
        int(np.argmax(np.where(M, var[:, 0], -np.inf)))
    dt = time.perf_counter() - t0
    del U
    threads = visible
    try:
        from threadpoolctl import threadpool_info
        blas = [p["num_threads"] for p in threadpool_info() if p.get("user_api") == "blas"]
        threads = max(blas) if blas else threads
    except Exception:
        pass
    rec = {"value": sample / dt, "unit": "candidates/s", "cores": threads,
           "sample": f"first {sample} candidates, the NumPy oracle ({threads} BLAS threads), {dt:.2f} s"}
    if out is None:
        return {**rec, "kind": "port", "sample": rec["sample"] + "; posterior + bounds + S/U/M masks + u* + arg-max, expander excluded"}
    out["numpy"] = rec
    return out


def result_record(res, kind):
    """What a sweep returned, with the guard-band bookkeeping of the approximating posteriors (K1b / K1t): `guard_band` =
    decisions that fell inside the band of the posterior's measured deviation on the fast path (0: every mask byte and index is
    the exact evaluator's, unconditionally), `guard_rechecks` = candidates re-evaluated with the exact formula when it was not."""
    if kind == "safeopt":
        out = {"count_S": res["count_S"], "count_M": res["count_M"], "count_G": [int(x) for x in res["count_G"]],
               "minimizer_index": res["minimizer_index"], "expander_index": res["expander_index"],
               "expander_index_c": [int(x) for x in res["expander_index_c"]]}
    else:
        out = {"count_S": res["count_S"], "count_O": [int(x) for x in res["count_O"]], "safe_min_index": res["safe_min_index"],
               "target_index": res["target_index"], "explore_index": res["explore_index"]}
    out.update({"exact_rechecks": res["n_exact_rechecks"], "guard_band": int(res["guard_band"]), "guard_rechecks": int(res["guard_rechecks"]),
                "guard_passes": int(res["guard_passes"])})
    return out


def table_kernel_record(eng, cfg, step, kind, res, kernel_name, n_total, barrier, steps=5):
    """The same resident-model sweep with the exact O(n^2)-per-candidate separable-table kernel K1g, same process, and whether
    the two posteriors' sweeps AGREE: every count and every index of the result records must be equal."""
    opt = "bilinear" if kernel_name.startswith("k_bpost") else "tensor_cheb"
    # (the iteration record before this one alternated two models: this config's own model again, and a fresh result of it)
    eng.set_model(cfg["ds"], dtype=cfg["dtype"], use_invK=cfg["dtype"] == "f64")
    res = step()
    eng.set_option(opt, 0)
    el_t, rows_t, res_t = timed_resident(eng, step, steps, 1, barrier)
    eng.set_option(opt, 1)
    rt, _ = mfma_roofline(cfg, rows_t, n_total)
    a, b_ = result_record(res, kind), result_record(res_t, kind)
    keys = [k for k in a if not k.startswith("guard_") and k != "exact_rechecks"]
    diff = [k for k in keys if a[k] != b_[k]]
    return {"kernel": "k_posterior_grid", "device_ms_per_step": rt["device_ms_per_step"], "kernel_ms": rt["kernel_ms"],
            "value": n_total * steps / el_t, "unit": "candidates/s", "achieved": rt["algorithmic"]["achieved"],
            "frac": rt["algorithmic"]["frac"], "frac_definition": "SURVEY.md 8(d) algorithmic flops / kernel time / peak",
            "agreement": {"verdict": "identical" if not diff else "DIFFERENT: " + ", ".join(diff), "compared": keys,
                          "table_kernel_result": {k: b_[k] for k in keys},
                          "note": "counts (|S|, |M|, |G_i| or |O_i|) and indices of this config's sweep with " + kernel_name.split(" ")[0]
                                  + " against the exact table kernel's, same model and grid, same run"}}


def sweep_fn(eng, kind, b, lean=0):
    return (lambda: eng.sweep_safeopt(b, lean=lean)) if kind == "safeopt" else (lambda: eng.sweep_goose(b))


def hbm_roofline(q, es, n_local, rows):
    """the classification / expander / arg-max passes are HBM-bound, ~ 2 q s + 4 bytes per candidate
    (mean and var of every output read once, three mask bytes and the transform's verdict written), no reuse;
    ``set_phase_ms`` = device time between the K1 stop event and the end of the sweep."""
    per_cand = 2 * q * es + 4
    byts = float(per_cand) * n_local
    set_ms = float(np.mean([p["total_ms"] - p["posterior_ms"] - p["recheck_ms"] for p in rows]))
    tbs = byts / (set_ms * 1e-3) / 1e12 if set_ms > 0 else 0.0
    return {"bound": "hbm", "kernels": "set phase K3-K5: classification, distance transforms, verdicts, arg-reductions",
            "bytes_per_candidate": per_cand, "bytes": byts, "set_phase_ms": set_ms, "achieved": tbs, "unit": "TB/s",
            "peak": HBM_PEAK_TBS, "frac": tbs / HBM_PEAK_TBS, "frac_vs_measured_copy": tbs / HBM_MEASURED_TBS,
            "definition": "SURVEY.md 8(d): (2 q s + 4) bytes per candidate / device time between the K1 stop event and the end of the sweep"}


def mfma_roofline(cfg, prof_rows, n_local):
    k1 = float(np.mean([p["posterior_ms"] for p in prof_rows]))
    flops = float(np.mean([p["posterior_flops"] for p in prof_rows]))
    issued = float(np.mean([p["posterior_executed_flops"] for p in prof_rows]))
    peak = FP64_MATRIX_PEAK_TFLOPS if cfg["dtype"] == "f64" else FP32_MATRIX_PEAK_TFLOPS
    alg = flops / (k1 * 1e-3) / 1e12
    exe = issued / (k1 * 1e-3) / 1e12
    achieved = min(alg, exe) if exe > 0 else alg
    kind = prof_rows[-1]["posterior_kernel"]
    return {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "frac_definition": "min(algorithmic flops, flops issued on the matrix cores) / K1 device time / peak: "
                               + ("the ISSUED term binds (K1b)" if 0 < exe < alg else "the ALGORITHMIC term binds"),
            "kernel": K1_NAMES.get(kind, "?"), "kernel_ms": k1,
            "executed": {"achieved": exe, "frac": exe / peak, "flops_per_candidate": issued / n_local},
            "algorithmic": {"achieved": alg, "frac": alg / peak, "flops_per_candidate": flops / n_local,
                            "definition": "SURVEY.md 8(d): q (n^2 + (2 d + 10) n) per candidate"},
            "peak_source": "AMD MI355X datasheet FP64 matrix (no f64 row in MI355X_MICROARCH.md)" if cfg["dtype"] == "f64" else "MI355X_MICROARCH.md f32 MFMA",
            "peak_measured_mfma_f64": FP64_MFMA_MEASURED_TFLOPS if cfg["dtype"] == "f64" else None,
            "device_ms_per_step": float(np.mean([p["total_ms"] for p in prof_rows]))}, kind


def timed_resident(eng, step, steps, warmup, barrier):
    for _ in range(warmup):
        step()
    rows = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = step()
        rows.append(eng.profile_struct())         # (the HIP-event times of this sweep: read here, turned into dicts behind the timed region)
    barrier()
    el = time.perf_counter() - t0
    return el, [eng.profile_dict(p) for p in rows], res


def timed_iterations(eng, models, dtype, step, steps, warmup, barrier):
    """Every step installs the OTHER model (upload, factorisation, alpha), which invalidates the K1b tables, and sweeps:
    the cost of one iteration of the reference loop on a resident candidate grid."""
    use_invK = dtype == "f64"
    for i in range(warmup):
        eng.set_model(models[i % 2], dtype=dtype, use_invK=use_invK)
        step()
    t_set, t_sweep, builds, dev, k1, kinds = [], [], [], [], [], set()
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        ta = time.perf_counter()
        eng.set_model(models[i % 2], dtype=dtype, use_invK=use_invK)
        tb = time.perf_counter()
        step()
        tc = time.perf_counter()
        p = eng.profile_struct()
        t_set.append(tb - ta)
        t_sweep.append(tc - tb)
        builds.append(p.posterior_setup_ms)
        dev.append(p.total_ms)
        k1.append(p.posterior_ms)
        kinds.add(K1_NAMES.get(p.posterior_kernel, "?"))
    barrier()
    el = time.perf_counter() - t0
    return {"ms_per_step": el * 1e3 / steps, "steps": steps,
            "set_model_ms": float(np.mean(t_set)) * 1e3, "sweep_call_ms": float(np.mean(t_sweep)) * 1e3,
            "table_build_ms": float(np.mean(builds)), "sweep_device_ms": float(np.mean(dev)),
            "k1_incl_tables_ms": float(np.mean(k1)), "k1_kernels": sorted(kinds), "ms_per_step_min": float(np.min(np.add(t_set, t_sweep))) * 1e3,
            "ms_per_step_median": float(np.median(np.add(t_set, t_sweep))) * 1e3, "ms_per_step_max": float(np.max(np.add(t_set, t_sweep))) * 1e3,
            "slow_steps": [(i, round(float(a + b) * 1e3, 3)) for i, (a, b) in enumerate(zip(t_set, t_sweep)) if a + b > 2.0 * np.median(np.add(t_set, t_sweep))],
            "definition": "two data sets alternate; every timed step = set_model (upload, alpha, the plan of the model's first sweep: "
                          "exact node values + Chebyshev coefficients, K1i) + one full sweep; wall clock between barriers"}


def make_configs(name, n=None):
    """(config, alternate data set of the same config).  Built with a handful of BLAS threads and before any timed region:
    the BLAS pool's idle workers spin for tens of milliseconds after a call, and on a box whose CPU quota is smaller than
    its visible cores that gets the timing thread throttled."""
    from safebo_amd import synthetic
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=4)
    except Exception:
        import contextlib
        ctx = contextlib.nullcontext()
    with ctx:
        cfg = synthetic.make_config(name, n=n)
        alt = synthetic.make_config(name, n=n, seed=synthetic.SEED0 + 100 + cfg["index"])
    return cfg, alt


def max_over_ranks(group, x):
    return float(group.all_reduce(np.array([x], dtype=np.float64), "max")[0]) if group is not None else x


def comm_record(eng, step, transport):
    """Collectives of one sweep on this rank: calls, bytes handed over, event-timed microseconds (three extra sweeps with an
    event pair around every collective -- option comm_events -- outside the timed region: the pairs cost bubbles)."""
    eng.set_option("comm_events", 1)
    rows = []
    for _ in range(3):
        step()
        rows.append(eng.profile())
    eng.set_option("comm_events", 0)
    calls = int(rows[-1]["comm_calls"])
    ms = float(np.mean([p["comm_ms"] for p in rows]))
    return {"transport": transport, "collectives_per_sweep": calls, "bytes_sent_per_rank": int(rows[-1]["comm_bytes"]),
            "us_per_sweep": ms * 1e3, "us_per_collective": ms * 1e3 / calls if calls else 0.0, "host_syncs_per_sweep": int(rows[-1]["host_syncs"]),
            "definition": "event pairs around every RCCL call of a sweep on the library's stream (rank 0); for the host relay of one-GPU "
                          "rehearsals the wall clock of the staged calls"}


def extra_record(eng, cfg, alt, name, kind, steps, barrier, points=None, group=None, table_kernel=False, transport="none", lean=0):
    """Resident-model sweep rate (+ iteration cost for the 2-D grids) of another config, same process.  ``points``: size of the
    explicit candidate list of a scattered config (E).  ``group``: the ranks of an N > 1 run -- the grid is then sharded over
    them (strong scaling) and the time is the slowest rank's."""
    from safebo_amd import synthetic
    use_invK = cfg["dtype"] == "f64"
    world = group.get_world_size() if group is not None else 1
    eng.set_model(cfg["ds"], dtype=cfg["dtype"], use_invK=use_invK)
    if cfg["count"] is None:
        n_total = int(points)
        eng.set_points(synthetic.scattered_points(cfg, n_total))
        where = f"explicit list of {n_total} scattered candidates"
    else:
        count = list(cfg["count"])
        n_total = int(np.prod(count))
        if group is not None:
            eng.set_grid_sharded(cfg["bound"][:, 0], cfg["bound"][:, 1], count)
        else:
            eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], count)
        where = f"implicit grid {'x'.join(map(str, count))} ({n_total} candidates" + (f", sharded over {world} ranks)" if group is not None else ")")
    step = sweep_fn(eng, kind, cfg["b"], lean=lean)
    el, rows, res = timed_resident(eng, step, steps, max(1, steps // 5), barrier)
    el = max_over_ranks(group, el)
    mf, _ = mfma_roofline(cfg, rows, n_total // world)
    out = {"config": f"config {name}: {cfg['plant']} {cfg['d']}-D {kind} sweep, {where}, n={cfg['n']}, q={cfg['q']}, b={cfg['b']}, {cfg['dtype']}",
           "sweep": kind, "value": n_total * steps / el, "unit": "candidates/s", "steps": steps, "ms_per_step": el * 1e3 / steps,
           "roofline": {k: mf[k] for k in ("achieved", "peak", "frac", "kernel", "kernel_ms", "device_ms_per_step", "frac_definition")},
           "roofline_hbm": hbm_roofline(cfg["q"], 8 if cfg["dtype"] == "f64" else 4, n_total // world, rows)}
    out["lean"] = int(lean) if kind == "safeopt" else 0          # (the sweep option this record was measured with)
    if group is not None:
        out["n_gpus"], out["scaling"] = world, "strong"
        out["comm"] = comm_record(eng, step, transport)
    out["roofline"]["executed_frac"] = mf["executed"]["frac"]
    out["roofline"]["algorithmic_frac"] = mf["algorithmic"]["frac"]
    if cfg["dtype"] == "f32":
        out["fp64_recheck"] = {"candidates_reevaluated": int(rows[-1]["fp64_rechecks"]), "ms": float(np.mean([p["recheck_ms"] for p in rows])),
                               "note": "fp32 posterior; candidates its bounds cannot decide are re-evaluated in fp64 so that the masks "
                                       "equal the fp64 result"}
    if alt is not None and cfg["count"] is not None and group is None:
        it = timed_iterations(eng, [alt["ds"], cfg["ds"]], cfg["dtype"], step, max(4, steps // 3), 2, barrier)
        it["value"] = n_total / (it["ms_per_step"] * 1e-3)
        it["unit"] = "candidates/s"
        out["iteration"] = it
    if table_kernel and (mf["kernel"].startswith("k_bpost") or mf["kernel"].startswith("k_t_final")):
        # the same sweep with the separable-table kernel (the O(n^2)-per-candidate contraction on MFMA), same run
        out["table_kernel"] = table_kernel_record(eng, cfg, step, kind, res, mf["kernel"], n_total, barrier, steps=5 if n_total <= 1 << 23 else 3)
    out["result"] = result_record(res, kind)
    return out


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")

    import safebo_amd
    from safebo_amd import synthetic
    safebo_amd._lib.load()        # libsafebo.so (and the RCCL it is linked against)

    group = None
    if world > 1:
        from safebo_amd import distributed
        group = distributed.init_from_env()       # stdlib TCP rendezvous among the launcher's ranks (no torch)

    cfg, alt = make_configs(args.config, n=args.n)
    scattered = cfg["count"] is None                      # config E: explicit list of scattered candidates
    default_run = args.config == "H" and args.n is None and args.sweep == "safeopt" and args.posterior == "auto"
    extras = world == 1 and not args.no_extra and not scattered
    extra_cfgs = {name: make_configs(name) for name in (("B", "C", "D", "E") if extras and default_run else ())}
    dcfg = make_configs("D")[0] if (world > 1 and not args.no_extra and default_run and args.scaling == "strong") else None
    if scattered:
        per_rank = args.points if args.scaling == "weak" else args.points // world
        n_total = per_rank * world
        count = None
    else:
        count = list(cfg["count"])
        if args.scaling == "weak":
            count[-1] *= world                           # weak scaling: slowest axis grows with the ranks
        n_total = int(np.prod(count))
    lo, hi = cfg["bound"][:, 0].copy(), cfg["bound"][:, 1].copy()

    # one GPU per process; SBO_BENCH_DEVICE pins every rank to one card (single-GPU rehearsal of the N>1 plumbing only)
    eng = safebo_amd.SweepEngine(int(os.environ.get("SBO_BENCH_DEVICE", local_rank)))
    for kv in filter(None, os.environ.get("SBO_BENCH_OPTIONS", "").split(",")):   # tuning knobs, e.g. "scan_waves=16"
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    transport = "none"
    if world > 1:
        # RCCL (unique id broadcast from rank 0).  Only a one-GPU rehearsal of the N > 1 plumbing (SBO_BENCH_DEVICE pins
        # every rank to one card, which RCCL refuses) may fall back to the host relay; ranks on distinct devices that
        # cannot form the communicator fail the run.
        transport = distributed.join_with_fallback(eng, group, allow_relay="SBO_BENCH_DEVICE" in os.environ)
    eng.set_model(cfg["ds"], dtype=cfg["dtype"], use_invK=(cfg["dtype"] == "f64"))
    if scattered:
        pts = synthetic.scattered_points(cfg, n_total)[rank * per_rank:(rank + 1) * per_rank]
        eng.set_points(pts, first=rank * per_rank)       # uploaded to HBM before the timed region
    else:
        eng.set_grid_sharded(lo, hi, count)              # candidates are implicit: resident by construction

    def barrier():
        eng.synchronize()
        if group is not None:
            group.barrier()
        eng.synchronize()

    step = sweep_fn(eng, args.sweep, cfg["b"], lean=args.lean)
    if args.posterior == "table":
        eng.set_option("bilinear", 0)
    elapsed, rows, res = timed_resident(eng, step, args.steps, args.warmup, barrier)
    elapsed = max_over_ranks(group, elapsed)
    comm = comm_record(eng, step, transport) if world > 1 else None
    # N > 1: config D (BASELINE.json configs[3]) strong-scaled over the same ranks, every rank takes part
    d_rec = extra_record(eng, dcfg, None, "D", "safeopt", 3, barrier, group=group, transport=transport) if dcfg is not None else None

    if rank == 0:
        n_local = n_total // world
        es = 8 if cfg["dtype"] == "f64" else 4
        roof, k1_kind = mfma_roofline(cfg, rows, n_local)
        roof["hbm"] = hbm_roofline(cfg["q"], es, n_local, rows)
        # what binds the posterior kernel.  K1b / K1i issue ~300 flops per candidate on the matrix cores whatever n is: their roof is the
        # 2 q s bytes per candidate they write and the instruction issue of their epilogues, not the matrix pipe -- `frac` above is kept
        # as the issued-flop utilisation (`executed`), the store fraction is the one to read; the SURVEY 8(d)-conformant figure of this
        # config is the exact table kernel's (table_kernel_frac below, same run)
        store_bytes = 2.0 * cfg["q"] * es * n_local
        roof["store_bytes_algorithmic"] = store_bytes
        roof["store_roofline_frac"] = store_bytes / (roof["kernel_ms"] * 1e-3) / 1e12 / HBM_PEAK_TBS
        roof["store_roofline_definition"] = ("2 q s bytes per candidate (the posterior API's output, SURVEY.md 8(d)) / K1 device time / 8 TB/s; a lean sweep "
                                             "writes fewer bytes than that (config.lean), the figure prices the time against the full output")
        if args.sweep == "safeopt" and args.lean and int(rows[-1].get("set_path", 0)) == 1 and not scattered and world == 1:
            # ... and against what THIS sweep writes: every constraint's mean / var, the objective's on the 64 x 128 posterior tiles that
            # hold a safe candidate (counted from the S mask of one more sweep, outside the timed region)
            eng.sweep_safeopt(cfg["b"], want_masks=True, lean=args.lean)
            Sm = eng.mask("S").reshape(count[1] // 64, 64, count[0] // 128, 128)
            tiles_S = int(Sm.any(axis=(1, 3)).sum())
            written = 2.0 * es * (n_local * (cfg["q"] - 1) + 8192.0 * tiles_S)
            roof["store_bytes_written"] = written
            roof["objective_tiles_with_a_safe_candidate"] = [tiles_S, (count[1] // 64) * (count[0] // 128)]
            roof["store_roofline_frac_written"] = written / (roof["kernel_ms"] * 1e-3) / 1e12 / HBM_PEAK_TBS
        if k1_kind in (4, 6):
            roof["bound"] = "hbm-store/issue"
        # HBM bytes of the K1 launch(es) are NOT measured by this run: they come from separate rocprofv3 --pmc passes of the
        # same command (tools/gpu_bench_profile.sh), committed with the kernel they belong to; null when no record matches
        roof["traffic"], roof["traffic_source"] = None, None
        if os.path.exists(PMC_TRAFFIC_FILE) and world == 1:
            key = f"{args.config}:n={cfg['ds']['X_norm'].shape[0]}" + (":K1b" if k1_kind == 4 else "")
            rec = json.load(open(PMC_TRAFFIC_FILE)).get(key)
            if rec:
                roof["traffic"] = rec["hbm_bytes_per_launch"]
                roof["traffic_source"] = (f"OFFLINE: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, "
                                          f"profiles/pmc_traffic.json[{key}] ({rec.get('kernel')}; collected {rec.get('collected', 'round 1')})")
        result = result_record(res, args.sweep)
        out = {
            "metric": "candidate-points/sec, SafeOpt posterior+safe-set sweep",
            "value": n_total * args.steps / elapsed, "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": f"config {args.config}: {cfg['plant']} {cfg['d']}-D {args.sweep} sweep of one RESIDENT model, "
                                   + (f"explicit list of {n_total} scattered candidates" if scattered else
                                      f"implicit grid {'x'.join(str(c) for c in count)} ({n_total} candidates)")
                                   + f", n={cfg['ds']['X_norm'].shape[0]} observations, q={cfg['q']} outputs, b={cfg['b']}"
                                   + (f", sharded over {world} ranks by rows of the slowest axis" if world > 1 else ""),
                       "per_gpu_candidates": n_local, "sweep": args.sweep, "collectives": transport, "result": result,
                       "lean": args.lean if args.sweep == "safeopt" else 0,
                       "set_path": {0: "byte masks", 1: "column words written by the GEMM posterior (sets_colpath.inc.hpp)"}[int(rows[-1].get("set_path", 0))]},
            "roofline": roof,
        }
        if args.sweep == "safeopt" and world == 1 and not args.no_extra:
            # the same sweep at the other lean levels, same process (short timed runs): what storing / evaluating the objective's
            # posterior where no stage of the sweep reads it costs
            for lv, key in ((0, "full_posterior"), (1, "lean_1"), (2, "lean_2")):
                if lv == args.lean:
                    continue
                el_v, rows_v, res_v = timed_resident(eng, sweep_fn(eng, args.sweep, cfg["b"], lean=lv), max(20, args.steps // 4), 3, barrier)
                same = result_record(res_v, args.sweep) == result
                out["config"][key] = {"lean": lv, "ms_per_step": el_v * 1e3 / max(20, args.steps // 4), "value": n_total * max(20, args.steps // 4) / el_v,
                                      "device_ms_per_step": float(np.mean([p["total_ms"] for p in rows_v])),
                                      "kernel_ms": float(np.mean([p["posterior_ms"] for p in rows_v])), "result_identical": bool(same)}
        if args.sweep == "safeopt" and world == 1 and not args.no_extra and int(rows[-1].get("set_path", 0)) == 1:
            # column path: the expander chain runs on a second stream BESIDE the objective's posterior launch, so the time between the K1
            # stop event and the end of the sweep (`set_phase_ms` above) is only what sticks out behind K1 -- priced against the survey's
            # bytes it would exceed the copy rate.  The same sweep on ONE stream (option col_overlap = 0) gives the set phase's own
            # duration: that is the figure `hbm.frac` is quoted on; the overlapped one stays as `exposed`
            eng.set_option("col_overlap", 0)
            try:
                el_s, rows_s, res_s = timed_resident(eng, step, max(20, args.steps // 4), 3, barrier)
            finally:
                eng.set_option("col_overlap", 1)
            h1 = hbm_roofline(cfg["q"], es, n_local, rows_s)
            h = roof["hbm"]
            h["exposed"] = {"set_phase_ms": h["set_phase_ms"], "achieved": h["achieved"], "frac": h["frac"], "frac_vs_measured_copy": h["frac_vs_measured_copy"],
                            "definition": "device time between the K1 stop event and the end of the sweep with the expander chain overlapped (the default run)"}
            for k in ("set_phase_ms", "achieved", "frac", "frac_vs_measured_copy"):
                h[k] = h1[k]
            h["definition"] = ("SURVEY.md 8(d): (2 q s + 4) bytes per candidate / the set phase's own device time, measured with everything on one stream "
                               "(col_overlap = 0: K1 stop event -> end of the sweep); `exposed` = the same in the default, overlapped run")
            h["one_stream"] = {"ms_per_step": el_s * 1e3 / max(20, args.steps // 4), "device_ms_per_step": float(np.mean([p["total_ms"] for p in rows_s])),
                               "kernel_ms": float(np.mean([p["posterior_ms"] for p in rows_s])), "result_identical": bool(result_record(res_s, args.sweep) == result)}
        if comm is not None:
            out["comm"] = comm
        if d_rec is not None:
            out["extra"] = [d_rec]
        if extras:
            it = timed_iterations(eng, [alt["ds"], cfg["ds"]], cfg["dtype"], step, max(10, args.steps // 4), 4, barrier)
            it["value"] = n_total / (it["ms_per_step"] * 1e-3)
            it["unit"] = "candidates/s"
            out["iteration"] = it
            out["config"]["iteration_ms"] = it["ms_per_step"]        # (one iteration of the reference loop: new model + tables + sweep)
            if k1_kind == 4:
                roof["table_build_ms"] = it["table_build_ms"]
            if k1_kind in (4, 5) and default_run:
                out["table_kernel"] = table_kernel_record(eng, cfg, step, args.sweep, res, roof["kernel"], n_total, barrier, steps=3)
                roof["table_kernel_frac"] = out["table_kernel"]["frac"]
                roof["table_kernel_ms"] = out["table_kernel"]["kernel_ms"]
            out["config"]["iteration_value"] = it["value"]
        if extras and default_run:
            # every other BASELINE.json config on this one GPU: B (configs[1], with its K1g figure), C (the Williams-Otto plant: GoOSE
            # and SafeOpt), D (the whole 128^4 grid of the 8-GPU config: Chebyshev-node interpolation K1t, with its K1g figure), E (10^7 scattered fp32 points)
            out["extra"] = [extra_record(eng, *extra_cfgs["B"], "B", "safeopt", 100, barrier, table_kernel=True, lean=args.lean),
                            extra_record(eng, *extra_cfgs["C"], "C", "goose", 40, barrier),
                            extra_record(eng, *extra_cfgs["C"], "C", "safeopt", 40, barrier, lean=args.lean),
                            extra_record(eng, *extra_cfgs["D"], "D", "safeopt", 10, barrier, table_kernel=True, lean=args.lean),
                            extra_record(eng, extra_cfgs["E"][0], None, "E", "safeopt", 3, barrier, points=10_000_000)]
        if world == 1 and args.cpu_sample > 0 and not scattered:
            # the workload of `value` itself: a prefix of this config's grid (a bounded sample -- H whole would be minutes of CPU
            # work: n = 512 costs 16 x config B's flops per candidate), and beside it config B's whole grid as in earlier rounds
            whole = n_total <= 1 << 22
            out["cpu_baseline"] = cpu_baseline(cfg, count, min(args.cpu_sample, 1 << 18 if not whole else args.cpu_sample),
                                               prefix=None if whole else max(1 << 20, min(args.cpu_sample, 1 << 21)))
            out["cpu_baseline"]["config"] = f"config {args.config}: n={cfg['n']}, grid {'x'.join(map(str, count))}"
            if "B" in extra_cfgs and args.config != "B":
                bc = extra_cfgs["B"][0]
                out["cpu_baseline"]["config_B"] = cpu_baseline(bc, list(bc["count"]), 0)
                out["cpu_baseline"]["config_B"]["config"] = f"config B: n={bc['n']}, grid {'x'.join(map(str, bc['count']))}"
        # the standing audit of the guard band over everything this process swept (sbo_profile.guard_audit_*, csrc/guard.hip): value
        # pairs re-evaluated with the reference formula on a side stream, how many lay outside the band, the worst in units of the band
        # (flat copies of what lives in nested records: a reader of the top-level scalars alone can recompute the headline fractions)
        hb = roof["hbm"]
        roof["set_phase_ms"] = hb["set_phase_ms"]
        roof["set_phase_hbm_frac"] = hb["frac"]
        roof["set_phase_bytes_algorithmic"] = hb["bytes"]
        roof["set_phase_exposed_ms"] = hb.get("exposed", {}).get("set_phase_ms", hb["set_phase_ms"])
        eng.synchronize()
        pa = eng.profile()
        out["config"]["guard_audit_samples"] = int(pa["guard_audit_samples"])
        out["config"]["guard_audit_violations"] = int(pa["guard_audit_violations"])
        out["config"]["guard_audit"] = {"samples": int(pa["guard_audit_samples"]), "violations": int(pa["guard_audit_violations"]),
                                        "worst_over_band": float(pa["guard_audit_worst"])}
        print(json.dumps(out))
    if group is not None:
        group.barrier()
        group.destroy()
    eng.close()


if __name__ == "__main__":
    main()
