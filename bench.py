#!/usr/bin/env python3
"""bench.py -- candidate-points/sec of the SafeOpt posterior + safe-set sweep on MI355X.

Contract (see the task statement): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 the
driver launches one rank per GPU with ``torch.distributed.run``.  One *step* is one full SafeOpt sweep
(K1 posterior, S/U/M masks, u*, Lipschitz bound, expander sets, masked arg-max, result to host) over the
candidates resident in HBM.

Workload at N = 1: BASELINE.json configs[1] -- Benoit 2-D, 2048 x 2048 implicit grid, n = 128 observations,
q = 2 outputs, fp64.  For N > 1 the grid grows along its slowest axis (2048 x 2048 N): each rank sweeps
2048^2 candidates ("weak" scaling); the ranks exchange u*/L (one RCCL max all-reduce), the fully-unsafe
mask (one all-gather) and the arg-max candidates (one sum all-reduce).

The posterior (K1) of that workload runs as two dense fp64 GEMMs in a reduced basis of the separable RBF kernel (K1b,
bilinear.hip) -- inner dimension ~280 whatever n is -- instead of the O(n^2)-per-candidate triangular contraction the
algorithmic flop count of SURVEY.md 8(d) assumes.  ``roofline.achieved`` counts, per launch, min(algorithmic flops,
flops the matrix cores actually issued) / K1 time: wasted flops earn nothing (the contract's point) and neither do
flops a better algorithm no longer executes, so ``frac`` stays a hardware utilisation <= 1; ``roofline.algorithmic``
gives the literal SURVEY figure (which exceeds the peak with K1b), and ``table_kernel`` times the same sweep with the
O(n^2) kernel (K1g) in the same run.

torch is used only as the launcher's rendezvous (gloo group: unique-id broadcast, barriers, max over
ranks); device memory, streams and the collectives on the data path belong to libsafebo.so.
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
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")   # HBM bytes per K1 launch from rocprofv3 --pmc


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="B", help="synthetic config (B = BASELINE.json configs[1]; H = 4096^2, n = 512)")
    ap.add_argument("--n", type=int, default=None, help="override the number of observations")
    ap.add_argument("--cpu-sample", type=int, default=1 << 21, help="candidates timed by the CPU baseline (0 = skip)")
    ap.add_argument("--points", type=int, default=10_000_000, help="candidates per rank for the scattered config E")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: the grid's slowest axis grows with the ranks (default); strong: fixed grid, sharded")
    ap.add_argument("--posterior", choices=["auto", "table"], default="auto",
                    help="auto: fp64 2-D grids use the bilinear GEMM posterior (K1b) when its bases qualify; "
                         "table: force the separable-table kernel (K1g), the O(n^2)-per-candidate contraction")
    return ap.parse_args()


def cpu_baseline(cfg, count, sample):
    """The NumPy oracle (reference formulation: explicit invK, (Nc x n) @ (n x n), row-dot) on a contiguous
    prefix of the same grid: posterior, bounds, S/U/M masks, u*, arg-max.  The oracle's brute-force
    expander is quadratic in the candidate count and is left out (that favours the CPU figure)."""
    import oracle
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    total = int(np.prod(count))
    sample = min(sample, total)
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
    threads = len(os.sched_getaffinity(0))
    try:
        from threadpoolctl import threadpool_info
        blas = [p["num_threads"] for p in threadpool_info() if p.get("user_api") == "blas"]
        threads = max(blas) if blas else threads
    except Exception:
        pass
    return {"value": sample / dt, "unit": "candidates/s", "cores": threads, "kind": "port",
            "sample": f"first {sample} candidates of the same grid: posterior + bounds + S/U/M masks + u* + arg-max "
                      f"in NumPy ({threads} BLAS threads, {len(os.sched_getaffinity(0))} cores visible), {dt:.2f} s; "
                      f"the oracle's quadratic brute-force expander is excluded"}


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

    dist = None
    if world > 1:
        from safebo_amd import distributed
        dist = distributed.init_gloo_from_env()   # rendezvous only (gloo over 127.0.0.1)

    cfg = synthetic.make_config(args.config, n=args.n)
    scattered = cfg["count"] is None                      # config E: explicit list of scattered candidates
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
        # every rank to one card, which RCCL refuses) may fall back to the gloo relay; ranks on distinct devices that
        # cannot form the communicator fail the run.
        transport = distributed.join_with_fallback(eng, allow_relay="SBO_BENCH_DEVICE" in os.environ)
    eng.set_model(cfg["ds"], dtype=cfg["dtype"], use_invK=(cfg["dtype"] == "f64"))
    if scattered:
        pts = synthetic.scattered_points(cfg, n_total)[rank * per_rank:(rank + 1) * per_rank]
        eng.set_points(pts, first=rank * per_rank)       # uploaded to HBM before the timed region
    else:
        eng.set_grid_sharded(lo, hi, count)              # candidates are implicit: resident by construction

    def barrier():
        eng.synchronize()
        if dist is not None:
            dist.barrier()
        eng.synchronize()

    def step():
        return eng.sweep_safeopt(cfg["b"])

    if args.posterior == "table":
        eng.set_option("bilinear", 0)
    for _ in range(args.warmup):
        step()
    setup_ms = eng.profile()["posterior_setup_ms"]      # K1b: host build of the per-(model, grid) tables, paid in the warm-up
    k1_ms, k1_flops, k1_exec, tot_ms = [], [], [], []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        p = eng.profile()
        k1_ms.append(p["posterior_ms"])
        k1_flops.append(p["posterior_flops"])
        k1_exec.append(p["posterior_executed_flops"])
        tot_ms.append(p["total_ms"])
    k1_kind = p["posterior_kernel"]
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = n_total * args.steps / elapsed
        k1 = float(np.mean(k1_ms))
        alg_tflops = float(np.mean(k1_flops)) / (k1 * 1e-3) / 1e12
        peak = FP64_MATRIX_PEAK_TFLOPS if cfg["dtype"] == "f64" else FP32_MATRIX_PEAK_TFLOPS
        kname = {1: "k_posterior", 2: "k_posterior_chunked", 3: "k_posterior_grid", 4: "k_bpost (+ k_bstage1, stage 1)"}.get(k1_kind, "?")
        executed = float(np.mean(k1_exec)) / (k1 * 1e-3) / 1e12
        achieved = min(alg_tflops, executed) if executed > 0 else alg_tflops
        traffic = None
        if os.path.exists(PMC_TRAFFIC_FILE):       # collected by tools/gpu_bench_profile.sh in separate --pmc passes
            key = f"{args.config}:n={cfg['ds']['X_norm'].shape[0]}" + (":K1b" if k1_kind == 4 else "")
            rec = json.load(open(PMC_TRAFFIC_FILE)).get(key)
            traffic = rec["hbm_bytes_per_launch"] if rec else None
        out = {
            "metric": "candidate-points/sec, SafeOpt posterior+safe-set sweep",
            "value": value, "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": cfg["dtype"], "data": "synthetic",
            "config": {"workload": f"config {args.config}: {cfg['plant']} {cfg['d']}-D SafeOpt sweep, "
                                   + (f"explicit list of {n_total} scattered candidates" if scattered else
                                      f"implicit grid {'x'.join(str(c) for c in count)} ({n_total} candidates)")
                                   + f", n={cfg['ds']['X_norm'].shape[0]} observations, q={cfg['q']} outputs, b={cfg['b']}",
                       "per_gpu_candidates": n_total // world, "sweep": "safeopt", "collectives": transport,
                       "result": {"count_S": res["count_S"], "count_M": res["count_M"], "count_G": [int(x) for x in res["count_G"]],
                                  "minimizer_index": res["minimizer_index"], "exact_rechecks": res["n_exact_rechecks"]}},
            # achieved = min(ALGORITHMIC flops of SURVEY.md 8d, flops issued on the matrix cores) / K1 time.  With K1g the
            # first term binds (padding and the dense blocks of the triangle are not counted); with K1b the second one
            # does: its two GEMMs (inner dimension ~r(r+1)/2, independent of n) issue far fewer flops than the O(n^2)
            # count, and "algorithmic" below is that count over the same time -- above the peak by construction.
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "kernel": kname, "kernel_ms": k1,
                         "executed": {"achieved": executed, "frac": executed / peak,
                                      "flops_per_candidate": float(np.mean(k1_exec)) / (n_total // world)},
                         "algorithmic": {"achieved": alg_tflops, "frac": alg_tflops / peak,
                                         "flops_per_candidate": float(np.mean(k1_flops)) / (n_total // world),
                                         "definition": "SURVEY.md 8(d): q (n^2 + (2 d + 10) n) per candidate"},
                         "peak_source": "AMD MI355X datasheet FP64 matrix (no f64 row in MI355X_MICROARCH.md)" if cfg["dtype"] == "f64" else "MI355X_MICROARCH.md f32 MFMA",
                         "peak_measured_mfma_f64": FP64_MFMA_MEASURED_TFLOPS if cfg["dtype"] == "f64" else None,
                         "device_ms_per_step": float(np.mean(tot_ms))},
        }
        if k1_kind == 4:
            out["roofline"]["table_build_ms"] = setup_ms
        if world == 1 and k1_kind == 4:
            # the same sweep with the separable-table kernel (the O(n^2)-per-candidate contraction on MFMA), same run
            eng.set_option("bilinear", 0)
            step()
            t_ms, t_k1 = [], []
            for _ in range(3):
                step()
                pt = eng.profile()
                t_ms.append(pt["total_ms"])
                t_k1.append(pt["posterior_ms"])
            eng.set_option("bilinear", 1)
            tk1 = float(np.mean(t_k1))
            out["table_kernel"] = {"kernel": "k_posterior_grid", "device_ms_per_step": float(np.mean(t_ms)), "kernel_ms": tk1,
                                   "value": n_total / (float(np.mean(t_ms)) * 1e-3), "unit": "candidates/s (device time)",
                                   "achieved": float(np.mean(k1_flops)) / (tk1 * 1e-3) / 1e12, "frac": float(np.mean(k1_flops)) / (tk1 * 1e-3) / 1e12 / peak}
        if world == 1 and args.cpu_sample > 0 and not scattered:
            out["cpu_baseline"] = cpu_baseline(cfg, count, args.cpu_sample)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
