#!/usr/bin/env python3
"""Per-phase time of k_bpost from the diagnostic build (make -C safe-bayesian-optimization_amd/csrc phaseclk; run through gpurun as
  cp safe-bayesian-optimization_amd/libsafebo_phaseclk.so safe-bayesian-optimization_amd/libsafebo.so && python tools/dev_phase_clocks.py H B
on the box's scratch copy).  Prints, per config, the mean time a workgroup spends in: variance phase, mean phase (+ fused
classification), gradient phases, partial rows -- 100 MHz counter at the phase boundaries, summed over the workgroups of 50 sweeps."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import safebo_amd                                   # noqa: E402
from safebo_amd import synthetic                    # noqa: E402

lib = safebo_amd._lib.load()
clk = lib.sbo_debug_phase_clocks
clk.restype, clk.argtypes = C.c_int, [C.POINTER(C.c_uint64), C.c_int, C.c_int]
buf = (C.c_uint64 * 16)()
nofuse = "--nofuse" in sys.argv


def sweep(eng, b):
    try:
        eng.sweep_safeopt(b)
    except Exception as exc:          # (experiment builds that skip stores leave the set phase nothing sensible)
        sweep.err = exc


for name in [a for a in sys.argv[1:] if not a.startswith("--")] or ["H"]:
    cfg = synthetic.make_config(name)
    eng = safebo_amd.SweepEngine(0)
    eng.set_model(cfg["ds"], dtype="f64", use_invK=True)
    eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], list(cfg["count"]))
    if nofuse:
        eng.set_option("fuse_classify", 0)
    for _ in range(5):
        sweep(eng, cfg["b"])
    eng.synchronize()
    split = (cfg["count"][0] // 128) * (cfg["count"][1] // 64)          # rows of output 0 (the objective's tiles)
    assert clk(buf, 1, split) == 0
    steps, k1 = 50, []
    for _ in range(steps):
        sweep(eng, cfg["b"])
        k1.append(eng.profile()["posterior_ms"])
    eng.synchronize()
    assert clk(buf, 0, split) == 0
    both = np.array(list(buf), dtype=np.float64)
    names = ["variance phase (GEMM + epilogue)", "mean phase (+ fused classification on the constraint)", "gradient phases (tiles that run them)",
             "partial rows"]
    for out_i, v in ((0, both[:8]), (1, both[8:])):
        wgs = v[5]
        if wgs == 0:
            continue
        print(f"config {name}, output {out_i}: {int(wgs / steps)} workgroups per sweep, {v[4] / steps:.0f} gradient phases run per sweep, K1 {np.mean(k1) * 1e3:.1f} us")
        tot = v[:4].sum()
        for i, nm in enumerate(names):
            print(f"   {nm:58s} {v[i] / wgs * 1e-2:7.2f} us per workgroup   {100.0 * v[i] / tot:5.1f} % of a workgroup's life")
        print(f"   {'life of a workgroup':58s} {tot / wgs * 1e-2:7.2f} us")
    eng.close()
