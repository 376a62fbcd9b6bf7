#!/usr/bin/env python3
"""development: where the wall time of a resident-model sweep goes beyond its device time: tools/dev_host_overhead.py CONFIG"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import safebo_amd
from safebo_amd import synthetic, _lib as L
name = sys.argv[1] if len(sys.argv) > 1 else "B"
cfg = synthetic.make_config(name)
eng = safebo_amd.SweepEngine(0)
eng.set_model(cfg["ds"], dtype="f64")
eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], list(cfg["count"]))
for _ in range(20):
    eng.sweep_safeopt(cfg["b"], lean=1)
N = 400
t0 = time.perf_counter()
for _ in range(N):
    eng.sweep_safeopt(cfg["b"], lean=1)
t1 = time.perf_counter()
res = L.SafeOptResult()
opts = eng._opts(cfg["b"], True, False, False)
opts.lean = 1
fn = eng._lib.sbo_sweep_safeopt
ctx = eng._ctx
t2 = time.perf_counter()
for _ in range(N):
    fn(ctx, C.byref(opts), C.byref(res))
t3 = time.perf_counter()
dev = []
for _ in range(50):
    fn(ctx, C.byref(opts), C.byref(res))
    dev.append(eng.profile()["total_ms"])
print(f"{name}: python method {1e3 * (t1 - t0) / N:.4f} ms / sweep, raw C call {1e3 * (t3 - t2) / N:.4f}, device {np.mean(dev):.4f}")
for opt in ("spin_wait",):
    for v in (0, 1):
        try:
            eng.set_option(opt, v)
        except Exception as e:
            print("option", opt, e); break
        t2 = time.perf_counter()
        for _ in range(N):
            fn(ctx, C.byref(opts), C.byref(res))
        t3 = time.perf_counter()
        print(f"   {opt}={v}: raw C call {1e3 * (t3 - t2) / N:.4f}")
