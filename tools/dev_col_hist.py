#!/usr/bin/env python3
"""development: wave-lifetime histograms of the column path's kernels (run with SBO_COL_DBG=9 [SBO_COL_OVERLAP=0])"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import safebo_amd
from safebo_amd import synthetic
lib = C.CDLL(os.path.join(ROOT, "safe-bayesian-optimization_amd", "libsafebo.so"))
fn = lib.sbo_debug_col_hist
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_uint64), C.c_int]
buf = (C.c_uint64 * (6 * 64))()
name = sys.argv[1] if len(sys.argv) > 1 else "H"
cfg = synthetic.make_config(name)
eng = safebo_amd.SweepEngine(0)
eng.set_model(cfg["ds"], dtype="f64")
eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], list(cfg["count"]))
for _ in range(4):
    eng.sweep_safeopt(cfg["b"])
eng.synchronize()
fn(buf, 1)
steps = 20
for _ in range(steps):
    eng.sweep_safeopt(cfg["b"])
eng.synchronize()
fn(buf, 0)
for k, nm in enumerate(["k_col_a", "k_col_cs", "k_col_decide", "k_col_scan", "k_col_min (M)", "k_col_min (G)"]):
    h = [buf[k * 64 + i] / steps for i in range(64)]
    tot = sum(h)
    print(f"{nm}: {tot:.0f} waves per sweep; lifetime us: " + " ".join(f"{i}:{v:.0f}" for i, v in enumerate(h) if v >= 0.5))
