"""Developer driver: PCIe-inclusive rates of the host-buffer entry points."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd import synthetic
import oracle
eng = safebo_amd.SweepEngine(0)
cfg = synthetic.make_config("B")
eng.set_model(cfg["ds"])
eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"])
N = 2048 * 2048
eng.posterior(); 
t = time.perf_counter(); m, v = eng.posterior(); dt = time.perf_counter() - t
print(f"posterior() incl. download of mean/var [N,2] f64 (134 MB): {dt*1e3:.1f} ms -> {N/dt:.3e} candidates/s")
t = time.perf_counter(); l = eng.bounds(3.0, 1, "lcb"); dt = time.perf_counter() - t
print(f"bounds(lcb_1) incl. download (34 MB): {dt*1e3:.1f} ms -> {N/dt:.3e} candidates/s")
pts = oracle.grid_points(cfg["bound"][:, 0], cfg["bound"][:, 1], [1024, 1024])
t = time.perf_counter(); eng.set_points(pts); dt = time.perf_counter() - t
print(f"set_points 1M x 2 f64 (16.8 MB upload): {dt*1e3:.1f} ms -> {pts.shape[0]/dt:.3e} candidates/s")
t = time.perf_counter(); r = eng.sweep_safeopt(3.0); dt = time.perf_counter() - t
print(f"sweep on the uploaded list (1M, generic kernel): {dt*1e3:.1f} ms")
t = time.perf_counter(); S = eng.mask("S"); dt = time.perf_counter() - t
print(f"mask download 1 MB: {dt*1e3:.2f} ms")
