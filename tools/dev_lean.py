#!/usr/bin/env python3
"""development: device time of resident-model sweeps, plain vs lean: tools/dev_lean.py CONFIG"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import safebo_amd
from safebo_amd import synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "H"
cfg = synthetic.make_config(name)
eng = safebo_amd.SweepEngine(0)
eng.set_model(cfg["ds"], dtype="f64")
eng.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], list(cfg["count"]))
audits = [int(a) for a in os.environ.get("AUDIT", "4096").split(",")]
for audit, lean in [(a, l) for a in audits for l in (0, 1, 2, 0, 1, 2)]:
    eng.set_option("guard_audit", audit)
    eng.set_option("guard_audit_every", int(os.environ.get("EVERY", "16")))
    for _ in range(5):
        eng.sweep_safeopt(cfg["b"], lean=lean)
    tot, k1, sp = [], [], []
    t0 = time.perf_counter()
    for _ in range(100):
        eng.sweep_safeopt(cfg["b"], lean=lean)
        p = eng.profile()
        tot.append(p["total_ms"]); k1.append(p["posterior_ms"]); sp.append(p["set_phase_ms"])
    wall = (time.perf_counter() - t0) / 100 * 1e3
    print(f"{name} audit={audit} ({p['guard_audit_samples']} pairs, {p['guard_audit_violations']} violations, worst {p['guard_audit_worst']:.3g}) lean={lean}: wall {wall:.4f} ms/sweep, device {np.mean(tot):.4f}, K1 {np.mean(k1):.4f}, set phase {np.mean(sp):.4f}, path {p['set_path']}")
