"""Developer probe (CPU, NumPy oracle only): degree of the 2-D Chebyshev expansion the quadratic form k^T invK k needs over the
reference's hyper-parameter box -- the numerical basis of the Chebyshev core of K1b."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from safebo_amd import synthetic
def run(name, hyp, Ds):
    cfg = synthetic.make_config(name)
    d = 2; q = cfg["q"]
    h = cfg["ds"]["hypopt"].copy()
    h[:d, :] = hyp[1]; h[d, :] = hyp[2]; h[d + 1, :] = hyp[0]
    ds = oracle.make_inference_dataset(cfg["X"], cfg["Y"], h)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    rng = np.random.default_rng(1)
    tp = rng.uniform(lo, hi, size=(3000, 2))
    m2, v2 = oracle.gp_inference(tp, ds)
    o = 1
    sf2 = np.exp(2 * h[d, o])
    out = []
    for D in Ds:
        k = np.arange(D)
        xi = np.cos(np.pi * (k + 0.5) / D)
        x0 = 0.5 * (xi * (hi[0] - lo[0]) + (hi[0] + lo[0])); x1 = 0.5 * (xi * (hi[1] - lo[1]) + (hi[1] + lo[1]))
        X0, X1 = np.meshgrid(x0, x1, indexing="ij")
        pts = np.stack([X0.ravel(), X1.ravel()], 1)
        # unclipped quad: recompute via formula to avoid the clip at 0
        xn = (pts - ds["X_mean"]) / ds["X_std"]
        ell = np.exp(2 * h[:d, o])
        kk = oracle.calc_cov_mat(ds["X_norm"], xn, ell, sf2)        # [n, N]
        quad = np.einsum("jn,jk,kn->n", kk, ds["invKopt"][o], kk)
        F = quad.reshape(D, D)
        T = np.cos(np.outer(k, np.pi * (k + 0.5) / D))
        w = np.full(D, 2.0); w[0] = 1.0
        C = (w[:, None] * (T @ F @ T.T) * w[None, :]) / D ** 2
        t0 = (2 * tp[:, 0] - (hi[0] + lo[0])) / (hi[0] - lo[0]); t1 = (2 * tp[:, 1] - (hi[1] + lo[1])) / (hi[1] - lo[1])
        V0 = np.cos(np.outer(np.arccos(t0), k)); V1 = np.cos(np.outer(np.arccos(t1), k))
        q_ = np.einsum("na,ab,nb->n", V0, C, V1)
        vv = np.maximum(0.0, sf2 - q_)
        err = np.max(np.abs(vv - v2[:, o] / ds["Y_std"][o] ** 2))
        tail = max(np.max(np.abs(C[-2:, :])), np.max(np.abs(C[:, -2:]))) / np.max(np.abs(C))
        out.append(f"D{D}: {err:.1e} (tail {tail:.0e})")
    print(name, hyp, " | ".join(out), flush=True)
for name in ("B",):
    run(name, (-2.0, -0.5, 0.0), (24, 32, 40, 48, 64))
    run(name, (-5.0, -0.5, 0.0), (32, 48, 64, 96))
    run(name, (-5.0, 1.5, 0.0), (16, 24, 32, 48))
    run(name, (-5.0, -1.5, 0.0), (64, 96, 128, 160, 192))
    run(name, (-5.0, 0.5, 1.5), (24, 32, 48, 64))
    run(name, (-2.0, -1.5, 0.0), (64, 96, 128, 160))
