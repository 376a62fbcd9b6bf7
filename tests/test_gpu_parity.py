"""GPU parity tests (run with -m gpu on an MI355X): the HIP path behind the C ABI against the NumPy oracle and
the committed golden fixtures.  Tolerances: posterior mean/variance within 1e-10 (fp64) / 1e-4 (fp32) in
normalised units (mean / max(1, Y_std), var / max(1, Y_std)^2 -- the reference's GP works on normalised outputs,
models/GP_Safe.py:92-94, 346-347); masks and arg-max indices bit-exact."""
import glob
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle
import safebo_amd
from safebo_amd import synthetic

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(p for p in glob.glob(os.path.join(HERE, "golden", "*.npz")) if "contour_reference" not in p and "nll_population" not in p)
TOL64, TOL32 = 1e-10, 1e-4


def _nerr(got, ref, ystd, power):
    return float(np.max(np.abs(got - ref) / np.maximum(1.0, ystd) ** power))


def _check_posterior(eng, ds, pts, tol, dtype="f64"):
    mean, var = eng.posterior()
    om, ov = oracle.gp_inference(pts, ds)
    assert mean.shape == om.shape and var.shape == ov.shape
    assert mean.dtype == (np.float64 if dtype == "f64" else np.float32)
    em, ev = _nerr(mean, om, ds["Y_std"], 1), _nerr(var, ov, ds["Y_std"], 2)
    assert em < tol and ev < tol, (em, ev)
    return mean, var


# ---------------------------------------------------------------------------------------------- posterior
@pytest.mark.parametrize("cfg_name,n,count", [
    ("A", 20, [50, 50]), ("A", 4, [40, 40]), ("A", 2, [9, 7]), ("B", 128, [96, 64]), ("B", 100, [70, 33]),
    ("B", 17, [33, 5]), ("C", 256, [64, 48]), ("H", 512, [64, 32]), ("H", 300, [31, 17]), ("D", 128, [9, 8, 7, 6]),
])
@pytest.mark.parametrize("use_invK", [True, False])
def test_posterior_fp64_grid(engine, cfg_name, n, count, use_invK):
    cfg = synthetic.make_config(cfg_name, n=n)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    engine.set_model(cfg["ds"], dtype="f64", use_invK=use_invK)
    engine.set_grid(lo, hi, count)
    _check_posterior(engine, cfg["ds"], oracle.grid_points(lo, hi, count), TOL64)


@pytest.mark.parametrize("d,q,n", [(1, 1, 9), (3, 2, 40), (5, 3, 64), (6, 1, 200), (8, 2, 33)])
def test_posterior_fp64_scattered_points_any_dimension(engine, d, q, n):
    rng = np.random.default_rng(100 + d)
    X = rng.uniform(-1, 1, size=(n, d))
    Y = np.stack([np.sin(X.sum(1) * (i + 1)) + 0.5 * X[:, 0] for i in range(q)], axis=1)
    hyp = synthetic.default_hypopt(d, q) + rng.uniform(-0.2, 0.2, size=(d + 2, q)) * np.r_[np.ones(d + 1), 0][:, None]
    ds = synthetic.make_dataset(X, Y, hyp)
    pts = rng.uniform(-1.3, 1.3, size=(777, d))
    engine.set_model(ds, dtype="f64")
    engine.set_points(pts)
    _check_posterior(engine, ds, pts, TOL64)


@pytest.mark.parametrize("cfg_name,n,dtype", [("B", 128, "f64"), ("H", 300, "f64"), ("H", 512, "f64"), ("C", 200, "f64"),
                                               ("B", 128, "f32"), ("H", 512, "f32")])
def test_posterior_chunked_generic_kernel(engine, cfg_name, n, dtype):
    """The chunked generic kernel (row/column chunks, K* regenerated per chunk pair) forced on small models."""
    cfg = synthetic.make_config(cfg_name, n=n)
    pts = np.random.default_rng(n).uniform(cfg["bound"][:, 0], cfg["bound"][:, 1], size=(1500, 2))
    engine.set_option("posterior_path", 2)
    try:
        engine.set_model(cfg["ds"], dtype=dtype, use_invK=(dtype == "f64"))
        engine.set_points(pts)
        _check_posterior(engine, cfg["ds"], pts, TOL64 if dtype == "f64" else TOL32, dtype=dtype)
        if dtype == "f64":
            ref = oracle.safeopt_sweep(pts, cfg["ds"], cfg["b"])
            res = engine.sweep_safeopt(cfg["b"], want_masks=True, posterior_ready=True)
            assert np.array_equal(engine.mask("S"), ref["S"]) and res["minimizer_index"] == ref["minimizer_index"]
            assert np.allclose(res["L"], ref["L"], rtol=1e-9)
    finally:
        engine.set_option("posterior_path", 0)


# K1b: on fp64 2-D grids the posterior runs as two GEMMs in a reduced basis (bilinear.hip) when the axis bases qualify
@pytest.mark.parametrize("cfg_name,n,count", [("B", 128, [160, 96]), ("C", 64, [96, 130]), ("H", 300, [64, 72]), ("A", 64, [70, 65])])
def test_bilinear_posterior_matches_oracle_and_table_kernel(engine, cfg_name, n, count):
    cfg = synthetic.make_config(cfg_name, n=n)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    pts = oracle.grid_points(lo, hi, count)
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, count)
    try:
        engine.set_option("bilinear", 0)
        engine.posterior_run()
        assert engine.profile()["posterior_kernel"] == 3
        m_tab, v_tab = engine.posterior()
        engine.set_option("bilinear", 2)              # K1b's own plan for the first evaluation (1: node interpolation K1i first)
        engine.posterior_run()
        prof = engine.profile()
        assert prof["posterior_kernel"] == 4 and 0 < prof["posterior_executed_flops"] < prof["posterior_flops"] * 4
        mean, var = _check_posterior(engine, cfg["ds"], pts, TOL64)
    finally:
        engine.set_option("bilinear", 1)
    ystd = np.maximum(1.0, cfg["ds"]["Y_std"])
    assert np.max(np.abs(mean - m_tab) / ystd) < 1e-11 and np.max(np.abs(var - v_tab) / ystd ** 2) < 1e-11
    # the Lipschitz keys (max |grad mean|) come out of different kernels: same values to rounding
    r1 = engine.sweep_safeopt(cfg["b"], posterior_ready=True) if cfg_name != "H" else None
    if r1 is not None:
        engine.set_option("bilinear", 0)
        try:
            r0 = engine.sweep_safeopt(cfg["b"])
        finally:
            engine.set_option("bilinear", 1)
        assert np.allclose(r1["L"], r0["L"], rtol=1e-10)
        assert r1["minimizer_index"] == r0["minimizer_index"] and r1["count_S"] == r0["count_S"]


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_posterior_over_the_hyper_parameter_range(engine, seed):
    """Random hyper-parameters inside the reference's search box (models/GP_Safe.py:205-206: log length-scales and log
    signal std in [-1.5, 1.5]; the noise kept at log std >= -3 so that cond(K) stays below ~1e6 and the 1e-10 bar is about
    the kernels, not about the conditioning of the reference formula), random n and grid: whichever posterior kernel the
    library picks must match the oracle, and the two grid kernels must agree with each other."""
    rng = np.random.default_rng(900 + seed)
    n = int(rng.integers(30, 300))
    cfg = synthetic.make_config("B", n=n, seed=1000 + seed)
    hyp = np.empty((4, 2))
    hyp[:2] = rng.uniform(-1.0, 1.5, size=(2, 2))
    hyp[2] = rng.uniform(-1.0, 1.0, size=2)
    hyp[3] = rng.uniform(-3.0, -2.0, size=2)
    ds = synthetic.make_dataset(cfg["X"], cfg["Y"], hyp)
    count = [int(rng.integers(70, 150)), int(rng.integers(70, 150))]
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    pts = oracle.grid_points(lo, hi, count)
    engine.set_model(ds)
    engine.set_grid(lo, hi, count)
    mean, var = _check_posterior(engine, ds, pts, TOL64)
    kernel = engine.profile()["posterior_kernel"]
    assert kernel in (3, 4, 6)
    if kernel in (4, 6):
        engine.set_option("bilinear", 0)
        try:
            engine.posterior_run()
            m_t, v_t = engine.posterior()
        finally:
            engine.set_option("bilinear", 1)
        ystd = np.maximum(1.0, ds["Y_std"])
        assert np.max(np.abs(mean - m_t) / ystd) < TOL64 and np.max(np.abs(var - v_t) / ystd ** 2) < TOL64


def _tensor_model(d, n, log_ell, seed=7):
    """A smooth problem on [-2, 2]^d (objective + one constraint, safe near the origin; a second constraint and per-axis /
    per-output length scales when ``log_ell`` is a [d, 3] table) with fixed hyper-parameters."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(-2.0, 2.0, size=(n, d))
    cols = [np.sum(X ** 2, axis=1) + np.sin(2.0 * X[:, 0]), 3.0 - 0.5 * np.sum(X ** 2, axis=1) + X[:, 1]]
    if np.ndim(log_ell) == 2:
        cols.append(2.5 - 0.4 * np.sum((X - 0.3) ** 2, axis=1))
        hyp = synthetic.default_hypopt(d, 3)
        hyp[:d, :] = np.asarray(log_ell, dtype=np.float64)
        return synthetic.make_dataset(X, np.stack(cols, axis=1), hyp)
    return synthetic.make_dataset(X, np.stack(cols, axis=1), synthetic.default_hypopt(d, 2, log_ell=log_ell))


@pytest.mark.parametrize("d,count,n,log_ell,kernel", [
    (4, [64, 64, 64, 64], 128, -0.5, 5),        # BASELINE configs[3] hyper-parameters: 48 nodes per axis
    (3, [160, 168, 160], 96, -0.5, 5),          # three axes, ragged tiles (168 = 128 + 40)
    (3, [161, 165, 163], 64, -0.5, 5),          # odd counts: scalar stores, ragged everything
    (3, [176, 160, 152], 64, 0.3, 5),           # long length scale: 32 nodes
    (3, [160, 160, 168], 64, -1.3, 3),          # short length scale: more than 64 nodes on the first axes -- declined, K1g runs
    (3, [136, 176, 192], 80, [[-0.5, 0.3, -0.1], [0.4, -0.2, 0.2], [-0.8, 0.1, -0.6]], 5),   # three outputs, a length scale per axis and output (48 x 48 x 64 nodes)
])
def test_tensor_interpolation_equals_the_grid_kernel(engine, d, count, n, log_ell, kernel):
    """K1t (option tensor_cheb, default on): fp64 grids of three / four axes take the exact posterior at a tensor grid of
    Chebyshev nodes (K1g with explicit axis positions) and interpolate it to the candidates on the matrix cores.  Same function
    as K1g on the whole grid: posterior equal to rounding, Lipschitz constants too, sweep results and masks identical, and a
    sample of the grid within the bar of the NumPy oracle."""
    ds = _tensor_model(d, n, log_ell)
    lo, hi = np.full(d, -2.0), np.full(d, 2.0)
    engine.set_model(ds)
    out = {}
    try:
        for opt in (0, 1):
            engine.set_option("tensor_cheb", opt)
            engine.set_grid(lo, hi, count)
            mean, var = engine.posterior()
            kern = engine.profile()["posterior_kernel"]
            res = engine.sweep_safeopt(2.0, want_masks=True)
            masks = {k: engine.mask(k) for k in ("S", "U", "M")}
            masks["G"] = engine.mask("G", 1)
            if ds["hypopt"].shape[1] == 3:
                masks["G2"] = engine.mask("G", 2)
            goose = engine.sweep_goose(2.0, want_masks=True, posterior_ready=True)       # (the other sweeps read the same posterior)
            masks["O"] = engine.mask("O", 1)
            tr = engine.sweep_tr(2.0, np.zeros(d), 1.25, posterior_ready=True)
            out[opt] = (mean, var, kern, res, masks, goose, tr)
    finally:
        engine.set_option("tensor_cheb", 1)
    assert out[0][2] == 3 and out[1][2] == kernel
    for which in (5, 6):
        for k, v in out[0][which].items():
            if isinstance(v, (int, bool, np.integer)) or (isinstance(v, np.ndarray) and v.dtype.kind in "iub"):
                assert np.array_equal(np.asarray(v), np.asarray(out[1][which][k])), (which, k)
    ys = np.maximum(1.0, ds["Y_std"])
    assert np.max(np.abs(out[1][0] - out[0][0]) / ys) < TOL64 and np.max(np.abs(out[1][1] - out[0][1]) / ys ** 2) < TOL64
    r0, r1 = out[0][3], out[1][3]
    for k in ("minimizer_index", "expander_index", "count_S", "count_U", "count_M", "choose_minimizer"):
        assert r0[k] == r1[k], k
    assert np.array_equal(r0["count_G"], r1["count_G"]) and np.array_equal(r0["expander_index_c"], r1["expander_index_c"])
    assert np.allclose(r0["L"], r1["L"], rtol=1e-11, atol=0.0) and abs(r0["u_star"] - r1["u_star"]) <= 1e-10 * ys[0]
    for k in out[0][4]:
        assert np.array_equal(out[0][4][k], out[1][4][k]), k
    # a sample of the grid against the oracle
    rng = np.random.default_rng(3)
    total = int(np.prod(count))
    idx = np.unique(np.concatenate([rng.integers(0, total, size=3000), [0, total - 1, count[0] - 1, total - count[0]]]))
    axes = oracle.grid_axes(lo, hi, count)
    sub = np.empty((idx.size, d))
    f = idx.copy()
    for a in range(d):
        sub[:, a] = axes[a][f % count[a]]
        f //= count[a]
    om, ov = oracle.gp_inference(sub, ds)
    assert _nerr(out[1][0][idx], om, ds["Y_std"], 1) < TOL64 and _nerr(out[1][1][idx], ov, ds["Y_std"], 2) < TOL64


@pytest.mark.parametrize("seed", [31, 32, 33, 34, 35, 36])
def test_tensor_interpolation_random_models(engine, seed):
    """Random data, sizes and hyper-parameters from the box of the reference's fit (length scales per axis and output, signal
    and noise levels), three-axis grids of 4-5 M candidates: whatever the plan decides -- node counts per axis, a second
    attempt, declining to K1g -- the posterior equals K1g's to rounding and the SafeOpt sweep (masks, counts, arg-max indices) is
    identical; a sample of the grid is within the bar of the NumPy oracle."""
    rng = np.random.default_rng(9000 + seed)
    d, n = 3, int(rng.integers(24, 160))
    X = rng.uniform(-2.0, 2.0, size=(n, d))
    w = rng.normal(size=(2, d))
    Y = np.stack([np.sum((X - 0.2 * w[0]) ** 2, axis=1) + np.sin(X @ w[0]), 2.5 + 0.5 * np.cos(X @ w[1]) - 0.45 * np.sum(X ** 2, axis=1)], axis=1)
    hyp = np.empty((d + 2, 2))
    hyp[:d] = rng.uniform(-0.9, 0.8, size=(d, 2))
    hyp[d] = rng.uniform(-0.5, 0.5, size=2)
    hyp[d + 1] = rng.uniform(-3.0, -1.5, size=2)
    ds = synthetic.make_dataset(X, Y, hyp)
    count = [int(c) for c in rng.integers(150, 176, size=3)]
    b = float(rng.uniform(1.0, 3.0))
    lo, hi = np.full(d, -2.0), np.full(d, 2.0)
    engine.set_model(ds)
    out = {}
    try:
        for opt in (0, 1):
            engine.set_option("tensor_cheb", opt)
            engine.set_grid(lo, hi, count)
            mean, var = engine.posterior()
            kern = engine.profile()["posterior_kernel"]
            try:
                res = engine.sweep_safeopt(b, want_masks=True, posterior_ready=True)
                masks = {k: engine.mask(k) for k in ("S", "U", "M")}
                masks["G"] = engine.mask("G", 1)
            except safebo_amd.EmptySafeSetError:
                res, masks = None, {}
            out[opt] = (mean, var, kern, res, masks)
    finally:
        engine.set_option("tensor_cheb", 1)
    assert out[0][2] == 3 and out[1][2] in (3, 5)
    ys = np.maximum(1.0, ds["Y_std"])
    assert np.max(np.abs(out[1][0] - out[0][0]) / ys) < TOL64 and np.max(np.abs(out[1][1] - out[0][1]) / ys ** 2) < TOL64
    assert (out[0][3] is None) == (out[1][3] is None)
    if out[0][3] is not None:
        for k in ("minimizer_index", "expander_index", "count_S", "count_U", "count_M", "choose_minimizer"):
            assert out[0][3][k] == out[1][3][k], k
        assert np.allclose(out[0][3]["L"], out[1][3]["L"], rtol=1e-10, atol=0.0)
        for k in out[0][4]:
            assert np.array_equal(out[0][4][k], out[1][4][k]), k
    total = int(np.prod(count))
    idx = np.unique(rng.integers(0, total, size=2000))
    axes = oracle.grid_axes(lo, hi, count)
    sub = np.empty((idx.size, d))
    f = idx.copy()
    for a in range(d):
        sub[:, a] = axes[a][f % count[a]]
        f //= count[a]
    om, ov = oracle.gp_inference(sub, ds)
    assert _nerr(out[1][0][idx], om, ds["Y_std"], 1) < TOL64 and _nerr(out[1][1][idx], ov, ds["Y_std"], 2) < TOL64


def test_tensor_interpolation_second_attempt_and_decline(engine):
    """The plan's accuracy probe at work (option tensor_guess_pct scales the first guess of the node counts): a guess that is
    too short fails the probe, the second attempt one ladder step up passes -- and the next model on the same grid starts
    there --; a guess far too short fails twice and the O(n^2) kernel runs.  The posterior is K1g's to rounding every time."""
    lo, hi, count = np.full(3, -2.0), np.full(3, 2.0), [160, 168, 160]
    engine.set_grid(lo, hi, count)
    try:
        for log_ell, cases in ((-0.5, ((100, 5), (78, 5), (30, 5))), (-0.75, ((100, 5), (30, 3)))):
            ds = _tensor_model(3, 64, log_ell)
            ys = np.maximum(1.0, ds["Y_std"])
            engine.set_model(ds)
            engine.set_option("tensor_cheb", 0)
            m0, v0 = engine.posterior()
            engine.set_option("tensor_cheb", 1)
            for pct, kernel in cases:
                engine.set_option("tensor_guess_pct", pct)
                for rep in range(2):                          # (the second model build on the grid starts from the sticky bump)
                    engine.set_model(ds)
                    m, v = engine.posterior()
                    assert engine.profile()["posterior_kernel"] == kernel, (log_ell, pct, rep)
                    assert np.max(np.abs(m - m0) / ys) < 2e-11 and np.max(np.abs(v - v0) / ys ** 2) < 2e-11, (log_ell, pct, rep)
    finally:
        engine.set_option("tensor_guess_pct", 100)
        engine.set_option("tensor_cheb", 1)


def test_tensor_interpolation_shards_reproduce_the_whole_grid_bitwise(engine):
    """A rank's shard (whole hyper-planes of the last axis) interpolates the same node values with its own rows of the last
    axis' matrix: the same sums, bit for bit."""
    ds = _tensor_model(3, 64, -0.5)
    lo, hi = np.full(3, -2.0), np.full(3, 2.0)
    count = [160, 168, 160]
    engine.set_model(ds)
    engine.set_grid(lo, hi, count)
    m, v = engine.posterior()
    assert engine.profile()["posterior_kernel"] == 5
    plane = count[0] * count[1]
    for first, nloc in [(0, 80 * plane), (80 * plane, 80 * plane), (37 * plane, 64 * plane)]:
        engine.set_grid(lo, hi, count, first=first, n_local=nloc)
        ms, vs = engine.posterior()
        assert engine.profile()["posterior_kernel"] == 5
        assert np.array_equal(ms, m[first:first + nloc]) and np.array_equal(vs, v[first:first + nloc])


def test_bilinear_rank_range_and_declines(engine):
    """Short length-scales need larger bases (r up to 64) and higher degrees: still the GEMM path when that is cheaper than
    the O(n^2) contraction, and still within tolerance -- with the Chebyshev core (inner dimension 2 rc - 1 <= 255 instead of up
    to 2080 pair products) that now includes n = 128 at log ell = -1.5 on a small grid.  Bases beyond 64 directions and fp32
    models stay on the separable-table kernel."""
    lo_count = [96, 80]
    for cfg_name, n, shift, kernel in (("H", 512, -1.0, 4), ("B", 128, -1.0, 4), ("B", 128, -2.2, 3)):
        cfg = synthetic.make_config(cfg_name, n=n)
        lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
        pts = oracle.grid_points(lo, hi, lo_count)
        hyp = synthetic.default_hypopt(2, 2)
        hyp[:2] += shift
        ds = synthetic.make_dataset(cfg["X"], cfg["Y"], hyp)
        engine.set_model(ds)
        engine.set_grid(lo, hi, lo_count)
        _check_posterior(engine, ds, pts, TOL64)
        assert engine.profile()["posterior_kernel"] == kernel, (cfg_name, shift)
    cfg = synthetic.make_config("B", n=128)
    engine.set_model(cfg["ds"], dtype="f32", use_invK=False)
    engine.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], lo_count)
    engine.posterior_run()
    assert engine.profile()["posterior_kernel"] == 3


def test_bilinear_shards_reproduce_the_whole_grid_bitwise(engine):
    """Line ranges of a grid (what a rank of a sharded sweep holds) use the bases of the whole axis: identical values."""
    cfg = synthetic.make_config("B", n=128)
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], [80, 100]
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, count)
    m, v = engine.posterior()
    assert engine.profile()["posterior_kernel"] in (4, 6)
    for first_line, lines in [(0, 16), (16, 48), (37, 63), (84, 16)]:
        engine.set_grid(lo, hi, count, first=first_line * 80, n_local=lines * 80)
        ms, vs = engine.posterior()
        assert engine.profile()["posterior_kernel"] in (4, 6)
        sl = slice(first_line * 80, (first_line + lines) * 80)
        assert np.array_equal(ms, m[sl]) and np.array_equal(vs, v[sl])


def test_posterior_fp32(engine):
    cfg = synthetic.make_config("B", n=128)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    pts = oracle.grid_points(lo, hi, [96, 64])
    engine.set_model(cfg["ds"], dtype="f32", use_invK=False)      # fp32: contract with L^-1 (no cancellation)
    engine.set_grid(lo, hi, [96, 64])
    _check_posterior(engine, cfg["ds"], pts, TOL32, dtype="f32")
    engine.set_points(pts.astype(np.float32))
    _check_posterior(engine, cfg["ds"], pts.astype(np.float32).astype(np.float64), TOL32, dtype="f32")


def test_posterior_fp32_large_n_scattered(engine):
    cfg = synthetic.make_config("E", n=2048)
    pts = synthetic.scattered_points(cfg, 4096)
    engine.set_model(cfg["ds"], dtype="f32", use_invK=False)
    engine.set_points(pts)
    _check_posterior(engine, cfg["ds"], pts.astype(np.float64), TOL32, dtype="f32")


def test_grid_paths_agree(engine):
    """Implicit grid through the separable-table kernel (K1g), the same grid through the generic exp() kernel (K1),
    and the grid handed over as explicit points: K1 on a grid and on the same points is bit-identical, K1g differs
    from both only by the rounding of exp(a)exp(b) vs exp(a+b)."""
    cfg = synthetic.make_config("B", n=128)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, [61, 47])
    m1, v1 = engine.posterior()
    engine.set_option("posterior_path", 1)
    try:
        engine.set_grid(lo, hi, [61, 47])
        m2, v2 = engine.posterior()
    finally:
        engine.set_option("posterior_path", 0)
    engine.set_points(oracle.grid_points(lo, hi, [61, 47]))
    m3, v3 = engine.posterior()
    assert np.array_equal(m2, m3) and np.array_equal(v2, v3)
    assert np.max(np.abs(m1 - m2)) < 1e-12 and np.max(np.abs(v1 - v2)) < 1e-12


def test_shard_ranges_reproduce_the_whole_grid_bitwise(engine):
    cfg = synthetic.make_config("B", n=128)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, [64, 50])
    m, v = engine.posterior()
    # whole-line shards run the same separable kernel: bit-identical; ragged shards fall back to the generic
    # kernel: equal to rounding
    for first, nloc in [(0, 64), (0, 640), (640, 2560), (3136, 64), (64 * 7, 64 * 3)]:
        engine.set_grid(lo, hi, [64, 50], first=first, n_local=nloc)
        ms, vs = engine.posterior()
        assert np.array_equal(ms, m[first:first + nloc]) and np.array_equal(vs, v[first:first + nloc])
    for first, nloc in [(0, 1), (0, 1000), (1000, 2200), (3199, 1), (37, 64)]:
        engine.set_grid(lo, hi, [64, 50], first=first, n_local=nloc)
        ms, vs = engine.posterior()
        assert np.max(np.abs(ms - m[first:first + nloc])) < 1e-12 and np.max(np.abs(vs - v[first:first + nloc])) < 1e-12
    engine.set_grid(lo, hi, [64, 50], first=5, n_local=0)       # empty shard
    ms, vs = engine.posterior()
    assert ms.shape == (0, 2)


def test_bounds_match_oracle(engine):
    cfg = synthetic.make_config("C", n=64)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    pts = oracle.grid_points(lo, hi, [40, 30])
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, [40, 30])
    mean, var = engine.posterior()
    lcb, ucb = oracle.bounds(mean, var, 2.0)          # same unfused arithmetic on the device values: bit-exact
    for i in range(3):
        assert np.array_equal(engine.bounds(2.0, i, "lcb"), lcb[:, i])
        assert np.array_equal(engine.bounds(2.0, i, "ucb"), ucb[:, i])
        assert np.array_equal(engine.bounds(2.0, i, "mean"), mean[:, i])
    # the plot mask of test/test_SafeOpt.py:337-338: vmap(lcb)(points, 1) > 0.
    om, ov = oracle.gp_inference(pts, cfg["ds"])
    assert np.array_equal(engine.bounds(2.0, 1, "lcb") > 0.0, oracle.bounds(om, ov, 2.0)[0][:, 1] > 0.0)


# ---------------------------------------------------------------------------------------------- error behaviour
def test_errors_follow_reference_style(engine):
    cfg = synthetic.make_config("A")
    with pytest.raises(ValueError, match="no kernel with name"):
        engine.set_model(cfg["ds"], kernel="Matern")            # models/GP_Safe.py:161-162
    engine.set_model(cfg["ds"])
    engine.set_points(np.zeros((5, 3)))
    with pytest.raises(ValueError, match="dimension should be same"):
        engine.posterior()                                      # models/GP_Safe.py:159-160
    bad = dict(cfg["ds"])
    bad["hypopt"] = np.zeros((3, 2))
    with pytest.raises(ValueError):
        engine.set_model(bad)
    with pytest.raises(ValueError):
        engine.set_grid([0, 0], [1, 1], [0, 4])
    fresh = safebo_amd.SweepEngine(0)
    with pytest.raises(safebo_amd.SafeBOError):
        fresh.posterior_run()                                   # no model yet
    fresh.close()


def test_empty_safe_set_is_reported(engine):
    cfg = synthetic.make_config("D", n=40)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    pts = oracle.grid_points(lo, hi, [6, 5, 4, 3])
    assert oracle.safeopt_sweep(pts, cfg["ds"], 3.0)["empty_safe_set"]
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, [6, 5, 4, 3])
    with pytest.raises(safebo_amd.EmptySafeSetError):
        engine.sweep_safeopt(3.0)


# ---------------------------------------------------------------------------------------------- sweeps
def _check_safeopt(eng, ref, q):
    for k in ("S", "U", "M"):
        assert np.array_equal(eng.mask(k), ref[k]), k
    for c in range(1, q):
        assert np.array_equal(eng.mask("G", c), ref["G"][c - 1]), f"G{c}"


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_safeopt_sweep_against_golden(engine, path):
    z = np.load(path)
    ds = synthetic.make_dataset(z["X"], z["Y"], z["hypopt"])
    lo, hi, count = z["bound"][:, 0], z["bound"][:, 1], [int(c) for c in z["count"]]
    q = z["Y"].shape[1]
    engine.set_model(ds)
    engine.set_grid(lo, hi, count)
    mean, var = engine.posterior()
    assert _nerr(mean, z["mean"], ds["Y_std"], 1) < TOL64 and _nerr(var, z["var"], ds["Y_std"], 2) < TOL64
    res = engine.sweep_safeopt(float(z["b"]), quirk_L_index=bool(z["quirk"]), want_masks=True, posterior_ready=True)
    _check_safeopt(engine, {k: z[k] for k in ("S", "U", "M", "G")}, q)
    assert res["minimizer_index"] == int(z["minimizer_index"])
    assert abs(res["minimizer_std"] - float(z["minimizer_std"])) < 1e-9 * max(1.0, float(z["minimizer_std"]))
    assert np.array_equal(res["expander_index_c"], z["expander_index"])
    assert res["expander_best_c"] == int(z["expander_best"]) and res["choose_minimizer"] == bool(z["choose_minimizer"])
    assert abs(res["u_star"] - float(z["u_star"])) < 1e-9 * max(1.0, abs(float(z["u_star"])))
    assert np.allclose(res["L"], z["L"], rtol=1e-9)
    assert (res["count_S"], res["count_U"], res["count_M"]) == (z["S"].sum(), z["U"].sum(), z["M"].sum())
    assert np.array_equal(res["count_G"], z["G"].sum(1))
    pts = oracle.grid_points(lo, hi, count)
    assert np.array_equal(res["minimizer_x"], pts[res["minimizer_index"]])


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_goose_sweep_against_golden(engine, path):
    """GoOSE iteration (models/GoOSE.py:63-119): pessimistic safe minimum, optimistic sets O_c, target, explore point."""
    z = np.load(path)
    ds = synthetic.make_dataset(z["X"], z["Y"], z["hypopt"])
    lo, hi, count = z["bound"][:, 0], z["bound"][:, 1], [int(c) for c in z["count"]]
    q = z["Y"].shape[1]
    engine.set_model(ds)
    engine.set_grid(lo, hi, count)
    res = engine.sweep_goose(float(z["b"]), quirk_L_index=bool(z["quirk"]), want_masks=True)
    assert np.array_equal(engine.mask("S"), z["S"]) and np.array_equal(engine.mask("U"), z["U"])
    for c in range(1, q):
        assert np.array_equal(engine.mask("O", c), z["O"][c - 1]), f"O{c}"
    assert res["safe_min_index"] == int(z["safe_min_index"])
    assert abs(res["safe_min_lcb"] - float(z["safe_min_lcb"])) < 1e-9 * max(1.0, abs(float(z["safe_min_lcb"])))
    assert np.array_equal(res["target_index_c"], z["target_index_c"])
    assert res["target_index"] == int(z["target_index"]) and res["target_best_c"] == int(z["target_best"])
    assert res["explore_index"] == int(z["explore_index"]) and res["choose_safe_min"] == bool(z["choose_safe_min"])
    assert np.array_equal(res["count_O"], z["O"].sum(1))
    if res["target_index"] >= 0:
        pts = oracle.grid_points(lo, hi, count)
        assert np.array_equal(res["target_x"], pts[res["target_index"]])
        assert np.array_equal(res["explore_x"], pts[res["explore_index"]])


def test_goose_sweep_explicit_points(engine):
    cfg = synthetic.make_config("C", n=64)
    pts = np.random.default_rng(9).uniform(cfg["bound"][:, 0], cfg["bound"][:, 1], size=(3000, 2))
    ref = oracle.goose_sweep(pts, cfg["ds"], cfg["b"])
    engine.set_model(cfg["ds"])
    engine.set_points(pts)
    res = engine.sweep_goose(cfg["b"], want_masks=True)
    for c in (1, 2):
        assert np.array_equal(engine.mask("O", c), ref["O"][c - 1])
    assert res["safe_min_index"] == ref["safe_min_index"] and res["target_index"] == ref["target_index"]
    assert res["explore_index"] == ref["explore_index"]


def test_large_explicit_list_expander_equals_the_grid_transform(engine):
    """Explicit lists above 131 072 points (round 1's cap) take the exhaustive expander evaluation too (cap now 2 M, cost
    quadratic): a 640 x 480 grid handed over point by point must give the masks and indices of the same grid swept through
    the distance transform."""
    import time
    cfg = synthetic.make_config("B", n=48)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    count = [640, 480]
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, count)
    ref = engine.sweep_safeopt(cfg["b"], want_masks=True)
    rm = {k: engine.mask(k) for k in ("S", "U", "M")}
    rm["G"] = engine.mask("G", 1)
    engine.set_points(oracle.grid_points(lo, hi, count))
    t0 = time.perf_counter()
    res = engine.sweep_safeopt(cfg["b"], want_masks=True)
    dt = time.perf_counter() - t0
    for k in ("S", "U", "M"):
        assert np.array_equal(engine.mask(k), rm[k]), k
    assert np.array_equal(engine.mask("G", 1), rm["G"]) and rm["G"].any()
    for k in ("minimizer_index", "expander_index", "count_S", "count_M"):
        assert res[k] == ref[k], k
    assert list(res["count_G"]) == list(ref["count_G"])
    assert dt < 30.0


def test_safeopt_sweep_explicit_points_exhaustive_expander(engine):
    cfg = synthetic.make_config("A")
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    pts = oracle.grid_points(lo, hi, [50, 50])
    ref = oracle.safeopt_sweep(pts, cfg["ds"], cfg["b"])
    engine.set_model(cfg["ds"])
    engine.set_points(pts)
    res = engine.sweep_safeopt(cfg["b"], want_masks=True)
    _check_safeopt(engine, ref, 2)
    assert res["minimizer_index"] == ref["minimizer_index"]
    assert res["n_exact_rechecks"] == ref["S"].sum()


def test_safeopt_single_output_has_no_constraints(engine):
    cfg = synthetic.make_config("E", n=64)
    pts = synthetic.scattered_points(cfg, 5000, dtype=np.float64)
    ref = oracle.safeopt_sweep(pts, cfg["ds"], 2.0)
    engine.set_model(cfg["ds"])
    engine.set_points(pts)
    res = engine.sweep_safeopt(2.0, want_masks=True)
    assert ref["S"].all() and np.array_equal(engine.mask("M"), ref["M"])
    assert res["minimizer_index"] == ref["minimizer_index"] and res["expander_best_c"] == 0


@pytest.mark.parametrize("count", [[5, 7], [1, 9], [17, 3], [33, 1]])
def test_small_and_degenerate_grids(engine, count):
    cfg = synthetic.make_config("A", n=20)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    pts = oracle.grid_points(lo, hi, count)
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, count)
    _check_posterior(engine, cfg["ds"], pts, TOL64)
    ref = oracle.safeopt_sweep(pts, cfg["ds"], 1.0)
    if ref["empty_safe_set"]:
        with pytest.raises(safebo_amd.EmptySafeSetError):
            engine.sweep_safeopt(1.0)
    else:
        res = engine.sweep_safeopt(1.0, want_masks=True)
        _check_safeopt(engine, ref, 2)
        assert res["minimizer_index"] == ref["minimizer_index"]


def test_one_dimensional_model_and_grid(engine):
    rng = np.random.default_rng(4)
    X = rng.uniform(-1, 1, size=(15, 1))
    Y = np.stack([np.sin(3 * X[:, 0]), 0.6 - X[:, 0] ** 2], axis=1)
    ds = synthetic.make_dataset(X, Y, synthetic.default_hypopt(1, 2))
    pts = oracle.grid_points([-1.0], [1.0], [301])
    engine.set_model(ds)
    engine.set_grid([-1.0], [1.0], [301])
    _check_posterior(engine, ds, pts, TOL64)
    ref = oracle.safeopt_sweep(pts, ds, 2.0)
    res = engine.sweep_safeopt(2.0, want_masks=True)
    _check_safeopt(engine, ref, 2)
    assert res["minimizer_index"] == ref["minimizer_index"] and list(res["expander_index_c"]) == list(ref["expander_index"])


@pytest.mark.parametrize("d,count,n,off,ll,b", [(3, [13, 11, 10], 40, 0.9, -0.5, 1.0), (5, [6, 5, 4, 5, 4], 90, 1.1, 0.9, 2.0),
                                                (6, [4, 4, 3, 4, 3, 4], 120, 1.3, 0.9, 2.0)])
def test_sweeps_on_grids_of_any_dimension(engine, d, count, n, off, ll, b):
    """d = 3 (padded to 4), 5 and 6 (padded to 8): separable grid posterior, every middle axis of the distance and power
    transforms, and the coordinate decoding of the results, against the oracle."""
    rng = np.random.default_rng(200 + d)
    X = rng.uniform(-1, 1, size=(n, d))
    Y = np.stack([np.sin(X.sum(1)) + 0.5 * X[:, 0], off - 1.8 * (X ** 2).mean(1) + 0.3 * X[:, 1]], axis=1)
    hyp = synthetic.default_hypopt(d, 2)
    hyp[:d, :] = ll
    ds = synthetic.make_dataset(X, Y, hyp)
    lo, hi = -np.ones(d), np.ones(d)
    pts = oracle.grid_points(lo, hi, count)
    engine.set_model(ds)
    engine.set_grid(lo, hi, count)
    _check_posterior(engine, ds, pts, TOL64)
    ref = oracle.safeopt_sweep(pts, ds, b)
    assert ref["G"].any() and ref["M"].any()
    res = engine.sweep_safeopt(b, want_masks=True, posterior_ready=True)
    _check_safeopt(engine, ref, 2)
    assert res["minimizer_index"] == ref["minimizer_index"] and list(res["expander_index_c"]) == list(ref["expander_index"])
    assert np.array_equal(res["minimizer_x"], pts[ref["minimizer_index"]])
    assert np.allclose(res["L"], ref["L"], rtol=1e-9)
    gref = oracle.goose_sweep(pts, ds, b)
    assert gref["O"].any()
    g = engine.sweep_goose(b, want_masks=True, posterior_ready=True)
    assert np.array_equal(engine.mask("O", 1), gref["O"][0])
    assert (g["safe_min_index"], g["target_index"], g["explore_index"]) == (gref["safe_min_index"], gref["target_index"],
                                                                             gref["explore_index"])
    assert np.array_equal(g["target_x"], pts[gref["target_index"]])
    x0 = pts[ref["minimizer_index"]]
    tref = oracle.tr_sweep(pts, ds, b, x0, 0.8)
    t = engine.sweep_tr(b, x0, 0.8, posterior_ready=True)
    assert t["index"] == tref["index"] and t["count_T"] == int(tref["T"].sum())


def test_fp32_sweep_with_fp64_recheck_equals_the_fp64_oracle(engine):
    """dtype f32 (hard part 2): the posterior runs in fp32, candidates whose fp32 bounds -- within the 1e-4
    contract -- cannot decide S, U, u*, M or the minimiser are re-evaluated in fp64 by the model's fp64 twin, and the set
    phase runs in fp64 on the result (Lipschitz constant in fp64, expander verdicts that an unrefined ucb cannot settle
    deferred, re-evaluated and decided in a second pass).  Every mask -- S, U, M, G -- u*, L and both acquisition indices must
    then be the fp64 oracle's, bit for bit; the number of re-evaluated candidates is reported (a few per cent of the grid)."""
    cfg = synthetic.make_config("B", n=128)
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], [96, 80]
    pts = oracle.grid_points(lo, hi, count)
    engine.set_model(cfg["ds"], dtype="f32", use_invK=False)
    engine.set_grid(lo, hi, count)
    _check_posterior(engine, cfg["ds"], pts, TOL32, dtype="f32")
    res = engine.sweep_safeopt(cfg["b"], want_masks=True, posterior_ready=True)
    prof = engine.profile()
    ref64 = oracle.safeopt_sweep(pts, cfg["ds"], cfg["b"])
    assert 0 < prof["fp64_rechecks"] < 0.35 * pts.shape[0], prof["fp64_rechecks"]
    for k in ("S", "U", "M"):
        assert np.array_equal(engine.mask(k), ref64[k]), k
    assert res["minimizer_index"] == ref64["minimizer_index"]
    assert res["u_star"] == pytest.approx(ref64["u_star"], rel=1e-10)
    assert res["minimizer_std"] == pytest.approx(ref64["minimizer_std"], rel=1e-9)
    assert (res["count_S"], res["count_U"], res["count_M"]) == (ref64["S"].sum(), ref64["U"].sum(), ref64["M"].sum())
    assert np.array_equal(engine.mask("G", 1), ref64["G"][0]) and ref64["G"][0].any()
    assert list(res["expander_index_c"]) == list(ref64["expander_index"]) and res["count_G"][0] == ref64["G"][0].sum()
    assert res["expander_std"] == pytest.approx(ref64["expander_best_std"], rel=1e-9)
    assert np.allclose(res["L"], ref64["L"], rtol=1e-9)
    assert res["choose_minimizer"] == ref64["choose_minimizer"]
    # the posterior the caller reads stays the fp32 one
    mean, var = engine.posterior()
    assert mean.dtype == np.float32
    # a fresh sweep (posterior recomputed inside) gives the same answer, and so does the incremental model path
    res2 = engine.sweep_safeopt(cfg["b"])
    assert res2["minimizer_index"] == res["minimizer_index"] and res2["count_M"] == res["count_M"]


@pytest.mark.parametrize("case", ["wo3_grid", "benoit_list", "rosen4_grid"])
def test_fp32_recheck_on_other_shapes(engine, case):
    """Three outputs (two expander sets), an explicit candidate list (exhaustive expander path: every possibly-safe candidate
    is re-evaluated) and a 4-D grid: fp32 model, every SafeOpt mask and index equal to the fp64 oracle."""
    if case == "wo3_grid":
        cfg, count, b = synthetic.make_config("C", n=64), [72, 64], 2.0
    elif case == "benoit_list":
        cfg, count, b = synthetic.make_config("A", n=20), None, 3.0
    else:
        cfg, count, b = synthetic.make_config("D", n=128), [10, 9, 8, 7], 0.5
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    engine.set_model(cfg["ds"], dtype="f32", use_invK=False)
    if count is None:
        pts = oracle.grid_points(lo, hi, [45, 40])
        engine.set_points(pts)
    else:
        pts = oracle.grid_points(lo, hi, count)
        engine.set_grid(lo, hi, count)
    ref64 = oracle.safeopt_sweep(pts, cfg["ds"], b)
    assert not ref64["empty_safe_set"]
    res = engine.sweep_safeopt(b, want_masks=True)
    assert engine.profile()["fp64_rechecks"] > 0
    for k in ("S", "U", "M"):
        assert np.array_equal(engine.mask(k), ref64[k]), k
    for c in range(1, cfg["q"]):
        assert np.array_equal(engine.mask("G", c), ref64["G"][c - 1]), f"G{c}"
    assert res["minimizer_index"] == ref64["minimizer_index"] and list(res["expander_index_c"]) == list(ref64["expander_index"])
    assert np.allclose(res["L"], ref64["L"], rtol=1e-9) and res["u_star"] == pytest.approx(ref64["u_star"], rel=1e-10)


def test_fp32_unconstrained_scattered_sweep_equals_the_fp64_oracle(engine):
    """Config E's shape at test size: scattered 6-D points, one output, fp32 -- no constraints, so the recheck is about u*, M
    and the minimiser only."""
    cfg = synthetic.make_config("E", n=300)
    pts = synthetic.scattered_points(cfg, 20000)
    engine.set_model(cfg["ds"], dtype="f32", use_invK=False)
    engine.set_points(pts)
    res = engine.sweep_safeopt(2.0, want_masks=True)
    prof = engine.profile()
    ref64 = oracle.safeopt_sweep(pts.astype(np.float64), cfg["ds"], 2.0)
    assert ref64["S"].all() and 0 < prof["fp64_rechecks"] < 0.5 * pts.shape[0], prof["fp64_rechecks"]
    assert np.array_equal(engine.mask("M"), ref64["M"])
    assert res["minimizer_index"] == ref64["minimizer_index"] and res["u_star"] == pytest.approx(ref64["u_star"], rel=1e-10)
    assert res["expander_best_c"] == 0


@pytest.mark.parametrize("case", ["benoit_grid", "wo3_grid", "benoit_list", "rosen4_grid", "sines6_list"])
def test_fp32_goose_and_trust_region_sweeps_equal_the_fp64_oracle(engine, case):
    """GoOSE (models/GoOSE.py:63-119) and trust-region (models/GP_TR.py:43-51) sweeps of fp32 models: every possibly-safe
    candidate (the sources of the optimistic sets need exact radii), every candidate whose S / U membership the fp32 bounds
    cannot decide, and the contenders of the target's arg-min over the optimistic sets are re-evaluated in fp64; without
    constraints only the contenders of the one arg-min.  S, U, O_c, the safe minimum, the targets, the explore index, L, and
    the trust-region arg-min must be the fp64 oracle's."""
    if case == "benoit_grid":
        cfg, count, b, pts = synthetic.make_config("B", n=128), [96, 80], 3.0, None
    elif case == "wo3_grid":
        cfg, count, b, pts = synthetic.make_config("C", n=64), [72, 64], 2.0, None
    elif case == "benoit_list":
        cfg, count, b = synthetic.make_config("A", n=20), None, 3.0
        pts = oracle.grid_points(cfg["bound"][:, 0], cfg["bound"][:, 1], [45, 40])
    elif case == "rosen4_grid":
        cfg, count, b, pts = synthetic.make_config("D", n=128), [10, 9, 8, 7], 0.5, None
    else:
        cfg, count, b = synthetic.make_config("E", n=300), None, 2.0
        pts = synthetic.scattered_points(cfg, 20000).astype(np.float64)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    q = cfg["q"]
    engine.set_model(cfg["ds"], dtype="f32", use_invK=False)
    if count is None:
        engine.set_points(pts)
    else:
        pts = oracle.grid_points(lo, hi, count)
        engine.set_grid(lo, hi, count)
    ref = oracle.goose_sweep(pts, cfg["ds"], b)
    assert not ref["empty_safe_set"]
    g = engine.sweep_goose(b, want_masks=True)
    assert engine.profile()["fp64_rechecks"] > 0
    assert np.array_equal(engine.mask("S"), ref["S"]) and np.array_equal(engine.mask("U"), ref["U"])
    for c in range(1, q):
        assert np.array_equal(engine.mask("O", c), ref["O"][c - 1]), f"O{c}"
    assert g["safe_min_index"] == ref["safe_min_index"] and g["safe_min_lcb"] == pytest.approx(ref["safe_min_lcb"], rel=1e-9, abs=1e-12)
    if q > 1:
        assert list(g["target_index_c"])[:q - 1] == list(ref["target_index_c"]) and g["target_index"] == ref["target_index"]
        assert g["explore_index"] == ref["explore_index"] and g["choose_safe_min"] == ref["choose_safe_min"]
        assert np.allclose(g["L"][:q], ref["L"], rtol=1e-9)
    # trust region around the safe minimum: a ball that cuts through S
    x0 = pts[ref["safe_min_index"]]
    r = 0.25 * float(np.max(hi - lo))
    tref = oracle.tr_sweep(pts, cfg["ds"], b, x0, r)
    t = engine.sweep_tr(b, x0, r)
    assert t["index"] == tref["index"] and t["lcb"] == pytest.approx(tref["lcb_min"], rel=1e-9, abs=1e-12)
    assert (t["count_S"], t["count_T"]) == (int(tref["S"].sum()), int(tref["T"].sum()))
    # and the sweeps a caller interleaves on one posterior stay consistent with each other
    s_ = engine.sweep_safeopt(b, posterior_ready=True)
    sref = oracle.safeopt_sweep(pts, cfg["ds"], b)
    assert s_["minimizer_index"] == sref["minimizer_index"] and s_["count_S"] == int(sref["S"].sum())


def test_fp32_sweep_without_recheck_follows_the_fp32_posterior(engine):
    """Option fp64_recheck = 0: the classification is a function of the fp32 posterior alone.  Against the fp64 oracle the
    masks may then differ, but only inside the band the fp32 posterior error can move a deciding bound across its threshold
    (tau = 4 x the measured posterior difference; counted and bounded); given the fp32 mean / var they are exact (the oracle
    evaluates the same expressions in numpy float32)."""
    cfg = synthetic.make_config("B", n=128)
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], [96, 80]
    pts = oracle.grid_points(lo, hi, count)
    engine.set_option("fp64_recheck", 0)
    try:
        engine.set_model(cfg["ds"], dtype="f32", use_invK=False)
        engine.set_grid(lo, hi, count)
        mean, var = _check_posterior(engine, cfg["ds"], pts, TOL32, dtype="f32")
        res = engine.sweep_safeopt(cfg["b"], want_masks=True, posterior_ready=True)
        assert engine.profile()["fp64_rechecks"] == 0
        got = {k: engine.mask(k) for k in ("S", "U", "M")}
        G = engine.mask("G", 1)
    finally:
        engine.set_option("fp64_recheck", 1)
    ref64 = oracle.safeopt_sweep(pts, cfg["ds"], cfg["b"])
    b = cfg["b"]
    dl = np.abs(mean.astype(np.float64) - ref64["mean"]) + b * np.abs(np.sqrt(var.astype(np.float64)) - np.sqrt(ref64["var"]))
    tau = 4.0 * dl.max(axis=0)                                   # per output, raw units
    near_S = np.abs(ref64["lcb"][:, 1]) <= tau[1]
    near_M = near_S | (np.abs(ref64["lcb"][:, 0] - ref64["u_star"]) <= 2.0 * tau[0])
    for k, near in (("S", near_S), ("U", near_S), ("M", near_M)):
        diff = got[k] != ref64[k]
        assert not (diff & ~near).any(), (k, int((diff & ~near).sum()))
        assert diff.sum() <= 0.005 * pts.shape[0], (k, int(diff.sum()))
    assert abs(res["u_star"] - ref64["u_star"]) <= 2.0 * tau[0]
    ref = oracle.safeopt_sweep(pts, cfg["ds"], cfg["b"], mean_var=(mean, var))
    for k in ("S", "U", "M"):
        assert np.array_equal(got[k], ref[k]), k
    assert res["minimizer_index"] == ref["minimizer_index"]
    assert res["u_star"] == float(ref["u_star"])
    Gref = ref["G"][0]
    assert not (G & ~ref["S"]).any() and (G != Gref).sum() <= 2e-3 * max(1, Gref.sum())


def test_no_unsafe_witness_means_no_expander(engine):
    """Every candidate safe and none fully unsafe: G_c is empty, the expander is reported as absent and the loop rule
    falls back to the minimiser (test/test_SafeOpt.py:153-158 with std_expander = 0)."""
    rng = np.random.default_rng(8)
    X = rng.uniform(-1, 1, size=(30, 2))
    Y = np.stack([X[:, 0] ** 2 + X[:, 1] ** 2, 50.0 + 0.01 * X[:, 0]], axis=1)      # constraint far above zero everywhere
    ds = synthetic.make_dataset(X, Y, synthetic.default_hypopt(2, 2))
    pts = X + 1e-3            # candidates next to the observations: the pessimistic prior cannot pull them down
    ref = oracle.safeopt_sweep(pts, ds, 1.0)
    assert ref["S"].all() and not ref["U"].any() and not ref["G"].any()
    engine.set_model(ds)
    engine.set_points(pts)
    res = engine.sweep_safeopt(1.0, want_masks=True)
    _check_safeopt(engine, ref, 2)
    assert res["expander_best_c"] == 0 and res["expander_index"] == -1 and res["expander_std"] == 0.0
    assert res["choose_minimizer"] and res["minimizer_index"] == ref["minimizer_index"]
    g = engine.sweep_goose(1.0, want_masks=True)
    assert g["target_index"] == -1 and g["choose_safe_min"] and not engine.mask("O", 1).any()
    assert g["safe_min_index"] == oracle.goose_sweep(pts, ds, 1.0)["safe_min_index"]


def test_single_candidate_and_empty_shard(engine):
    cfg = synthetic.make_config("A", n=20)
    x = np.array([[1.4, -0.8]])                      # inside the sampled safe region
    ref = oracle.safeopt_sweep(x, cfg["ds"], 0.5)
    engine.set_model(cfg["ds"])
    engine.set_points(x)
    if ref["empty_safe_set"]:
        with pytest.raises(safebo_amd.EmptySafeSetError):
            engine.sweep_safeopt(0.5)
    else:
        res = engine.sweep_safeopt(0.5)
        assert res["minimizer_index"] == 0 and res["count_S"] == 1
    engine.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], [10, 10], first=30, n_local=0)
    with pytest.raises(safebo_amd.EmptySafeSetError):
        engine.sweep_safeopt(0.5)


@pytest.mark.parametrize("first_plane,planes", [(0, 20), (13, 24), (30, 20)])
def test_sweep_of_a_plane_range_treats_it_as_the_candidate_set(engine, first_plane, planes):
    """A single rank sweeping hyper-planes [first, first + planes) of a grid: the candidate set is that range (witnesses
    outside it do not exist), indices reported are the grid's global flat indices."""
    cfg = synthetic.make_config("B", n=128)
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], [64, 50]
    first, nloc = first_plane * 64, planes * 64
    pts = oracle.grid_points(lo, hi, count, first=first, n=nloc)
    ref = oracle.safeopt_sweep(pts, cfg["ds"], cfg["b"])
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, count, first=first, n_local=nloc)
    if ref["empty_safe_set"]:
        with pytest.raises(safebo_amd.EmptySafeSetError):
            engine.sweep_safeopt(cfg["b"])
        return
    res = engine.sweep_safeopt(cfg["b"], want_masks=True)
    _check_safeopt(engine, ref, 2)
    assert res["minimizer_index"] == first + ref["minimizer_index"]
    assert [int(i) for i in res["expander_index_c"]] == [first + int(i) if i >= 0 else -1 for i in ref["expander_index"]]
    # a range that is not made of whole planes goes through the exhaustive path
    engine.set_grid(lo, hi, count, first=first + 7, n_local=nloc - 20)
    pts2 = oracle.grid_points(lo, hi, count, first=first + 7, n=nloc - 20)
    ref2 = oracle.safeopt_sweep(pts2, cfg["ds"], cfg["b"])
    if not ref2["empty_safe_set"]:
        engine.sweep_safeopt(cfg["b"], want_masks=True)
        _check_safeopt(engine, ref2, 2)


def test_full_size_properties_config_B(engine):
    """BASELINE.json configs[1] at full size (2048^2, n = 128): properties that do not need the whole oracle."""
    cfg = synthetic.make_config("B")
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"]
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, count)
    res = engine.sweep_safeopt(cfg["b"], want_masks=True)
    N = count[0] * count[1]
    lcb1, lcb0, ucb0, var0 = (engine.bounds(cfg["b"], 1, "lcb"), engine.bounds(cfg["b"], 0, "lcb"),
                              engine.bounds(cfg["b"], 0, "ucb"), engine.bounds(cfg["b"], 0, "var"))
    S, U, M, G = engine.mask("S"), engine.mask("U"), engine.mask("M"), engine.mask("G", 1)
    assert np.array_equal(S, lcb1 >= 0) and np.array_equal(U, lcb1 <= 0)          # masks follow the bounds bit for bit
    assert res["u_star"] == ucb0[S].min()
    assert np.array_equal(M, S & (lcb0 <= res["u_star"]))
    assert res["minimizer_index"] == int(np.argmax(np.where(M, var0, -np.inf)))
    assert res["expander_index_c"][0] == int(np.argmax(np.where(G, var0, -np.inf)))
    assert not (G & ~S).any() and (res["count_S"], res["count_M"], res["count_G"][0]) == (S.sum(), M.sum(), G.sum())
    # posterior against the oracle on a random subset, and the expander predicate on a random subset of S
    rng = np.random.default_rng(5)
    sub = np.sort(rng.choice(N, size=4096, replace=False))
    pts = oracle.grid_points(lo, hi, count)
    om, ov = oracle.gp_inference(pts[sub], cfg["ds"])
    mean, var = engine.posterior()
    assert _nerr(mean[sub], om, cfg["ds"]["Y_std"], 1) < TOL64 and _nerr(var[sub], ov, cfg["ds"]["Y_std"], 2) < TOL64
    ucb1 = engine.bounds(cfg["b"], 1, "ucb")
    xh = pts[U]
    for g in rng.choice(np.nonzero(S)[0], size=64, replace=False):
        want = bool(np.any(ucb1[g] - res["L"][1] * oracle.shifted_norm(pts[g][None, :], xh) >= 0))
        assert bool(G[g]) == want
    # sweeping again is idempotent, and reusing the posterior gives the same answer
    res2 = engine.sweep_safeopt(cfg["b"], posterior_ready=True)
    assert res2["minimizer_index"] == res["minimizer_index"] and np.array_equal(res2["count_G"], res["count_G"])


def test_full_size_goose_properties_config_B(engine):
    """GoOSE iteration of BASELINE.json configs[1] at full size (2048^2, n = 128): the optimistic set of the transform path
    equals the pruned exact pair evaluation on all 4.2 M candidates, the oracle's predicate re-decides a sample of U on both
    sides of the boundary of O_1, and the arg-min outputs are exact functions of the device bounds and masks."""
    cfg = synthetic.make_config("B")
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"]
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, count)
    out = {}
    try:
        for pairs in (0, 1):
            engine.set_option("goose_pairs", pairs)
            res = engine.sweep_goose(cfg["b"], want_masks=True)
            out[pairs] = (res, engine.mask("O", 1))
    finally:
        engine.set_option("goose_pairs", 0)
    res, O = out[0]
    assert np.array_equal(O, out[1][1])
    for k in ("safe_min_index", "target_index", "explore_index", "choose_safe_min"):
        assert res[k] == out[1][0][k], k
    S, U = engine.mask("S"), engine.mask("U")
    lcb0, lcb1, ucb1 = engine.bounds(cfg["b"], 0, "lcb"), engine.bounds(cfg["b"], 1, "lcb"), engine.bounds(cfg["b"], 1, "ucb")
    assert np.array_equal(S, lcb1 >= 0) and np.array_equal(U, lcb1 <= 0) and not (O & ~U).any()
    assert res["count_O"][0] == O.sum() and res["count_S"] == S.sum()
    assert res["safe_min_index"] == int(np.argmin(np.where(S, lcb0, np.inf)))      # models/GoOSE.py:69-76 on the grid
    assert res["target_index"] == int(np.argmin(np.where(O, lcb0, np.inf)))        # models/GoOSE.py:100-112
    pts = oracle.grid_points(lo, hi, count)
    tgt = pts[res["target_index"]]
    dist = np.sqrt(((pts - tgt[None, :]) ** 2).sum(axis=1))
    assert res["explore_index"] == int(np.argmin(np.where(S, dist, np.inf)))       # models/GoOSE.py:116-119
    rng = np.random.default_rng(9)
    Lq = res["L"][cfg["q"] - 1]
    edge = np.nonzero(O[:-1] != O[1:])[0]
    pick = np.concatenate([rng.choice(np.nonzero(U)[0], size=24, replace=False), rng.choice(edge, size=24, replace=False)])
    src, usrc = pts[S], ucb1[S]
    for hidx in pick:
        if U[hidx]:
            want = bool(np.any(usrc - Lq * oracle.shifted_norm(src, pts[hidx][None, :]) >= 0))
            assert bool(O[hidx]) == want, int(hidx)


def test_full_size_goose_properties_config_C(engine):
    """BASELINE.json configs[2] as specified: the William-Otto reactor (problems/WilliamOttoReactor_Problem.py, the plant of
    test/test_GoOSE.py:276-280) on a 1024 x 1024 grid, n = 256 observations, q = 3 outputs, b = 2, GoOSE.  Properties that
    hold at full size without the brute-force oracle: masks are exact functions of the device bounds, the transform's
    optimistic sets equal the pruned exact pair evaluation on all candidates, the oracle's predicates re-decide samples on
    both sides of the boundaries of G_c and O_c, every arg-min is the NumPy arg-min of the device values, and the
    posterior matches the oracle on a random subset; the SafeOpt sweep of the same posterior is checked the same way."""
    cfg = synthetic.make_config("C")
    assert cfg["plant"] == "wo" and cfg["ds"]["X_norm"].shape == (256, 2) and cfg["q"] == 3 and cfg["count"] == [1024, 1024]
    lo, hi, count, b, q = cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"], cfg["b"], 3
    N = count[0] * count[1]
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, count)
    out = {}
    try:
        for pairs in (0, 1):
            engine.set_option("goose_pairs", pairs)
            res = engine.sweep_goose(b, want_masks=True)
            out[pairs] = (res, [engine.mask("O", c) for c in (1, 2)])
    finally:
        engine.set_option("goose_pairs", 0)
    res, O = out[0]
    assert engine.profile()["posterior_kernel"] in (4, 6)
    for c in range(2):
        assert np.array_equal(O[c], out[1][1][c]), f"O{c + 1}"
    for k in ("safe_min_index", "target_index", "explore_index", "choose_safe_min", "target_best_c"):
        assert res[k] == out[1][0][k], k
    S, U = engine.mask("S"), engine.mask("U")
    lcb = [engine.bounds(b, i, "lcb") for i in range(3)]
    ucb = [engine.bounds(b, i, "ucb") for i in range(3)]
    assert np.array_equal(S, (lcb[1] >= 0) & (lcb[2] >= 0)) and np.array_equal(U, (lcb[1] <= 0) & (lcb[2] <= 0))
    assert S.any() and U.any() and O[0].any() and O[1].any()
    assert res["count_S"] == S.sum() and res["count_U"] == U.sum() and list(res["count_O"]) == [O[0].sum(), O[1].sum()]
    assert res["safe_min_index"] == int(np.argmin(np.where(S, lcb[0], np.inf)))        # models/GoOSE.py:63-67
    tgt_c = [int(np.argmin(np.where(O[c], lcb[0], np.inf))) for c in range(2)]
    assert list(res["target_index_c"]) == tgt_c                                          # models/GoOSE.py:100-112
    best = int(np.argmin([lcb[0][t] for t in tgt_c]))
    assert res["target_best_c"] == best + 1 and res["target_index"] == tgt_c[best]
    pts = oracle.grid_points(lo, hi, count)
    dist = np.sqrt(((pts - pts[res["target_index"]][None, :]) ** 2).sum(axis=1))
    assert res["explore_index"] == int(np.argmin(np.where(S, dist, np.inf)))             # models/GoOSE.py:116-119
    assert res["choose_safe_min"] == bool(res["safe_min_lcb"] <= res["target_lcb"])      # test/test_GoOSE.py:158
    # SafeOpt sweep on the same posterior: expander masks, then the oracle's pair predicates around the set boundaries
    s = engine.sweep_safeopt(b, want_masks=True, posterior_ready=True)
    G = [engine.mask("G", c) for c in (1, 2)]
    M = engine.mask("M")
    var0 = engine.bounds(b, 0, "var")
    assert s["u_star"] == ucb[0][S].min() and np.array_equal(M, S & (lcb[0] <= s["u_star"]))
    assert s["minimizer_index"] == int(np.argmax(np.where(M, var0, -np.inf)))
    assert list(s["expander_index_c"]) == [int(np.argmax(np.where(G[c], var0, -np.inf))) if G[c].any() else -1 for c in range(2)]
    # (the posterior in place is the LAST GoOSE sweep's: the first sweep of the model ran on interpolated node values (K1i), the
    # sweeps after it on K1b's plan -- the two agree to ~1e-12, not bit for bit)
    assert np.array_equal(s["L"], out[1][0]["L"]) and np.allclose(s["L"], res["L"], rtol=1e-10, atol=0.0)
    rng = np.random.default_rng(13)
    Lq = res["L"][q - 1]                                   # reference quirk: every constraint uses L_{q-1}
    xu, xs = pts[U], pts[S]
    for c in range(2):
        assert not (G[c] & ~S).any() and not (O[c] & ~U).any()
        us = ucb[c + 1][S]
        for mask, dom in ((G[c], S), (O[c], U)):
            edge = np.nonzero((mask[:-1] != mask[1:]) & dom[:-1] & dom[1:])[0]
            pick = rng.choice(edge, size=min(10, edge.size), replace=False) if edge.size else np.array([], dtype=int)
            extra = rng.choice(np.nonzero(dom)[0], size=10, replace=False)
            for i in np.concatenate([pick, pick + 1, extra]):
                if mask is G[c]:
                    want = bool(np.any(ucb[c + 1][i] - Lq * oracle.shifted_norm(pts[i][None, :], xu) >= 0))
                else:
                    want = bool(np.any(us - Lq * oracle.shifted_norm(xs, pts[i][None, :]) >= 0))
                assert bool(mask[i]) == want, (c + 1, "G" if mask is G[c] else "O", int(i))
    mean, var = engine.posterior()
    sub = np.sort(rng.choice(N, size=4096, replace=False))
    om, ov = oracle.gp_inference(pts[sub], cfg["ds"])
    assert _nerr(mean[sub], om, cfg["ds"]["Y_std"], 1) < TOL64 and _nerr(var[sub], ov, cfg["ds"]["Y_std"], 2) < TOL64


def test_full_size_properties_config_H(engine):
    """The headline configuration (4096^2 grid, n = 512, fp64) at full size on one GPU: the masks are exact functions of
    the device posterior, the posterior matches the oracle on a random subset and the two posterior kernels agree on the
    whole grid (linearity-free cross-check of 16.8 M values each)."""
    cfg = synthetic.make_config("H")
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"]
    N = count[0] * count[1]
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, count)
    res = engine.sweep_safeopt(cfg["b"], want_masks=True)
    assert engine.profile()["posterior_kernel"] in (4, 6)
    lcb1, lcb0, ucb0, var0 = (engine.bounds(cfg["b"], 1, "lcb"), engine.bounds(cfg["b"], 0, "lcb"),
                              engine.bounds(cfg["b"], 0, "ucb"), engine.bounds(cfg["b"], 0, "var"))
    S, U, M, G = engine.mask("S"), engine.mask("U"), engine.mask("M"), engine.mask("G", 1)
    assert np.array_equal(S, lcb1 >= 0) and np.array_equal(U, lcb1 <= 0)
    assert res["u_star"] == ucb0[S].min() and np.array_equal(M, S & (lcb0 <= res["u_star"]))
    assert res["minimizer_index"] == int(np.argmax(np.where(M, var0, -np.inf)))
    assert res["expander_index_c"][0] == int(np.argmax(np.where(G, var0, -np.inf)))
    assert not (G & ~S).any() and (res["count_S"], res["count_M"], res["count_G"][0]) == (S.sum(), M.sum(), G.sum())
    mean, var = engine.posterior()
    rng = np.random.default_rng(6)
    sub = np.sort(rng.choice(N, size=2048, replace=False))
    ax = oracle.grid_axes(lo, hi, count)
    pts_sub = np.stack([ax[0][sub % count[0]], ax[1][sub // count[0]]], axis=1)
    om, ov = oracle.gp_inference(pts_sub, cfg["ds"])
    assert _nerr(mean[sub], om, cfg["ds"]["Y_std"], 1) < TOL64 and _nerr(var[sub], ov, cfg["ds"]["Y_std"], 2) < TOL64
    engine.set_option("bilinear", 0)
    try:
        res_t = engine.sweep_safeopt(cfg["b"])
        assert engine.profile()["posterior_kernel"] == 3
        m_t, v_t = engine.posterior()
    finally:
        engine.set_option("bilinear", 1)
    ystd = np.maximum(1.0, cfg["ds"]["Y_std"])
    assert np.max(np.abs(mean - m_t) / ystd) < 1e-11 and np.max(np.abs(var - v_t) / ystd ** 2) < 1e-11
    assert res_t["minimizer_index"] == res["minimizer_index"] and res_t["count_S"] == res["count_S"]
    assert np.array_equal(res_t["count_G"], res["count_G"])


@pytest.mark.parametrize("cfg_name,n,count", [("B", 128, [384, 320]), ("C", 64, [352, 416]), ("D", 128, [36, 34, 33, 32])])
def test_goose_transform_equals_pair_evaluation_with_coarse_bounds(engine, cfg_name, n, count):
    """Grids large enough for the coarse cell bounds of the power-distance transform: the optimistic sets must be the
    ones the pruned exact pair evaluation produces (the path the golden vectors pin on explicit lists), and a random
    subset of U is re-decided by the oracle's predicate."""
    cfg = synthetic.make_config(cfg_name, n=n)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, count)
    out = {}
    try:
        for pairs in (1, 0):
            engine.set_option("goose_pairs", pairs)
            res = engine.sweep_goose(cfg["b"], want_masks=True)
            out[pairs] = (res, [engine.mask("O", c) for c in range(1, cfg["q"])])
    finally:
        engine.set_option("goose_pairs", 0)
    for c in range(cfg["q"] - 1):
        assert np.array_equal(out[0][1][c], out[1][1][c]), f"O{c + 1}"
    for k in ("safe_min_index", "target_index", "explore_index", "choose_safe_min"):
        assert out[0][0][k] == out[1][0][k], k
    assert np.array_equal(out[0][0]["count_O"], out[1][0]["count_O"])
    S, U = engine.mask("S"), engine.mask("U")
    pts = oracle.grid_points(lo, hi, count)
    rng = np.random.default_rng(11)
    Lq = out[0][0]["L"][cfg["q"] - 1]                     # reference quirk: every constraint uses L_{q-1}
    for c in range(1, cfg["q"]):
        ucb = engine.bounds(cfg["b"], c, "ucb")
        O = out[0][1][c - 1]
        # near the boundary of O_c is where a wrong verdict would sit: sample covered points and their uncovered neighbours
        cand = np.nonzero(U)[0]
        pick = rng.choice(cand, size=min(48, cand.size), replace=False)
        edge = np.nonzero(O[:-1] != O[1:])[0]
        pick = np.concatenate([pick, rng.choice(edge, size=min(16, edge.size), replace=False)]) if edge.size else pick
        for hidx in pick:
            if not U[hidx]:
                continue
            want = bool(np.any(ucb[S] - Lq * oracle.shifted_norm(pts[S], pts[hidx][None, :]) >= 0))
            assert bool(O[hidx]) == want, (c, int(hidx))


@pytest.mark.parametrize("cfg_name,n,count", [("B", 128, [320, 300]), ("A", 64, [130, 70]), ("H", 300, [200, 144])])
def test_fused_classification_equals_the_separate_pass(engine, cfg_name, n, count):
    """One-constraint sweeps on the GEMM posterior take S / U, |S|, |U| and the radius key from the posterior kernel's mean
    epilogue (option fuse_classify; ragged tiles included: 130 x 70, 200 x 144): every mask, count and index must
    equal the separate classification pass, for the SafeOpt, GoOSE and trust-region sweeps."""
    cfg = synthetic.make_config(cfg_name, n=n)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    out = {}
    try:
        for fuse in (1, 0):
            engine.set_option("fuse_classify", fuse)
            engine.set_model(cfg["ds"])
            engine.set_grid(lo, hi, count)
            s_ = engine.sweep_safeopt(cfg["b"], want_masks=True)
            assert engine.profile()["posterior_kernel"] in (4, 6)
            masks = {k: engine.mask(k) for k in ("S", "U", "M")}
            masks["G"] = engine.mask("G", 1)
            engine.set_model(cfg["ds"])                       # a fresh posterior for each sweep kind: the fused path again
            g_ = engine.sweep_goose(cfg["b"], want_masks=True)
            masks["O"] = engine.mask("O", 1)
            masks["S_g"], masks["U_g"] = engine.mask("S"), engine.mask("U")
            engine.set_model(cfg["ds"])
            t_ = engine.sweep_tr(cfg["b"], s_["minimizer_x"], 0.5)
            out[fuse] = (s_, g_, t_, masks)
    finally:
        engine.set_option("fuse_classify", -1)
    for k, v in out[1][3].items():
        assert np.array_equal(v, out[0][3][k]), k
    assert out[1][3]["S"].any() and out[1][3]["U"].any()
    for which in (0, 1, 2):
        a, b_ = out[1][which], out[0][which]
        for k in a:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b_[k])), (which, k)
    # and against the oracle where it is affordable
    if count[0] * count[1] <= 20000:
        ref = oracle.safeopt_sweep(oracle.grid_points(lo, hi, count), cfg["ds"], cfg["b"])
        assert np.array_equal(out[1][3]["S"], ref["S"]) and np.array_equal(out[1][3]["G"], ref["G"][0])
        assert out[1][0]["u_star"] == pytest.approx(ref["u_star"], rel=1e-10) and out[1][0]["count_U"] == int(ref["U"].sum())


@pytest.mark.parametrize("cfg_name,n,count,b", [("B", 128, [1100, 1024], 3.0), ("C", 96, [1040, 1030], 2.0), ("H", 300, [1056, 1500], 3.0),
                                                  ("B", 40, [1032, 1027], 1.0)])
def test_shared_set_phase_launches_equal_one_launch_per_kernel(engine, cfg_name, n, count, b):
    """On 2-D grids of one rank the independent kernels of the set phase share launches (k_edt_axis0_pair: both axis-0
    passes + the merge of the classification partials and of K1b's Lipschitz partials; k_set_mid: coarse last-axis scan +
    minimiser + block minima; option set_fuse, default on).  Every mask, count, index, u* and L must equal the sweep
    with one launch per kernel -- one and two constraints, ragged line lengths."""
    cfg = synthetic.make_config(cfg_name, n=n)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    q = cfg["q"]
    out = {}
    try:
        for fuse in (1, 0):
            engine.set_option("set_fuse", fuse)
            engine.set_grid(lo, hi, count)
            engine.set_model(cfg["ds"])
            r = engine.sweep_safeopt(b, want_masks=True)
            assert engine.profile()["posterior_kernel"] in (4, 6)
            masks = {k: engine.mask(k) for k in ("S", "U", "M")}
            masks.update({f"G{c}": engine.mask("G", c) for c in range(1, q)})
            r2 = engine.sweep_safeopt(b, quirk_L_index=False, posterior_ready=True, want_masks=True)
            masks.update({f"G{c}nq": engine.mask("G", c) for c in range(1, q)})
            out[fuse] = (r, r2, masks)
    finally:
        engine.set_option("set_fuse", 1)
    for k, v in out[1][2].items():
        assert np.array_equal(v, out[0][2][k]), k
    assert out[1][2]["S"].any() and out[1][2]["U"].any() and out[1][2]["G1"].any()
    for which in (0, 1):
        a, b_ = out[1][which], out[0][which]
        for k in a:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b_[k])), (which, k)


def _sweep_bundle(engine, cfg, b, q, goose=True):
    """SafeOpt (+ GoOSE) sweep of the resident model / grid, each on a FRESH posterior (so that the posterior kernels run inside
    the sweep, which is what the overlapped path needs): results, masks and the profile flag."""
    engine.set_model(cfg["ds"])
    r = engine.sweep_safeopt(b, want_masks=True)
    prof = engine.profile()
    masks = {k: engine.mask(k) for k in ("S", "U", "M")}
    masks.update({f"G{c}": engine.mask("G", c) for c in range(1, q)})
    g = None
    if goose:
        engine.set_model(cfg["ds"])
        g = engine.sweep_goose(b, want_masks=True)
        masks.update({f"O{c}": engine.mask("O", c) for c in range(1, q)})
        masks["S_g"], masks["U_g"] = engine.mask("S"), engine.mask("U")
    return r, g, masks, prof


def _assert_same_bundle(a, b_):
    for k, v in a[2].items():
        assert np.array_equal(v, b_[2][k]), k
    for which in (0, 1):
        if a[which] is None:
            continue
        for k in a[which]:
            assert np.array_equal(np.asarray(a[which][k]), np.asarray(b_[which][k])), (which, k)


@pytest.mark.parametrize("count,opts", [([1024, 36], {}), ([2048, 100], {}), ([2048, 2048], {"scan_waves": 0}), ([2048, 2048], {"scan_blocks": 0}),
                                        ([2048, 2048], {"result_mirror": 0}), ([1100, 1024], {"scan_waves": 0})])
def test_shared_launch_path_with_plain_scans_and_short_last_axes(engine, count, opts):
    """The 16-bit image of the fine axis-0 pass is decoded by the block minima and the list scan only: where
    the verdict kernel scans the image itself -- last axes too short for block minima (1024 x 36, 2048 x 100), or the options
    scan_waves / scan_blocks = 0 -- the shared-launch path must keep squared distances as doubles.  G masks, counts and indices
    must equal the one-launch-per-kernel sweep (set_fuse = 0) in every combination, and the oracle's where it is affordable;
    result_mirror is toggled too."""
    cfg = synthetic.make_config("B", n=128)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    engine.set_grid(lo, hi, count)
    out = {}
    defaults = {"scan_waves": 1, "scan_blocks": 1, "result_mirror": 1, "set_fuse": 1}
    try:
        for key, o in (("opt", opts), ("plain", {"set_fuse": 0})):
            for k, v in o.items():
                engine.set_option(k, v)
            out[key] = _sweep_bundle(engine, cfg, cfg["b"], 2, goose=False)
            for k in o:
                engine.set_option(k, defaults[k])
    finally:
        for k, v in defaults.items():
            engine.set_option(k, v)
    assert out["opt"][2]["G1"].any() and out["opt"][2]["U"].any()
    _assert_same_bundle(out["opt"], out["plain"])
    if count[0] * count[1] <= 40000:
        ref = oracle.safeopt_sweep(oracle.grid_points(lo, hi, count), cfg["ds"], cfg["b"])
        assert np.array_equal(out["opt"][2]["S"], ref["S"]) and np.array_equal(out["opt"][2]["G1"], ref["G"][0])
        assert list(out["opt"][0]["expander_index_c"])[:1] == list(ref["expander_index"])


@pytest.mark.parametrize("cfg_name,n,count", [("B", 128, [1100, 1024]), ("A", 20, [50, 50]), ("H", 300, [200, 144])])
def test_late_exact_recheck_path_equals_the_eager_one(engine, cfg_name, n, count):
    """One-constraint SafeOpt sweeps on one rank launch the exhaustive recheck of in-band candidates only when the result
    block reports any (option exact_lazy, default 1).  The late path -- recheck, expanders' arg-max and finals once more --
    forced on every sweep (exact_lazy = 2) must give what the eager launch order (exact_lazy = 0) gives."""
    cfg = synthetic.make_config(cfg_name, n=n)
    engine.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], count)
    out = {}
    try:
        for lazy in (2, 1, 0):
            engine.set_option("exact_lazy", lazy)
            out[lazy] = _sweep_bundle(engine, cfg, cfg["b"], 2, goose=False)
    finally:
        engine.set_option("exact_lazy", 1)
    _assert_same_bundle(out[2], out[0])
    _assert_same_bundle(out[1], out[0])


def test_two_lanes_of_constraints_equal_one_after_the_other(engine):
    """Models with two constraints on one rank: the per-constraint chains of a sweep (expander; GoOSE: + optimistic set)
    run on two streams with their own scratch and their own snapshot of the scalar block (option set_lanes, default on).
    SafeOpt and GoOSE results and masks must equal the one-lane sweep -- on a sequence of candidate sets of changing size
    (small grid, larger explicit list, full-size grid, repeated), which is what exposes a lane reading the other's counters."""
    cfg = synthetic.make_config("C", n=64)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    rng = np.random.default_rng(11)
    lists = [rng.uniform(lo, hi, size=(m, 2)) for m in (3000, 700)]
    steps = [("grid", [48, 40]), ("list", 0), ("grid", [1030, 1024]), ("list", 1), ("grid", [264, 256]), ("list", 0), ("grid", [48, 40])]
    out = {}
    try:
        for lanes in (1, 0):
            engine.set_option("set_lanes", lanes)
            engine.set_model(cfg["ds"])
            rows = []
            for kind, arg in steps * 2:
                if kind == "grid":
                    engine.set_grid(lo, hi, arg)
                else:
                    engine.set_points(lists[arg])
                s_ = engine.sweep_safeopt(cfg["b"], want_masks=True)
                masks = [engine.mask(k) for k in ("S", "U", "M")] + [engine.mask("G", c) for c in (1, 2)]
                g_ = engine.sweep_goose(cfg["b"], want_masks=True, posterior_ready=True)
                masks += [engine.mask("O", c) for c in (1, 2)]
                rows.append((s_, g_, masks))
            out[lanes] = rows
    finally:
        engine.set_option("set_lanes", 1)
    for (s1, g1, m1), (s0, g0, m0) in zip(out[1], out[0]):
        for a, b_ in zip(m1, m0):
            assert np.array_equal(a, b_)
        for a, b_ in ((s1, s0), (g1, g0)):
            for k in a:
                assert np.array_equal(np.asarray(a[k]), np.asarray(b_[k])), k
    assert any(m[3].any() and m[4].any() and m[5].any() for _, _, m in out[1])


def test_repeated_sweeps_on_a_resident_posterior_are_idempotent(engine):
    """The host classes call several sweeps per iteration on one posterior (`posterior_ready=True`): SafeOpt, GoOSE
    (whose explore step parks its target next to the sweep scalars) and the trust-region step, in any order, must
    leave the resident state (posterior, Lipschitz keys) untouched -- every repetition returns the same answer."""
    cfg = synthetic.make_config("B", n=64)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, [320, 288])
    engine.posterior_run()
    first = {}
    for rep in range(3):
        s = engine.sweep_safeopt(cfg["b"], posterior_ready=True)
        g = engine.sweep_goose(cfg["b"], posterior_ready=True, want_masks=True)
        O = engine.mask("O", 1)
        t = engine.sweep_tr(cfg["b"], np.array([1.2, -0.6]), 0.4, posterior_ready=True)
        got = dict(s_min=s["minimizer_index"], s_exp=list(s["expander_index_c"]), s_cnt=[s["count_S"], s["count_M"]] + list(s["count_G"]),
                   s_L=list(s["L"]), g_idx=(g["safe_min_index"], g["target_index"], g["explore_index"]), g_cnt=list(g["count_O"]),
                   g_L=list(g["L"]), O=O, t_idx=t["index"], t_cnt=(t["count_S"], t["count_T"]))
        if rep == 0:
            first = got
            continue
        for k, v in got.items():
            assert np.array_equal(np.asarray(v), np.asarray(first[k])), (rep, k)


# ---------------------------------------------------------------------------------------------- two ranks, one GPU
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _wait_ranks(procs, timeout=600):
    """Exit codes of the rank workers; whatever happens (a failed rank, a timeout, an exception in the test body between
    launch and wait), no worker is left behind holding the GPU."""
    try:
        return [p.wait(timeout=timeout) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()


@pytest.mark.parametrize("world,cfg_name,n,count,b", [(2, "A", 20, [50, 37], 3.0), (2, "C", 64, [48, 41], 2.0),
                                                       (2, "D", 128, [9, 8, 7, 5], 0.5), (4, "C", 64, [40, 83], 2.0),
                                                       (3, "B", 128, [64, 50], 3.0)])
def test_multi_rank_sweep_on_one_gpu_matches_oracle(tmp_path, world, cfg_name, n, count, b):
    """2-4 ranks share the test box's single GPU (collectives over the gloo relay): uneven plane shards, halo windows
    of the expander transform for first / middle / last ranks, host merge of the arg-max slots."""
    port, out = _free_port(), str(tmp_path / "res.json")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gpu_rank_worker.py"), str(r), str(world), port, out, cfg_name,
                               str(n), json.dumps(count), str(b)]) for r in range(world)]
    assert _wait_ranks(procs) == [0] * world
    res = json.load(open(out))
    cfg = synthetic.make_config(cfg_name, n=n)
    pts = oracle.grid_points(cfg["bound"][:, 0], cfg["bound"][:, 1], count)
    ref = oracle.safeopt_sweep(pts, cfg["ds"], b)
    parts = [np.load(out + f".rank{r}.npz") for r in range(world)]
    assert int(parts[0]["first"]) == 0
    for r in range(1, world):
        assert int(parts[r]["first"]) == int(parts[r - 1]["first"]) + int(parts[r - 1]["n_local"])
    for k in ("S", "U", "M"):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), ref[k]), k
    for c in range(1, cfg["q"]):
        assert np.array_equal(np.concatenate([p[f"G{c}"] for p in parts]), ref["G"][c - 1])
    assert res["minimizer_index"] == ref["minimizer_index"]
    assert res["expander_index_c"] == [int(x) for x in ref["expander_index"]]
    assert res["count_S"] == int(ref["S"].sum()) and res["count_G"] == [int(x) for x in ref["G"].sum(1)]
    assert res["u_star"] == pytest.approx(ref["u_star"], rel=1e-10)
    # the GoOSE sweep on the same shards: optimistic sets from the all-gathered source weights, merged arg-min slots
    gref = oracle.goose_sweep(pts, cfg["ds"], b)
    g = res["goose"]
    assert g is not None and not g.get("empty_safe_set", False)
    for c in range(1, cfg["q"]):
        assert np.array_equal(np.concatenate([p[f"O{c}"] for p in parts]), gref["O"][c - 1]), f"O{c}"
    assert g["safe_min_index"] == gref["safe_min_index"]
    assert g["target_index_c"] == [int(x) for x in gref["target_index_c"]]
    assert g["target_index"] == gref["target_index"] and g["explore_index"] == gref["explore_index"]
    assert g["choose_safe_min"] == gref["choose_safe_min"]
    assert g["count_O"] == [int(x) for x in gref["O"].sum(1)]


@pytest.mark.parametrize("world,cfg_name,n,count,b", [(3, "C", 64, [256, 300], 2.0), (2, "D", 128, [34, 33, 32, 70], 0.5),
                                                      (2, "D", 128, [64, 64, 64, 64], 3.0),      # (Chebyshev-node posterior K1t on the shards)
                                                      (4, "H", 300, [384, 512], 3.0)])           # (a rank without a single safe candidate)
def test_multi_rank_large_grid_matches_single_rank(engine, tmp_path, world, cfg_name, n, count, b):
    """Shards big enough for the coarse cell bounds inside each rank's halo window: every mask and index must equal the
    single-rank sweep of the whole grid (itself pinned to the oracle by the tests above)."""
    port, out = _free_port(), str(tmp_path / "res.json")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gpu_rank_worker.py"), str(r), str(world), port, out, cfg_name,
                               str(n), json.dumps(count), str(b)]) for r in range(world)]
    try:
        cfg = synthetic.make_config(cfg_name, n=n)
        engine.set_model(cfg["ds"])
        engine.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], count)
        ref = engine.sweep_safeopt(b, want_masks=True)
        rmask = {k: engine.mask(k) for k in ("S", "U", "M")}
        rmask.update({f"G{c}": engine.mask("G", c) for c in range(1, cfg["q"])})
        gref = engine.sweep_goose(b, want_masks=True, posterior_ready=True)
        rmask.update({f"O{c}": engine.mask("O", c) for c in range(1, cfg["q"])})
    except BaseException:
        for p in procs:
            p.kill()
            p.wait()
        raise
    assert _wait_ranks(procs) == [0] * world
    res = json.load(open(out))
    parts = [np.load(out + f".rank{r}.npz") for r in range(world)]
    for k, want in rmask.items():
        assert np.array_equal(np.concatenate([p[k] for p in parts]), want), k
    assert rmask["O1"].any() and rmask["G1"].any()
    for k in ("minimizer_index", "expander_index", "count_S", "count_M"):
        assert res[k] == ref[k], k
    assert res["count_G"] == ref["count_G"].tolist() and res["u_star"] == ref["u_star"]
    # the guard band across ranks: what one rank decides outright, the shards decide outright (r04: a rank without safe candidates
    # used to put a band of sqrt(dv) into the max over the ranks, and every sweep of config H on four ranks took the slow path)
    assert ref["guard_band"] == 0 and res["guard_band"] == 0 and res["guard_passes"] == 0, (res["guard_band"], res["guard_passes"])
    if cfg_name == "H":
        assert not any(np.load(out + f".rank{r}.npz")["S"].any() for r in (world - 1,)), "the last rank was meant to hold no safe candidate"
    g = res["goose"]
    for k in ("safe_min_index", "target_index", "explore_index", "choose_safe_min", "target_best_c"):
        assert g[k] == gref[k], k
    assert g["count_O"] == gref["count_O"].tolist() and g["target_index_c"] == gref["target_index_c"].tolist()
    if count == [64, 64, 64, 64]:
        assert res["posterior_kernel"] == 5
    if res["posterior_kernel"] == 5:
        # (r04) K1t across ranks: each rank evaluates a slab of the Chebyshev nodes and one all-gather puts the node tensors
        # together -- the sweep's collectives carry at least this rank's slab (12 quantities x >= 32^3 x 16 nodes x 8 bytes)
        assert res["comm_bytes"] > 12 * 32 ** 3 * 16 * 8, res["comm_bytes"]


@pytest.mark.parametrize("world,cfg_name,n,count,b", [(2, "B", 128, [96, 81], 3.0), (3, "C", 64, [72, 65], 2.0)])
def test_multi_rank_fp32_sweep_with_fp64_recheck_equals_the_fp64_oracle(tmp_path, world, cfg_name, n, count, b):
    """fp32 models across ranks: the interval of u* and the variance guards are global quantities (three keys through the
    collectives), the undecided lists are per rank, and the number of set-phase passes is agreed on by all ranks -- every mask
    and index of the sharded fp32 sweep must be the fp64 oracle's."""
    port, out = _free_port(), str(tmp_path / "res.json")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gpu_rank_worker.py"), str(r), str(world), port, out, cfg_name + ":f32",
                               str(n), json.dumps(count), str(b)]) for r in range(world)]
    assert _wait_ranks(procs) == [0] * world
    cfg = synthetic.make_config(cfg_name, n=n)
    pts = oracle.grid_points(cfg["bound"][:, 0], cfg["bound"][:, 1], count)
    ref = oracle.safeopt_sweep(pts, cfg["ds"], b)
    res = json.load(open(out))
    parts = [np.load(out + f".rank{r}.npz") for r in range(world)]
    for k in ("S", "U", "M"):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), ref[k]), k
    for c in range(1, cfg["q"]):
        assert np.array_equal(np.concatenate([p[f"G{c}"] for p in parts]), ref["G"][c - 1]), f"G{c}"
    assert res["minimizer_index"] == ref["minimizer_index"] and res["expander_index_c"][:cfg["q"] - 1] == [int(x) for x in ref["expander_index"]]
    assert res["u_star"] == pytest.approx(ref["u_star"], rel=1e-10) and res["fp64_rechecks"] > 0
    assert np.allclose(res["L"][:cfg["q"]], ref["L"], rtol=1e-9)
    # (r04) the GoOSE and trust-region sweeps of the fp32 model across the ranks: rechecked in fp64 like the SafeOpt sweep --
    # optimistic sets, targets and the trust-region argmin are the fp64 oracle's, whatever the world size
    _assert_goose_tr(res, parts, cfg, pts, b)


def _assert_goose_tr(res, parts, cfg, pts, b):
    gref = oracle.goose_sweep(pts, cfg["ds"], b)
    g = res["goose"]
    assert g is not None and not g.get("empty_safe_set", False)
    for c in range(1, cfg["q"]):
        assert np.array_equal(np.concatenate([p[f"O{c}"] for p in parts]), gref["O"][c - 1]), f"O{c}"
    assert g["safe_min_index"] == gref["safe_min_index"]
    assert g["target_index_c"][:cfg["q"] - 1] == [int(x) for x in gref["target_index_c"]]
    assert g["target_index"] == gref["target_index"] and g["explore_index"] == gref["explore_index"]
    assert g["choose_safe_min"] == gref["choose_safe_min"] and g["count_O"][:cfg["q"] - 1] == [int(x) for x in gref["O"].sum(1)]
    x0 = cfg["bound"].mean(axis=1)
    tref = oracle.tr_sweep(pts, cfg["ds"], b, x0, 0.3 * float(np.min(cfg["bound"][:, 1] - cfg["bound"][:, 0])))
    assert res["tr"]["index"] == tref["index"] and res["tr"]["count_T"] == int(tref["T"].sum())


@pytest.mark.parametrize("world,cfg_name,n,count,b", [(2, "B", 128, [300, 257], 3.0), (3, "C", 64, [272, 265], 2.0),
                                                      (2, "D", 128, [64, 64, 64, 64], 3.0)])
def test_multi_rank_forced_guard_reevaluation_equals_the_exact_kernel(engine, tmp_path, world, cfg_name, n, count, b):
    """(r04) The guard band of the approximating posteriors (K1b on the 2-D shards, K1t on the 4-D ones) across ranks: with
    guard_band = 2 every sweep -- SafeOpt, GoOSE, trust region -- re-evaluates its in-band candidates with the exact kernel;
    band widths and pass counts are agreed on through the collectives, and every mask and index is what ONE rank gets with
    the exact table kernel K1g on the whole grid (itself pinned to the oracle by the single-rank tests)."""
    port, out = _free_port(), str(tmp_path / "res.json")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gpu_rank_worker.py"), str(r), str(world), port, out, cfg_name + ":guard",
                               str(n), json.dumps(count), str(b)]) for r in range(world)]
    try:
        cfg = synthetic.make_config(cfg_name, n=n)
        engine.set_option("bilinear", 0)
        engine.set_option("tensor_cheb", 0)
        engine.set_model(cfg["ds"])
        engine.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], count)
        ref = engine.sweep_safeopt(b, want_masks=True)
        assert engine.profile()["posterior_kernel"] == 3
        rmask = {k: engine.mask(k) for k in ("S", "U", "M")}
        rmask.update({f"G{c}": engine.mask("G", c) for c in range(1, cfg["q"])})
        gref = engine.sweep_goose(b, want_masks=True, posterior_ready=True)
        rmask.update({f"O{c}": engine.mask("O", c) for c in range(1, cfg["q"])})
        x0 = cfg["bound"].mean(axis=1)
        tref = engine.sweep_tr(b, x0, 0.3 * float(np.min(cfg["bound"][:, 1] - cfg["bound"][:, 0])), posterior_ready=True)
    except BaseException:
        for p in procs:
            p.kill()
            p.wait()
        raise
    finally:
        engine.set_option("bilinear", 1)
        engine.set_option("tensor_cheb", 1)
    assert _wait_ranks(procs) == [0] * world
    res = json.load(open(out))
    assert res["posterior_kernel"] in (4, 5, 6), "the shards must run an approximating posterior for this test to mean anything"
    assert res["guard_passes"] >= 1
    parts = [np.load(out + f".rank{r}.npz") for r in range(world)]
    for k, want in rmask.items():
        assert np.array_equal(np.concatenate([p[k] for p in parts]), want), k
    for k in ("minimizer_index", "expander_index", "count_S", "count_M"):
        assert res[k] == ref[k], k
    assert res["count_G"] == ref["count_G"].tolist() and res["expander_index_c"] == ref["expander_index_c"].tolist()
    g = res["goose"]
    for k in ("safe_min_index", "target_index", "explore_index", "choose_safe_min", "target_best_c"):
        assert g[k] == gref[k], k
    assert g["count_O"] == gref["count_O"].tolist() and g["target_index_c"] == gref["target_index_c"].tolist()
    assert res["tr"]["index"] == tref["index"] and res["tr"]["count_T"] == tref["count_T"]


def test_multi_rank_uneven_shards_pick_one_posterior_kernel(engine, tmp_path):
    """(r05) 31 lines over two ranks are shards of 15 and 16 lines: 16 qualify for the GEMM posterior (K1b / K1i), 15 do
    not.  The recheck behind an approximating posterior contains collectives, so the choice must be ONE decision of all ranks (the
    smallest shard decides) -- a rank that alone took the exact kernel used to return while the other waited in the recheck's
    all-reduce.  Forced re-evaluation (guard_band = 2): the run must end, every rank on the same kernel, the result the exact one."""
    world, cfg_name, n, count, b = 2, "B", 128, [256, 31], 3.0
    port, out = _free_port(), str(tmp_path / "res.json")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gpu_rank_worker.py"), str(r), str(world), port, out, cfg_name + ":guard",
                               str(n), json.dumps(count), str(b)]) for r in range(world)]
    try:
        cfg = synthetic.make_config(cfg_name, n=n)
        engine.set_option("bilinear", 0)
        engine.set_model(cfg["ds"])
        engine.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], count)
        ref = engine.sweep_safeopt(b, want_masks=True)
        rmask = {k: engine.mask(k) for k in ("S", "U", "M")}
        rmask["G1"] = engine.mask("G", 1)
    except BaseException:
        for p in procs:
            p.kill()
            p.wait()
        raise
    finally:
        engine.set_option("bilinear", 1)
    assert _wait_ranks(procs, timeout=150) == [0] * world
    res = json.load(open(out))
    parts = [np.load(out + f".rank{r}.npz") for r in range(world)]
    assert [int(p["n_local"]) // count[0] for p in parts] == [15, 16]
    assert len({int(p["kernel"]) for p in parts}) == 1, [int(p["kernel"]) for p in parts]
    for k, want in rmask.items():
        assert np.array_equal(np.concatenate([p[k] for p in parts]), want), k
    for k in ("minimizer_index", "expander_index", "count_S", "count_M"):
        assert res[k] == ref[k], k


@pytest.mark.parametrize("world,cfg_name,n,count,bs", [(2, "B", 128, [320, 600], [2.0, 2.0, 3.5, 3.5, 1.0]),
                                                        (3, "C", 64, [256, 300], [2.0, 2.0, 3.0, 1.5])])
def test_multi_rank_speculative_halo_has_no_wait_inside_the_sweep(engine, tmp_path, world, cfg_name, n, count, bs):
    """Ranks > 1: a sweep sizes the halo windows of its transforms from the global keys of the PREVIOUS sweep (plus a quarter)
    instead of waiting for its own (option halo_spec) -- one host synchronisation per sweep, the result read-back; the device
    checks the guess against this sweep's keys and a window that is too narrow (radii grown: larger b) reruns the set phase
    the waiting way.  Every sweep of the sequence must equal the single-rank sweep of the whole grid with the same b."""
    port, out = _free_port(), str(tmp_path / "seq.json")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gpu_rank_worker.py"), str(r), str(world), port, out, cfg_name,
                               str(n), json.dumps(count), json.dumps(bs)]) for r in range(world)]
    try:
        cfg = synthetic.make_config(cfg_name, n=n)
        q = cfg["q"]
        engine.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], count)
        refs = []
        for b in bs:
            engine.set_model(cfg["ds"])
            r = engine.sweep_safeopt(b, want_masks=True)
            m = {k: engine.mask(k) for k in ("S", "M")}
            m.update({f"G{c}": engine.mask("G", c) for c in range(1, q)})
            g = engine.sweep_goose(b, want_masks=True, posterior_ready=True)
            m.update({f"O{c}": engine.mask("O", c) for c in range(1, q)})
            refs.append((r, g, m))
    except BaseException:
        for p in procs:
            p.kill()
            p.wait()
        raise
    assert _wait_ranks(procs) == [0] * world
    rows = json.load(open(out))
    parts = [np.load(out + f".rank{r}.npz") for r in range(world)]
    for i, (row, (r, g, m)) in enumerate(zip(rows, refs)):
        for k, want in m.items():
            assert np.array_equal(np.concatenate([p[f"{i}_{k}"] for p in parts]), want), (i, k)
        for k in ("minimizer_index", "expander_index", "count_S", "count_M", "u_star"):
            assert row[k] == r[k], (i, k)
        assert row["count_G"] == r["count_G"].tolist() and row["count_O"] == g["count_O"].tolist()
        assert row["target_index"] == g["target_index"] and row["explore_index"] == g["explore_index"], i
    # the first sweep waits for its keys; a repeated b rides on the previous sweep's keys: the read-back is the only wait
    assert rows[0]["host_syncs"] >= 2
    assert rows[1]["host_syncs"] == 1 and rows[3]["host_syncs"] == 1, [x["host_syncs"] for x in rows]
    assert rows[-1]["host_syncs"] == 1                  # smaller radii: the old window is wider than needed, still exact


def test_multi_rank_short_halo_guess_on_one_rank_reruns_every_rank(engine, tmp_path):
    """the rerun after a speculative window that was too narrow is a GLOBAL decision (the flag travels in the C3 row):
    rank 1 alone guesses one plane (test hook SBO_TEST_HALO_SHORT) -- the others' windows are fine --, and still every rank runs
    the set phase again, the job neither hangs nor diverges, the results equal the single-rank sweep, and the profile says what
    happened (halo_reruns, with the discarded pass's collectives and waits kept in the counters).  SafeOpt sweeps only: the
    sizes of the GoOSE sweep's all-gather follow the window, so ranks whose guesses differ -- which the library never produces:
    guesses are functions of the global keys alone -- cannot even enter it."""
    world, cfg_name, n, count, bs = 3, "C", 64, [256, 300], [2.0, 2.0, 2.0]
    port, out = _free_port(), str(tmp_path / "seq.json")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gpu_rank_worker.py"), str(r), str(world), port, out, cfg_name,
                               str(n), json.dumps(count), json.dumps(bs)],
                              env={**os.environ, "SBO_TEST_SEQ_SAFEOPT_ONLY": "1", **({"SBO_TEST_HALO_SHORT": "1"} if r == 1 else {})})
             for r in range(world)]
    try:
        cfg = synthetic.make_config(cfg_name, n=n)
        engine.set_grid(cfg["bound"][:, 0], cfg["bound"][:, 1], count)
        engine.set_model(cfg["ds"])
        r = engine.sweep_safeopt(bs[0], want_masks=True)
        m = {k: engine.mask(k) for k in ("S", "M")}
        m.update({f"G{c}": engine.mask("G", c) for c in range(1, cfg["q"])})
    except BaseException:
        for p in procs:
            p.kill()
            p.wait()
        raise
    assert _wait_ranks(procs) == [0] * world
    rows = json.load(open(out))
    parts = [np.load(out + f".rank{r}.npz") for r in range(world)]
    for i, row in enumerate(rows):
        for k, want in m.items():
            assert np.array_equal(np.concatenate([p[f"{i}_{k}"] for p in parts]), want), (i, k)
        assert row["minimizer_index"] == r["minimizer_index"] and row["expander_index"] == r["expander_index"], i
    # rank 0 reports: its own guess was fine, and yet sweeps 2 and 3 (the speculative ones) ran their set phase twice
    assert rows[0]["halo_reruns"] == 0
    assert rows[1]["halo_reruns"] == 1 and rows[2]["halo_reruns"] == 1, [x["halo_reruns"] for x in rows]
    assert rows[1]["host_syncs"] >= 2 and rows[1]["comm_calls"] > rows[0]["comm_calls"]


@pytest.mark.parametrize("cfg_name,n,count", [("B", 128, [320, 300]), ("C", 64, [256, 300]), ("D", 128, [20, 18, 17, 24])])
def test_rccl_collectives_on_a_one_rank_communicator(engine, cfg_name, n, count):
    """The RCCL calls themselves on the one GPU of the test box: a genuine one-rank communicator (ncclCommInitRank) and
    option "comm_selftest", which makes the sweeps take their multi-rank path -- C1 ncclAllReduce(ncclUint64, ncclMax, in
    place), C2 ncclAllGather of the packed U mask (and of the GoOSE source slabs), C3 ncclAllReduce(ncclDouble, ncclSum)
    with the host merge.  Results must equal the plain single-rank sweep of the same grid bit for bit."""
    cfg = synthetic.make_config(cfg_name, n=n)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    q = cfg["q"]
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, count)
    ref = engine.sweep_safeopt(cfg["b"], want_masks=True)
    rmask = {k: engine.mask(k) for k in ("S", "U", "M")}
    rmask.update({f"G{c}": engine.mask("G", c) for c in range(1, q)})
    gref = engine.sweep_goose(cfg["b"], want_masks=True, posterior_ready=True)
    rmask.update({f"O{c}": engine.mask("O", c) for c in range(1, q)})
    tref = engine.sweep_tr(cfg["b"], ref["minimizer_x"], 0.7, posterior_ready=True)
    with safebo_amd.SweepEngine(0) as eng:
        with pytest.raises(ValueError, match="needs a communicator"):
            eng.set_option("comm_selftest", 1)
        eng.comm_init(1, 0, eng.comm_unique_id())
        eng.set_option("comm_selftest", 1)
        eng.comm_barrier()
        eng.set_model(cfg["ds"])
        eng.set_grid_sharded(lo, hi, count)
        assert (eng.first, eng.n_local) == (0, int(np.prod(count)))
        eng.set_option("comm_events", 1)                         # an event pair around every RCCL call (sbo_profile.comm_*)
        res = eng.sweep_safeopt(cfg["b"], want_masks=True)
        prof = eng.profile()
        assert prof["comm_calls"] == 2 and prof["comm_bytes"] > 0 and 0.0 < prof["comm_ms"] < 50.0, prof
        eng.set_option("comm_events", 0)
        for k in ("S", "U", "M"):
            assert np.array_equal(eng.mask(k), rmask[k]), k
        for c in range(1, q):
            assert np.array_equal(eng.mask("G", c), rmask[f"G{c}"]), f"G{c}"
        for k in ("minimizer_index", "expander_index", "count_S", "count_U", "count_M", "u_star", "minimizer_std", "expander_best_c",
                  "n_exact_rechecks"):
            assert res[k] == ref[k], k
        assert np.array_equal(res["L"], ref["L"]) and np.array_equal(res["count_G"], ref["count_G"])
        g = eng.sweep_goose(cfg["b"], want_masks=True, posterior_ready=True)
        for c in range(1, q):
            assert np.array_equal(eng.mask("O", c), rmask[f"O{c}"]), f"O{c}"
        for k in ("safe_min_index", "target_index", "explore_index", "choose_safe_min", "target_best_c", "safe_min_lcb"):
            assert g[k] == gref[k], k
        assert np.array_equal(g["count_O"], gref["count_O"])
        t = eng.sweep_tr(cfg["b"], ref["minimizer_x"], 0.7, posterior_ready=True)
        assert (t["index"], t["count_T"], t["lcb"]) == (tref["index"], tref["count_T"], tref["lcb"])


def test_comm_init_failure_leaves_the_context_single_rank(engine):
    """A communicator that cannot be formed (here: rank out of range / no id) must not leave world > 1 behind with no
    transport (the next sweep then called a NULL relay callback)."""
    with safebo_amd.SweepEngine(0) as eng:
        with pytest.raises(ValueError):
            eng.comm_init(2, 5, eng.comm_unique_id())
        with pytest.raises(ValueError):
            eng.comm_init(2, 0, None)
        cfg = synthetic.make_config("A")
        eng.set_model(cfg["ds"])
        eng.set_grid_sharded(cfg["bound"][:, 0], cfg["bound"][:, 1], [50, 50])
        assert (eng.world, eng.n_local) == (1, 2500)
        assert eng.sweep_safeopt(cfg["b"])["count_S"] > 0


@pytest.mark.parametrize("use_invK", [True, False])
@pytest.mark.parametrize("n", [512, 300, 128])
def test_blocked_model_build_repeats_bitwise(engine, use_invK, n):
    """n >= 96 builds the model with the multi-workgroup blocked factorisation (one launch per panel, workgroups sharing the
    panel's diagonal block and the pending update of the previous panel): repeated builds of one data set must give the same
    posterior bit for bit (a workgroup that read a block another one had already factored once made every few builds fail),
    full and ragged last panels (300 = 9 x 32 + 12)."""
    cfg = synthetic.make_config("H", n=n)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    first = None
    for rep in range(8):
        engine.set_model(cfg["ds"], use_invK=use_invK)
        engine.set_grid(lo, hi, [48, 40])
        mean, var = engine.posterior()
        if first is None:
            first = (mean, var)
            _check_posterior(engine, cfg["ds"], oracle.grid_points(lo, hi, [48, 40]), TOL64)
        else:
            assert np.array_equal(mean, first[0]) and np.array_equal(var, first[1]), rep


@pytest.mark.parametrize("cfg_name,n,count", [("H", 512, [96, 80]), ("B", 128, [96, 80]), ("C", 256, [80, 96]), ("H", 300, [70, 66])])
def test_caller_invK_tables_and_deferred_factor(engine, cfg_name, n, count):
    """A caller's invK on a grid the GEMM posterior takes (option chol_async, default on): the tables contract with invK as
    given (W = invK Z, models/GP_Safe.py:341-343) and the reverse Cholesky factor -- needed by the O(n^2) kernels and by
    sbo_model_append only -- is built on a side stream after sbo_model_set has returned.  (a) The posterior equals the
    factor-based tables (chol_async = 0) to rounding and the oracle within the bar; (b) a consumer of the factor right behind
    the model change (K1g posterior; an append) waits for it and is correct; (c) a new model right behind a model change
    (its deferred chain still running) is correct too."""
    cfg = synthetic.make_config(cfg_name, n=n)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    pts = oracle.grid_points(lo, hi, count)
    engine.set_grid(lo, hi, count)
    engine.set_model(cfg["ds"])                                  # deferred factor, direct tables
    m1, v1 = _check_posterior(engine, cfg["ds"], pts, TOL64)
    assert engine.profile()["posterior_kernel"] in (4, 6)
    try:
        engine.set_option("chol_async", 0)
        engine.set_model(cfg["ds"])
        m0, v0 = _check_posterior(engine, cfg["ds"], pts, TOL64)
        assert engine.profile()["posterior_kernel"] in (4, 6)
    finally:
        engine.set_option("chol_async", 1)
    ys = np.maximum(1.0, cfg["ds"]["Y_std"])
    assert np.max(np.abs(m0 - m1) / ys) < 1e-11 and np.max(np.abs(v0 - v1) / ys ** 2) < 1e-11
    # (b) the factor's consumers right behind the model change
    try:
        engine.set_model(cfg["ds"])
        engine.set_option("bilinear", 0)                         # K1g: contracts with the factor images
        mg, vg = _check_posterior(engine, cfg["ds"], pts, TOL64)
        assert engine.profile()["posterior_kernel"] == 3
    finally:
        engine.set_option("bilinear", 1)
    assert np.max(np.abs(mg - m1) / ys) < 1e-11 and np.max(np.abs(vg - v1) / ys ** 2) < 1e-11
    # (c) model changes back to back, then a sweep against the oracle
    alt = synthetic.make_config(cfg_name, n=n, seed=synthetic.SEED0 + 321)
    for ds in (alt["ds"], cfg["ds"], alt["ds"], cfg["ds"]):
        engine.set_model(ds)
    res = engine.sweep_safeopt(cfg["b"], want_masks=True)
    ref = oracle.safeopt_sweep(pts, cfg["ds"], cfg["b"])
    assert np.array_equal(engine.mask("S"), ref["S"]) and np.array_equal(engine.mask("M"), ref["M"])
    assert res["minimizer_index"] == ref["minimizer_index"] and list(res["expander_index_c"]) == list(ref["expander_index"])


def test_deferred_factor_reports_an_indefinite_invK_where_it_is_needed(engine):
    """With the factor deferred, sbo_model_set cannot say that invK is not positive definite -- the GEMM posterior does not need
    it to be (neither does the reference: var is clipped at 0, models/GP_Safe.py:343).  The O(n^2) kernels do: the error
    surfaces, as ValueError, at the first call that needs the factor."""
    cfg = synthetic.make_config("B", n=128)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    ds = dict(cfg["ds"])
    bad = [a.copy() for a in ds["invKopt"]]
    bad[1][5, 5] = -abs(bad[1][5, 5])                             # an indefinite "inverse"
    ds["invKopt"] = bad
    engine.set_grid(lo, hi, [96, 80])
    engine.set_model(ds)                                          # accepted: nothing on this path factors it
    mean, var = engine.posterior()
    assert engine.profile()["posterior_kernel"] in (4, 6) and np.all(var >= 0) and np.all(np.isfinite(mean))
    om, ov = oracle.gp_inference(oracle.grid_points(lo, hi, [96, 80]), ds)
    assert _nerr(mean, om, ds["Y_std"], 1) < 1e-9 and _nerr(var, ov, ds["Y_std"], 2) < 1e-9
    try:
        engine.set_option("bilinear", 0)
        with pytest.raises(ValueError, match="not positive definite"):
            engine.posterior_run()
    finally:
        engine.set_option("bilinear", 1)
    try:
        engine.set_option("chol_async", 0)
        with pytest.raises(ValueError, match="not positive definite"):
            engine.set_model(ds)
    finally:
        engine.set_option("chol_async", 1)
    engine.set_model(cfg["ds"])                                   # and the context is usable again


@pytest.mark.parametrize("use_invK,dtype", [(True, "f64"), (False, "f64"), (False, "f32")])
def test_model_append_equals_a_rebuild(engine, use_invK, dtype):
    """sbo_model_append: five observations appended one by one under frozen normalisation and
    hyper-parameters give the posterior of a model rebuilt from all observations with the same constants."""
    cfg = synthetic.make_config("B", n=100)
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], [72, 66]
    ds = cfg["ds"]
    rng = np.random.default_rng(21)
    Xnew = rng.uniform(lo, hi, size=(5, 2))
    Ynew = synthetic.benoit(Xnew)
    xn = (Xnew - ds["X_mean"]) / ds["X_std"]
    yn = (Ynew - ds["Y_mean"]) / ds["Y_std"]
    engine.set_model(ds, dtype=dtype, use_invK=use_invK)
    for i in range(5):
        engine.append_sample(xn[i], yn[i])
    assert engine.n == 105
    engine.set_grid(lo, hi, count)
    mean, var = engine.posterior()
    # the same model built in one go: normalised data extended with the frozen constants, inverse recomputed
    X_norm = np.vstack([ds["X_norm"], xn])
    Y_norm = np.vstack([ds["Y_norm"], yn])
    ds2 = dict(ds)
    ds2["X_norm"], ds2["Y_norm"] = X_norm, Y_norm
    ds2["invKopt"] = oracle.build_invK(X_norm, ds["hypopt"])
    pts = oracle.grid_points(lo, hi, count)
    om, ov = oracle.gp_inference(pts, ds2)
    tol = TOL64 if dtype == "f64" else TOL32
    assert _nerr(mean, om, ds["Y_std"], 1) < tol and _nerr(var, ov, ds["Y_std"], 2) < tol
    engine.set_model(ds2, dtype=dtype, use_invK=use_invK)
    engine.set_grid(lo, hi, count)
    m2, v2 = engine.posterior()
    assert _nerr(mean, m2, ds["Y_std"], 1) < tol and _nerr(var, v2, ds["Y_std"], 2) < tol
    res_a = engine.sweep_safeopt(cfg["b"])
    assert res_a["count_S"] > 0


def test_wo_plant_on_the_device_matches_the_oracle_and_the_reference_table(engine):
    """sbo_plant_wo: the batched steady-state solve against the NumPy restatement (same Newton
    iteration: 1e-12) and against the reference project's own 100 x 100 table (made with fsolve: 1e-5 / 1e-8)."""
    from oracle import plants as oplants
    from safebo_amd.plants import WilliamOttoReactor
    z = np.load(os.path.join(HERE, "golden", "wo_contour_reference.npz"), allow_pickle=False)
    U = np.stack([z["X_0"].ravel(), z["X_1"].ravel()], axis=1)
    Y = engine.plant_wo(U)
    ref = oplants.wo_outputs(U)
    assert np.max(np.abs(Y - ref) / np.maximum(1.0, np.abs(ref))) < 1e-12
    assert np.max(np.abs(Y[:, 0] - z["Y_objective"].ravel())) < 1e-5
    assert np.max(np.abs(Y[:, 1] - z["Y_constraint1"].ravel())) < 1e-8
    assert np.max(np.abs(Y[:, 2] - z["Y_constraint2"].ravel())) < 1e-8
    plant = WilliamOttoReactor(engine=engine)
    u = np.array([4.8, 83.0])
    assert plant.get_objective(u) == pytest.approx(float(oplants.wo_outputs(u[None])[0, 0]), rel=1e-12)
    assert plant.get_constraint1(u) == pytest.approx(float(oplants.wo_outputs(u[None])[0, 1]), abs=1e-14)
    rng = np.random.default_rng(3)
    big = np.stack([rng.uniform(4, 7, 200000), rng.uniform(70, 100, 200000)], axis=1)
    Yb = engine.plant_wo(big)
    assert np.isfinite(Yb).all()
    sub = rng.choice(200000, 500, replace=False)
    assert np.max(np.abs(Yb[sub] - oplants.wo_outputs(big[sub])) / np.maximum(1.0, np.abs(Yb[sub]))) < 1e-12


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16, 17, 18])
def test_random_models_full_sweeps_against_the_oracle(engine, seed):
    """Random models (n, hyper-parameters, b) on a 72 x 70 grid -- large enough for the GEMM posterior (K1b) whenever it
    pays, small enough for the oracle's brute-force sets: every SafeOpt and GoOSE mask and index against the oracle.
    A mask bit may differ from the oracle's only where the deciding bound is within the posterior tolerance of zero
    (the oracle's own rounding decides those); none of these seeds has such a candidate, so equality is exact."""
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(40, 220))
    cfg = synthetic.make_config("B" if seed % 2 else "C", n=n, seed=7100 + seed)
    q = cfg["Y"].shape[1]
    hyp = np.empty((4, q))
    hyp[:2] = rng.uniform(-0.8, 0.6, size=(2, q))
    hyp[2] = rng.uniform(-0.5, 0.5, size=q)
    hyp[3] = rng.uniform(-2.5, -2.0, size=q)
    ds = synthetic.make_dataset(cfg["X"], cfg["Y"], hyp)
    b = float(rng.uniform(1.0, 3.0))
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], [72, 70]
    pts = oracle.grid_points(lo, hi, count)
    engine.set_model(ds)
    engine.set_grid(lo, hi, count)
    mean, var = _check_posterior(engine, ds, pts, TOL64)
    ref = oracle.safeopt_sweep(pts, ds, b)
    if ref["empty_safe_set"]:
        with pytest.raises(safebo_amd.EmptySafeSetError):
            engine.sweep_safeopt(b, posterior_ready=True)
        return
    res = engine.sweep_safeopt(b, want_masks=True, posterior_ready=True)
    lcb_min = np.min(np.abs(ref["lcb"][:, 1:]), axis=1)
    for k in ("S", "U", "M"):
        diff = engine.mask(k) != ref[k]
        assert not diff.any(), (k, int(diff.sum()), float(lcb_min[diff].max()) if diff.any() else 0.0)
    for c in range(1, q):
        assert np.array_equal(engine.mask("G", c), ref["G"][c - 1]), f"G{c}"
    assert res["minimizer_index"] == ref["minimizer_index"]
    assert list(res["expander_index_c"]) == list(ref["expander_index"])
    gref = oracle.goose_sweep(pts, ds, b)
    g = engine.sweep_goose(b, want_masks=True, posterior_ready=True)
    for c in range(1, q):
        assert np.array_equal(engine.mask("O", c), gref["O"][c - 1]), f"O{c}"
    assert (g["safe_min_index"], g["target_index"], g["explore_index"]) == (gref["safe_min_index"], gref["target_index"],
                                                                             gref["explore_index"])


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_random_models_large_grid_paths_agree(engine, seed):
    """Random models on a 512 x 2048 grid (1 M candidates: every blocked / listed / LDS-line kernel of the set phase is
    on): the expander masks of the blocked + listed scans equal those of the plain step-by-step scans, the optimistic
    masks of the power transform equal the pruned exact pair evaluation, and the oracle's predicates re-decide a sample
    on both sides of the boundaries of G_1 and O_1."""
    rng = np.random.default_rng(8000 + seed)
    n = int(rng.integers(48, 200))
    cfg = synthetic.make_config("B" if seed % 2 else "C", n=n, seed=8100 + seed)
    q = cfg["Y"].shape[1]
    hyp = np.empty((4, q))
    hyp[:2] = rng.uniform(-0.8, 0.6, size=(2, q))
    hyp[2] = rng.uniform(-0.5, 0.5, size=q)
    hyp[3] = rng.uniform(-2.5, -2.0, size=q)
    ds = synthetic.make_dataset(cfg["X"], cfg["Y"], hyp)
    b = float(rng.uniform(1.0, 3.0))
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], [512, 2048]
    engine.set_model(ds)
    engine.set_grid(lo, hi, count)
    engine.posterior_run()
    out = {}
    try:
        for plain in (0, 1):
            engine.set_option("scan_blocks", 0 if plain else 1)
            engine.set_option("scan_waves", 0 if plain else 1)
            engine.set_option("goose_pairs", plain)
            try:
                s_ = engine.sweep_safeopt(b, want_masks=True, posterior_ready=True)
            except safebo_amd.EmptySafeSetError:
                return
            G = [engine.mask("G", c) for c in range(1, q)]
            g_ = engine.sweep_goose(b, want_masks=True, posterior_ready=True)
            out[plain] = (s_, G, g_, [engine.mask("O", c) for c in range(1, q)])
    finally:
        engine.set_option("scan_blocks", 1)
        engine.set_option("scan_waves", 1)
        engine.set_option("goose_pairs", 0)
    for c in range(q - 1):
        assert np.array_equal(out[0][1][c], out[1][1][c]), f"G{c + 1}"
        assert np.array_equal(out[0][3][c], out[1][3][c]), f"O{c + 1}"
    assert list(out[0][0]["expander_index_c"]) == list(out[1][0]["expander_index_c"])
    for k in ("safe_min_index", "target_index", "explore_index"):
        assert out[0][2][k] == out[1][2][k], k
    # oracle predicates on samples around the boundaries (reference quirk: every constraint uses L of the last output)
    S, U = engine.mask("S"), engine.mask("U")
    pts = oracle.grid_points(lo, hi, count)
    Lq = out[0][0]["L"][q - 1]
    ucb1 = engine.bounds(b, 1, "ucb")
    G1, O1 = out[0][1][0], out[0][3][0]
    xu, xs, us = pts[U], pts[S], ucb1[S]
    for mask, dom in ((G1, S), (O1, U)):
        edge = np.nonzero((mask[:-1] != mask[1:]) & dom[:-1] & dom[1:])[0]
        pick = rng.choice(edge, size=min(12, edge.size), replace=False) if edge.size else np.array([], dtype=int)
        for i in np.concatenate([pick, pick + 1]):
            if mask is G1:
                want = bool(np.any(ucb1[i] - Lq * oracle.shifted_norm(pts[i][None, :], xu) >= 0))
            else:
                want = bool(np.any(us - Lq * oracle.shifted_norm(xs, pts[i][None, :]) >= 0))
            assert bool(mask[i]) == want, ("G" if mask is G1 else "O", int(i))


def test_long_last_axis_blocks_of_64_steps(engine):
    """Weak-scaling grids get finer along the slowest axis only: from 8192 planes on the last-axis scans use blocks of 64
    steps (four rounds of a 16-lane group per block).  Blocked + listed scans against the plain step-by-step scans on a
    256 x 8192 grid, SafeOpt expanders and GoOSE optimistic sets."""
    cfg = synthetic.make_config("B")
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], [256, 8192]
    engine.set_model(cfg["ds"])
    engine.set_grid(lo, hi, count)
    engine.posterior_run()
    keep = {}
    try:
        for plain in (0, 1):
            engine.set_option("scan_blocks", 0 if plain else 1)
            engine.set_option("scan_waves", 0 if plain else 1)
            s_ = engine.sweep_safeopt(cfg["b"], want_masks=True, posterior_ready=True)
            G = engine.mask("G", 1)
            g_ = engine.sweep_goose(cfg["b"], want_masks=True, posterior_ready=True)
            keep[plain] = (G, engine.mask("O", 1), s_["expander_index_c"][0], g_["target_index"], g_["explore_index"])
    finally:
        engine.set_option("scan_blocks", 1)
        engine.set_option("scan_waves", 1)
    assert keep[0][0].any() and keep[0][1].any()
    assert np.array_equal(keep[0][0], keep[1][0]) and np.array_equal(keep[0][1], keep[1][1])
    assert keep[0][2:] == keep[1][2:]


@pytest.mark.parametrize("seed,count", [(31, [160, 140]), (32, [264, 256])])
def test_random_models_mid_grids_against_the_oracle(engine, seed, count):
    """The oracle's brute-force sets on grids large enough for the blocked last-axis scans and the listed open candidates
    (160 x 140: every safe / unsafe candidate goes through the list kernels; 264 x 256: the coarse bounds decide most and
    list the rest): every SafeOpt and GoOSE mask and index of a random model, bit for bit."""
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.integers(40, 160))
    cfg = synthetic.make_config("B", n=n, seed=9100 + seed)
    q = cfg["Y"].shape[1]
    hyp = np.empty((4, q))
    hyp[:2] = rng.uniform(-0.8, 0.6, size=(2, q))
    hyp[2] = rng.uniform(-0.5, 0.5, size=q)
    hyp[3] = rng.uniform(-2.5, -2.0, size=q)
    ds = synthetic.make_dataset(cfg["X"], cfg["Y"], hyp)
    b = float(rng.uniform(1.5, 3.0))
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    pts = oracle.grid_points(lo, hi, count)
    engine.set_model(ds)
    engine.set_grid(lo, hi, count)
    _check_posterior(engine, ds, pts, TOL64)
    ref = oracle.safeopt_sweep(pts, ds, b)
    assert not ref["empty_safe_set"]
    res = engine.sweep_safeopt(b, want_masks=True, posterior_ready=True)
    for k in ("S", "U", "M"):
        assert np.array_equal(engine.mask(k), ref[k]), k
    assert np.array_equal(engine.mask("G", 1), ref["G"][0]) and ref["G"][0].any()
    assert res["minimizer_index"] == ref["minimizer_index"] and list(res["expander_index_c"]) == list(ref["expander_index"])
    gref = oracle.goose_sweep(pts, ds, b)
    g = engine.sweep_goose(b, want_masks=True, posterior_ready=True)
    assert np.array_equal(engine.mask("O", 1), gref["O"][0]) and gref["O"][0].any()
    assert (g["safe_min_index"], g["target_index"], g["explore_index"]) == (gref["safe_min_index"], gref["target_index"],
                                                                             gref["explore_index"])

