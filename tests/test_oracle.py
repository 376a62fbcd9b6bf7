"""CPU tests of the oracle (oracle/gp_oracle.py): what pins it, given that the reference's own tests assert
nothing and the reference cannot run here (no jax)."""
import glob
import os

import numpy as np
import pytest

import oracle
from safebo_amd import synthetic

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")) if "contour_reference" not in p and "nll_population" not in p)


def _small(n=12, d=2, q=2, seed=3):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, size=(n, d))
    Y = np.stack([np.sin(2 * X[:, 0]) + X[:, -1] ** 2 + 0.3 * i * X[:, 0] for i in range(q)], axis=1)
    hyp = synthetic.default_hypopt(d, q)
    hyp[:d] += rng.uniform(-0.3, 0.3, size=(d, q))
    return X, Y, hyp


def test_reference_adjacent_pins_benoit():
    # utils/utils_SafeOpt.py:31 optimum, test/test_GoOSE.py:182 value and acceptance band
    x = np.array([[0.36845785, -0.39299271]])
    y = synthetic.benoit(x)[0]
    assert abs(y[0] - 0.145249) <= 0.005
    assert abs(y[1]) < 1e-6           # the tight constraint is active at the optimum


def test_posterior_matches_independent_gp_implementation():
    """scikit-learn's GaussianProcessRegressor (Cholesky-based, fixed hyper-parameters) is an independent
    implementation of the same posterior: latent variance, RBF-ARD, noise sn2 + float32 eps on the diagonal."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel

    X, Y, hyp = _small(n=25, d=3, q=2)
    ds = oracle.make_inference_dataset(X, Y, hyp)
    pts = np.random.default_rng(0).uniform(-1.2, 1.2, size=(200, 3))
    mean, var = oracle.gp_inference(pts, ds)
    mp = oracle.mean_prior(ds)
    xn = (pts - ds["X_mean"]) / ds["X_std"]
    for i in range(2):
        ell = np.exp(hyp[:3, i])                      # length-scale = sqrt(exp(2h))
        sf2, sn2 = np.exp(2 * hyp[3, i]), np.exp(2 * hyp[4, i]) + oracle.FLOAT32_EPS
        gp = GaussianProcessRegressor(ConstantKernel(sf2, "fixed") * RBF(ell, "fixed"), alpha=sn2, optimizer=None)
        gp.fit(ds["X_norm"], ds["Y_norm"][:, i] - mp[i])
        m, s = gp.predict(xn, return_std=True)
        assert np.allclose((m + mp[i]) * ds["Y_std"][i] + ds["Y_mean"][i], mean[:, i], rtol=0, atol=1e-8)
        assert np.allclose(s ** 2 * ds["Y_std"][i] ** 2, var[:, i], rtol=0, atol=1e-8)


def test_prior_mean_and_far_field():
    # models/GP_Safe.py:331-332: far from data MEAN_0 -> Y_mean[0], MEAN_i -> -Y_mean[i] for constraints
    X, Y, hyp = _small()
    ds = oracle.make_inference_dataset(X, Y, hyp)
    mean, var = oracle.gp_inference(np.array([[1e3, 1e3]]), ds)
    assert np.allclose(mean[0, 0], ds["Y_mean"][0])
    assert np.allclose(mean[0, 1], -ds["Y_mean"][1])
    assert np.allclose(var[0], np.exp(2 * hyp[2]) * ds["Y_std"] ** 2)


def test_single_point_matches_batch_and_reference_shapes():
    X, Y, hyp = _small()
    ds = oracle.make_inference_dataset(X, Y, hyp)
    pts = np.random.default_rng(1).uniform(-1, 1, size=(7, 2))
    mean, var = oracle.gp_inference(pts, ds)
    for p in range(7):
        m1, v1 = oracle.gp_inference(pts[p:p + 1], ds)
        assert np.allclose(m1[0], mean[p], atol=1e-13) and np.allclose(v1[0], var[p], atol=1e-13)
    assert mean.shape == (7, 2) and var.shape == (7, 2)
    assert np.all(var >= 0)


def test_cov_mat_errors_follow_reference():
    with pytest.raises(ValueError):
        oracle.cov_mat(np.zeros((3, 2)), np.zeros((3, 2)), np.ones(3), 1.0)   # models/GP_Safe.py:134-135


def test_grid_order_is_meshgrid_xy_ravel():
    # test/test_SafeOpt.py:325-334: linspace, meshgrid (default 'xy'), ravel, column_stack
    lo, hi, count = [-0.6, -1.0], [1.5, 1.0], [7, 5]
    x0, x1 = np.linspace(lo[0], hi[0], count[0]), np.linspace(lo[1], hi[1], count[1])
    X0, X1 = np.meshgrid(x0, x1)
    ref = np.column_stack((X0.ravel(), X1.ravel()))
    assert np.array_equal(oracle.grid_points(lo, hi, count), ref)
    assert np.array_equal(oracle.grid_points(lo, hi, count, first=9, n=11), ref[9:20])


def test_mean_gradient_matches_finite_differences():
    X, Y, hyp = _small(n=15, d=2, q=3)
    ds = oracle.make_inference_dataset(X, Y, hyp)
    pts = np.random.default_rng(2).uniform(-1, 1, size=(5, 2))
    g = oracle.mean_grad(pts, ds)
    eps = 1e-6
    for a in range(2):
        dp = np.zeros(2)
        dp[a] = eps
        fd = (oracle.gp_inference(pts + dp, ds)[0] - oracle.gp_inference(pts - dp, ds)[0]) / (2 * eps)
        assert np.allclose(g[:, :, a], fd, rtol=1e-6, atol=1e-7)


def test_sets_against_plain_loops():
    """S / M / U / G definitions re-evaluated with pure-Python loops on a tiny grid."""
    cfg = synthetic.make_config("A", n=10)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    pts = oracle.grid_points(lo, hi, [13, 11])
    for quirk in (True, False):
        r = oracle.safeopt_sweep(pts, cfg["ds"], 1.5, quirk_L_index=quirk)
        lcb, ucb, var = r["lcb"], r["ucb"], r["var"]
        N, q = lcb.shape
        S = [all(lcb[g, i] >= 0 for i in range(1, q)) for g in range(N)]
        U = [max(lcb[g, i] for i in range(1, q)) <= 0 for g in range(N)]
        assert list(r["S"]) == S and list(r["U"]) == U
        u_star = min(ucb[g, 0] for g in range(N) if S[g])
        M = [S[g] and lcb[g, 0] <= u_star for g in range(N)]
        assert list(r["M"]) == M and r["u_star"] == u_star
        best = max((g for g in range(N) if M[g]), key=lambda g: (var[g, 0], -g))
        assert r["minimizer_index"] == best
        L = r["L"][q - 1] if quirk else r["L"][1]
        G = []
        for g in range(N):
            ok = False
            if S[g]:
                for h in range(N):
                    if U[h] and ucb[g, 1] - L * np.linalg.norm(pts[g] - pts[h] + 1e-8) >= 0:   # models/SafeOpt.py:87
                        ok = True
                        break
            G.append(ok)
        assert list(r["G"][0]) == G


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    z = np.load(path)
    ds = oracle.make_inference_dataset(z["X"], z["Y"], z["hypopt"])
    pts = oracle.grid_points(z["bound"][:, 0], z["bound"][:, 1], z["count"])
    r = oracle.safeopt_sweep(pts, ds, float(z["b"]), quirk_L_index=bool(z["quirk"]))
    scale = np.maximum(1.0, ds["Y_std"])
    assert np.max(np.abs(r["mean"] - z["mean"]) / scale) < 1e-11
    assert np.max(np.abs(r["var"] - z["var"]) / scale ** 2) < 1e-11
    for k in ("S", "U", "M"):
        assert np.array_equal(r[k], z[k]), k
    assert np.array_equal(r["G"], z["G"])
    assert r["minimizer_index"] == int(z["minimizer_index"])
    assert np.array_equal(r["expander_index"], z["expander_index"])
    g = oracle.goose_sweep(pts, ds, float(z["b"]), quirk_L_index=bool(z["quirk"]), mean_var=(r["mean"], r["var"]))
    assert np.array_equal(g["O"], z["O"])
    assert g["safe_min_index"] == int(z["safe_min_index"]) and g["target_index"] == int(z["target_index"])
    assert g["explore_index"] == int(z["explore_index"])


def test_fp32_mode_close_to_fp64():
    cfg = synthetic.make_config("A", n=20)
    pts = oracle.grid_points(cfg["bound"][:, 0], cfg["bound"][:, 1], [20, 20])
    m64, v64 = oracle.gp_inference(pts, cfg["ds"])
    m32, v32 = oracle.gp_inference(pts, cfg["ds"], dtype=np.float32)
    assert m32.dtype == np.float32
    assert np.max(np.abs(m32 - m64)) < 5e-4 and np.max(np.abs(v32 - v64)) < 5e-4


def test_wo_plant_restatement_matches_the_reference_table():
    """oracle/plants.py against the reference project's own 100 x 100 table of the William-Otto reactor outputs
    (tests/golden/wo_contour_reference.npz, a verbatim copy of its data/data_contour_WilliamOttoReactor.npz): the table was
    made with SciPy fsolve at its default tolerance (~1.5e-8 relative in the states), hence the bars."""
    from oracle import plants
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "wo_contour_reference.npz"), allow_pickle=False)
    U = np.stack([z["X_0"].ravel(), z["X_1"].ravel()], axis=1)
    assert U.shape == (10000, 2) and U[:, 0].min() == 4.0 and U[:, 1].max() == 100.0
    Y = plants.wo_outputs(U)
    assert np.max(np.abs(Y[:, 0] - z["Y_objective"].ravel())) < 1e-5          # values up to 268
    assert np.max(np.abs(Y[:, 1] - z["Y_constraint1"].ravel())) < 1e-8
    assert np.max(np.abs(Y[:, 2] - z["Y_constraint2"].ravel())) < 1e-8
    # the known optimum of the reference (test/test_WilliamOttoReactor.py:250): objective -76.036 at the constrained optimum
    f, J = plants.wo_residual_jacobian(plants.wo_steady_state(U[:7]), U[:7, 0], U[:7, 1])
    assert np.max(np.abs(f)) < 1e-15


def test_config_C_plant_labels_match_the_reference_table():
    """The host generator of BASELINE.json configs[2]'s observations (safebo_amd.synthetic.williams_otto) against the
    reference's own 100 x 100 table of the William-Otto plant and against the oracle's restatement."""
    from oracle import plants
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "wo_contour_reference.npz"), allow_pickle=False)
    U = np.stack([z["X_0"].ravel(), z["X_1"].ravel()], axis=1)
    Y = synthetic.williams_otto(U)
    assert np.max(np.abs(Y[:, 0] - z["Y_objective"].ravel())) < 1e-5
    assert np.max(np.abs(Y[:, 1] - z["Y_constraint1"].ravel())) < 1e-8
    assert np.max(np.abs(Y[:, 2] - z["Y_constraint2"].ravel())) < 1e-8
    assert np.max(np.abs(Y - plants.wo_outputs(U)) / np.maximum(1.0, np.abs(Y))) < 1e-12
    cfg = synthetic.make_config("C")
    assert cfg["Y"].shape == (256, 3) and np.array_equal(cfg["Y"], synthetic.williams_otto(cfg["X"]))
    assert cfg["b"] == 2.0 and cfg["bound"].tolist() == [[4.0, 7.0], [70.0, 100.0]]        # test/test_GoOSE.py:279-280


def test_extended_precision_evaluation_brackets_the_fp64_oracle():
    """oracle/extended.py: in a well-conditioned model the three evaluations agree to fp64 rounding; at the bottom of the
    reference's noise range the fp64 formula visibly rounds (the quantity the GPU envelope tests are scaled by) while the
    two extended-precision evaluations still agree with each other to the accuracy of the stored inverse."""
    from oracle import extended
    cfg = synthetic.make_config("B", n=60)
    pts = np.random.default_rng(1).uniform(cfg["bound"][:, 0], cfg["bound"][:, 1], size=(64, 2))
    ds = cfg["ds"]
    om, ov = oracle.gp_inference(pts, ds)
    gm, gv = extended.posterior_given_invK(pts, ds)
    tm, tv = extended.posterior_true(pts, ds)
    assert gm.dtype == np.longdouble
    assert np.max(np.abs(om - gm)) < 1e-12 and np.max(np.abs(ov - gv)) < 1e-12
    assert np.max(np.abs(gm - tm)) < 1e-12 and np.max(np.abs(gv - tv)) < 1e-12
    hard = synthetic.make_dataset(cfg["X"], cfg["Y"], synthetic.default_hypopt(2, 2, log_ell=0.5, log_sf=1.5, log_sn=-5.0))
    assert np.linalg.cond(np.linalg.inv(hard["invKopt"][0])) > 1e6
    om, ov = oracle.gp_inference(pts, hard)
    gm, gv = extended.posterior_given_invK(pts, hard)
    e_formula = float(np.max(np.abs(ov - gv)))
    assert 1e-11 < e_formula < 1e-5


@pytest.mark.parametrize("cfg_name,n,count,b", [("B", 40, [61, 47], 3.0), ("C", 64, [40, 33], 2.0), ("D", 48, [7, 6, 5, 4], 0.5)])
def test_openmp_restatement_matches_the_numpy_oracle(cfg_name, n, count, b):
    """oracle/c/sweep_omp.c (the multi-threaded CPU column of bench.py) against oracle.safeopt_sweep: posterior to rounding
    (the sums of k^T invK run in another order), masks / u* / minimiser equal, whole grid and a ragged sub-range."""
    from oracle import omp
    cfg = synthetic.make_config(cfg_name, n=n)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    pts = oracle.grid_points(lo, hi, count)
    ref = oracle.safeopt_sweep(pts, cfg["ds"], b)
    got = omp.safeopt_sweep(lo, hi, count, cfg["ds"], b, want_posterior=True, want_masks=True)
    ys = np.maximum(1.0, cfg["ds"]["Y_std"])
    assert np.max(np.abs(got["mean"] - ref["mean"]) / ys) < 1e-10 and np.max(np.abs(got["var"] - ref["var"]) / ys ** 2) < 1e-10
    for k in ("S", "U", "M"):
        assert np.array_equal(got[k], ref[k]), k
    assert got["count_S"] == int(ref["S"].sum()) and got["count_M"] == int(ref["M"].sum())
    assert got["minimizer_index"] == ref["minimizer_index"] and got["u_star"] == pytest.approx(ref["u_star"], rel=1e-10)
    first, m = 37, 301
    sub = omp.safeopt_sweep(lo, hi, count, cfg["ds"], b, first=first, n=m, want_posterior=True, want_masks=True)
    assert np.array_equal(sub["mean"], got["mean"][first:first + m]) and np.array_equal(sub["S"], got["S"][first:first + m])
    assert omp.threads() >= 1
