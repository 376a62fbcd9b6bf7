"""Host-side mirrors of the reference classes (GP_Safe.GP, SafeOpt.BO, GoOSE.BO): CPU logic here, device-backed
behaviour under the gpu marker."""
import os

import numpy as np
import pytest

import oracle
from safebo_amd import GoOSE, GP_TR, SafeOpt, synthetic
from safebo_amd.GP_Safe import GP


def benoit_f(u, noise=0):
    return u[0] ** 2 + u[1] ** 2 + u[0] * u[1]


def benoit_g(u, noise=0):
    return -(1. - u[0] + u[1] ** 2 + 2. * u[1])


BOUND = np.array([[-.6, 1.5], [-1., 1.]])


def _init(cls, n=12, grid=(50, 50), b=3.0, fixed=True, **kw):
    if "TR_parameters" in kw:
        m = cls([benoit_f, benoit_g], BOUND, b, kw.pop("TR_parameters"), grid=grid, **kw)
    else:
        m = cls([benoit_f, benoit_g], BOUND, b, grid=grid, **kw)
    X, Y = m.Data_sampling(n, np.array([1.4, -.8]), 0.3)
    if fixed:
        m.fixed_hyper = synthetic.default_hypopt(2, 2)
    m.GP_initialization(X, Y, "RBF", multi_hyper=5, var_out=True)
    return m


def test_gp_state_matches_oracle_restatement():
    m = _init(SafeOpt.BO)
    ds = oracle.make_inference_dataset(m.X, m.Y, m.hypopt)
    for k in ("X_mean", "X_std", "Y_mean", "Y_std", "X_norm", "Y_norm", "hypopt"):
        assert np.array_equal(np.asarray(m.inference_datasets[k]), ds[k]), k
    for a, b in zip(m.inference_datasets["invKopt"], ds["invKopt"]):
        assert np.array_equal(a, b)
    assert m.n_point == 12 and m.nx_dim == 2 and m.ny_dim == 2 and m.n_fun == 2
    # every sample lies in the ball of radius 0.3 around x_0 (models/GP_Safe.py:30-50)
    assert np.all(np.linalg.norm(m.X - np.array([1.4, -.8]), axis=1) <= 0.3 + 1e-12)


def test_oracle_nll_matches_its_fixture_and_the_host_objective():
    """oracle.negative_loglikelihood (the checker of the device objective) against tests/golden/nll_population.npz, and the
    product class's own NumPy objective against the oracle -- on the CPU, no device involved."""
    import oracle
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nll_population.npz"))
    for n, d in ((4, 2), (20, 2), (45, 2)):
        Xn, y, H, golden = fx[f"X_{n}_{d}"], fx[f"y_{n}_{d}"], fx[f"H_{n}_{d}"], fx[f"nll_{n}_{d}"]
        want = np.array([oracle.negative_loglikelihood(h, Xn, y) for h in H])
        ok = np.isfinite(golden)
        assert np.array_equal(np.isfinite(want), ok) and np.allclose(want[ok], golden[ok], rtol=1e-11, atol=1e-11)
        m = GP([lambda u, noise=0: 0.0])
        m.kernel, m.nx_dim, m.n_point = "RBF", d, n
        host = np.array([m.negative_loglikelihood(h, Xn, y[:, None]) for h in H])
        assert np.allclose(host[ok], want[ok], rtol=1e-11, atol=1e-11)


def test_nll_matches_direct_formula():
    m = _init(SafeOpt.BO)
    hyper = np.array([-0.3, 0.2, 0.1, -3.0])
    y = m.Y_norm[:, 1:2]
    K = m.Cov_mat("RBF", m.X_norm, m.X_norm, np.exp(2 * hyper[:2]), np.exp(2 * hyper[2])) + (np.exp(2 * hyper[3]) + 1e-8) * np.eye(12)
    want = float((y.T @ np.linalg.solve(K, y))[0, 0] + np.linalg.slogdet(K)[1])      # models/GP_Safe.py:186-190
    assert m.negative_loglikelihood(hyper, m.X_norm, y) == pytest.approx(want, rel=1e-10)


def test_reference_error_messages():
    m = _init(SafeOpt.BO)
    with pytest.raises(ValueError, match="W and X_norm dimension"):
        m.Cov_mat("RBF", m.X_norm, m.X_norm, np.ones(3), 1.0)
    with pytest.raises(ValueError, match="no kernel with name"):
        m.Cov_mat("Matern", m.X_norm, m.X_norm, np.ones(2), 1.0)
    with pytest.raises(ValueError):
        m.calc_Cov_mat("RBF", m.X_norm, np.zeros(2), np.ones(3), 1.0)


def test_hyperparameter_fit_stays_in_reference_bounds():
    m = _init(SafeOpt.BO, n=6, fixed=False)
    m.de_options = {"seed": 1, "maxiter": 15, "tol": 1e-3}
    m.add_sample(np.array([1.3, -0.7]), m.calculate_plant_outputs(np.array([1.3, -0.7])))
    assert m.n_point == 7 and m.hypopt.shape == (4, 2)
    assert np.all(m.hypopt[:3] >= -1.5) and np.all(m.hypopt[:3] <= 1.5)          # models/GP_Safe.py:205-206
    assert np.all(m.hypopt[3] >= -5.0) and np.all(m.hypopt[3] <= -2.0)
    for i in range(2):
        K = m.Cov_mat("RBF", m.X_norm, m.X_norm, np.exp(2 * m.hypopt[:2, i]), np.exp(2 * m.hypopt[2, i])) \
            + (np.exp(2 * m.hypopt[3, i]) + np.finfo(np.float32).eps) * np.eye(7)
        assert np.allclose(m.invKopt[i] @ K, np.eye(7), atol=1e-6)


def test_incremental_host_state_is_the_bordered_inverse():
    """CPU part of add_sample(incremental=True): invKopt after the append equals inv(K_{n+1}) under the frozen hyper-parameters
    (the device append itself is checked under the gpu marker)."""
    m = _init(SafeOpt.BO, n=10)

    class _NoDevice:                      # the append's device call is not under test here
        def append_sample(self, xn, yn):
            self.got = (xn.copy(), yn.copy())
    m._engine = _NoDevice()
    m._uploaded_version = m._model_version
    v0 = m._model_version
    x_new = np.array([1.35, -0.75])
    m.add_sample(x_new, m.calculate_plant_outputs(x_new), incremental=True)
    assert m._model_version == v0 + 1 and m._uploaded_version == m._model_version
    d = 2
    for i in range(2):
        K = m.Cov_mat("RBF", m.X_norm, m.X_norm, np.exp(2 * m.hypopt[:d, i]), np.exp(2 * m.hypopt[d, i])) \
            + (np.exp(2 * m.hypopt[d + 1, i]) + np.finfo(np.float32).eps) * np.eye(11)
        assert np.allclose(m.inference_datasets["invKopt"][i] @ K, np.eye(11), atol=1e-9)
    assert np.allclose(m._engine.got[0], (x_new - m.X_mean) / m.X_std)


def test_infnorm_mean_grad_matches_oracle():
    m = _init(GoOSE.BO)
    pts = np.array([[1.2, -0.5], [0.1, 0.3], [1.45, -0.9]])
    want = oracle.mean_grad_infnorm(pts, m.inference_datasets)
    for p in range(3):
        for i in range(2):
            assert m.infnorm_mean_grad(pts[p], i) == pytest.approx(want[p, i], rel=1e-10)


# ------------------------------------------------------------------------------------------------------------ device
@pytest.mark.gpu
@pytest.mark.parametrize("cls", [SafeOpt.BO, GoOSE.BO])
def test_incremental_add_sample_invalidates_the_cached_sweep(cls):
    """sweep -> add_sample(incremental=True) -> sweep must be the sweep of the (n + 1)-point model: the host classes key their
    caches on the model version, which the incremental path bumps without re-uploading; the host copy of the model
    (bordered inverse) stays a complete state that can be uploaded again and that the oracle evaluates."""
    m = _init(cls, n=14, grid=(60, 50))
    first = m.sweep()
    x_new = np.asarray(first["minimizer_x"])
    y_new = m.calculate_plant_outputs(x_new)
    m.add_sample(x_new, y_new, incremental=True)
    assert m.n_point == 15 and m.inference_datasets["X_norm"].shape == (15, 2) and m.inference_datasets["invKopt"][0].shape == (15, 15)
    second = m.sweep()
    assert second is not first
    pts = oracle.grid_points(BOUND[:, 0], BOUND[:, 1], [60, 50])
    ref = oracle.safeopt_sweep(pts, m.inference_datasets, 3.0)       # frozen normalisation + hyper-parameters, 15 rows
    assert second["minimizer_index"] == ref["minimizer_index"] and second["count_S"] == int(ref["S"].sum())
    assert second["minimizer_std"] == pytest.approx(ref["minimizer_std"], rel=1e-7)
    assert second["minimizer_std"] < first["minimizer_std"]           # the sampled point is no longer the most uncertain
    if cls is GoOSE.BO:
        g = m.goose_sweep()
        gref = oracle.goose_sweep(pts, m.inference_datasets, 3.0)
        assert g["safe_min_index"] == gref["safe_min_index"] and g["target_index"] == gref["target_index"]
    # a posterior query with another dataset forces a re-upload of the own one afterwards: it must still be consistent
    other = _init(cls, n=9)
    m.GP_inference(np.array([1.2, -0.6]), other.inference_datasets)
    m._grid_resident()
    again = m.engine.sweep_safeopt(3.0)
    assert again["minimizer_index"] == ref["minimizer_index"]
    # and a rebuild under the same frozen state agrees with the appended device model
    mean_inc, var_inc = m.GP_inference(pts[::37], None)
    om, ov = oracle.gp_inference(pts[::37], m.inference_datasets)
    assert np.max(np.abs(mean_inc - om)) < 1e-8 and np.max(np.abs(var_inc - ov)) < 1e-8


@pytest.mark.gpu
def test_bo_bounds_single_and_batched():
    m = _init(SafeOpt.BO)
    pts = np.random.default_rng(0).uniform(BOUND[:, 0], BOUND[:, 1], size=(300, 2))
    om, ov = oracle.gp_inference(pts, m.inference_datasets)
    ol, ou = oracle.bounds(om, ov, 3.0)
    assert np.max(np.abs(m.lcb(pts, 1) - ol[:, 1])) < 1e-10            # vmap(GP_m.lcb, (0, None))(points, 1)
    assert np.max(np.abs(m.ucb(pts, 0) - ou[:, 0])) < 1e-10
    assert abs(m.mean(pts[7], 0) - om[7, 0]) < 1e-10 and np.ndim(m.lcb(pts[7], 1)) == 0
    mean, var = m.GP_inference(pts[3], m.inference_datasets)          # reference call shape (models/SafeOpt.py:30)
    assert mean.shape == (2,) and np.max(np.abs(mean - om[3])) < 1e-10 and np.max(np.abs(var - ov[3])) < 1e-10
    assert m.lcb_constraint_min(pts[3]) == pytest.approx(ol[3, 1], abs=1e-10)
    m.var_out = False
    assert m.GP_inference(pts[3], m.inference_datasets) == pytest.approx(om[3, 0], abs=1e-10)


@pytest.mark.gpu
def test_safeopt_methods_follow_oracle():
    m = _init(SafeOpt.BO, n=20, grid=(60, 50))
    pts = oracle.grid_points(BOUND[:, 0], BOUND[:, 1], [60, 50])
    ref = oracle.safeopt_sweep(pts, m.inference_datasets, 3.0)
    x, std = m.Minimizer()
    assert np.array_equal(x, pts[ref["minimizer_index"]]) and std == pytest.approx(ref["minimizer_std"], rel=1e-9)
    x, std = m.Expander()
    assert np.array_equal(x, pts[ref["expander_best_index"]]) and std == pytest.approx(ref["expander_best_std"], rel=1e-9)
    x, val = m.minimize_obj_ucb()
    assert val == pytest.approx(ref["u_star"], rel=1e-10)
    assert m.maximize_infnorm_mean_grad(1) == pytest.approx(ref["L"][1], rel=1e-9)
    masks = m.masks()
    assert np.array_equal(masks["S"].ravel(), ref["S"]) and np.array_equal(masks["G1"].ravel(), ref["G"][0])
    xx = np.concatenate([pts[10], pts[500]])
    want = oracle.bounds(*oracle.gp_inference(pts[10:11], m.inference_datasets), 3.0)[1][0, 1] \
        - 2.5 * np.linalg.norm(pts[10] - pts[500] + 1e-8)
    assert m.Lipschitz_continuity_constraint(xx, 1, 2.5) == pytest.approx(want, abs=1e-9)


@pytest.mark.gpu
def test_goose_methods_follow_oracle():
    m = _init(GoOSE.BO, n=20, grid=(48, 40), b=2.0)
    pts = oracle.grid_points(BOUND[:, 0], BOUND[:, 1], [48, 40])
    ref = oracle.goose_sweep(pts, m.inference_datasets, 2.0)
    x, lcb = m.minimize_obj_lcb()
    assert np.array_equal(x, pts[ref["safe_min_index"]]) and lcb == pytest.approx(ref["safe_min_lcb"], abs=1e-9)
    t, tl = m.Target()
    assert np.array_equal(t, pts[ref["target_index"]]) and tl == pytest.approx(ref["target_lcb"], abs=1e-9)
    assert np.array_equal(m.explore_safeset(t), pts[ref["explore_index"]])
    other = np.array([0.2, 0.1])            # an arbitrary target goes through the host fallback
    d = np.where(ref["S"], np.linalg.norm(pts - other, axis=1), np.inf)
    assert np.array_equal(m.explore_safeset(other), pts[int(np.argmin(d))])


@pytest.mark.gpu
@pytest.mark.parametrize("n,d", [(4, 2), (20, 2), (45, 2), (128, 4), (300, 3)])
def test_device_nll_matches_host_objective(engine, n, d):
    """sbo_nll_batch against the oracle's restatement of the reference objective (oracle.negative_loglikelihood,
    models/GP_Safe.py:169-192) and against the committed fixture tests/golden/nll_population.npz (made by
    tests/golden/make_golden.py from the same oracle): a population inside the reference's search box (:205-206)."""
    import oracle
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nll_population.npz"))
    Xn, y, H, golden = fx[f"X_{n}_{d}"], fx[f"y_{n}_{d}"], fx[f"H_{n}_{d}"], fx[f"nll_{n}_{d}"]
    got = engine.nll_batch(Xn, y, H)
    want = np.array([oracle.negative_loglikelihood(h, Xn, y) for h in H])
    ok = np.isfinite(want)
    assert ok.sum() >= 35
    assert np.allclose(want[ok], golden[ok], rtol=1e-11, atol=1e-11)          # the oracle against its fixture
    assert np.allclose(got[ok], golden[ok], rtol=1e-9, atol=1e-9)             # the device against the fixture
    # and the host class's own objective (what SciPy's DE calls when the fit stays on the host) is the same function
    m = GP([lambda u, noise=0: 0.0])
    m.kernel, m.nx_dim, m.n_point = "RBF", d, n
    host = np.array([m.negative_loglikelihood(h, Xn, y[:, None]) for h in H])
    assert np.allclose(host[ok], want[ok], rtol=1e-11, atol=1e-11)
    with pytest.raises(ValueError):
        engine.nll_batch(Xn, y, H[:, :-1])


@pytest.mark.gpu
def test_fit_on_device_reaches_the_host_optimum():
    m_host = _init(SafeOpt.BO, n=12, fixed=False)
    m_dev = _init(SafeOpt.BO, n=12, fixed=False)
    opts = {"seed": 3, "maxiter": 40, "tol": 1e-6, "popsize": 20}
    m_host.de_options, m_dev.de_options = dict(opts), dict(opts)
    m_dev.fit_on_device = True
    m_host.GP_initialization(m_host.X, m_host.Y, "RBF", multi_hyper=5)
    m_dev.X, m_dev.Y = m_host.X.copy(), m_host.Y.copy()
    m_dev.GP_initialization(m_dev.X, m_dev.Y, "RBF", multi_hyper=5)
    for i in range(2):
        f_host = m_host.negative_loglikelihood(m_host.hypopt[:, i], m_host.X_norm, m_host.Y_norm[:, i:i + 1])
        f_dev = m_host.negative_loglikelihood(m_dev.hypopt[:, i], m_host.X_norm, m_host.Y_norm[:, i:i + 1])
        assert f_dev <= f_host + 1e-3 * max(1.0, abs(f_host))     # same objective, equally good minimum
    assert np.all(m_dev.hypopt[:3] >= -1.5) and np.all(m_dev.hypopt[:3] <= 1.5) and np.all(m_dev.hypopt[3] <= -2.0)


@pytest.mark.gpu
def test_device_differential_evolution_reaches_the_host_optimum():
    """fit_on_device = "de": the whole DE search on the device (sbo_fit_de) + SciPy's polish step must find a minimum of
    the same objective that is as good as SciPy's own DE (the searches are stochastic: the values, not the paths, agree)."""
    import time
    m_host = _init(SafeOpt.BO, n=14, fixed=False)
    m_dev = _init(SafeOpt.BO, n=14, fixed=False)
    m_host.de_options = {"seed": 5, "tol": 1e-4}
    m_dev.de_options = {"seed": 5, "tol": 1e-4}
    m_dev.fit_on_device = "de"
    t0 = time.perf_counter()
    m_host.GP_initialization(m_host.X, m_host.Y, "RBF", multi_hyper=5)
    t_host = time.perf_counter() - t0
    m_dev.X, m_dev.Y = m_host.X.copy(), m_host.Y.copy()
    m_dev.GP_initialization(m_dev.X, m_dev.Y, "RBF", multi_hyper=5)        # (includes first-use allocations)
    t0 = time.perf_counter()
    m_dev.GP_initialization(m_dev.X, m_dev.Y, "RBF", multi_hyper=5)
    t_dev = time.perf_counter() - t0
    for i in range(2):
        f_host = m_host.negative_loglikelihood(m_host.hypopt[:, i], m_host.X_norm, m_host.Y_norm[:, i:i + 1])
        f_dev = m_host.negative_loglikelihood(m_dev.hypopt[:, i], m_host.X_norm, m_host.Y_norm[:, i:i + 1])
        assert f_dev <= f_host + 1e-3 * max(1.0, abs(f_host)), (i, f_dev, f_host)
    assert np.all(m_dev.hypopt[:3] >= -1.5) and np.all(m_dev.hypopt[:3] <= 1.5) and np.all(m_dev.hypopt[3] <= -2.0)
    print(f"DE fit of 2 outputs, n = 14: SciPy {t_host * 1e3:.0f} ms, device {t_dev * 1e3:.0f} ms")


TR_PARAMS = {"radius": 0.5, "radius_max": 1, "radius_red": 0.8, "radius_inc": 1.1, "rho_lb": 0.2, "rho_ub": 0.8}


def test_update_TR_rule_matches_oracle_restatement():
    # models/GP_TR.py:56-91 on synthetic numbers (no device: GP_inference is replaced by fixed values)
    m = GP_TR.BO([benoit_f, benoit_g], BOUND, 3.0, dict(TR_PARAMS), grid=(20, 20))
    m.inference_datasets = None
    vals = {}
    m.GP_inference = lambda x, ds: (np.array([vals[tuple(np.round(x, 6))], 0.0]), np.zeros(2))
    x0, x1 = np.array([1.0, -0.5]), np.array([0.8, -0.4])
    for p_old, p_new, g_old, g_new in [([1.0, 0.1], [0.5, 0.1], 1.0, 0.4), ([1.0, 0.1], [0.9, 0.1], 1.0, 0.2),
                                       ([1.0, 0.1], [0.7, 0.1], 1.0, 0.4), ([1.0, 0.1], [1.2, 0.1], 1.0, 0.4),
                                       ([1.0, 0.1], [0.5, -0.1], 1.0, 0.4)]:
        vals[tuple(np.round(x0, 6))], vals[tuple(np.round(x1, 6))] = g_old, g_new
        got = m.update_TR(x0, x1, 0.5, p_old, p_new)
        want = oracle.update_TR(TR_PARAMS, x0, x1, 0.5, p_old, p_new, g_old, g_new)
        assert np.array_equal(got[0], want[0]) and got[1] == pytest.approx(want[1])


@pytest.mark.gpu
def test_trust_region_acquisition_follows_oracle():
    m = _init(GP_TR.BO, n=20, grid=(60, 50), TR_parameters=dict(TR_PARAMS))
    pts = oracle.grid_points(BOUND[:, 0], BOUND[:, 1], [60, 50])
    for x_0, r in [(np.array([1.4, -0.8]), 0.3), (np.array([1.2, -0.6]), 0.1), (np.array([-0.5, 0.9]), 0.05)]:
        ref = oracle.tr_sweep(pts, m.inference_datasets, 3.0, x_0, r)
        x, val = m.minimize_obj_lcb(r, x_0)
        if ref["empty"]:
            assert np.array_equal(x, x_0) and val == np.inf
        else:
            assert np.array_equal(x, pts[ref["index"]]) and val == pytest.approx(ref["lcb_min"], abs=1e-9)
            assert np.array_equal(m.engine.mask("M"), ref["T"])
    x_new, _ = m.minimize_obj_lcb(0.3, np.array([1.4, -0.8]))
    c, r = m.update_TR(np.array([1.4, -0.8]), x_new, 0.3, m.calculate_plant_outputs(np.array([1.4, -0.8])),
                       m.calculate_plant_outputs(x_new))
    assert r in (0.3 * 0.8, 0.3, min(0.3 * 1.1, 1)) and (np.array_equal(c, x_new) or np.array_equal(c, [1.4, -0.8]))


@pytest.mark.gpu
def test_safeopt_campaign_plumbing_config_A():
    """BASELINE.json configs[0]: Benoit 2-D, 50x50 grid, n <= 20 -- the loop of test/test_SafeOpt.py:135-186 with
    its decision rule (:153-158) and stopping test (:178), three iterations, hyper-parameters fitted by DE."""
    m = _init(SafeOpt.BO, n=4, grid=(50, 50), fixed=False)
    m.de_options = {"seed": 0, "maxiter": 10, "tol": 1e-2}
    for it in range(3):
        minimizer, std_min = m.Minimizer()
        expander, std_exp = m.Expander()
        x_new = minimizer if std_min > std_exp else expander
        assert np.all(x_new >= BOUND[:, 0]) and np.all(x_new <= BOUND[:, 1])
        assert m.lcb(x_new, 1) >= 0.0                         # the chosen point is in the safe set
        y = m.calculate_plant_outputs(x_new)
        assert y[1] >= -0.05                                  # and (nearly) safe on the true plant
        m.add_sample(x_new, y)
        if std_exp < 0.01 and std_min < 0.01:
            break
    assert m.n_point == 4 + it + 1


@pytest.mark.gpu
def test_goose_campaign_reaches_the_benoit_optimum_safely():
    """The loop of test/test_GoOSE.py:142-190 end to end -- pessimistic safe minimum vs optimistic target, explore_safeset,
    add_sample with a refit (DE evaluated on the device) -- on a 400 x 400 candidate grid, with the reference's own stopping
    rule |f(x_new) - 0.145249| <= 0.005 (test/test_GoOSE.py:182).  No evaluated point may violate the plant constraint."""
    m = GoOSE.BO([benoit_f, benoit_g], BOUND, 2.0, grid=(400, 400), seed=1)
    m.fit_on_device = True
    m.de_options = {"seed": 0, "maxiter": 60, "tol": 1e-4}
    X, Y = m.Data_sampling(4, np.array([1.4, -.8]), 0.3)
    m.GP_initialization(X, Y, "RBF", multi_hyper=5)
    reached, worst = False, np.inf
    for it in range(25):
        x_safe, lcb_safe = m.minimize_obj_lcb()
        x_t, lcb_t = m.Target()
        x_new = x_safe if lcb_safe <= lcb_t else m.explore_safeset(x_t)          # test/test_GoOSE.py:158-162
        y = m.calculate_plant_outputs(x_new)
        worst = min(worst, y[1])
        m.add_sample(x_new, y)
        if abs(y[0] - 0.145249) <= 0.005:
            reached = True
            break
    assert reached, f"optimum not reached in 25 iterations (last f = {y[0]})"
    assert worst >= -1e-3, f"unsafe evaluation: constraint value {worst}"
