"""GPU tests of the column-word set phase (r05, csrc/sets_colpath.inc.hpp): one-constraint SafeOpt sweeps on 2-D grids of whole
64 x 128 posterior tiles take S / U as 64-bit column words from the GEMM posterior's epilogue, u* from the objective's tiles,
and decide M / G_1 and the two arg-max reductions on those words (models/SafeOpt.py:47-66, 85-124).  Every mask, count and index
must equal the oracle's (small grids, brute-force expander) and the byte-mask pipeline's (option col_path = 0) at full size."""
import numpy as np
import pytest

import oracle
from safebo_amd import synthetic

pytestmark = pytest.mark.gpu

KEYS = ("minimizer_index", "expander_index", "expander_best_c", "choose_minimizer", "count_S", "count_U", "count_M", "u_star",
        "minimizer_std", "expander_std")


def _masks(eng):
    return {"S": eng.mask("S"), "U": eng.mask("U"), "M": eng.mask("M"), "G": eng.mask("G", 1)}


def _check_oracle(eng, res, masks, ref, lean=0):
    for k in ("S", "U", "M"):
        assert np.array_equal(masks[k], ref[k]), k
    assert np.array_equal(masks["G"], ref["G"][0])
    assert res["minimizer_index"] == ref["minimizer_index"]
    assert list(res["expander_index_c"]) == list(ref["expander_index"])
    assert res["u_star"] == ref["u_star"] or abs(res["u_star"] - ref["u_star"]) < 1e-10
    assert (res["count_S"], res["count_U"], res["count_M"], res["count_G"][0]) == (ref["S"].sum(), ref["U"].sum(), ref["M"].sum(), ref["G"][0].sum())
    # (a lean sweep reports L[0] = 0: no sweep reads the objective's Lipschitz key, include/safebo.h)
    assert np.allclose(res["L"][1 if lean else 0:], ref["L"][1 if lean else 0:], rtol=1e-9) and (not lean or res["L"][0] == 0.0)
    assert res["choose_minimizer"] == ref["choose_minimizer"]


@pytest.mark.parametrize("cfg_name,n,count,b", [
    ("A", 20, [128, 64], None), ("A", 64, [128, 128], None), ("B", 64, [256, 192], None), ("B", 128, [384, 128], None),
    ("H", 96, [128, 320], None), ("B", 40, [256, 64], 1.0), ("A", 30, [512, 64], 2.0),
])
def test_column_path_matches_oracle(engine, cfg_name, n, count, b):
    """First sweep (K1i) and second sweep (K1b) of a model on small grids, fused classification forced: oracle parity of every
    mask / index, and the posterior the sweeps leave resident."""
    cfg = synthetic.make_config(cfg_name, n=n)
    b = cfg["b"] if b is None else b
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    pts = oracle.grid_points(lo, hi, count)
    ref = oracle.safeopt_sweep(pts, cfg["ds"], b)
    engine.set_option("fuse_classify", 1)
    engine.set_option("col_path", 2)
    try:
        engine.set_grid(lo, hi, count)
        engine.set_model(cfg["ds"], dtype="f64")
        kernels = []
        for sweep in range(2):
            res = engine.sweep_safeopt(b, want_masks=True)
            prof = engine.profile()
            kernels.append(prof["posterior_kernel"])
            assert prof["set_path"] == 1, prof
            if ref["empty_safe_set"]:
                continue
            _check_oracle(engine, res, _masks(engine), ref)
            assert res["guard_band"] == 0
            mean, var = engine.posterior()
            ys = np.maximum(1.0, cfg["ds"]["Y_std"])
            assert np.max(np.abs(mean - ref["mean"]) / ys) < 1e-10 and np.max(np.abs(var - ref["var"]) / ys ** 2) < 1e-10
        assert set(kernels) <= {4, 6}, kernels
    finally:
        engine.set_option("fuse_classify", -1)
        engine.set_option("col_path", 1)


@pytest.mark.parametrize("overlap", [0, 1])
def test_column_path_one_stream_and_two(engine, overlap):
    """Option col_overlap: the expander chain beside the objective's posterior launch (1) or everything on one stream (0)."""
    cfg = synthetic.make_config("B", n=96)
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], [384, 192]
    ref = oracle.safeopt_sweep(oracle.grid_points(lo, hi, count), cfg["ds"], cfg["b"])
    engine.set_option("fuse_classify", 1)
    engine.set_option("col_path", 2)
    engine.set_option("col_overlap", overlap)
    try:
        engine.set_grid(lo, hi, count)
        engine.set_model(cfg["ds"], dtype="f64")
        for sweep in range(4):
            res = engine.sweep_safeopt(cfg["b"], want_masks=True, lean=max(0, sweep - 1))
            assert engine.profile()["set_path"] == 1
            _check_oracle(engine, res, _masks(engine), ref, lean=max(0, sweep - 1))
        # explore_safeset with a caller's target (models/GoOSE.py:116-119) reads the safe set out of the column words
        pts = oracle.grid_points(lo, hi, count)
        target = np.array([0.9 * hi[0], 0.8 * lo[1]])
        dist = np.sqrt(((pts - target) ** 2).sum(axis=1))
        idx, x = engine.explore_safeset(target)
        assert idx == int(np.argmin(np.where(ref["S"], dist, np.inf))) and np.array_equal(x, pts[idx])
    finally:
        engine.set_option("col_overlap", 1)
        engine.set_option("fuse_classify", -1)
        engine.set_option("col_path", 1)


@pytest.mark.parametrize("lean", [0, 1, 2])
def test_column_path_lean_and_late_recheck(engine, lean):
    """`lean` sweeps (objective tiles without a safe candidate store nothing) return the same result, and the posterior asked for
    afterwards is complete; exact_lazy = 2 forces the late exhaustive recheck of the byte-mask path behind the column kernels."""
    cfg = synthetic.make_config("B", n=64)
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], [256, 128]
    pts = oracle.grid_points(lo, hi, count)
    ref = oracle.safeopt_sweep(pts, cfg["ds"], cfg["b"])
    engine.set_option("fuse_classify", 1)
    engine.set_option("col_path", 2)
    try:
        engine.set_grid(lo, hi, count)
        engine.set_model(cfg["ds"], dtype="f64")
        for lazy in (1, 2, 1):
            engine.set_option("exact_lazy", lazy)
            res = engine.sweep_safeopt(cfg["b"], want_masks=True, lean=lean)
            assert engine.profile()["set_path"] == 1
            _check_oracle(engine, res, _masks(engine), ref, lean=lean)
        mean, var = engine.posterior()
        ys = np.maximum(1.0, cfg["ds"]["Y_std"])
        assert np.max(np.abs(mean - ref["mean"]) / ys) < 1e-10 and np.max(np.abs(var - ref["var"]) / ys ** 2) < 1e-10
        # a sweep on the resident posterior (byte-mask path) agrees
        res2 = engine.sweep_safeopt(cfg["b"], want_masks=True, posterior_ready=True)
        assert engine.profile()["set_path"] == 0
        _check_oracle(engine, res2, _masks(engine), ref)
    finally:
        engine.set_option("exact_lazy", 1)
        engine.set_option("fuse_classify", -1)
        engine.set_option("col_path", 1)


def test_column_path_guard_reevaluation(engine):
    """guard_band = 2: the sweep re-evaluates exactly behind a column-path first pass (lean or not) and returns the same sets."""
    cfg = synthetic.make_config("B", n=64)
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], [256, 128]
    pts = oracle.grid_points(lo, hi, count)
    ref = oracle.safeopt_sweep(pts, cfg["ds"], cfg["b"])
    engine.set_option("fuse_classify", 1)
    engine.set_option("col_path", 2)
    engine.set_option("guard_band", 2)
    try:
        engine.set_grid(lo, hi, count)
        engine.set_model(cfg["ds"], dtype="f64")
        for lean in (0, 1, 2, 2):
            res = engine.sweep_safeopt(cfg["b"], want_masks=True, lean=lean)
            assert res["guard_passes"] >= 1
            _check_oracle(engine, res, _masks(engine), ref, lean=lean)
    finally:
        engine.set_option("guard_band", 1)
        engine.set_option("fuse_classify", -1)
        engine.set_option("col_path", 1)


@pytest.mark.parametrize("cfg_name", ["B", "H"])
def test_column_path_equals_byte_mask_path_full_size(engine, cfg_name):
    """BASELINE configs B (2048^2, n = 128) and H (4096^2, n = 512) at full size: the column-word set phase and the byte-mask
    pipeline return identical masks (all 4.2 M / 16.8 M candidates), counts, indices and keys -- on the model's first sweep
    (K1i) and on its second (K1b), plain and lean."""
    cfg = synthetic.make_config(cfg_name)
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], cfg["count"]
    engine.set_grid(lo, hi, count)
    out = {}
    try:
        for path in (0, 1):
            engine.set_option("col_path", 2 * path)
            engine.set_model(cfg["ds"], dtype="f64")                 # (a new model: the first sweep runs K1i again)
            for sweep in range(4):
                res = engine.sweep_safeopt(cfg["b"], want_masks=True, lean=max(0, sweep - 1))
                prof = engine.profile()
                assert prof["set_path"] == path
                out[(path, sweep)] = (res, _masks(engine), prof["posterior_kernel"])
    finally:
        engine.set_option("col_path", 1)
    for sweep in range(4):
        r0, m0, k0 = out[(0, sweep)]
        r1, m1, k1 = out[(1, sweep)]
        assert k0 == k1
        for k in KEYS:
            assert r0[k] == r1[k], (sweep, k, r0[k], r1[k])
        assert list(r0["count_G"]) == list(r1["count_G"]) and list(r0["expander_index_c"]) == list(r1["expander_index_c"])
        assert np.array_equal(r0["L"], r1["L"])
        assert r0["guard_band"] == r1["guard_band"] == 0
        for k in ("S", "U", "M", "G"):
            assert np.array_equal(m0[k], m1[k]), (sweep, k)
    # the posterior behind a lean sweep is complete when asked for
    mean, var = engine.posterior()
    rng = np.random.default_rng(11)
    sub = np.sort(rng.choice(count[0] * count[1], size=2048, replace=False))
    om, ov = oracle.gp_inference(oracle.grid_points(lo, hi, count)[sub], cfg["ds"])
    ys = np.maximum(1.0, cfg["ds"]["Y_std"])
    assert np.max(np.abs(mean[sub] - om) / ys) < 1e-10 and np.max(np.abs(var[sub] - ov) / ys ** 2) < 1e-10


@pytest.mark.parametrize("col", [0, 2])
def test_deferred_gradient_launch_equals_the_gate_in_front(engine, col):
    """K1i's first sweep of a model (option grad_defer, default 1): the gradient gate's kernels and a launch of the gradient phases alone
    run beside the posterior launches on stream3 instead of in front of them -- the Lipschitz keys (models/SafeOpt.py:68-83), every
    count and index, and both outputs' posterior are those of the r04 order, on the byte-mask path and on the column path; GoOSE and
    trust-region sweeps on such a first posterior read the same keys."""
    cfg = synthetic.make_config("B", n=96)
    lo, hi, count = cfg["bound"][:, 0], cfg["bound"][:, 1], [512, 256]
    out = {}
    engine.set_option("col_path", col)
    engine.set_grid(lo, hi, count)
    try:
        for defer in (0, 3, 2):                                  # (3: behind the gate; 2: no gate at all, both gradient phases on every tile)
            engine.set_option("grad_defer", defer)
            rows = []
            for kind in ("safeopt", "goose", "tr"):
                engine.set_model(cfg["ds"], dtype="f64")         # (a new model: K1i again)
                if kind == "safeopt":
                    res = engine.sweep_safeopt(cfg["b"], want_masks=True)
                elif kind == "goose":
                    res = engine.sweep_goose(cfg["b"])
                else:
                    res = engine.sweep_tr(cfg["b"], 0.5 * (lo + hi), 0.3 * float(np.min(hi - lo)))
                assert engine.profile()["posterior_kernel"] == 6
                rows.append((res, engine.posterior()))
            out[defer] = rows
    finally:
        engine.set_option("grad_defer", 1)
        engine.set_option("col_path", 1)
    for (r0, (m0, v0)), (r1, (m1, v1)) in list(zip(out[0], out[3])) + list(zip(out[0], out[2])):
        assert np.array_equal(m0, m1) and np.array_equal(v0, v1)
        for k in r0:
            a, b = r0[k], r1[k]
            if k.endswith("_ms"):
                continue
            if isinstance(a, np.ndarray) or isinstance(a, (list, tuple)):
                assert np.array_equal(np.asarray(a), np.asarray(b)), k
            else:
                assert a == b or (isinstance(a, float) and np.isnan(a) and np.isnan(b)), (k, a, b)
    assert np.all(np.asarray(out[3][0][0]["L"]) > 0)
