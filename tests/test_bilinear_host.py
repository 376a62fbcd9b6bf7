"""CPU check of the host numerics behind the bilinear (reduced-basis) posterior path: bases reproduce the axis factors of
the RBF kernel to rounding, and the two bilinear forms evaluated in NumPy reproduce the oracle's posterior on a grid."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle
from safebo_amd import synthetic

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "safe-bayesian-optimization_amd", "csrc")


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("bl") / "libblt.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", os.path.join(CSRC, "bilinear_host_test.cpp"), "-o", so])
    return C.CDLL(so)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _axis(shim, As_col, vinv, xn):
    n, count = As_col.shape[0], xn.shape[0]
    r, rc = C.c_int(), C.c_int()
    U, S = np.zeros(n * 64), np.zeros(64 * count)
    rcode = shim.blt_axis(n, _p(As_col), C.c_double(vinv), _p(xn), count, C.byref(r), C.byref(rc), _p(U), _p(S))
    if rcode != 0:
        return None
    return U[:n * r.value].reshape(r.value, n).T.copy(), S[:r.value * count].reshape(r.value, count).copy(), rc.value


@pytest.mark.parametrize("name,n,count,shift", [("B", 128, [96, 80], 0.0), ("H", 300, [64, 72], 0.0), ("C", 64, [80, 64], 0.0),
                                                ("B", 20, [64, 64], 0.3), ("B", 128, [96, 80], -0.5)])
def test_bilinear_forms_reproduce_the_oracle_posterior(shim, name, n, count, shift):
    cfg = synthetic.make_config(name, n=n)
    hyp = synthetic.default_hypopt(2, cfg["Y"].shape[1])
    hyp[:2] += shift
    ds = synthetic.make_dataset(cfg["X"], cfg["Y"], hyp)
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    pts = oracle.grid_points(lo, hi, count)
    om, ov = oracle.gp_inference(pts, ds)
    axes = oracle.grid_axes(lo, hi, count)
    Xn = ds["X_norm"]
    for i in range(cfg["Y"].shape[1]):
        ell = np.exp(2 * hyp[:2, i]); sf2 = float(np.exp(2 * hyp[2, i]))
        vinv = ell ** -0.5
        bases = []
        for a in range(2):
            xn = (axes[a] - ds["X_mean"][a]) / ds["X_std"][a]
            As_col = np.ascontiguousarray(Xn[:, a] * vinv[a])
            basis = _axis(shim, As_col, vinv[a], np.ascontiguousarray(xn))
            assert basis is not None
            U, S, rc = basis
            exact = np.exp(-0.5 * (xn[None, :] * vinv[a] - As_col[:, None]) ** 2)
            assert np.max(np.abs(U @ S - exact)) < 5e-14
            assert np.max(np.abs(U.T @ U - np.eye(U.shape[1]))) < 1e-13
            bases.append((U, S))
        (U0, S0), (U1, S1) = bases
        invK = ds["invKopt"][i]
        Lc = np.linalg.cholesky(np.linalg.inv(invK))
        M = np.ascontiguousarray(np.linalg.inv(Lc))
        mp = 0.0 if i == 0 else -2 * ds["Y_mean"][i] / ds["Y_std"][i]
        alpha = invK @ (ds["Y_norm"][:, i] - mp)
        r0, r1 = U0.shape[1], U1.shape[1]
        K0, K1 = r0 * (r0 + 1) // 2, r1 * (r1 + 1) // 2
        T4, Mb = np.zeros(K0 * K1), np.zeros(r0 * r1)
        shim.blt_forms(n, _p(M), r0, _p(np.ascontiguousarray(U0.T)), r1, _p(np.ascontiguousarray(U1.T)), C.c_double(sf2 * sf2), 1,
                       _p(np.ascontiguousarray(alpha)), C.c_double(sf2), _p(T4), _p(Mb))
        P0, P1 = np.zeros(K0 * count[0]), np.zeros(K1 * count[1])
        shim.blt_pairs(r0, count[0], _p(np.ascontiguousarray(S0)), _p(P0))
        shim.blt_pairs(r1, count[1], _p(np.ascontiguousarray(S1)), _p(P1))
        quad = P1.reshape(K1, -1).T @ T4.reshape(K0, K1).T @ P0.reshape(K0, -1)          # [x1, x0]
        var = np.maximum(sf2 - quad, 0) * ds["Y_std"][i] ** 2
        mean = (mp + S1.T @ Mb.reshape(r0, r1).T @ S0) * ds["Y_std"][i] + ds["Y_mean"][i]
        errv = np.max(np.abs(var.ravel() - ov[:, i])) / max(1.0, ds["Y_std"][i]) ** 2
        errm = np.max(np.abs(mean.ravel() - om[:, i])) / max(1.0, ds["Y_std"][i])
        assert errv < 1e-11 and errm < 1e-11, (errm, errv, r0, r1)


def test_short_length_scales_are_declined(shim):
    """Length-scales much shorter than the axis: the factor family needs more than 64 directions, the builder says so and
    the library keeps the separable-table kernel for such models."""
    cfg = synthetic.make_config("B", n=128)
    xn = np.linspace(-1.9, 1.6, 96)
    vinv = float(np.exp(2.2))
    As_col = np.ascontiguousarray(cfg["ds"]["X_norm"][:, 0] * vinv)
    assert _axis(shim, As_col, vinv, xn) is None
