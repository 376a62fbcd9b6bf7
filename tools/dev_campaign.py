"""Developer driver: full GoOSE / SafeOpt campaigns on the Benoit problem (the loops of test/test_GoOSE.py:142-190 and
test/test_SafeOpt.py:135-186) with the device sweep and the device-evaluated hyper-parameter fit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from safebo_amd import GoOSE, SafeOpt

def f(u, noise=0): return u[0] ** 2 + u[1] ** 2 + u[0] * u[1]
def g(u, noise=0): return -(1. - u[0] + u[1] ** 2 + 2. * u[1])
bound = np.array([[-.6, 1.5], [-1., 1.]])

fit = sys.argv[1] if len(sys.argv) > 1 else "batch"        # "batch": SciPy DE with the device objective; "de": device DE
for algo in ("goose", "safeopt"):
    cls = GoOSE.BO if algo == "goose" else SafeOpt.BO
    m = cls([f, g], bound, 2.0 if algo == "goose" else 3.0, grid=(400, 400), seed=1)
    m.fit_on_device = "de" if fit == "de" else True
    m.de_options = {"seed": 0, "maxiter": 60, "tol": 1e-4}
    X, Y = m.Data_sampling(4, np.array([1.4, -.8]), 0.3)
    t0 = time.perf_counter()
    m.GP_initialization(X, Y, "RBF", multi_hyper=5)
    worst = 1e9
    for it in range(30):
        if algo == "goose":
            x_safe, lcb_safe = m.minimize_obj_lcb()
            x_t, lcb_t = m.Target()
            x_new = x_safe if lcb_safe <= lcb_t else m.explore_safeset(x_t)
        else:
            xm, sm = m.Minimizer(); xe, se = m.Expander()
            x_new = xm if sm > se else xe
        y = m.calculate_plant_outputs(x_new)
        worst = min(worst, y[1])
        m.add_sample(x_new, y)
        if algo == "goose" and abs(y[0] - 0.145249) <= 0.005: break
        if algo == "safeopt" and se < 0.01 and sm < 0.01: break
    print(f"{algo} [{fit}]: {it + 1} iterations, {time.perf_counter() - t0:.1f} s, final f={y[0]:.5f}, worst constraint value seen {worst:+.4f}, n={m.n_point}")
