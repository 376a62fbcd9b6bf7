"""Developer driver: device vs host evaluation of the hyper-parameter objective."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import safebo_amd
from safebo_amd.GP_Safe import GP
eng = safebo_amd.SweepEngine(0)
for n, d, P in [(20, 2, 60), (45, 2, 60), (128, 2, 60), (256, 2, 60), (512, 2, 60), (512, 2, 240)]:
    rng = np.random.default_rng(n)
    X = rng.uniform(-1, 1, size=(n, d)); Xn = (X - X.mean(0)) / X.std(0)
    y = np.sin(Xn.sum(1)); y = (y - y.mean()) / y.std()
    H = np.column_stack([rng.uniform(-1.5, 1.5, size=(P, d + 1)), rng.uniform(-5.0, -2.0, size=P)])
    m = GP([lambda u, noise=0: 0.0]); m.kernel, m.nx_dim, m.n_point = "RBF", d, n
    eng.nll_batch(Xn, y, H)
    t = time.perf_counter()
    for _ in range(5): got = eng.nll_batch(Xn, y, H)
    t_dev = (time.perf_counter() - t) / 5
    t = time.perf_counter(); want = np.array([m.negative_loglikelihood(h, Xn, y[:, None]) for h in H]); t_host = time.perf_counter() - t
    ok = np.isfinite(want)
    print(f"n={n} P={P}: device {t_dev*1e3:.2f} ms, host NumPy {t_host*1e3:.2f} ms, max rel diff {np.max(np.abs(got[ok]-want[ok])/np.abs(want[ok])):.2e}")
