"""world_size-2 gloo test on CPU: shard layout, exchange protocol (C1/C2/C3) and host merge rule."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

import oracle
from safebo_amd import synthetic
from safebo_amd.distributed import merge_slots, shard_planes

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def test_shard_planes_cover_the_grid():
    for planes, stride, world in [(2048, 2048, 8), (23, 30, 2), (128, 128 ** 3, 8), (5, 7, 8), (1, 10, 4)]:
        f = shard_planes(planes, stride, world)
        assert f[0] == 0 and f[-1] == planes * stride and len(f) == world + 1
        assert all(b >= a and (b - a) % stride == 0 for a, b in zip(f, f[1:]))
        sizes = [(b - a) // stride for a, b in zip(f, f[1:])]
        assert max(sizes) - min(sizes) <= 1


def test_merge_slots_tie_breaks_to_lowest_index():
    rows = [[(1.0, 40), (2.0, 7)], [(1.0, 12), (0.0, -1)], [(0.5, 3), (3.0, 90)]]
    assert merge_slots(rows, [True, True]) == [(1.0, 12), (3.0, 90)]
    assert merge_slots(rows, [False, False]) == [(0.5, 3), (2.0, 7)]
    assert merge_slots([[(0.0, -1)], [(0.0, -1)]], [True]) == [(0.0, -1)]


def test_two_rank_gloo_sweep_matches_single_rank_oracle(tmp_path):
    port, out = _free_port(), str(tmp_path / "merged.json")
    env = dict(os.environ, OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(r), "2", port, out], env=env)
             for r in range(2)]
    try:
        codes = [p.wait(timeout=300) for p in procs]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
                p.wait()
    assert codes == [0, 0]
    got = json.load(open(out))
    cfg = synthetic.make_config("A", n=20)
    pts = oracle.grid_points(cfg["bound"][:, 0], cfg["bound"][:, 1], [30, 23])
    ref = oracle.safeopt_sweep(pts, cfg["ds"], cfg["b"])
    assert got["first_of"] == [0, 11 * 30, 23 * 30]
    assert got["u_star"] == ref["u_star"]
    assert np.allclose(got["L"], ref["L"], rtol=1e-13)
    assert got["minimizer"] == ref["minimizer_index"]
    assert got["expander"] == int(ref["expander_index"][0])
    assert (got["count_S"], got["count_M"], got["count_G"]) == (ref["S"].sum(), ref["M"].sum(), ref["G"][0].sum())
