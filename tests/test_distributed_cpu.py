"""world_size-2 tests on CPU (gloo, and the package's own stdlib TCP rendezvous): shard layout, exchange protocol (C1/C2/C3),
host merge rule, and the rendezvous' collectives themselves."""
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


import pytest


@pytest.mark.parametrize("transport", ["gloo", "tcp"])
def test_two_rank_gloo_sweep_matches_single_rank_oracle(tmp_path, transport):
    port, out = _free_port(), str(tmp_path / "merged.json")
    env = dict(os.environ, OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(r), "2", port, out, transport], env=env)
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


def _tcp_rank(rank, world, port, q):
    from safebo_amd.distributed import TcpGroup
    g = TcpGroup(rank, world, "127.0.0.1", port, timeout=60.0)
    out = {}
    out["bcast"] = g.broadcast_object({"id": b"x" * 128, "n": 7} if rank == 0 else None, src=0)
    out["sum"] = g.all_reduce(np.array([rank + 1.5, 2.0 * rank]), "sum").tolist()
    out["max_u64"] = g.all_reduce(np.array([(1 << 63) + rank, 5 - rank], dtype=np.uint64), "max").tolist()
    out["min_i64"] = g.all_reduce(np.array([rank - 1], dtype=np.int64), "min").tolist()
    out["gather"] = g.all_gather_bytes(bytes([rank]) * (rank + 1))
    g.barrier()
    g.destroy()
    q.put((rank, out))


def test_tcp_rendezvous_collectives_three_ranks():
    """The stdlib rendezvous of safebo_amd.distributed (what bench.py and the multi-rank GPU tests use instead of
    torch.distributed): broadcast, sum / max / min all-reduce (uint64 keys included), ragged all-gather, barrier."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = int(_free_port())
    q = ctx.Queue()
    procs = [ctx.Process(target=_tcp_rank, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    try:
        got = dict(q.get(timeout=120) for _ in procs)
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for r in range(3):
        o = got[r]
        assert o["bcast"] == {"id": b"x" * 128, "n": 7}
        assert o["sum"] == [1.5 + 2.5 + 3.5, 0.0 + 2.0 + 4.0]
        assert o["max_u64"] == [(1 << 63) + 2, 5] and o["min_i64"] == [-1]
        assert o["gather"] == [b"\x00", b"\x01\x01", b"\x02\x02\x02"]
