"""Worker for tests/test_distributed_cpu.py (world_size-2 rehearsal of the shard / exchange protocol over gloo, or -- argv[5] ==
"tcp" -- over the package's own stdlib rendezvous).

Each rank evaluates its own shard with the NumPy oracle standing in for the HIP kernels (there is no GPU
in the CPU test environment), then runs the *same* exchange steps libsafebo.so performs -- C1 max of order
keys, C2 all-gather of the padded U mask, C3 sum of per-rank rows -- through the HostRelay callbacks the
1-GPU rehearsal transport uses, and merges with the host rule.  Rank 0 writes the merged result.
"""
import ctypes as C
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def ord_key(v):
    b = struct.unpack("<Q", struct.pack("<d", float(v)))[0]
    return (~b) & 0xFFFFFFFFFFFFFFFF if b >> 63 else b | (1 << 63)


def ord_val(k):
    b = k & 0x7FFFFFFFFFFFFFFF if k >> 63 else (~k) & 0xFFFFFFFFFFFFFFFF
    return struct.unpack("<d", struct.pack("<Q", b))[0]


def main():
    rank, world, port, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import oracle
    from safebo_amd import synthetic
    from safebo_amd.distributed import HostRelay, merge_slots, shard_planes, init_from_env

    if len(sys.argv) > 5 and sys.argv[5] == "tcp":
        dist = init_from_env()
    else:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from _gloo_group import GlooGroup
        dist = GlooGroup()
    relay = HostRelay(dist)
    cfg = synthetic.make_config("A", n=20)
    count = [30, 23]                                   # 23 planes over 2 ranks: uneven shards
    lo, hi = cfg["bound"][:, 0], cfg["bound"][:, 1]
    first_of = shard_planes(count[1], count[0], world)
    first, n_local = first_of[rank], first_of[rank + 1] - first_of[rank]
    pts = oracle.grid_points(lo, hi, count, first=first, n=n_local)
    mean, var = oracle.gp_inference(pts, cfg["ds"])
    lcb, ucb = oracle.bounds(mean, var, cfg["b"])
    S = np.all(lcb[:, 1:] >= 0, axis=1)
    U = np.all(lcb[:, 1:] <= 0, axis=1)
    gn = oracle.mean_grad_infnorm(pts, cfg["ds"])
    q = mean.shape[1]

    # C1: [~u*_key, L keys] with max
    keys = (C.c_uint64 * (1 + q))()
    ukey = min([ord_key(u) for u in ucb[S, 0]], default=0xFFFFFFFFFFFFFFFF)
    keys[0] = (~ukey) & 0xFFFFFFFFFFFFFFFF
    for i in range(q):
        keys[1 + i] = struct.unpack("<Q", struct.pack("<d", float(gn[:, i].max())))[0]
    assert relay._allreduce(None, C.addressof(keys), 1 + q, 0, 1) == 0
    u_star = ord_val((~keys[0]) & 0xFFFFFFFFFFFFFFFF)
    L = np.array([struct.unpack("<d", struct.pack("<Q", keys[1 + i]))[0] for i in range(q)])

    # C2: padded all-gather of the U mask, then compaction to the whole grid
    maxlocal = max(first_of[r + 1] - first_of[r] for r in range(world))
    send = (C.c_uint8 * maxlocal)(*([int(x) for x in U] + [0] * (maxlocal - n_local)))
    recv = (C.c_uint8 * (maxlocal * world))()
    assert relay._allgather(None, C.addressof(send), C.addressof(recv), maxlocal) == 0
    total = first_of[-1]
    Ufull = np.zeros(total, dtype=bool)
    for r in range(world):
        nl = first_of[r + 1] - first_of[r]
        Ufull[first_of[r]:first_of[r + 1]] = np.frombuffer(recv, dtype=np.uint8)[r * maxlocal:r * maxlocal + nl] != 0

    # local M / G and their arg-max slots
    M = S & (lcb[:, 0] <= u_star)
    all_pts = oracle.grid_points(lo, hi, count)
    G = np.zeros(n_local, dtype=bool)
    xh = all_pts[Ufull]
    for g in np.nonzero(S)[0]:
        G[g] = bool(np.any(ucb[g, 1] - L[q - 1] * oracle.shifted_norm(pts[g][None, :], xh) >= 0))

    def slot(mask):
        if not mask.any():
            return (0.0, -1)
        i = int(np.argmax(np.where(mask, var[:, 0], -np.inf)))
        return (float(var[i, 0]), first + i)
    mine = [slot(M), slot(G)]

    # C3: one sum all-reduce of [world][row]; each rank fills only its own row
    row = 2 * len(mine) + 3
    buf = (C.c_double * (world * row))()
    for t, (v, i) in enumerate(mine):
        buf[rank * row + t] = v
        buf[rank * row + len(mine) + t] = float(i)
    buf[rank * row + 2 * len(mine) + 0] = float(S.sum())
    buf[rank * row + 2 * len(mine) + 1] = float(M.sum())
    buf[rank * row + 2 * len(mine) + 2] = float(G.sum())
    assert relay._allreduce(None, C.addressof(buf), world * row, 1, 0) == 0
    rows = [[(buf[r * row + t], int(buf[r * row + len(mine) + t])) for t in range(len(mine))] for r in range(world)]
    merged = merge_slots(rows, [True, True])
    counts = [sum(buf[r * row + 2 * len(mine) + k] for r in range(world)) for k in range(3)]
    if rank == 0:
        json.dump({"u_star": u_star, "L": L.tolist(), "minimizer": merged[0][1], "expander": merged[1][1],
                   "count_S": counts[0], "count_M": counts[1], "count_G": counts[2], "first_of": first_of}, open(out_path, "w"))
    dist.barrier()
    dist.destroy()


if __name__ == "__main__":
    main()
