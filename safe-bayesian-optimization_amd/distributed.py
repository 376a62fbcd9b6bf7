"""Rank plumbing for multi-GPU sweeps: one process per GPU.

``torch.distributed`` (gloo) is used only as the launcher-side rendezvous -- broadcasting the RCCL unique id,
barriers, and, for rehearsals on a 1-GPU box, carrying the library's collectives through host memory
(``relay=True``).  The data path of a real multi-GPU run is RCCL inside libsafebo.so.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L


def shard_planes(planes: int, stride: int, world: int):
    """Canonical shard layout (same arithmetic as sbo_candidates_grid_sharded): rank r owns hyper-planes
    [r P / W, (r+1) P / W) of the slowest axis -> flat offsets first_of[0..W]."""
    return [(planes * r // world) * stride for r in range(world + 1)]


def merge_slots(rows, is_max):
    """Merge per-rank (value, index) arg-reduce slots -- the host half of collective C3.
    rows: list over ranks of lists of (value, index) with index < 0 meaning "none"; ties -> lowest index."""
    nslots = len(rows[0])
    out = []
    for t in range(nslots):
        best_v, best_i = 0.0, -1
        for row in rows:
            v, i = row[t]
            if i < 0:
                continue
            take = best_i < 0 or (v > best_v if is_max[t] else v < best_v) or (v == best_v and i < best_i)
            if take:
                best_v, best_i = v, i
        out.append((best_v, best_i))
    return out


class GlooRelay:
    """Callbacks for sbo_comm_init_relay backed by a torch.distributed (gloo) group."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist
        self._torch, self._dist, self._group = torch, dist, group
        self.allreduce = L.RELAY_ALLREDUCE(self._allreduce)
        self.allgather = L.RELAY_ALLGATHER(self._allgather)

    def _allreduce(self, user, buf, count, elem, op):
        try:
            torch, dist = self._torch, self._dist
            ops = {0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}
            if elem == 0:   # uint64 compared through an order-preserving int64 image
                arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint64)), shape=(count,))
                img = (arr ^ np.uint64(1 << 63)).view(np.int64).copy()
                t = torch.from_numpy(img)
                dist.all_reduce(t, op=ops[op], group=self._group)
                arr[:] = t.numpy().view(np.uint64) ^ np.uint64(1 << 63)
            else:
                arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double)), shape=(count,))
                t = torch.from_numpy(arr.copy())
                dist.all_reduce(t, op=ops[op], group=self._group)
                arr[:] = t.numpy()
            return 0
        except Exception as exc:   # never let an exception cross the C boundary
            print("relay all-reduce failed:", exc)
            return 1

    def _allgather(self, user, send, recv, nbytes):
        try:
            torch, dist = self._torch, self._dist
            world = dist.get_world_size(self._group)
            src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
            dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,))
            outs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(outs, torch.from_numpy(src.copy()), group=self._group)
            for r, t in enumerate(outs):
                dst[r * nbytes:(r + 1) * nbytes] = t.numpy()
            return 0
        except Exception as exc:
            print("relay all-gather failed:", exc)
            return 1


def join(engine, relay: bool = False, group=None):
    """Join ``engine`` to the ranks of the initialised torch.distributed group: RCCL by default
    (unique id broadcast from rank 0), or the host relay for rehearsals."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if relay:
        cb = GlooRelay(group)
        engine._relay = cb   # keep the ctypes thunks alive as long as the engine
        L.check(engine._lib.sbo_comm_init_relay(engine._ctx, world, rank, cb.allreduce, cb.allgather, None))
        engine.world, engine.rank = world, rank
        return
    # the broadcast always happens, also when rank 0 could not create the id: every rank then raises the same error
    # instead of some of them waiting in the broadcast for a rank that has already left
    uid = [None]
    if rank == 0:
        try:
            uid[0] = engine.comm_unique_id()
        except Exception as exc:                  # noqa: BLE001
            uid[0] = ("error", str(exc))
    dist.broadcast_object_list(uid, src=0, group=group)
    if isinstance(uid[0], tuple):
        raise L.SafeBOError(L.SBO_E_COMM, f"rank 0 could not create the RCCL unique id: {uid[0][1]}")
    engine.comm_init(world, rank, uid[0])


def join_with_fallback(engine, group=None, allow_relay: bool = True) -> str:
    """RCCL if every rank can form the communicator.  Otherwise, with ``allow_relay`` (one-GPU rehearsals of the N > 1
    plumbing, where RCCL refuses two ranks on one device), every rank falls back to the gloo relay; without it every rank
    raises -- a real multi-GPU run must never publish a relay-speed number.  Returns the transport used."""
    import torch
    import torch.distributed as dist
    ok, err = 1, None
    try:
        join(engine, relay=False, group=group)
    except Exception as exc:                      # noqa: BLE001 - any failure means "no RCCL here"
        err = exc
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag[0]) == 1:
        return "rccl"
    if not allow_relay:
        raise L.SafeBOError(L.SBO_E_COMM, f"[rank {dist.get_rank(group)}] the RCCL communicator could not be formed on every rank"
                            + (f" (this rank: {err})" if err is not None else " (this rank was fine)"))
    if err is not None:
        print(f"[rank {dist.get_rank(group)}] RCCL communicator failed ({err}); using the gloo relay")
    join(engine, relay=True, group=group)
    return "gloo-relay"


def init_gloo_from_env():
    """init_process_group(gloo) from the torchrun environment (RANK / WORLD_SIZE / MASTER_*)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group(backend="gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    return dist
