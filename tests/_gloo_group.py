"""A torch.distributed (gloo) group with the five methods safebo_amd.distributed expects of a group -- test-side only: the
product package's own rendezvous is the stdlib TcpGroup; this adapter lets the world_size-2 CPU test drive the same relay /
merge code over gloo."""
import os

import numpy as np


class GlooGroup:
    def __init__(self):
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        self._torch, self._dist = torch, dist

    def get_rank(self):
        return self._dist.get_rank()

    def get_world_size(self):
        return self._dist.get_world_size()

    def barrier(self):
        self._dist.barrier()

    def broadcast_bytes(self, payload, src=0):
        torch, dist = self._torch, self._dist
        n = torch.tensor([len(payload) if dist.get_rank() == src else 0], dtype=torch.int64)
        dist.broadcast(n, src=src)
        buf = torch.frombuffer(bytearray(payload), dtype=torch.uint8) if dist.get_rank() == src and len(payload) else torch.empty(int(n[0]), dtype=torch.uint8)
        if int(n[0]):
            dist.broadcast(buf, src=src)
        return bytes(buf.numpy().tobytes())

    def all_reduce(self, arr, op="sum"):
        torch, dist = self._torch, self._dist
        ops = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}
        a = np.ascontiguousarray(arr)
        if a.dtype == np.uint64:      # compared through an order-preserving int64 image (gloo has no uint64)
            img = (a ^ np.uint64(1 << 63)).view(np.int64).copy()
            t = torch.from_numpy(img)
            dist.all_reduce(t, op=ops[op])
            return (t.numpy().view(np.uint64) ^ np.uint64(1 << 63)).reshape(a.shape)
        t = torch.from_numpy(a.copy())
        dist.all_reduce(t, op=ops[op])
        return t.numpy().reshape(a.shape)

    def all_gather_bytes(self, payload):
        torch, dist = self._torch, self._dist
        src = torch.frombuffer(bytearray(payload), dtype=torch.uint8) if len(payload) else torch.empty(0, dtype=torch.uint8)
        outs = [torch.empty(len(payload), dtype=torch.uint8) for _ in range(dist.get_world_size())]
        dist.all_gather(outs, src)
        return [bytes(t.numpy().tobytes()) for t in outs]

    def destroy(self):
        self._dist.destroy_process_group()
