"""Rank plumbing for multi-GPU sweeps: one process per GPU.

The data path of a multi-GPU run is RCCL inside libsafebo.so.  What the host side needs besides is a rendezvous among the
ranks a launcher started (``torch.distributed.run`` or anything else that sets RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT):
broadcasting the 128-byte RCCL unique id, barriers, a max over the ranks' timings, and -- for rehearsals on a 1-GPU box only
-- carrying the library's collectives through host memory.  ``TcpGroup`` does that with the standard library (a star over TCP
through rank 0, struct-framed byte strings, connections accepted on an HMAC challenge over the job's secret): no PyTorch in the
product package.  ``HostRelay`` and ``join`` accept any group object with the same five
methods (the CPU tests drive them with a torch.distributed / gloo adapter that lives under tests/).
"""
from __future__ import annotations

import ctypes as C
import hashlib
import hmac
import os
import socket
import struct
import time

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


_MAGIC = b"SBO-RDZV2"
_OPS = {"sum": np.add, "max": np.maximum, "min": np.minimum}
_MAX_FRAME = 1 << 30          # no message of this rendezvous is anywhere near it: a larger length word is a stranger or a bug
_HELLO_TIMEOUT = 2.0          # seconds a connection gets to say who it is


def _send(sock, payload: bytes):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv(sock, limit: int = _MAX_FRAME) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > limit:
        raise ConnectionError(f"rendezvous frame of {n} bytes refused (limit {limit})")
    return _recv_exact(sock, n)


def _pack_list(parts) -> bytes:
    """[bytes] -> count, lengths, bodies (fixed little-endian framing: nothing on this wire is ever unpickled)"""
    return struct.pack("<I", len(parts)) + b"".join(struct.pack("<Q", len(p_)) for p_ in parts) + b"".join(parts)


def _unpack_list(blob: bytes) -> list:
    (n,) = struct.unpack_from("<I", blob, 0)
    if 4 + 8 * n > len(blob):
        raise ValueError("rendezvous list frame is truncated")
    lens = struct.unpack_from(f"<{n}Q", blob, 4)
    off, out = 4 + 8 * n, []
    if off + sum(lens) != len(blob):
        raise ValueError("rendezvous list frame has the wrong length")
    for ln in lens:
        out.append(blob[off:off + ln])
        off += ln
    return out


def _is_loopback(addr: str) -> bool:
    return addr in ("localhost", "::1") or addr.startswith("127.")


def job_secret(base_port: int, world: int, addr: str = "127.0.0.1") -> bytes:
    """The key both sides of the hello prove they hold.  SBO_RDZV_SECRET (set by whoever launches the ranks) or, failing that, the
    launcher's run id -- mixed with the job's port and size so that two jobs never accept each other's ranks.  Without either the key
    would be derivable from the public port and world size and the hello would authenticate nobody: across hosts (`addr` not
    loopback) that is refused outright; on loopback, where only local users can reach the port, a constant is used with a warning
    (r05)."""
    raw = os.environ.get("SBO_RDZV_SECRET") or os.environ.get("TORCHELASTIC_RUN_ID")
    if not raw:
        if world > 1 and not _is_loopback(addr):
            raise RuntimeError("rendezvous across hosts needs a job secret: set SBO_RDZV_SECRET (or launch with torch.distributed.run, "
                               "whose TORCHELASTIC_RUN_ID serves) -- without one any process that can reach the port could claim a rank")
        if world > 1:
            import warnings
            warnings.warn("safebo rendezvous without SBO_RDZV_SECRET / TORCHELASTIC_RUN_ID: any local process can join this job's ranks",
                          RuntimeWarning, stacklevel=3)
        raw = "safebo-rendezvous"
    return hashlib.sha256(raw.encode() + struct.pack("<II", base_port, world)).digest()


_PORT_WINDOW = 16


def rendezvous_ports(base_port: int) -> list:
    """Where rank 0 listens, known to every rank before anyone connects: exactly SBO_RDZV_PORT when the launcher sets it; else
    the first port of [MASTER_PORT + 1, MASTER_PORT + 16] that rank 0 can bind (MASTER_PORT itself belongs to the launcher's
    store; a neighbour in the ephemeral range may be somebody's outgoing connection).  Whoever else listens in that window is
    harmless: a rank joins only the listener that proves the job's secret (and only struct-framed bytes ever cross)."""
    if "SBO_RDZV_PORT" in os.environ:
        return [int(os.environ["SBO_RDZV_PORT"])]
    return [base_port + k for k in range(1, _PORT_WINDOW + 1)]


class TcpGroup:
    """Star rendezvous over TCP: rank 0 listens on the first free port of `rendezvous_ports`, the others connect to it; each connection is accepted
    only after a challenge-response on the job's secret (HMAC-SHA256 over a fresh nonce, both directions), a rank in
    [1, world) and no rank twice.  Every collective is "send to rank 0, combine there, send back", framed with struct: byte
    strings and NumPy buffers only.  Small payloads: ids, scalars, and the rehearsal relay."""

    def __init__(self, rank: int, world: int, addr: str = "127.0.0.1", base_port: int = 29500, timeout: float = 120.0):
        if world < 1 or not 0 <= rank < world:
            raise ValueError(f"rank {rank} outside a world of {world}")
        self.rank, self.world, self._peers, self._sock = rank, world, [], None
        key = job_secret(base_port, world, addr)
        ports = rendezvous_ports(base_port)
        deadline = time.time() + timeout
        if world == 1:
            return

        def mac(*parts):
            return hmac.new(key, b"".join(parts), hashlib.sha256).digest()

        if rank == 0:
            srv, why = None, None
            for port in ports:
                try:
                    srv = socket.create_server((addr, port), reuse_port=False)
                    break
                except OSError as exc:
                    why = exc
            if srv is None:
                raise OSError(f"no rendezvous port free on {addr} among {ports[0]}..{ports[-1]} ({why}); set SBO_RDZV_PORT on every rank")
            srv.settimeout(0.25)
            peers = {}
            try:
                while len(peers) < world - 1:
                    if time.time() > deadline:
                        raise TimeoutError(f"rendezvous: {world - 1 - len(peers)} rank(s) did not arrive")
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        continue
                    # a connection has _HELLO_TIMEOUT to prove itself; silence, strangers and repeats are dropped and cost
                    # the others that much at most
                    conn.settimeout(_HELLO_TIMEOUT)
                    try:
                        nonce = os.urandom(32)
                        conn.sendall(_MAGIC + nonce)
                        hello = _recv_exact(conn, len(_MAGIC) + 4 + 32 + 32)
                        r = struct.unpack_from("<I", hello, len(_MAGIC))[0]
                        theirs = hello[len(_MAGIC) + 4:len(_MAGIC) + 36]
                        good = (hello.startswith(_MAGIC) and 1 <= r < world and r not in peers and
                                hmac.compare_digest(hello[-32:], mac(b"rank", nonce, theirs, struct.pack("<I", r))))
                        if not good:
                            conn.close()
                            continue
                        conn.sendall(mac(b"root", theirs, nonce))
                    except (OSError, ConnectionError, struct.error):
                        conn.close()
                        continue
                    conn.settimeout(timeout)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    peers[r] = conn
            except BaseException:
                for conn in peers.values():
                    conn.close()
                raise
            finally:
                srv.close()
            self._peers = [peers[r] for r in range(1, world)]
        else:
            attempt = 0
            while self._sock is None:
                s_ = None
                port = ports[attempt % len(ports)]
                attempt += 1
                try:
                    s_ = socket.create_connection((addr, port), timeout=_HELLO_TIMEOUT)
                    s_.settimeout(_HELLO_TIMEOUT)
                    first = _recv_exact(s_, len(_MAGIC) + 32)
                    if not first.startswith(_MAGIC):
                        raise ConnectionError(f"port {port} on {addr} does not speak this rendezvous")
                    nonce, mine = first[len(_MAGIC):], os.urandom(32)
                    rk = struct.pack("<I", rank)
                    s_.sendall(_MAGIC + rk + mine + mac(b"rank", nonce, mine, rk))
                    if not hmac.compare_digest(_recv_exact(s_, 32), mac(b"root", mine, nonce)):
                        raise ConnectionError("rank 0 did not prove the job secret")
                    s_.settimeout(timeout)
                    s_.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self._sock = s_
                except (OSError, ConnectionError, struct.error) as exc:
                    if s_ is not None:
                        s_.close()
                    if time.time() > deadline:
                        raise TimeoutError(f"rendezvous: rank 0 was not reached on {addr}:{ports[0]}..{ports[-1]} ({exc})") from exc
                    if attempt % len(ports) == 0:
                        time.sleep(0.05)

    # -- the five methods a "group" has ----------------------------------------------------------------------------------
    def get_rank(self) -> int:
        return self.rank

    def get_world_size(self) -> int:
        return self.world

    def _exchange(self, payload: bytes, combine):
        """every rank contributes `payload`; rank 0 computes combine([payload_0, ..]) -> list of per-rank answers"""
        if self.world == 1:
            return combine([payload])[0]
        if self.rank == 0:
            parts = [payload] + [_recv(p) for p in self._peers]
            answers = combine(parts)
            for p, a in zip(self._peers, answers[1:]):
                _send(p, a)
            return answers[0]
        _send(self._sock, payload)
        return _recv(self._sock)

    def barrier(self):
        self._exchange(b"", lambda parts: [b""] * len(parts))

    def broadcast_bytes(self, payload, src: int = 0) -> bytes:
        """the byte string of rank `src` on every rank (the other ranks' argument is ignored)"""
        if not 0 <= src < self.world:
            raise ValueError(f"broadcast source {src} outside a world of {self.world}")

        def combine(parts):
            return [parts[src]] * len(parts)
        return self._exchange(bytes(payload) if self.rank == src else b"", combine)

    def all_reduce(self, arr: np.ndarray, op: str = "sum") -> np.ndarray:
        """element-wise sum / max / min over the ranks of a NumPy array (same shape and dtype everywhere)"""
        a = np.ascontiguousarray(arr)
        fn = _OPS[op]

        def combine(parts):
            if any(len(p_) != a.nbytes for p_ in parts):
                raise ValueError("all_reduce: the ranks sent buffers of different sizes")
            acc = np.frombuffer(parts[0], dtype=a.dtype).copy()
            for p_ in parts[1:]:
                acc = fn(acc, np.frombuffer(p_, dtype=a.dtype))
            return [acc.tobytes()] * len(parts)
        out = self._exchange(a.tobytes(), combine)
        if len(out) != a.nbytes:
            raise ValueError("all_reduce: answer of the wrong size")
        return np.frombuffer(out, dtype=a.dtype).reshape(a.shape).copy()

    def all_gather_bytes(self, payload: bytes) -> list:
        def combine(parts):
            blob = _pack_list(parts)
            return [blob] * len(parts)
        out = _unpack_list(self._exchange(bytes(payload), combine))
        if len(out) != self.world:
            raise ValueError("all_gather: answer with the wrong number of parts")
        return out

    def destroy(self):
        for p in self._peers:
            p.close()
        if self._sock is not None:
            self._sock.close()
        self._peers, self._sock = [], None


class HostRelay:
    """Callbacks for sbo_comm_init_relay: the library's collectives staged through host memory and carried by a group object
    (TcpGroup, or the gloo adapter of the CPU tests).  A rehearsal transport -- RCCL over xGMI is the production one."""

    def __init__(self, group):
        self._group = group
        self.allreduce = L.RELAY_ALLREDUCE(self._allreduce)
        self.allgather = L.RELAY_ALLGATHER(self._allgather)

    def _allreduce(self, user, buf, count, elem, op):
        try:
            opname = {0: "sum", 1: "max", 2: "min"}[op]
            if elem == 0:
                arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint64)), shape=(count,))
            else:
                arr = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double)), shape=(count,))
            arr[:] = self._group.all_reduce(arr.copy(), opname)
            return 0
        except Exception as exc:   # never let an exception cross the C boundary
            print("relay all-reduce failed:", exc)
            return 1

    def _allgather(self, user, send, recv, nbytes):
        try:
            world = self._group.get_world_size()
            src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
            dst = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,))
            for r, part in enumerate(self._group.all_gather_bytes(src.tobytes())):
                dst[r * nbytes:(r + 1) * nbytes] = np.frombuffer(part, dtype=np.uint8)
            return 0
        except Exception as exc:
            print("relay all-gather failed:", exc)
            return 1


def join(engine, group, relay: bool = False):
    """Join ``engine`` to the ranks of ``group``: RCCL by default (unique id broadcast from rank 0), or the host relay for
    rehearsals."""
    world, rank = group.get_world_size(), group.get_rank()
    if relay:
        cb = HostRelay(group)
        engine._relay = cb   # keep the ctypes thunks alive as long as the engine
        L.check(engine._lib.sbo_comm_init_relay(engine._ctx, world, rank, cb.allreduce, cb.allgather, None))
        engine.world, engine.rank = world, rank
        return
    # the broadcast always happens, also when rank 0 could not create the id: every rank then raises the same error
    # instead of some of them waiting in the broadcast for a rank that has already left
    msg = b""
    if rank == 0:
        try:
            msg = b"\x00" + bytes(engine.comm_unique_id())
        except Exception as exc:                  # noqa: BLE001
            msg = b"\x01" + str(exc).encode("utf-8", "replace")
    msg = group.broadcast_bytes(msg, src=0)
    if msg[:1] != b"\x00":
        raise L.SafeBOError(L.SBO_E_COMM, f"rank 0 could not create the RCCL unique id: {msg[1:].decode('utf-8', 'replace')}")
    uid = msg[1:]
    engine.comm_init(world, rank, uid)


def join_with_fallback(engine, group, allow_relay: bool = True) -> str:
    """RCCL if every rank can form the communicator.  Otherwise, with ``allow_relay`` (one-GPU rehearsals of the N > 1
    plumbing, where RCCL refuses two ranks on one device), every rank falls back to the host relay; without it every rank
    raises -- a real multi-GPU run must never publish a relay-speed number.  Returns the transport used."""
    ok, err = 1, None
    try:
        join(engine, group, relay=False)
    except Exception as exc:                      # noqa: BLE001 - any failure means "no RCCL here"
        err = exc
        ok = 0
    flag = group.all_reduce(np.array([ok], dtype=np.int64), "min")
    if int(flag[0]) == 1:
        return "rccl"
    if not allow_relay:
        raise L.SafeBOError(L.SBO_E_COMM, f"[rank {group.get_rank()}] the RCCL communicator could not be formed on every rank"
                            + (f" (this rank: {err})" if err is not None else " (this rank was fine)"))
    if err is not None:
        print(f"[rank {group.get_rank()}] RCCL communicator failed ({err}); using the host relay")
    join(engine, group, relay=True)
    return "host-relay"


def init_from_env(timeout: float = 120.0) -> TcpGroup:
    """The ranks of a launcher (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, as torch.distributed.run sets them) as a TcpGroup."""
    return TcpGroup(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
                    os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")), timeout)
