"""Multi-GPU plumbing: one process per GPU, RCCL over xGMI through the C ABI (include/fsdp.h fsdp_comm_*).  No PyTorch.

Frames are independent, so there is NO data-path collective: each rank plans its own contiguous shard.
RCCL carries only the start-up broadcast of constant tables (skidpad track table, consistency check of the constant
previous path) and the benchmark's barrier / max-reduction.

Launch contract: the launcher (``python -m torch.distributed.run``, or anything else) sets RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT; this module only reads the environment.  The one thing RCCL needs out of band is its 128-byte
unique id: rank 0 creates it (``fsdp_comm_unique_id``) and serves it to the other ranks over one TCP connection each
(stdlib sockets).  MASTER_PORT itself belongs to the launcher's store, so the exchange uses the first free port of
``[MASTER_PORT + 1, MASTER_PORT + PORT_SPAN]``; clients find it by a handshake that carries a launch key (the launcher's pid,
which all ranks of one launch share, plus WORLD_SIZE), so a port held by a stranger is skipped.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import socket
import struct
import time

import numpy as np

ID_BYTES = 128
PORT_SPAN = 32
_MAGIC = b"FSDPID1\0"


def _launch_key(world: int) -> bytes:
    """16 bytes every rank of one launch computes alike: run id / launcher pid / world size."""
    run = os.environ.get("FSDP_LAUNCH_KEY") or f"{os.environ.get('TORCHELASTIC_RUN_ID', '')}:{os.getppid()}"
    return hashlib.sha256(f"{run}:{world}:{os.environ.get('MASTER_PORT', '')}".encode()).digest()[:16]


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = b""
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed during the unique-id exchange")
        buf += chunk
    return buf


def serve_unique_id(uid: bytes, world: int, addr: str, base_port: int, timeout: float = 300.0) -> None:
    """Rank 0: hand `uid` to the world - 1 other ranks (each connects once, proves the launch key, names its rank)."""
    assert len(uid) == ID_BYTES
    key = _launch_key(world)
    srv = None
    for port in range(base_port + 1, base_port + 1 + PORT_SPAN):
        s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            s.bind((addr, port))
            s.listen(world)
            srv = s
            break
        except OSError:
            s.close()
    if srv is None:
        raise RuntimeError(f"no free port in [{base_port + 1}, {base_port + PORT_SPAN}] for the RCCL unique-id exchange")
    deadline = time.monotonic() + timeout
    served = set()
    try:
        while len(served) < world - 1:
            srv.settimeout(max(0.1, deadline - time.monotonic()))
            try:
                conn, _ = srv.accept()
            except socket.timeout:
                raise TimeoutError(f"unique-id exchange: only ranks {sorted(served)} of {world - 1} connected") from None
            with conn:
                conn.settimeout(10.0)
                try:
                    hello = _recv_exact(conn, len(_MAGIC) + 16 + 4)
                except (ConnectionError, socket.timeout):
                    continue
                rank = struct.unpack("<i", hello[-4:])[0]
                if hello[: len(_MAGIC)] != _MAGIC or hello[len(_MAGIC) : len(_MAGIC) + 16] != key or not (0 < rank < world):
                    conn.sendall(b"NO")  # a stranger, or a rank of another launch
                    continue
                conn.sendall(b"OK" + uid)
                served.add(rank)
    finally:
        srv.close()


def fetch_unique_id(rank: int, world: int, addr: str, base_port: int, timeout: float = 300.0) -> bytes:
    """Ranks > 0: find rank 0's port in the span and fetch the id."""
    hello = _MAGIC + _launch_key(world) + struct.pack("<i", rank)
    deadline = time.monotonic() + timeout
    while time.monotonic() < deadline:
        for port in range(base_port + 1, base_port + 1 + PORT_SPAN):
            try:
                with socket.create_connection((addr, port), timeout=2.0) as s:
                    s.settimeout(10.0)
                    s.sendall(hello)
                    if _recv_exact(s, 2) == b"OK":
                        return _recv_exact(s, ID_BYTES)
            except (OSError, ConnectionError):
                continue
        time.sleep(0.05)
    raise TimeoutError("unique-id exchange: rank 0 not found")


def exchange_unique_id(rank: int, world: int, make_id, addr: str | None = None, base_port: int | None = None) -> bytes:
    """All ranks call this; returns the same 128 bytes everywhere.  make_id() runs on rank 0 only."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    base_port = int(os.environ.get("MASTER_PORT", "29531")) if base_port is None else base_port
    if rank == 0:
        uid = bytes(make_id())
        if world > 1:
            serve_unique_id(uid, world, addr, base_port)
        return uid
    return fetch_unique_id(rank, world, addr, base_port)


def frame_range(rank: int, world: int, n_total: int):
    """Contiguous shard [lo, hi) of a global batch (strong-scaling form, SURVEY.md 8e: [g*B/G, (g+1)*B/G))."""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    return lo, min(lo + per, n_total)


class Dist:
    """Rank bookkeeping + the RCCL communicator of one context.  ``Dist(ctx)`` with WORLD_SIZE == 1 does nothing
    unless FSDP_FORCE_DIST=1 (a one-rank communicator, to exercise the RCCL plumbing on a single GPU)."""

    def __init__(self, ctx=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self._ctx = None
        self._active = False
        if ctx is not None:
            self.attach(ctx)

    def attach(self, ctx) -> None:
        """Create this rank's communicator on ctx's GPU (collective: every rank must call)."""
        self._ctx = ctx
        if not (self.world > 1 or os.environ.get("FSDP_FORCE_DIST") == "1"):
            return
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC
        lib = ctx._lib

        def make_id():
            buf = ctypes.create_string_buffer(ID_BYTES)
            ctx._check(lib.fsdp_comm_unique_id(buf), "fsdp_comm_unique_id")
            return buf.raw

        uid = exchange_unique_id(self.rank, self.world, make_id)
        ctx._check(lib.fsdp_comm_init(ctx._h, self.rank, self.world, uid), "fsdp_comm_init")
        self._active = True
        assert self.comm_size == self.world and int(lib.fsdp_comm_rank(ctx._h)) == self.rank

    @property
    def comm_size(self) -> int:
        """Rank count RCCL reports (ncclCommCount); 1 without a communicator."""
        return int(self._ctx._lib.fsdp_comm_size(self._ctx._h)) if self._active else 1

    def shard_seed(self, base_seed: int) -> int:
        """Weak scaling: each rank replays its own synthetic track (fixed frames per GPU)."""
        return base_seed + self.rank

    def frame_range(self, n_total: int):
        return frame_range(self.rank, self.world, n_total)

    def broadcast_array(self, arr, shape, dtype=np.float64, src: int = 0) -> np.ndarray:
        """Track-map broadcast: rank `src` owns a constant table (skidpad known path 5786 x 2 f64 =
        92 576 B, noise table) and broadcasts it once at start-up; the other ranks pass arr=None."""
        if self.rank == src:
            buf = np.ascontiguousarray(arr, dtype=dtype).reshape(shape).copy()
        else:
            buf = np.zeros(shape, dtype=dtype)
        if self._active:
            self._ctx._check(self._ctx._lib.fsdp_comm_broadcast(self._ctx._h, ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes), src),
                             "fsdp_comm_broadcast")
        return buf

    def broadcast_check_table(self, table: np.ndarray) -> bool:
        """Rank 0 broadcasts its copy of a constant table; True iff every rank's own copy has the same bits."""
        mine = np.ascontiguousarray(table, dtype=np.float64)
        ref = self.broadcast_array(mine if self.rank == 0 else None, mine.shape)
        same = 1.0 if np.array_equal(ref.view(np.uint64), mine.view(np.uint64)) else 0.0
        return self._reduce(same, 2) == 1.0

    def _reduce(self, value: float, op: int) -> float:
        if not self._active:
            return float(value)
        v = np.array([value], dtype=np.float64)
        self._ctx._check(self._ctx._lib.fsdp_comm_allreduce(self._ctx._h, v.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 1, op), "fsdp_comm_allreduce")
        return float(v[0])

    def barrier(self):
        """Waits for this rank's passes in flight, then for every rank (all-reduce rendezvous)."""
        if self._active:
            self._ctx._check(self._ctx._lib.fsdp_comm_barrier(self._ctx._h), "fsdp_comm_barrier")
        elif self._ctx is not None:
            self._ctx.sync()

    def max_over_ranks(self, value: float) -> float:
        return self._reduce(value, 1)

    def sum_over_ranks(self, value: float) -> float:
        return self._reduce(value, 0)

    def close(self):
        if self._active:
            self._ctx._lib.fsdp_comm_destroy(self._ctx._h)
            self._active = False
