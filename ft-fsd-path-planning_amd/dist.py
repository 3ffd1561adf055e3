"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).  Frames are independent, so there is NO data-path collective: each
rank plans its own shard.  The only collectives are the start-up broadcast of the constant
previous-path table (consistency check across GPUs) and the timing barrier / max-reduction.
"""
from __future__ import annotations

import os

import numpy as np


class Dist:
    def __init__(self, backend: str | None = None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = None
        self._dist = None
        self._torch = None
        # FSDP_FORCE_DIST=1 initialises the process group even for a single rank (exercises the RCCL plumbing on one GPU)
        self._active = self.world > 1 or os.environ.get("FSDP_FORCE_DIST") == "1"
        if self._active:
            import torch
            import torch.distributed as dist

            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            self._torch, self._dist = torch, dist
            self.backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29531")
                dist.init_process_group(backend=self.backend, rank=self.rank, world_size=self.world)

    @property
    def device(self):
        return "cuda" if self.backend == "nccl" else "cpu"

    def shard_seed(self, base_seed: int) -> int:
        """Each rank replays its own synthetic track (weak scaling: fixed frames per GPU)."""
        return base_seed + self.rank

    def frame_range(self, n_total: int):
        """Contiguous shard [lo, hi) of a global batch (strong-scaling form)."""
        per = (n_total + self.world - 1) // self.world
        lo = min(self.rank * per, n_total)
        return lo, min(lo + per, n_total)

    def broadcast_check_table(self, table: np.ndarray) -> bool:
        """Rank 0 broadcasts the constant previous-path table; every rank compares with its own copy."""
        if not self._active:
            return True
        t = self._torch.from_numpy(np.ascontiguousarray(table, dtype=np.float64)).to(self.device)
        ref = t.clone()
        self._dist.broadcast(ref, src=0)
        same = bool(self._torch.equal(ref, t))
        flag = self._torch.tensor([1 if same else 0], dtype=self._torch.int32, device=self.device)
        self._dist.all_reduce(flag, op=self._dist.ReduceOp.MIN)
        return bool(flag.item() == 1)

    def broadcast_array(self, arr: np.ndarray | None, shape, src: int = 0) -> np.ndarray:
        """Track-map broadcast: rank `src` owns the constant skidpad tables (known path 5786x2 f64 =
        92 576 B, noise table, reference centres) and broadcasts them once at start-up; the other ranks pass arr=None."""
        if not self._active:
            return np.ascontiguousarray(arr, dtype=np.float64)
        t = self._torch.zeros(tuple(shape), dtype=self._torch.float64, device=self.device)
        if self.rank == src:
            t.copy_(self._torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64)))
        self._dist.broadcast(t, src=src)
        return t.cpu().numpy()

    def barrier(self):
        if self._active:
            if self.backend == "nccl":
                self._torch.cuda.synchronize()
            self._dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if not self._active:
            return float(value)
        t = self._torch.tensor([value], dtype=self._torch.float64, device=self.device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value: float) -> float:
        if not self._active:
            return float(value)
        t = self._torch.tensor([value], dtype=self._torch.float64, device=self.device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self._active and self._dist.is_initialized():
            self._dist.destroy_process_group()
