"""Several GPUs from ONE process and ONE host thread ("one handle + stream per GPU ... inputs H2D and
outputs D2H per GPU").

``MultiPlanner(devices=None)`` owns one context (= ``fsdp_ctx``: streams, pass slots, device buffers, the constant
tables) per device, cuts a batch of independent frames into contiguous ranges ``[g*B/G, (g+1)*B/G)``, hands every range
to its context with the asynchronous ``fsdp_submit`` and collects the tickets in order.  No launcher, no socket, no RCCL:
frames never talk to each other, so the only thing the GPUs share is the host thread that feeds them.  Results are the
bytes of one ``plan_batch`` over the whole batch (a frame's result does not depend on the batch it travels in:
tests/test_gpu_parity.py::test_batch_composition_invariance, tests/test_multi_gpu.py).

``MultiSkidpadBatch`` shards stateful skidpad planner INSTANCES the same way (BASELINE config 5): instance i lives on
one GPU for its whole life.

The process-per-GPU mode of bench.py / dist.py (RANK / LOCAL_RANK / WORLD_SIZE, RCCL for the start-up table broadcast)
stays what it is; this is the form a user of the Python API calls.

Reference: there is none — full_pipeline/full_pipeline.py:84-207 plans one frame per call on one CPU thread.  The
semantics mirrored here are those of ``PathPlanner.plan_batch`` (planner.py): independent frames, fresh-planner
previous path unless ``prev_paths`` is given.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from . import _capi


def shard_ranges(n: int, parts: int) -> List[Tuple[int, int]]:
    """Contiguous ranges [g*n/G, (g+1)*n/G) — the same cut as dist.Dist.frame_range for ranks."""
    return [(g * n // parts, (g + 1) * n // parts) for g in range(parts)]


def visible_devices() -> List[int]:
    return list(range(int(_capi.load().fsdp_device_count())))


class _Staging:
    """Page-locked input buffers of one shard: fsdp_submit reads page-locked memory from inside the slot's own kernels
    (no blocking copy), pageable memory is staged by the runtime and blocks the host thread — which would serialise the
    GPUs behind one another.  Only batches that are NOT already page-locked come through here (MultiPlanner.submit)."""

    def __init__(self):
        self.off = self.cones = self.poses = self.prev = None

    @staticmethod
    def _fit(buf, shape, dtype):
        n = int(np.prod(shape))
        if buf is None or buf.size < n:
            buf = _capi.pinned_empty(max(n, 1) * 5 // 4 + 16, dtype)
        return buf, buf[:n].reshape(shape)

    def load(self, off, cones, poses, prev):
        self.off, o = self._fit(self.off, off.shape, np.int32)
        self.cones, c = self._fit(self.cones, cones.shape, np.float64)
        self.poses, p = self._fit(self.poses, poses.shape, np.float64)
        np.subtract(off, off[0], out=o)
        c[...] = cones
        p[...] = poses
        q = None
        if prev is not None:
            self.prev, q = self._fit(self.prev, prev.shape, np.float64)
            q[...] = prev
        return o, c, p, q


class MultiTicket:
    __slots__ = ("parts", "out", "host_s")

    def __init__(self, parts, out, host_s=0.0):
        self.parts, self.out, self.host_s = parts, out, host_s


class MultiPlanner:
    """plan_batch over every visible GPU (or the given ``devices``; a device may appear more than once — two contexts
    on one GPU is how the 1-GPU test box exercises this class).

    What the one calling thread does per batch (round 4 copied ~5.5 KB per frame through it — 26 GB/s of memcpy at ONE GPU's
    streaming rate, so the class could not scale): a batch whose arrays are page-locked (``pinned_empty`` / ``pinned_copy`` /
    ``fsdp_host_register``) is sharded ZERO-COPY — every context gets pointers into the caller's own arrays
    (``Context.submit_slice``; include/fsdp.h: cone_offsets[0] need not be 0) and reads its slice over PCIe from inside its
    sorting kernel; a pageable batch is staged into page-locked buffers by one worker thread per context, in parallel (NumPy's
    copies release the GIL); results are written by the GPUs straight into one page-locked array of the whole batch, which is
    what ``collect`` / ``plan_batch`` / ``plan_stream`` RETURN — no copy; the block goes back to a free list when the last
    reference to it is dropped (``_capi.PinnedPool``)."""

    STAGE_THREADS_MIN_FRAMES = 256  # below this a shard is staged inline (a thread hand-off costs ~50 us)

    def __init__(self, devices: Sequence[int] | None = None, params: dict | None = None, mission: int = 4, overlap: int = 2):
        devices = visible_devices() if devices is None else [int(d) for d in devices]
        if not devices:
            raise _capi.FsdpError("MultiPlanner: no GPU visible (this package has no CPU fallback)")
        self.devices = devices
        self.ctx = [_capi.Context(device=d, mission=mission, params=params) for d in devices]
        self._pool = _capi.PinnedPool()
        self._workers = None
        self.host_seconds = 0.0   # time the calling thread spent inside submit() since the last reset_host_time()
        self.host_frames = 0
        self.zero_copy_batches = 0
        self.staged_batches = 0
        # staging sets per context: a batch's inputs must stay untouched until its ticket is collected, and `overlap`
        # batches per context may be in flight
        self._overlap = 0
        self.set_overlap(overlap)

    def set_overlap(self, depth: int):
        """Batches in flight per context (fsdp_set_overlap): a stream of batches keeps `depth` of them on every GPU."""
        depth = max(1, int(depth))
        for c in self.ctx:
            c.set_overlap(depth)
        self._overlap = depth
        self._stage = [[_Staging() for _ in range(2 * depth)] for _ in self.ctx]
        self._turn = [0] * len(self.ctx)

    @property
    def horizon(self) -> int:
        return self.ctx[0].horizon

    def set_global_path(self, xy):
        for c in self.ctx:
            c.set_global_path(xy)

    def reset_host_time(self):
        self.host_seconds, self.host_frames = 0.0, 0

    def close(self):
        if self._workers is not None:
            for w in self._workers:
                w.shutdown(wait=True)
            self._workers = None
        for c in self.ctx:
            c.close()
        self._pool.close()

    # ---- one batch --------------------------------------------------------------------------------------------------
    def _stage_and_submit(self, g, lo, hi, offsets, cones, poses, prev, out, compact=False):
        st = self._stage[g][self._turn[g] % len(self._stage[g])]
        self._turn[g] += 1
        o, c, p, q = st.load(offsets[lo : hi + 1], cones[offsets[lo] : offsets[hi]], poses[lo:hi], None if prev is None else prev[lo:hi])
        return self.ctx[g].submit(o, c, p, q, out=out[lo:hi], compact=compact)

    def submit(self, offsets, cones, poses, prev_paths=None, out: np.ndarray | None = None, compact: bool = False) -> MultiTicket:
        """Cut the batch, enqueue every shard on its GPU, return at once.  ``out``: page-locked RESULT_DTYPE array of the
        whole batch (``pinned_empty``); every GPU writes its range of it.  Default: a block of the planner's pool.
        ``compact``: fsdp_compact_result records (path, sorted indices, status: ``Context.compact_dtype``, 1384 instead of 2408 bytes
        per frame over PCIe).

        Contract: page-locked input arrays are NOT copied (the GPUs read their slices in place) — they, and ``out``, must stay
        untouched until ``collect``.  If a shard cannot be submitted, the shards already on their GPUs are waited for before the
        exception leaves this method, so that neither the caller's arrays nor the result block are written behind his back."""
        import time

        t0 = time.perf_counter()
        offsets, cones, poses, n = _capi.Context._prep(offsets, cones, poses)
        prev = None if prev_paths is None else self.ctx[0].pad_paths(prev_paths)
        if prev is not None and len(prev) != n:
            raise ValueError("prev_paths: one (horizon, 4) path per frame")
        dt = self.ctx[0].compact_dtype if compact else self.ctx[0].result_dtype
        if out is None:
            out = self._pool.get(n, dt)
        assert out.dtype == dt and len(out) == n and out.flags.c_contiguous
        ranges = [(g, lo, hi) for g, (lo, hi) in enumerate(shard_ranges(n, len(self.ctx))) if hi > lo]
        zero_copy = n > 0 and _capi.is_pinned(offsets) and _capi.is_pinned(poses) and (len(cones) == 0 or _capi.is_pinned(cones)) and (
            prev is None or _capi.is_pinned(prev))
        parts, futs = [], []
        try:
            if zero_copy:
                self.zero_copy_batches += 1
                for g, lo, hi in ranges:
                    parts.append((g, self.ctx[g].submit_slice(lo, hi, offsets, cones, poses, prev, out, compact=compact)))
            elif len(ranges) > 1 and n >= self.STAGE_THREADS_MIN_FRAMES * len(ranges):
                # pageable input: every context's worker copies its shard into page-locked staging and submits it; the caller's
                # arrays are his again when submit returns
                self.staged_batches += 1
                if self._workers is None:
                    from concurrent.futures import ThreadPoolExecutor

                    self._workers = [ThreadPoolExecutor(max_workers=1, thread_name_prefix=f"fsdp-stage-{g}") for g in range(len(self.ctx))]
                futs = [(g, self._workers[g].submit(self._stage_and_submit, g, lo, hi, offsets, cones, poses, prev, out, compact)) for g, lo, hi in ranges]
                for g, f in futs:
                    parts.append((g, f.result()))
            else:
                self.staged_batches += 1
                for g, lo, hi in ranges:
                    parts.append((g, self._stage_and_submit(g, lo, hi, offsets, cones, poses, prev, out, compact)))
        except BaseException:
            # (round-5 advisor) shards already submitted write into `out` / read the caller's arrays: wait for them — and for the
            # workers still staging — before the block can go back to the pool and the exception to the caller
            for g, f in futs:
                try:
                    t = f.result()
                    if all(t is not pt for _, pt in parts):
                        parts.append((g, t))
                except BaseException:
                    pass
            self._drain(parts)
            raise
        el = time.perf_counter() - t0
        self.host_seconds += el
        self.host_frames += n
        return MultiTicket(parts, out, el)

    def _drain(self, parts):
        """Wait for shards whose results nobody will read (error paths): errors of the wait itself are swallowed."""
        for g, t in parts:
            try:
                self.ctx[g].collect(t)
            except Exception:
                pass

    def collect(self, ticket: MultiTicket) -> np.ndarray:
        """Wait for the batch; returns the page-locked result array the GPUs wrote (not a copy)."""
        parts, ticket.parts = ticket.parts, []
        for k, (g, t) in enumerate(parts):
            try:
                self.ctx[g].collect(t)
            except BaseException:
                self._drain(parts[k + 1 :])  # (the other GPUs still write their ranges of the block)
                raise
        return ticket.out

    def plan_batch(self, offsets, cones, poses, prev_paths=None, compact: bool = False) -> np.ndarray:
        """The bytes of ``Context.plan_batch`` / ``plan_batch_sequential`` over the whole batch, planned on all GPUs."""
        return self.collect(self.submit(offsets, cones, poses, prev_paths, compact=compact))

    # ---- a stream of batches ----------------------------------------------------------------------------------------
    def plan_stream(self, batches, depth: int | None = None, compact: bool = False):
        """Yield the results of an iterable of batches ``(offsets, cones, poses)`` in order, `depth` batches in flight on
        every GPU (default: two per pass slot — the contexts' ticket capacity: a slot's next batch is then already queued on
        its stream when the current one ends).  Every yielded array is the page-locked block its GPUs wrote; drop it (and
        its views) and the block serves a later batch."""
        depth = 2 * self._overlap if depth is None else max(1, min(int(depth), 2 * self._overlap))
        inflight = []
        try:
            for b in batches:
                if len(inflight) == depth:
                    yield self.collect(inflight.pop(0))
                inflight.append(self.submit(*b, compact=compact))
            while inflight:
                yield self.collect(inflight.pop(0))
        finally:
            # a generator closed early (or an exception part-way): the batches still in flight write into pool blocks that
            # would otherwise be handed to a later batch while the GPUs are not done with them (round-5 advisor)
            for t in inflight:
                self._drain(t.parts)
                t.parts = []


class MultiSkidpadBatch:
    """n stateful skidpad planners (BASELINE config 5) sharded over GPUs by instance: planner i lives in context
    ``g`` with ``lo_g <= i < hi_g``.  Same methods as ``SkidpadBatch``; results are those of one ``SkidpadBatch`` holding
    all n planners."""

    def __init__(self, n_instances: int, devices: Sequence[int] | None = None, table: np.ndarray | None = None, params: dict | None = None):
        from .skidpad import INFO_DTYPE, SkidpadBatch

        devices = visible_devices() if devices is None else [int(d) for d in devices]
        self.n = int(n_instances)
        self.devices = devices
        self._info_dtype = INFO_DTYPE
        self.ranges = [(lo, hi) for lo, hi in shard_ranges(self.n, len(devices)) if hi > lo]
        self.parts = [SkidpadBatch(hi - lo, device=d, table=table, params=params) for (lo, hi), d in zip(self.ranges, devices)]
        self.tables = self.parts[0].tables
        self._depth = 1
        self._stage, self._turn = None, 0
        self._pool = _capi.PinnedPool()

    @property
    def constants(self):
        return self.parts[0].constants

    def reset(self):
        for p in self.parts:
            p.reset()

    def set_overlap(self, depth: int):
        self._depth = max(1, int(depth))
        for p in self.parts:
            p.set_overlap(depth)
        self._stage = [[_Staging() for _ in range(self._depth + 1)] for _ in self.parts]
        self._turn = 0

    def submit(self, cone_offsets, cones_xyt, poses, out=None, info=None, compact: bool = False):
        """One step of every planner as a ticket.  ``out``: page-locked array of the whole batch, RESULT_DTYPE or — compact —
        PATH_RESULT_DTYPE (default: a block of this object's pool, returned to it when the last reference is dropped)."""
        off, cones, poses, n = _capi.Context._prep(cone_offsets, cones_xyt, poses)
        assert n == self.n
        want = self.parts[0]._ctx.path_result_dtype if compact else self.parts[0]._ctx.result_dtype
        if out is None:
            out = self._pool.get(n, want)
        if out.dtype != want:
            raise ValueError(f"out.dtype is {out.dtype}, compact={compact} asks for {want}")
        if info is None:
            info = np.zeros(n, dtype=self._info_dtype)
        if self._stage is None:
            self.set_overlap(self._depth)
        k = self._turn % len(self._stage[0])
        self._turn += 1
        tickets = []
        for g, ((lo, hi), part) in enumerate(zip(self.ranges, self.parts)):
            o, c, p, _ = self._stage[g][k].load(off[lo : hi + 1], cones[off[lo] : off[hi]], poses[lo:hi], None)
            tickets.append(part.submit(o, c, p, out=out[lo:hi], info=info[lo:hi], compact=compact))
        return MultiTicket(tickets, (out, info))

    def collect(self, ticket: MultiTicket):
        for part, t in zip(self.parts, ticket.parts):
            part.collect(t)
        return ticket.out

    def step(self, cone_offsets, cones_xyt, poses):
        out, info = self.collect(self.submit(cone_offsets, cones_xyt, poses))
        return out, info

    def replay(self, frames, depth: int = 32, compact: bool = False):
        """SkidpadBatch.replay over all GPUs: `depth` steps submitted ahead on every context.  The yielded result arrays are
        the page-locked blocks the GPUs wrote (a pool: a block serves a later step once its array has been dropped)."""
        self.set_overlap(depth)
        inflight = []
        for f in frames:
            if len(inflight) == depth:
                yield self.collect(inflight.pop(0))
            inflight.append(self.submit(*f, compact=compact))
        for t in inflight:
            yield self.collect(t)

    def close(self):
        for p in self.parts:
            p.close()
        self._pool.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self._pool.close()
        except Exception:
            pass
