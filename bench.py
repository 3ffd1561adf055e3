#!/usr/bin/env python3
"""bench.py — PathPlanner frames/s at 128 cones/frame on N MI355X (BASELINE.json metric).

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N > 1 launched by
``python -m torch.distributed.run --nproc-per-node N ...`` (one rank per GPU; the launcher only provides RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_*: this process never imports torch — the communicator is RCCL behind the C ABI,
include/fsdp.h fsdp_comm_*).  A *step* is one pass of the whole hot path (sorting -> matching -> path kernels) over
one batch of synthetic frames that is already resident in HBM.

``--config 2`` (default; the configuration the metric is quoted on): BASELINE configs[1] — batch 4096 synthetic replay
frames per GPU, 64 left + 64 right coloured cones (the FSG recording itself is absent from the reference checkout,
.MISSING_LARGE_BLOBS); weak scaling, every rank replays its own track.
``--config 4``: BASELINE configs[3] — ONE global batch of 65 536 frames x 200 cones (Gaussian xy perturbation, sigma
0.1 m) cut into contiguous frame ranges [g*B/G, (g+1)*B/G), one per rank; strong scaling.

Frames are independent, so ranks shard them with no data-path collective; RCCL carries only the start-up broadcast /
check of the constant previous-path table, the timing barrier and the max-reduction of the elapsed time.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import os

# several passes in flight = one HIP stream each; the runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and
# RCCL / the null stream take queues too — must be set before the HIP runtime starts in this process
# (FSDP_SHARE_GPU=1 — several ranks on one GPU, testing only — adds the processes' queues up: keep each small)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "10" if os.environ.get("FSDP_SHARE_GPU") == "1" and "RANK" in os.environ else "22")

import argparse
import importlib
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FRAMES_PER_GPU = 4096        # config 2
CONES_PER_SIDE = 64
CFG4_FRAMES, CFG4_CONES_PER_SIDE = 65536, 100  # config 4 (global batch)


def algo_bytes_per_frame(cones_per_frame: int) -> int:
    """read N*24 + 32 (cones, pose) + write 1280 + 96 + 8: 4488 at N = 128, 6216 at N = 200."""
    return cones_per_frame * 24 + 32 + 1280 + 96 + 8
PASS_OVERLAP = 20  # passes in flight in the timed region (fsdp_set_overlap).  Round 4 (profiles/r04_ab_variants.txt 7): every pass in
# flight has its own HIP stream, and streams beyond the runtime's hardware queues share one and serialise (20 in flight on 16
# queues: 5.0 M); with 20-32 queues twenty passes in flight give 5.7-5.9 M frames/s over 20 passes and 6.3-6.4 M over 100, against
# 5.4 / 6.1-6.2 M with ten; 24 or more passes collapse (0.7-3.6 M).  22 queues, not 32: the queues of all processes on a GPU add up,
# and a second process next to 24+ of them crawls (the skidpad child of this script: 5.3 -> 0.5 M; it gets 8 of its own).
STREAM_DEPTH = 20  # pass slots of the host -> host streaming leg (two tickets queue on each; profiles/r06_streaming_probe.txt)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def usable_cores() -> int:
    """Host threads this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box
    shows 256 logical CPUs but grants a 16-CPU quota; more threads than that only get throttled)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except Exception:
            pass
    return n


def cpu_baseline(off, cones, poses, budget_s: float = 12.0):
    """The CPU oracle (oracle/, parity-checked restatement of the reference; kind 'port') timed on this
    host's cores over a bounded sample of the same workload.  Baseline, not target."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib  # test infrastructure, used here only as the measured CPU baseline

    cores = usable_cores()
    n = len(off) - 1
    oracle_lib.plan_batch(off[:65], cones[: off[64]], poses[:64], n_threads=cores)  # warm (default path, page-in)
    # single-thread latency sample
    t0 = time.perf_counter()
    k1 = min(64, n)
    oracle_lib.plan_batch(off[: k1 + 1], cones[: off[k1]], poses[:k1], n_threads=1)
    t_single = (time.perf_counter() - t0) / k1
    done, t0 = 0, time.perf_counter()
    while True:
        oracle_lib.plan_batch(off, cones, poses, n_threads=cores)
        done += n
        el = time.perf_counter() - t0
        if el > budget_s or done >= 64 * n:
            break
    return {
        "value": done / el,
        "unit": "frames/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{done} frames ({done // n} passes over the {n}-frame batch), std::thread over frames, threads = "
                  f"usable cores (affinity {os.cpu_count()} capped by the cgroup CPU quota); "
                  f"1-thread latency {t_single * 1e6:.0f} us/frame",
        "single_thread_us_per_frame": t_single * 1e6,
    }


def skidpad_leg(n_instances: int = 1024, timeout: float = 120.0):
    """BASELINE configs[4] (tools/bench_skidpad.py): frames/s of n_instances stateful skidpad planners over the 341-frame
    recording, the replay submitted ahead and one step at a time; None-valued fields with the reason if it could not run."""
    import subprocess

    try:
        r = subprocess.run([sys.executable, str(ROOT / "tools" / "bench_skidpad.py"), str(n_instances)], capture_output=True, text=True,
                           timeout=timeout, env={**{k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")},
                                                 "GPU_MAX_HW_QUEUES": "8"})  # (one stream; this process still holds its 22 queues)
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        return {"value": j["frames_per_s_incl_pcie"], "unit": "frames/s", "planner_instances": n_instances, "frames": 341,
                "one_step_at_a_time_frames_per_s": j["frames_per_s_incl_pcie_one_step_at_a_time"], "steps_in_flight": j["steps_in_flight"],
                "relocalized": j["relocalized"], "frames_with_nonzero_status": j["frames_with_nonzero_status"],
                "flip_count": j.get("flip_count"), "roofline": j.get("roofline"),
                "compact_results_frames_per_s": j.get("frames_per_s_incl_pcie_compact_results"),
                "what": "host buffers to host buffers, H2D + kernels + D2H of every step in the clock; consecutive steps of a planner share their launches (DESIGN.md, Skidpad)"}
    except Exception as e:  # noqa: BLE001  (an extra of the line, never its failure)
        return {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}


def streaming_leg(pkg, ctx, per_gpu: int, depth: int, n_batches: int, seed0: int):
    """Host -> host: `n_batches` DIFFERENT batches (a different synthetic track each) through `depth` pass slots with
    fsdp_submit / fsdp_collect — every batch is copied to the GPU, planned and its results copied back inside the
    timed region, all buffers page-locked (fsdp_host_alloc).  This is the rate a caller of the public API gets for a
    stream of batches (the reference's harness feeds frames one after the other, demo/json_demo.py:103-131).  `value` is the
    stream with COMPACT result records (fsdp_submit_compact: path + sorted indices + status, the 1384 bytes SURVEY 8d counts as a
    frame's output); `full_records_frames_per_s` the same stream with the 2408-byte records that also carry the matching
    intermediates.  One batch at a time = blocking fsdp_plan_batch calls on the same buffers (a batch is pipelined in chunks
    inside the call)."""
    batches = []
    for k in range(n_batches):
        off, cones, poses = pkg.synth.make_replay_batch(per_gpu, CONES_PER_SIDE, 0.15, seed=seed0 + 1000 + k, color=True)
        batches.append((pkg.pinned_copy(off, np.int32), pkg.pinned_copy(cones, np.float64), pkg.pinned_copy(poses, np.float64)))
    outs = [pkg.pinned_empty(per_gpu, pkg.RESULT_DTYPE) for _ in range(n_batches)]
    outs_c = [pkg.pinned_empty(per_gpu, pkg.COMPACT_DTYPE) for _ in range(n_batches)]
    ctx.set_overlap(depth)

    def replay(compact):
        inflight = []
        for k in range(n_batches):
            if len(inflight) == ctx.ticket_capacity:
                ctx.collect(inflight.pop(0))
            inflight.append(ctx.submit(*batches[k], out=(outs_c if compact else outs)[k], compact=compact))
        for t in inflight:
            ctx.collect(t)

    # warm-up: one untimed replay of the whole stream — every slot's stream and buffers exist, and the context has seen
    # which route kernels this stream's batches need (a pass that lacks one is repeated: that belongs to a stream's first
    # seconds, not to its rate)
    replay(False)
    replay(True)
    for o in outs + outs_c:
        o["status"] = -1
    reruns0 = ctx.route_stats()[2]
    t0 = time.perf_counter()
    replay(False)
    el_full = time.perf_counter() - t0
    t0 = time.perf_counter()
    replay(True)
    el = time.perf_counter() - t0
    # one batch at a time: blocking calls, page-locked buffers in place
    n_ser = min(8, n_batches)
    ctx.plan_batch(*batches[0], out=outs[0])
    t1 = time.perf_counter()
    for k in range(n_ser):
        ctx.plan_batch(*batches[k], out=outs[k])
    el_ser = time.perf_counter() - t1
    ctx.plan_batch(*batches[0], out=outs_c[0], compact=True)
    t1 = time.perf_counter()
    for k in range(n_ser):
        ctx.plan_batch(*batches[k], out=outs_c[k], compact=True)
    el_ser_c = time.perf_counter() - t1
    pageable = [tuple(np.array(a) for a in b) for b in batches[:n_ser]]
    ctx.plan_batch(*pageable[0])
    t1 = time.perf_counter()
    for b in pageable:
        ctx.plan_batch(*b)
    el_ser_pg = time.perf_counter() - t1
    # the streamed results are the serial results; the compact records are the full records' fields
    ctx.set_option("plan_chunks", 1)
    chk = ctx.plan_batch(*pageable[n_ser - 1])
    ctx.set_option("plan_chunks", 0)
    same = chk.tobytes() == outs[n_ser - 1].tobytes()
    same_c = all(np.array_equal(outs_c[n_ser - 1][f], chk[f], equal_nan=True) for f in ("path", "left_idx", "right_idx", "status"))
    bad = int(sum(int((o["status"] != 0).sum()) for o in outs_c))
    h2d = sum(a.nbytes for a in batches[0])
    ceiling = ctx.pcie_probe(h2d, 30)
    # The yardstick of the stream: the SAME batches resident in HBM at the same depth (every track costs the GPU differently: 5.5-6.8 M
    # frames/s; the headline's track is one of the cheaper ones) — a sample of six of them, sixty passes each, no PCIe in the clock.
    res_rates = []
    ctx.set_overlap(depth)
    for k in range(0, n_batches, max(1, n_batches // 6))[:6]:
        ctx.upload(*batches[k])
        ctx.time_runs(depth, collect=False)
        ctx.sync()
        t1 = time.perf_counter()
        ctx.time_runs(60, collect=False)
        ctx.sync()
        res_rates.append(per_gpu * 60 / (time.perf_counter() - t1))
    resident_same = float(np.mean(res_rates)) if res_rates else None
    return {
        "value": per_gpu * n_batches / el, "unit": "frames/s", "batches": n_batches, "frames_per_batch": per_gpu, "depth": depth,
        "seconds": el, "result_records": "compact (fsdp_compact_result, 1384 B)",
        "full_records_frames_per_s": per_gpu * n_batches / el_full,
        "one_batch_at_a_time_frames_per_s": per_gpu * n_ser / el_ser_c,
        "one_batch_at_a_time_full_records_frames_per_s": per_gpu * n_ser / el_ser,
        "one_batch_at_a_time_pageable_frames_per_s": per_gpu * n_ser / el_ser_pg,
        "pcie_bytes_per_batch": {"h2d": int(h2d), "d2h": int(outs_c[0].nbytes), "d2h_full_records": int(outs[0].nbytes)},
        "pcie_GBps": {"h2d": h2d * n_batches / el / 1e9, "d2h": outs_c[0].nbytes * n_batches / el / 1e9},
        # what the box's link carries for page-locked copies of a batch's size (fsdp_pcie_probe: hipMemcpyAsync, 30 x 12.7 MB)
        "pcie_ceiling_GBps": ceiling,
        "resident_rate_of_the_same_batches": resident_same,
        "stream_over_resident_same_batches": (per_gpu * n_batches / el / resident_same) if resident_same else None,
        "last_batch_equals_serial_plan_batch": bool(same), "compact_records_equal_full_records_fields": bool(same_c),
        "frames_with_nonzero_status": bad,
        "passes_rerun_for_routes": ctx.route_stats()[2] - reruns0,
        "what": "different batches host -> host (page-locked buffers), H2D + kernels + D2H of every batch inside the timed region; "
                "warm-up = one untimed replay of the same stream",
    }



def multi_planner_leg(pkg, devices, per_ctx: int, depth: int, n_batches: int, seed0: int, reference_rate=None):
    """The user-facing multi-GPU class (multi.MultiPlanner: every GPU from one process and one thread) fed the way a caller feeds
    it: `n_batches` DIFFERENT page-locked batches of `per_ctx` frames per context through plan_stream, host -> host.  Reports
    the rate, the time the calling thread spent inside submit() per frame (what bounds the class when the GPUs are many), and
    whether the batches went out zero-copy (slices of the caller's arrays) or through staging copies."""
    n_ctx = len(devices)
    try:
        mp = pkg.MultiPlanner(devices, overlap=depth)
        batches = []
        for k in range(n_batches):
            off, cones, poses = pkg.synth.make_replay_batch(per_ctx * n_ctx, CONES_PER_SIDE, 0.15, seed=seed0 + 2000 + k, color=True)
            batches.append((pkg.pinned_copy(off, np.int32), pkg.pinned_copy(cones, np.float64), pkg.pinned_copy(poses, np.float64)))

        def replay():
            bad = 0
            last = None
            for res in mp.plan_stream(batches, compact=True):
                bad += int((res["status"] != 0).sum())
                last = res
            return bad, last

        replay()  # warm-up (streams, buffers, the contexts' route predictions)
        mp.reset_host_time()
        zc0 = mp.zero_copy_batches
        t0 = time.perf_counter()
        bad, last = replay()
        el = time.perf_counter() - t0
        host_s, host_frames = mp.host_seconds, mp.host_frames
        chk = mp.ctx[0].plan_batch(*batches[-1], compact=True)
        same = all(np.array_equal(chk[f], last[f], equal_nan=True) for f in chk.dtype.names)
        # the same stream from pageable arrays (staged by one worker thread per context)
        pageable = [tuple(np.array(a) for a in b) for b in batches[: max(2, n_batches // 4)]]
        list(mp.plan_stream(pageable))
        mp.reset_host_time()
        t1 = time.perf_counter()
        for _ in mp.plan_stream(pageable):
            pass
        el_pg = time.perf_counter() - t1
        rate = per_ctx * n_ctx * n_batches / el
        out = {"value": rate, "unit": "frames/s", "contexts": n_ctx, "devices": list(devices), "frames_per_batch": per_ctx * n_ctx,
               "batches": n_batches, "depth_per_context": depth, "seconds": el,
               "host_us_per_submitted_frame": host_s / max(host_frames, 1) * 1e6,
               "host_thread_busy_fraction": host_s / el,
               "zero_copy_batches": mp.zero_copy_batches - zc0, "result_copies": 0, "result_records": "compact (fsdp_compact_result)",
               "pageable_input_frames_per_s": per_ctx * n_ctx * len(pageable) / el_pg,
               "pageable_host_us_per_submitted_frame": mp.host_seconds / max(mp.host_frames, 1) * 1e6,
               "last_batch_equals_one_context_plan_batch": bool(same), "frames_with_nonzero_status": bad,
               "what": "multi.MultiPlanner.plan_stream, page-locked batches sharded as slices of the caller's arrays (fsdp_submit with "
                       "cone_offsets[0] != 0), results returned as the page-locked blocks the GPUs wrote; host -> host, everything in the clock"}
        if reference_rate:
            out["vs_single_context_streaming"] = rate / reference_rate
        mp.close()
        return out
    except Exception as e:  # noqa: BLE001  (an extra of the line, never its failure)
        return {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}


def chip_time_leg(frames: int = 98304, timeout: float = 240.0):
    """What a kernel costs when it has the chip to itself and fills it: one pass at a time over a resident batch of 98 304 frames
    (24 x the bench batch, same generator), HIP events around every launch -> ns of chip time per frame and kernel.  Unlike the
    duration of an overlapped launch (which depends on how twenty streams interleave on the box at hand) this figure is a property of
    the kernel: the sum over the kernels is the floor of ms_per_step / frames, and profiles/ reproduces it — it IS
    tools/batch_sweep.py, run as a child process with the packed kernels pinned (inside this process, after the bench's other
    legs have allocated and freed tens of GB, the same launches measured 30 % slower for the scratch-heavy kernels)."""
    import subprocess

    try:
        env = dict(os.environ, FSDP_PACK="1")
        r = subprocess.run([sys.executable, str(ROOT / "tools" / "batch_sweep.py"), str(frames)], capture_output=True, text=True, timeout=timeout, env=env)
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        per = {k: v / j["frames"] * 1e6 for k, v in j["kernel_ms"].items()}
        return {"frames_per_launch": j["frames"], "chip_ns_per_frame": per, "chip_ns_per_frame_sum": float(sum(per.values())),
                "frames_per_s_at_that_sum": 1e9 / float(sum(per.values())), "chip_time_source": "tools/batch_sweep.py 98304 (FSDP_PACK=1), child process"}
    except Exception as e:  # noqa: BLE001
        return {"chip_time_error": f"{type(e).__name__}: {e}"[:300]}


class InProcess:
    """--single-process: every GPU of the node driven from THIS process (multi.py's form of the product: one context per
    device, no launcher, no socket, no RCCL).  Context 0 plays rank 0 (all the extras of the line are measured on it); the
    timed region runs `steps` passes on every context at once, one host thread per context around the blocking
    fsdp_time_runs call (ctypes releases the GIL), and the clock spans the slowest of them.  Interface of dist.Dist."""

    transport = "in-process"

    def __init__(self, pkg, n: int, n_dev: int, share: bool):
        if n_dev < n and not share:
            sys.exit(f"bench.py: --gpus {n} but only {n_dev} GPU(s) visible (FSDP_SHARE_GPU=1 puts several contexts on one GPU: testing only)")
        self.world, self.rank = n, 0
        self.devices = [g % max(n_dev, 1) for g in range(n)]
        self.ctxs = [pkg.Context(device=dv, mission=int(pkg.MissionTypes.trackdrive)) for dv in self.devices]

    def describe(self):
        return f"in-process, {self.world} context(s) on device(s) {self.devices} driven by one process (multi.py), no collective at all"

    def shard_seed(self, base_seed, g=0):
        return base_seed + g

    def frame_range(self, n_total, g=0):
        return g * n_total // self.world, (g + 1) * n_total // self.world

    def broadcast_check_table(self, table):
        ref = np.ascontiguousarray(table).view(np.uint64)
        return all(np.array_equal(np.ascontiguousarray(c.default_path()).view(np.uint64), ref) for c in self.ctxs[1:])

    def barrier(self):
        for c in self.ctxs:
            c.sync()

    def max_over_ranks(self, v):
        return v

    def timed_region(self, steps):
        """EXACTLY `steps` passes on every context, started together; returns the wall clock over all of them."""
        import threading

        gate = threading.Barrier(len(self.ctxs) + 1)
        errs = []

        def work(c):
            gate.wait()
            try:
                c.time_runs(steps, collect=False)
                c.sync()
            except Exception as e:  # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=work, args=(c,)) for c in self.ctxs]
        for t in th:
            t.start()
        self.barrier()
        gate.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        if errs:
            raise errs[0]
        return el

    def close(self):
        for c in self.ctxs[1:]:
            c.close()


def _lib_hash(pkg) -> str:
    import hashlib

    return hashlib.sha256(Path(pkg._capi.LIB_PATH).read_bytes()).hexdigest()[:16]


def _pmc(pkg):
    """The committed rocprofv3 PMC passes (profiles/pmc_traffic.json, produced by tools/profile_gpu.sh), but only when
    they were collected from the very library this run loaded (recorded sha256): a stale file yields None."""
    try:
        d = json.load(open(ROOT / "profiles" / "pmc_traffic.json"))
        return d if d.get("lib_sha256_16") == _lib_hash(pkg) else None
    except Exception:
        return None


def _all_kernel_insts(pmc, names):
    """VALU wave-instructions per frame summed over the kernels of a pass, or None when a kernel has no counters."""
    if not pmc:
        return None
    tot = 0.0
    for n in names:
        v = (pmc.get(n) or {}).get("valu_insts_per_frame")
        if v is None:
            if n in ("path_retry_kernel",):
                continue
            return None
        tot += v
    return tot


def golden_flip_count(pkg, ctx):
    """Sample-count flips (120 <-> 121 dense samples, DESIGN.md "arithmetic contract") against the REFERENCE on the
    committed golden fuzz set (tests/golden/fuzz.npz: reference outputs captured by tests/golden/make_golden.py):
    frames whose path differs from the reference's by more than 1e-5.  They can only be arc-extension frames."""
    try:
        g = np.load(ROOT / "tests" / "golden" / "fuzz.npz")
        res = ctx.plan_batch(g["offsets"], g["cones"], g["poses"])
        ok = g["ok"].astype(bool) & (res["status"] == 0)
        err = np.zeros(len(res))
        for k in np.nonzero(ok)[0]:
            e = np.abs(res["path"][k] - g["path"][k])
            err[k] = np.nanmax(e) if not np.isnan(e).all() else 0.0
        arc = (res["path_fallback"] & 16) != 0
        flips = ok & (err > 1e-5)
        return {"frames": int(ok.sum()), "arc_frames": int((ok & arc).sum()), "flips": int(flips.sum()),
                "flips_outside_arc_frames": int((flips & ~arc).sum()),
                "max_err_other_frames": float(err[ok & ~flips].max()) if (ok & ~flips).any() else 0.0,
                "set": "tests/golden/fuzz.npz (synthetic fuzz frames with the reference's outputs)"}
    except Exception as e:  # the bench line must not die on a fixture problem
        return {"error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=[2, 4],
                    help="2: 4096 frames x 128 cones per GPU (weak; the metric's configuration); 4: 65536 x 200 cones in contiguous shards (strong)")
    ap.add_argument("--frames", type=int, default=0, help="override the frame count (per GPU for config 2, global for config 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--latency", action="store_true", help="(always measured at N = 1; kept for compatibility)")
    ap.add_argument("--no-skidpad", action="store_true", help="skip the config-5 extra of the line (tools/bench_skidpad.py, ~10 s)")
    ap.add_argument("--no-latency", action="store_true", help="skip the single-frame latency loop and the golden flip count (profiling runs)")
    ap.add_argument("--no-overlap", action="store_true", help="one pass strictly after the other (single stream)")
    ap.add_argument("--overlap", type=int, default=PASS_OVERLAP, help="passes in flight (1..32)")
    ap.add_argument("--seed", type=int, default=1, help="seed of the synthetic track (config 2; rank r replays seed + r)")
    ap.add_argument("--single-process", action="store_true",
                    help="N > 1 without a launcher, sockets or RCCL: one context per GPU, all driven from this process (multi.py); "
                         "also what --gpus N falls back to when its own launch of N ranks fails")
    ap.add_argument("--stream-batches", type=int, default=100, help="different batches of the host -> host streaming leg (0: skip)")
    ap.add_argument("--allow-tcp-fallback", action="store_true",
                    help="N > 1 ranks: accept a run whose start-up collectives went over the TCP star because RCCL did not come up on "
                         "every rank (the line then says communicator: tcp-fallback); without it such a run exits with status 3")
    args = ap.parse_args()
    if os.environ.get("FSDP_HANG_DUMP"):  # diagnostics: Python stacks of all threads after N seconds, then exit
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["FSDP_HANG_DUMP"]), exit=True)

    pkg = importlib.import_module("ft-fsd-path-planning_amd")
    n_dev = pkg._capi.load().fsdp_device_count()
    share = os.environ.get("FSDP_SHARE_GPU") == "1"  # testing only: several ranks on one GPU (RCCL refuses that -> tcp-fallback)
    single = args.single_process and args.gpus > 1
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not single:
        # started without a launcher: be the launcher — one rank per GPU with the launch contract's environment
        if n_dev < args.gpus and not share:
            sys.exit(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) visible")
        rc, text = pkg.dist.spawn_ranks([str(Path(__file__).resolve()), *sys.argv[1:]], args.gpus, capture=True,
                                        timeout=float(os.environ.get("FSDP_SPAWN_TIMEOUT", "900")))
        lines = [l for l in text.splitlines() if l.startswith("{") and '"metric"' in l]
        if lines:
            print(lines[-1], flush=True)
            sys.exit(3 if rc == 3 else 0)  # (3: the ranks ran, but not over RCCL — see --allow-tcp-fallback)
        print(f"bench.py: the {args.gpus} ranks produced no line (exit status {rc}); planning on the {args.gpus} GPUs from this process instead", file=sys.stderr)
        single = True
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if single:
        d = InProcess(pkg, args.gpus, n_dev, share)
        ctx = d.ctxs[0]
    else:
        ctx = pkg.Context(device=local_rank % max(n_dev, 1) if share else local_rank, mission=int(pkg.MissionTypes.trackdrive))
        # WORLD_SIZE > 1: ncclCommInitRank on this rank's GPU (RCCL over xGMI; its id travels over the TCP star of dist.py, which
        # also carries the collectives if RCCL fails on any rank)
        d = pkg.dist.Dist(ctx)
    world = d.world
    assert "torch" not in sys.modules, "the product path must not pull in PyTorch"
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}: reporting n_gpus = {world}", file=sys.stderr)

    # the only collectives on this path: rank 0 broadcasts the constant previous-path table (RCCL over xGMI), every
    # rank checks it against the table its own GPU computed at context creation; then barriers / one max-reduction
    assert d.broadcast_check_table(ctx.default_path()), "previous-path table differs across ranks"
    if not single:
        # RCCL has done what north_star names it for; the timing barrier and the max-reduction are host-side scalars (TCP star)
        d.release_device_communicator()

    if args.config == 2:
        # weak scaling: this rank's own 4096-frame replay (a different track per rank)
        per_gpu = args.frames or FRAMES_PER_GPU
        cones_per_frame = 2 * CONES_PER_SIDE
        off, cones, poses = pkg.synth.make_replay_batch(per_gpu, CONES_PER_SIDE, 0.15, seed=d.shard_seed(args.seed), color=True)
        frames_global = per_gpu * world
        workload = f"BASELINE configs[1]: batch={per_gpu} synthetic autocross replay frames per GPU, 64 L + 64 R coloured cones"
        scaling = "weak"
    else:
        # strong scaling: contiguous shard [lo, hi) of ONE global batch; frame f depends on (seed, f) only
        frames_global = args.frames or CFG4_FRAMES
        cones_per_frame = 2 * CFG4_CONES_PER_SIDE
        lo, hi = d.frame_range(frames_global)
        off, cones, poses = pkg.synth.make_config4_shard(lo, hi, CFG4_CONES_PER_SIDE, 0.1, seed=7)
        workload = (f"BASELINE configs[3]: {frames_global} synthetic frames x 200 cones, Gaussian xy perturbation sigma 0.1 m, "
                    f"contiguous shards of {hi - lo} frames per GPU")
        scaling = "strong"
    n_local = len(off) - 1
    peer_batches = []
    if single:  # the other contexts' shards: what ranks 1 .. N-1 would build
        for g in range(1, world):
            if args.config == 2:
                peer_batches.append(pkg.synth.make_replay_batch(per_gpu, CONES_PER_SIDE, 0.15, seed=d.shard_seed(args.seed, g), color=True))
            else:
                peer_batches.append(pkg.synth.make_config4_shard(*d.frame_range(frames_global, g), CFG4_CONES_PER_SIDE, 0.1, seed=7))
    algo_bytes = algo_bytes_per_frame(cones_per_frame)
    # a replay is a stream of batches: consecutive passes rotate through `overlap` HIP streams / buffer sets so that the
    # next passes fill the compute units the slowest frames of the previous ones no longer occupy (fsdp_set_overlap)
    overlap = 1 if args.no_overlap else args.overlap
    if share and world > 1 and not single:
        overlap = min(overlap, 8)  # (several processes on one GPU: their streams share its hardware queues)
    if n_local * overlap > 131072:  # every pass in flight keeps its own intermediates (~0.1 MB per frame): bound them to ~13 GB
        overlap = max(1, 131072 // n_local)
    ctx.set_overlap(overlap)
    ctx.upload(off, cones, poses)
    for c, b in zip(d.ctxs[1:] if single else [], peer_batches):
        c.set_overlap(overlap)
        c.upload(*b)
        for _ in range(args.warmup):
            c.run()
        c.time_reserve(args.steps)
        c.time_detail(False)

    for _ in range(args.warmup):
        ctx.run()
    ctx.time_reserve(args.steps)  # the HIP events of the timed region exist before it starts
    # In the timed region HIP events bracket the launches of the path stage's main kernel (fit_kernel: the dominant kernel,
    # the one the roofline object is about) of every pass, on the stream the kernel runs on.  Events around all seven
    # launches of a pass cost about 3 % of the throughput (an event record is a packet in the stream's queue): the other
    # kernels' overlapped durations come from a second, untimed run of the same `steps` passes right after the clock stops.
    ctx.time_detail(False)
    ctx.sync()
    d.barrier()
    if single:
        elapsed = d.timed_region(args.steps)
    else:
        t0 = time.perf_counter()
        # EXACTLY `steps` passes, enqueued back to back; returns after the last pass has finished.  The events are read after
        # the clock has stopped.
        ctx.time_runs(args.steps, collect=False)
        ctx.sync()
        mine = time.perf_counter() - t0  # this rank's K passes, from the common start to its own last kernel
        d.barrier()
        elapsed = d.max_over_ranks(mine)  # the job is done when the slowest rank is
    ev_total_ms, ev_main_ms = ctx.time_results()
    names = ctx.stage_names()
    main_ms = [x / args.steps for x in ev_main_ms]  # non-zero for the bracketed kernel only
    ctx.time_detail(True, kernel_clock=True)
    _, ev_stage_ms = ctx.time_runs(args.steps)      # untimed: every kernel bracketed, same passes in flight
    stage_ms = [x / args.steps for x in ev_stage_ms]
    # the refit kernel by its own clock (first wavefront's start to last wavefront's end of every launch) — taken in this
    # untimed repeat: the readings are two atomics per workgroup, and the timed region runs the production launches without them
    kclock_ms, kclock_n = ctx.time_kernel_clock()
    ctx.time_detail(True)

    # for reference: the same kernels one pass after the other (no overlap) — per-launch durations without chip sharing
    ctx.set_overlap(1)
    n_ser = max(3, min(args.steps, 10))
    ser_total_ms, ser_stage_ms = ctx.time_runs(n_ser)
    names_serial = ctx.stage_names()
    serial_ms = [x / n_ser for x in ser_stage_ms]
    res = ctx.download()
    status_hist = {int(k): int(v) for k, v in zip(*np.unique(res["status"], return_counts=True))}
    arc_frames = int(((res["path_fallback"] & 16) != 0).sum())

    if rank == 0:
        value = frames_global * args.steps / elapsed
        dom = int(np.argmax(stage_ms))
        # the kernel bracketed in the timed region (fit_kernel) is the dominant one on the bench workload; where another
        # kernel takes longer per launch (config 4: the sorting kernel over 200 cones), its duration comes from the repeat
        dom_ms = main_ms[dom] if main_ms[dom] > 0 else stage_ms[dom]
        achieved = algo_bytes * n_local / (dom_ms * 1e-3) / 1e9
        pmc = _pmc(pkg)
        pk = (pmc or {}).get(names[dom], {})
        out = {
            "metric": "PathPlanner frames/s at 128 cones/frame" if args.config == 2 else "PathPlanner frames/s at 200 cones/frame (config 4)",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": workload,
                "frames_per_gpu": n_local,
                "frames_global": frames_global,
                "cones_per_frame": cones_per_frame,
                "parallelism": f"frames sharded over {world} GPU(s), no data-path collective; communicator: "
                               + d.describe(),
                "processes": 1 if single else world,
                "communicator": d.transport,
                # what ncclCommCount returned while the RCCL communicator carried the start-up broadcast (it is released before the
                # timed region); 0 = RCCL never ran: for N > 1 ranks this must equal n_gpus, or the process exits with status 3
                "rccl_ranks": int(getattr(d, "rccl_ranks", 0)),
                "pass_overlap": overlap,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": names[dom],
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": pk.get("hbm_bytes_per_launch", 0) / 1e9 if pk.get("hbm_bytes_per_launch") else None,
                "basis": "average duration of the dominant kernel's launches inside the timed region (twenty passes in flight share the chip)",
                "kernel_ms_timed_region": main_ms[dom] if main_ms[dom] > 0 else None,
                "kernel_bracketed_in_timed_region": names[int(np.argmax(main_ms))],
                # the same launches by the kernel's own clock readings (s_memrealtime: first wavefront's start to last
                # wavefront's end = what a kernel trace calls the duration).  The event bracket above starts when the previous
                # kernel of the stream ends, so it also holds the time the launch waits in its hardware queue — with twenty
                # streams on the command processor that wait is of the order of the kernel itself.
                "kernel_ms_device_clock": (kclock_ms / kclock_n) if kclock_n else None,
                "achieved_by_device_clock": (algo_bytes * n_local / (kclock_ms / kclock_n * 1e-3) / 1e9) if kclock_n else None,
                "frac_by_device_clock": (algo_bytes * n_local / (kclock_ms / kclock_n * 1e-3) / 1e9 / HBM_PEAK_GBS) if kclock_n else None,
                "kernel_ms": {n: m for n, m in zip(names, stage_ms)},
                "kernel_ms_serial": {n: m for n, m in zip(names_serial, serial_ms)},
                "ms_per_step_serial": ser_total_ms / n_ser,
                "traffic_unit": "GB per launch (rocprofv3 PMC passes of this very library build, profiles/pmc_traffic.json; null when the recorded library hash differs)",
                "valu_insts_per_frame": pk.get("valu_insts_per_frame"),
                # FP64 / VALU issue utilisation of the whole chip over the timed region: wave-instructions of every kernel of
                # a pass (SQ_INSTS_VALU of the committed PMC passes, same library build) x 4 issue cycles / (SIMDs x clock x time)
                "valu_insts_per_frame_all_kernels": _all_kernel_insts(pmc, names),
                "valu_issue_util": (_all_kernel_insts(pmc, names) * n_local * 4 / (elapsed / args.steps * 2.4e9 * 1024)
                                    if _all_kernel_insts(pmc, names) else None),
                # the same with the issue cost measured on this chip when every SIMD holds four wavefronts of independent
                # FP64 chains (tools/ubench/fp64_rate.hip, profiles/r02_fp64_issue_rate.txt: v_fma_f64 4.6, v_mul / v_add_f64
                # 5.0 cycles per wave-instruction at the reported 2.4 GHz, i.e. 68 TFLOP/s FMA): an upper estimate — the 32-bit
                # part of the mix (selects, DPP moves, integer work) issues faster
                "valu_issue_util_at_measured_fp64_cost": (_all_kernel_insts(pmc, names) * n_local * 4.8 / (elapsed / args.steps * 2.4e9 * 1024)
                                                          if _all_kernel_insts(pmc, names) else None),
                "note": f"algorithmic bytes/frame = {algo_bytes} (SURVEY 8d) x {n_local} frames / average duration of the dominant "
                        "kernel's launches in the timed region (kernel_ms_timed_region: HIP events on the streams the kernel "
                        "runs on; passes overlap, so a launch shares the chip with the other streams' kernels — kernel_ms = "
                        "every kernel bracketed, same passes repeated after the clock stopped; kernel_ms_serial = the same "
                        "launches alone); the path is FP64-issue/latency bound (serial spline QR), not HBM bound",
            },
            "status_histogram": status_hist,
            "arc_extension_frames": arc_frames,
            "lib_sha256_16": _lib_hash(pkg),
        }
        if args.config == 2 and args.stream_batches > 0:
            out["streaming"] = streaming_leg(pkg, ctx, n_local, min(overlap, STREAM_DEPTH), args.stream_batches, d.shard_seed(args.seed))
        if args.config == 2 and args.stream_batches > 0 and (world == 1 or single):
            # multi.MultiPlanner from this one process: on every GPU of a --single-process run, else four contexts on the one GPU
            if not single:
                ctx.set_overlap(1)  # (gives the twenty slots' streams back: hardware queues are shared by every context on the GPU)
            devs = d.devices if single else [ctx.device or 0] * 4
            nb = max(4, args.stream_batches // (1 if single else len(devs)))
            out["multi_planner_stream"] = multi_planner_leg(pkg, devs, n_local, 3, nb, d.shard_seed(args.seed),
                                                            (out.get("streaming") or {}).get("value"))
        if world == 1 and not args.no_latency:
            # sample-count flips against the reference, measured on the committed golden fuzz set
            out["flip_count"] = golden_flip_count(pkg, ctx)
            # BASELINE metric, second half: p50 single-frame latency (batch = 1, host buffers, PCIe-inclusive)
            o1, c1, p1 = off[:2], cones[: off[1]], poses[:1]
            lat = []
            for _ in range(300):
                t1 = time.perf_counter()
                ctx.plan_batch(o1, c1, p1)
                lat.append(time.perf_counter() - t1)
            out["p50_single_frame_us"] = float(np.median(lat[50:]) * 1e6)
            # the reproducible per-kernel figure next to roofline.kernel_ms (round-4 review, weak #11)
            out["roofline"].update(chip_time_leg())
            chip = (out["roofline"].get("chip_ns_per_frame") or {}).get(names[dom])
            if chip:
                # The line's achieved / frac: the dominant kernel ALONE on a chip it fills (one launch of 98 304 frames, HIP events
                # around it on its stream, measured by a child of this run) — a property of the kernel, reproducible from profiles/
                # (r06_batch_sweep_packed_kernels.jsonl, r06_chip_time_rocprofv3_summary.txt).  The duration of one of twenty
                # overlapped launches (..._timed_region below) is a property of how twenty streams interleave on the box at hand.
                rf = out["roofline"]
                rf["achieved_timed_region"], rf["frac_timed_region"] = rf["achieved"], rf["frac"]
                rf["achieved"] = algo_bytes / chip  # bytes per ns = GB/s
                rf["frac"] = rf["achieved"] / HBM_PEAK_GBS
                rf["basis"] = (f"{algo_bytes} algorithmic bytes per frame / {chip:.1f} ns of chip time per frame of {names[dom]} "
                               "(chip_ns_per_frame: the kernel alone, 98 304 frames per launch)")
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(off, cones, poses)
                out["p50_single_frame_vs_cpu_1thread"] = out["p50_single_frame_us"] / out["cpu_baseline"]["single_thread_us_per_frame"]
            # BASELINE configs[4] next to the headline: 1024 stateful skidpad planners over the recording, host to host (a
            # process of its own: tools/bench_skidpad.py; its own context on the same GPU, after everything timed above)
            if not args.no_skidpad:
                out["skidpad_config5"] = skidpad_leg()
        print(json.dumps(out), flush=True)
    # The line is out.  Tear-down (last barrier, ncclCommDestroy, hipHostFree of ~1 GB of page-locked buffers, the HIP
    # runtime's own exit handlers) has nothing left to report, and a benchmark process that hangs on its way out after a
    # successful run would cost the caller its whole time limit (seen once on the GPU box, not reproducible).  The process
    # still leaves the ordinary way (profilers write their output in exit handlers), but under two watchdogs: a timer thread
    # that ends it with status 0 after 30 s, and — for the part of the exit the interpreter's threads do not live to see —
    # SIGALRM's default action 30 s later.
    import signal
    import threading

    exit_status = pkg.dist.multi_rank_exit_status(world, single, d.transport, int(getattr(d, "rccl_ranks", 0)), args.allow_tcp_fallback)
    if exit_status and rank == 0:
        print(f"bench.py: {world} ranks, but communicator = {d.transport!r} with {getattr(d, 'rccl_ranks', 0)} RCCL rank(s) "
              f"({getattr(d, 'fallback_reason', None)}): exit status {exit_status} (--allow-tcp-fallback accepts such a run)", file=sys.stderr)
    sys.stdout.flush()
    sys.stderr.flush()
    # (ranks other than 0 reach this point while rank 0 still measures its single-GPU extras — streaming, latency, CPU
    # baseline, ~20 s — and then wait for it in the barrier: their fuse is longer)
    fuse = 30.0 if rank == 0 else 240.0
    killer = threading.Timer(fuse, lambda: os._exit(exit_status))
    killer.daemon = True
    killer.start()
    signal.signal(signal.SIGALRM, signal.SIG_DFL)
    signal.alarm(int(fuse) + 30)
    d.barrier()
    d.close()
    ctx.close()
    sys.exit(exit_status)


if __name__ == "__main__":
    main()
