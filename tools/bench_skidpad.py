#!/usr/bin/env python3
"""BASELINE config 5: skidpad planner instances replaying demo/skidpad.json (golden copy) with rigidly perturbed starts.
Reports frames/s over the whole stateful sequence, relocalization success, kernel time.

One process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher, like bench.py): the planner instances are
sharded across the ranks, rank 0 alone loads the constant tables (known skidpad path 5786 x 2 f64 = 92 576 B, fixed-seed noise
table) and every other rank receives them through the RCCL broadcast of the C ABI (fsdp_comm_broadcast) — the only collective
besides the timing barrier.   python tools/bench_skidpad.py [n_instances_total]"""
import importlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import skidpad_support as sk  # noqa: E402

pkg = importlib.import_module("ft-fsd-path-planning_amd")

pkg._capi.DEFAULT_OPTIONS.update(pkg._capi.options_from_env())
if os.environ.get("AB_LIB"):  # an experiment build (tools/build_variant.sh)
    pkg._capi.LIB_PATH = Path(os.environ["AB_LIB"])  # FSDP_PACK / FSDP_PATH_MODE / ... of this tool's shell -> fsdp_set_option
n_total = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
comm_ctx = pkg.Context(device=local_rank, mission=4)  # carries the communicator
d = pkg.dist.Dist(comm_ctx)
lo, hi = d.frame_range(n_total)  # contiguous shard of the planner instances
n = hi - lo
# the track map travels over RCCL: only rank 0 touches the data file
table = d.broadcast_array(pkg.skidpad.load_tables()[0] if rank == 0 else None, (5786, 2))
noise = d.broadcast_array(pkg.skidpad.load_tables()[1] if rank == 0 else None, (1140, 3, 2))
g = sk.load_sequence(ROOT / "tests" / "golden")
tf = sk.perturbed_instances(g, n_total)[lo:hi]
T = len(g["poses"])
batches = [sk.batch_for_step(g, t, tf) for t in range(T)]
batch = pkg.SkidpadBatch(n, device=local_rank, table=table)
assert np.array_equal(batch.tables[1], noise)
DEPTH = int(os.environ.get("FSDP_SKID_DEPTH", "32"))
batches = [tuple(pkg.pinned_copy(a, dt) for a, dt in zip(b, (np.int32, np.float64, np.float64))) for b in batches]
outs = [pkg.pinned_empty(n, pkg.RESULT_DTYPE) for _ in range(DEPTH + 1)]
batch.set_overlap(DEPTH)
for t in range(3):  # warm-up on a throw-away state (reference demo does the same, json_demo.py:89-94)
    batch.step(*batches[t])
for tk in [batch.submit(*batches[t], out=outs[t]) for t in range(DEPTH)]:  # (and on the group's kernels and workspace)
    batch.collect(tk)
# (a) one step at a time: submit + collect (what a live car does)
ONLY_AHEAD = os.environ.get("FSDP_SKID_BENCH_LEGS") == "ahead"  # (kernel traces of the submitted-ahead leg alone)
batch.reset()
d.barrier()
t0 = time.perf_counter()
status = np.zeros(n, np.int64)
for t in range(0 if ONLY_AHEAD else T):
    res, info = batch.step(*batches[t])
    status += res["status"] != 0
d.barrier()
el_step = d.max_over_ranks(time.perf_counter() - t0)
ref_last = None if ONLY_AHEAD else res["path"].copy()
# (b) the replay as a stream: up to DEPTH steps submitted ahead (fsdp_skidpad_submit): consecutive steps share their launches
# (csrc/skidpad_kernel.h "steps in flight"), the next group's inputs go up and the previous one's results come down while a
# group's kernels run; the planner states chain on the device
def replay_ahead(keep0=None):
    status = np.zeros(n, np.int64)
    inflight = []
    done = 0
    for t in range(T):
        if len(inflight) == DEPTH:
            res, info = batch.collect(inflight.pop(0))
            status += res["status"] != 0
            if keep0 is not None:
                keep0.append(res["path"][0].copy())
        inflight.append(batch.submit(*batches[t], out=outs[t % (DEPTH + 1)]))
    for tk in inflight:
        res, info = batch.collect(tk)
        status += res["status"] != 0
        if keep0 is not None:
            keep0.append(res["path"][0].copy())
    return status, res, info


batch.reset()
d.barrier()
t0 = time.perf_counter()
status, res, info = replay_ahead()
d.barrier()
el = d.max_over_ranks(time.perf_counter() - t0)
assert ONLY_AHEAD or np.array_equal(res["path"], ref_last, equal_nan=True), "pipelined steps differ from one-at-a-time steps"
# (c) the same stream with compact results (fsdp_skidpad_submit_compact: 1296-byte records — a skidpad step has no sorting /
# matching outputs — instead of the 2408-byte fsdp_frame_result)
outs_c = [pkg.pinned_empty(n, pkg.PATH_RESULT_DTYPE) for _ in range(DEPTH + 1)]
def replay_compact():
    inflight = []
    for t in range(T):
        if len(inflight) == DEPTH:
            res_c, _ = batch.collect(inflight.pop(0))
        inflight.append(batch.submit(*batches[t], out=outs_c[t % (DEPTH + 1)]))
    for tk in inflight:
        res_c, _ = batch.collect(tk)
    return res_c
batch.reset()
replay_compact()  # (the copy kernel's first launches)
batch.reset()
d.barrier()
t0 = time.perf_counter()
res_c = replay_compact()
d.barrier()
el_c = d.max_over_ranks(time.perf_counter() - t0)
assert np.array_equal(res_c["path"], res["path"], equal_nan=True), "compact results differ"
# the same replay once more, untimed, with HIP events around every group's packed kernels (dominant kernel -> roofline) and
# planner 0 — the unperturbed recording — held against the reference's golden sequence (flip count)
paths0 = []
extra = {}
if rank == 0:
    batch.reset()
    batch.time_groups(True)
    replay_ahead(paths0)
    kernel_ms, n_groups, n_pairs = batch.group_times()
    batch.time_groups(False)
    err0 = np.array([np.abs(p - g["path"][t]).max() for t, p in enumerate(paths0)])
    extra["flip_count"] = {"frames": int(T), "flips": int((err0 > 1e-5).sum()), "max_err": float(err0.max()),
                           "set": "planner 0 = the recording, against tests/golden/skidpad_sequence.npz (the reference's outputs)"}
    if n_groups:
        dom = max(kernel_ms, key=kernel_ms.get)
        # algorithmic bytes per (planner, step) once relocalized: pose in (32 B) + path out (1280 B) + relocalization information out (40 B)
        algo = 32 + 1280 + 40
        achieved = algo * n_pairs / (kernel_ms[dom] * 1e-3) / 1e9
        extra["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                             "kernel_ms_per_group": {k: v / n_groups for k, v in kernel_ms.items()}, "groups": n_groups, "instance_step_pairs": n_pairs,
                             "note": f"algorithmic bytes per (planner, step) = {algo} x pairs / summed duration of the dominant kernel's launches "
                                     "(HIP events on the context's stream around the packed kernels of every group of steps; untimed repeat of the replay)"}
kms = batch.time_path(10) / 10  # the path kernel repeated on the LAST frame of the replay (the car stands at the end of the track: the shortest path of the run)
reloc = d.sum_over_ranks(float(info["relocalized"].sum()))
bad = d.sum_over_ranks(float(status.sum()))
if rank == 0:
    print(json.dumps({"config": "BASELINE configs[4]: skidpad, batch=%d perturbed starts x %d frames on %d GPU(s)" % (n_total, T, d.world),
                      "frames_per_s_incl_pcie": n_total * T / el, "seconds": el, "steps_in_flight": DEPTH,
                      "frames_per_s_incl_pcie_compact_results": n_total * T / el_c,
                      "frames_per_s_incl_pcie_one_step_at_a_time": None if ONLY_AHEAD else n_total * T / el_step, "relocalized": int(reloc),
                      "frames_with_nonzero_status": int(bad), "ms_per_step": el / T * 1e3, "skid_path_kernel_ms_on_the_last_frame": kms,
                      "note": "submitted ahead, DEPTH / 2 consecutive steps share one group of launches (packed path-stage kernels from 2048 (instance, step) pairs, else a wavefront per pair); one step at a time = the latency of one path stage on a wavefront per planner",
                      **extra,
                      "tables": ("rank 0 loads them, broadcast to the others; communicator " + d.describe()) if d.world > 1 or d._active else "single process"}))
d.close()
