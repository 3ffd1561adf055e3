"""JSON replay loader + timing CLI — this build's counterpart of the reference's only benchmark harness
(demo/json_demo.py: load_data_json :255-275, select_mission_by_filename :38-51, warm-up :89-94, timed per-frame
loop :103-131).  Same file schema: a list of {"car_position":[x,y], "car_direction":[dx,dy], "slam_cones":[5 lists of [x,y]]}.

  python -m fsd_path_planning_amd.replay --data-path fsg_19_2_laps.json [--remove-color-info] [--batched] [--output-path out.npz]

Two replay modes:
  per-frame : one PathPlanner, one calculate_path_in_global_frame call per frame, wall-clock per call (what the
              reference's demo measures; for the skidpad mission this is the stateful sequence);
  --batched : the frames of the recording as a stream of batches of independent frames (fresh-planner semantics per
              frame; trackdrive/autocross recordings only), several batches in flight (fsdp_submit / fsdp_collect);
              with --stateful the frames that read the previous path are planned again in order with the path their
              predecessor left: the per-frame replay's results at the batched replay's speed.
"""
from __future__ import annotations

import argparse
import json
import time
from pathlib import Path
from typing import List, Tuple

import numpy as np

from .planner import ConeTypes, MissionTypes, PathPlanner, pack_frames


def select_mission_by_filename(filename: str) -> MissionTypes:
    """"skidpad" in the name -> skidpad, "accel" -> acceleration, else trackdrive (json_demo.py:38-51)."""
    if "skidpad" in filename:
        return MissionTypes.skidpad
    if "accel" in filename:
        return MissionTypes.acceleration
    return MissionTypes.trackdrive


class Recording:
    """A JSON recording in the layout the C ABI takes (include/fsdp.h): one (N, 3) [x, y, type] cone array for all frames
    in the reference's flatten order (UNKNOWN, RIGHT, LEFT, ORANGE_SMALL, ORANGE_BIG — core_trace_sorter.py:37-54), CSR
    offsets (F + 1,) and poses (F, 4) [px, py, dx, dy].  File schema (what demo/json_demo.py reads): a list of frames
    {"car_position": [x, y], "car_direction": [dx, dy], "slam_cones": [5 lists of [x, y], one per ConeTypes value]}."""

    N_TYPES = len(ConeTypes)

    def __init__(self, offsets: np.ndarray, cones: np.ndarray, poses: np.ndarray):
        self.offsets, self.cones, self.poses = offsets, cones, poses

    @classmethod
    def from_json(cls, data_path: Path) -> "Recording":
        frames = json.loads(Path(data_path).read_text())
        counts = np.array([[len(lst) for lst in f["slam_cones"]] for f in frames], dtype=np.int64).reshape(len(frames), cls.N_TYPES)
        offsets = np.zeros(len(frames) + 1, dtype=np.int32)
        np.cumsum(counts.sum(axis=1), out=offsets[1:])
        cones = np.empty((int(offsets[-1]), 3))
        poses = np.empty((len(frames), 4))
        for k, f in enumerate(frames):
            poses[k, :2] = f["car_position"]
            poses[k, 2:] = f["car_direction"]
            at = int(offsets[k])
            for cone_type, lst in enumerate(f["slam_cones"]):
                if lst:
                    cones[at:at + len(lst), :2] = lst
                    cones[at:at + len(lst), 2] = cone_type
                    at += len(lst)
        return cls(offsets, cones, poses)

    def without_color(self) -> "Recording":
        """Every cone UNKNOWN.  The cones of a frame are already stacked in type order, which is the order the reference's
        colour stripping leaves them in (json_demo.py:266-273), so only the type column changes."""
        cones = self.cones.copy()
        cones[:, 2] = float(ConeTypes.UNKNOWN)
        return Recording(self.offsets, cones, self.poses)

    def __len__(self) -> int:
        return len(self.poses)

    def frame_cones_by_type(self, k: int) -> List[np.ndarray]:
        """Frame k as the five per-type (n, 2) arrays `calculate_path_in_global_frame` takes."""
        block = self.cones[self.offsets[k]:self.offsets[k + 1]]
        return [block[block[:, 2] == t, :2] for t in range(self.N_TYPES)]


def load_data_json(data_path: Path, remove_color_info: bool = False) -> Tuple[np.ndarray, np.ndarray, List[List[np.ndarray]]]:
    """(positions (F, 2), directions (F, 2), per frame the five per-type cone arrays) — the tuple the reference's loader
    returns, cut out of a `Recording`."""
    rec = Recording.from_json(data_path)
    if remove_color_info:
        rec = rec.without_color()
    return rec.poses[:, :2].copy(), rec.poses[:, 2:].copy(), [rec.frame_cones_by_type(k) for k in range(len(rec))]


def replay_per_frame(mission, positions, directions, observations, device=None):
    warm = PathPlanner(mission, device=device)  # warm-up on a throw-away planner (json_demo.py:89-94)
    warm.calculate_path_in_global_frame(observations[0], positions[0], directions[0])
    planner = PathPlanner(mission, device=device)
    paths, times, reloc_frame = [], [], None
    for i, (p, d, c) in enumerate(zip(positions, directions, observations)):
        if reloc_frame is None and planner.relocalization_info is not None:
            reloc_frame = i
        t0 = time.perf_counter()
        paths.append(planner.calculate_path_in_global_frame(c, p, d))
        times.append(time.perf_counter() - t0)
    return np.array(paths), np.array(times), reloc_frame, planner.relocalization_info


def replay_batched(mission, positions, directions, observations, device=None, repeats: int = 5, batch_frames: int = 4096, depth: int = 4,
                   devices=None):
    """The recording as a stream of batches of `batch_frames` frames, `depth` of them in flight (fsdp_submit /
    fsdp_collect: a batch's transfers run under the other batches' kernels; page-locked buffers).  Returns the results of
    all frames in recording order and the seconds one replay of the whole recording took (host buffers to host buffers).
    devices (a list of GPU indices or "all"): every batch is cut into contiguous frame ranges, one per GPU, all driven from
    this process (multi.MultiPlanner.plan_stream) — the same bytes."""
    from . import _capi

    if devices is not None:
        from .multi import MultiPlanner

        mp = MultiPlanner(None if devices == "all" else devices, mission=int(mission), overlap=depth)
        frames = list(zip(observations, positions, directions))
        chunks = [pack_frames(frames[lo:lo + batch_frames]) for lo in range(0, len(frames), batch_frames)]
        got = list(mp.plan_stream(chunks))  # warm-up
        t0 = time.perf_counter()
        for _ in range(repeats):
            got = list(mp.plan_stream(chunks))
        sec = (time.perf_counter() - t0) / repeats
        mp.close()
        return (np.concatenate(got) if got else np.zeros(0, _capi.RESULT_DTYPE)), sec
    planner = PathPlanner(mission, device=device)
    ctx = planner._ctx
    frames = list(zip(observations, positions, directions))
    chunks = []
    for lo in range(0, len(frames), batch_frames):
        off, cones, poses = pack_frames(frames[lo:lo + batch_frames])
        chunks.append((_capi.pinned_copy(off, np.int32), _capi.pinned_copy(cones, np.float64), _capi.pinned_copy(poses, np.float64),
                       _capi.pinned_empty(len(poses), ctx.result_dtype)))
    depth = max(1, min(depth, len(chunks)))
    ctx.set_overlap(depth)

    def one_replay():
        inflight = []
        for off, cones, poses, out in chunks:
            if len(inflight) == ctx.ticket_capacity:
                ctx.collect(inflight.pop(0))
            inflight.append(ctx.submit(off, cones, poses, out=out))
        for t in inflight:
            ctx.collect(t)

    one_replay()  # warm-up
    t0 = time.perf_counter()
    for _ in range(repeats):
        one_replay()
    sec = (time.perf_counter() - t0) / repeats
    res = np.concatenate([np.array(c[3]) for c in chunks]) if chunks else np.zeros(0, ctx.result_dtype)
    ctx.set_overlap(1)
    return res, sec


FB_READ_PREVIOUS = 1 | 2 | 4 | 8  # path_fallback bits of the branches that read previous_paths[-1] (include/fsdp.h)


def replay_stateful_batched(mission, positions, directions, observations, device=None, batch_frames: int = 4096, depth: int = 4):
    """The recording as ONE planner sees it — consecutive frames chain through previous_paths[-1] (core_calculate_path.py:
    572-573), which the reference reads in its fallbacks only (:202-203, 218-221, 235-236, 531-536, 564-570) — at the speed
    of the batched replay: all frames as independent frames first (every one with the fresh planner's previous path), then
    the frames that did read the previous path (their path_fallback bits say so), in recording order, once more with the
    path their predecessor really left (a frame the reference raises on leaves none).  Returns the results in recording
    order, the seconds of the batched replay and the number of frames planned again."""
    res, sec = replay_batched(mission, positions, directions, observations, device, repeats=1, batch_frames=batch_frames, depth=depth)
    res = res.copy()
    planner = PathPlanner(mission, device=device)
    ctx = planner._ctx
    prev, again = None, 0
    for k in range(len(res)):
        if (res["path_fallback"][k] & FB_READ_PREVIOUS) and prev is not None:
            off, cones, poses = pack_frames([(observations[k], positions[k], directions[k])])
            res[k] = ctx.plan_batch_sequential(off, cones, poses, prev[None])[0]
            again += 1
        if res["status"][k] == 0:
            prev = np.array(res["path"][k][: ctx.horizon])
    return res, sec, again


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--data-path", "-i", type=Path, required=True)
    ap.add_argument("--remove-color-info", action="store_true")
    ap.add_argument("--batched", action="store_true")
    ap.add_argument("--stateful", action="store_true", help="--batched: one planner's view of the recording (frames chain through the previous path)")
    ap.add_argument("--output-path", "-o", type=Path, default=None)
    ap.add_argument("--device", type=int, default=None)
    ap.add_argument("--devices", type=str, default=None, help='--batched: GPUs the stream is sharded over from this process, e.g. "0,1,2,3" or "all"')
    ap.add_argument("--batch-frames", type=int, default=4096, help="--batched: frames per batch of the stream")
    ap.add_argument("--depth", type=int, default=4, help="--batched: batches in flight")
    a = ap.parse_args(argv)
    mission = select_mission_by_filename(a.data_path.name)
    positions, directions, observations = load_data_json(a.data_path, a.remove_color_info)
    out = {"file": str(a.data_path), "mission": mission.name, "frames": len(positions)}
    if a.batched:
        if a.stateful:
            res, sec, again = replay_stateful_batched(mission, positions, directions, observations, a.device, batch_frames=a.batch_frames, depth=a.depth)
            out.update(frames_planned_again_with_their_predecessors_path=again)
        else:
            devs = None if a.devices is None else ("all" if a.devices == "all" else [int(x) for x in a.devices.split(",")])
            res, sec = replay_batched(mission, positions, directions, observations, a.device, batch_frames=a.batch_frames, depth=a.depth, devices=devs)
        out.update(mode="batched", seconds_per_batch=sec, frames_per_s=len(res) / sec,
                   status_histogram={int(k): int(v) for k, v in zip(*np.unique(res["status"], return_counts=True))})
        paths = res["path"]
    else:
        paths, times, reloc_frame, info = replay_per_frame(mission, positions, directions, observations, a.device)
        out.update(mode="per-frame", p50_us=float(np.median(times) * 1e6), mean_us=float(times.mean() * 1e6),
                   max_us=float(times.max() * 1e6), frames_over_100ms=int((times > 0.1).sum()), relocalized_at_frame=reloc_frame)
        if info is not None:
            out.update(translation=[float(x) for x in info.translation], rotation_deg=float(np.rad2deg(info.rotation)))
    if a.output_path:
        np.savez_compressed(a.output_path, path=paths)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
