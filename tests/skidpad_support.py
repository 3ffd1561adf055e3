"""Helpers for the skidpad (BASELINE config 5) tests: the golden sequence and rigidly perturbed starts."""
import numpy as np


def load_sequence(golden_dir):
    g = np.load(golden_dir / "skidpad_sequence.npz")
    return g


def frame(g, t):
    o = g["offsets"]
    return g["cones"][o[t] : o[t + 1]], g["poses"][t]


def perturbed_instances(g, n_inst, seed=3, max_shift=0.5, max_rot_deg=5.0):
    """Config 5: each planner instance replays the recording under its own rigid perturbation of the start pose
    (uniform +-0.5 m, +-5 deg), applied to poses AND cones.  Returns per-instance (R (2,2), t (2,))."""
    rng = np.random.default_rng(seed)
    ang = np.deg2rad(rng.uniform(-max_rot_deg, max_rot_deg, n_inst))
    sh = rng.uniform(-max_shift, max_shift, (n_inst, 2))
    ang[0] = 0.0
    sh[0] = 0.0  # instance 0 = the unperturbed recording
    c0 = g["poses"][0, :2]
    tf = []
    for a, s in zip(ang, sh):
        R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
        tf.append((R, c0 + s - R @ c0))  # rotate about the first car position, then shift
    return tf


def batch_for_step(g, t, tf):
    """(offsets, cones_xyt, poses) of frame t for all instances."""
    xyt, pose = frame(g, t)
    cones, poses, off = [], [], [0]
    for R, tr in tf:
        c = xyt.copy()
        c[:, :2] = xyt[:, :2] @ R.T + tr
        cones.append(c)
        poses.append(np.concatenate([R @ pose[:2] + tr, R @ pose[2:]]))
        off.append(off[-1] + len(c))
    return np.array(off, np.int32), np.concatenate(cones).reshape(-1, 3), np.array(poses)


def awkward_frames(g, tf, n_frames=56):
    """The first frames of the golden sequence for the instances tf, with the poses that break a planner's routine: a car
    60 m off the track (the too-far check hands the step the previous path), positions / directions that are not finite
    (the reference raises for the former and moves the window index first: skidpad_calculate_path.py:66-67), a car moved
    15 m along the track (another window index than the recorded poses lead to)."""
    frames = []
    for t in range(n_frames):
        off, cones, poses = batch_for_step(g, t, tf)
        poses = poses.copy()
        if t in (5, 30):
            poses[:, 1] += 60.0
        if t == 9:
            poses[0, 0] = np.inf
        if t == 35:
            poses[1, 0] = np.nan
        if t == 37:
            poses[2, 1] = -np.inf
        if t == 38:
            poses[0, 2] = np.nan
        if t in (41, 42):
            poses[:, 0] += 15.0 * poses[:, 2]
            poses[:, 1] += 15.0 * poses[:, 3]
        frames.append((off, cones, poses))
    return frames
