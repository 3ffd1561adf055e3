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
