"""The kernel SOURCES of ft-fsd-path-planning_amd/csrc executed on the CPU by the host SIMT emulator
(tests/emu/: one fiber per lane, rendezvous for ballot/shuffle/barrier) and compared with the oracle.
This is how kernel logic is checked where no GPU exists; the -m gpu tests repeat the comparison on
hardware through the real library."""
import numpy as np
import pytest

import emu_lib
import oracle_lib
import parity


@pytest.mark.parametrize("name,stride", [("scenarios", 1), ("cfg2_color", 4), ("cfg3_nocolor", 8), ("cfg4_200cones", 8),
                                          ("cfg4_noisy_nocolor", 6), ("fuzz", 5), ("big_frames", 4), ("lattice", 2), ("nonfinite_poses", 1), ("nonfinite_cones", 2), ("odd_inputs", 1)])
@pytest.mark.parametrize("group", emu_lib.PATH_GROUP_SIZES)
def test_emulated_kernels_equal_oracle(golden_dir, name, stride, group):
    g = np.load(golden_dir / f"{name}.npz")
    if group != 8:
        stride *= 3  # the other two instantiations on a thinner sample
    idx = np.arange(0, len(g["ok"]), stride)
    off = np.concatenate([[0], np.cumsum([g["offsets"][i + 1] - g["offsets"][i] for i in idx])]).astype(np.int32)
    cones = np.concatenate([g["cones"][g["offsets"][i] : g["offsets"][i + 1]] for i in idx])
    poses = g["poses"][idx]
    res, n_dense = emu_lib.plan(off, cones, poses, group)
    with oracle_lib.math_mode(1):  # the kernels use det_math.h for the arc extension
        ref = oracle_lib.plan_batch(off, cones, poses)
    assert np.array_equal(res["status"], ref["status"])
    ok = ref["status"] == 0
    for f in ("left_idx", "right_idx", "n_left_v", "n_right_v", "l2r", "r2l", "left_v", "right_v", "path_fallback"):
        assert np.array_equal(res[f][ok], ref[f][ok]), f
    parity.assert_intermediates_equal(res, ref, ok)
    if name in ("big_frames", "lattice") and group == 8:
        assert emu_lib.lib().emu_last_big() > 0  # the sorting stage took the global-memory route for some frames
    # no libm value enters the float chain -> bit-identical
    assert np.array_equal(res["path"][ok], ref["path"][ok])


def test_emulated_default_path_equals_reference(golden_dir):
    ref = np.load(golden_dir / "default_path.npz")["path"]
    assert np.abs(emu_lib.default_path() - ref).max() < 1e-12


@pytest.mark.parametrize("name", ["params_sort", "params_path", "params_monotonic", "params_deg2", "params_deg1", "params_horizon", "params_no_unknown"])
def test_emulated_kernels_with_non_default_parameters(golden_dir, name):
    """The parameter block (include/fsdp.h fsdp_params) through the kernel sources: non-default constructor kwargs of the
    reference's stage classes, kernels == oracle bit for bit, and both == the reference's goldens."""
    g = np.load(golden_dir / f"{name}.npz")
    prm = dict(zip(g["param_names"].tolist(), g["param_values"].tolist()))
    idx = np.arange(0, len(g["ok"]), 3)
    off = np.concatenate([[0], np.cumsum([g["offsets"][i + 1] - g["offsets"][i] for i in idx])]).astype(np.int32)
    cones = np.concatenate([g["cones"][g["offsets"][i] : g["offsets"][i + 1]] for i in idx])
    poses = g["poses"][idx]
    with emu_lib.params(prm):
        res, _ = emu_lib.plan(off, cones, poses, 1008)
    with oracle_lib.params(prm), oracle_lib.math_mode(1):
        ref = oracle_lib.plan_batch(off, cones, poses)
    assert np.array_equal(res["status"], ref["status"])
    ok = ref["status"] == 0
    for f in ("left_idx", "right_idx", "n_left_v", "n_right_v", "l2r", "r2l", "left_v", "right_v", "path_fallback"):
        assert np.array_equal(res[f][ok], ref[f][ok]), f
    parity.assert_intermediates_equal(res, ref, ok)
    assert np.array_equal(res["path"][ok], ref["path"][ok], equal_nan=True)  # (rows beyond a shorter horizon are NaN on both sides)
    arc = parity.ArcLibm(golden_dir, name)
    flips = []
    for j, k in enumerate(idx):
        cat, detail = parity.compare_frame(res[j], g, int(k), arc=arc)
        assert cat in ("ok", "ref_undefined", "flip"), (k, cat, detail)
        flips += [int(k)] if cat == "flip" else []
    assert flips == [f for f in arc.flips("det") if f in set(idx.tolist())], flips


@pytest.mark.parametrize("name,group", [("params_wide_sort", 1008), ("params_wide_horizon", 1008), ("params_wide_all", 1004), ("params_wide_all", 64),
                                        ("params_wide_sort", 16)])
def test_emulated_wide_build_with_parameters_beyond_the_standard_shapes(golden_dir, name, group):
    """The kernel sources compiled with -DFSDP_WIDE_SHAPES (libfsdp_hip_wide.so: max_n_neighbors <= 8 — two rounds of
    (candidate, neighbour) pairs per pop —, max_length <= 16 — four configurations per cost round, matching on the whole
    wavefront —, horizon <= 64 — up to 193 dense samples): kernels == the oracle's wide build bit for bit, and both == the
    reference's goldens for those parameters (make_golden.py --params-r5)."""
    import emu_lib_wide
    import oracle_lib_wide

    g = np.load(golden_dir / f"{name}.npz")
    prm = dict(zip(g["param_names"].tolist(), g["param_values"].tolist()))
    idx = np.arange(0, len(g["ok"]), 3 if group == 1008 else 5)
    off = np.concatenate([[0], np.cumsum([g["offsets"][i + 1] - g["offsets"][i] for i in idx])]).astype(np.int32)
    cones = np.concatenate([g["cones"][g["offsets"][i] : g["offsets"][i + 1]] for i in idx])
    poses = g["poses"][idx]
    with emu_lib_wide.params(prm):
        res, _ = emu_lib_wide.plan(off, cones, poses, group)
    with oracle_lib_wide.params(prm), oracle_lib_wide.math_mode(1):
        ref = oracle_lib_wide.plan_batch(off, cones, poses)
    assert np.array_equal(res["status"], ref["status"])
    ok = ref["status"] == 0
    for f in ("left_idx", "right_idx", "n_left_v", "n_right_v", "l2r", "r2l", "left_v", "right_v", "path_fallback"):
        assert np.array_equal(res[f][ok], ref[f][ok]), f
    parity.assert_intermediates_equal(res, ref, ok)
    assert np.array_equal(res["path"][ok], ref["path"][ok], equal_nan=True)
    if name != "params_wide_horizon":
        assert max(int(ref["n_left"].max()), int(ref["n_right"].max())) > 12  # sides longer than the standard shapes hold
    arc = parity.ArcLibm(golden_dir, name)
    flips = []
    for j, k in enumerate(idx):
        cat, detail = parity.compare_frame(res[j], g, int(k), arc=arc)
        assert cat in ("ok", "ref_undefined", "flip"), (k, cat, detail)
        flips += [int(k)] if cat == "flip" else []
    assert flips == [f for f in arc.flips("det") if f in set(idx.tolist())], flips


@pytest.mark.parametrize("seed,frame,knots,group", [(511, 901, 68, 64), (516, 1001, 171, 1004)])
def test_refits_beyond_64_knots_are_planned_not_refused(seed, frame, knots, group):
    """FITPACK lets a smoothing spline take nest = m + 2k knots (utils/spline_fit.py:117 -> splprep); the exact kernels' last level (a
    frame with the whole wavefront: path_kernel<64>, the second level of path_retry_kernel) keeps 256 of them.  The two noisiest
    frames of round 5's wide fuzz (sides of 16 cones; refits of 68 and 171 knots, refused with FSDP_OVERFLOW_KNOTS while the
    capacity was 64) now come back planned, equal to the oracle bit for bit — through the one-kernel path stage and through the
    three-kernel one (packed kernels -> retry list -> four frames per wavefront -> the whole wavefront)."""
    import importlib

    import emu_lib_wide
    import oracle_lib_wide

    pkg = importlib.import_module("ft-fsd-path-planning_amd")
    per_side, track, noise, colour = {511: (64, 0.1, 0.3, True), 516: (64, 0.3, 0.3, False)}[seed]
    off, cones, poses = pkg.synth.make_replay_batch(1024, per_side, track, seed=seed, frame_noise=noise, random_pose=True, color=colour)
    lo, hi = frame - 2, frame + 2
    o = (off[lo : hi + 1] - off[lo]).astype(np.int32)
    c, p = cones[off[lo] : off[hi]], poses[lo:hi]
    prm = dict(max_n_neighbors=8, max_length=16)
    with oracle_lib_wide.params(prm), oracle_lib_wide.math_mode(1):
        ref = oracle_lib_wide.plan_batch(o, c, p)
        _, nf, fits = oracle_lib_wide.plan_frame_capture(c[o[2] : o[3]], p[2])
    assert max(f[1] for f in fits[:nf]) == knots and ref["status"][2] == 0
    with emu_lib_wide.params(prm):
        res, _ = emu_lib_wide.plan(o, c, p, group)
    assert np.array_equal(res["status"], ref["status"])
    ok = ref["status"] == 0
    assert np.array_equal(res["path"][ok], ref["path"][ok], equal_nan=True) and np.array_equal(res["path_fallback"][ok], ref["path_fallback"][ok])


@pytest.mark.parametrize("name,stride", [("big_frames", 3), ("lattice", 7)])
def test_emulated_wide_build_routes_beyond_the_lds_capacities(golden_dir, name, stride):
    """The wide build's sort_big_kernel (frame state in global memory: 300 / 600 cones per frame, more than 64 raw end
    configurations per side) under max_n_neighbors = 8, max_length = 16: equal to the oracle's wide build; where eight neighbours
    on a lattice make a side grow beyond the 4096 raw end configurations the state holds (the oracle counts 37 000 and more) the
    frame is refused with FSDP_OVERFLOW_ENDS, never truncated (include/fsdp.h)."""
    import emu_lib_wide
    import oracle_lib_wide

    # (on the lattice eight neighbours make most sides explode; six with length 14 leave frames on either side of the capacity)
    prm = dict(max_n_neighbors=8, max_length=16, mpc_prediction_horizon=64) if name == "big_frames" else dict(max_n_neighbors=6, max_length=14)
    g = np.load(golden_dir / f"{name}.npz")
    idx = np.arange(0, len(g["ok"]), stride)
    off = np.concatenate([[0], np.cumsum([g["offsets"][i + 1] - g["offsets"][i] for i in idx])]).astype(np.int32)
    cones = np.concatenate([g["cones"][g["offsets"][i] : g["offsets"][i + 1]] for i in idx])
    poses = g["poses"][idx]
    with emu_lib_wide.params(prm):
        res, _ = emu_lib_wide.plan(off, cones, poses, 8)
        assert emu_lib_wide.lib().emu_last_big() > 0
    with oracle_lib_wide.params(prm), oracle_lib_wide.math_mode(1):
        ref = oracle_lib_wide.plan_batch(off, cones, poses, n_threads=4)
    refused = res["status"] == 202
    assert (np.maximum(ref["n_configs_left"], ref["n_configs_right"])[refused] > 1024).all()  # (the oracle's counts AFTER the post-filters: the raw ones, which the state holds, are larger)
    ok = (ref["status"] == 0) & ~refused
    assert ok.sum() >= 2 and np.array_equal(res["status"][~refused], ref["status"][~refused])
    for f in ("left_idx", "right_idx", "n_left_v", "n_right_v", "l2r", "r2l", "left_v", "right_v", "path_fallback"):
        assert np.array_equal(res[f][ok], ref[f][ok]), f
    parity.assert_intermediates_equal(res, ref, ok)
    assert np.array_equal(res["path"][ok], ref["path"][ok], equal_nan=True)


@pytest.mark.parametrize("name,stride", [("scenarios", 1), ("cfg3_nocolor", 16), ("lattice", 6), ("odd_inputs", 1)])
def test_emulated_wide_build_with_the_default_parameters(golden_dir, name, stride):
    """The wide build under the reference's DEFAULT parameters (a stage object fed more than 12 sorted cones per side runs on it
    whatever its parameters, stages.py): the same results as the standard build's oracle, in the wider records."""
    import emu_lib_wide

    g = np.load(golden_dir / f"{name}.npz")
    idx = np.arange(0, len(g["ok"]), stride)
    off = np.concatenate([[0], np.cumsum([g["offsets"][i + 1] - g["offsets"][i] for i in idx])]).astype(np.int32)
    cones = np.concatenate([g["cones"][g["offsets"][i] : g["offsets"][i + 1]] for i in idx])
    poses = g["poses"][idx]
    res, _ = emu_lib_wide.plan(off, cones, poses, 8)
    with oracle_lib.math_mode(1):
        ref = oracle_lib.plan_batch(off, cones, poses)
    assert np.array_equal(res["status"], ref["status"])
    ok = ref["status"] == 0
    for f, cap in (("left_idx", 12), ("right_idx", 12), ("l2r", 24), ("r2l", 24), ("left_v", 24), ("right_v", 24), ("path", 40)):
        assert np.array_equal(res[f][ok][:, :cap], ref[f][ok], equal_nan=True), f
    assert np.isnan(res["path"][ok][:, 40:]).all() and (res["left_idx"][ok][:, 12:] == -1).all()
    for f in ("n_left", "n_right", "n_left_v", "n_right_v", "path_fallback"):
        assert np.array_equal(res[f][ok], ref[f][ok]), f
    parity.assert_intermediates_equal(res, ref, ok)


@pytest.mark.parametrize("group", [1004, 1008, 1016])
def test_knot_capacity_reached_in_the_middle_of_a_round(group):
    """A fit whose round of new knots crosses the workspace's capacity (16 knots in the three-kernel path stage) must be
    handed to the exact kernel: stopping at the capacity and going on would converge on a knot set the reference never
    has (frame 182 of this noisy colourless set: 15 -> 17 knots in one round; the capped fit ended with 16 and a
    different spline — found when the GPU test of the kernel instantiations was pinned to the three-kernel stage)."""
    import importlib

    pkg = importlib.import_module("ft-fsd-path-planning_amd")
    off, cones, poses = pkg.synth.make_replay_batch(211, 100, 0.0, seed=22, frame_noise=0.3, random_pose=True, color=False)
    lo, hi = 176, 192
    o = (off[lo : hi + 1] - off[lo]).astype(np.int32)
    c, p = cones[off[lo] : off[hi]], poses[lo:hi]
    res, _ = emu_lib.plan(o, c, p, group)
    assert emu_lib.last_retries() >= 1
    with oracle_lib.math_mode(1):
        ref = oracle_lib.plan_batch(o, c, p)
    assert np.array_equal(res["status"], ref["status"])
    ok = ref["status"] == 0
    assert np.array_equal(res["path"][ok], ref["path"][ok])


def test_sorting_state_sizes_agree(golden_dir):
    """The library sorts a batch whose frames hold at most 128 cones with the 128-cone frame state (sort_kernel_128, four
    wavefronts per SIMD) and any other batch with the 255-cone state: same code, same results (every output field)."""
    g = np.load(golden_dir / "fuzz.npz")
    idx = np.arange(0, len(g["ok"]), 3)
    off = np.concatenate([[0], np.cumsum([g["offsets"][i + 1] - g["offsets"][i] for i in idx])]).astype(np.int32)
    cones = np.concatenate([g["cones"][g["offsets"][i] : g["offsets"][i + 1]] for i in idx])
    poses = g["poses"][idx]
    assert np.diff(off).max() <= 128
    small = emu_lib.sort(off, cones, poses)
    emu_lib.lib().emu_set_no_sort128(1)  # (the library's option "no_sort128")
    try:
        large = emu_lib.sort(off, cones, poses)
    finally:
        emu_lib.lib().emu_set_no_sort128(0)
    assert small.tobytes() == large.tobytes()
    # a frame with exactly 128 cones fits the small state, 129 cones do not (the batch then takes the 255-cone kernel)
    pkg_synth = __import__("importlib").import_module("ft-fsd-path-planning_amd.synth")
    for per_side in (64, 65):
        o, c, p = pkg_synth.make_replay_batch(6, per_side, 0.15, seed=5, color=True)
        with oracle_lib.math_mode(1):
            ref = oracle_lib.plan_batch(o, c, p)
        got = emu_lib.sort(o, c, p)
        assert np.array_equal(got["status"], ref["status"])
        assert np.array_equal(got["left_idx"], ref["left_idx"]) and np.array_equal(got["right_idx"], ref["right_idx"])

