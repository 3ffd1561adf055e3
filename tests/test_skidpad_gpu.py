"""Skidpad mission (BASELINE config 5) on the MI355X through the C ABI: stateful planner instances with rigidly
perturbed starts against oracle planners (det-math mode) and against the reference's golden sequence."""
import importlib

import numpy as np
import pytest

import oracle_lib
import skidpad_support as sk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return importlib.import_module("ft-fsd-path-planning_amd")


def test_reference_shaped_skidpad_planner_replays_golden_sequence(pkg, golden_dir):
    g = sk.load_sequence(golden_dir)
    planner = pkg.PathPlanner(pkg.MissionTypes.skidpad, device=0)
    for t in range(len(g["poses"])):
        xyt, pose = sk.frame(g, t)
        cones_by_type = [xyt[xyt[:, 2] == k, :2] for k in range(5)]
        path = planner.calculate_path_in_global_frame(cones_by_type, pose[:2], pose[2:])
        ri = planner.relocalization_info
        assert (ri is not None) == bool(g["relocalized"][t]), t
        if ri is not None:
            assert np.abs(np.concatenate([ri.translation, [ri.rotation]]) - g["info"][t]).max() < 1e-9
        assert np.abs(path - g["path"][t]).max() <= 1e-5, t  # every one of the 341 frames, no allowance


def test_perturbed_instances_equal_oracle(pkg, golden_dir):
    """64 instances x 80 frames, every output bit-identical to 64 stateful oracle planners."""
    g = sk.load_sequence(golden_dir)
    n = 64
    tf = sk.perturbed_instances(g, n)
    batch = pkg.SkidpadBatch(n, device=0)
    table, noise = batch.tables
    # the constants the device derived from the table == the reference's (golden) bits
    ref, md = batch.constants
    assert np.array_equal(ref, g["reference_centers"])
    assert md == float(np.mean(np.linalg.norm(np.diff(table[::2][:10], axis=-2), axis=-1)))
    with oracle_lib.math_mode(1):
        ops = [oracle_lib.SkidpadPlanner(table, noise) for _ in range(n)]
        for t in range(80):
            off, cones, poses = sk.batch_for_step(g, t, tf)
            res, info = batch.step(off, cones, poses)
            for i, op in enumerate(ops):
                r, oi = op.step(cones[off[i] : off[i + 1]], poses[i])
                assert int(res[i]["status"]) == int(r["status"]), (t, i)
                assert int(info[i]["relocalized"]) == int(oi[0]) and int(info[i]["index_along_path"]) == int(oi[4]), (t, i)
                assert np.abs(res[i]["path"] - r["path"]).max() <= 1e-9, (t, i)
    assert info["relocalized"].mean() > 0.9


def test_skidpad_on_the_wide_build_with_a_long_horizon(pkg, golden_dir):
    """mpc_prediction_horizon = 56 (config.py:58; beyond the standard build's 40 rows): the planners live in a context of the
    wide build (64-row states, previous paths and results) — single steps and a replay submitted ahead (the packed path-stage
    route included from 2048 pairs) against oracle planners of the wide oracle build, bit for bit."""
    import oracle_lib_wide

    g = sk.load_sequence(golden_dir)
    n = 48
    prm = dict(mpc_prediction_horizon=56)
    tf = sk.perturbed_instances(g, n)
    batch = pkg.SkidpadBatch(n, device=0, params=prm)
    assert batch._ctx.shapes is pkg.WIDE
    table, noise = batch.tables
    frames = [sk.batch_for_step(g, t, tf) for t in range(60)]
    with oracle_lib_wide.params(prm), oracle_lib_wide.math_mode(1):
        ops = [oracle_lib_wide.SkidpadPlanner(table, noise) for _ in range(n)]
        want = []
        for off, cones, poses in frames:
            want.append([op.step(cones[off[i] : off[i + 1]], poses[i]) for i, op in enumerate(ops)])
    for t, (off, cones, poses) in enumerate(frames[:30]):
        res, info = batch.step(off, cones, poses)
        assert res["path"].shape == (n, 64, 4)
        for i in range(n):
            r, oi = want[t][i]
            assert int(res[i]["status"]) == int(r["status"]) == 0 and int(info[i]["relocalized"]) == int(oi[0])
            assert np.array_equal(res[i]["path"], r["path"], equal_nan=True), (t, i)
    batch.reset()
    for t, (res, info) in enumerate(batch.replay(frames, depth=16)):
        for i in range(n):
            r, oi = want[t][i]
            assert int(info[i]["index_along_path"]) == int(oi[4]) and np.array_equal(res[i]["path"], r["path"], equal_nan=True), (t, i)


def test_reset_gives_fresh_planners(pkg, golden_dir):
    g = sk.load_sequence(golden_dir)
    batch = pkg.SkidpadBatch(2, device=0)
    tf = sk.perturbed_instances(g, 2)
    first = None
    for rep in range(2):
        outs = []
        for t in range(25):
            off, cones, poses = sk.batch_for_step(g, t, tf)
            res, info = batch.step(off, cones, poses)
            outs.append(res["path"].copy())
        if first is None:
            first = outs
        else:
            assert all(np.array_equal(a, b) for a, b in zip(first, outs))
        batch.reset()


def test_full_size_config5(pkg, golden_dir):
    """BASELINE config 5 at its full size: 1024 planner instances (rigidly perturbed starts) x all 341 frames of the
    recording.  Properties on every instance (relocalization, window index, finite paths), instance 0 (the unperturbed
    recording) against the reference's golden sequence, and 12 sampled instances against stateful oracle planners."""
    g = sk.load_sequence(golden_dir)
    n, T = 1024, len(g["poses"])
    tf = sk.perturbed_instances(g, n)
    batch = pkg.SkidpadBatch(n, device=0)
    table, noise = batch.tables
    sample = [0, 1, 2, 3, 100, 257, 511, 512, 700, 901, 1000, 1023]
    # vectorised form of skidpad_support.batch_for_step (every instance sees the same cones under its own rigid transform)
    R = np.stack([r for r, _ in tf])
    tr = np.stack([t for _, t in tf])
    reloc_frame = np.full(n, -1)
    last_idx = np.zeros(n, np.int64)
    with oracle_lib.math_mode(1):
        ops = {i: oracle_lib.SkidpadPlanner(table, noise) for i in sample}
        for t in range(T):
            xyt, pose = sk.frame(g, t)
            m = len(xyt)
            cones = np.empty((n, m, 3))
            cones[:, :, :2] = np.einsum("nij,mj->nmi", R, xyt[:, :2]) + tr[:, None, :]
            cones[:, :, 2] = xyt[:, 2]
            poses = np.concatenate([np.einsum("nij,j->ni", R, pose[:2]) + tr, np.einsum("nij,j->ni", R, pose[2:])], axis=1)
            off = (np.arange(n + 1) * m).astype(np.int32)
            res, info = batch.step(off, cones.reshape(-1, 3), poses)
            assert (res["status"] == 0).all(), (t, np.unique(res["status"]))
            assert np.isfinite(res["path"]).all()
            newly = (info["relocalized"] != 0) & (reloc_frame < 0)
            reloc_frame[newly] = t
            rel = info["relocalized"] != 0
            # the window index never runs backwards by more than the search radius and stays inside the table
            assert (info["index_along_path"][rel] >= 0).all() and (info["index_along_path"][rel] < len(table) // 2 + 1).all()
            last_idx[rel] = info["index_along_path"][rel]
            # instance 0 = the reference's recording
            assert bool(info["relocalized"][0]) == bool(g["relocalized"][t])
            assert np.abs(res["path"][0] - g["path"][t]).max() <= 1e-5, t
            for i, op in ops.items():
                r, oi = op.step(cones[i], poses[i])
                assert int(res[i]["status"]) == int(r["status"]) and int(info[i]["relocalized"]) == int(oi[0]), (t, i)
                assert int(info[i]["index_along_path"]) == int(oi[4]), (t, i)
                assert np.abs(res[i]["path"] - r["path"]).max() <= 1e-9, (t, i)
    assert (reloc_frame >= 0).mean() > 0.95, float((reloc_frame >= 0).mean())   # relocalization success count
    assert np.median(reloc_frame[reloc_frame >= 0]) <= 40


def _same_fields(a, b):
    """bit-equal field by field (the records' padding bytes are nobody's)"""
    return a.dtype == b.dtype and all(np.ascontiguousarray(a[k]).tobytes() == np.ascontiguousarray(b[k]).tobytes() for k in a.dtype.names)


@pytest.mark.parametrize("n,group,pack_min", [(3, None, None), (64, 3, None), (700, None, None), (1100, 5, None), (40, 16, 1), (1600, None, None), (24, 7, 100)])
def test_steps_submitted_ahead_equal_single_steps(pkg, golden_dir, monkeypatch, n, group, pack_min):
    """Steps in flight (csrc/skidpad_kernel.h): a replay that submits ahead has up to 16 consecutive steps planned by one
    group of launches — a wavefront per (instance, step), each working from the window index its predecessors' poses lead
    to and waiting for its predecessor's published state before it keeps or repeats its result; or, from 2048 (instance,
    step) pairs (option "skid_pack_min"), the packed kernels of the autocross path stage with the planners' own wavefronts
    committing the steps in order.  Results, planner information and the states' further course must be those of one
    launch per step — bit for bit, through the relocalization, through steps that read the previous path (a car 60 m off
    the track), steps that fail (positions that are not finite) and jumps of the window index."""
    if group:
        monkeypatch.setitem(pkg._capi.DEFAULT_OPTIONS, "skid_group", group)
    if pack_min:
        monkeypatch.setitem(pkg._capi.DEFAULT_OPTIONS, "skid_pack_min", pack_min)
    g = sk.load_sequence(golden_dir)
    tf = sk.perturbed_instances(g, n)
    frames = sk.awkward_frames(g, tf, 64)
    one = pkg.SkidpadBatch(n, device=0)
    ref = []
    for f in frames:
        res, info = one.step(*f)
        ref.append((res.copy(), info.copy()))
    assert any(r["path_fallback"].any() for r, _ in ref) and any((r["status"] != 0).any() for r, _ in ref)
    batch = pkg.SkidpadBatch(n, device=0)
    for depth in (32, 3):
        batch.reset()
        got = list(batch.replay(frames, depth))
        for t, ((res, info), (r0, i0)) in enumerate(zip(got, ref)):
            assert _same_fields(res, r0) and _same_fields(info, i0), (depth, t)
    # and the states carry on identically: ten more steps, one at a time
    for t in range(64, 74):
        f = sk.batch_for_step(g, t, tf)
        a, b = batch.step(*f), one.step(*f)
        assert _same_fields(a[0], b[0]) and _same_fields(a[1], b[1]), t


def test_random_skidpad_traffic(pkg, golden_dir, monkeypatch):
    """Stress of the deferred launches: 50 planners over 150 frames (the awkward ones first), steps submitted in bursts of
    random length, collected in random order and at random times (a collect launches whatever group is pending, so groups
    of every size 1 ... 16 occur and both routes — option "skid_pack_min" lowered to 300 pairs — alternate), single blocking
    steps in between.  Every result must be the one-launch-per-step result, bit for bit."""
    monkeypatch.setitem(pkg._capi.DEFAULT_OPTIONS, "skid_pack_min", 300)
    n = 50
    g = sk.load_sequence(golden_dir)
    tf = sk.perturbed_instances(g, n)
    frames = sk.awkward_frames(g, tf, 64) + [sk.batch_for_step(g, t, tf) for t in range(64, 150)]
    one = pkg.SkidpadBatch(n, device=0)
    ref = []
    for f in frames:
        res, info = one.step(*f)
        ref.append((res.copy(), info.copy()))
    rng = np.random.default_rng(7)
    for depth in (32, 5):
        batch = pkg.SkidpadBatch(n, device=0)
        batch.set_overlap(depth)
        inflight, t, checked = [], 0, 0
        while t < len(frames) or inflight:
            r = rng.random()
            if t < len(frames) and not inflight and r < 0.1:
                res, info = batch.step(*frames[t])  # a blocking step (nothing outstanding)
                assert _same_fields(res, ref[t][0]) and _same_fields(info, ref[t][1]), (depth, t)
                t += 1
                checked += 1
            elif t < len(frames) and len(inflight) < depth and r < 0.7:
                for _ in range(int(rng.integers(1, depth + 1))):
                    if t == len(frames) or len(inflight) == depth:
                        break
                    inflight.append((t, batch.submit(*frames[t])))
                    t += 1
            elif inflight:
                # tickets of one slot ring: the oldest first or any other one
                k = 0 if rng.random() < 0.5 else int(rng.integers(len(inflight)))
                j, tk = inflight.pop(k)
                res, info = batch.collect(tk)
                assert _same_fields(res, ref[j][0]) and _same_fields(info, ref[j][1]), (depth, j)
                checked += 1
        assert checked == len(frames)


def test_timing_launches_leave_no_trace(pkg, golden_dir, monkeypatch):
    """fsdp_skidpad_time_path repeats the last step's path kernel and restores the planners' state — and the publish
    counters the wavefronts of grouped launches wait on (round-3 advisor: with stale counters the second step of the next
    group did not wait for the first one's state).  Steps, timing launches, then a replay submitted ahead through the
    wavefront-per-(instance, step) kernel: every result must equal the one-launch-per-step planners that were never timed."""
    monkeypatch.setitem(pkg._capi.DEFAULT_OPTIONS, "skid_group", 5)
    monkeypatch.setitem(pkg._capi.DEFAULT_OPTIONS, "skid_pack_min", 1000000)  # keep the groups on skid_path_kernel (the kernel that waits)
    n = 48
    g = sk.load_sequence(golden_dir)
    tf = sk.perturbed_instances(g, n)
    frames = sk.awkward_frames(g, tf, 64)
    one = pkg.SkidpadBatch(n, device=0)
    ref = []
    for f in frames:
        res, info = one.step(*f)
        ref.append((res.copy(), info.copy()))
    batch = pkg.SkidpadBatch(n, device=0)
    for t in range(30):
        res, info = batch.step(*frames[t])
        assert _same_fields(res, ref[t][0]) and _same_fields(info, ref[t][1]), t
    assert batch.time_path(7) > 0.0
    got = list(batch.replay(frames[30:], 16))
    for t, (res, info) in enumerate(got, start=30):
        assert _same_fields(res, ref[t][0]) and _same_fields(info, ref[t][1]), t
    with oracle_lib.math_mode(1):
        table, noise = batch.tables
        op = oracle_lib.SkidpadPlanner(table, noise)
        for t, f in enumerate(frames):
            off, cones, poses = f
            r, oi = op.step(cones[off[0] : off[1]], poses[0])
            assert int(ref[t][0][0]["status"]) == int(r["status"]), t
            if int(r["status"]) == 0:
                assert np.abs(ref[t][0][0]["path"] - r["path"]).max() <= 1e-9, t


@pytest.mark.parametrize("n,pinned", [(40, True), (40, False), (1100, True)])
def test_compact_results_are_the_path_fields_of_the_full_ones(pkg, golden_dir, n, pinned):
    """fsdp_skidpad_submit_compact: path, status, fallback bits and dense-sample count of every planner and step — what a
    skidpad step produces (sorting and matching are skipped, full_pipeline.py:138-140) — in 1296-byte records, into page-locked
    or pageable memory, for groups on both routes; equal to those fields of the full 2408-byte results, information included."""
    g = sk.load_sequence(golden_dir)
    tf = sk.perturbed_instances(g, n)
    frames = sk.awkward_frames(g, tf, 48)
    full = pkg.SkidpadBatch(n, device=0)
    ref = [(r.copy(), i.copy()) for r, i in full.replay(frames, 16)]
    batch = pkg.SkidpadBatch(n, device=0)
    batch.set_overlap(16)
    inflight, got = [], []
    for f in frames:
        if len(inflight) == 16:
            r, i = batch.collect(inflight.pop(0))
            got.append((r.copy(), i.copy()))
        out = pkg.pinned_empty(n, pkg.PATH_RESULT_DTYPE) if pinned else np.zeros(n, pkg.PATH_RESULT_DTYPE)
        inflight.append(batch.submit(*f, out=out))
    for tk in inflight:
        r, i = batch.collect(tk)
        got.append((r.copy(), i.copy()))
    assert len(got) == len(ref)
    for t, ((r, i), (r0, i0)) in enumerate(zip(got, ref)):
        assert r.dtype == pkg.PATH_RESULT_DTYPE and r.itemsize == 1296
        for k in ("path", "status", "path_fallback", "n_dense"):
            assert np.ascontiguousarray(r[k]).tobytes() == np.ascontiguousarray(r0[k]).tobytes(), (t, k)
        assert _same_fields(i, i0), t
