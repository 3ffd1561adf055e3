"""The N > 1 plumbing (ft-fsd-path-planning_amd/dist.py, used by bench.py) on CPU with 2 and 3 real processes: the rank
rendezvous with its launch-key handshake, the three collectives the bench needs — broadcast of a constant table
(``broadcast_check_table``), a scalar all-reduce (``max_over_ranks`` / ``sum_over_ranks``) and the barrier — over the
transport that needs no GPU (the TCP star, which is also what every run falls back to when RCCL fails on any rank),
contiguous frame-range sharding of a global batch (BASELINE config 4), weak-scaling seeds, and the launcher-less start
(``dist.spawn_ranks``, what ``python bench.py --gpus N`` uses when no launcher set WORLD_SIZE).  The RCCL calls themselves
need GPUs: tests/test_gpu_parity.py::test_rccl_single_rank_communicator runs them with a one-rank communicator."""
import importlib
import json
import multiprocessing as mp
import os
import socket
import subprocess
import sys
import textwrap
import time
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p if p < 65000 else 29531


def _worker(rank, world, port, key, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      FSDP_LAUNCH_KEY=key)
    pkg = importlib.import_module("ft-fsd-path-planning_amd")
    d = pkg.dist.Dist()  # no context (no GPU here): the TCP star carries the collectives
    assert (d.rank, d.world, d.transport, d.comm_size) == (rank, world, "tcp-fallback", world)
    # broadcast of a constant table from rank 0 (on a GPU box: the skidpad track table / the previous-path table)
    table = np.arange(5786 * 2, dtype=np.float64).reshape(5786, 2) * 0.25
    got = d.broadcast_array(table if rank == 0 else None, (5786, 2))
    same = d.broadcast_check_table(table)
    # a rank whose own copy differs by one bit makes the check fail on EVERY rank
    mine = table.copy()
    if rank == world - 1:
        mine.view(np.uint64)[17, 1] ^= 1
    differs = d.broadcast_check_table(mine)
    # broadcast from a non-zero source goes through rank 0
    from_last = d.broadcast_array(np.full(3, 7.5) if rank == world - 1 else None, (3,), src=world - 1)
    d.barrier()
    mx = d.max_over_ranks(10.0 + rank)
    sm = d.sum_over_ranks(float(rank + 1))
    t0 = time.perf_counter()
    for _ in range(200):
        d.max_over_ranks(1.0)
    rtt = (time.perf_counter() - t0) / 200
    d.barrier()
    lo, hi = d.frame_range(65536)
    off, cones, poses = pkg.synth.make_config4_shard(lo, lo + 4, 100, 0.1, seed=7)  # the first 4 frames of this rank's shard
    off2, cones2, _ = pkg.synth.make_replay_batch(8, 16, 0.1, seed=d.shard_seed(5))
    q.put((rank, bool(np.array_equal(got, table)), same, differs, from_last.tolist(), mx, sm, (lo, hi), float(cones[:, :2].sum()), cones.shape,
           float(cones2[:, :2].sum()), "torch" in sys.modules, rtt))
    d.close()


@pytest.mark.parametrize("world", [2, 3])
def test_collectives_and_sharding_with_real_processes(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, f"test-{port}", q)) for r in range(world)]
    for p in procs[::-1]:  # clients first: they must retry until rank 0 listens
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and r[2] for r in res), "table broadcast / check"
    assert not any(r[3] for r in res), "a one-bit difference on one rank must fail the check everywhere"
    assert all(r[4] == [7.5, 7.5, 7.5] for r in res)
    assert all(r[5] == 10.0 + world - 1 for r in res) and all(r[6] == world * (world + 1) / 2 for r in res)
    per = -(-65536 // world)
    assert [r[7] for r in res] == [(min(k * per, 65536), min((k + 1) * per, 65536)) for k in range(world)]
    assert res[-1][7][1] == 65536 and sum(r[7][1] - r[7][0] for r in res) == 65536
    assert len({r[8] for r in res}) == world and len({r[10] for r in res}) == world  # different shards / tracks
    assert all(r[9] == (800, 3) for r in res)
    assert not any(r[11] for r in res), "dist.py must not import torch"
    assert all(r[12] < 0.05 for r in res), "a scalar all-reduce over the star takes well under 50 ms"


def test_stranger_and_other_launch_on_the_port_span_are_skipped():
    """A listener that is not rank 0 of this launch (garbage answers), and a rank 0 of ANOTHER launch (answers NO), sit on
    the first ports of the span: the client must find its own rank 0 behind them."""
    pkg = importlib.import_module("ft-fsd-path-planning_amd")
    import threading

    port = _free_port()
    stranger = socket.socket()
    stranger.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    stranger.bind(("127.0.0.1", port + 1))
    stranger.listen(4)

    def babble():
        while True:
            try:
                c, _ = stranger.accept()
                c.recv(64)
                c.sendall(b"??")
                c.close()
            except OSError:
                return

    threading.Thread(target=babble, daemon=True).start()
    stars, errors = {}, {}

    def rank0(key, name, timeout):
        os.environ["FSDP_LAUNCH_KEY"] = key  # (read at construction)
        try:
            stars[name] = pkg.dist.Star(0, 2, "127.0.0.1", port, connect_timeout=timeout)
        except Exception as e:  # launch B's rank 0 never sees its rank 1: kept for the assertion below, not left to the thread
            errors[name] = e

    os.environ["FSDP_LAUNCH_KEY"] = "launch-B"
    tb = threading.Thread(target=rank0, args=("launch-B", "B", 6.0), daemon=True)
    tb.start()
    time.sleep(0.5)  # B's rank 0 holds port + 2
    ta = threading.Thread(target=rank0, args=("launch-A", "A", 30.0), daemon=True)
    ta.start()
    time.sleep(0.5)  # A's rank 0 holds port + 3
    os.environ["FSDP_LAUNCH_KEY"] = "launch-A"
    client = pkg.dist.Star(1, 2, "127.0.0.1", port, connect_timeout=30)
    ta.join(timeout=30)
    assert "A" in stars and not ta.is_alive()
    got = []
    th = threading.Thread(target=lambda: got.append(stars["A"].allreduce(np.array([1.0, 5.0]), 1)), daemon=True)
    th.start()
    mine = client.allreduce(np.array([3.0, 2.0]), 1)
    th.join(timeout=10)
    assert mine.tolist() == [3.0, 5.0] and got[0].tolist() == [3.0, 5.0]
    # launch B never gets its rank 1: a clear timeout, not a hang — and the helper thread is joined, not abandoned
    tb.join(timeout=20)
    assert not tb.is_alive() and "B" not in stars
    assert isinstance(errors.get("B"), TimeoutError) and "only ranks" in str(errors["B"])
    client.close()
    stars["A"].close()
    stranger.close()
    del os.environ["FSDP_LAUNCH_KEY"]


def test_missing_rank_times_out_with_a_clear_message():
    pkg = importlib.import_module("ft-fsd-path-planning_amd")
    port = _free_port()
    with pytest.raises(TimeoutError, match="rank 0 not found"):
        pkg.dist.Star(1, 2, "127.0.0.1", port, connect_timeout=1.0)
    with pytest.raises(TimeoutError, match="only ranks"):
        pkg.dist.Star(0, 2, "127.0.0.1", port, connect_timeout=1.0)


def test_launcherless_start_spawns_the_ranks(tmp_path):
    """dist.spawn_ranks (bench.py --gpus N without a launcher): N processes with the launch contract's environment that
    find each other and run the collectives; rank 0's stdout passes through."""
    script = tmp_path / "ranks.py"
    script.write_text(textwrap.dedent(f"""
        import importlib, json, os, sys
        sys.path.insert(0, {str(ROOT)!r})
        pkg = importlib.import_module("ft-fsd-path-planning_amd")
        d = pkg.dist.Dist()
        tot = d.sum_over_ranks(float(d.rank + 1))
        d.barrier()
        if d.rank == 0:
            print(json.dumps({{"world": d.world, "sum": tot, "transport": d.transport, "spawned": os.environ.get("FSDP_SPAWNED")}}))
        d.close()
    """))
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(f"""
        import importlib, sys
        sys.path.insert(0, {str(ROOT)!r})
        pkg = importlib.import_module("ft-fsd-path-planning_amd")
        sys.exit(pkg.dist.spawn_ranks([{str(script)!r}], 3, timeout=120))
    """)], capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line == {"world": 3, "sum": 6.0, "transport": "tcp-fallback", "spawned": "1"}


def test_config4_union_does_not_depend_on_rank_count():
    pkg = importlib.import_module("ft-fsd-path-planning_amd")
    whole = pkg.synth.make_config4_shard(0, 12, 100, 0.1, seed=7)
    parts = [pkg.synth.make_config4_shard(*pkg.dist.frame_range(r, 3, 12), 100, 0.1, seed=7) for r in range(3)]
    assert np.array_equal(np.concatenate([p[1] for p in parts]), whole[1])
    assert np.array_equal(np.concatenate([p[2] for p in parts]), whole[2])


def test_a_multi_rank_line_without_rccl_is_an_error_unless_accepted():
    """bench.py's exit status for N > 1 (dist.multi_rank_exit_status; round-5 review item 8): a scaling curve must not quietly be
    made of ranks that met over the TCP star — status 3 unless RCCL carried the start-up collectives with every rank in the
    communicator (`rccl_ranks` of the line = ncclCommCount before the release) or the caller passed --allow-tcp-fallback."""
    pkg = importlib.import_module("ft-fsd-path-planning_amd")
    st = pkg.dist.multi_rank_exit_status
    assert st(1, False, "none", 0, False) == 0                 # one rank: nothing to claim
    assert st(8, False, "rccl", 8, False) == 0                 # the real thing
    assert st(8, False, "tcp-fallback", 0, False) == 3         # RCCL did not come up
    assert st(8, False, "rccl", 4, False) == 3                 # a communicator that does not hold every rank
    assert st(8, False, "tcp-fallback", 0, True) == 0          # accepted explicitly
    assert st(8, True, "in-process", 0, False) == 0            # --single-process: one process by design, no collective at all
    src = (ROOT / "bench.py").read_text()
    assert "--allow-tcp-fallback" in src and "multi_rank_exit_status" in src and '"rccl_ranks"' in src


def _fallback_worker(rank, world, port, key, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      FSDP_LAUNCH_KEY=key)
    pkg = importlib.import_module("ft-fsd-path-planning_amd")
    d = pkg.dist.Dist()  # no GPU here: the TCP star is the only transport, exactly the state of a rank whose RCCL bootstrap failed
    q.put((rank, d.transport, d.rccl_ranks, pkg.dist.multi_rank_exit_status(d.world, False, d.transport, d.rccl_ranks, False),
           pkg.dist.multi_rank_exit_status(d.world, False, d.transport, d.rccl_ranks, True)))
    d.barrier()
    d.close()


def test_ranks_on_the_tcp_star_report_zero_rccl_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, 2, port, f"test-{port}", q)) for r in range(2)]
    for p in procs[::-1]:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, "tcp-fallback", 0, 3, 0), (1, "tcp-fallback", 0, 3, 0)]
