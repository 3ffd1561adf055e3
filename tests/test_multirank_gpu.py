"""More than one rank on the GPU box.  The box has ONE GPU, so the ranks share it: RCCL refuses (or cannot finish) a
communicator with two ranks on one device, which is exactly the situation dist.py's agreement step is for — every rank must
end up on the same transport (normally the TCP star), the collectives must still give the right answers and every rank's
planner must work next to the others'.  On a real multi-GPU node the same code path keeps RCCL."""
import importlib
import json
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_two_ranks_agree_on_a_transport_and_plan(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent(f"""
        import importlib, json, os, sys
        import numpy as np
        sys.path.insert(0, {str(ROOT)!r})
        pkg = importlib.import_module("ft-fsd-path-planning_amd")
        ctx = pkg.Context(device=0)
        d = pkg.dist.Dist(ctx)
        ok = d.broadcast_check_table(ctx.default_path())
        lo, hi = d.frame_range(512)
        off, cones, poses = pkg.synth.make_config4_shard(lo, hi, 100, 0.1, seed=7)
        res = ctx.plan_batch(off, cones, poses)
        d.barrier()
        frames = d.sum_over_ranks(float(len(res)))
        good = d.sum_over_ranks(float((res["status"] == 0).sum()))
        code = d.max_over_ranks({{"none": 0.0, "rccl": 1.0, "tcp-fallback": 2.0}}[d.transport])
        lowest = -d.max_over_ranks(-{{"none": 0.0, "rccl": 1.0, "tcp-fallback": 2.0}}[d.transport])
        if d.rank == 0:
            print(json.dumps({{"world": d.world, "table_ok": bool(ok), "frames": frames, "good": good, "transport": d.transport,
                              "same_transport_everywhere": code == lowest, "why": d.fallback_reason, "torch": "torch" in sys.modules}}))
        d.close()
    """))
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(f"""
        import importlib, sys
        sys.path.insert(0, {str(ROOT)!r})
        pkg = importlib.import_module("ft-fsd-path-planning_amd")
        sys.exit(pkg.dist.spawn_ranks([{str(script)!r}], 2, env_extra={{"FSDP_RCCL_INIT_TIMEOUT": "60"}}, timeout=400))
    """)], capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["world"] == 2 and line["table_ok"] and line["frames"] == 512.0 and line["good"] >= 500 and line["same_transport_everywhere"]
    assert line["transport"] in ("rccl", "tcp-fallback") and not line["torch"]
