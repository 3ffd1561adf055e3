#!/usr/bin/env python3
"""Kernel timeline of one bench run from a rocprofv3 --kernel-trace database (rocpd sqlite): per dispatch the stream /
queue, start and end relative to the first dispatch of the timed region, and the number of kernels running at every
start.  Run on the GPU box:
  rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency
  python tools/timeline.py gpurun_out/tl [first_n]"""
import glob
import re
import sqlite3
import sys

out = sys.argv[1]
con = sqlite3.connect(glob.glob(f"{out}/**/*.db", recursive=True)[0])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
want = [c for c in ("name", "start", "end", "queue_id", "stream_id", "grid_x") if c in cols]
rows = list(con.execute(f"select {','.join(want)} from kernels order by start"))
idx = {c: i for i, c in enumerate(want)}


def short(name):
    s = name.split("(")[0].replace("void ", "").split("::")[-1].strip()
    return re.sub(r"<(\d+), \d+>", r"<\1>", s)


rows = [r for r in rows if "fsdp::" in r[idx["name"]]]
# the timed region = the last run of 20 passes before the serial reference leg: take dispatches of 4096-frame launches
t0 = rows[0][idx["start"]]
print("columns:", want)
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)]:
    q = r[idx["stream_id"]] if "stream_id" in idx else r[idx["queue_id"]]
    print(f"{short(r[idx['name']]):<26} q={q!s:<6} {1e-3 * (r[idx['start']] - t0):10.1f} {1e-3 * (r[idx['end']] - t0):10.1f}  {1e-3 * (r[idx['end']] - r[idx['start']]):8.1f} us")
