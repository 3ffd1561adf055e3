#!/usr/bin/env python3
"""Condense rocprofv3 (rocpd sqlite) outputs — kernel stats + separate PMC passes — into a short text
summary that is committed under profiles/.  Usage: summarize_prof.py <gpurun_out/prof_TAG>"""
import glob
import json
import sqlite3
import sys

out = sys.argv[1]
json_out = sys.argv[2] if len(sys.argv) > 2 else None
traffic = {}
FRAMES = 4096
ALGO_BYTES = 4488  # SURVEY.md 8d, N = 128


def db(sub):
    files = glob.glob(f"{out}/{sub}/*.db")
    return sqlite3.connect(files[0]) if files else None


for sub, label in (("trace", "python bench.py --steps 5 --warmup 1 --no-cpu-baseline  [passes overlap 4 deep, as the bench runs]"),
                   ("trace_serial", "same command with --no-overlap  [one pass after the other]")):
  con = db(sub)
  print(f"== rocprofv3 --kernel-trace --stats ({label}) ==")
  if con:
      print(f"{'kernel':<28}{'calls':>6}{'avg_us':>12}{'total_ms':>11}{'pct':>7}")
      for name, calls, tot, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
          print(f"{name.split('(')[0]:<28}{calls:>6}{avg:>12.1f}{tot / 1e3:>11.2f}{pct:>7.2f}")
      print("-- per-kernel resources --")
      seen = set()
      for r in con.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x, grid_x, duration from kernels"):
          if r[0] in seen:
              continue
          seen.add(r[0])
          print(f"{r[0].split('(')[0]:<28} vgpr={r[1]} agpr={r[2]} sgpr={r[3]} lds={r[4]}B scratch={r[5]}B/lane wg={r[6]} grid={r[7]}")
      for name, avg in con.execute("select name, avg(duration) from kernels group by name"):
          if "fsdp::" in name and "default" not in name:
              gbs = ALGO_BYTES * FRAMES / (avg * 1e-9) / 1e9
              print(f"roofline[{name.split('(')[0]}]: avg {avg / 1e3:.1f} us -> algorithmic {gbs:.3f} GB/s = {gbs / 8000:.2e} of 8 TB/s HBM peak")

for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    con = db(sub)
    if not con:
        continue
    print(f"== rocprofv3 --pmc pass: {sub} (per-dispatch averages) ==")
    q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"
    for k, c, v, n in con.execute(q):
        if "fsdp::" not in k or "default" in k:
            continue
        extra = ""
        if c == "FETCH_SIZE":
            extra = f"  (KB; x2 gfx950 correction for wide coalesced reads -> <= {2 * v / 1024:.2f} MB/launch; algorithmic read {FRAMES * (128 * 24 + 32) / 1e6:.2f} MB)"
        if c == "WRITE_SIZE":
            extra = f"  (KB -> {v / 1024:.2f} MB/launch; algorithmic write {FRAMES * 1384 / 1e6:.2f} MB)"
        print(f"{k.split('(')[0]:<28}{c:<22}{v:>16.1f}  n={n}{extra}")
        if c in ("FETCH_SIZE", "WRITE_SIZE"):
            short = k.split("(")[0].replace("void ", "").split("::")[-1]
            traffic.setdefault(short, {})[c + "_KB"] = round(v, 1)

if json_out and traffic:
    for k, d in traffic.items():
        # HBM bytes per launch: FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md) + WRITE_SIZE
        d["hbm_bytes_per_launch"] = int(1024 * (2 * d.get("FETCH_SIZE_KB", 0.0) + d.get("WRITE_SIZE_KB", 0.0)))
    doc = {"source": f"{out} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per-dispatch averages; "
                     "tools/profile_gpu.sh)", **traffic}
    json.dump(doc, open(json_out, "w"), indent=1)
