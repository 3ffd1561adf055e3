#!/usr/bin/env python3
"""Per-kernel statistics of a rocprofv3 --kernel-trace --stats run (rocpd sqlite):  python tools/kernel_stats.py <dir> [title]"""
import glob, re, sqlite3, sys
files = glob.glob(f"{sys.argv[1]}/*.db") + glob.glob(f"{sys.argv[1]}/*/*.db")
con = sqlite3.connect(files[0])
short = lambda n: re.sub(r"<(\d+), \d+>", r"<\1>", n.split("(")[0].replace("void ", "").split("::")[-1].strip())
print(f"== rocprofv3 --kernel-trace --stats ({sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]}) ==")
print(f"{'kernel':<30}{'calls':>6}{'avg_us':>12}{'total_ms':>11}{'pct':>7}")
for name, calls, tot, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{short(name):<30}{calls:>6}{avg:>12.1f}{tot / 1e3:>11.2f}{pct:>7.2f}")
seen = set()
for r in con.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, workgroup_x, grid_x from kernels order by grid_x desc"):
    if r[0] not in seen:
        seen.add(r[0])
        print(f"{short(r[0]):<30} vgpr={r[1]} agpr={r[2]} sgpr={r[3]} lds={r[4]}B scratch={r[5]}B/lane wg={r[6]} grid={r[7]}")
t0, t1 = con.execute("select min(start), max(end) from kernels").fetchone()
busy = con.execute("select sum(duration) from kernels").fetchone()[0]
print(f"first kernel start to last kernel end: {(t1 - t0) / 1e6:.1f} ms, sum of kernel durations {busy / 1e6:.1f} ms")
# time with at least one kernel running (kernels of different streams may overlap)
iv = sorted(con.execute("select start, end from kernels").fetchall())
union, cur_s, cur_e = 0, None, None
for a, b in iv:
    if cur_e is None or a > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = a, b
    else:
        cur_e = max(cur_e, b)
union += (cur_e - cur_s) if cur_e else 0
print(f"time with at least one kernel running: {union / 1e6:.1f} ms")
