#!/bin/bash
# SQ_INSTS_VALU of the sorting kernel when the stage stops after phase k (k = 1: staging + per-cone scalars, 2: + start
# cones and adjacency of both sides, 3: + the search, 4: + post filters / side counting / costs, product: + combine)
export TMPDIR=/tmp
R=$(pwd)
cd /tmp
for so in $R/ft-fsd-path-planning_amd/lib/variants/stop1.so $R/ft-fsd-path-planning_amd/lib/variants/stop2.so $R/ft-fsd-path-planning_amd/lib/variants/stop3.so $R/ft-fsd-path-planning_amd/lib/variants/stop4.so $R/ft-fsd-path-planning_amd/lib/libfsdp_hip.so; do
  d=$R/gpurun_out/sortstop/$(basename $so .so)
  rm -rf $d
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $d -o pmc -- python $R/tools/sort_phase_insts.py $so > /dev/null 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob("$d/**/*.db", recursive=True)
con = sqlite3.connect(db[0])
rows = list(con.execute("select kernel_name, counter_name, avg(value) from counters_collection where kernel_name like '%sort_kernel_128%' group by kernel_name, counter_name"))
print("$(basename $so)", {c: round(v / 4096, 1) for _, c, v in rows})
PY
done
