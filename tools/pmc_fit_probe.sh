#!/bin/bash
# PMC break-down of the kernels alone on a chip they fill (FSDP_PACK=1 tools/batch_sweep.py 98304): where do wave cycles go?
# (--pmc passes with --kernel-trace only; never combined with sys / hip / hsa tracing)
R=$(pwd); O=$R/gpurun_out/prof_r06/pmc_fit; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
CMD="python $R/tools/batch_sweep.py 98304"
FSDP_PACK=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/a -o pmc -- $CMD > $O/a.log 2>&1
FSDP_PACK=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $O/b -o pmc -- $CMD > $O/b.log 2>&1
FSDP_PACK=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC -d $O/c -o pmc -- $CMD > $O/c.log 2>&1
cd $R
python - <<'PY'
import sqlite3, glob, collections, re
out = collections.defaultdict(dict)
for d in "abc":
    for f in glob.glob(f"gpurun_out/prof_r06/pmc_fit/{d}/**/*.db", recursive=True):
        con = sqlite3.connect(f)
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
        pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]; info = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
        kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
        q = f"select s.kernel_name, i.name, sum(e.value), count(distinct k.id) from {pmc} e join {info} i on e.pmc_id=i.id join {kd} k on e.event_id=k.event_id join {ks} s on k.kernel_id=s.id group by 1,2"
        for name, ctr, val, n in con.execute(q):
            short = re.sub(r"\(.*", "", name).replace("void fsdp::", "").replace("fsdp::", "")
            out[short][ctr] = val / max(n, 1)
for k, v in out.items():
    if "WAVE_CYCLES" in "".join(v): pass
    print(k, {c: round(x) for c, x in sorted(v.items())})
PY
