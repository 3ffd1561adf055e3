#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for bench.py.
# Usage: tools/profile_gpu.sh <tag>   -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline"
# kernel trace of the bench command itself (passes overlap as in the bench), then one with --no-overlap
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
echo "trace rc=$?"
rocprofv3 --kernel-trace --stats -d $OUT/trace_serial -o trace -- $BENCH --no-overlap > $OUT/trace_serial.log 2>&1
# (the PMC passes run the bench command as is: its timed region launches path_kernel<8>, its serial reference leg path_kernel<16>)
echo "trace rc=$?"
# PMC passes (separate runs; FETCH_SIZE and WRITE_SIZE do not fit one pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
echo "write rc=$?"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
echo "sq rc=$?"
cd $REPO
find $OUT -name "*.csv" | head -40
python tools/summarize_prof.py $OUT $OUT/pmc_traffic.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
