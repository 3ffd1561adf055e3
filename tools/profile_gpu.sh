#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes for bench.py, plus the FETCH_SIZE /
# WRITE_SIZE calibration kernels.  Usage: tools/profile_gpu.sh <tag>   -> gpurun_out/prof_<tag>/...
# (PMC passes never combine with sys/hip/hsa tracing: --kernel-trace only.)
set -u
TAG=${1:-r03}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# the profiled command = the bench's timed region (passes overlapped) followed by its serial reference leg; the single-frame
# latency loop and the CPU baseline are switched off so that per-dispatch averages are those of the 4096-frame launches
BENCH="python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-latency --stream-batches 0"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
echo "trace rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace_serial -o trace -- $BENCH --no-overlap > $OUT/trace_serial.log 2>&1
echo "trace_serial rc=$?"
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?"
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
echo "write rc=$?"
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1
echo "sq rc=$?"
# counter calibration: kernels that move exactly 1 GiB each with 4 / 8 / 16 bytes per lane (a build output, not in the history:
# compiled here when the snapshot came without it)
[ -x $REPO/tools/ubench/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off $REPO/tools/ubench/fetch_calib.hip -o $REPO/tools/ubench/fetch_calib > $OUT/calib_build.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/calib_fetch -o pmc -- $REPO/tools/ubench/fetch_calib > $OUT/calib_fetch.log 2>&1
echo "calib fetch rc=$?"
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/calib_write -o pmc -- $REPO/tools/ubench/fetch_calib > $OUT/calib_write.log 2>&1
echo "calib write rc=$?"
cd $REPO
python tools/summarize_prof.py $OUT $OUT/pmc_traffic.json > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
