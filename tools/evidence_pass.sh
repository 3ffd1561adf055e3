#!/bin/bash
# evidence pass of round 6: everything profiles/r06_* holds, taken with the library build of this snapshot.
# Every command runs under `timeout` (a hung process would cost the box's whole limit).  Usage (on the GPU box, via gpurun):
#   bash tools/evidence_pass.sh [part ...]      parts: tests profile bench streams routes fuzz   (default: all)
set -u
R=gpurun_out/r06
mkdir -p $R
T="timeout 600"
PARTS=${*:-tests profile bench streams routes fuzz}
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -5 > $R/gpu_tests.txt
  $T python __graft_entry__.py smoke >> $R/gpu_tests.txt 2>&1
  python tools/kernel_resources.py > $R/kernel_resources.txt 2>&1
  python tools/kernel_resources.py ft-fsd-path-planning_amd/lib/libfsdp_hip_wide.so > $R/kernel_resources_wide.txt 2>&1
fi
if has profile; then
  timeout 2400 bash tools/profile_gpu.sh r06 > $R/profile.log 2>&1
  cp gpurun_out/prof_r06/pmc_traffic.json profiles/pmc_traffic.json   # so that the bench lines below carry traffic / insts
  # the chip-time command of the bench line (roofline.frac): the kernels alone on a chip they fill, as a kernel trace
  ( export TMPDIR=/tmp; REPO=$(pwd); cd /tmp; FSDP_PACK=1 timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_r06/chip -o trace -- python $REPO/tools/batch_sweep.py 98304 > $REPO/$R/chip_trace.log 2>&1 )
  python tools/kernel_stats.py gpurun_out/prof_r06/chip "FSDP_PACK=1 python tools/batch_sweep.py 98304 (one pass at a time over a resident 98 304-frame batch: 12 launches per kernel)" > $R/chip_time_rocprofv3_summary.txt 2>&1
fi
if has bench; then
  $T python bench.py --steps 20 --warmup 3 2>$R/bench.err | tail -1 > $R/bench_line.json
  $T python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 > $R/bench_line_100steps.json
  $T python bench.py --config 4 --frames 8192 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 > $R/bench_cfg4_shard.json
  FSDP_FORCE_DIST=1 $T python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-latency --stream-batches 0 2>/dev/null | grep '^{' | tail -1 > $R/bench_line_torchrun_rccl_1rank.json
  # two ranks on the one GPU: RCCL cannot come up, the ranks agree on the TCP star.  Without --allow-tcp-fallback that is exit status 3
  FSDP_SHARE_GPU=1 FSDP_RCCL_INIT_TIMEOUT=40 $T python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-latency --stream-batches 0 > $R/bench_2ranks_refused.out 2>$R/bench_2ranks_refused.err
  echo "exit status without --allow-tcp-fallback: $?" > $R/bench_2ranks_exit_status.txt
  FSDP_SHARE_GPU=1 FSDP_RCCL_INIT_TIMEOUT=40 $T python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-latency --stream-batches 0 --allow-tcp-fallback 2>$R/bench_2ranks.err | grep '^{' | tail -1 > $R/bench_line_2ranks_one_gpu_tcp_fallback.json
  echo "exit status with --allow-tcp-fallback: ${PIPESTATUS[0]}" >> $R/bench_2ranks_exit_status.txt
  FSDP_SHARE_GPU=1 $T python bench.py --gpus 2 --single-process --steps 20 --warmup 3 --no-cpu-baseline --no-latency --stream-batches 0 2>$R/bench_single_process.err | grep '^{' | tail -1 > $R/bench_line_2contexts_one_process.json
  $T python tools/bench_configs.py > $R/bench_configs.jsonl 2>&1
  $T python tools/batch_sweep.py 1024 2048 4096 8192 16384 32768 65536 98304 > $R/batch_sweep.jsonl 2>&1
  FSDP_PACK=1 $T python tools/batch_sweep.py 2048 4096 8192 16384 32768 98304 > $R/batch_sweep_packed_kernels.jsonl 2>&1
  $T python tools/latency_breakdown.py > $R/latency_breakdown.txt 2>&1
  $T python tools/overlap_depths.py default > $R/overlap_depths.txt 2>&1
fi
if has streams; then
  $T python tools/stream_probe.py > $R/streaming.jsonl 2>&1
  ( $T python tools/bench_skidpad.py 1024; $T python tools/bench_skidpad.py 4096 ) > $R/skidpad.jsonl 2>&1
fi
if has routes; then
  $T python tools/ab_routes.py > $R/routes.txt 2>&1
  $T python tools/wide_probe.py > $R/wide_build.jsonl 2>&1
fi
if has fuzz; then
  timeout 1500 python tests/fuzz_gpu_vs_oracle.py 2048 > $R/fuzz_gpu_vs_oracle.txt 2>&1
  timeout 900 python tests/fuzz_skidpad_gpu_vs_oracle.py 192 90 > $R/fuzz_skidpad_gpu_vs_oracle.txt 2>&1
  timeout 900 python tests/fuzz_gpu_vs_oracle_wide.py 1024 > $R/fuzz_gpu_vs_oracle_wide.txt 2>&1
fi
for f in $R/gpu_tests.txt $R/bench_2ranks_exit_status.txt; do [ -f $f ] && cat $f; done
[ -f $R/bench_line.json ] && cut -c1-300 $R/bench_line.json
[ -f $R/fuzz_gpu_vs_oracle.txt ] && tail -2 $R/fuzz_gpu_vs_oracle.txt
exit 0
