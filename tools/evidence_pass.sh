#!/bin/bash
# evidence pass of round 5: everything profiles/ holds for the round, taken with the library build of this snapshot.
# Every command runs under `timeout` (a hung process would cost the box's whole limit).
set -u
R=gpurun_out/r05
mkdir -p $R
T="timeout 600"
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $R/gpu_tests.txt
$T python __graft_entry__.py smoke >> $R/gpu_tests.txt 2>&1
timeout 2400 bash tools/profile_gpu.sh r05 > $R/profile.log 2>&1
cp gpurun_out/prof_r05/pmc_traffic.json profiles/pmc_traffic.json   # so that the bench lines below carry traffic / insts
$T python bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $R/bench_line.json
$T python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 > $R/bench_line_100steps.json
$T python bench.py --config 4 --frames 8192 --no-cpu-baseline --no-latency 2>/dev/null | tail -1 > $R/bench_cfg4_shard.json
FSDP_FORCE_DIST=1 $T python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-latency --stream-batches 0 2>/dev/null | grep '^{' | tail -1 > $R/bench_line_torchrun_rccl_1rank.json
FSDP_SHARE_GPU=1 FSDP_RCCL_INIT_TIMEOUT=40 $T python bench.py --gpus 2 --steps 20 --warmup 3 --no-cpu-baseline --no-latency --stream-batches 0 2>$R/bench_2ranks.err | grep '^{' | tail -1 > $R/bench_line_2ranks_one_gpu_tcp_fallback.json
$T python tools/stream_probe.py > $R/streaming.jsonl 2>&1
( $T python tools/bench_skidpad.py 1024; $T python tools/bench_skidpad.py 4096 ) > $R/skidpad.jsonl 2>&1
# skidpad by instance count and by steps per group (csrc/skidpad_kernel.h "steps in flight")
( for n in 64 256 512 1024 2048 4096; do echo "instances $n"; $T python tools/bench_skidpad.py $n; done
  for g in 1 2 4 8 12 16; do echo "1024 instances, groups of $g steps"; FSDP_SKID_GROUP=$g $T python tools/bench_skidpad.py 1024; done
  for g in 1 2 3 4; do echo "1024 instances, a wavefront per (instance, step), groups of $g steps"; FSDP_SKID_PACK_MIN=100000000 FSDP_SKID_GROUP=$g $T python tools/bench_skidpad.py 1024; done
  echo "4096 instances, a wavefront per (instance, step)"; FSDP_SKID_PACK_MIN=100000000 $T python tools/bench_skidpad.py 4096 ) > $R/skidpad_groups.txt 2>&1
( export TMPDIR=/tmp; REPO=$(pwd); cd /tmp; FSDP_SKID_BENCH_LEGS=ahead timeout 600 rocprofv3 --kernel-trace --stats -d $REPO/gpurun_out/prof_r05/skid -o trace -- python $REPO/tools/bench_skidpad.py 1024 > $REPO/$R/skid_trace.log 2>&1 )
python tools/kernel_stats.py gpurun_out/prof_r05/skid "python tools/bench_skidpad.py 1024, the replay submitted ahead only" > $R/skidpad_rocprofv3_summary.txt 2>&1
$T python tools/bench_configs.py > $R/bench_configs.jsonl 2>&1
# one process, two contexts on the one GPU (multi.py; the form the driver's 8-GPU node can run without a launcher)
FSDP_SHARE_GPU=1 $T python bench.py --gpus 2 --single-process --steps 20 --warmup 3 --no-cpu-baseline --no-latency --stream-batches 0 2>$R/bench_single_process.err | grep '^{' | tail -1 > $R/bench_line_2contexts_one_process.json
$T python tools/ab_routes.py > $R/routes.txt 2>&1
python tools/kernel_resources.py > $R/kernel_resources.txt 2>&1
python tools/kernel_resources.py ft-fsd-path-planning_amd/lib/libfsdp_hip_wide.so > $R/kernel_resources_wide.txt 2>&1
$T python tools/wide_probe.py > $R/wide_build.jsonl 2>&1
$T tools/ubench/mfma_f64_order > $R/mfma_f64_order.txt 2>&1
$T python tools/batch_sweep.py 1024 2048 4096 8192 16384 32768 65536 98304 > $R/batch_sweep.jsonl 2>&1
FSDP_PACK=1 $T python tools/batch_sweep.py 2048 4096 8192 16384 32768 98304 > $R/batch_sweep_packed_kernels.jsonl 2>&1
$T python tools/latency_breakdown.py > $R/latency_breakdown.txt 2>&1
$T python tools/overlap_depths.py default > $R/overlap_depths.txt 2>&1
timeout 1500 python tests/fuzz_gpu_vs_oracle.py 2048 > $R/fuzz_gpu_vs_oracle.txt 2>&1
timeout 900 python tests/fuzz_skidpad_gpu_vs_oracle.py 192 90 > $R/fuzz_skidpad_gpu_vs_oracle.txt 2>&1
timeout 900 python tests/fuzz_gpu_vs_oracle_wide.py 1024 > $R/fuzz_gpu_vs_oracle_wide.txt 2>&1
cat $R/gpu_tests.txt; cut -c1-300 $R/bench_line.json; tail -3 $R/fuzz_gpu_vs_oracle.txt
