#!/bin/bash
# copies what tools/evidence_pass.sh left under gpurun_out/ into profiles/ under this round's names
R=gpurun_out/r06
for f in bench_line.json bench_line_100steps.json bench_cfg4_shard.json bench_line_torchrun_rccl_1rank.json bench_line_2ranks_one_gpu_tcp_fallback.json bench_line_2contexts_one_process.json bench_2ranks_exit_status.txt bench_configs.jsonl batch_sweep.jsonl batch_sweep_packed_kernels.jsonl latency_breakdown.txt overlap_depths.txt streaming.jsonl skidpad.jsonl routes.txt gpu_tests.txt kernel_resources.txt kernel_resources_wide.txt chip_time_rocprofv3_summary.txt fuzz_gpu_vs_oracle.txt fuzz_gpu_vs_oracle_wide.txt fuzz_skidpad_gpu_vs_oracle.txt; do [ -f $R/$f ] && cp $R/$f profiles/r06_$f; done
# (r06_wide_build.jsonl carries a hand-written note at its end: refreshed by hand)
cp gpurun_out/prof_r06/summary.txt profiles/r06_rocprofv3_summary.txt
cp gpurun_out/prof_r06/pmc_traffic.json profiles/pmc_traffic.json
