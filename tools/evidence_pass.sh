#!/bin/bash
# evidence pass: everything profiles/ holds for the round, with the library build of this snapshot
set -u
R=gpurun_out/r02
mkdir -p $R
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $R/gpu_tests.txt
python __graft_entry__.py smoke >> $R/gpu_tests.txt 2>&1
bash tools/profile_gpu.sh r02 > $R/profile.log 2>&1
cp gpurun_out/prof_r02/pmc_traffic.json profiles/pmc_traffic.json   # so that the bench lines below carry traffic / insts
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $R/bench_line.json
python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > $R/bench_line_100steps.json
python bench.py --config 4 --frames 8192 --no-cpu-baseline 2>/dev/null | tail -1 > $R/bench_cfg4_shard.json
FSDP_FORCE_DIST=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $R/bench_line_rccl_1rank.json
python tools/bench_configs.py > $R/bench_configs.jsonl 2>&1
python tools/batch_sweep.py > $R/batch_sweep.jsonl 2>&1
python tools/overlap_depths.py default > $R/overlap_depths.txt 2>&1
python tools/latency_breakdown.py > $R/latency_breakdown.txt 2>&1
( echo "== fit_kernel<4> =="; python tools/section_profile.py; echo; echo "== path_prep_kernel<8> =="; python tools/section_profile.py --kernel=prep | tail -22; echo; echo "== path_finish_kernel<8> =="; python tools/section_profile.py --kernel=finish | tail -22; echo; echo "== sort_kernel, coloured =="; python tools/section_profile_sort.py ) > $R/kernel_sections.txt 2>&1
( python tools/bench_skidpad.py; FSDP_FORCE_DIST=1 python tools/bench_skidpad.py ) > $R/skidpad.jsonl 2>&1

cat $R/gpu_tests.txt; cut -c1-300 $R/bench_line.json
