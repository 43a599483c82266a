#!/bin/bash
# developer tool: copy the summaries tools/profile_r06.sh left under gpurun_out/r06 into profiles/ (tracked) under their round-6 names
set -e
cd "$(dirname "$0")/.."; s=gpurun_out/r06/set; d=profiles
last() { tail -n 1 "$1"; }
last $s/bench_cfg2_s20.json > $d/r06_bench_cfg2_steps20.json
last $s/bench_cfg2_s200.json > $d/r06_bench_cfg2_steps200.json
last $s/bench_cfg3.json > $d/r06_bench_cfg3.json
last $s/bench_cfg4_1gpu.json > $d/r06_bench_cfg4_1gpu.json
last $s/bench_cfg2_analytic.json > $d/r06_bench_cfg2_analytic.json
last $s/bench_cfg2_gloo2.json > $d/r06_bench_cfg2_2ranks_gloo_one_gpu.json
for c in cfg2 cfg3 cfg4; do last $s/bench_${c}_probing.json; done > $d/r06_bench_probing_kernels.jsonl
cp $s/sweep.jsonl $d/r06_sweep.jsonl
cp $s/trace_cfg2_kernel_stats.csv $d/r06_bench_cfg2_kernel_stats.csv
cp $s/trace_cfg3_kernel_stats.csv $d/r06_bench_cfg3_kernel_stats.csv
cp $s/pmc_summary.txt $d/r06_pmc_summary.txt
cp $s/resource_usage.txt $d/r06_resource_usage.txt
cp $s/frame_trace.txt $d/r06_frame_trace.txt
cp $s/next_rows.jsonl $d/r06_next_rows.jsonl
cp $s/process_frame_async_ab.jsonl $d/r06_process_frame_async_ab.jsonl
last $s/sequence_200.json > $d/r06_sequence_200.json
last $s/sequence_800.json > $d/r06_sequence_800.json
last $s/sequence_60_gloo2.json > $d/r06_sequence_60_2ranks_gloo.json
cp $s/wf0_iteration.json $d/r06_wf0_iteration.json
cp $s/sequence_track.jsonl $d/r06_sequence_track.jsonl
cp $s/track_eval_timing.jsonl $d/r06_track_eval_timing.jsonl
cp $s/sort_phase_timing.txt $d/r06_sort_phase_timing.txt
python tools/make_hbm_traffic.py $d/r06_pmc_summary.txt $d/r06_resource_usage.txt
ls -la $d | grep r06_
