#!/bin/bash
# copies the small summaries of scripts/gpu_r3_final.sh from gpurun_out/ into profiles/ (tracked)
cd "$(dirname "$0")/.."
P=gpurun_out/prof_r03_C2; O=gpurun_out/r3
for f in r03_C2_rocprofv3_summary.txt r03_C2_kernel_stats.csv r03_C2_trials_gemm_durations.json r03_C2_traffic_trials_gemm.json r03_C2_pmc_trials_gemm.txt r03_C2_bench_under_trace.json; do cp $P/$f profiles/ 2>/dev/null; done
for f in r03_bench_c2.json r03_bench_c3.json r03_bench_c4_one_gpu.json r03_bench_2ranks_one_gpu_host_transport.json r03_bench_emulate8.json r03_transform_sweep.txt r03_gemm_sweep.txt r03_em_kernel_stats.txt; do cp $O/$f profiles/ 2>/dev/null; done
grep -v "^$" $O/pytest_gpu.txt | tail -14 > profiles/r03_pytest_gpu_tail.txt
ls -la profiles | grep r03
