#!/bin/bash
# Developer helper (build container): copy the evidence set tools/profile_round.sh left in gpurun_out/round/ into profiles/<round>_*
# usage: tools/collect_profiles.sh r03
R=${1:?round tag}
S=gpurun_out/round
D=profiles
cp $S/bench_n1.json $D/${R}_bench_n1.json
cp $S/bench_c4.json $D/${R}_bench_c4.json
cp $S/bench_fp8.json $D/${R}_bench_fp8.json
cp $S/stats/s_kernel_stats.csv $D/${R}_kernel_stats_c2_bf16_N5120.csv
cp $S/trace_step.txt $D/${R}_trace_one_step.txt
cp $S/stream_busy.txt $D/${R}_stream_busy.txt
cp $S/pmc_per_dispatch.txt $D/${R}_pmc_per_dispatch.txt
cp $S/pmc_hbm.csv $D/${R}_pmc_hbm.csv
cp $S/top_kernels.json $D/${R}_top_kernels.json
cp $S/pmc_mfma_busy.txt $D/${R}_pmc_mfma_busy.txt
cp $S/deep_layers.txt $D/${R}_deep_layers.txt
cp $S/wgrad_layers.txt $D/${R}_wgrad_layers.txt
cat $(ls $S/census.* | grep -v seq | head -1) > $D/${R}_launch_census.txt
ls -la $D/${R}_*
