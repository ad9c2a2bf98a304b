#!/bin/bash
# Developer helper (GPU box): the per-round evidence set -> gpurun_out/round/  (copy what is to be judged into profiles/)
#   bench lines (config 2 with roofline + cpu_baseline + c4 / fp8 sub-records), rocprofv3 --kernel-trace --stats of the bench command with the
#   launch census (algorithmic bytes per launch), per-launch trace of one step with stream tags, HBM PMC passes (FETCH_SIZE, WRITE_SIZE in
#   separate runs, as the microarch guide prescribes) -> per-dispatch table + top_kernels.json + pmc_hbm.csv keyed by the build hash,
#   MFMA-busy PMC pass, per-layer tables of the channel-heavy convs and the big weight gradients.
OUT=$GRAFT_REPO_ROOT/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
MMVAE_LAUNCH_STATS=$OUT/census MMVAE_LAUNCH_SEQ=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_m -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_m.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_step.py $OUT/stats/s_kernel_trace.csv > $OUT/trace_step.txt
python tools/stream_busy.py $OUT/stats/s_kernel_trace.csv > $OUT/stream_busy.txt
python tools/pmc_per_dispatch.py $OUT/pmc_f/p_counter_collection.csv $OUT/pmc_w/p_counter_collection.csv > $OUT/pmc_per_dispatch.txt
python tools/pmc_hbm_csv.py $OUT/pmc_f/p_counter_collection.csv $OUT/pmc_w/p_counter_collection.csv 5120 > $OUT/pmc_hbm.csv
python tools/top_kernels.py $OUT/stats/s_kernel_trace.csv $(ls $OUT/census.*.seq | head -1) $OUT/pmc_per_dispatch.txt > $OUT/top_kernels.json
python tools/pmc_summary.py $OUT/pmc_m/p_counter_collection.csv > $OUT/pmc_mfma_busy.txt
# bench.py picks the `roofline` kernel from the newest profiles/*_top_kernels.json and quotes PMC traffic from profiles/*_pmc_hbm.csv of the same
# build: place this run's tables there (box-local copy; tools/collect_profiles.sh does the same in the build container) before the bench lines
R=${ROUND_TAG:-r04}
cp $OUT/top_kernels.json profiles/${R}_top_kernels.json
cp $OUT/pmc_hbm.csv profiles/${R}_pmc_hbm.csv
python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cut -c1-300 $OUT/bench_n1.json
python bench.py --config c4 --no-cpu-baseline > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --dtype fp8 --no-cpu-baseline > $OUT/bench_fp8.json 2> $OUT/bench_fp8.err
python tools/deep_probe.py > $OUT/deep_layers.txt 2>&1
python tools/wgrad_probe.py > $OUT/wgrad_layers.txt 2>&1
cat $OUT/pmc_hbm.csv
tail -2 $OUT/pmc_per_dispatch.txt
ls $OUT $OUT/stats
