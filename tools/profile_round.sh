#!/bin/bash
# Developer helper (GPU box): the per-round evidence set -> gpurun_out/round/
#   bench line (with roofline + cpu_baseline), rocprofv3 --kernel-trace --stats of the bench command, per-launch trace of
#   one step, HBM PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs, as the microarch guide prescribes).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench_n1.json 2> $OUT/bench.err
cut -c1-400 $OUT/bench_n1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc_w.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_step.py $OUT/stats/s_kernel_trace.csv > $OUT/trace_step.txt
ls $OUT $OUT/stats
