#!/bin/bash
# Developer helper (GPU box): HBM PMC passes of one train step (FETCH_SIZE and WRITE_SIZE in separate runs, as the microarch guide prescribes)
# -> gpurun_out/pmc/{f,w}/p_counter_collection.csv and the per-dispatch table gpurun_out/pmc/per_dispatch.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $OUT/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $OUT/w.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_per_dispatch.py $OUT/f/p_counter_collection.csv $OUT/w/p_counter_collection.csv > $OUT/per_dispatch.txt
tail -3 $OUT/per_dispatch.txt
