#!/bin/bash
# Developer helper (GPU box): MFMA busy cycles per kernel ("per-layer MFMA-util for the MFMA-bound layers").
# One --pmc pass with --kernel-trace only; summary -> gpurun_out/mfma/mfma_util.txt
OUT=$GRAFT_REPO_ROOT/gpurun_out/mfma
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $OUT/pmc/p_counter_collection.csv > $OUT/mfma_util.txt
head -40 $OUT/mfma_util.txt
