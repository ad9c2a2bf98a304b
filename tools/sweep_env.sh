#!/bin/bash
# Developer helper (GPU box): per-launch traces of one step for several values of one environment variable.
# usage: tools/sweep_env.sh VAR v1 v2 ...   -> gpurun_out/sweep_<VAR>_<v>.txt
VAR=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export $VAR=$v
  rm -rf /tmp/sw_trace
  rocprofv3 --kernel-trace --output-format csv -d /tmp/sw_trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > /tmp/sw.log 2>&1 || { tail -5 /tmp/sw.log; exit 1; }
  python $GRAFT_REPO_ROOT/tools/trace_step.py /tmp/sw_trace/t_kernel_trace.csv > $GRAFT_REPO_ROOT/gpurun_out/sweep_${VAR}_$v.txt
  grep "^step:" $GRAFT_REPO_ROOT/gpurun_out/sweep_${VAR}_$v.txt
done
