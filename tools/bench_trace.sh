#!/bin/bash
# Developer helper (GPU box): bench line + per-launch kernel trace of one step -> gpurun_out/
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-roofline "$@" > gpurun_out/bench.json 2>gpurun_out/bench.err || { tail -5 gpurun_out/bench.err; exit 1; }
cut -c1-330 gpurun_out/bench.json
rm -rf gpurun_out/trace
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline "$@" > $GRAFT_REPO_ROOT/gpurun_out/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_step.py gpurun_out/trace/t_kernel_trace.csv > gpurun_out/trace_step.txt
sed -n '/^step:/,$p' gpurun_out/trace_step.txt | head -${TOPN:-30}
