#!/bin/bash
# GPU timeline of the pipelined single-GPU step (rocprofv3 --kernel-trace of tools/pipeline_trace.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
rm -rf gpurun_out/pipetrace
timeout -k 5 120 rocprofv3 --kernel-trace -d $R/gpurun_out/pipetrace -o t --output-format csv -- python3 tools/pipeline_trace.py > gpurun_out/pipetrace.log 2>&1
echo "rocprof rc $?"
ls gpurun_out/pipetrace 2>/dev/null
[ -f gpurun_out/pipetrace/t_kernel_trace.csv ] && timeout 60 python3 tools/slab_step_timeline.py gpurun_out/pipetrace/t_kernel_trace.csv 600 | head -60
