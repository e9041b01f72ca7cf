#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the default bench; prints name, calls, avg ns.
# usage: tools/gpu_prof.sh <tag> [bench args...]
tag=${1:-p}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o $tag --output-format csv -- \
    python3 bench.py --steps 20 --warmup 3 --no-cpu "$@" > gpurun_out/bench_$tag.log 2>&1
cut -d, -f1,2,4 gpurun_out/prof_$tag/${tag}_kernel_stats.csv | cut -c1-110
grep "^{" gpurun_out/bench_$tag.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"ms_per_step\"], d[\"kernels_us\"])"
