#!/bin/bash
# the sharded step at world 1 over RCCL: timings per configuration
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
run() { echo "== $*"; env "$@" PROBE_STEPS=400 timeout 120 python3 tools/slab_step_trace.py 2>&1 | grep "us per step"; }
run PROBE_N=128
run PROBE_N=128 PROBE_TORCH=1
run PROBE_N=128 PROBE_TORCH=1 PROBE_BIND=1
run PROBE_N=128 PROBE_TORCH=1 PROBE_BIND=1 GPU_MAX_HW_QUEUES=4
run PROBE_N=128 PROBE_DEPTH=5
run PROBE_N=512
run PROBE_N=512 PROBE_TORCH=1 PROBE_BIND=1
run PROBE_N=512 PROBE_LANES=0
