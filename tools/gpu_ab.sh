#!/bin/bash
# A/B of library variants on the GPU box: tools/gpu_ab.sh NAME... (see tools/variants.sh); "default" = the in-tree library
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
for v in "$@"; do tools/variants.sh run $v --minimal --steps 200 --warmup 5; done
