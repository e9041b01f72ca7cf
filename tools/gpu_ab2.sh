#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
for rep in 1 2; do for v in "$@"; do tools/variants.sh run $v --minimal --steps 200 --warmup 5; done; done
for v in "$@"; do tools/variants.sh run $v --minimal --steps 100 --warmup 5 --scene repeatxy; done
