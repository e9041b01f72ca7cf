#!/bin/bash
# On the GPU box: the -m gpu suite in the default configuration (+ optionally extra test selections), logs under gpurun_out/$1
tag=${1:-suite}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/$tag; mkdir -p $O
timeout 1700 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
