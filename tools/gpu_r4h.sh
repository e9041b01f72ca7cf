#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r04h; mkdir -p $O
timeout 1700 python3 -m pytest tests/test_gpu_elide_volume.py -m gpu -x -q > $O/pytest_elide.log 2>&1; echo "pytest rc $?" >> $O/pytest_elide.log
tail -12 $O/pytest_elide.log
timeout 600 python3 tools/elide_kernels_probe.py 2>/dev/null | tee $O/elide_kernels.txt
