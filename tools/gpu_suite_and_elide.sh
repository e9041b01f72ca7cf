#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r04e; mkdir -p $O
timeout 1700 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
SDFK_ELIDE_VOLUME=1 timeout 1700 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_alt_elide.log 2>&1; echo "pytest rc $?" >> $O/pytest_alt_elide.log
tail -5 $O/pytest_alt_elide.log
