#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r04g; mkdir -p $O
{
for v in $VARIANTS; do tools/variants.sh run $v --minimal --steps 200 --warmup 5; done
for v in $VARIANTS_XY; do tools/variants.sh run $v --minimal --steps 100 --warmup 5 --scene repeatxy; done
} > $O/ab.log 2>&1
grep -v "^  File\|^    \|Traceback\|json.decoder" $O/ab.log
if [ -n "$TESTS" ]; then timeout 1500 python3 -m pytest $TESTS -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -5 $O/pytest.log; fi
