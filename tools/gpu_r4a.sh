#!/bin/bash
# round 4: A/B of library variants against the round-3 library, then the parity suites
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r04a; mkdir -p $O
{
for v in $VARIANTS; do tools/variants.sh run $v --minimal --steps 200 --warmup 5; done
for v in $VARIANTS_XY; do tools/variants.sh run $v --minimal --steps 100 --warmup 5 --scene repeatxy; done
} > $O/ab.log 2>&1
cat $O/ab.log
timeout 1500 python3 -m pytest $TESTS -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
