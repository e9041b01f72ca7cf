#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
{
for v in head default head default; do tools/variants.sh run $v --minimal --steps 200 --warmup 5; done
for v in head default; do tools/variants.sh run $v --minimal --steps 100 --warmup 5 --scene repeatxy; done
for v in base head default; do tools/variants.sh run $v --minimal --steps 10 --warmup 3 --scene union8 --grid 1024; done
} > $O/ab.log 2>&1
cat $O/ab.log
timeout 1500 python3 -m pytest tests/test_gpu_param_programs.py tests/test_gpu_parity.py tests/test_golden_fixtures.py tests/test_reference_vectors.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
