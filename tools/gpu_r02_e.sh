#!/bin/bash
# round 2: captured launch graphs behind sdfk_sample_march -- the new tests, A/B on launch-bound grids, then the full suite
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r02e; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_graphs.py -x -q > $O/pytest_graphs.log 2>&1; echo "pytest graphs rc $?" >> $O/pytest_graphs.log
tail -15 $O/pytest_graphs.log
run() { echo -n "$* : "; env "${@:2}" timeout 300 python3 bench.py --no-cpu --minimal --grid $1 --steps 400 --warmup 50 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for n in 96 192 256; do
  for rep in 1 2; do
    run $n SDFK_GRAPHS=0
    run $n SDFK_GRAPHS=1
    run $n SDFK_GRAPHS=2
  done
done 2>&1 | tee $O/ab.log
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
