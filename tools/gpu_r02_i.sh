#!/bin/bash
# round 2: lanes x depth after the sampler lost its 12 KB of LDS
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r02i; mkdir -p $O
run() { echo -n "$* : "; env "${@:4}" timeout 300 python3 bench.py --no-cpu --minimal --scene $1 --grid $2 --steps $3 --warmup 5 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for cfg in "sphere 512 200" "repeatxy 512 100" "union8 512 100" "sphere 1024 40" "sphere 768 80" "sphere 384 200"; do
  for ld in "2 3" "3 3" "3 4" "3 6"; do
    set -- $ld
    run $cfg SDFK_LANES=$1 SDFK_BENCH_DEPTH=$2
  done
done 2>&1 | tee $O/lanes.log
