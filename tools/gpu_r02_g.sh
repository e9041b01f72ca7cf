#!/bin/bash
# round 2: rows-per-wavefront knob of the fused sampler for the colour scenes (store-bound at 76 % of HBM peak)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r02g; mkdir -p $O
run() { echo -n "$* : "; env "${@:2}" SDFK_NO_CACHE=1 timeout 300 python3 bench.py --no-cpu --minimal --scene $1 --steps 20 --warmup 5 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k: v['avg_us'] for k, v in d['kernels_us'].items() if k.startswith('sdfk_sample')})"; }
for sc in repeatxy union8 sphere; do
  for rpw in 1 2 4; do run $sc SDFK_SAMPLE_RPW=$rpw; done
  run $sc SDFK_SAMPLE_NT=0
done 2>&1 | tee $O/rpw.log
