#!/bin/bash
# vertex colours by re-evaluation (sdfk_vertex_colors) against the gather path: parity suites, then A/B on the colour scenes
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/vcol; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
run() { echo -n "$* : "; env "${@:2}" timeout 300 python3 bench.py --no-cpu --minimal --scene $1 --steps 100 --warmup 5 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], {k: v['avg_us'] for k, v in d['kernels_us'].items() if k in ('k_vertices','sdfk_vertex_colors','sdfk_corners_eval')})"; }
for sc in repeatxy union8; do
  run $sc SDFK_NO_VCOLOR_EVAL=1
  run $sc X=0
  run $sc SDFK_NO_VCOLOR_EVAL=1
  run $sc X=0
done
