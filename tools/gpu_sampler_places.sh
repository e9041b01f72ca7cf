#!/bin/bash
# On the GPU box: the pipelined step and the sampler alone with the sampling kernels' occupancy capped (SDFK_SAMPLE_VREG: the
# kernel allocates N + 1 vector registers, at most 512 / (N + 1) wavefronts per SIMD).  usage: tools/gpu_sampler_places.sh [scene]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
scene=${1:-sphere}
O=gpurun_out/places; mkdir -p $O
for v in none 71 79 95 103 127; do
  flags=""; [ "$v" != none ] && flags="-DSDFK_SAMPLE_VREG=$v"
  for rep in 1 2; do
  echo -n "vreg $v  "
  SDFK_JIT_FLAGS="$flags" timeout 300 python3 bench.py --no-cpu --minimal --steps 200 --warmup 5 --scene $scene 2>/dev/null | grep "^{" | \
     python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pipelined ms/step', d['ms_per_step'], 'cold', d['ms_per_step_cold_clocks'], 'sampler serial us', [v['avg_us'] for k,v in d['kernels_us'].items() if k.startswith('sdfk_sample')], 'k_vertices', d['kernels_us'].get('k_vertices',{}).get('avg_us'))"
  done
done 2>&1 | tee $O/places_$scene.log
