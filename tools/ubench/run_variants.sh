#!/bin/bash
# GPU box: bench each library variant in gpurun_variants/ alternately (A/B on the same box)
for rep in 1 2 3; do
for f in gpurun_variants/lib_*.so; do
  cp $f sdfkit_amd/libsdfkit_hip.so; touch sdfkit_amd/libsdfkit_hip.so
  python bench.py --steps 40 --warmup 3 --no-cpu 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['ms_per_step'], d['roofline']['avg_launch_us'])"
done; done
