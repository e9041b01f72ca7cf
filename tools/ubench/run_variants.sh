#!/bin/bash
# GPU box: bench each library variant in gpurun_variants/ (serial lanes, per-kernel times)
for f in gpurun_variants/lib_*.so; do
  cp $f sdfkit_amd/libsdfkit_hip.so; touch sdfkit_amd/libsdfkit_hip.so
  echo "== $f"
  SDFK_LANES=0 python bench.py --steps 20 --warmup 3 --no-cpu 2>&1 | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k:v['avg_us'] for k,v in d['kernels_us'].items()})"
done
