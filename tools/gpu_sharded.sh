#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/sharded; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_multirank.py -x -q 2>&1 | tail -4
for i16 in 0 1; do for g in 128 512; do
  SDFK_BENCH_INDEX16=$i16 SDFK_BENCH_FORCE_DIST=1 timeout 300 python3 bench.py --steps 200 --warmup 5 --no-cpu --minimal --grid $g > $O/dist1_${g}_$i16.json 2> $O/err.txt
  python3 - $O/dist1_${g}_$i16.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s=d["sharded"]
print(sys.argv[1], d["ms_per_step"], s["host_us_per_step"]["submit"], s["slab_kernels_only_ms"], s["gather_stride_bytes_per_rank"], s["payload"][:40])
PY
done; done
SDFK_BENCH_ONE_GPU=1 timeout 600 python3 bench.py --gpus 2 --no-cpu --minimal > $O/two.json 2>> $O/err.txt; grep -o '"payload": "[^"]*"' $O/two.json | cut -c1-120
