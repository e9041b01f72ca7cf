#!/bin/bash
# round 2: sampler without the colour staging LDS for .W-only programs -- parity, bench A/B (lanes x depth)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r02h; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -k "w_only or sample_bit_exact or golden" > $O/pytest_a.log 2>&1; echo "pytest a rc $?" >> $O/pytest_a.log
tail -3 $O/pytest_a.log
run() { echo -n "$* : "; env "$@" timeout 300 python3 bench.py --no-cpu --minimal --steps 100 --warmup 5 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
run X=0
run X=0
run SDFK_LANES=2 SDFK_BENCH_DEPTH=3
run SDFK_LANES=3 SDFK_BENCH_DEPTH=4
run SDFK_LANES=3 SDFK_BENCH_DEPTH=6
run SDFK_LANES=4 SDFK_BENCH_DEPTH=6
run SDFK_LANES=4 SDFK_BENCH_DEPTH=8
run SDFK_LANES=0
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep "^{" > $O/bench.json
python3 -c "
import json; d=json.loads(open('$O/bench.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['latency_ms_single_stream'])"
