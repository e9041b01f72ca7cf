#!/bin/bash
# round 2: rsq-based sdfk_sqrt -- exhaustive check, parity suite, bench A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r02f; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_sqrt_exhaustive.py tests/test_gpu_parity.py tests/test_raymarch.py -x -q > $O/pytest_a.log 2>&1; echo "pytest a rc $?" >> $O/pytest_a.log
tail -5 $O/pytest_a.log
for i in 1 2 3; do timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu 2>/dev/null | grep "^{" > $O/bench_$i.json; done
for sc in repeatxy union8; do timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu --minimal --scene $sc 2>/dev/null | grep "^{" > $O/bench_$sc.json; done
python3 - <<PY
import json
for f in ("bench_1","bench_2","bench_3","bench_repeatxy","bench_union8"):
    d=json.loads(open("gpurun_out/r02f/%s.json"%f).read())
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d.get("latency_ms_single_stream"), {k: v["avg_us"] for k, v in d["kernels_us"].items()})
PY
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
