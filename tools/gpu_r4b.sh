#!/bin/bash
# round 4: whole -m gpu suite, default bench line, C3 traces
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
timeout 1700 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -3 $O/bench.err
python3 - <<'PY'
import json
ls=[l for l in open("gpurun_out/r04b/bench.json") if l.startswith("{")]
d=json.loads(ls[-1])
for k in ("value","ms_per_step","latency_ms_single_stream","first_call_ms","first_call_new_constants_ms","one_step_incl_mesh_d2h_ms","frac_measured_bytes","c3_repeatxy"):
    print(k, d.get(k))
print(d["first_call"])
print({k:v["median_ms"] for k,v in d["one_step_incl_mesh_d2h"].items() if isinstance(v,dict)})
print(d["roofline"])
PY
SDFK_BENCH_ONE_GPU=1 timeout 600 python3 bench.py --gpus 2 --no-cpu --minimal --grid 256 > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two.err; echo "two ranks rc $?"; tail -3 $O/bench_two.err
SDFK_BENCH_FORCE_DIST=1 timeout 300 python3 bench.py --steps 100 --warmup 5 --no-cpu --minimal > $O/bench_dist1_512.json 2> $O/bench_dist1.err; echo "dist1 rc $?"; tail -2 $O/bench_dist1.err
