#!/bin/bash
# full -m gpu suite + smoke + default bench (what the driver runs at round end)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/full; mkdir -p $O
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/full/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["latency_ms_single_stream"], d["one_step_incl_mesh_d2h_ms"], d["cpu_baseline"]["value"], d["first_call"])
PY
