#!/bin/bash
# round 2, second GPU call: new host-IO tests, probe on private anonymous memory, bench with the three D2H kinds, PMC counters per kernel
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r02b; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_host_io.py tests/test_shim_oplists.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
for t in 8 16; do for h in 0 1; do echo "== SDFK_COPY_THREADS=$t PROBE_HUGE=$h"; PROBE_HUGE=$h SDFK_COPY_THREADS=$t timeout 300 python3 tools/host_io_probe.py 2>&1 | grep -v amdgpu.ids | grep -v upload; done; done > $O/host_io.log 2>&1
cat $O/host_io.log
timeout 600 python3 bench.py --cpu-passes 1 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r02b/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["latency_ms_single_stream"], d["one_step_incl_mesh_d2h_ms"]); print(d["one_step_incl_mesh_d2h"])
PY
bash tools/gpu_pmc.sh > $O/pmc.log 2>&1; cat $O/pmc.log | head -150
