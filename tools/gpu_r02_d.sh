#!/bin/bash
# round 2: full -m gpu suite after the host-copy changes, grid-cap sweep of the meshing kernels, host-IO probe, bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r02d; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
run() { echo -n "$* : "; env "$@" SDFK_LANES=0 timeout 300 python3 bench.py --no-cpu 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels_us'].items() if k in ('k_resolve','k_vertices','k_triangles')})"; }
run X=0
for g in 1536 1024 768 512; do run SDFK_GRID_RESOLVE=$g; done
for g in 1536 1024 768; do run SDFK_GRID_VERT=$g; done
for g in 1536 1024 768 512; do run SDFK_GRID_TRI=$g; done
for h in 0 1; do echo "== PROBE_HUGE=$h"; PROBE_HUGE=$h timeout 300 python3 tools/host_io_probe.py 2>&1 | grep -v amdgpu.ids | grep "mesh_copy\|THP"; done > $O/host_io.log 2>&1
cat $O/host_io.log
timeout 600 python3 bench.py --cpu-passes 1 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r02d/bench.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["latency_ms_single_stream"], d["one_step_incl_mesh_d2h_ms"]); print({k: v for k, v in d["one_step_incl_mesh_d2h"].items() if k != "what"})
PY
