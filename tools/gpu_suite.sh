#!/bin/bash
# full -m gpu suite, default bench, the sharded step behind the C ABI (RCCL world 1; two ranks on one GPU)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/suite; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -30 $O/pytest.log
timeout 600 python3 bench.py --steps 20 --warmup 5 --cpu-n 128 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
for g in 128 512; do
  SDFK_BENCH_FORCE_DIST=1 timeout 300 python3 bench.py --steps 200 --warmup 5 --no-cpu --minimal --grid $g > $O/bench_dist1_$g.json 2> $O/bench_dist1_$g.err; echo "dist1 $g rc $?"
done
SDFK_BENCH_ONE_GPU=1 timeout 600 python3 bench.py --gpus 2 --no-cpu --minimal > $O/bench_two_ranks_one_gpu.json 2> $O/bench_two.err; echo "two ranks rc $?"
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/suite/bench*.json")):
    ls=[l for l in open(f) if l.startswith("{")]
    if not ls: print(f,"NO LINE"); continue
    d=json.loads(ls[-1])
    print(f, d["value"], d["ms_per_step"], d.get("value_cold_clocks"), d.get("latency_ms_single_stream"), d.get("frac_measured_bytes"), d.get("frac_single_call"), d.get("one_step_incl_mesh_d2h_ms"))
    if "sharded" in d:
        s=d["sharded"]; print("   ", s["host_us_per_step"]["submit"], s["host_us_per_step"]["collect"], s["slab_kernels_only_ms"], s["exchange"], s["steps_redone_on_the_exact_path"])
PY
for f in $O/*.err; do echo "== $f"; tail -n 3 $f; done
