#!/bin/bash
# round 4: whole -m gpu suite; C3 bench + kernel trace of its back-to-back sampler pass
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r04c; mkdir -p $O
timeout 1700 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 600 python3 bench.py --no-cpu --scene repeatxy > $O/bench_repeatxy.json 2> $O/bench_repeatxy.err; echo "c3 rc $?"
python3 - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r04c/bench_repeatxy.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["kernels_us"])
PY
cd /tmp; timeout 600 rocprofv3 --kernel-trace -d $R/$O/trace -o t --output-format csv -- python3 $R/bench.py --no-cpu --scene repeatxy --steps 12 > /dev/null 2>&1; cd $R
f=$(ls $O/trace/*/t_kernel_trace.csv $O/trace/t_kernel_trace.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# runs of consecutive sampler launches (the back-to-back roofline pass): duration and the gap to the previous sampler
run=[]; prev=None
for r in rows:
    nm=r["Kernel_Name"]
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if nm.startswith("sdfk_sample_bits") and prev is not None and prev[0].startswith("sdfk_sample_bits"):
        run.append(((e-s)/1e3,(s-prev[2])/1e3))
    prev=(nm,s,e)
if run:
    d=sorted(x[0] for x in run); g=sorted(x[1] for x in run)
    print(f"back-to-back sampler launches: n={len(run)} dur med {d[len(d)//2]:.1f} us (min {d[0]:.1f} max {d[-1]:.1f}); gap-before med {g[len(g)//2]:.2f} us (max {g[-1]:.2f})")
PY
rm -rf $O/trace
