#!/bin/bash
# On the GPU box: BASELINE config C5 (RayMarcher 1920 x 1080 x 256, README scene) -- its timing line, the rocprofv3 kernel stats and one SQ
# counter pass of the same command; the counters bench.py's c5_raymarch.valu_issue_frac is derived from go into profiles/pmc_traffic.json.
# usage: tools/gpu_c5_profile.sh r05
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/c5_$tag; mkdir -p $O
python3 tools/bench_raymarch.py 20 > $O/bench_raymarch.json 2>$O/bench_raymarch.err
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats -o s --output-format csv -- python3 tools/bench_raymarch.py 20 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU \
    -d $R/$O/pmc -o p --output-format csv -- python3 tools/bench_raymarch.py 5 > /dev/null 2>&1
cp $O/stats/s_kernel_stats.csv $O/kernel_stats_c5_raymarch.csv
python3 tools/pmc_summary.py $O/pmc/p_counter_collection.csv > $O/pmc_sq_counters_c5_raymarch.txt
python3 - "$O" <<'PY'
import json, os, sys
sys.path.insert(0, "tools")
from pmc_summary import summarise
O = sys.argv[1]
res = summarise([os.path.join(O, "pmc", "p_counter_collection.csv")])
p = "profiles/pmc_traffic.json"
d = json.load(open(p))
for k, c in res.items():
    if k.startswith("sdfk_raymarch"):
        for name in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES"):
            if name in c:
                d[f"sdfk_raymarch:{name}@c5@1080"] = int(round(c[name]))
json.dump(d, open(p, "w"), indent=1, sort_keys=True)
json.dump(d, open(os.path.join(O, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
PY
rm -rf $O/stats $O/pmc
cat $O/bench_raymarch.json; cat $O/pmc_sq_counters_c5_raymarch.txt | head -20
