#!/bin/bash
# On the GPU box: the counters behind bench.py's `elided.roofline` -- the PRODUCT DEFAULT (volume-less) step of the 512^3 sphere, one
# in-order stream (SDFK_LANES=0): rocprofv3 kernel stats, the two SQ counter passes of tools/gpu_pmc.sh and the two HBM passes, all of
# `bench.py --minimal --elide`; per-kernel means go to profiles-style text files, the figures bench.py reads into
# profiles/pmc_traffic.json under `...@sphere_elided@512`.   usage: tools/gpu_elided_counters.sh r06
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp SDFK_LANES=0
O=gpurun_out/elided_$tag; mkdir -p $O
B="python3 bench.py --steps 3 --warmup 1 --no-cpu --minimal --elide"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats -o s --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu --minimal --elide > $O/bench_elided_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS \
    -d $R/$O/pmcA -o p --output-format csv -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU \
    -d $R/$O/pmcB -o p --output-format csv -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/$O/pmc_w -o p --output-format csv -- $B > /dev/null 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/$O/pmc_f -o p --output-format csv -- $B > /dev/null 2>&1
cp $O/stats/s_kernel_stats.csv $O/kernel_stats_serial_elided.csv
python3 tools/pmc_summary.py $O/pmcA/p_counter_collection.csv $O/pmcB/p_counter_collection.csv > $O/pmc_sq_counters_elided.txt
python3 tools/pmc_summary.py $O/pmc_w/p_counter_collection.csv $O/pmc_f/p_counter_collection.csv > $O/pmc_hbm_traffic_elided.txt
python3 - "$O" <<'PY'
import json, os, sys
sys.path.insert(0, "tools")
from pmc_summary import summarise
O = sys.argv[1]
sq = summarise([os.path.join(O, "pmcA", "p_counter_collection.csv"), os.path.join(O, "pmcB", "p_counter_collection.csv")])
hb = summarise([os.path.join(O, "pmc_w", "p_counter_collection.csv"), os.path.join(O, "pmc_f", "p_counter_collection.csv")])
p = "profiles/pmc_traffic.json"
d = json.load(open(p))
def short(k):
    return k.replace("void ", "").replace("sdfk::", "").split("(")[0].strip()
for k, c in sq.items():
    name = short(k)
    if name.startswith("k_vertices"):
        for cn in ("SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS"):
            if cn in c:
                d[f"{name}:{cn}@sphere_elided@512"] = int(round(c[cn]))
step = 0.0
for k, c in hb.items():
    name = short(k)
    if not (name.startswith("k_") or name.startswith("sdfk_")) or name.startswith("k_spin"):
        continue
    b = c.get("WRITE_SIZE", 0.0) * 1024 + 2 * c.get("FETCH_SIZE", 0.0) * 1024
    d[f"{name}@sphere_elided@512"] = int(round(b))
    if name.startswith(("sdfk_cull_blocks", "sdfk_eval_blocks", "k_compact", "k_blockscan", "k_chunkscan", "sdfk_corners_eval", "k_resolve", "k_vertices", "k_triangles")):
        step += b
d["pipeline_step@sphere_elided@512"] = int(round(step))
json.dump(d, open(p, "w"), indent=1, sort_keys=True)
json.dump(d, open(os.path.join(O, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
print(f"pipeline_step@sphere_elided@512 = {step / 1e6:.1f} MB")
PY
rm -rf $O/stats $O/pmcA $O/pmcB $O/pmc_w $O/pmc_f
cut -d, -f1,2,4 $O/kernel_stats_serial_elided.csv | cut -c1-110 | head -14
head -30 $O/pmc_sq_counters_elided.txt
head -14 $O/pmc_hbm_traffic_elided.txt
