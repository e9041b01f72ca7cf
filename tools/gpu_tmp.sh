#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/profiles_r02b; mkdir -p $O
timeout 200 python3 bench.py --no-cpu --scene union8 --grid 1024 --steps 5 --warmup 2 > $O/bench_c4_1024.json 2>/dev/null; echo "c4 rc $?"
SDFK_BENCH_ONE_GPU=1 timeout 300 python3 bench.py --gpus 2 --no-cpu > $O/bench_two_ranks_one_gpu.json 2>/dev/null; echo "two ranks rc $?"
SDFK_LANES=0 timeout 200 rocprofv3 --pmc WRITE_SIZE -d $R/$O/pmc_w3 -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --minimal --scene repeatxy > /dev/null 2>&1; echo "pmc w3 rc $?"
SDFK_LANES=0 timeout 200 rocprofv3 --pmc FETCH_SIZE -d $R/$O/pmc_f3 -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --minimal --scene repeatxy > /dev/null 2>&1; echo "pmc f3 rc $?"
if [ -f $O/pmc_w3/p_counter_collection.csv ] && [ -f $O/pmc_f3/p_counter_collection.csv ]; then
  python3 tools/pmc_summary.py $O/pmc_w3/p_counter_collection.csv $O/pmc_f3/p_counter_collection.csv > $O/pmc_hbm_traffic_repeatxy.txt
  cp profiles/pmc_traffic.json $O/pmc_traffic.json
  python3 tools/pmc_to_json.py $O/pmc_w3/p_counter_collection.csv $O/pmc_f3/p_counter_collection.csv repeatxy 512 $O/pmc_traffic.json
fi
rm -rf $O/pmc_w3 $O/pmc_f3
grep "^{" $O/bench_c4_1024.json | cut -c1-200; grep "^{" $O/bench_two_ranks_one_gpu.json | cut -c1-200
