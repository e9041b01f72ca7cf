#!/bin/bash
# On the GPU box: everything profiles/ needs for a round.  usage: tools/gpu_round_profiles.sh r04
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/profiles_$tag; mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
# (the kernel statistics of the HEADLINE workload only: without the C3 and elided-volume legs of the default line, whose launches
# -- another scene, another sampler -- would mix into the per-kernel means)
export SDFK_BENCH_NO_C3=1 SDFK_BENCH_NO_C2=1 SDFK_BENCH_NO_C4=1 SDFK_BENCH_NO_C5=1 SDFK_BENCH_NO_ELIDED=1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats -o s --output-format csv -- python3 bench.py --no-cpu > $O/bench_under_rocprof.json 2>/dev/null
SDFK_LANES=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats_serial -o s --output-format csv -- python3 bench.py --no-cpu > $O/bench_serial_under_rocprof.json 2>/dev/null
unset SDFK_BENCH_NO_C3 SDFK_BENCH_NO_C2 SDFK_BENCH_NO_C4 SDFK_BENCH_NO_C5 SDFK_BENCH_NO_ELIDED
SDFK_LANES=0 timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/$O/pmc_w -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --minimal > /dev/null 2>&1
SDFK_LANES=0 timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/$O/pmc_f -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --minimal > /dev/null 2>&1
python3 tools/pmc_summary.py $O/pmc_w/p_counter_collection.csv $O/pmc_f/p_counter_collection.csv > $O/pmc_hbm_traffic.txt
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python3 tools/pmc_to_json.py $O/pmc_w/p_counter_collection.csv $O/pmc_f/p_counter_collection.csv sphere 512 $O/pmc_traffic.json
# config C3 (RepeatXY with colours, clipToBounds): serial kernel stats + the two PMC passes
python3 bench.py --no-cpu --scene repeatxy > $O/bench_repeatxy.json 2>/dev/null
SDFK_LANES=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/stats_c3 -o s --output-format csv -- python3 bench.py --no-cpu --scene repeatxy > /dev/null 2>&1
SDFK_LANES=0 timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/$O/pmc_w3 -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --minimal --scene repeatxy > /dev/null 2>&1
SDFK_LANES=0 timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/$O/pmc_f3 -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --minimal --scene repeatxy > /dev/null 2>&1
python3 tools/pmc_summary.py $O/pmc_w3/p_counter_collection.csv $O/pmc_f3/p_counter_collection.csv > $O/pmc_hbm_traffic_repeatxy.txt
python3 tools/pmc_to_json.py $O/pmc_w3/p_counter_collection.csv $O/pmc_f3/p_counter_collection.csv repeatxy 512 $O/pmc_traffic.json
cp $O/stats_c3/s_kernel_stats.csv $O/kernel_stats_serial_repeatxy.csv
rm -rf $O/stats_c3 $O/pmc_w3 $O/pmc_f3
cp $O/stats/s_kernel_stats.csv $O/kernel_stats.csv
cp $O/stats_serial/s_kernel_stats.csv $O/kernel_stats_serial.csv
rm -rf $O/stats $O/stats_serial $O/pmc_w $O/pmc_f
# C2 (launch-bound), C4 at its full size on one GPU, the sharded step over real RCCL at world 1 (small slab and 512^3), and the
# sharded path as the driver starts it (every rank on this one GPU: the library's host transport over gloo)
python3 bench.py --no-cpu --grid 256 > $O/bench_c2_256.json 2>/dev/null
SDFK_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu --minimal --grid 128 --steps 200 > $O/bench_rccl_world1_128.json 2>/dev/null
SDFK_BENCH_FORCE_DIST=1 python3 bench.py --no-cpu --minimal --steps 100 > $O/bench_rccl_world1_512.json 2>/dev/null
python3 bench.py --no-cpu --scene union8 --grid 1024 --steps 5 --warmup 2 > $O/bench_c4_1024.json 2>/dev/null
SDFK_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --no-cpu > $O/bench_two_ranks_one_gpu.json 2>/dev/null
grep "^{" $O/bench.json | cut -c1-400
cut -d, -f1,2,4 $O/kernel_stats_serial.csv | cut -c1-100
cat $O/pmc_hbm_traffic.txt | head -12
grep "^{" $O/bench_repeatxy.json | cut -c1-300
grep "^{" $O/bench_c4_1024.json | cut -c1-300
head -4 $O/pmc_hbm_traffic_repeatxy.txt
# instruction / wait counters per kernel (two more PMC passes)
bash tools/gpu_pmc.sh 2>&1 | grep -v "^W2026\|^E2026" > $O/pmc_sq_counters.txt
# the volume-less (product default) step: kernel stats, SQ and HBM counters (-> profiles/pmc_traffic.json `...@sphere_elided@512`)
bash tools/gpu_elided_counters.sh $tag > $O/elided_counters.log 2>&1
cp gpurun_out/elided_$tag/*.csv gpurun_out/elided_$tag/*.txt $O/ 2>/dev/null
cp profiles/pmc_traffic.json $O/pmc_traffic.json
# the default line once more, now with the counters of THIS box behind roofline.traffic / elided.roofline
python3 bench.py --no-cpu > $O/bench_no_cpu_with_fresh_counters.json 2>/dev/null
