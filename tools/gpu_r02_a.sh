#!/bin/bash
# round 2, first GPU call: the whole -m gpu suite, default bench, self-launched 2-rank bench (one-GPU mode),
# host-IO probe, serial kernel stats as this round's starting point
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
grep "^{" $O/bench.json | cut -c1-1500
SDFK_BENCH_ONE_GPU=1 timeout 600 python3 bench.py --gpus 2 --no-cpu > $O/bench_g2.json 2> $O/bench_g2.err; echo "bench g2 rc $?"
grep "^{" $O/bench_g2.json | cut -c1-600; tail -3 $O/bench_g2.err
for t in 4 16 32; do echo "== SDFK_COPY_THREADS=$t"; SDFK_COPY_THREADS=$t timeout 300 python3 tools/host_io_probe.py 2>&1 | grep -v amdgpu.ids; done > $O/host_io.log 2>&1
cat $O/host_io.log | head -60
SDFK_LANES=0 rocprofv3 --kernel-trace --stats -d $R/$O/stats_serial -o s --output-format csv -- python3 bench.py --no-cpu > $O/bench_serial_under_rocprof.json 2>/dev/null
cut -d, -f1,2,4 $O/stats_serial/s_kernel_stats.csv | cut -c1-100
cp $O/stats_serial/s_kernel_stats.csv $O/kernel_stats_serial.csv; rm -rf $O/stats_serial
