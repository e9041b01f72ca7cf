#!/bin/bash
# On the GPU box: k_vertices with runs of G consecutive chunks per XCD (SDFK_KV_XCD_GROUP, tools/variants.sh build xgG "-DSDFK_KV_XCD_GROUP=G")
# against chunk = workgroup: serial per-kernel times + pipelined step (both scenes), then FETCH_SIZE of the launch for each build.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/kvxcd; mkdir -p $O
V=${VARIANTS:-"default xg4 xg16 xg64 default"}
{
for v in $V; do tools/variants.sh run $v --minimal --steps 200 --warmup 5; done
for v in $V; do tools/variants.sh run $v --minimal --steps 100 --warmup 5 --scene repeatxy; done
for v in $V; do
    lib=$R/sdfkit_amd/_ablate/$v.so; [ "$v" = default ] && lib=$R/sdfkit_amd/libsdfkit_hip.so
    export SDFKIT_HIP_LIBRARY=$lib SDFK_LANES=0
    rm -rf $O/pmc_f
    timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/$O/pmc_f -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu --minimal > /dev/null 2>&1
    echo "FETCH_SIZE (KiB, x2 = bytes/1024) $v:"; python3 tools/pmc_summary.py $O/pmc_f/p_counter_collection.csv 2>/dev/null | grep -A1 "k_vertices\|k_resolve\|k_compact" | grep -v "^--"
    unset SDFKIT_HIP_LIBRARY SDFK_LANES
done
rm -rf $O/pmc_f
} > $O/ab.log 2>&1
grep -v "^  File\|^    \|Traceback\|json.decoder" $O/ab.log
