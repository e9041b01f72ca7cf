#!/bin/bash
# On the GPU box: k_blockscan / k_chunkscan (prefixes of many compaction blocks / chunks in one workgroup) against every consumer workgroup
# summing its predecessors itself (tools/variants.sh build noscan "-DSDFK_SCAN_BLOCKS=1000000000 -DSDFK_SCAN_CHUNKS=4000000000u"):
# C4 (8-primitive union, 1024^3) serial per-kernel times + pipelined step, stored and volume-less; the 512^3 sphere for reference.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/scan; mkdir -p $O
{
for v in noscan default noscan default; do
    tools/variants.sh run $v --scene union8 --grid 1024 --steps 5 --warmup 2
    lib=$R/sdfkit_amd/_ablate/$v.so; [ "$v" = default ] && lib=$R/sdfkit_amd/libsdfkit_hip.so
    SDFKIT_HIP_LIBRARY=$lib timeout 300 python3 bench.py --no-cpu --scene union8 --grid 1024 --steps 5 --warmup 2 2>/dev/null | grep "^{" | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   volume-less: pipelined ms/step', d['elided_volume_ms_per_step'], 'one call', d['latency_ms_single_stream_product_default'])"
done
for v in noscan default noscan default; do tools/variants.sh run $v --minimal --steps 200 --warmup 5; done
} > $O/ab.log 2>&1
grep -v "^  File\|^    \|Traceback\|json.decoder" $O/ab.log
