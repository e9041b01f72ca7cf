#!/bin/bash
# On the GPU box: the volume-less legs of the bench line (pipelined step, one synchronous call) for library builds: $VARIANTS
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
V=${VARIANTS:-"memset default memset default"}
for v in $V; do
    lib=$R/sdfkit_amd/_ablate/$v.so; [ "$v" = default ] && lib=$R/sdfkit_amd/libsdfkit_hip.so
    echo -n "$v  "
    SDFKIT_HIP_LIBRARY=$lib SDFK_BENCH_NO_C3=1 timeout 300 python3 bench.py --no-cpu 2>/dev/null | grep "^{" | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('stored', d['ms_per_step'], 'volume-less', d['elided_volume_ms_per_step'], 'one call', d['latency_ms_single_stream_product_default'], 'one call stored', d['latency_ms_single_stream'])"
done
