#!/bin/bash
# Experiment builds of the library with extra -D flags, for A/B timing on the GPU box.
#   build (anywhere):   tools/variants.sh build NAME "-DSDFK_KV_MINWAVES=4" [NAME2 "-D..."] ...
#   run (GPU box):      tools/variants.sh run NAME [bench args]      (serial bench: per-kernel times)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
D=sdfkit_amd/_ablate
if [ "$1" = build ]; then
    shift; mkdir -p $D
    while [ $# -ge 2 ]; do
        python3 -m sdfkit_amd.build $R/$D/$1.so $2 &    # (the same translation units with the extra flags, objects of their own)
        shift 2
    done
    wait; ls -la $D
else
    shift; name=$1; shift
    lib=$R/$D/$name.so; [ "$name" = default ] && lib=$R/sdfkit_amd/libsdfkit_hip.so
    echo -n "$name  "
    SDFKIT_HIP_LIBRARY=$lib SDFK_LANES=0 timeout 300 python3 bench.py --no-cpu "$@" 2>/dev/null | grep "^{" | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial ms/step', d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels_us'].items()})"
    SDFKIT_HIP_LIBRARY=$lib timeout 300 python3 bench.py --no-cpu "$@" 2>/dev/null | grep "^{" | \
        python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   pipelined ms/step', d['ms_per_step'], 'latency', d['latency_ms_single_stream'])"
fi
