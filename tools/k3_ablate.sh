#!/bin/bash
# Where does k_resolve spend its time?  Debug builds (-DSDFK_K3_ABLATE=1: no tiling resolution,
# 2: no vertex-ownership loop; 0: full; meshes of 1 and 2 are wrong by construction), serial bench.
#   build (anywhere):  tools/k3_ablate.sh build      run (GPU box):  tools/k3_ablate.sh run [scene]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
D=sdfkit_amd/_ablate
if [ "$1" = build ]; then
    mkdir -p $D
    for A in 0 1 2; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fno-fast-math \
            -Wno-unused-function -DSDFK_K3_ABLATE=$A -o $D/k3_$A.so sdfkit_amd/csrc/sdfkit_hip.hip -lhiprtc &
    done
    wait; ls -la $D
else
    cp sdfkit_amd/libsdfkit_hip.so /tmp/lib_good.so
    for A in 0 1 2; do
        cp $D/k3_$A.so sdfkit_amd/libsdfkit_hip.so
        echo -n "ablate=$A  "
        SDFK_LANES=0 timeout 120 python3 bench.py --no-cpu --scene ${2:-sphere} 2>/dev/null | grep "^{" | \
            python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'k_resolve', d['kernels_us']['k_resolve']['avg_us'])"
    done
    cp /tmp/lib_good.so sdfkit_amd/libsdfkit_hip.so
fi
