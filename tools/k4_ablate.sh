#!/bin/bash
# Where does k_vertices spend its time?  Ablation: debug builds of the library that leave the kernel
# early at successive points (-DSDFK_K4_ABLATE=1..5; their meshes are wrong by construction), timed
# with the serial bench.   1: after setup   2: + window staging, chunk prefix, scan
# 3: + owner search and sharer lookup   4: + vertex-id pushes   5: + position/colour/normal maths   0: full
#   build (anywhere):  tools/k4_ablate.sh build      run (GPU box):  tools/k4_ablate.sh run [scene]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
D=sdfkit_amd/_ablate
if [ "$1" = build ]; then
    mkdir -p $D
    for A in 0 1 2 3 4 5; do
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fno-fast-math \
            -Wno-unused-function -DSDFK_K4_ABLATE=$A -o $D/lib_$A.so sdfkit_amd/csrc/sdfkit_hip.hip -lhiprtc &
    done
    wait; ls -la $D
else
    cp sdfkit_amd/libsdfkit_hip.so /tmp/lib_good.so
    for A in 0 1 2 3 4 5; do
        cp $D/lib_$A.so sdfkit_amd/libsdfkit_hip.so
        echo -n "ablate=$A  "
        SDFK_LANES=0 timeout 120 python3 bench.py --no-cpu --scene ${2:-sphere} 2>/dev/null | grep "^{" | \
            python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'k_vertices', d['kernels_us']['k_vertices']['avg_us'], 'k_triangles', d['kernels_us']['k_triangles']['avg_us'])"
    done
    cp /tmp/lib_good.so sdfkit_amd/libsdfkit_hip.so
fi
