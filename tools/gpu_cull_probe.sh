#!/bin/bash
# On the GPU box: per-kernel times of the volume-less path at 512^3 (and 1024^3: $1 = 1024) for coarse-box culling variants
# (tools/elide_kernels_probe.py: SDFK_LANES=0, HIP events around each launch)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
for cfg in ${CFGS:-"default SDFK_CULL_COARSE=0 SDFK_CULL_CPW=1 SDFK_CULL_CPW=2 SDFK_CULL_CPW=4 SDFK_CULL_CPW=8 SDFK_CULL_CPW=16"}; do
    echo "== $cfg"
    [ "$cfg" = default ] && cfg="SDFK_NOTHING=1"
    env $cfg python3 tools/elide_kernels_probe.py $1 2>&1 | grep "elide 2" | sed -e "s/'k_compact.*'sdfk_cull/'sdfk_cull/"
done
