#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
bash tools/gpu_kern.sh
PROBE_WORLD=8 PROBE_RANK=1 timeout 200 python3 tools/slab_chain_probe.py 2>&1 | grep -v amdgpu.ids
