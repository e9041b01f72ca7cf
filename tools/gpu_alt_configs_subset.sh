#!/bin/bash
# the -m gpu suite in the default configuration and under the alternate start-up configurations that take other code paths
# through the meshing kernels (the full list: tools/gpu_alt_configs.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/alt; mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout 1500 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/$tag.log 2>&1
  echo "$tag ($*): rc $? -- $(tail -1 $O/$tag.log)"
}
run default SDFK_UNUSED=0
run stored_volume SDFK_ELIDE_VOLUME=0
run lanes0 SDFK_LANES=0
run graphs_off SDFK_GRAPHS=0
run dist_index16 SDFK_DIST_INDEX16=1 SDFK_DIST_EXCHANGE=2
run gather_paths SDFK_NO_CORNER_EVAL=1 SDFK_NO_VCOLOR_EVAL=1
