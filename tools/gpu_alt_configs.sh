#!/bin/bash
# the -m gpu suite under alternate start-up configurations (each one a path the default run does not take)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/alt; mkdir -p $O
run_impl() {
  tag=$1; shift
  env "$@" timeout 1500 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/$tag.log 2>&1
  echo "$tag ($*): rc $? -- $(tail -1 $O/$tag.log)"
}
# usage: tools/gpu_alt_configs.sh [tag ...]   (no argument: every configuration)
want() { [ ${#SEL[@]} -eq 0 ] && return 0; for t in "${SEL[@]}"; do [ "$t" = "$1" ] && return 0; done; return 1; }
SEL=("$@")
run() { if want "$1"; then run_impl "$@"; fi; }
run placement_off SDFK_STREAM_PLACEMENT=0
run lanes4 SDFK_LANES=4
run lanes0 SDFK_LANES=0
run hwq4 GPU_MAX_HW_QUEUES=4
run graphs_off SDFK_GRAPHS=0
run dist_conservative SDFK_DIST_LANES=0 SDFK_DIST_EXCHANGE=0
run dist_index16 SDFK_DIST_INDEX16=1 SDFK_DIST_EXCHANGE=2
run gather_paths SDFK_NO_CORNER_EVAL=1 SDFK_NO_VCOLOR_EVAL=1
run elide_volume SDFK_ELIDE_VOLUME=1
run stored_volume SDFK_ELIDE_VOLUME=0
run dist_direct SDFK_DIST_EXCHANGE=1
run color_two_passes SDFK_COLOR_PASSES=2
run color_one_pass SDFK_COLOR_PASSES=1
run dist_sharded SDFK_DIST_EXCHANGE=3
run dist_to_root SDFK_DIST_EXCHANGE=2
