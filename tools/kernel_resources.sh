#!/bin/bash
# Per-kernel VGPR / scratch / LDS usage of the product kernels (device-only compile, no GPU needed).
set -e
T=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -c \
    "$(dirname "$0")/../sdfkit_amd/csrc/mc_kernels.hip" -o $T/dev.o
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/dev.o \
    --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | \
    grep -E "\.name:|\.vgpr_count|private_segment_fixed|vgpr_spill|group_segment_fixed" | paste - - - - - | \
    sed -e 's/ \+/ /g'
[ -n "$KEEP" ] && echo "$T/k.co" || rm -rf $T
