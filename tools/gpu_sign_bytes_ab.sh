#!/bin/bash
# On the GPU box: A/B of the sampling kernels' sign-byte assembly (profiles/r06_ab_sign_bytes.txt).  The two libraries are built in the container first:
#   sdfkit_amd/_ablate/oldsign.so   = the product           (python3 -m sdfkit_amd.build $PWD/sdfkit_amd/_ablate/oldsign.so -DSDFK_VARIANT_OLD=1)
#   sdfkit_amd/libsdfkit_hip.so     = the tree with the variant applied to csrc/sample_codegen.h (docs/history.md "Round 6": the nibble of a row
#                                     spread to one bit per byte with ((n * 0x204081) & 0x01010101) << r, wavefront 0 ORs one word per wavefront)
# then: the parity suites on the variant, and three alternating default bench runs of each.
export TMPDIR=/tmp SDFK_BENCH_NO_C2=1 SDFK_BENCH_NO_C4=1 SDFK_BENCH_NO_C5=1 SDFK_BENCH_NO_CONTROL=1
mkdir -p gpurun_out/r06f
timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_elide_volume.py tests/test_gpu_color_passes.py tests/test_reference_vectors.py tests/test_golden_fixtures.py -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2 3; do
  for lib in sdfkit_amd/_ablate/oldsign.so sdfkit_amd/libsdfkit_hip.so; do
    SDFKIT_HIP_LIBRARY=$PWD/$lib python3 bench.py --no-cpu 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', 'step', d['ms_per_step'], 'min', d['blocks']['ms_per_step_min'], 'sampler', d['roofline']['avg_launch_us'], d['roofline']['frac'], 'serial kernels', d['kernels_us']['sdfk_sample_bits']['avg_us'], 'c3', d['c3_repeatxy']['ms_per_step'], d['c3_repeatxy']['sampler_us_back_to_back'], 'elided', d['elided_volume_ms_per_step'], 'lat', d['latency_ms_single_stream'])"
  done
done
