#!/bin/bash
# On the GPU box: the sampling kernels' 16-byte stores as "sc1 nt" (SDFK_SAMPLE_NT=2, the product) against "nt" (=1): serial per-kernel
# times + pipelined step, sphere and README scene at 512^3, C4 at 1024^3; bench.py's own back-to-back roofline figure for the sphere.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/storeflav; mkdir -p $O
OLD="-USDFK_SAMPLE_NT -DSDFK_SAMPLE_NT=1"
{
for rep in 1 2; do
    echo "== nt";      SDFK_JIT_FLAGS="$OLD" tools/variants.sh run default --minimal --steps 200 --warmup 5
    echo "== sc1 nt";  tools/variants.sh run default --minimal --steps 200 --warmup 5
done
for rep in 1 2; do
    echo "== nt";      SDFK_JIT_FLAGS="$OLD" tools/variants.sh run default --minimal --steps 100 --warmup 5 --scene repeatxy
    echo "== sc1 nt";  tools/variants.sh run default --minimal --steps 100 --warmup 5 --scene repeatxy
done
echo "== nt";      SDFK_JIT_FLAGS="$OLD" tools/variants.sh run default --scene union8 --grid 1024 --steps 5 --warmup 2
echo "== sc1 nt";  tools/variants.sh run default --scene union8 --grid 1024 --steps 5 --warmup 2
for f in "$OLD" ""; do
    echo "== roofline leg, SDFK_JIT_FLAGS='$f'"
    SDFK_JIT_FLAGS="$f" python3 bench.py --no-cpu 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['ms_per_step'], r['avg_launch_us'], r['frac'], 'c3 sampler', d['c3_repeatxy']['sampler_us_back_to_back'], d['c3_repeatxy']['sampler_frac'])"
done
} > $O/ab.log 2>&1
grep -v "^  File\|^    \|Traceback\|json.decoder" $O/ab.log
