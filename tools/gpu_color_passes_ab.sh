#!/bin/bash
# On the GPU box: SDFK_OPT_COLOR_PASSES A/B -- a colour volume sampled in one fused pass (1) against values + sign bytes, then the colour
# array as one linear stream (2): the sampler alone (tools/c3_sampler_probe.py: back-to-back launches, event-timed) for the README scene
# at 512^3 and the 8-primitive union at 512^3 and 1024^3, then the pipelined stored step of both scenes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/colorpasses; mkdir -p $O
{
for rep in 1 2; do
    for passes in 1 2; do
        echo "== SDFK_COLOR_PASSES=$passes (sampler alone)"
        SDFK_COLOR_PASSES=$passes python3 tools/c3_sampler_probe.py repeatxy 512 2 20 2>/dev/null | tail -1
        SDFK_COLOR_PASSES=$passes python3 tools/c3_sampler_probe.py union8 512 2 20 2>/dev/null | tail -1
        SDFK_COLOR_PASSES=$passes python3 tools/c3_sampler_probe.py union8 1024 2 6 2>/dev/null | tail -1
    done
done
for passes in 1 2; do
    echo "== SDFK_COLOR_PASSES=$passes (pipelined stored step, bench.py --minimal)"
    for sc in "--scene repeatxy" "--scene union8 --grid 1024 --steps 5"; do
        SDFK_COLOR_PASSES=$passes python3 bench.py --no-cpu --minimal $sc 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['grid'], d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels_us'].items() if k.startswith('sdfk_sample')})"
    done
done
} > $O/ab.log 2>&1
cat $O/ab.log
