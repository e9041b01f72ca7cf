#!/usr/bin/env python3
"""GPU probe: the fused sampling kernel alone on slab-shaped volumes (512 x 512 x planes), as the Z-slab ranks of a
sharded 512^3 run see them (planes = 64 + context at 8 ranks, 128 + context at 4, ...).  SDFK_SAMPLE_MODE=0 forces the
row-tiled form for comparison."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import sdfkit_amd as S
from sdfkit_amd import _native as N
from sdfkit_amd.api import Voxels
N.init(); L = N.lib()
dev = torch.device("cuda:0"); stream = N.bind_torch_stream(dev)
sdf = S.Sdfs.Sphere(1.0)
for nz in (68, 132, 260, 512):
    vols = [Voxels([-1.5] * 3, [1.5] * 3, 512, 512, nz) for _ in range(4)]
    for k in range(8): vols[k % 4]._sample(sdf, clip=False)
    N.check(L.sdfk_profile_enable(2))
    for k in range(4): vols[k % 4]._sample(sdf, clip=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    K = 40
    for k in range(K): vols[k % 4]._sample(sdf, clip=False)
    e1.record(stream); torch.cuda.synchronize()
    N.check(L.sdfk_profile_enable(0))
    us = e0.elapsed_time(e1) * 1e3 / K
    print(f"512x512x{nz}: {us:.1f} us per launch, {512*512*nz*4.125/us/1e6:.2f} TB/s")
    for v in vols: v._free()
