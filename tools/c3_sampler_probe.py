#!/usr/bin/env python3
"""The fused sampling kernel alone, back to back, event-timed: scene (sphere | repeatxy | union8), grid edge, number of resident
volumes it rotates over, launches.  Prints one line.  Used under rocprofv3 --pmc by tools/gpu_c3_counters.sh."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from sdfkit_amd import _native as N
from sdfkit_amd.api import Voxels
import bench

scene = sys.argv[1] if len(sys.argv) > 1 else "repeatxy"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
nvol = int(sys.argv[3]) if len(sys.argv) > 3 else 2
k = int(sys.argv[4]) if len(sys.argv) > 4 else 20
N.init(0)
L = N.lib()
stream = N.bind_torch_stream(torch.device("cuda", 0))
sdf, mn, mx, clip = bench.scene_for(scene)
vols = [Voxels(mn, mx, n, n, n) for _ in range(nvol)]
for i in range(2 * nvol):
    vols[i % nvol]._sample(sdf, clip=clip)
N.check(L.sdfk_profile_enable(2))       # the sampling kernel alone
t_w = time.perf_counter()
i = 0
while (time.perf_counter() - t_w) < 0.08:
    vols[i % nvol]._sample(sdf, clip=clip)
    i += 1
    if i % 32 == 0:
        torch.cuda.synchronize()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for i in range(k):
    vols[i % nvol]._sample(sdf, clip=clip)
e1.record(stream)
torch.cuda.synchronize()
N.check(L.sdfk_profile_enable(0))
us = e0.elapsed_time(e1) * 1e3 / k
nbytes = n ** 3 * (16 if sdf.writes_color else 4) + n ** 3 // 8
print(f"{scene} {n}^3 over {nvol} volumes: {us:.1f} us per launch, {nbytes / us / 1e6:.2f} TB/s = {nbytes / us / 1e6 / 8:.3f} of 8 TB/s  (JIT flags: {os.environ.get('SDFK_JIT_FLAGS', '')})")
