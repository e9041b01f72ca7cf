#!/usr/bin/env python3
"""Per-kernel times of sdfk_sample_march with SDFK_OPT_ELIDE_VOLUME = 0 / 1 / 2 (one in-order stream, HIP events), 512^3."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sdfkit_amd import _native as N

N.init(0)
L = N.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for scene in ("sphere", "repeatxy"):
    sdf, mn, mx, clip = bench.scene_for(scene)
    for mode in (0, 1, 2):
        with N.option(N.OPT_ELIDE_VOLUME, mode), N.option(N.OPT_LANES, 0):
            for _ in range(6):
                sdf.ToMesh(mn, mx, n, n, n, clipToBounds=clip).Recycle()
            N.check(L.sdfk_profile_reset()); N.check(L.sdfk_profile_enable(1))
            for _ in range(10):
                sdf.ToMesh(mn, mx, n, n, n, clipToBounds=clip).Recycle()
            N.check(L.sdfk_profile_enable(0))
            prof = N.profile_snapshot()
            print(scene, "elide", mode, {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items() if v[1]}, flush=True)
