#!/usr/bin/env python3
"""GPU probe: time of the fused sampling kernel (sdfk_sample_bits) at 512^3 for SDFs of
different arithmetic cost -- separates the store-bound part from the VALU part."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
from sdfkit_amd import _native as N, Sdfs, SdfExprs, Voxels

N.init()
lib = N.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cases = {
    "plane_xy": Sdfs.PlaneXY(0.1),
    "sphere": Sdfs.Sphere(1.0),
    "box": Sdfs.Box((0.7, 0.8, 0.9)),
    "cylinder": Sdfs.Cylinder(0.6, 0.9),
}
for name, sdf in cases.items():
    vol = Voxels((-1.5,) * 3, (1.5,) * 3, n, n, n)
    for _ in range(3):
        vol._sample(sdf)
    lib.sdfk_synchronize()
    lib.sdfk_profile_enable(1)
    lib.sdfk_profile_reset()
    for _ in range(10):
        vol._sample(sdf)
    lib.sdfk_synchronize()
    snap = N.profile_snapshot()
    lib.sdfk_profile_enable(0)
    print(name, {k: round(ms * 1000.0 / max(c, 1), 2) for k, (ms, c) in snap.items()})
    vol._free()
