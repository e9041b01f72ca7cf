#!/usr/bin/env python3
"""GPU probe: CreateMesh on a volume that did NOT come from the fused sampler (device copy of a
sampled volume whose cached views are dropped): per-kernel times of the non-fused path at 512^3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
from sdfkit_amd import _native as N, Sdfs, Voxels, MarchingCubes
N.init(); L = N.lib()
n = 512
vol = Voxels.SampleSdf(Sdfs.Sphere(1.0), (-1.5,) * 3, (1.5,) * 3, n, n, n)
pv, pc = C.c_void_p(), C.c_void_p()
N.check(L.sdfk_volume_device_ptrs(vol._h, C.byref(pv), C.byref(pc)))   # drops the cached bits / program
for it in range(6):
    if it == 2:
        N.check(L.sdfk_profile_reset()); N.check(L.sdfk_profile_enable(1))
    m = C.c_void_p()
    N.check(L.sdfk_march(vol._h, C.c_float(0.0), 1, C.byref(m)))
    a, b = C.c_int64(), C.c_int64()
    N.check(L.sdfk_mesh_counts(m, C.byref(a), C.byref(b)))
    L.sdfk_mesh_free(m)
N.check(L.sdfk_profile_enable(0))
print(a.value, b.value, {k: round(ms * 1000 / max(c, 1), 1) for k, (ms, c) in N.profile_snapshot().items()})
