#!/usr/bin/env python3
"""GPU probe: host<->device transfer rates of the C ABI (pageable caller arrays, as a C# shim
pins them): sdfk_mesh_copy, sdfk_volume_upload, sdfk_volume_download at 512^3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from sdfkit_amd import _native as N, Sdfs, Voxels

N.init(); L = N.lib()
n = 512
sdf = Sdfs.Sphere(1.0)
m = C.c_void_p()
for _ in range(2):
    N.check(L.sdfk_sample_march(sdf.program(), N.f3([-1.5] * 3), N.f3([1.5] * 3), n, n, n, 0, C.c_float(0.0), 1, C.byref(m)))
    nv, ni = C.c_int64(), C.c_int64()
    N.check(L.sdfk_mesh_counts(m, C.byref(nv), C.byref(ni)))
v = np.empty((nv.value, 3), np.float32); c = np.empty_like(v); nn = np.empty_like(v); t = np.empty(ni.value, np.int32)
for rep in range(3):
    t0 = time.perf_counter()
    N.check(L.sdfk_mesh_copy(m, v.ctypes.data, c.ctypes.data, nn.ctypes.data, t.ctypes.data))
    dt = time.perf_counter() - t0
    print(f"mesh_copy   {(v.nbytes*3+t.nbytes)/1e6:8.1f} MB  {dt*1e3:8.2f} ms  {(v.nbytes*3+t.nbytes)/dt/1e9:6.2f} GB/s")
vol = Voxels((-1.5,) * 3, (1.5,) * 3, n, n, n)
vol._sample(sdf)
host = np.empty((n, n, n), np.float32)
for rep in range(3):
    t0 = time.perf_counter()
    N.check(L.sdfk_volume_download(vol._h, host.ctypes.data, None))
    dt = time.perf_counter() - t0
    print(f"vol download {host.nbytes/1e6:8.1f} MB  {dt*1e3:8.2f} ms  {host.nbytes/dt/1e9:6.2f} GB/s")
for rep in range(3):
    t0 = time.perf_counter()
    N.check(L.sdfk_volume_upload(vol._h, host.ctypes.data, None))
    dt = time.perf_counter() - t0
    print(f"vol upload   {host.nbytes/1e6:8.1f} MB  {dt*1e3:8.2f} ms  {host.nbytes/dt/1e9:6.2f} GB/s")
