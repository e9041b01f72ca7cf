#!/usr/bin/env python3
"""GPU probe: host time of one sdfk_sample_march call in the steady state (three jobs in flight, counts read three
steps late), for small grids where the step is bound by the host: python3 tools/host_cost_probe.py [grid edge]."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.getcwd())
import sdfkit_amd as S
from sdfkit_amd import _native as N
N.init(); L = N.lib()
sdf = S.Sdfs.Sphere(1.0); prog = sdf.program()
mn, mx = N.f3([-1.5]*3), N.f3([1.5]*3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
hs = []
def one():
    h = C.c_void_p()
    N.check(L.sdfk_sample_march(prog, mn, mx, n, n, n, 0, C.c_float(0.0), 1, C.byref(h)))
    return h
for _ in range(20): hs.append(one())
nv = C.c_int64(); ni = C.c_int64()
for h in hs: N.check(L.sdfk_mesh_counts(h, C.byref(nv), C.byref(ni))); L.sdfk_mesh_free(h)
hs = []
K = 2000
t0 = time.perf_counter(); tc = 0.0
for i in range(K):
    t1 = time.perf_counter()
    hs.append(one())
    tc += time.perf_counter() - t1
    if len(hs) > 3:
        h = hs.pop(0)
        N.check(L.sdfk_mesh_counts(h, C.byref(nv), C.byref(ni))); L.sdfk_mesh_free(h)
N.check(L.sdfk_synchronize())
dt = time.perf_counter() - t0
print(f"{n}^3: {dt/K*1e6:.1f} us per step, of which sdfk_sample_march call {tc/K*1e6:.1f} us; nv={nv.value}")
