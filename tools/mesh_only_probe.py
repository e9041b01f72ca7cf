"""Throughput of the meshing chain alone (sign bits cached, corners re-evaluated) on a resident 512^3 sphere volume:
serial on one stream vs rotating over the library's lanes, several jobs in flight.  And of the sampling kernel alone.
(What part of a pipelined step is the meshing chain, once it overlaps with other jobs' chains?)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

from sdfkit_amd import Sdfs  # noqa: E402
from sdfkit_amd import _native as N  # noqa: E402
from sdfkit_amd.api import Voxels  # noqa: E402

n = int(os.environ.get("PROBE_N", "512"))
L = N.lib()
N.check(L.sdfk_init(0))
sdf = Sdfs.Sphere(1.0)
mn, mx = [-1.5] * 3, [1.5] * 3
vols = [Voxels(mn, mx, n, n, n) for _ in range(4)]
for v in vols:
    v._sample(sdf, clip=False)


def march(v, lane):
    if lane:
        N.check(L.sdfk_lane_begin(lane, None))
    m = C.c_void_p()
    try:
        N.check(L.sdfk_march(v._h, C.c_float(0.0), 1, C.byref(m)))
    finally:
        if lane:
            N.check(L.sdfk_lane_end(0))
    return m


def retire(m):
    a, b = C.c_int64(), C.c_int64()
    N.check(L.sdfk_mesh_counts(m, C.byref(a), C.byref(b)))
    L.sdfk_mesh_free(m)
    return a.value


def run(steps, lanes, depth):
    q = []
    for i in range(steps):
        q.append(march(vols[i % 4], 1 + i % lanes if lanes else 0))
        while len(q) > depth:
            retire(q.pop(0))
    while q:
        retire(q.pop(0))


for lanes, depth in ((0, 1), (0, 3), (2, 3), (3, 4), (4, 6), (4, 8)):
    run(300, lanes, depth)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(600, lanes, depth)
    torch.cuda.synchronize()
    print(f"meshing chain only, lanes {lanes} depth {depth}: {(time.perf_counter() - t0) / 600 * 1e6:.1f} us per job", flush=True)
