#!/usr/bin/env python3
"""GPU probe: host<->device transfer rates of the C ABI with pageable caller arrays (what the C# shim
pins for a call): sdfk_mesh_copy into FRESH (never touched) and into already-touched arrays for each copy
strategy (SDFK_OPT_COPY_MODE 0 / 1 / 2) and pool size, sdfk_volume_download / upload at 512^3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import mmap
import numpy as np
from sdfkit_amd import _native as N, Sdfs, Voxels

N.init(); L = N.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sdf = Sdfs.Sphere(1.0)
m = C.c_void_p()
for _ in range(2):
    N.check(L.sdfk_sample_march(sdf.program(), N.f3([-1.5] * 3), N.f3([1.5] * 3), n, n, n, 0, C.c_float(0.0), 1, C.byref(m)))
    nv, ni = C.c_int64(), C.c_int64()
    N.check(L.sdfk_mesh_counts(m, C.byref(nv), C.byref(ni)))
try:
    print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| cpus:", os.cpu_count())
except OSError:
    pass


def fresh(nbytes):
    """an anonymous mapping nobody has touched (what a brand-new managed / numpy array is)"""
    mm = mmap.mmap(-1, nbytes + 4096, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)   # (Python's default is MAP_SHARED: shmem pages)
    if os.environ.get("PROBE_HUGE") == "1":
        mm.madvise(mmap.MADV_HUGEPAGE)      # what numpy does for its large arrays
    else:
        mm.madvise(mmap.MADV_NOHUGEPAGE)
    return mm, np.frombuffer(mm, dtype=np.uint8, count=nbytes)


vb, tb = nv.value * 12, ni.value * 4
for mode in ("2", "0", "1"):
    N.set_option(N.OPT_COPY_MODE, int(mode))
    for touched in (False, True):
        ts = []
        for rep in range(4):
            keep = [fresh(vb), fresh(vb), fresh(vb), fresh(tb)]
            arrs = [a for _, a in keep]
            if touched:
                for a in arrs:
                    a[::4096] = 1
            t0 = time.perf_counter()
            N.check(L.sdfk_mesh_copy(m, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data))
            ts.append(time.perf_counter() - t0)
            del arrs, keep
        tot = vb * 3 + tb
        print(f"mesh_copy mode {mode} {'touched' if touched else 'fresh  '} {tot/1e6:7.1f} MB  best {min(ts)*1e3:7.2f} ms  ({tot/min(ts)/1e9:5.1f} GB/s)  all {[round(t*1e3, 2) for t in ts]}")
N.set_option(N.OPT_COPY_MODE, 1)
vol = Voxels((-1.5,) * 3, (1.5,) * 3, n, n, n)
vol._sample(sdf)
for mode in ("2", "0", "1"):
    N.set_option(N.OPT_COPY_MODE, int(mode))
    for touched in (False, True):
        ts = []
        for rep in range(2):
            mm, host = fresh(n * n * n * 4)
            if touched:
                host[::4096] = 1
            t0 = time.perf_counter()
            N.check(L.sdfk_volume_download(vol._h, host.ctypes.data, None))
            ts.append(time.perf_counter() - t0)
            del host, mm
        print(f"vol download mode {mode} {'touched' if touched else 'fresh  '} {n**3*4/1e6:8.1f} MB  best {min(ts)*1e3:8.2f} ms  {n**3*4/min(ts)/1e9:6.2f} GB/s")
N.set_option(N.OPT_COPY_MODE, 1)
host = np.ones((n, n, n), np.float32)
for rep in range(3):
    t0 = time.perf_counter()
    N.check(L.sdfk_volume_upload(vol._h, host.ctypes.data, None))
    dt = time.perf_counter() - t0
    print(f"vol upload   {host.nbytes/1e6:8.1f} MB  {dt*1e3:8.2f} ms  {host.nbytes/dt/1e9:6.2f} GB/s")
