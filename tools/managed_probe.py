#!/usr/bin/env python3
"""GPU probe: what a managed-array caller (C#: Mesh.cs:10-13 are `new Vector3[n]` / `new int[n]`) pays per call at 512^3, for the
ways the hand-off can be arranged:
  A  sample_march -> counts (wait) -> allocate fresh arrays -> mesh_copy
  B  sample_march -> size_hint (no wait) -> allocate fresh arrays -> host_prefault each (GPU still busy) -> counts -> mesh_copy
each with SDFK_OPT_PREFAULT_HUGE 0 / 1 (2 MiB blocks of the destination advised MADV_HUGEPAGE before first touch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import mmap
import numpy as np
from sdfkit_amd import _native as N, Sdfs

N.init(); L = N.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sdf = Sdfs.Sphere(1.0)
prog = sdf.program()
mn, mx = N.f3([-1.5] * 3), N.f3([1.5] * 3)
try:
    print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:",
          open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip(), "| cpus:", os.cpu_count())
except OSError:
    pass


def fresh(nbytes):
    mm = mmap.mmap(-1, max(nbytes, 1), flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    mm.madvise(mmap.MADV_NOHUGEPAGE)      # a managed runtime's arrays: plain 4 KiB pages unless somebody says otherwise
    return mm, np.frombuffer(mm, dtype=np.uint8, count=nbytes)


def call(kind):
    N.check(L.sdfk_synchronize())
    t0 = time.perf_counter()
    m = C.c_void_p()
    N.check(L.sdfk_sample_march(prog, mn, mx, n, n, n, 0, C.c_float(0.0), 1, C.byref(m)))
    a, b, ex = C.c_int64(), C.c_int64(), C.c_int32()
    if kind == "B":
        N.check(L.sdfk_mesh_size_hint(m, C.byref(a), C.byref(b), C.byref(ex)))
        keep = [fresh(a.value * 12) for _ in range(3)] + [fresh(b.value * 4)]
        for _, x in keep:
            N.check(L.sdfk_host_prefault(x.ctypes.data, x.nbytes))
        t_alloc = time.perf_counter()
        ha, hb = a.value, b.value
        N.check(L.sdfk_mesh_counts(m, C.byref(a), C.byref(b)))
        assert (a.value, b.value) == (ha, hb)
    else:
        N.check(L.sdfk_mesh_counts(m, C.byref(a), C.byref(b)))
        keep = [fresh(a.value * 12) for _ in range(3)] + [fresh(b.value * 4)]
        t_alloc = time.perf_counter()
    t_counts = time.perf_counter()
    arrs = [x for _, x in keep]
    N.check(L.sdfk_mesh_copy(m, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data))
    t_copy = time.perf_counter()
    lo, hi = (C.c_float * 3)(), (C.c_float * 3)()
    N.check(L.sdfk_mesh_bounds(m, lo, hi))
    L.sdfk_mesh_free(m)
    t1 = time.perf_counter()
    st = (C.c_int64 * 5)()
    L.sdfk_copy_stats(st)
    call.last = f"copy call {(t_copy - t_counts) * 1e3:.2f} ms: queued {st[1] / 1e6:.2f} present {st[2] / 1e6:.2f} done {st[3] / 1e6:.2f} (dma wait {st[4] / 1e6:.2f}); bounds+free {(t1 - t_copy) * 1e3:.2f}"
    return (t1 - t0) * 1e3, (t_alloc - t0) * 1e3, (t_counts - t0) * 1e3


reuse = None


def call_reuse():
    """floor: the same (touched) arrays every call"""
    global reuse
    N.check(L.sdfk_synchronize())
    t0 = time.perf_counter()
    m = C.c_void_p()
    N.check(L.sdfk_sample_march(prog, mn, mx, n, n, n, 0, C.c_float(0.0), 1, C.byref(m)))
    a, b = C.c_int64(), C.c_int64()
    N.check(L.sdfk_mesh_counts(m, C.byref(a), C.byref(b)))
    tc = time.perf_counter()
    if reuse is None:
        reuse = [fresh(a.value * 12) for _ in range(3)] + [fresh(b.value * 4)]
    arrs = [x for _, x in reuse]
    N.check(L.sdfk_mesh_copy(m, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data))
    L.sdfk_mesh_free(m)
    t1 = time.perf_counter()
    return (t1 - t0) * 1e3, (tc - t0) * 1e3, (tc - t0) * 1e3


for _ in range(3):
    call("A")
for mode in (1, 0, 2):
    N.set_option(N.OPT_COPY_MODE, mode)
    N.set_option(N.OPT_PREFAULT_HUGE, 0)
    ts = [call_reuse() for _ in range(7)]
    print(f"mode {mode} reused arrays: copy alone median {np.median([t[0] - t[1] for t in ts[1:]]):.3f} ms  all {[round(t[0] - t[1], 2) for t in ts]}")
    ts = [call("A") for _ in range(7)]
    print(f"mode {mode} fresh arrays (A): copy alone median {np.median([t[0] - t[2] for t in ts[1:]]):.3f} ms  total {np.median([t[0] for t in ts[1:]]):.3f}")
N.set_option(N.OPT_COPY_MODE, 1)
for huge in (0, 1):
    N.set_option(N.OPT_PREFAULT_HUGE, huge)
    for kind in ("A", "B"):
        ts = [call(kind) for _ in range(7)]
        tot = sorted(t[0] for t in ts[1:])
        print("   ", call.last)
        print(f"huge {huge} {kind}: median {tot[len(tot) // 2]:.3f} ms  min {tot[0]:.3f}  (until arrays ready {np.median([t[1] for t in ts[1:]]):.3f}, until counts {np.median([t[2] for t in ts[1:]]):.3f})  all {[round(t[0], 2) for t in ts]}")
for threads_note in ():
    pass
