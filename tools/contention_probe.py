"""What slows the marching-cubes chain when another job's sampling kernel runs beside it -- the competition for instruction
issue or the saturated memory path?  The meshing chain alone on a resident 512^3 sphere volume (serial, one stream), timed
(a) alone, (b) beside a stream of plain 512 MiB fills (HBM write path saturated, almost no VALU), (c) beside a stream of
VALU-heavy kernels on a cache-resident tensor (sin/cos chains on 4 MB: no HBM traffic)."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdfkit_amd import Sdfs
from sdfkit_amd import _native as N
from sdfkit_amd.api import Voxels

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
L = N.lib()
N.check(L.sdfk_init(0))
sdf = Sdfs.Sphere(1.0)
mn, mx, n = [-1.5] * 3, [1.5] * 3, 512
vol = Voxels(mn, mx, n, n, n)
vol._sample(sdf, clip=False)


def mesh_chain(k):
    for _ in range(k):
        m = C.c_void_p()
        N.check(L.sdfk_march(vol._h, C.c_float(0.0), 1, C.byref(m)))
        a, b = C.c_int64(), C.c_int64()
        N.check(L.sdfk_mesh_counts(m, C.byref(a), C.byref(b)))
        L.sdfk_mesh_free(m)


side = torch.cuda.Stream(dev)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
small = torch.rand(1 << 20, device=dev)


def background(kind, stop_after_s):
    t_end = time.perf_counter() + stop_after_s
    with torch.cuda.stream(side):
        while time.perf_counter() < t_end:
            if kind == "fill":
                for _ in range(8):
                    big.fill_(1)
            else:
                y = small
                for _ in range(8):
                    y = torch.sin(torch.cos(torch.sin(torch.cos(y))))
            # keep at most ~1 ms queued
            ev = torch.cuda.Event(); ev.record(side); ev.synchronize()


import threading
mesh_chain(20)
torch.cuda.synchronize()
for kind in ("alone", "fill", "valu"):
    th = None
    if kind != "alone":
        th = threading.Thread(target=background, args=(kind, 1.5)); th.start()
        time.sleep(0.3)
    t0 = time.perf_counter()
    mesh_chain(300)
    dt = (time.perf_counter() - t0) / 300
    if th:
        th.join()
    torch.cuda.synchronize()
    print(f"meshing chain, one call at a time, {kind:6s}: {dt * 1e6:7.1f} us per job", flush=True)
