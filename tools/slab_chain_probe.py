#!/usr/bin/env python3
"""GPU probe: one rank's share of an 8-rank (or N-rank) sharded step -- the slab chain of a 512^3 grid cut N ways -- without any
exchange: sdfk_slab_enqueue in a loop on 0..4 lanes (captured step graphs), and per kernel on one stream.  PROBE_WORLD, PROBE_RANK, PROBE_N, PROBE_LANES (lane counts to try, in order)."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdfkit_amd import _native as N, Sdfs
from sdfkit_amd import dist as D

N.init(0)
L = N.lib()
n = int(os.environ.get("PROBE_N", "512"))
world = int(os.environ.get("PROBE_WORLD", "8"))
rank = int(os.environ.get("PROBE_RANK", "3"))
sdf = Sdfs.Sphere(1.0)
mn, mx = N.f3([-1.5] * 3), N.f3([1.5] * 3)
lb, le, z0, nzl = D.slab(n, world, rank)
NBUF = 12   # (a multiple of every lane count tried: a buffer is always reused on the lane that used it last)
vols, bufs = [], []
for k in range(NBUF):
    v = C.c_void_p()
    N.check(L.sdfk_volume_create_slab(n, n, n, mn, mx, z0, nzl, 0, C.byref(v)))
    vols.append(v)
    bufs.append(torch.zeros(8 << 20, dtype=torch.uint8, device="cuda"))
prog = sdf.program()
iso = C.c_float(0.0)


def step(i, lane):
    k = i % NBUF
    N.check(L.sdfk_slab_enqueue(prog, vols[k], 0, iso, lb, le, C.c_void_p(bufs[k].data_ptr()), bufs[k].numel(), lane, None))


for lanes in [int(x) for x in os.environ.get("PROBE_LANES", "2,3,4,1,0").split(",")]:
    for i in range(4 * NBUF):   # (the captured step graph of every (buffer, lane) pair exists after this)
        step(i, (1 + i % lanes) if lanes else 0)
    N.check(L.sdfk_synchronize())
    torch.cuda.synchronize()
    steps = 300
    t0 = time.perf_counter()
    for i in range(steps):
        step(i, (1 + i % lanes) if lanes else 0)
    t1 = time.perf_counter()
    N.check(L.sdfk_synchronize())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"slab {n}x{n}x{nzl} (layers [{lb},{le}) of {world} ranks): {dt / steps * 1e6:.1f} us per step (host {(t1 - t0) / steps * 1e6:.1f}), lanes {lanes}")
N.check(L.sdfk_profile_reset())
N.check(L.sdfk_profile_enable(1))
for i in range(50):
    step(i, 0)
N.check(L.sdfk_synchronize())
N.check(L.sdfk_profile_enable(0))
prof = N.profile_snapshot()
print({k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in prof.items() if v[1]})
hdr = bufs[0][:16].cpu().numpy().view("int64")
print("counts", hdr)
