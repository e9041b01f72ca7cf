"""A short pipelined run of the headline step (512^3 sphere, three lanes, five jobs in flight) for a rocprofv3
--kernel-trace: how do the kernels of consecutive jobs interleave on the GPU?  (tools/slab_step_timeline.py prints it.)"""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sdfkit_amd import Sdfs
from sdfkit_amd import _native as N

n = int(os.environ.get("PROBE_N", "512"))
torch.cuda.set_device(0)
torch.zeros(4, device="cuda")
L = N.lib()
N.check(L.sdfk_init(0))
sdf = Sdfs.Sphere(1.0)   # (kept alive: the handle dies with the object)
prog = sdf.program()
mn, mx = [-1.5] * 3, [1.5] * 3
q = []
for i in range(150):
    m = C.c_void_p()
    N.check(L.sdfk_sample_march(prog, N.f3(mn), N.f3(mx), n, n, n, 0, C.c_float(0.0), 1, C.byref(m)))
    q.append(m)
    while len(q) > 5:
        h = q.pop(0)
        a, b = C.c_int64(), C.c_int64()
        N.check(L.sdfk_mesh_counts(h, C.byref(a), C.byref(b)))
        L.sdfk_mesh_free(h)
for h in q:
    L.sdfk_mesh_free(h)
torch.cuda.synchronize()
