"""How long does the GPU take to reach its sustained step rate after an idle period?  Queues the headline step
(512^3 sphere, three in flight) for a while and prints the mean step time per window of 20 retired steps, after idle
gaps of different lengths.  (Why: the driver's bench run times 20 steps = 4 ms after 5 warm-up steps.)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (device memory / context like bench.py)

from sdfkit_amd import Sdfs  # noqa: E402
from sdfkit_amd import _native as N  # noqa: E402

n = int(os.environ.get("PROBE_N", "512"))
L = N.lib()
N.check(L.sdfk_init(0))
sdf = Sdfs.Sphere(1.0)
prog = sdf.program()
mn, mx = [-1.5] * 3, [1.5] * 3


def submit():
    m = C.c_void_p()
    N.check(L.sdfk_sample_march(prog, N.f3(mn), N.f3(mx), n, n, n, 0, C.c_float(0.0), 1, C.byref(m)))
    return m


def retire(m):
    a, b = C.c_int64(), C.c_int64()
    N.check(L.sdfk_mesh_counts(m, C.byref(a), C.byref(b)))
    L.sdfk_mesh_free(m)


def run(steps, depth=3):
    q, stamps = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        q.append(submit())
        while len(q) > depth:
            retire(q.pop(0)); stamps.append(time.perf_counter() - t0)
    while q:
        retire(q.pop(0)); stamps.append(time.perf_counter() - t0)
    return stamps


for _ in range(3):
    retire(submit())
for idle in (0.0, 0.002, 0.02, 0.2, 1.0):
    torch.cuda.synchronize()
    time.sleep(idle)
    s = run(400)
    w = [(s[min(i + 20, len(s)) - 1] - (s[i - 1] if i else 0.0)) / 20 * 1e3 for i in range(0, 400, 20)]
    print(f"idle {idle * 1e3:6.0f} ms: first 20 steps {w[0]:.4f} ms/step, then " + " ".join(f"{x:.3f}" for x in w[1:]), flush=True)
