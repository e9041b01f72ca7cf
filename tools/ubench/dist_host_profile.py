import os, sys, time
sys.path.insert(0, "/root/repo")
import ctypes as C
import torch, torch.distributed as dist
from sdfkit_amd import _native as N, Sdfs
from sdfkit_amd import dist as D
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
N.init(0); L = N.lib()
N.bind_torch_stream()
n = 64
ses = D.SlabSession(Sdfs.Sphere(1.0), [-1.5]*3, [1.5]*3, n, n, n, False, 0.0, None, dev, depth=3)
for _ in range(10):
    if len(ses.queue) == ses.depth: ses.collect()
    ses.submit()
ses.drain(); torch.cuda.synchronize()
import collections
T = collections.defaultdict(float)
def timed(name, f):
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[name] += time.perf_counter() - t; return r
    return g
w = ses.workers
for x in w:
    x.enqueue = timed("enqueue", x.enqueue)
ses._start_gather = timed("start_gather", ses._start_gather)
ses._finish_gather = timed("finish_gather", ses._finish_gather)
ses._headers = timed("headers", ses._headers)
ses.rebase = timed("rebase(in finish)", ses.rebase)
K = 300
t0 = time.perf_counter()
for _ in range(K):
    if len(ses.queue) == ses.depth: ses.collect()
    ses.submit()
ses.drain(); torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"total {tot/K*1e6:.1f} us/step")
for k, v in T.items(): print(f"  {k:20s} {v/K*1e6:7.1f} us")
# inside enqueue: the three C calls
m = C.c_void_p()
x = w[0]
t=time.perf_counter()
for _ in range(200):
    x.release()
    N.check(L.sdfk_sample_march_slab(x.prog, x.vol, 0, C.c_float(0.0), x.lb, x.le, 0, C.byref(m)))
    x.mesh = m
torch.cuda.synchronize()
print(f"  sample_march_slab+free alone {(time.perf_counter()-t)/200*1e6:.1f} us")
dist.destroy_process_group()
