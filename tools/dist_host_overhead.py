#!/usr/bin/env python3
"""GPU probe: host-side cost of one pipelined SlabSession step (world = 1 over RCCL) on a small
grid, where the GPU work is negligible -- what bounds the 8-GPU step rate from the host side."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
import torch.distributed as dist
from sdfkit_amd import _native as N, Sdfs
from sdfkit_amd import dist as D

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
N.init(0)
N.bind_torch_stream()
for n in (64, 128, 512):
    ses = D.SlabSession(Sdfs.Sphere(1.0), [-1.5] * 3, [1.5] * 3, n, n, n, False, 0.0, None, dev, depth=3)
    for _ in range(10):
        if len(ses.queue) == ses.depth:
            ses.collect()
        ses.submit()
    ses.drain()
    torch.cuda.synchronize()
    K = 200
    t0 = time.perf_counter()
    for _ in range(K):
        if len(ses.queue) == ses.depth:
            ses.collect()
        ses.submit()
    ses.drain()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
    N.check(N.lib().sdfk_graph_stats(C.byref(a), C.byref(b), C.byref(c)))
    print(f"n={n}: {dt*1e6:.1f} us per step (host + GPU, pipelined); captured step graphs alive {a.value}, graph launches so far {b.value}, steps redone {ses.redone}")
    ses.close()
dist.destroy_process_group()
