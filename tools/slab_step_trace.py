"""Runs a short pipelined SlabSession (world = 1 over RCCL, small grid) for a rocprofv3 --kernel-trace: what does the GPU
timeline of a sharded step look like when the GPU work is negligible?  (tools/slab_step_timeline.py reads the trace.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
n = int(os.environ.get("PROBE_N", "64"))
ses = D.SlabSession(Sdfs.Sphere(1.0), [-1.5] * 3, [1.5] * 3, n, n, n, False, 0.0, None, dev, depth=3)
for _ in range(60):
    if len(ses.queue) == ses.depth:
        ses.collect()
    ses.submit()
ses.drain()
torch.cuda.synchronize()
ses.close()
dist.destroy_process_group()
