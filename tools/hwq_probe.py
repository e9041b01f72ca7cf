"""Why does the number of hardware queues (GPU_MAX_HW_QUEUES) help one process and hurt another?  The pipelined sharded
step at world = 1 (RCCL), small grid, after PROBE_PRE single-GPU jobs (which use lanes 1..3) and with PROBE_EXTRA extra idle
torch streams created first."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from sdfkit_amd import _native as N, Sdfs
from sdfkit_amd import dist as D

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
extra = [torch.cuda.Stream(dev) for _ in range(int(os.environ.get("PROBE_EXTRA", "0")))]
for s in extra:
    with torch.cuda.stream(s):
        torch.zeros(4, device=dev)
order = os.environ.get("PROBE_ORDER", "dist_first")
if order == "dist_first":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
N.init(0)
N.bind_torch_stream()
sdf = Sdfs.Sphere(1.0)
for _ in range(int(os.environ.get("PROBE_PRE", "0"))):
    sdf.ToMesh([-1.5] * 3, [1.5] * 3, 96, 96, 96, clipToBounds=False)
if order != "dist_first":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = 128
ses = D.SlabSession(sdf, [-1.5] * 3, [1.5] * 3, n, n, n, False, 0.0, None, dev, depth=3)
for _ in range(10):
    if len(ses.queue) == ses.depth:
        ses.collect()
    ses.submit()
ses.drain()
torch.cuda.synchronize()
K = 300
t0 = time.perf_counter()
for _ in range(K):
    if len(ses.queue) == ses.depth:
        ses.collect()
    ses.submit()
ses.drain()
torch.cuda.synchronize()
print(f"queues {os.environ.get('GPU_MAX_HW_QUEUES')} pre {os.environ.get('PROBE_PRE', '0')} extra {os.environ.get('PROBE_EXTRA', '0')} {order}: {(time.perf_counter() - t0) / K * 1e6:.1f} us per step", flush=True)
ses.close()
dist.destroy_process_group()
