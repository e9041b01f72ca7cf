"""Launched by tests/test_gpu_multirank.py under torch.distributed.run (every rank on GPU 0, gloo):
pipelined SlabSession over `world` ranks; every rank compares the gathered, rebased mesh with the
oracle's mesh of the whole grid, array by array.  Exit code 0 = identical on this rank."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

os.environ["LOCAL_RANK"] = "0"
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from oracle import oracle as O          # noqa: E402
from sdfkit_amd import _native as N     # noqa: E402
from sdfkit_amd import dist as D        # noqa: E402
from tests import scenes as S           # noqa: E402

name, dims = sys.argv[1], tuple(int(x) for x in sys.argv[2:5])
scene, sdf = S.CATALOGUE[name]()
mn, mx = [-2.8125] * 3, [2.8125] * 3
ov, oc = O.sample(scene, mn, mx, *dims)
O.clip_to_bounds(ov, mn, mx)
om = O.march(ov, oc, mn, mx)
N.init(0)
ses = D.SlabSession(sdf, mn, mx, *dims, True, 0.0, None, torch.device("cuda", 0), depth=3)
ok = True
for it in range(16):
    if len(ses.queue) == ses.depth:
        ses.collect()
        m = ses.mesh()
        ok &= (np.array_equal(m.Triangles, om.triangles) and np.array_equal(m.Vertices, om.vertices) and
               np.array_equal(m.Colors, om.colors) and np.array_equal(m.Normals, om.normals, equal_nan=True) and
               np.array_equal(m.Min, om.min) and np.array_equal(m.Max, om.max))
    ses.submit()
while ses.queue:
    ses.collect()
    m = ses.mesh()
    ok &= np.array_equal(m.Triangles, om.triangles) and np.array_equal(m.Vertices, om.vertices)
import ctypes as C
jobs, launches = C.c_int64(), C.c_int64()
N.check(N.lib().sdfk_graph_stats(C.byref(jobs), C.byref(launches), None))
if os.environ.get("SDFK_GRAPHS", "1") != "0" and os.environ.get("SDFK_LANES", "2") not in ("0", "1"):
    # the repeat steps were captured step graphs (one per slot and lane), replayed from their second use on
    ok &= jobs.value >= 1 and launches.value >= 1
ses.close()
dist.barrier()
dist.destroy_process_group()
print(f"rank {os.environ.get('RANK')}: {'identical' if ok else 'DIFFERENT'} ({len(om.vertices)} vertices)")
sys.exit(0 if ok else 1)
