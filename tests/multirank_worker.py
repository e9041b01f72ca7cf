"""Launched by tests/test_gpu_multirank.py under torch.distributed.run: a pipelined sharded session (sdfk_dist_session_*, the
library's own step driver) over `world` ranks; every rank compares the whole mesh of every step with the oracle's mesh of the
whole grid, array by array.  Exit code 0 = identical on this rank.
  default               every rank on GPU 0, the HOST transport (RCCL refuses two ranks on one device; the exchange goes through
                        the gloo group, everything else is the production path)
  SDFK_TEST_REAL_RCCL=1 one GPU per rank (LOCAL_RANK), the library's own RCCL communicator: ncclAllGather, the grouped
                        ncclSend / ncclRecv exchange, gather-to-root and the 16-bit-index payloads BETWEEN GPUs
                        (SDFK_DIST_EXCHANGE / SDFK_DIST_INDEX16 of the environment select; needs >= world GPUs)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

REAL = os.environ.get("SDFK_TEST_REAL_RCCL") == "1"
if not REAL:
    os.environ["LOCAL_RANK"] = "0"
DEVICE = int(os.environ.get("LOCAL_RANK", "0"))
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
if REAL:
    D.init(device=DEVICE)
else:
    D.init_host(device=0)
OPT_MODE = N.get_option(N.OPT_DIST_EXCHANGE)
MODE = OPT_MODE if REAL else -1      # (the host transport reports -1; SDFK_DIST_EXCHANGE = 2 / 3 still mean "to rank 0 only" / "headers only")
HOLDS_MESH = not (OPT_MODE == 2 and dist.get_rank() != 0)   # gather-to-root: only rank 0 holds the whole mesh (with either transport)
graphs_on = N.get_option(N.OPT_GRAPHS) != 0 and N.get_option(N.OPT_DIST_LANES) != 0


def same(m, full=True):
    ok = np.array_equal(m.Triangles, om.triangles) and np.array_equal(m.Vertices, om.vertices)
    if full:
        ok &= (np.array_equal(m.Colors, om.colors) and np.array_equal(m.Normals, om.normals, equal_nan=True) and
               np.array_equal(m.Min, om.min) and np.array_equal(m.Max, om.max))
    return bool(ok)


ok = True
if HOLDS_MESH:
    ok = same(D.sharded_to_mesh(sdf, mn, mx, *dims))      # the one-off form first (sdfk_dist_to_mesh)
else:   # (collective: every rank makes the call; the ranks that hold no mesh get SDFK_ERR_UNSUPPORTED from the extraction)
    try:
        D.sharded_to_mesh(sdf, mn, mx, *dims)
        ok = False
    except N.SdfKitNativeError as e:
        ok = "only rank 0 holds the mesh" in str(e)
def own_slab_is_its_slice(ses):
    """sdfk_dist_slab_mesh: this rank's slab, indices global = the slice of the whole (oracle) mesh the headers' counts delimit"""
    counts, r = ses.counts(), dist.get_rank()
    vb, ib = sum(c[0] for c in counts[:r]), sum(c[1] for c in counts[:r])
    nv, ni = counts[r]
    sm = ses.slab_mesh()
    return bool(np.array_equal(sm.Vertices, om.vertices[vb:vb + nv]) and np.array_equal(sm.Triangles, om.triangles[ib:ib + ni]) and
                np.array_equal(sm.Colors, om.colors[vb:vb + nv]) and np.array_equal(sm.Normals, om.normals[vb:vb + nv], equal_nan=True))


ses = D.SlabSession(sdf, mn, mx, *dims, True, 0.0, depth=3)
for it in range(16):
    if ses.in_flight == ses.depth:
        ses.collect()
        if it % 3 == 0:
            ok &= own_slab_is_its_slice(ses)      # (before the whole mesh is asked for: exchange mode 3 has moved no payload yet)
            # a rank that received headers only (mode 3; mode 2 on a rank other than 0) must not hand out the gather buffer: the
            # foreign sections hold a fresh header followed by stale bytes
            if dist.get_world_size() > 1 and (OPT_MODE == 3 or not HOLDS_MESH):
                try:
                    ses.gathered()
                    ok = False
                except N.SdfKitNativeError as e:
                    ok &= "headers only" in str(e)
            elif dist.get_world_size() > 1:
                ok &= ses.gathered()[1] == ses.stats()["stride_bytes"]
        if HOLDS_MESH:
            ok &= same(ses.mesh())
        if it % 3 == 1:
            ok &= own_slab_is_its_slice(ses)      # (and after: the payloads have been gathered and rebased in place)
    ses.submit()
while ses.in_flight:
    ses.collect()
    if HOLDS_MESH:
        ok &= same(ses.mesh(), full=False)
counts = ses.counts()
ok &= sum(c[0] for c in counts) == len(om.vertices) and sum(c[1] for c in counts) == len(om.triangles)
import ctypes as C                      # noqa: E402
jobs, launches = C.c_int64(), C.c_int64()
N.check(N.lib().sdfk_graph_stats(C.byref(jobs), C.byref(launches), None))
if graphs_on:
    # the repeat steps were captured step graphs (one per slot and lane), replayed from their second use on
    ok &= jobs.value >= 1 and launches.value >= 1
st = ses.stats()
ok &= st["steps"] == 16 and st["exchange_mode"] == MODE and st["index16"] == (os.environ.get("SDFK_DIST_INDEX16") == "1")
ses.close()
dist.barrier()
D.shutdown()
dist.destroy_process_group()
print(f"rank {os.environ.get('RANK')}: {'identical' if ok else 'DIFFERENT'} ({len(om.vertices)} vertices, {st})")
sys.exit(0 if ok else 1)
