"""Runs a short pipelined sharded session (sdfk_dist_*, world = 1 over RCCL, small grid) for a rocprofv3 --kernel-trace: what
does the GPU timeline of a sharded step look like when the GPU work is negligible?  (tools/slab_step_timeline.py reads
the trace.)  PROBE_N = grid edge, PROBE_MODE = exchange mode, PROBE_LANES = SDFK_OPT_DIST_LANES."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdfkit_amd import _native as N, Sdfs
from sdfkit_amd import dist as D

if os.environ.get("PROBE_TORCH"):   # what bench.py has in the process: torch's runtime state, optionally its stream as lane 0
    import torch
    torch.cuda.set_device(0)
    torch.cuda.synchronize()
N.init(0)
L = N.lib()
if os.environ.get("PROBE_BIND"):
    N.bind_torch_stream(torch.device("cuda", 0))
buf = (C.c_ubyte * 128)()
N.check(L.sdfk_dist_unique_id(buf))
N.check(L.sdfk_dist_init(1, 0, buf))
n = int(os.environ.get("PROBE_N", "64"))
N.set_option(N.OPT_DIST_EXCHANGE, int(os.environ.get("PROBE_MODE", "1")))
N.set_option(N.OPT_DIST_LANES, int(os.environ.get("PROBE_LANES", "2")))
ses = D.SlabSession(Sdfs.Sphere(1.0), [-1.5] * 3, [1.5] * 3, n, n, n, False, 0.0, depth=int(os.environ.get("PROBE_DEPTH", "3")))
steps = int(os.environ.get("PROBE_STEPS", "60"))
for _ in range(12):
    if ses.in_flight == ses.depth:
        ses.collect()
    ses.submit()
ses.drain()
N.check(L.sdfk_synchronize())
st0 = ses.stats()
t0 = time.perf_counter()
for _ in range(steps):
    if ses.in_flight == ses.depth:
        ses.collect()
    ses.submit()
ses.drain()
dt = time.perf_counter() - t0
st = ses.stats()
print(f"n {n}: {dt / steps * 1e6:.1f} us per step; host submit {(st['host_ns_submit'] - st0['host_ns_submit']) / steps / 1e3:.1f} us, "
      f"collect {(st['host_ns_collect'] - st0['host_ns_collect']) / steps / 1e3:.1f} us; stride {st['stride_bytes']} mode {st['exchange_mode']} redone {st['redone']}")
ses.close()
D.shutdown()
