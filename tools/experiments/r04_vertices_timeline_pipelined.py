"""k_vertices' per-chunk timeline (library from r04_vertices_timeline.patch, -DSDFK_KV_TIME=1) while the library PIPELINES jobs:
five jobs in flight on the three lanes, as in the headline pass of bench.py -- what the chunks of the last launches look like
beside another job's sampling kernel.  (g_kt holds the stamps of whichever launch wrote a chunk last.)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from sdfkit_amd import _native as N
N.init(0)
L = N.lib()
n = 512
sdf, mn, mx, clip = bench.scene_for("sphere")
prog = sdf.program()
mn_c, mx_c = N.f3(mn), N.f3(mx)
infl = []
a, b = C.c_int64(), C.c_int64()
def retire(m):
    N.check(L.sdfk_mesh_counts(m, C.byref(a), C.byref(b))); L.sdfk_mesh_free(m)
for it in range(400):
    m = C.c_void_p()
    N.check(L.sdfk_sample_march(prog, mn_c, mx_c, n, n, n, 0, C.c_float(0.0), 1, C.byref(m)))
    infl.append(m)
    while len(infl) > 5: retire(infl.pop(0))
# read the stamps while the pipeline is still full: the last complete launches
buf = np.zeros(8192 * 8, dtype=np.int64)
while len(infl) > 2: retire(infl.pop(0))
assert L.sdfk_dbg_kt(buf.ctypes.data_as(C.POINTER(C.c_longlong)), buf.size) == 0
while infl: retire(infl.pop(0))
t = buf.reshape(-1, 8); t = t[t[:, 0] != 0]
us = lambda x: x / 100.0
# keep the chunks of ONE launch: those whose workgroup entered within 150 us of the latest entry
latest = t[:, 6].max(); sel = t[:, 6] > latest - 15000; t = t[sel]
k0 = t[:, 6].min()
print("chunks of the last launch(es):", len(t), " span %.1f us" % us(t[:, 5].max() - k0))
for name, d in (("entry->chunk start", t[:, 0] - t[:, 6]), ("stage (loads -> LDS)", t[:, 1] - t[:, 0]), ("chunk prefix", t[:, 2] - t[:, 1]), ("creator table", t[:, 3] - t[:, 2]),
                ("vertex loop (wave 0)", t[:, 4] - t[:, 3]), ("wait for the other waves", t[:, 5] - t[:, 4]), ("whole chunk", t[:, 5] - t[:, 0])):
    d = d[(t[:, 0] - t[:, 6]) < 3000] if name == "entry->chunk start" else d
    print("   %-26s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (name, us(d.mean()), us(np.percentile(d, 10)), us(np.percentile(d, 50)), us(np.percentile(d, 90)), us(d.max())))
cs, ce = us(t[:, 0] - k0), us(t[:, 5] - k0)
print("   chunks in progress every 5 us:", [int(((cs <= x) & (ce > x)).sum()) for x in np.arange(0, ce.max() + 5, 5.0)])
