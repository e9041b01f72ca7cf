"""Timeline inside k_vertices (library built from r04_vertices_timeline.patch with -DSDFK_KV_TIME=1): per chunk, the 100 MHz wall
clock at: 0 chunk start, 1 staging loads landed + stored to LDS, 2 chunk prefix known, 3 scan + creator table done, 4 this
wavefront's vertices done, 5 whole workgroup done; 6 = the workgroup's kernel entry, 7 = HW_ID | XCC_ID << 32."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
from sdfkit_amd import _native as N
N.init(0)
L = N.lib()
for scene, n in (("sphere", 512), ("repeatxy", 512)):
    sdf, mn, mx, clip = bench.scene_for(scene)
    with N.option(N.OPT_LANES, 0):
        for _ in range(3):
            m = sdf.ToMesh(mn, mx, n, n, n, clipToBounds=clip)
        buf = np.zeros(8192 * 8, dtype=np.int64)
        assert L.sdfk_dbg_kt(buf.ctypes.data_as(C.POINTER(C.c_longlong)), buf.size) == 0
    t = buf.reshape(-1, 8)
    t = t[t[:, 0] != 0]
    k0 = t[:, 6].min()
    us = lambda a: a / 100.0
    print(scene, n, "chunks", len(t), "kernel span %.1f us" % us(t[:, 5].max() - k0))
    ph = [("entry->chunk start", t[:, 0] - t[:, 6]), ("phase 0->1", t[:, 1] - t[:, 0]), ("phase 1->2", t[:, 2] - t[:, 1]), ("phase 2->3", t[:, 3] - t[:, 2]),
          ("main loop (wave 0)", t[:, 4] - t[:, 3]), ("wait for the other waves", t[:, 5] - t[:, 4]), ("whole chunk", t[:, 5] - t[:, 0])]
    for name, d in ph:
        print("   %-26s mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us" % (name, us(d.mean()), us(np.percentile(d, 10)), us(np.percentile(d, 50)), us(np.percentile(d, 90)), us(d.max())))
    start = us(t[:, 6] - k0)
    end = us(t[:, 5] - k0)
    print("   workgroup entry times: p50 %.1f p90 %.1f max %.1f us;  ends: p10 %.1f p50 %.1f p90 %.1f max %.1f us" % (
        np.percentile(start, 50), np.percentile(start, 90), start.max(), np.percentile(end, 10), np.percentile(end, 50), np.percentile(end, 90), end.max()))
    e_us = us(t[:, 6] - k0); d_us = us(t[:, 0] - t[:, 6]); first = t[:, 0] - t[:, 6] < 40 * 100   # (a workgroup's first chunk: start soon after entry)
    for lo, hi in ((0, 1), (1, 3), (3, 8), (8, 15), (15, 25), (25, 60)):
        sel = (e_us >= lo) & (e_us < hi) & (t[:, 0] - t[:, 6] < 3000)
        if sel.any():
            print("   workgroups entering at %2d-%2d us: %4d chunks, entry->start p10 %.2f p50 %.2f p90 %.2f max %.2f;  stage p50 %.2f  vertex loop p50 %.2f" % (
                lo, hi, sel.sum(), np.percentile(d_us[sel], 10), np.percentile(d_us[sel], 50), np.percentile(d_us[sel], 90), d_us[sel].max(),
                np.percentile(us(t[sel, 1] - t[sel, 0]), 50), np.percentile(us(t[sel, 4] - t[sel, 3]), 50)))
    cs, ce = us(t[:, 0] - k0), us(t[:, 5] - k0)
    print("   chunks in progress every 2 us:", [int(((cs <= b) & (ce > b)).sum()) for b in np.arange(0, ce.max() + 2, 2.0)])
    # occupancy over time: chunks in flight per 2 us bin
    bins = np.arange(0, end.max() + 2, 2.0)
    inflight = [(int(((start <= b) & (end > b)).sum())) for b in bins]
    print("   workgroups in flight every 2 us:", inflight)
    hw = t[:, 7]
    xcc = (hw >> 32) & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = xcc * 1000 + se * 100 + sh * 16 + cu
    u, c = np.unique(key, return_counts=True)
    print("   distinct (xcc,se,sh,cu):", len(u), " chunks per CU: min %d p50 %d max %d" % (c.min(), np.median(c), c.max()), " per XCC:", np.bincount(xcc.astype(int)).tolist())
