import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import bench
from sdfkit_amd import _native as N
N.init(0)
L = N.lib()
names = ["vertex wave-iterations", "  lanes", "wave-iters with e==12", "  lanes", "sharer wave-iterations", "  lanes", "sharer iters with a window miss", "  lanes",
         "bsearch trips (wave)", "  lanes", "sharer iters with 'no cell' lanes", "  lanes", "sharer iters with global-path lanes", "  lanes", "misses outside both row slices", "  lanes"]
for scene, n in (("sphere", 512), ("repeatxy", 512), ("sphere", 256)):
    sdf, mn, mx, clip = bench.scene_for(scene)
    with N.option(N.OPT_LANES, 0):
        sdf.ToMesh(mn, mx, n, n, n, clipToBounds=clip)
        out = (C.c_ulonglong * 16)()
        L.sdfk_dbg_kv(out, 1)
        m = sdf.ToMesh(mn, mx, n, n, n, clipToBounds=clip)
        nv = len(m.Vertices)
        L.sdfk_dbg_kv(out, 1)
    print(scene, n, "vertices", nv)
    for i, nm in enumerate(names): print("   %-40s %d" % (nm, out[i]))
