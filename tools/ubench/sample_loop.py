import sys, os, time
sys.path.insert(0, "/root/repo")
import ctypes as C, torch
from sdfkit_amd import _native as N, Sdfs, Voxels
N.init(); L = N.lib()
N.bind_torch_stream()
n = 512
vol = Voxels((-1.5,)*3, (1.5,)*3, n, n, n)
sdf = Sdfs.Sphere(1.0)
for _ in range(5): vol._sample(sdf)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 50
a.record()
for _ in range(K): vol._sample(sdf)
b.record(); torch.cuda.synchronize()
print("sample+transpose per call: %.2f us" % (a.elapsed_time(b) * 1e3 / K))
