#!/usr/bin/env python3
"""BASELINE config C5 (stretch): RayMarcher 1920x1080, 256 depth iterations, README RepeatXY
scene, camera (-2,2,4) -> origin (Perf/Program.cs:43-62 convention: discard the first loop).
Prints one JSON line: Mrays/s, SDF evaluations/s, ms per frame (device-resident images)."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sdfkit_amd import Matrix4x4, RayMarcher, SdfExprs, Vec3
from sdfkit_amd import _native as N

w, h, iters, loops = 1920, 1080, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 10
sdf = SdfExprs.Sphere(0.5).RepeatXY(1.125, 1.125, lambda i, p, d: 0.9 * Vec3.of(p.x.b, 1.0) - Vec3.Abs(i) / 6.0).ToSdf()
rm = RayMarcher(w, h, sdf)
rm.DepthIterations = iters
rm.ViewTransform = Matrix4x4.CreateLookAt((-2, 2, 4), (0, 0, 0), (0, 1, 0))
pos, vpi = rm.camera()
N.init()
L = N.lib()
rgb = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
N.bind_torch_stream()
args = (sdf.program(), w, h, N.f3(pos), (C.c_float * 16)(*[float(x) for x in vpi.ravel()]), C.c_float(1.0), C.c_float(100.0), iters,
        None, C.c_void_p(rgb.data_ptr()))
N.check(L.sdfk_raymarch_device(*args))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(loops):
    N.check(L.sdfk_raymarch_device(*args))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / loops
evals = w * h * (iters + 6)
print(json.dumps({"metric": "RayMarcher.Render 1920x1080x256, RepeatXY scene", "ms_per_frame": round(dt * 1e3, 3),
                  "mrays_per_s": round(w * h / dt / 1e6, 1), "gevals_per_s": round(evals / dt / 1e9, 2), "loops": loops,
                  "checksum": float(torch.nan_to_num(rgb.double()).sum().item())}))
