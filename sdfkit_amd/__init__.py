"""sdfkit_amd -- MI355X (gfx950) implementation of SdfKit's Voxels.SampleSdf ->
MarchingCubes.CreateMesh hot path behind the reference's Sdf / Voxels / Mesh API.

`csrc/` holds the hand-written HIP kernels and the C ABI (include/sdfkit_hip.h);
`api` mirrors the reference's public types over that ABI.  There is no CPU path.
"""
from .api import (DefaultBatchSize, MarchingCubes, Mesh, Sdf, SdfExprs, SdfFunc, SdfFuncs, Sdfs, Voxels)
from .expr import MathF, Mod, VMax, Vec3, Vec4
from .raymarch import FloatData, Matrix4x4, RayMarcher, Vec3Data

__all__ = ["DefaultBatchSize", "MarchingCubes", "Mesh", "Sdf", "SdfExprs", "SdfFunc", "SdfFuncs", "Sdfs",
           "Voxels", "MathF", "Mod", "VMax", "Vec3", "Vec4", "FloatData", "Matrix4x4", "RayMarcher", "Vec3Data"]
