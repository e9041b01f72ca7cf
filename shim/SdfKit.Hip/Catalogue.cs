// Catalogue.cs -- where the reference's factories gain their GPU tag.  These are the bodies the patched
// SdfKit/Sdf.cs and SdfKit/SdfExpr.cs members get (same signatures, same returned delegate -- still callable on the
// CPU exactly as before -- plus the tag).  UNCOMPILED HERE (no .NET in the build image).
using System;
using System.Numerics;
using SdfKit.Hip;

namespace SdfKit
{
    public static partial class Sdfs
    {
        // Sdf.cs:202-215: the reference delegate is kept (CPU callers, RayMarcher on CPU, opaque compositions)
        public static Sdf Sphere(float radius) => GpuSdf.Tag(SphereCpu(radius), GpuSdf.Sphere(radius));
        // Sdf.cs:125-139
        public static Sdf Box(Vector3 bounds) => GpuSdf.Tag(BoxCpu(bounds), GpuSdf.Box(bounds));
        // Sdf.cs:144-156
        public static Sdf Plane(Vector3 normal, float distanceFromOrigin) => GpuSdf.Tag(PlaneCpu(normal, distanceFromOrigin), GpuSdf.Plane(normal, distanceFromOrigin));
        // Sdfs.Cylinder (Sdf.cs:141-142) goes through SdfExprs.Cylinder(...).ToSdf() and is tagged there.
        // Sdfs.Solid(SdfFunc) / Solid(SdfDistFunc) (Sdf.cs:172-200) take COMPILED delegates: opaque, no tag.
        // (SphereCpu / BoxCpu / PlaneCpu = the reference bodies, renamed.)
    }

    public static partial class SdfEx
    {
        // Sdf.cs:101-110: colour constant, distance of the inner SDF
        public static Sdf WithColor(this Sdf sdf, Vector3 color)
        {
            Sdf cpu = WithColorCpu(sdf, color);                       // the reference body
            var inner = GpuSdf.ProgramOf(sdf);
            return inner == null ? cpu : GpuSdf.Tag(cpu, GpuSdf.WithColor(inner, color));
        }

        // Sdf.cs:22-47: the SDF at arbitrary points.  A tagged SDF goes to the GPU in one call (sdfk_eval_points: the delegate's
        // contract exactly -- a .W-only SDF leaves X, Y, Z of the caller's elements alone); an opaque delegate keeps the reference's
        // batched Parallel.For (SampleCpu = the reference body, renamed).  Small batches stay on the CPU: a PCIe round trip costs more.
        public static unsafe void Sample(this Sdf sdf, Memory<Vector3> points, Memory<Vector4> distances, int batchSize = SdfConfig.DefaultBatchSize, int maxDegreeOfParallelism = -1)
        {
            var prog = GpuSdf.ProgramOf(sdf);
            if (prog == null || distances.Length < 4096) { SampleCpu(sdf, points, distances, batchSize, maxDegreeOfParallelism); return; }
            using var hp = points.Pin(); using var hd = distances.Pin();
            Native.Check(Native.sdfk_eval_points(prog.Handle, (float*)hp.Pointer, distances.Length, (float*)hd.Pointer));
        }

        // Sdf.cs:59-63.  With a tagged SDF the whole chain stays on the device: ONE native call
        // (sample + ClipToBounds + sign bits fused, marching cubes, mesh left in HBM), then the copy into the
        // managed Mesh arrays.  batchSize / maxDegreeOfParallelism have no GPU meaning.
        public static unsafe Mesh ToMesh(this Sdf sdf, Vector3 min, Vector3 max, int nx, int ny, int nz, int batchSize = SdfConfig.DefaultBatchSize,
                                         int maxDegreeOfParallelism = -1, bool clipToBounds = true, float isoValue = 0.0f, int step = 1, IProgress<float>? progress = null)
        {
            var prog = GpuSdf.ProgramOf(sdf);
            if (prog == null)   // opaque delegate: the reference's own path (CPU sampler), meshing still on the GPU via the host arrays
                return ToVoxels(sdf, min, max, nx, ny, nz, batchSize, maxDegreeOfParallelism, clipToBounds).ToMesh(isoValue, step, progress);
            // one process per GPU with Dist.Init done: the same call shards the grid by Z slab over the GPUs of the node and
            // every rank gets the whole mesh (collective: every rank makes the call) -- Dist.cs, sdfk_dist_to_mesh
            // a node of several GPUs opened by THIS process (new Node()): one call from this thread uses all of them -- Dist.cs, sdfk_node_to_mesh
            if (Node.Current != null && Node.Current.World > 1 && step == 1 && (long)nx * ny * nz >= Node.MinVoxels) {
                var whole = Node.Current.ToMesh(prog, min, max, nx, ny, nz, clipToBounds, isoValue);
                MarchingCubes.ReportProgress(progress, nz, step);
                return whole;
            }
            if (Dist.World > 1 && step == 1) {
                var whole = Dist.ToMesh(prog, min, max, nx, ny, nz, clipToBounds, isoValue);
                MarchingCubes.ReportProgress(progress, nz, step);
                return whole;
            }
            Native.Check(Native.sdfk_sample_march(prog.Handle, (float*)&min, (float*)&max, nx, ny, nz, clipToBounds ? 1 : 0, isoValue, step, out var h));
            MarchingCubes.ReportProgress(progress, nz, step);
            try { return Mesh.FromNative(h, prog.WritesColor); } finally { Native.sdfk_mesh_free(h); }
        }
    }

    public static partial class SdfExprEx
    {
        // SdfExpr.cs:208-211.  The CPU delegate is still produced (SdfExprCompiler.Compile) so that the result can be
        // called like any Sdf; the GPU program is lowered from the same tree.  A tree with a node the lowering does
        // not know (NotSupportedException) simply stays untagged = CPU only.
        public static Sdf ToSdf(this SdfExpr sdf)
        {
            Sdf cpu = SdfExprCompiler.Compile(sdf);
            try { return GpuSdf.Tag(cpu, Lowering.Lower(sdf)); }
            catch (NotSupportedException) { return cpu; }
        }
    }
}
