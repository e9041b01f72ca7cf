// GpuProgram.cs -- an SDF as a flat float32 SSA program (sdfk_op list) + its native handle, and the registry that
// tags `Sdf` delegates with the program they stand for.  UNCOMPILED HERE (no .NET in the build image).
using System;
using System.Collections.Generic;
using System.Runtime.CompilerServices;

namespace SdfKit.Hip
{
    /// <summary>Builder + owner of one sdfk_program.  Emission conventions (shared with the Python mirror
    /// sdfkit_amd/expr.py so that both hosts produce the same programs): ops 0..2 are X, Y, Z; a constant is
    /// emitted once, at its first use; everything else is emitted in C# evaluation order (operands left to right,
    /// then the operation) with NO common-subexpression folding -- the reference evaluates `sizeX * 0.5f` twice
    /// (SdfExpr.cs:151), so does the program.</summary>
    public sealed class GpuProgram : IDisposable
    {
        readonly List<SdfkOp> ops = new List<SdfkOp>();
        // (constants are not pooled: see Const)
        public int[] OutRgbw = { -1, -1, -1, -1 };
        public bool WritesColor;
        IntPtr handle;

        public GpuProgram()
        {
            Emit(Op.X); Emit(Op.Y); Emit(Op.Z);
        }

        public int X => 0;
        public int Y => 1;
        public int Z => 2;
        public IReadOnlyList<SdfkOp> Ops => ops;

        public int Emit(Op opcode, int a = -1, int b = -1, int c = -1, int d = -1, float imm = 0.0f)
        {
            ops.Add(new SdfkOp { Opcode = (int)opcode, A = a, B = b, C = c, D = d, Imm = imm });
            return ops.Count - 1;
        }

        /// <summary>One Op.Const per mention, never pooled by VALUE: the constants of a program are kernel arguments and the compiled
        /// kernels are shared by every program of one structure (sdfk_program_create, include/sdfkit_hip.h) -- the structure must not
        /// change because two parameters of a scene happen to be equal in one frame of an animation.</summary>
        public int Const(float x) => Emit(Op.Const, imm: x);

        /// <summary>sdfk_program_create on first use (hiprtc, or the library's on-disk code-object cache).</summary>
        public unsafe IntPtr Handle
        {
            get {
                if (handle == IntPtr.Zero) {
                    Native.EnsureInit();
                    var arr = ops.ToArray();
                    fixed (SdfkOp* p = arr)
                    fixed (int* o = OutRgbw)
                        Native.Check(Native.sdfk_program_create(p, arr.Length, o, WritesColor ? 1 : 0, out handle));
                }
                return handle;
            }
        }

        public void Dispose()
        {
            if (handle != IntPtr.Zero) Native.sdfk_program_destroy(handle);
            handle = IntPtr.Zero;
        }

        ~GpuProgram() { Dispose(); }
    }

    /// <summary>`Sdf` is a delegate type (Sdf.cs:8): it cannot carry a field, so GPU-lowerable delegates are tagged
    /// on the side.  Untagged delegates are opaque: they have no GPU form and keep the reference's CPU sampler.</summary>
    public static class GpuSdf
    {
        static readonly ConditionalWeakTable<Sdf, GpuProgram> programs = new ConditionalWeakTable<Sdf, GpuProgram>();

        public static Sdf Tag(Sdf cpuDelegate, GpuProgram program)
        {
            programs.Add(cpuDelegate, program);
            return cpuDelegate;
        }

        public static GpuProgram? ProgramOf(Sdf sdf) => programs.TryGetValue(sdf, out var p) ? p : null;

        /// <summary>[GpuProgram] tags of the batched catalogue (Sdf.cs:118-215).  These delegates only assign `.W`,
        /// so WritesColor = false: colour stays at the zero-initialised scratch value (Voxels.cs:88-92).</summary>
        public static GpuProgram Sphere(float radius)
        {
            var g = new GpuProgram();
            g.OutRgbw[3] = g.Emit(Op.Sub, Lowering.Length(g, g.X, g.Y, g.Z), g.Const(radius));   // p[i].Length() - radius  (Sdf.cs:211)
            return g;
        }

        public static GpuProgram Box(System.Numerics.Vector3 bounds)
        {
            var g = new GpuProgram();
            g.OutRgbw[3] = Lowering.BoxDistance(g, new[] { g.X, g.Y, g.Z }, bounds);                 // Sdf.cs:134-136
            return g;
        }

        public static GpuProgram Plane(System.Numerics.Vector3 n, float d)
        {
            var g = new GpuProgram();
            // Vector3.Dot(p, normal) + distanceFromOrigin  (Sdf.cs:153); Dot = (x*x' + y*y') + z*z'
            int dot = g.Emit(Op.Add, g.Emit(Op.Add, g.Emit(Op.Mul, g.X, g.Const(n.X)), g.Emit(Op.Mul, g.Y, g.Const(n.Y))), g.Emit(Op.Mul, g.Z, g.Const(n.Z)));
            g.OutRgbw[3] = g.Emit(Op.Add, dot, g.Const(d));
            return g;
        }

        /// <summary>SdfEx.WithColor (Sdf.cs:101-115): same distance, constant colour; the result writes colour.</summary>
        public static GpuProgram WithColor(GpuProgram inner, System.Numerics.Vector3 color)
        {
            var g = new GpuProgram();
            int w = Lowering.Inline(g, inner, new[] { g.X, g.Y, g.Z })[3];
            g.OutRgbw = new[] { g.Const(color.X), g.Const(color.Y), g.Const(color.Z), w };
            g.WritesColor = true;
            return g;
        }
    }
}
