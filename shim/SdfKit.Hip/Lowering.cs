// Lowering.cs -- SdfExpr (= Expression<SdfFunc>, GlobalUsings.cs:16) -> sdfk_op list.
// The GPU counterpart of SdfExprCompiler.CreateBatchedLambda/Compile (SdfExpr.cs:225-273): instead of wrapping
// the per-point expression in a batch loop and handing it to the CLR JIT, the tree is flattened into one float32
// SSA instruction per float operation, in C# evaluation order, and libsdfkit_hip.so JIT-compiles THAT (hiprtc)
// into the grid-sampling kernel.  UNCOMPILED HERE (no .NET in the build image); the emission rules are the ones
// sdfkit_amd/expr.py + api.py implement, and tests/test_shim_oplists.py replays the op lists this file emits for
// the README scene (README.md:24-30), Union and the catalogue through the C ABI.
//
// Node shapes handled = everything SdfExpr.cs builds:
//   Lambda / Invoke                 SdfExprEx.ModifyInput (:79-89) nests Invoke(sdf, Invoke(changePosition, p))
//   Block + Assign + variables      ModifyOutput (:91-111), ModifyInputAndOutput (:113-141), SdfExprs.Union (:53-68)
//   Conditional(LessThan(da.W,db.W), da, db)   Union (:61-66) -> one SEL_LT per component
//   New Vector3(x,y,z) / Vector3(v) / Vector4(Vector3,w) / Vector4(x,y,z,w)
//   MemberInit SdfIndexedInput { Position = ..., Index = ... }   RepeatXY / RepeatXZ with colour lambda (:163-195)
//   MemberAccess .X .Y .Z .W, SdfIndexedInput.Position/.Index, Vector3.One/Zero/UnitX.., closure fields (constants)
//   + - * / unary -  on float and Vector3 (BCL operators are component-wise; float*Vector3 and Vector3/float
//   broadcast the scalar), Call: MathF.Sqrt/Abs/Floor/Max/Min, Math.Max/Min/Abs/Sqrt/Floor on floats,
//   Vector3.Abs/Max/Min/Dot, v.Length(), v.LengthSquared(), VectorOps.Mod (VectorData.cs:697-698),
//   VectorOps.VMax (:860-861), Convert(int|double constant -> float).
// Anything else throws NotSupportedException: the caller then keeps the reference's CPU path for that SDF.
using System;
using System.Collections.Generic;
using System.Linq.Expressions;
using System.Numerics;
using System.Reflection;

namespace SdfKit.Hip
{
    /// <summary>A lowered value: 1 (float), 3 (Vector3) or 4 (Vector4) value ids, or an SdfIndexedInput pair.</summary>
    sealed class Val
    {
        public int[] Ids = Array.Empty<int>();
        public Val? Position, Index;             // SdfIndexedInput
        public bool IsIndexedInput => Position != null;
        public static Val Of(params int[] ids) => new Val { Ids = ids };
    }

    public static class Lowering
    {
        /// <summary>SdfExprEx.ToSdf's GPU half: the whole tree as one program.  An SdfExpr always builds a full
        /// Vector4 (colour + distance), so the program writes colour.</summary>
        public static GpuProgram Lower(Expression<SdfFunc> sdf)
        {
            var g = new GpuProgram { WritesColor = true };
            var env = new Dictionary<ParameterExpression, Val>();
            var r = new Visitor(g).InvokeLambda(sdf, new[] { Val.Of(g.X, g.Y, g.Z) }, env);
            if (r.Ids.Length != 4) throw new NotSupportedException("an SdfExpr must produce a Vector4");
            g.OutRgbw = r.Ids;
            return g;
        }

        // ---- shared arithmetic (also used by the [GpuProgram] tags in GpuProgram.cs) ---------------------------
        /// <summary>Vector3.Length(): MathF.Sqrt((x*x + y*y) + z*z)  (BCL, scalar order)</summary>
        public static int Length(GpuProgram g, int x, int y, int z) =>
            g.Emit(Op.Sqrt, g.Emit(Op.Add, g.Emit(Op.Add, g.Emit(Op.Mul, x, x), g.Emit(Op.Mul, y, y)), g.Emit(Op.Mul, z, z)));

        /// <summary>Vector3.Max(wd, Zero).Length() + VMax(Vector3.Min(wd, Zero)), wd = Abs(p) - bounds (Sdf.cs:134-136)</summary>
        public static int BoxDistance(GpuProgram g, int[] p, Vector3 bounds)
        {
            int[] b = { g.Const(bounds.X), g.Const(bounds.Y), g.Const(bounds.Z) };
            var wd = new int[3];
            for (int k = 0; k < 3; k++) wd[k] = g.Emit(Op.Sub, g.Emit(Op.Abs, p[k]), b[k]);
            int zero = g.Const(0.0f);
            var hi = new int[3];
            for (int k = 0; k < 3; k++) hi[k] = g.Emit(Op.MaxSel, wd[k], zero);          // Vector3.Max: (a > b) ? a : b
            int len = Length(g, hi[0], hi[1], hi[2]);
            var lo = new int[3];
            for (int k = 0; k < 3; k++) lo[k] = g.Emit(Op.MinSel, wd[k], zero);          // Vector3.Min: (a < b) ? a : b
            int vmax = g.Emit(Op.MaxIeee, g.Emit(Op.MaxIeee, lo[0], lo[1]), lo[2]);     // Math.Max(Math.Max(x, y), z)
            return g.Emit(Op.Add, len, vmax);
        }

        /// <summary>Copies program `inner` into `g` with its inputs X, Y, Z bound to the value ids `p`; returns the
        /// ids of inner's (r, g, b, w) outputs (-1 where inner writes no colour).</summary>
        public static int[] Inline(GpuProgram g, GpuProgram inner, int[] p)
        {
            var map = new int[inner.Ops.Count];
            for (int i = 0; i < inner.Ops.Count; i++) {
                var o = inner.Ops[i];
                int M(int id) => id < 0 ? -1 : map[id];
                switch ((Op)o.Opcode) {
                case Op.X: map[i] = p[0]; break;
                case Op.Y: map[i] = p[1]; break;
                case Op.Z: map[i] = p[2]; break;
                case Op.Const: map[i] = g.Const(o.Imm); break;
                default: map[i] = g.Emit((Op)o.Opcode, M(o.A), M(o.B), M(o.C), M(o.D)); break;
                }
            }
            var r = new int[4];
            for (int k = 0; k < 4; k++) r[k] = inner.OutRgbw[k] < 0 ? -1 : map[inner.OutRgbw[k]];
            return r;
        }

        sealed class Visitor
        {
            readonly GpuProgram g;
            public Visitor(GpuProgram g) { this.g = g; }

            public Val InvokeLambda(LambdaExpression f, IReadOnlyList<Val> args, Dictionary<ParameterExpression, Val> outer)
            {
                // lexical scoping: the lambda body sees its own parameters (+ whatever encloses the lambda)
                var env = new Dictionary<ParameterExpression, Val>(outer);
                for (int i = 0; i < f.Parameters.Count; i++) env[f.Parameters[i]] = args[i];
                return Visit(f.Body, env);
            }

            Val Visit(Expression e, Dictionary<ParameterExpression, Val> env)
            {
                switch (e) {
                case ParameterExpression p:
                    return env.TryGetValue(p, out var v) ? v : throw new NotSupportedException($"unbound parameter {p.Name}");
                case ConstantExpression c:
                    return Constant(c.Value);
                case InvocationExpression inv: {
                    // Invoke(sdf, arg...): `sdf` is a quoted lambda held in a ConstantExpression or a closure field
                    // (SdfExpr.cs:83-88 passes the Expression<SdfFunc> object itself), or an inline LambdaExpression
                    var target = inv.Expression is LambdaExpression l ? l : Evaluate(inv.Expression) as LambdaExpression
                                 ?? throw new NotSupportedException("Invoke of a compiled delegate: opaque to the GPU (wrap it with SdfExprs.Solid of an expression)");
                    var args = new List<Val>();
                    foreach (var a in inv.Arguments) args.Add(Visit(a, env));
                    return InvokeLambda(target, args, env);
                }
                case BlockExpression b: {
                    var inner = new Dictionary<ParameterExpression, Val>(env);
                    Val last = Val.Of();
                    foreach (var s in b.Expressions) last = Visit(s, inner);
                    return last;
                }
                case BinaryExpression { NodeType: ExpressionType.Assign } asg: {
                    var v = Visit(asg.Right, env);
                    env[(ParameterExpression)asg.Left] = v;
                    return v;
                }
                case ConditionalExpression cond: {
                    // (a < b) ? t : f  -- SdfExprs.Union (SdfExpr.cs:61-66); the test must be a float LessThan
                    if (cond.Test is not BinaryExpression { NodeType: ExpressionType.LessThan } lt) throw new NotSupportedException("only `a < b ? x : y` conditionals");
                    int a = Scalar(Visit(lt.Left, env)), b = Scalar(Visit(lt.Right, env));
                    var t = Visit(cond.IfTrue, env); var f = Visit(cond.IfFalse, env);
                    var ids = new int[t.Ids.Length];
                    for (int k = 0; k < ids.Length; k++) ids[k] = g.Emit(Op.SelLt, a, b, t.Ids[k], f.Ids[k]);
                    return Val.Of(ids);
                }
                case NewExpression n: return New(n, env);
                case MemberInitExpression mi: {
                    // new SdfIndexedInput { Position = ..., Index = ... }   (SdfExpr.cs:166-177)
                    var r = new Val();
                    foreach (var bnd in mi.Bindings) {
                        var ma = (MemberAssignment)bnd;
                        var v = Visit(ma.Expression, env);
                        if (ma.Member.Name == nameof(SdfIndexedInput.Position)) r.Position = v; else r.Index = v;
                    }
                    return r;
                }
                case MemberExpression m: return Member(m, env);
                case UnaryExpression { NodeType: ExpressionType.Negate } neg: {
                    var v = Visit(neg.Operand, env);
                    var ids = new int[v.Ids.Length];
                    for (int k = 0; k < ids.Length; k++) ids[k] = g.Emit(Op.Neg, v.Ids[k]);
                    return Val.Of(ids);
                }
                case UnaryExpression { NodeType: ExpressionType.Convert or ExpressionType.Quote } cv:
                    return cv.NodeType == ExpressionType.Quote ? throw new NotSupportedException("quoted lambda outside Invoke")
                         : cv.Operand.Type == typeof(float) ? Visit(cv.Operand, env) : Constant(Convert.ToSingle(Evaluate(cv.Operand)));
                case BinaryExpression bin: return Binary(bin, env);
                case MethodCallExpression call: return Call(call, env);
                default:
                    throw new NotSupportedException($"expression node {e.NodeType} has no GPU lowering");
                }
            }

            // ---- leaves -------------------------------------------------------------------------------------------
            Val Constant(object? o) => o switch {
                float f => Val.Of(g.Const(f)),
                int i => Val.Of(g.Const(i)),
                double d => Val.Of(g.Const((float)d)),
                Vector3 v => Val.Of(g.Const(v.X), g.Const(v.Y), g.Const(v.Z)),
                Vector4 v => Val.Of(g.Const(v.X), g.Const(v.Y), g.Const(v.Z), g.Const(v.W)),
                _ => throw new NotSupportedException($"constant of type {o?.GetType()}"),
            };

            /// <summary>Closure fields, static fields and anything else that does not depend on the sample point:
            /// evaluated on the host, once, like the CLR would when the delegate runs.</summary>
            static object? Evaluate(Expression e) => e switch {
                ConstantExpression c => c.Value,
                MemberExpression { Member: FieldInfo f } m => f.GetValue(m.Expression == null ? null : Evaluate(m.Expression)),
                MemberExpression { Member: PropertyInfo p } m => p.GetValue(m.Expression == null ? null : Evaluate(m.Expression)),
                _ => Expression.Lambda(e).Compile().DynamicInvoke(),
            };

            static bool DependsOnPoint(Expression? e) => e != null && new PointFinder().Found(e);
            sealed class PointFinder : ExpressionVisitor
            {
                bool hit;
                public bool Found(Expression e) { Visit(e); return hit; }
                protected override Expression VisitParameter(ParameterExpression p) { hit = true; return p; }
            }

            Val Member(MemberExpression m, Dictionary<ParameterExpression, Val> env)
            {
                if (!DependsOnPoint(m.Expression)) return Constant(Evaluate(m));   // closure field (sizeX, color, r), Vector3.One, ...
                var o = Visit(m.Expression!, env);
                if (o.IsIndexedInput) return m.Member.Name == nameof(SdfIndexedInput.Position) ? o.Position! : o.Index!;
                return m.Member.Name switch {
                    "X" => Val.Of(o.Ids[0]), "Y" => Val.Of(o.Ids[1]), "Z" => Val.Of(o.Ids[2]), "W" => Val.Of(o.Ids[3]),
                    _ => throw new NotSupportedException($"member {m.Member.Name}"),
                };
            }

            Val New(NewExpression n, Dictionary<ParameterExpression, Val> env)
            {
                var a = new List<Val>();
                foreach (var x in n.Arguments) a.Add(Visit(x, env));
                var ids = new List<int>();
                foreach (var v in a) ids.AddRange(v.Ids);
                if (n.Type == typeof(Vector3) && ids.Count == 1) return Val.Of(ids[0], ids[0], ids[0]);   // new Vector3(float)
                if ((n.Type == typeof(Vector3) && ids.Count == 3) || (n.Type == typeof(Vector4) && ids.Count == 4)) return Val.Of(ids.ToArray());
                throw new NotSupportedException($"constructor {n.Constructor}");
            }

            static int Scalar(Val v) => v.Ids.Length == 1 ? v.Ids[0] : throw new NotSupportedException("float expected");

            // ---- operators: float op float; Vector3 op Vector3 component-wise; float*Vector3, Vector3*float,
            // Vector3/float broadcast the scalar (System.Numerics semantics) --------------------------------------
            Val Binary(BinaryExpression b, Dictionary<ParameterExpression, Val> env)
            {
                Op op = b.NodeType switch {
                    ExpressionType.Add => Op.Add, ExpressionType.Subtract => Op.Sub, ExpressionType.Multiply => Op.Mul, ExpressionType.Divide => Op.Div,
                    _ => throw new NotSupportedException($"operator {b.NodeType}"),
                };
                var l = Visit(b.Left, env); var r = Visit(b.Right, env);     // left operand first: C# evaluation order
                int n = Math.Max(l.Ids.Length, r.Ids.Length);
                var ids = new int[n];
                for (int k = 0; k < n; k++) ids[k] = g.Emit(op, l.Ids[l.Ids.Length == 1 ? 0 : k], r.Ids[r.Ids.Length == 1 ? 0 : k]);
                return Val.Of(ids);
            }

            Val Call(MethodCallExpression c, Dictionary<ParameterExpression, Val> env)
            {
                var a = new List<Val>();
                if (c.Object != null) a.Add(Visit(c.Object, env));
                foreach (var x in c.Arguments) a.Add(Visit(x, env));
                var t = c.Method.DeclaringType; string name = c.Method.Name;
                int[] Map1(Op op, Val v) { var r = new int[v.Ids.Length]; for (int k = 0; k < r.Length; k++) r[k] = g.Emit(op, v.Ids[k]); return r; }
                int[] Map2(Op op, Val u, Val v) { var r = new int[u.Ids.Length]; for (int k = 0; k < r.Length; k++) r[k] = g.Emit(op, u.Ids[k], v.Ids[k]); return r; }
                if (t == typeof(MathF) || t == typeof(Math)) {
                    switch (name) {
                    case "Sqrt": return Val.Of(g.Emit(Op.Sqrt, Scalar(a[0])));
                    case "Abs": return Val.Of(g.Emit(Op.Abs, Scalar(a[0])));
                    case "Floor": return Val.Of(g.Emit(Op.Floor, Scalar(a[0])));
                    case "Max": return Val.Of(g.Emit(Op.MaxIeee, Scalar(a[0]), Scalar(a[1])));   // IEEE 754:2019 maximum (NaN-propagating, -0 < +0)
                    case "Min": return Val.Of(g.Emit(Op.MinIeee, Scalar(a[0]), Scalar(a[1])));
                    }
                } else if (t == typeof(Vector3)) {
                    switch (name) {
                    case "Abs": return Val.Of(Map1(Op.Abs, a[0]));
                    case "Max": return Val.Of(Map2(Op.MaxSel, a[0], a[1]));                        // (a > b) ? a : b per component
                    case "Min": return Val.Of(Map2(Op.MinSel, a[0], a[1]));
                    case "Length": return Val.Of(Length(g, a[0].Ids[0], a[0].Ids[1], a[0].Ids[2]));
                    case "LengthSquared": { var v = a[0].Ids; return Val.Of(g.Emit(Op.Add, g.Emit(Op.Add, g.Emit(Op.Mul, v[0], v[0]), g.Emit(Op.Mul, v[1], v[1])), g.Emit(Op.Mul, v[2], v[2]))); }
                    case "Dot": { var u = a[0].Ids; var v = a[1].Ids; return Val.Of(g.Emit(Op.Add, g.Emit(Op.Add, g.Emit(Op.Mul, u[0], v[0]), g.Emit(Op.Mul, u[1], v[1])), g.Emit(Op.Mul, u[2], v[2]))); }
                    case "op_Addition": return Val.Of(Map2(Op.Add, a[0], a[1]));
                    case "op_Subtraction": return Val.Of(Map2(Op.Sub, a[0], a[1]));
                    case "op_UnaryNegation": return Val.Of(Map1(Op.Neg, a[0]));
                    }
                } else if (t == typeof(VectorOps)) {
                    switch (name) {
                    case "Mod": {   // a - b * MathF.Floor(a / b)   (VectorData.cs:697-698)
                        int x = Scalar(a[0]), y = Scalar(a[1]);
                        return Val.Of(g.Emit(Op.Sub, x, g.Emit(Op.Mul, y, g.Emit(Op.Floor, g.Emit(Op.Div, x, y)))));
                    }
                    case "VMax": { var v = a[0].Ids; return Val.Of(g.Emit(Op.MaxIeee, g.Emit(Op.MaxIeee, v[0], v[1]), v[2])); }   // :860-861
                    }
                }
                throw new NotSupportedException($"call to {t?.Name}.{name} has no GPU lowering");
            }
        }
    }
}
