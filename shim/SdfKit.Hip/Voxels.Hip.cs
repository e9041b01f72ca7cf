// Voxels.Hip.cs -- SdfKit.Voxels with a device-resident twin.  This is the patched SdfKit/Voxels.cs: every public
// member keeps its name, signature and meaning (Voxels.cs:6-190).  The ONE source-level change is that `Values` and
// `Colors` turn from `public readonly` fields into get-only properties of the same types, so that a volume that
// was sampled on the GPU does not pay a 16 B/voxel download unless somebody actually reads the arrays.
// UNCOMPILED HERE (no .NET in the build image); sdfkit_amd/api.py's Voxels is the same state machine, tested.
//
// State: hostArrays (managed, allocated lazily) and device (sdfk_volume*).  Exactly one of them, or both, hold the
// truth:
//   deviceIsNewer   set by SampleSdf(tagged sdf) / ClipToBounds on the device; cleared by EnsureHost() (download)
//   hostMayBeNewer  set whenever `Values`/`Colors` (or the indexers' setters) have been handed out -- the arrays
//                   are plain managed arrays, writes to them cannot be observed, so from then on the host copy wins:
//                   CreateMesh uploads it (sdfk_march_host) instead of trusting the device twin.
using System;
using System.Numerics;
using SdfKit.Hip;

namespace SdfKit
{
    public partial class Voxels : IBoundedVolume, IDisposable
    {
        float[,,]? values;
        Vector3[,,]? colors;
        IntPtr device;                 // sdfk_volume* or zero
        bool deviceHasColors, deviceIsNewer, hostMayBeNewer;

        public float[,,] Values { get { EnsureHost(); hostMayBeNewer = true; return values!; } }
        public Vector3[,,] Colors { get { EnsureHost(); hostMayBeNewer = true; return colors!; } }

        // Voxels.cs:23-35: caller-supplied arrays are the truth
        public Voxels(float[,,] values, Vector3[,,] colors, Vector3 min, Vector3 max) : this(min, max, values.GetLength(0), values.GetLength(1), values.GetLength(2))
        {
            this.values = values; this.colors = colors; hostMayBeNewer = true;
        }

        // Voxels.cs:37-40 (+ the DX/DY/DZ arithmetic of :32-34, unchanged); no array is allocated yet
        public Voxels(Vector3 min, Vector3 max, int nx, int ny, int nz)
        {
            Min = min; Max = max; NX = nx; NY = ny; NZ = nz;
            DX = NX >= 1 ? (max.X - min.X) / NX : 0.0f;
            DY = NY >= 1 ? (max.Y - min.Y) / NY : 0.0f;
            DZ = NZ >= 1 ? (max.Z - min.Z) / NZ : 0.0f;
        }

        unsafe void EnsureHost()
        {
            values ??= new float[NX, NY, NZ];
            colors ??= new Vector3[NX, NY, NZ];
            if (!deviceIsNewer) return;
            fixed (float* pv = values) fixed (Vector3* pc = colors)
                Native.Check(Native.sdfk_volume_download(device, pv, (float*)pc));   // colours of a W-only SDF come back as zeros
            deviceIsNewer = false;
        }

        unsafe IntPtr EnsureDevice(bool withColors)
        {
            Native.EnsureInit();
            if (device != IntPtr.Zero && withColors && !deviceHasColors) { Native.sdfk_volume_free(device); device = IntPtr.Zero; }
            if (device == IntPtr.Zero) {
                var mn = Min; var mx = Max;
                Native.Check(Native.sdfk_volume_create(NX, NY, NZ, (float*)&mn, (float*)&mx, withColors ? 1 : 0, out device));
                deviceHasColors = withColors;
            }
            return device;
        }

        /// <summary>Voxels.SampleSdf (Voxels.cs:72-125).  A tagged SDF runs as one kernel launch on the device
        /// (cell centres min + 0.5 D + i D in float32, exactly :81,104-106); an opaque delegate runs the reference's
        /// own Parallel.For body into the managed arrays (SampleSdfCpu = that body, unchanged).</summary>
        public void SampleSdf(Sdf sdf, int batchSize = SdfConfig.DefaultBatchSize, int maxDegreeOfParallelism = -1)
        {
            var prog = GpuSdf.ProgramOf(sdf);
            if (prog == null) { EnsureHost(); SampleSdfCpu(sdf, batchSize, maxDegreeOfParallelism); hostMayBeNewer = true; return; }
            Native.Check(Native.sdfk_sample(prog.Handle, EnsureDevice(prog.WritesColor), 0));
            deviceIsNewer = true; hostMayBeNewer = false;
        }

        /// <summary>Voxels.ClipToBounds (Voxels.cs:133-167): all six faces become Size.X / NX.</summary>
        public void ClipToBounds()
        {
            Native.Check(Native.sdfk_volume_clip_to_bounds(SyncToDevice()));
            deviceIsNewer = true; hostMayBeNewer = false;
        }

        /// <summary>The device twin with the current truth in it (uploads the managed arrays when they may be newer).</summary>
        internal unsafe IntPtr SyncToDevice()
        {
            if (!hostMayBeNewer && device != IntPtr.Zero) return device;
            EnsureHost();
            bool anyColor = false;
            foreach (var c in colors!) if (c != Vector3.Zero) { anyColor = true; break; }
            var h = EnsureDevice(anyColor || deviceHasColors);
            fixed (float* pv = values) fixed (Vector3* pc = colors)
                Native.Check(Native.sdfk_volume_upload(h, pv, deviceHasColors ? (float*)pc : null));
            hostMayBeNewer = false;
            return h;
        }

        // indexers (Voxels.cs:42-65) go through Values; ToMesh (:67-70) and the static SampleSdf overloads (:169-189)
        // are unchanged: they call the members above.

        public void Dispose()
        {
            if (device != IntPtr.Zero) Native.sdfk_volume_free(device);
            device = IntPtr.Zero;
            GC.SuppressFinalize(this);
        }

        ~Voxels() { if (device != IntPtr.Zero) Native.sdfk_volume_free(device); }
    }

    // MarchingCubes.CreateMesh (MarchingCubes.cs:39-92): same signature, same defaults.
    public static partial class MarchingCubes
    {
        public static Mesh CreateMesh(Voxels volume, float isoValue = 0.0f, int step = 1, IProgress<float>? progress = null)
        {
            // device twin when it holds the truth (sampled on the GPU, arrays never handed out): no 4 B/voxel upload,
            // sign bits already there; otherwise the managed arrays are uploaded first (SyncToDevice)
            Native.Check(Native.sdfk_march(volume.SyncToDevice(), isoValue, step, out var h));
            ReportProgress(progress, volume.NZ, step);
            try { return Mesh.FromNative(h); } finally { Native.sdfk_mesh_free(h); }
        }

        /// <summary>IProgress contract of MarchingCubes.cs:53-81: one report per z layer, (float)z / (nz - 2*step),
        /// values in [0,1] including ~0 and ~1 (MarchingCubesTests.cs:150-168).  Delivered after the native call.</summary>
        internal static void ReportProgress(IProgress<float>? progress, int nz, int step)
        {
            if (progress == null) return;
            int zb = nz - 2 * step;
            for (int z = 0; z < nz - step; z += step) progress.Report((float)z / zb);
        }
    }

    /// <summary>Exact-length arrays of meshes that were handed back with <see cref="Mesh.Recycle"/>, keyed by element type and
    /// length.  Mesh.cs:10-13 makes a mesh four managed arrays of exactly Vertices.Length / Triangles.Length elements, and on a
    /// growing heap a new array is memory nobody has touched: 26 MB of page faults, 1.0 of the 1.6 ms one ToMesh call costs at
    /// 512^3, against 0.75 ms into resident pages.  A host that meshes the same grid again and again gets the same counts again
    /// and again: the arrays of the mesh it is done with serve the next one (bench.py: one_step_incl_mesh_d2h.pooled; the Python
    /// mirror has the same class, sdfkit_amd/api.py).  A miss allocates with GC.AllocateUninitializedArray (no zeroing pass).
    /// Like System.Buffers.ArrayPool: whoever recycles a mesh must not touch its arrays afterwards.</summary>
    public static class MeshArrayPool
    {
        public static int PerLength = 4;              // arrays kept per (type, length)
        public static long MaxBytes = 1L << 30;
        static readonly object gate = new object();
        static readonly System.Collections.Generic.Dictionary<(Type, int), System.Collections.Generic.Stack<Array>> free = new();
        static readonly System.Collections.Generic.LinkedList<(Type, int)> order = new();   // least recently returned first
        static long bytes;
        public static long Hits, Misses;

        public static T[] Rent<T>(long n, out bool recycled) where T : unmanaged
        {
            lock (gate) {
                if (free.TryGetValue((typeof(T), (int)n), out var st) && st.Count > 0) {
                    var a = (T[])st.Pop();
                    bytes -= n * System.Runtime.CompilerServices.Unsafe.SizeOf<T>();
                    Hits++; recycled = true;
                    return a;
                }
                Misses++;
            }
            recycled = false;
            return GC.AllocateUninitializedArray<T>((int)n);
        }

        /// <summary>A recycled array of exactly n elements, or null -- without allocating on a miss (for callers that want a
        /// ZEROED array then: `new T[n]`, not the uninitialised one Rent would hand out).</summary>
        public static T[]? TryRent<T>(long n) where T : unmanaged
        {
            lock (gate) {
                if (free.TryGetValue((typeof(T), (int)n), out var st) && st.Count > 0) {
                    var a = (T[])st.Pop();
                    bytes -= n * System.Runtime.CompilerServices.Unsafe.SizeOf<T>();
                    Hits++;
                    return a;
                }
                Misses++;
            }
            return null;
        }

        public static void Return<T>(T[]? a) where T : unmanaged
        {
            if (a == null || a.Length == 0) return;
            long sz = (long)a.Length * System.Runtime.CompilerServices.Unsafe.SizeOf<T>();
            lock (gate) {
                var key = (typeof(T), a.Length);
                if (!free.TryGetValue(key, out var st)) free[key] = st = new System.Collections.Generic.Stack<Array>();
                if (st.Count >= PerLength) return;
                st.Push(a); bytes += sz;
                order.Remove(key); order.AddLast(key);
                while (bytes > MaxBytes && order.First != null) {     // the length nobody has returned to for longest goes first
                    var old = order.First.Value; order.RemoveFirst();
                    if (free.Remove(old, out var dead))
                        foreach (var d in dead) bytes -= Buffer.ByteLength(d);
                }
            }
        }
    }

    public partial class Mesh
    {
        /// <summary>sdfk_mesh_counts waits for the (deferred) job and verifies its size guess; the four managed arrays
        /// are exact-length like the reference's (tests assert Vertices.Length) and come from <see cref="MeshArrayPool"/> (a miss:
        /// fresh, uninitialised); Min/Max come from the device (Mesh.Measure, Mesh.cs:30-45, is fused into the vertex kernel)
        /// through an internal constructor that skips Measure().</summary>
        internal static unsafe Mesh FromNative(IntPtr h, bool colors = true)
        {
            Native.Check(Native.sdfk_mesh_counts(h, out long nv, out long ni));
            var v = MeshArrayPool.Rent<Vector3>(nv, out _); var n = MeshArrayPool.Rent<Vector3>(nv, out _); var t = MeshArrayPool.Rent<int>(ni, out _);
            // colors == false: the program only writes .W, every colour is (0,0,0) (Voxels.cs:88-92).  A NEW array is zero already:
            // no colour destination is passed, the library has nothing to clear or to fault in (0.7 of 2.4 ms at 512^3); a recycled
            // one holds an old mesh's colours: it is passed, and the library clears it (resident pages: a memset on its pool)
            Vector3[] c;
            bool clear = true;
            if (colors) c = MeshArrayPool.Rent<Vector3>(nv, out _);
            else {
                var recycled = MeshArrayPool.TryRent<Vector3>(nv);   // (no allocation on a miss: a ZEROED array is wanted then)
                clear = recycled != null;
                c = recycled ?? new Vector3[nv];
            }
            fixed (Vector3* pv = v, pc = c, pn = n) fixed (int* pt = t)
                Native.Check(Native.sdfk_mesh_copy(h, (float*)pv, clear ? (float*)pc : null, (float*)pn, pt));
            Vector3 mn, mx;
            Native.Check(Native.sdfk_mesh_bounds(h, (float*)&mn, (float*)&mx));
            return new Mesh(v, c, n, t, mn, mx);
        }

        /// <summary>Hands the four arrays to <see cref="MeshArrayPool"/>: the next mesh of the same size gets them instead of new,
        /// untouched memory.  Not in the reference (a Mesh is simply collected); opt-in for hosts that mesh repeatedly.  The mesh
        /// and its arrays must not be used afterwards.  A second call on the same mesh does nothing (the arrays go to the pool ONCE:
        /// pushed twice, two later meshes would be handed the same memory); two Mesh objects built over the SAME arrays must not
        /// both be recycled -- the pool cannot see that (the Python mirror empties the mesh instead: Vertices etc. are get-only here,
        /// Mesh.cs:10-13).</summary>
        public void Recycle()
        {
            if (recycled) return;
            recycled = true;
            MeshArrayPool.Return(Vertices); MeshArrayPool.Return(Colors); MeshArrayPool.Return(Normals); MeshArrayPool.Return(Triangles);
        }
        bool recycled;
    }
}
