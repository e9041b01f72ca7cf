// Dist.cs -- SdfEx.ToMesh (Sdf.cs:59-63) over the GPUs of one node.  Two forms: `Node` (bottom of the file) = several GPUs from THIS
// process, one call from one managed thread; `Dist` = one process per GPU, the grid cut into Z slabs, the
// slab meshes exchanged by the LIBRARY (it calls RCCL itself; include/sdfkit_hip.h, "Z-slab sharding").  What is left for
// the host is what only the host can do: start one process per GPU and hand the 128-byte RCCL id from rank 0 to the others.
// UNCOMPILED IN THIS REPOSITORY (no .NET toolchain in the build image); the same calls, in the same order, are what
// sdfkit_amd/dist.py makes through ctypes, and tests/test_gpu_multirank.py runs them on the GPU.
using System;
using System.IO;
using System.Numerics;
using System.Threading;

namespace SdfKit.Hip
{
    public static unsafe class Dist
    {
        public static int World { get; private set; } = 1;
        public static int Rank { get; private set; }

        /// <summary>One process per GPU, started by any launcher that sets WORLD_SIZE / RANK / LOCAL_RANK (mpirun, torchrun,
        /// a shell loop).  The id travels through a file both sides can see (`rendezvousPath`); a host with MPI or sockets of
        /// its own passes the bytes that way instead (InitWithId).</summary>
        public static void Init(string rendezvousPath)
        {
            int world = int.Parse(Environment.GetEnvironmentVariable("WORLD_SIZE") ?? "1");
            int rank = int.Parse(Environment.GetEnvironmentVariable("RANK") ?? "0");
            Native.EnsureInit();                       // sdfk_init(LOCAL_RANK)
            var id = new byte[128];
            if (rank == 0) {
                fixed (byte* p = id) Native.Check(Native.sdfk_dist_unique_id(p));
                File.WriteAllBytes(rendezvousPath + ".tmp", id);
                File.Move(rendezvousPath + ".tmp", rendezvousPath);            // appears atomically
            } else {
                while (!File.Exists(rendezvousPath)) Thread.Sleep(5);
                id = File.ReadAllBytes(rendezvousPath);
            }
            InitWithId(world, rank, id);
        }

        public static void InitWithId(int world, int rank, byte[] id128)
        {
            Native.EnsureInit();
            fixed (byte* p = id128) Native.Check(Native.sdfk_dist_init(world, rank, p));
            World = world; Rank = rank;
        }

        /// <summary>SdfEx.ToMesh on all ranks: COLLECTIVE (every rank makes the call), every rank gets the whole mesh --
        /// bit for bit the mesh of the single-GPU call.</summary>
        public static Mesh ToMesh(GpuProgram program, Vector3 min, Vector3 max, int nx, int ny, int nz, bool clipToBounds = true, float isoValue = 0)
        {
            float* mn = stackalloc float[3] { min.X, min.Y, min.Z };
            float* mx = stackalloc float[3] { max.X, max.Y, max.Z };
            Native.Check(Native.sdfk_dist_to_mesh(program.Handle, mn, mx, nx, ny, nz, clipToBounds ? 1 : 0, isoValue, out var mesh));
            try { return SdfKit.Mesh.FromNative(mesh, program.WritesColor); } finally { Native.sdfk_mesh_free(mesh); }   // (Voxels.Hip.cs)
        }

        /// <summary>Repeated sharded meshing of one grid (an animation, a parameter sweep): up to `depth` steps in flight,
        /// no host wait inside a step.  Submit() queues a step, Collect() waits for the oldest one.</summary>
        public sealed class Session : IDisposable
        {
            IntPtr h;
            readonly bool writesColor;
            public Session(GpuProgram program, Vector3 min, Vector3 max, int nx, int ny, int nz, bool clipToBounds = true, float isoValue = 0, int depth = 4)
            {
                float* mn = stackalloc float[3] { min.X, min.Y, min.Z };
                float* mx = stackalloc float[3] { max.X, max.Y, max.Z };
                writesColor = program.WritesColor;
                Native.Check(Native.sdfk_dist_session_create(program.Handle, mn, mx, nx, ny, nz, clipToBounds ? 1 : 0, isoValue, depth, out h));
            }
            public void Submit() => Native.Check(Native.sdfk_dist_submit(h));
            public (long vertices, long indices) Collect()
            {
                Native.Check(Native.sdfk_dist_collect(h, out var nv, out var ni));
                return (nv, ni);
            }
            /// <summary>Measures both exchanges, with plain and with compact payloads, on this node's fabric and keeps the fastest configuration (collective, nothing in flight).
            /// A session whose exchange mode is 2 (only rank 0 holds the mesh) or 3 (the mesh stays sharded) is refused -- InvalidOperationException --: who holds the
            /// mesh is that session's contract, not a candidate (ABI 6).</summary>
            public void Tune(int stepsPerMode = 20) => Native.Check(Native.sdfk_dist_tune(h, stepsPerMode, null));
            /// <summary>THIS rank's own slab of the step collected last, indices global (no payload exchange, not collective): with exchange mode 3 the whole mesh is the
            /// ranks' slabs in rank order.</summary>
            public Mesh SlabMesh()
            {
                Native.Check(Native.sdfk_dist_slab_mesh(h, out var mesh));
                try { return SdfKit.Mesh.FromNative(mesh, writesColor); } finally { Native.sdfk_mesh_free(mesh); }
            }
            /// <summary>The whole mesh of the step collected last (exchange mode 3: collective -- the payloads of that step are gathered here; mode 2: rank 0 only).</summary>
            public Mesh Mesh()
            {
                Native.Check(Native.sdfk_dist_mesh(h, out var mesh));
                try { return SdfKit.Mesh.FromNative(mesh, writesColor); } finally { Native.sdfk_mesh_free(mesh); }
            }
            public void Dispose()
            {
                if (h != IntPtr.Zero) Native.sdfk_dist_session_free(h);
                h = IntPtr.Zero;
            }
        }
    }

    /// <summary>Several GPUs from THIS process (sdfk_node_*): the library gives every device a context of its own and a host thread
    /// of its own that is the Z-slab rank; the threads join RCCL exactly as one process per GPU would.  `SdfEx.ToMesh` (Sdf.cs:59-63)
    /// dispatches here when a node is open: one call from one managed thread uses the whole node.
    /// <code>using var node = new Node();            // every GPU of the process
    /// var mesh = node.ToMesh(program, min, max, 1024, 1024, 1024);</code></summary>
    public sealed unsafe class Node : IDisposable
    {
        IntPtr h;
        public int World { get; }
        public static Node? Current { get; private set; }      // what the SdfEx.ToMesh facade consults (Catalogue.cs)
        /// <summary>Grids below this stay on one GPU: a sharded step costs an exchange of the whole mesh between the GPUs (DESIGN.md,
        /// "the fabric bound": at 512^3 every rank receives 23 MB over xGMI, more than one GPU needs for the whole job).</summary>
        public static long MinVoxels = 1L << 28;

        public Node(int[]? devices = null)
        {
            if (devices == null || devices.Length == 0) Native.Check(Native.sdfk_node_open(null, 0, out h));
            else fixed (int* d = devices) Native.Check(Native.sdfk_node_open(d, devices.Length, out h));
            Native.Check(Native.sdfk_node_info(h, out var world, out _));
            World = world;
            Current = this;
        }

        public Mesh ToMesh(GpuProgram program, Vector3 min, Vector3 max, int nx, int ny, int nz, bool clipToBounds = true, float isoValue = 0)
        {
            float* mn = stackalloc float[3] { min.X, min.Y, min.Z };
            float* mx = stackalloc float[3] { max.X, max.Y, max.Z };
            var ops = new SdfkOp[program.Ops.Count];
            for (int i = 0; i < ops.Length; i++) ops[i] = program.Ops[i];
            // two phases, as every managed mesh needs them (Mesh.cs:10-13: exact-length arrays): the sharded step + totals, then every GPU
            // copies its own slab into its slice of the arrays over its own PCIe link (the slabs never cross xGMI)
            long nv, ni; int hasColors;
            fixed (SdfkOp* po = ops) fixed (int* o = program.OutRgbw)
                Native.Check(Native.sdfk_node_mesh_begin(h, po, ops.Length, o, program.WritesColor ? 1 : 0, mn, mx, nx, ny, nz, clipToBounds ? 1 : 0, isoValue,
                                                         out nv, out ni, out hasColors));
            var v = MeshArrayPool.Rent<Vector3>(nv, out _); var nrm = MeshArrayPool.Rent<Vector3>(nv, out _); var t = MeshArrayPool.Rent<int>(ni, out _);
            var recycled = hasColors != 0 ? MeshArrayPool.Rent<Vector3>(nv, out _) : MeshArrayPool.TryRent<Vector3>(nv);
            var c = recycled ?? new Vector3[nv];                        // (no colours and no recycled array: a NEW array is zero already)
            bool passColors = hasColors != 0 || recycled != null;       // (a recycled one is cleared by the library)
            Vector3 lo, hi;
            fixed (Vector3* pv = v, pc = c, pn = nrm) fixed (int* pt = t)
                Native.Check(Native.sdfk_node_mesh_copy(h, (float*)pv, passColors ? (float*)pc : null, (float*)pn, pt, (float*)&lo, (float*)&hi));
            return new Mesh(v, c, nrm, t, lo, hi);                      // (the internal constructor that skips Measure: Voxels.Hip.cs)
        }

        public void Dispose()
        {
            if (Current == this) Current = null;
            if (h != IntPtr.Zero) Native.sdfk_node_close(h);
            h = IntPtr.Zero;
        }
    }
}
