// Native.cs -- P/Invoke surface of libsdfkit_hip.so (include/sdfkit_hip.h), one [DllImport] per entry point the
// shim uses.  UNCOMPILED IN THIS REPOSITORY: the build image has no .NET toolchain (DESIGN.md section 1); the same
// ABI is driven end to end by the ctypes mirror sdfkit_amd/_native.py, whose SIGNATURES table this file follows
// entry for entry (tests/test_abi.py checks header <-> exports <-> ctypes table).
using System;
using System.Numerics;
using System.Runtime.InteropServices;

namespace SdfKit.Hip
{
    /// <summary>struct sdfk_op (include/sdfkit_hip.h): one float32 SSA instruction; value id = index.</summary>
    [StructLayout(LayoutKind.Sequential)]
    public struct SdfkOp
    {
        public int Opcode, A, B, C, D;
        public float Imm;
    }

    /// <summary>enum sdfk_opcode</summary>
    public enum Op
    {
        Const = 0, X = 1, Y = 2, Z = 3, Add = 4, Sub = 5, Mul = 6, Div = 7, Neg = 8, Abs = 9, Sqrt = 10, Floor = 11,
        MinSel = 12, MaxSel = 13, MinIeee = 14, MaxIeee = 15, SelLt = 16,
    }

    static unsafe class Native
    {
        const string Lib = "sdfkit_hip";   // libsdfkit_hip.so next to the assembly or on LD_LIBRARY_PATH

        // lifetime
        [DllImport(Lib)] public static extern int sdfk_abi_version();
        [DllImport(Lib)] public static extern int sdfk_init(int device);
        [DllImport(Lib)] public static extern void sdfk_shutdown();
        [DllImport(Lib)] public static extern int sdfk_synchronize();
        [DllImport(Lib)] public static extern IntPtr sdfk_last_error();
        // programs (Sdf delegate / SdfExprCompiler.Compile, Sdf.cs:8, SdfExpr.cs:225-273)
        [DllImport(Lib)] public static extern int sdfk_program_create(SdfkOp* ops, int nOps, int* outRgbw, int writesColor, out IntPtr program);
        [DllImport(Lib)] public static extern int sdfk_program_check(SdfkOp* ops, int nOps, int* outRgbw, int writesColor);
        [DllImport(Lib)] public static extern void sdfk_program_destroy(IntPtr program);
        // Voxels (Voxels.cs)
        [DllImport(Lib)] public static extern int sdfk_volume_create(int nx, int ny, int nz, float* min, float* max, int withColors, out IntPtr volume);
        [DllImport(Lib)] public static extern int sdfk_volume_upload(IntPtr volume, float* values, float* colors3);
        [DllImport(Lib)] public static extern int sdfk_volume_download(IntPtr volume, float* values, float* colors3);
        [DllImport(Lib)] public static extern int sdfk_volume_clip_to_bounds(IntPtr volume);
        [DllImport(Lib)] public static extern void sdfk_volume_free(IntPtr volume);
        [DllImport(Lib)] public static extern int sdfk_sample(IntPtr program, IntPtr volume, int clipToBounds);
        // MarchingCubes.CreateMesh (MarchingCubes.cs:39-92), SdfEx.ToMesh (Sdf.cs:59-63)
        [DllImport(Lib)] public static extern int sdfk_march(IntPtr volume, float iso, int step, out IntPtr mesh);
        [DllImport(Lib)] public static extern int sdfk_march_host(float* values, float* colors3, int nx, int ny, int nz, float* min, float* max, float iso, int step, out IntPtr mesh);
        [DllImport(Lib)] public static extern int sdfk_sample_march(IntPtr program, float* min, float* max, int nx, int ny, int nz, int clip, float iso, int step, out IntPtr mesh);
        // Mesh (Mesh.cs)
        [DllImport(Lib)] public static extern int sdfk_mesh_counts(IntPtr mesh, out long nVertices, out long nIndices);
        [DllImport(Lib)] public static extern int sdfk_mesh_bounds(IntPtr mesh, float* min, float* max);
        [DllImport(Lib)] public static extern int sdfk_mesh_copy(IntPtr mesh, float* v, float* c, float* n, int* tri);
        [DllImport(Lib)] public static extern void sdfk_mesh_free(IntPtr mesh);
        [DllImport(Lib)] public static extern int sdfk_mesh_size_hint(IntPtr mesh, out long nVertices, out long nIndices, out int exact);
        [DllImport(Lib)] public static extern int sdfk_mesh_transform(IntPtr mesh, float* matrix16, float* normalMatrix16);
        [DllImport(Lib)] public static extern int sdfk_host_prefault(void* p, long nBytes);
        // options (what used to be environment variables): enum sdfk_option
        [DllImport(Lib)] public static extern int sdfk_set_option(int key, long value);
        [DllImport(Lib)] public static extern int sdfk_get_option(int key, out long value);
        [DllImport(Lib)] public static extern int sdfk_set_cache_dir([MarshalAs(UnmanagedType.LPStr)] string path);
        [DllImport(Lib)] public static extern int sdfk_stream_placement(int* out8);   // diagnostics: which streams run side by side
        // Z-slab sharding over the GPUs of one node, one process per GPU (SdfEx.ToMesh, Sdf.cs:59-63): Dist.cs
        [DllImport(Lib)] public static extern int sdfk_dist_unique_id(byte* id128);
        [DllImport(Lib)] public static extern int sdfk_dist_init(int world, int rank, byte* id128);
        [DllImport(Lib)] public static extern int sdfk_dist_info(out int world, out int rank, out int backend);
        [DllImport(Lib)] public static extern void sdfk_dist_shutdown();
        [DllImport(Lib)] public static extern int sdfk_dist_to_mesh(IntPtr program, float* min, float* max, int nx, int ny, int nz, int clip, float iso, out IntPtr mesh);
        [DllImport(Lib)] public static extern int sdfk_dist_session_create(IntPtr program, float* min, float* max, int nx, int ny, int nz, int clip, float iso, int depth, out IntPtr session);
        [DllImport(Lib)] public static extern int sdfk_dist_submit(IntPtr session);
        [DllImport(Lib)] public static extern int sdfk_dist_collect(IntPtr session, out long nVerticesMine, out long nIndicesMine);
        [DllImport(Lib)] public static extern int sdfk_dist_counts(IntPtr session, long* counts2PerRank);
        [DllImport(Lib)] public static extern int sdfk_dist_mesh(IntPtr session, out IntPtr mesh);
        [DllImport(Lib)] public static extern int sdfk_dist_tune(IntPtr session, int stepsPerMode, long* nsPerConfig4);
        [DllImport(Lib)] public static extern int sdfk_dist_stats(IntPtr session, long* stats8);
        [DllImport(Lib)] public static extern int sdfk_dist_slab_mesh(IntPtr session, out IntPtr mesh);
        [DllImport(Lib)] public static extern void sdfk_dist_session_free(IntPtr session);
        [DllImport(Lib)] public static extern int sdfk_eval_points(IntPtr program, float* points3, long n, float* rgbw4);   // SdfEx.Sample, Sdf.cs:22-47
        // several GPUs from ONE process (the managed host is one process): include/sdfkit_hip.h, "one process, several GPUs"
        [DllImport(Lib)] public static extern int sdfk_node_open(int* devices, int nDevices, out IntPtr node);
        [DllImport(Lib)] public static extern int sdfk_node_info(IntPtr node, out int world, out int backend);
        [DllImport(Lib)] public static extern int sdfk_node_to_mesh(IntPtr node, SdfkOp* ops, int nOps, int* outRgbw, int writesColor, float* min, float* max,
                                                                    int nx, int ny, int nz, int clip, float iso, out IntPtr mesh);
        [DllImport(Lib)] public static extern int sdfk_node_mesh_begin(IntPtr node, SdfkOp* ops, int nOps, int* outRgbw, int writesColor, float* min, float* max,
                                                                       int nx, int ny, int nz, int clip, float iso, out long nVertices, out long nIndices, out int hasColors);
        [DllImport(Lib)] public static extern int sdfk_node_mesh_copy(IntPtr node, float* vertices3, float* colors3, float* normals3, int* triangles, float* min, float* max);
        [DllImport(Lib)] public static extern void sdfk_node_close(IntPtr node);
        // RayMarcher (RayMarcher.cs:45-211)
        [DllImport(Lib)] public static extern int sdfk_raymarch(IntPtr program, int width, int height, float* cameraPosition, float* viewProjectionInverse,
                                                                float near, float far, int depthIterations, float* depth, float* rgb);

        static readonly object initLock = new object();
        static bool inited;

        /// <summary>enum sdfk_option</summary>
        public const int OptLanes = 1, OptTokens = 2, OptGraphs = 3, OptCopyMode = 4, OptCornerEval = 5, OptVcolorEval = 6,
                         OptDistExchange = 7, OptDistLanes = 8, OptHwQueues = 9, OptCodeCache = 10, OptPrefaultHuge = 11, OptDistIndex16 = 12, OptStreamPlacement = 13, OptIdleLane = 14, OptIdlePrograms = 15, OptElideVolume = 16, OptColorPasses = 17;

        // libc's own setenv: on Unix Environment.SetEnvironmentVariable only edits the runtime's managed copy of the environment,
        // which the getenv of native code -- the HIP runtime's -- never sees
        [DllImport("libc", EntryPoint = "setenv")] static extern int libc_setenv(string name, string value, int overwrite);

        /// <summary>The HIP runtime maps all streams of a process onto GPU_MAX_HW_QUEUES in-order hardware queues (default 4) and reads
        /// the variable when IT initialises; the library's streams want 8 (sdfk_init in csrc/lib_context.hip says why).  A library must not
        /// edit the environment of its process; this binding does, once, before its first native call -- which is this
        /// process's first HIP call unless the host used HIP before (then the host's launcher exports the variable itself).  It has to
        /// be the NATIVE environment: setenv(3) through P/Invoke (overwrite = 0: a value the launcher exported stands).</summary>
        static Native()
        {
            try { libc_setenv("GPU_MAX_HW_QUEUES", "8", 0); }
            catch (Exception) { /* no libc under that name: EnsureInit's check below says what the process ended up with */ }
            Environment.SetEnvironmentVariable("GPU_MAX_HW_QUEUES", Environment.GetEnvironmentVariable("GPU_MAX_HW_QUEUES") ?? "8");
        }

        /// <summary>sdfk_init once per process; device = LOCAL_RANK (one process per GPU) or 0.</summary>
        public static void EnsureInit()
        {
            if (inited) return;
            lock (initLock) {
                if (inited) return;
                // (the entry points this shim binds -- sdfk_set_option, sdfk_dist_*, sdfk_mesh_size_hint -- are ABI 4; ABI 5 changed a default)
                if (sdfk_abi_version() != 6) throw new InvalidOperationException("libsdfkit_hip.so does not have the ABI version (6) this shim was written for");
                int device = int.TryParse(Environment.GetEnvironmentVariable("LOCAL_RANK"), out var r) ? r : 0;
                Check(sdfk_init(device));
                // what the library saw when it came up (0 = unset): with fewer than 8 hardware queues its lanes share queues --
                // correct, slower (a sharded step on a small slab takes twice as long): say so once
                if (sdfk_get_option(OptHwQueues, out long hwq) == 0 && hwq < 8)
                    Console.Error.WriteLine($"SdfKit.Hip: GPU_MAX_HW_QUEUES is {(hwq == 0 ? "unset" : hwq.ToString())} in the native environment " +
                                            "(the HIP runtime was initialised before this binding could export it): export GPU_MAX_HW_QUEUES=8 in the launcher");
                inited = true;
            }
        }

        /// <summary>The reference throws nothing on this path; a native failure becomes InvalidOperationException.</summary>
        public static void Check(int status)
        {
            if (status != 0)
                throw new InvalidOperationException($"sdfkit_hip status {status}: {Marshal.PtrToStringAnsi(sdfk_last_error())}");
        }
    }
}
