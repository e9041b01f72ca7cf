/* sdfkit_hip.h -- C ABI of libsdfkit_hip.so, the MI355X (gfx950) implementation of
 * SdfKit's hot path  Voxels.SampleSdf -> [Voxels.ClipToBounds] -> MarchingCubes.CreateMesh.
 *
 * The reference (praeclarum/SdfKit, 100 % managed C#) has no FFI boundary of its own;
 * these entry points are what a `[DllImport("sdfkit_hip")]` shim behind the reference's
 * public Sdf / Voxels / MarchingCubes / Mesh API binds.  Each entry
 * point cites the reference interface it replaces (file:line relative to the reference).
 *
 * Conventions: plain pointers and sizes only; every function returns an sdfk_status
 * (0 = OK) and records a thread-local message readable with sdfk_last_error(); the
 * library never retains caller (host) pointers after a call returns; volumes are
 * [nx][ny][nz] row-major (z fastest) exactly like C# `float[nx,ny,nz]` /
 * `Vector3[nx,ny,nz]` (Voxels.cs:8-9).  There is NO CPU fallback: without a HIP device
 * every compute entry point fails with SDFK_ERR_NO_DEVICE.
 */
#ifndef SDFKIT_HIP_H
#define SDFKIT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the entry points declared in this header are exported
 * (tests/test_abi.py compares `nm -D --defined-only` with the declarations). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define SDFK_ABI_VERSION 6   /* 2: sdfk_jit_stats, sdfk_host_alloc, sdfk_host_free; 3: sdfk_graph_stats; 4: sdfk_set_option, sdfk_dist_*, sdfk_mesh_transform, sdfk_mesh_size_hint;
                                5: SDFK_OPT_ELIDE_VOLUME defaults to 2 (the temporary volume of sdfk_sample_march is not stored); sdfk_init makes a
                                   context per device and per-thread current; sdfk_node_* (several GPUs from one process); sdfk_dist_slab_mesh and exchange
                                   mode 3 (the mesh stays sharded); sdfk_eval_points (SdfEx.Sample);
                                6: SDFK_OPT_COLOR_PASSES; sdfk_dist_gathered refuses a step that brought this rank headers only and sdfk_dist_tune
                                   a session whose exchange mode is 2 or 3 (SDFK_ERR_UNSUPPORTED); gather-to-root has the same who-receives-what on
                                   the host transport as over RCCL; sdfk_host_alloc works in a process whose only contexts are a node's */

typedef enum sdfk_status {
    SDFK_OK = 0,
    SDFK_ERR_INVALID = 1,    /* bad argument */
    SDFK_ERR_NO_DEVICE = 2,  /* no HIP device / sdfk_init not called */
    SDFK_ERR_HIP = 3,        /* a HIP runtime call failed */
    SDFK_ERR_COMPILE = 4,    /* hiprtc failed on a generated SDF kernel */
    SDFK_ERR_NOMEM = 5,
    SDFK_ERR_UNSUPPORTED = 6
} sdfk_status;

/* ---- SDF programs ----------------------------------------------------------
 * Replaces the `Sdf` delegate (Sdf.cs:8) for SDFs that can run on the GPU: the shim
 * lowers an SdfExpr tree (SdfExpr.cs:16-212) or a tagged Sdfs.* factory (Sdf.cs:118-215)
 * to a flat SSA list of scalar float32 operations.  Value id = instruction index.
 * sdfk_program_create validates the list and generates the HIP source of the program's kernels
 * (grid sampler in its row-length instantiations, cell-corner evaluator, ray marcher); each kernel is
 * JIT-compiled with hiprtc -- or loaded from the on-disk code-object cache -- the first time a call
 * needs it (SDFK_ERR_COMPILE is then reported by that call).  The counterpart of
 * SdfExprCompiler.Compile (SdfExpr.cs:225-273).  All arithmetic is IEEE binary32, one rounding
 * per op, no FMA contraction.
 * CONSTANTS ARE ARGUMENTS.  The generated source -- hence the compiled module and its entry in the on-disk cache --
 * depends on the program's STRUCTURE only (opcodes, operand ids, outputs): the `imm` of an SDFK_OP_CONST travels with
 * every launch in the kernel-argument block.  A program with a known structure and other constants (another radius,
 * another period, the next frame of an animation) is created in microseconds and launches the kernels that are already
 * loaded -- in the reference Sdfs.Sphere(radius) is a closure and a new radius costs nothing (Sdf.cs:202-214).  The
 * exceptions are constants one of whose uses the compiler can fold EXACTLY when it sees the value (x * +-1, x / +-1,
 * x / 2^k, x + -0, x - +0, -0 - x): those stay literals and belong to the structure, so that they cost what they always
 * cost; results are bit-identical either way (IEEE-exact foldings only: -ffp-contract=off, no fast-math).  Programs with
 * more than 28 constants keep ALL of them as literals (arguments live in scalar registers: beyond ~30 the sampler spills them,
 * csrc/sample_codegen.h has the measurement) -- such a program is its own structure.  SDFK_OPT_IDLE_PROGRAMS structures stay loaded after their last
 * program has been destroyed. */
typedef enum sdfk_opcode {
    SDFK_OP_CONST = 0,   /* imm (a kernel argument: see above) */
    SDFK_OP_X = 1, SDFK_OP_Y = 2, SDFK_OP_Z = 3,   /* sample point (Voxels.cs:104-106) */
    SDFK_OP_ADD = 4, SDFK_OP_SUB = 5, SDFK_OP_MUL = 6, SDFK_OP_DIV = 7,  /* a ? b */
    SDFK_OP_NEG = 8, SDFK_OP_ABS = 9, SDFK_OP_SQRT = 10, SDFK_OP_FLOOR = 11, /* f(a) */
    SDFK_OP_MIN_SEL = 12, /* (a < b) ? a : b   -- Vector3.Min component */
    SDFK_OP_MAX_SEL = 13, /* (a > b) ? a : b   -- Vector3.Max component */
    SDFK_OP_MIN_IEEE = 14,/* Math.Min / MathF.Min (IEEE 754:2019 minimum) */
    SDFK_OP_MAX_IEEE = 15,/* Math.Max / MathF.Max (IEEE 754:2019 maximum) */
    SDFK_OP_SEL_LT = 16   /* (a < b) ? c : d   -- SdfExprs.Union (SdfExpr.cs:63-66) */
} sdfk_opcode;

typedef struct sdfk_op {
    int32_t opcode;
    int32_t a, b, c, d;  /* operand value ids (unused = -1) */
    float imm;
} sdfk_op;

typedef struct sdfk_program sdfk_program;
typedef struct sdfk_volume sdfk_volume;
typedef struct sdfk_mesh sdfk_mesh;
typedef struct sdfk_march_job sdfk_march_job;

/* ---- lifetime -------------------------------------------------------------- */
int sdfk_abi_version(void);
/* The library's context of HIP device `device` (ordinal) -- created on first use: stream, lanes, pools -- becomes the CALLING THREAD's
 * current context, like hipSetDevice: everything the thread creates afterwards lives on that device.  Idempotent.  A thread that
 * never called sdfk_init works in the context of the first device the process initialised (a host whose calls arrive on pool
 * threads needs nothing else).  Since ABI 5 a second device is not an error: one process may drive several GPUs, one host thread
 * per device (handles are used by threads whose current context is the one they were made in; a mesh's accessors work from any
 * thread).  sdfk_node_open below does exactly that for a whole node.  sdfk_shutdown releases the calling thread's current context. */
int sdfk_init(int device);
void sdfk_shutdown(void);
/* Run on a caller-owned hipStream_t; NULL = the library's own (non-blocking) stream.  NB: the
 * legacy default stream IS the null handle (torch's default stream, for one): passing it selects
 * the library's own stream, which nothing orders against the default stream -- callers that mix
 * their own stream work with library calls create a real stream and pass that
 * (sdfkit_amd/_native.py: bind_torch_stream). */
int sdfk_set_stream(void* hip_stream);
/* Waits for everything the library has queued (the caller's stream and the internal ones). */
int sdfk_synchronize(void);
/* Lane sections.  The calls made between sdfk_lane_begin(lane, ...) and sdfk_lane_end(...) are
 * queued on internal stream `lane` (1..4) instead of the caller's stream, buffers they allocate
 * come from that lane's pool: independent call sequences issued in sections of different lanes
 * overlap on the GPU (what sdfk_sample_march does by itself; a sharded step -- sample a slab,
 * mesh it, pack it -- is such a sequence too).  begin: the lane first waits for
 * `wait_hip_event` (a hipEvent_t the caller recorded, or NULL) -- e.g. "the collective that read
 * this section's output buffer last time has finished".  end(1): the caller's stream waits for
 * everything the section queued, so work the caller queues next (a collective on the packed
 * buffer) sees its results.  Objects created inside a section are used inside sections of the
 * same lane, or after one of their accessors has returned (that waits on the host).  One
 * section at a time per process. */
int sdfk_lane_begin(int32_t lane, void* wait_hip_event);
int sdfk_lane_end(int32_t caller_stream_waits);
const char* sdfk_last_error(void);

/* ---- options ---------------------------------------------------------------
 * Run-time switches of the library (what used to be environment variables read per call).  sdfk_init takes the
 * DEFAULTS from the environment once (SDFK_LANES, SDFK_TOKENS, SDFK_GRAPHS, SDFK_COPY_MODE, SDFK_NO_CORNER_EVAL,
 * SDFK_NO_VCOLOR_EVAL, SDFK_DIST_EXCHANGE, SDFK_NO_CACHE); after that only these calls change or read them. */
typedef enum sdfk_option {
    SDFK_OPT_LANES = 1,         /* internal streams sdfk_sample_march rotates over: 0 (caller's stream), 2, 3 (default), 4 */
    SDFK_OPT_TOKENS = 2,        /* phase tokens, bit 0: sampling kernels of consecutive jobs apart, bit 1: k_vertices apart;
                                   -1 (default) = bit 0 for grids of >= 2^27 voxels */
    SDFK_OPT_GRAPHS = 3,        /* captured launch graphs: 0 never, 1 (default) launch-bound grids, 2 every grid */
    SDFK_OPT_COPY_MODE = 4,     /* device -> caller arrays: 1 (default) bounded pinned ring + thread pool, 0 pre-fault + runtime copy, 2 runtime copy */
    SDFK_OPT_CORNER_EVAL = 5,   /* 1 (default): cell corners of a freshly sampled volume are re-evaluated; 0: gathered */
    SDFK_OPT_VCOLOR_EVAL = 6,   /* 1 (default): vertex colours of a freshly sampled volume are re-evaluated; 0: gathered */
    SDFK_OPT_DIST_EXCHANGE = 7, /* sharded step: 0 (default) ncclAllGather, in place -- the plainest collective; opt-ins: 1 grouped
                                   ncclSend/ncclRecv to every peer (all xGMI links at once), 2 payloads to rank 0 only (headers to all),
                                   3 THE MESH STAYS SHARDED: only the 64-byte headers travel in a step (the counts of every slab on every
                                   rank), the slab payloads stay on their GPUs -- sdfk_dist_slab_mesh hands out a rank's own slab,
                                   sdfk_dist_mesh runs the payload exchange of that one step on demand (collective).  A step is then
                                   no longer bound by what every rank has to receive over xGMI.
                                   sdfk_dist_tune measures 0 against 1 on the node it runs on */
    SDFK_OPT_DIST_LANES = 8,    /* sharded step: internal streams consecutive steps rotate over: 0..3, default 3 (measured: a step on an
                                   8-rank slab of 512^3 takes 82 / 46 / 35 us with 1 / 2 / 3; a fourth shares a hardware queue: 98 us) */
    SDFK_OPT_HW_QUEUES = 9,     /* read-only: GPU_MAX_HW_QUEUES as the process had it when the library came up (0 = unset).  The
                                   HIP runtime maps all streams of a process onto that many in-order hardware queues (default 4);
                                   the library's streams want 8 (sdfk_init in csrc/lib_context.hip says why), and the variable only
                                   counts if it is set before the process's FIRST HIP call: host bindings export it, the
                                   library itself never edits the environment */
    SDFK_OPT_CODE_CACHE = 10,   /* 1 (default): compiled code objects are kept on disk (sdfk_set_cache_dir); 0: off */
    SDFK_OPT_PREFAULT_HUGE = 11,/* 1: whole 2 MiB blocks of a pageable destination are advised MADV_HUGEPAGE before they are first
                                   touched (sdfk_mesh_copy, sdfk_volume_download, sdfk_host_prefault); 0 (default): left as they are
                                   (measured slower where the kernel compacts memory inside the fault) */
    SDFK_OPT_DIST_INDEX16 = 12, /* sharded step, sessions created afterwards: 1 = compact payloads -- the slab's indices travel as uint16 offsets
                                   against one int32 base per 1024 indices (48 -> 36 bytes per vertex of a colourless mesh: what every rank
                                   has to receive from every other rank per step); sdfk_dist_mesh decodes, sdfk_dist_gathered shows the
                                   encoded form.  A slab whose ids do not fit sends the session back to int32 indices (every rank
                                   sees it in the headers).  0 (default): int32 indices, rebased in place by the step */
    SDFK_OPT_STREAM_PLACEMENT = 13, /* at sdfk_init (set it before): 1 (default) = the library measures which of its streams run side by side
                                   on this process's hardware queues / pipes and puts its lanes and the exchange stream where they do not
                                   get in each other's -- or the caller's stream's -- way (about 10 ms); 0 = streams as they come */
    SDFK_OPT_IDLE_LANE = 14,    /* 1 (default): sdfk_sample_march jobs on launch-bound grids (the captured-graph ones) rotate over a FOURTH
                                   internal stream while the caller's stream has nothing queued (hipStreamQuery at the call): that stream
                                   shares the caller's stream's hardware pipe, so it is only used when the caller is not; 0 = never */
    SDFK_OPT_IDLE_PROGRAMS = 15,/* compiled kernel sets are shared by every program of one STRUCTURE (the constants of a program are kernel
                                   arguments, sdfk_program_create); this many structures stay loaded after their last program has been
                                   destroyed (default 32; 0: unloaded at once) -- the next frame of an animation asks for the same one */
    SDFK_OPT_ELIDE_VOLUME = 16, /* The temporary volume of sdfk_sample_march (SdfEx.ToMesh, Sdf.cs:59-63: the Voxels is a local nobody sees),
                                   on grids above the captured-graph limit.  2 (DEFAULT since ABI 5): it is neither stored nor, for the most
                                   part, even evaluated -- 64 x 4 x 4 blocks whose values provably lie on one side of the iso value (the
                                   program evaluated in interval arithmetic over the block: rigorous for the float operations themselves,
                                   no assumption about the SDF) get constant sign words, only the blocks the surface passes through are
                                   evaluated voxel by voxel; cell corners and vertex colours are re-evaluated by the program.  1: every
                                   voxel is evaluated, nothing but the sign bits is stored (4, or 16 with colours, bytes per voxel of HBM
                                   writes less than 0).  0: the volume is evaluated and stored, as the reference does and as the headline
                                   benchmark's step is defined (bench.py sets 0 for its headline).  Meshes are bit-identical in all three.
                                   Never elided: a volume the caller can see (sdfk_sample / sdfk_sample_march_slab / Voxels.SampleSdf:
                                   those always store), step > 1, a NaN iso value, a program one of whose volumes had case-13 sign words
                                   (the dead-cell test of the meshing reads voxels: that job is redone on a stored volume) */
    SDFK_OPT_COLOR_PASSES = 17, /* How Voxels.SampleSdf writes a COLOUR volume (16 B per voxel into two arrays: Voxels.cs:112-120).  One fused pass
                                 * from workgroups that own 8 x rows runs at 0.70-0.75 of the HBM peak even without arithmetic; values (+ sign
                                 * bytes) by that kernel and then the colour array as one linear stream by a second kernel reaches 0.78-0.84
                                 * -- when the program is cheap enough to be evaluated a second time, one voxel per lane.  0 (default): two
                                 * passes for programs of at most 24 operations (one primitive with a constant colour) on grids of at least
                                 * 2^21 voxels, one pass otherwise; 1: always one pass; 2: always two.  Bit-identical results either way. */
    SDFK_OPT_COUNT_ = 18
} sdfk_option;
int sdfk_set_option(int32_t key, int64_t value);
int sdfk_get_option(int32_t key, int64_t* value);
/* Directory of the on-disk code-object cache; NULL = default ($SDFK_CACHE_DIR | $XDG_CACHE_HOME/sdfkit_hip |
 * ~/.cache/sdfkit_hip as the process had them at start-up).  The directory must belong to the calling user and be
 * writable by nobody else, otherwise the cache stays off. */
int sdfk_set_cache_dir(const char* path);

/* out_rgbw = value ids of (colour.X, colour.Y, colour.Z, distance W) -- the Vector4 the
 * delegate writes (Sdf.cs:8).  writes_color = 0 for delegates that only assign `.W`
 * (Sdfs.Sphere/Box/Plane, Sdf.cs:134,153,211): colour stays (0,0,0) (Voxels.cs:88-92). */
int sdfk_program_create(const sdfk_op* ops, int32_t n_ops, const int32_t out_rgbw[4],
                        int32_t writes_color, sdfk_program** out);
/* Generate + hiprtc-compile EVERY kernel of the program for gfx950 without loading (needs no
 * device, never uses the cache): a lowering check the shim can run at build time. */
int sdfk_program_check(const sdfk_op* ops, int32_t n_ops, const int32_t out_rgbw[4], int32_t writes_color);
/* the generated HIP source of the program's STRUCTURE (constants appear as K.k[i]) */
const char* sdfk_program_source(const sdfk_program* p);
/* JIT bookkeeping of this process.  Compiled code objects are kept on disk (see csrc/lib_jit.hip,
 * "on-disk cache": $SDFK_CACHE_DIR | $XDG_CACHE_HOME/sdfkit_hip | ~/.cache/sdfkit_hip; SDFK_NO_CACHE=1
 * switches it off), so only the first process that sees a program pays for hiprtc -- the counterpart
 * of SdfExprCompiler.Compile (SdfExpr.cs:234-238) paying the expression JIT once per delegate. */
int sdfk_jit_stats(int64_t* n_compiled, int64_t* n_cache_hits, double* compile_ms_total);
void sdfk_program_destroy(sdfk_program* p);

/* ---- Voxels (Voxels.cs:6-65) ------------------------------------------------
 * Device-resident `Values` (+ `Colors` when with_colors != 0).  A *slab* volume holds the
 * voxel planes [z0, z0+nz_local) of a global nx*ny*nz_global grid (multi-GPU Z sharding);
 * sdfk_volume_create is the slab z0 = 0, nz_local = nz. */
int sdfk_volume_create(int32_t nx, int32_t ny, int32_t nz, const float min[3], const float max[3],
                       int32_t with_colors, sdfk_volume** out);
int sdfk_volume_create_slab(int32_t nx, int32_t ny, int32_t nz_global, const float min[3],
                            const float max[3], int32_t z0, int32_t nz_local,
                            int32_t with_colors, sdfk_volume** out);
/* host <-> device copies of Voxels.Values / Voxels.Colors (colors may be NULL) */
int sdfk_volume_upload(sdfk_volume* v, const float* values, const float* colors3);
int sdfk_volume_download(const sdfk_volume* v, float* values, float* colors3);
/* Raw device pointers (the caller may write through them: cached sign bits are dropped and
 * meshes that still depend on the volume are completed first).  The DEVICE arrays pad every z row to
 * sdfk_volume_row_pitch voxels (nz rounded up to a multiple of 4: 16-byte aligned rows for the sampling
 * kernel): voxel (x, y, z) is at ((x * ny + y) * pitch + z), its colour at 3x that.  upload / download
 * convert from / to the dense host layout of Voxels.Values / Voxels.Colors. */
int sdfk_volume_device_ptrs(const sdfk_volume* v, void** values, void** colors3);
int sdfk_volume_row_pitch(const sdfk_volume* v, int32_t* pitch_voxels);
void sdfk_volume_free(sdfk_volume* v);

/* Voxels.SampleSdf(Sdf, batchSize, maxDegreeOfParallelism) (Voxels.cs:72-125): evaluates
 * the program at every cell centre and stores W -> Values, XYZ -> Colors.  batchSize and
 * maxDegreeOfParallelism have no GPU meaning and are not part of the ABI.
 * clip_to_bounds != 0 fuses Voxels.ClipToBounds (SdfEx.ToVoxels, Sdf.cs:49-57). */
int sdfk_sample(const sdfk_program* p, sdfk_volume* v, int32_t clip_to_bounds);
/* Voxels.ClipToBounds (Voxels.cs:133-167) on an existing volume. */
int sdfk_volume_clip_to_bounds(sdfk_volume* v);

/* ---- MarchingCubes.CreateMesh (MarchingCubes.cs:39-92) ----------------------
 * One-shot forms.  The mesh is left on the device; query with sdfk_mesh_*.
 *
 * Completion is DEFERRED: a repeat call for a grid shape sizes its buffers from the previous
 * mesh of that shape, queues every kernel and returns the handle without waiting for the GPU.
 * The first sdfk_mesh_* accessor waits, verifies the size guess and, if it was too small, redoes
 * the job exactly -- the caller never sees a difference except in timing, and errors of the job
 * are reported by that accessor.  The library keeps the source volume's contents alive for that:
 * modifying or freeing a volume (upload, sample, clip, free, device_ptrs) first completes the
 * meshes that still depend on it.  sdfk_mesh_free on an unread mesh does not wait. */
int sdfk_march(const sdfk_volume* v, float iso_value, int32_t step, sdfk_mesh** out);
/* Host-array form: `values`/`colors3` are the managed Voxels.Values / Voxels.Colors arrays
 * pinned by the shim for the duration of the call (colors3 may be NULL = zeros). */
int sdfk_march_host(const float* values, const float* colors3, int32_t nx, int32_t ny, int32_t nz,
                    const float min[3], const float max[3], float iso_value, int32_t step,
                    sdfk_mesh** out);
/* SdfEx.ToMesh (Sdf.cs:59-63): sample (+clip) and mesh without leaving the device.
 * Self-contained jobs: consecutive calls are queued on three internal streams in turn (not on the
 * sdfk_set_stream stream) and overlap on the GPU; their results are safe to use from any stream
 * once an accessor has returned.  sdfk_set_option(SDFK_OPT_LANES, 0) keeps them on the caller's stream.
 * Repeat calls on launch-bound grids (<= 2^24 voxels, step 1): the job of a (program, bounds, grid, clip,
 * iso) is built once per internal stream -- volume, workspace and mesh arrays sized from the previous
 * result of that grid shape -- its kernel launches are captured in a hipGraph, and every later call is
 * ONE hipGraphLaunch; the returned handle borrows the job's arrays until it is freed (up to 3 live
 * handles per key and stream, further calls take the ordinary path).  Same results, same deferred
 * completion; a result that outgrows the captured capacities is redone exactly as always.
 * SDFK_OPT_GRAPHS = 0 switches this off, 2 applies it to every grid size. */
int sdfk_sample_march(const sdfk_program* p, const float min[3], const float max[3],
                      int32_t nx, int32_t ny, int32_t nz, int32_t clip_to_bounds,
                      float iso_value, int32_t step, sdfk_mesh** out);
/* Captured jobs alive, graph launches so far, device bytes the captured jobs hold. */
int sdfk_graph_stats(int64_t* jobs, int64_t* launches, int64_t* device_bytes);

/* Two-phase form for Z-slab sharding (one process per GPU).  `v` is a slab whose planes
 * cover the cell layers [layer_begin, layer_end) (global layer indices) plus context:
 * two planes below layer_begin (unless that reaches plane 0) and one plane above
 * layer_end (unless that reaches the last plane).  begin: classify + count, returns the
 * numbers of vertices and triangle indices this slab owns.  finish: emit, with
 * `vertex_base` = sum of the vertex counts of all lower slabs (exchanged by the caller,
 * e.g. with an RCCL all-gather), so that indices are global and the concatenation of the
 * slab meshes in rank order equals the single-device mesh. step must be 1. */
int sdfk_march_begin(const sdfk_volume* v, float iso_value, int32_t layer_begin, int32_t layer_end,
                     sdfk_march_job** job, int64_t* n_vertices, int64_t* n_indices);
int sdfk_march_finish(sdfk_march_job* job, int64_t vertex_base, sdfk_mesh** out);
void sdfk_march_job_free(sdfk_march_job* job);

/* One-call slab forms (deferred completion as above): buffers are sized from the previous call with the
 * same slab shape, classification and emit are queued back to back, and the exact two-phase
 * path is taken only when that guess was too small.  `vertex_base` is added to every index;
 * pass 0 to get slab-local indices and rebase after the gather (sdfk_slabs_rebase). */
int sdfk_march_slab(const sdfk_volume* v, float iso_value, int32_t layer_begin, int32_t layer_end,
                    int64_t vertex_base, sdfk_mesh** out);
int sdfk_sample_march_slab(const sdfk_program* p, sdfk_volume* slab, int32_t clip_to_bounds, float iso_value,
                           int32_t layer_begin, int32_t layer_end, int64_t vertex_base, sdfk_mesh** out);

/* Self-describing slab payload for a single padded all-gather: 64-byte header
 * { int64 n_vertices; int64 n_indices; float min[3]; float max[3]; int32 vertex_bytes; int32 cap_v; int32 idx_bits; int32 flags; pad }
 * followed by Vertices | Colors | Normals (3 floats per vertex each) | Triangles (int32).
 * vertex_bytes = 36, or 24 when the volume had no colours: Colors (all zero) is then left out.
 * cap_v = vertex slots each section is laid out for (sections at 64, 64 + 12 cap_v, ...; indices at
 * 64 + vertex_bytes * cap_v); 0 = dense (cap_v = n_vertices), what sdfk_mesh_pack writes.  Written device
 * to device into `dst` (capacity_bytes); *needed_bytes = header + arrays.  If it does not fit,
 * only the header is written and SDFK_OK is still returned (the caller sees needed > capacity).
 * For a mesh whose job is still queued (deferred completion) nothing waits: the device packs
 * from the job's own counters, *needed_bytes = -1, and the header says what happened -- counts
 * >= 0 (arrays present if they fit the capacity), or nv = ni = -1 when the speculative buffers
 * of that job were too small (complete the mesh with any accessor and pack again). */
#define SDFK_SLAB_HEADER_BYTES 64
int sdfk_mesh_pack(const sdfk_mesh* m, void* dst, int64_t capacity_bytes, int64_t* needed_bytes);
/* One sharded step of a rank in ONE call (what a pipelined driver queues per step; no host wait):
 * [sdfk_lane_begin(lane, wait_hip_event) if lane > 0] sample the slab -> mesh it [sdfk_lane_end(1)].  From the
 * second call for a slab shape on, the mesh is EMITTED STRAIGHT INTO the payload at dst: its arrays are the
 * payload's sections, laid out for the capacities guessed from the previous mesh (cap_v in the header), and the
 * last kernel writes the header -- no pack launch, no second copy.  (The first call, and hints that do not fit
 * capacity_bytes, mesh into buffers of the library and pack.)  The payload header tells the counts (-1: a
 * speculative capacity was too small, redo the step on the exact path). */
int sdfk_slab_enqueue(const sdfk_program* p, sdfk_volume* slab, int32_t clip_to_bounds, float iso_value,
                      int32_t layer_begin, int32_t layer_end, void* dst, int64_t capacity_bytes,
                      int32_t lane, void* wait_hip_event);
/* `gathered` = world payloads of stride_bytes each (device memory, as produced by an
 * all-gather of sdfk_mesh_pack buffers with slab-local indices): adds to the indices of slab r
 * the vertex counts of slabs 0..r-1, in one launch, reading the counts from the headers. */
int sdfk_slabs_rebase(void* gathered, int32_t world, int64_t stride_bytes);
/* Same, and the `world` 64-byte headers are also written to `headers_mirror`: device-accessible
 * (pinned, mapped) HOST memory, so that the host can read the counts after waiting for one event
 * on the stream, without a copy of its own. */
int sdfk_slabs_rebase_mirror(void* gathered, int32_t world, int64_t stride_bytes, void* headers_mirror);

/* ---- Z-slab sharding over the GPUs of one node (SdfEx.ToMesh, Sdf.cs:59-63, one process per GPU) -------------
 * The reference has no distributed path; this is the multi-GPU form of the same call.  The grid is cut into Z slabs
 * of cell layers (rank r owns a contiguous range of the serial z sweep, MarchingCubes.cs:53-82, so the global vertex /
 * triangle order is the concatenation of the slabs in rank order); every rank samples its planes + a 2-plane context
 * (sampling is a pure function of the voxel index, Voxels.cs:99-108: no halo is exchanged), meshes its layers with
 * slab-local vertex ids straight into its section of a gather buffer, and ONE exchange per step moves the slab meshes:
 * the library calls RCCL itself (librccl.so.1, loaded on first use) on a stream of its own -- ncclAllGather, or the
 * same bytes as grouped ncclSend / ncclRecv over every xGMI link at once (SDFK_OPT_DIST_EXCHANGE) -- then one kernel
 * rebases the gathered indices and mirrors the `world` payload headers to the host.
 *
 *   rank 0:  sdfk_dist_unique_id(id)  -> the host hands `id` (128 bytes) to every rank (its own launcher / MPI / a file)
 *   all:     sdfk_init(local_device); sdfk_dist_init(world, rank, id)
 *   one-off: sdfk_dist_to_mesh(...)   -> the whole mesh on every rank, an ordinary sdfk_mesh handle
 *   repeat:  sdfk_dist_session_create(...); { sdfk_dist_submit(s) ... sdfk_dist_collect(s, ...) } with up to `depth`
 *            steps in flight; sdfk_dist_mesh(s, &m) = the mesh of the step collected last
 *
 * Every call below except unique_id / info / slab is COLLECTIVE: all ranks make the same calls in the same order.
 * Decisions that must agree (payload stride, "a rank's speculative buffers were too small: redo this step exactly")
 * are taken from the gathered headers, which every rank sees (csrc/slab_protocol.h). */
#define SDFK_DIST_ID_BYTES 128
int sdfk_dist_unique_id(void* id_out /* SDFK_DIST_ID_BYTES */);
int sdfk_dist_init(int32_t world, int32_t rank, const void* id /* SDFK_DIST_ID_BYTES */);
/* The same with the exchange done by the HOST (ranks that share one GPU, hosts with a transport of their own, tests):
 * `allgather(ctx, send, recv, bytes_per_rank)` gathers `bytes_per_rank` bytes of every rank's HOST buffer `send` into
 * `recv` (world x bytes_per_rank, rank order) and returns 0; blocking.  The library stages the payloads through
 * pinned memory around it; everything else is the same code. */
typedef int (*sdfk_allgather_fn)(void* ctx, const void* send, void* recv, int64_t bytes_per_rank);
int sdfk_dist_init_host(int32_t world, int32_t rank, sdfk_allgather_fn allgather, void* ctx);
/* backend: 0 = not initialised, 1 = RCCL, 2 = host transport */
int sdfk_dist_info(int32_t* world, int32_t* rank, int32_t* backend);
void sdfk_dist_shutdown(void);
/* The partition (no device needed): cell layers [layer_begin, layer_end) of rank `rank`, and the voxel planes
 * [z0, z0 + nz_local) its slab volume holds for them (context planes included, widened to a multiple of 4). */
int sdfk_dist_slab(int32_t nz, int32_t world, int32_t rank, int32_t* layer_begin, int32_t* layer_end, int32_t* z0, int32_t* nz_local);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop    /* (an opaque handle: the type itself -- in the library a class with members -- is not exported) */
#endif
typedef struct sdfk_dist_session sdfk_dist_session;
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif
/* depth = steps that may be in flight (slots: own slab volume and gather buffer each), 1..8. */
int sdfk_dist_session_create(const sdfk_program* p, const float min[3], const float max[3], int32_t nx, int32_t ny, int32_t nz,
                             int32_t clip_to_bounds, float iso_value, int32_t depth, sdfk_dist_session** out);
/* Queue one step (no host wait from the second step on).  SDFK_ERR_INVALID when `depth` steps are in flight. */
int sdfk_dist_submit(sdfk_dist_session* s);
/* Wait for the OLDEST queued step: this rank's slab counts; a step some rank could not fit is redone here by all. */
int sdfk_dist_collect(sdfk_dist_session* s, int64_t* n_vertices_mine, int64_t* n_indices_mine);
/* Per-rank (vertices, indices) of the step collected last: counts[2 * world]. */
int sdfk_dist_counts(const sdfk_dist_session* s, int64_t* counts);
/* The whole mesh of the step collected last (valid call until the next submit reuses that slot): the slab sections of the
 * gather buffer concatenated into an ordinary device-resident mesh -- identical to the single-GPU sdfk_sample_march.
 * (Exchange mode 3: the payloads of that step are exchanged HERE, so the call is collective -- every rank makes it.) */
int sdfk_dist_mesh(sdfk_dist_session* s, sdfk_mesh** out);
/* This rank's OWN slab of the step collected last as a mesh of its own (indices global: + the vertex counts of the slabs before it).
 * Not collective, no payload exchange: with SDFK_OPT_DIST_EXCHANGE = 3 a host assembles the whole mesh from the ranks' slabs, each
 * copied over its own PCIe link to the offsets sdfk_dist_counts gives (vertices: sum of nv of the ranks before; likewise indices). */
int sdfk_dist_slab_mesh(sdfk_dist_session* s, sdfk_mesh** out);
/* The raw gather buffer of the step collected last: world payloads of stride_bytes each (header + sections, indices
 * rebased), device memory owned by the session.  SDFK_ERR_UNSUPPORTED on a rank that received the headers only (exchange mode 3;
 * mode 2 on a rank other than 0): the foreign sections would be a fresh header followed by stale bytes. */
int sdfk_dist_gathered(const sdfk_dist_session* s, void** device_ptr, int64_t* stride_bytes);
/* stats[8] = { stride_bytes, steps submitted, steps redone on the exact path, stride regrowths, exchange mode used,
 *              host nanoseconds spent inside sdfk_dist_submit (total), inside sdfk_dist_collect (total),
 *              depth | 0x100 if the payloads are in the 16-bit index form | fall-backs to int32 indices << 16 } */
int sdfk_dist_stats(const sdfk_dist_session* s, int64_t stats[8]);
/* Which exchange and which payload form are faster on this node's fabric is a measurement: runs steps_per_mode pipelined steps
 * with ncclAllGather and with the direct grouped sends, each with plain payloads and with compact ones (16-bit index offsets:
 * fewer bytes per peer, an encode and a decode pass more), takes the slowest rank's time for each -- the same numbers on every
 * rank -- and keeps the fastest configuration for this session (a change of the payload form agrees the stride again).
 * Collective, nothing in flight; ns_per_config[4] (may be NULL): the agreed times, index = mode + 2 * (compact), -1 = the scene
 * does not fit the compact form; all 0 with the host transport, which has one exchange only.  SDFK_ERR_UNSUPPORTED for a session
 * whose exchange mode is 2 or 3: who holds the mesh is that session's contract, not a candidate (the mode is left alone; after any
 * failure the session keeps the mode it had). */
int sdfk_dist_tune(sdfk_dist_session* s, int32_t steps_per_mode, int64_t* ns_per_config);
/* This rank's slab step WITHOUT the exchange, queued like a step (measurement: "kernel-only" time of a sharded step). */
int sdfk_dist_enqueue_only(sdfk_dist_session* s);
void sdfk_dist_session_free(sdfk_dist_session* s);
/* SdfEx.ToMesh (Sdf.cs:59-63) over all ranks, one call: every rank gets the full mesh. */
int sdfk_dist_to_mesh(const sdfk_program* p, const float min[3], const float max[3], int32_t nx, int32_t ny, int32_t nz,
                      int32_t clip_to_bounds, float iso_value, sdfk_mesh** out);

/* ---- one process, several GPUs (SdfEx.ToMesh, Sdf.cs:59-63: a library call of ONE host process) --------------------------------
 * The sharded step above wants one rank per GPU; a managed host is one process.  A NODE gives every listed device a context of its
 * own and a host thread of the library's own that is the rank: the threads join one RCCL communicator per device (ncclCommInitRank
 * on a shared id -- what one process per GPU does) and run the same Z-slab step side by side (sdfk_dist_session_*: same
 * partition, same exchange, same protocol); the caller posts a scene and gets the WHOLE mesh back as an ordinary sdfk_mesh on
 * the first device (its accessors -- counts, copy, bounds, transform, free -- work from the calling thread).
 *   devices / n_devices: the HIP ordinals of the ranks, in rank order; NULL / 0 = every device of the process.  A device listed
 *   more than once makes ranks that share a GPU: they exchange through host memory (RCCL refuses two ranks on one device) --
 *   for bring-up and tests on a one-GPU box.
 * sdfk_node_to_mesh takes the SDF program as its op list (each rank compiles / loads it on its own device; a repeated scene keeps
 * its programs and its session: the second call is one sharded step).  One call at a time per node.  Free every mesh before
 * sdfk_node_close.  The node's contexts are private: a thread's own sdfk_init context on the same device is independent of them. */
typedef struct sdfk_node sdfk_node;
int sdfk_node_open(const int32_t* devices, int32_t n_devices, sdfk_node** out);
/* world = ranks; backend = 1: RCCL between the ranks, 2: the host transport (ranks share a device) */
int sdfk_node_info(const sdfk_node* node, int32_t* world, int32_t* backend);
int sdfk_node_to_mesh(sdfk_node* node, const sdfk_op* ops, int32_t n_ops, const int32_t out_rgbw[4], int32_t writes_color,
                      const float min[3], const float max[3], int32_t nx, int32_t ny, int32_t nz,
                      int32_t clip_to_bounds, float iso_value, sdfk_mesh** out);
/* The mesh on the HOST, in the two phases every managed caller needs (Mesh.cs:10-13: four exact-length arrays): begin runs the
 * sharded step and returns the totals; copy fills the caller's arrays -- every rank copies ITS slab into its slice (indices global)
 * over its own PCIe link, all ranks at once.  The node's steps leave the mesh sharded (SDFK_OPT_DIST_EXCHANGE = 3 semantics: only
 * the 64-byte headers cross xGMI), so nothing is bound by what a rank would have to receive from the others.  colors3 may be NULL
 * (has_colors = 0: every colour is zero); min / max = Mesh.Measure over the slabs (may be NULL). */
int sdfk_node_mesh_begin(sdfk_node* node, const sdfk_op* ops, int32_t n_ops, const int32_t out_rgbw[4], int32_t writes_color,
                         const float min[3], const float max[3], int32_t nx, int32_t ny, int32_t nz, int32_t clip_to_bounds,
                         float iso_value, int64_t* n_vertices, int64_t* n_indices, int32_t* has_colors);
int sdfk_node_mesh_copy(sdfk_node* node, float* vertices3, float* colors3, float* normals3, int32_t* triangles, float min[3], float max[3]);
void sdfk_node_close(sdfk_node* node);

/* ---- SdfEx.Sample (Sdf.cs:22-47) ----------------------------------------------
 * The SDF at `n` arbitrary points (x, y, z triples): rgbw4[4 i .. 4 i + 3] = (colour, distance) of point i, exactly what the
 * delegate writes into the caller's Vector4 buffer -- a program that only assigns .W (Sdfs.Sphere, Sdf.cs:211) leaves X, Y, Z of
 * every element as the caller had them.  batchSize / maxDegreeOfParallelism of the reference have no meaning here.
 * sdfk_eval_points: host arrays, synchronous; sdfk_eval_points_device: caller-owned device buffers, asynchronous on the
 * library stream. */
int sdfk_eval_points(const sdfk_program* p, const float* points3, int64_t n, float* rgbw4);
int sdfk_eval_points_device(const sdfk_program* p, const void* points3_dev, int64_t n, void* rgbw4_dev);

/* ---- RayMarcher (RayMarcher.cs:45-211) ---------------------------------------
 * RenderDepth (depth != NULL) and / or Render (rgb != NULL) of the program's SDF by sphere
 * tracing: one ray per pixel, `depth_iterations` steps, 6 more evaluations for the normal,
 * Lambert shading and the sky colour exactly as RayMarcher.Render (RayMarcher.cs:134-169).
 * camera_position and view_projection_inverse (row-major M11..M44) are what GetCameraRays
 * (RayMarcher.cs:97-112) derives from ViewTransform / field of view / planes with
 * System.Numerics; the shim computes them with the BCL and passes them in.  Images are
 * FloatData.Values / Vec3Data.Values layout: pixel (column i, row j) at j*width + i, 3 floats
 * per pixel for rgb.  sdfk_raymarch fills host arrays (synchronous); sdfk_raymarch_device
 * leaves the images in caller-owned device buffers, asynchronously on the library stream. */
int sdfk_raymarch(const sdfk_program* p, int32_t width, int32_t height, const float camera_position[3],
                  const float view_projection_inverse[16], float near_plane, float far_plane,
                  int32_t depth_iterations, float* depth, float* rgb);
int sdfk_raymarch_device(const sdfk_program* p, int32_t width, int32_t height, const float camera_position[3],
                         const float view_projection_inverse[16], float near_plane, float far_plane,
                         int32_t depth_iterations, void* depth_dev, void* rgb_dev);

/* ---- pinned host arena -------------------------------------------------------
 * Host memory the GPU can write directly (hipHostMalloc), recycled through size-class free lists:
 * destinations inside such a block make sdfk_mesh_copy / sdfk_volume_download / sdfk_raymarch plain
 * DMA transfers at the link rate (512^3 sphere mesh, 33 MB: 0.6 ms).  Into ordinary pageable
 * memory (managed arrays pinned by the shim for the call) the same entry points first touch the
 * destination pages on a small thread pool, which is what a copy into FRESHLY allocated arrays
 * is otherwise dominated by (SDFK_COPY_THREADS in the environment at start-up, SDFK_OPT_COPY_MODE).  The Python mirror allocates
 * Mesh.Vertices/Colors/Normals/Triangles here; a C# shim can do the same for Span<T>/Memory<T>
 * based accessors, while Mesh's public arrays (Mesh.cs:10-13) have to stay managed arrays.
 * sdfk_host_free returns the block to the arena; blocks that are still out at sdfk_shutdown stay valid (they are
 * not reclaimed: sdfk_host_free after a shutdown ignores them). */
int sdfk_host_alloc(int64_t n_bytes, void** out);
void sdfk_host_free(void* p);
/* Makes [p, p + n_bytes) of the caller's pageable memory (freshly allocated managed arrays, pinned for the call) present and
 * writable on the library's thread pool -- what sdfk_mesh_copy does to its destinations anyway, offered separately so that
 * the host can do it WHILE the GPU computes the mesh: sdfk_sample_march (returns at once) -> sdfk_mesh_size_hint ->
 * allocate Vertices / Colors / Normals / Triangles -> sdfk_host_prefault each -> sdfk_mesh_counts (waits) -> sdfk_mesh_copy. */
int sdfk_host_prefault(void* p, int64_t n_bytes);
/* Phases of the last staged device -> pageable-host copy (measurement): stats[5] = { bytes, ns until every chunk was queued,
 * ns until the destination pages were present, ns until done, ns of that spent waiting for the DMA }. */
int sdfk_copy_stats(int64_t stats[5]);
/* What the stream placement found (diagnostics): out[0] = 1 if measured, out[1] = classes found (streams of one class must not be
 * busy together: same hardware queue or same pipe), out[2] = class of lane 0 (the caller's stream), out[3..6] = classes of lanes
 * 1..4 (-1: no placed stream), out[7] = class of the exchange stream of a sharded rank (-1: none kept). */
int sdfk_stream_placement(int32_t out[8]);

/* ---- Mesh (Mesh.cs:8-64) ----------------------------------------------------
 * Vertices/Colors/Normals: 3 floats each per vertex; Triangles: int32 indices
 * (Mesh.cs:10-13).  Vertices/Normals are already transformed to world space
 * (MarchingCubes.cs:85-90, Mesh.cs:47-64). */
int sdfk_mesh_counts(const sdfk_mesh* m, int64_t* n_vertices, int64_t* n_indices);
/* The same WITHOUT waiting: for a mesh whose job is still queued, the counts of the previous mesh of the same grid shape
 * (what its buffers were sized from: exact whenever the scene repeats), *exact = 0; for a finished mesh its counts, *exact = 1.
 * Lets a host allocate (and sdfk_host_prefault) the managed arrays of Mesh.cs:10-13 while the GPU still works. */
int sdfk_mesh_size_hint(const sdfk_mesh* m, int64_t* n_vertices, int64_t* n_indices, int32_t* exact);
int sdfk_mesh_bounds(const sdfk_mesh* m, float min[3], float max[3]);      /* Mesh.Measure */
int sdfk_mesh_copy(const sdfk_mesh* m, float* vertices3, float* colors3, float* normals3,
                   int32_t* triangles);                                      /* any may be NULL */
int sdfk_mesh_device_ptrs(const sdfk_mesh* m, void** vertices3, void** colors3, void** normals3,
                          void** triangles);
/* device-to-device copy into caller-owned device buffers (e.g. torch tensors used as RCCL
 * all-gather inputs), asynchronous on the library stream; any pointer may be NULL */
int sdfk_mesh_copy_device(const sdfk_mesh* m, void* vertices3, void* colors3, void* normals3,
                          void* triangles);
/* Mesh.Transform(Matrix4x4) (Mesh.cs:47-64), in place on the device-resident mesh: positions by Vector3.Transform with
 * `matrix`, normals by Vector3.TransformNormal with `normal_matrix` and Vector3.Normalize, then Mesh.Measure.  Both
 * matrices row-major M11..M44 (row-vector convention of System.Numerics); normal_matrix = Transpose(Invert(matrix with
 * M41 = M42 = M43 = 0, M44 = 1)) exactly as Mesh.cs:49-55 derives it -- BCL calls the shim makes with the BCL itself.
 * Works on every mesh sdfk_march / sdfk_sample_march / sdfk_dist_mesh return, whatever the call history (a handle that borrows a
 * captured job's arrays is transformed in place: the job is busy until the handle is freed); SDFK_ERR_UNSUPPORTED only for the
 * per-rank slab meshes of sdfk_sample_march_slab whose arrays are sections of a gather buffer. */
int sdfk_mesh_transform(sdfk_mesh* m, const float matrix[16], const float normal_matrix[16]);
/* diagnostics: number of active cells, and of case-13 cells with no tiling
 * ("Impossible case 13?", MarchingCubes.cs:365) seen while meshing */
int sdfk_mesh_stats(const sdfk_mesh* m, int64_t* n_active_cells, int64_t* n_case13_cells);
void sdfk_mesh_free(sdfk_mesh* m);

/* ---- measurement hooks (bench.py) ------------------------------------------
 * on = 1: every kernel launch is bracketed by hipEvents on the launch stream (sdfk_profile_get).
 * on = 2: sdfk_sample launches the fused sampling kernel ONLY (no sign-bit transposition, cached
 *         views left invalid), so that the caller can time K back-to-back launches of that one
 *         kernel between two events of its own.  on = 0: normal operation. */
int sdfk_profile_enable(int32_t on);
int sdfk_profile_reset(void);
/* number of distinct kernels recorded; fills name/total milliseconds/launch count */
int sdfk_profile_count(void);
int sdfk_profile_get(int32_t i, const char** name, double* total_ms, int64_t* launches);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SDFKIT_HIP_H */
