// lib_internal.h -- what the translation units of libsdfkit_hip.so share: configuration, the device contexts (streams, lanes, pools, result
// slots), the opaque handles of the C ABI (include/sdfkit_hip.h) and the internal functions that cross a file boundary.
//   lib_context.hip  errors, configuration, contexts / streams / lanes / phase tokens / stream placement, options, copies, pinned arena
//   lib_jit.hip      SDF programs: source generation, hiprtc, the on-disk code-object cache
//   lib_volume.hip   volumes, Voxels.SampleSdf (+ ClipToBounds), SdfEx.Sample, RayMarcher
//   lib_march.hip    MarchingCubes.CreateMesh: the job driver, deferred completion, captured graphs, sdfk_sample_march, slab forms
//   lib_mesh.hip     the accessors of a device-resident Mesh
//   lib_dist.hip     the Z-slab sharded step (dist_rccl.h) and several GPUs from one process (node_local.h)
//   mc_kernels.hip   the marching-cubes kernels (declared in mc_kernels.h)
// Build: sdfkit_amd/build.py (the seven units in parallel, then one link with csrc/exports.map).  gfx950 only; there is no CPU path.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <rccl/rccl.h>   // (types only: the library is loaded with dlopen when a process shards, dist_rccl.h)

#include "../../include/sdfkit_hip.h"
#include "mc_kernels.h"
#include "sample_codegen.h"

using namespace sdfk;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
extern thread_local std::string t_err;
// x rows per wavefront of sdfk_sample_bits (compiled into the programs: 1 and 4 were measured slower)
static constexpr int kSampleRpw = 2;

// Run-time configuration.  The environment supplies DEFAULTS, read ONCE (config_from_env, called by sdfk_init and by the
// device-less sdfk_program_check); afterwards only sdfk_set_option / sdfk_set_cache_dir change it.  Nothing else in this
// file calls getenv.

struct Config {
    bool loaded = false;
    int lanes = 3;            // SDFK_OPT_LANES
    int tokens = -1;          // SDFK_OPT_TOKENS (-1: by grid size)
    int graphs = 1;           // SDFK_OPT_GRAPHS
    int copy_mode = 1;        // SDFK_OPT_COPY_MODE
    int corner_eval = 1;      // SDFK_OPT_CORNER_EVAL
    int vcolor_eval = 1;      // SDFK_OPT_VCOLOR_EVAL
    int dist_exchange = 0;    // SDFK_OPT_DIST_EXCHANGE (0: ncclAllGather; the direct / gather-to-root exchanges are opt-ins)
    int dist_lanes = 3;       // SDFK_OPT_DIST_LANES
    int dist_index16 = 0;     // SDFK_OPT_DIST_INDEX16
    int code_cache = 1;       // SDFK_OPT_CODE_CACHE
    int idle_programs = 32;   // SDFK_OPT_IDLE_PROGRAMS
    int color_passes = 0;     // SDFK_OPT_COLOR_PASSES: 0 = by the program's size and the grid's (default), 1 = always one pass, 2 = always two
    int elide_volume = 2;     // SDFK_OPT_ELIDE_VOLUME (default: the temporary volume of sdfk_sample_march is not stored, blocks are culled)
    int prefault_huge = 0;    // SDFK_OPT_PREFAULT_HUGE
    int place_streams = 1;    // SDFK_OPT_STREAM_PLACEMENT
    int idle_lane = 1;        // SDFK_OPT_IDLE_LANE
    int copy_threads = 0;     // SDFK_COPY_THREADS (0: min(16, cores / 2)); fixed once the pool has started
    int sample_mode = -1;     // SDFK_SAMPLE_MODE (debugging: force the row-tiled (0) / plane-chunk (1) sampler)
    int hw_queues = 0;        // GPU_MAX_HW_QUEUES as the process had it when the library initialised (0: unset)
    std::string cache_dir;    // resolved lazily (cache_dir()); "" = default resolution
    bool cache_dir_set = false;
    std::string jit_flags;    // SDFK_JIT_FLAGS: extra hiprtc options (space separated)
    std::string dump_source;  // SDFK_DUMP_SOURCE: file that receives the generated source of the last program
    std::string env_cache_dir, env_xdg, env_home, rccl_lib;
};
extern Config g_cfg;

int fail(int code, const char* fmt, ...);
#define HIPCHK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(SDFK_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---------------------------------------------------------------------------
// context: device, stream, caching device allocator, profiling events
// ---------------------------------------------------------------------------
struct sdfk_mesh;
struct sdfk_volume;
struct ProgCode;



struct ProfSpan { int name_id; hipEvent_t a, b; };

struct Context {
    bool inited = false;
    int device = -1;
    hipStream_t stream = nullptr;        // the stream work is queued on right now (= the current lane's)
    hipStream_t user_stream = nullptr;   // lane 0: the caller's stream (sdfk_set_stream) or own_stream
    hipStream_t own_stream = nullptr;
    // Lanes.  Lane 0 is the caller-visible stream.  Lanes 1..NSIDE are internal streams that
    // self-contained jobs (sdfk_sample_march: no input but the program, output read only after
    // a host-side wait) alternate between, so that the store-bound sampling kernel of one job
    // overlaps the latency-bound marching-cubes kernels of the previous one.
    // The caching allocator is stream-ordered PER LANE: a block goes back to the pool of the
    // lane it was allocated on and is only handed out again to work queued on that lane.
    static constexpr int NSIDE = 4;
    // (clean_cull_headers: counter blocks of the culling kernel that are known to be all zero IN THIS LANE'S STREAM ORDER -- the count pass
    // of the job that used one last cleared it --, so that a volume-less job needs no memset in front of its first kernel)
    struct Lane { hipStream_t stream = nullptr; std::multimap<size_t, void*> free_blocks; std::vector<uint32_t*> clean_cull_headers; };
    Lane lanes[1 + NSIDE];
    // stream placement (place_streams): the streams the library made for its lanes and the exchange, the class -- set of
    // streams that must not be busy together -- each was measured to be in, and who uses which
    struct Placed { hipStream_t s; int cls; int user; };   // user: 0 none, 1..NSIDE lane, 100 exchange
    std::vector<Placed> pool;
    std::map<hipStream_t, int> foreign_cls;   // classes of caller streams seen by sdfk_set_stream
    int cls_lane0 = -1, n_classes = 0;
    bool placed = false;
    int* spin_sink = nullptr;
    hipEvent_t lane_done[1 + NSIDE] = {};   // reused by sdfk_lane_end
    int cur_lane = 0;
    int side_lanes = 2;         // SDFK_LANES=0 disables the side lanes (everything on lane 0)
    int next_side = 0;
    // phase tokens of the self-contained jobs on the lanes (see phase_token): kind 0 = sampling kernel, 1 = k_vertices
    struct Token { hipEvent_t ring[8] = {}; int next = 0; hipEvent_t last = nullptr; int last_lane = -1; };
    Token tokens[2];
    int token_mask = 0;         // kinds active for the job being queued (set by sdfk_sample_march)
    struct Block { size_t size; int lane; };
    std::map<void*, Block> live_blocks;
    // profiling
    bool prof_on = false;
    bool sampler_only = false;   // sdfk_profile_enable(2): sdfk_sample launches the fused sampling kernel only
    std::vector<std::string> prof_names;
    std::vector<double> prof_ms;
    std::vector<int64_t> prof_n;
    std::vector<ProfSpan> prof_pending;
    std::vector<hipEvent_t> prof_event_pool;
    // pinned, device-mapped result slots: kernels mirror their counters / mesh bounds here,
    // the host reads them after its single stream sync (no copy kernel, no memset)
    struct HostSlot { McCounters c; float bounds[8]; };
    static constexpr int NSLOTS = 64;
    HostSlot* slots = nullptr;      // host view
    HostSlot* slots_dev = nullptr;  // device view
    int slot_next = 0;
    // A result slot belongs to ONE job from its creation until the job is released; a slot whose
    // job was dropped with kernels still queued (an unread mesh was freed, a sharded step retired
    // its mesh right after packing it) is handed out again once the event recorded on the job's
    // lane at drop time has completed (those kernels still write their counters into it): no
    // stream is ever synchronised for that.
    struct SlotState { bool busy = false; bool drop_pending = false; hipEvent_t dropped = nullptr; };
    SlotState slot_state[NSLOTS];
    // pinned staging for sdfk_mesh_copy / sdfk_volume_download (grown on demand, kept)
    void* stage = nullptr;
    size_t stage_bytes = 0;
    // pinned host arena (sdfk_host_alloc): size-class free lists like the device pool; a block in
    // `host_live` is in the caller's hands
    // (the arena itself is process-wide: HostArena below -- a block may be freed, or be the destination of a copy, in any context)
    // sizes seen last time for a (shape, iso-independent) key: lets a repeat call launch the
    // whole pipeline speculatively and synchronise once
    struct Hint { uint32_t n_active, nv, ni; };
    std::map<uint64_t, Hint> hints;
    // meshes returned by the speculative path whose kernels may still be queued (oldest first)
    std::deque<sdfk_mesh*> pending;
    static constexpr size_t MAX_PENDING = 6;
    // captured launch graphs of repeat sdfk_sample_march jobs on launch-bound grids (struct GraphJob below)
    std::vector<struct GraphJob*> graph_jobs;
    uint64_t graph_clock = 0;
    size_t graph_bytes = 0;
    int64_t graph_launches = 0;
    std::map<uint64_t, uint32_t> graph_sightings;   // full job key (+ lane) -> times asked for without a captured job
};

// ---- the sharding state of a device context (dist_rccl.h) ----------------------------------------------------------------
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct DistContext {
    int backend = 0;   // 0: none, 1: RCCL, 2: host transport
    int world = 1, rank = 0;
    RcclApi nccl;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;        // the exchange stream
    bool stream_owned = false;           // (created here, not one of the placed streams of lib_context.hip)
    sdfk_allgather_fn host_fn = nullptr;
    void* host_ctx = nullptr;
    int64_t* agree_dev = nullptr;        // [1 + world]
    int64_t* agree_host = nullptr;       // pinned, [1 + world]
    char* stage = nullptr;               // host transport: pinned staging, (1 + world) x stage_stride
    int64_t stage_stride = 0;
    int sessions = 0;
    // ranks that are threads of ONE process (node_local.h) can agree on a status for the price of a thread barrier: SlabOps::consensus
    int (*consensus_fn)(void* ctx, int mine) = nullptr;
    void* consensus_ctx = nullptr;
};

// ---- device contexts --------------------------------------------------------------------------------------------------------
// Everything the library keeps per GPU -- streams, lanes, pools, result slots, captured jobs, loaded kernel modules, the sharding
// state -- lives in a DeviceState, and every THREAD has a current one, exactly like the HIP runtime's current device: sdfk_init(d)
// creates the context of device d (once) and makes it the calling thread's; a thread that never called sdfk_init uses the context
// of the first device the process initialised (a host whose calls arrive on thread-pool threads keeps working as in rounds 1-4).
// Handles (programs, volumes, meshes, sessions) belong to the context they were made in and are used by threads whose current
// context that is.  One process can therefore drive several GPUs, one host thread per device (sdfk_node_*, dist_rccl.h; the
// reference is a library one .NET process calls: Sdf.cs:59-63); each context has its own lock, so the threads do not serialise.
struct DeviceState {
    Context ctx;
    std::recursive_mutex mu;
    DistContext dist;
    std::unordered_map<std::string, ProgCode*> codes;   // kernel sets of the program structures loaded on this device (modules are per device)
    uint64_t code_clock = 0;
    int graph_build_failures = 0;
    int claimed_device = -1;   // the device sdfk_init is making / has made this context for (-1: free); under g_registry_mu
    bool listed = true;        // found by sdfk_init(device) (false: the private context of a local node's virtual rank)
};
extern DeviceState g_state0;                                   // the first device's context (storage; `inited` says whether it is in use)
extern std::atomic<DeviceState*> g_default_state;   // current context of threads that never chose one (written under g_registry_mu,
                                                        // read without it by every call of such a thread: atomic)
extern std::mutex g_registry_mu;                               // guards g_states / writes of g_default_state / process-wide settings
extern std::atomic<int> g_contexts_up;                      // initialised contexts of the process, listed or private (a node's ranks)
extern std::vector<DeviceState*> g_states;          // the contexts sdfk_init made, by device (a local node's private ones are not listed)
extern thread_local DeviceState* t_state;
inline DeviceState& cur_state() { return *(t_state ? t_state : g_default_state.load(std::memory_order_acquire)); }
// (the names the rest of this file has always used for "the" context, its lock and its sharding state)
#define g (cur_state().ctx)
#define g_mu (cur_state().mu)
#define gd (cur_state().dist)
#define g_codes (cur_state().codes)
#define g_code_clock (cur_state().code_clock)
#define g_graph_build_failures (cur_state().graph_build_failures)
// Pinned host arena (sdfk_host_alloc): process-wide -- pinned memory belongs to no device context (hipHostMallocPortable), and a host
// may free a block, or name it as the destination of a copy, from a thread whose current context is another one than the allocator's.
struct HostArena {
    std::mutex mu;
    std::multimap<size_t, void*> free_blocks;   // size-class free lists, equal keys in order of return
    std::map<void*, size_t> live;               // blocks in the callers' hands
};
extern HostArena g_arena;
// the calling thread works in context `st` for the lifetime of the scope (accessors of a handle that belongs to another thread's
// context: the mesh a local node hands back, sdfk_node_to_mesh)
struct StateScope {
    DeviceState* saved;
    explicit StateScope(DeviceState* st) : saved(t_state) { if (st) t_state = st; }
    ~StateScope() { t_state = saved; }
};


hipStream_t lane_stream(int k);
int prof_name_id(const char* name);
hipEvent_t prof_event();
void prof_drain();

struct LaneScope {
    int saved;
    explicit LaneScope(int k) : saved(g.cur_lane) { g.cur_lane = k; g.stream = lane_stream(k); }
    ~LaneScope() { g.cur_lane = saved; g.stream = g.lanes[saved].stream; }
};

struct ProfScope {
    ProfSpan s;
    bool on;
    explicit ProfScope(const char* name) : on(g.prof_on && name != nullptr)   // (nullptr: no span)
    {
        if (on) {
            s.name_id = prof_name_id(name);
            s.a = prof_event();
            s.b = prof_event();
            (void)hipEventRecord(s.a, g.stream);
        }
    }
    ~ProfScope()
    {
        if (on) {
            (void)hipEventRecord(s.b, g.stream);
            g.prof_pending.push_back(s);
            if (g.prof_pending.size() > 4096) prof_drain();
        }
    }
};

class HostPool {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::function<void(int)> fn;
    int ntasks = 0, next = 0, running = 0;
    uint64_t generation = 0;
    bool stopping = false;

    void worker()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return stopping || (generation != seen && next < ntasks); });
            if (stopping) return;
            seen = generation;
            while (next < ntasks) {
                const int t = next++;
                running++;
                lk.unlock();
                fn(t);
                lk.lock();
                running--;
            }
            if (running == 0) cv_done.notify_all();
        }
    }

public:
    int size()
    {
        static const int n = [] {
            int v = g_cfg.copy_threads;
            if (v <= 0) v = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency() / 2));
            return std::min(v, 64);
        }();
        return n;
    }
    // starts `n` tasks fn(0..n-1) on the pool and returns; wait() blocks until they are done
    void start(int n, std::function<void(int)> f)
    {
        std::unique_lock<std::mutex> lk(mu);
        if (threads.empty())
            for (int i = 0; i < size(); i++) threads.emplace_back([this] { worker(); });
        fn = std::move(f);
        ntasks = n;
        next = 0;
        generation++;
        cv_work.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        while (next < ntasks) {   // the caller works too
            const int t = next++;
            running++;
            lk.unlock();
            fn(t);
            lk.lock();
            running--;
        }
        cv_done.wait(lk, [&] { return running == 0; });
        ntasks = 0;
    }
    ~HostPool()
    {
        {
            std::unique_lock<std::mutex> lk(mu);
            stopping = true;
            cv_work.notify_all();
        }
        for (auto& t : threads) t.join();
    }
};

extern HostPool g_pool;
extern std::recursive_mutex g_pool_mu;   // one client at a time (the pool is shared by the device contexts of the process)

struct CopyPiece { const void* src; void* dst; size_t bytes; };

// ---------------------------------------------------------------------------
// opaque objects
// ---------------------------------------------------------------------------
// Kernels of a program (bit k of SDFK_KERNELS in the generated source, csrc/sample_codegen.h)
// SDFK_OPT_COLOR_PASSES = 0: a colour volume is sampled in two passes when its program has at most kTwoPassMaxOps operations -- ONE primitive
// with a constant colour (Sdfs.Cylinder: 16, a sphere .WithColor: ~20) -- and the grid at least kTwoPassMinVoxels voxels.  The second pass
// evaluates one voxel per lane (one store per lane is what makes it a plain fill), without the fused kernel's sharing of everything
// that depends on x and y only among a lane's four z: measured at 512^3 (profiles/r06_ab_color_passes.txt, us per sampling, one / two
// passes): sphere with a constant colour 383 / 345, two coloured spheres under a Union (43 operations) 382 / 396, the README scene (64)
// 384 / 480, the 8-primitive union of BASELINE C4 (213) 389 / 578.
constexpr int kTwoPassMaxOps = 24;
constexpr size_t kTwoPassMinVoxels = size_t(1) << 21;
enum ProgKernel { PK_BITS = 0, PK_BITS_FLAT = 1, PK_SIGNS = 2, PK_BITS_CLIP = 3, PK_BITS_CLIP_FLAT = 4, PK_VCOLORS = 5, PK_CORNERS = 6, PK_RAYMARCH = 7,
                  PK_SIGNS_FLAT = 8, PK_CULL = 9, PK_EVAL_BLOCKS = 10, PK_POINTS = 11,
                  // two-pass sampling of a colour volume (SDFK_OPT_COLOR_PASSES): the fused samplers without their colour half, then the colours
                  PK_BITS_NC = 12, PK_BITS_NC_FLAT = 13, PK_BITS_NC_CLIP = 14, PK_BITS_NC_CLIP_FLAT = 15, PK_COLORS = 16, PK_COUNT = 17 };
inline bool pk_is_sampler(int k) { return k <= PK_BITS_CLIP_FLAT || k == PK_SIGNS_FLAT || k == PK_CULL || k == PK_EVAL_BLOCKS || (k >= PK_BITS_NC && k <= PK_BITS_NC_CLIP_FLAT); }

// The compiled kernels of one program STRUCTURE (opcodes, operand ids, outputs -- the generated source; a program's
// constants are kernel arguments, csrc/sample_codegen.h): shared by every program of that structure, so that a scene whose
// constants change per call (an animated radius, a parameter sweep) compiles ONCE -- in the reference Sdfs.Sphere(radius) is a
// closure and a new radius costs nothing (Sdf.cs:202-214).
struct ProgCode {
    std::string source;
    // The entry points are compiled ON DEMAND, one hiprtc module per kernel set: what a caller pays on the first call
    // is the sampler instantiation its grid needs + sdfk_corners_eval (they share a module), not all eight kernels
    // (512^3 sphere on the bench box: 150 instead of 310 ms; an 8-primitive union: a third).
    std::vector<hipModule_t> modules;
    hipFunction_t fn[PK_COUNT] = {};
    int refs = 0;            // programs of this structure that are alive
    uint64_t last_use = 0;   // (structures without a program are kept for a while: the next frame of an animation asks again)
};

struct sdfk_program {
    ProgCode* code = nullptr;
    std::vector<float> params;   // the constants, in the order of the K.k[] slots of the generated source (never empty)
    int writes_color = 0;
    int n_ops = 0;  // operations of the program (SDFK_OPT_COLOR_PASSES = 0 decides by it whether evaluating twice is cheap enough)
    int refs = 1;   // the caller's handle + volumes it has sampled + queued jobs that launch from its module
    bool orphaned = false;   // the caller's handle is gone (sdfk_program_destroy): captured jobs keyed on it can never be asked for again
    bool no_elide = false;   // a volume of this program had case-13 sign words (the dead-cell test reads voxels): its volumes are stored from then on
    void* kargs() const { return const_cast<float*>(params.data()); }   // the by-value SdfkK argument of every generated kernel
};

struct sdfk_volume {
    int nx = 0, ny = 0, nz = 0;       // local dims (nz = planes held)
    int nz_global = 0, z0 = 0;
    float gmin[3], gmax[3];
    float* values = nullptr;
    float* colors = nullptr;          // nullptr: colours are all zero
    // SDFK_OPT_ELIDE_VOLUME: a volume sdfk_sample_march made for itself and never hands out has NO storage for Values / Colors
    // (values == colors == nullptr): its sampler leaves the sign bits only, corners and vertex colours are re-evaluated.
    // elided_colors: the program writes colours (the mesh has a colour array although the volume has none).
    bool elided = false, elided_colors = false;
    uint32_t* cull_list = nullptr;    // SDFK_OPT_ELIDE_VOLUME = 2: the 64 sub-lists of undecided blocks, then their sub-box masks (sdfk_cull_blocks)
    uint32_t* cull_header = nullptr;  // ... and their 64 counters, 128 B apart (from the lane's clean blocks: Context::Lane)
    mutable bool cull_header_clean = false;   // all zero again (the count pass of the meshing job has been queued behind the kernels that used it)
    int cull_header_lane = 0;
    // sign bits (value > bits_iso) packed along X, written by the fused sampling kernel;
    // valid until Values change (upload / ClipToBounds)
    uint64_t* bits = nullptr;
    uint8_t* bits8 = nullptr;         // the sampling kernel's byte form of the same bits ([y][x/8][z])
    float bits_iso = 0.0f;
    bool bits_valid = false;
    // the program whose output `values` still is, with the arguments it ran with (nullptr once
    // the values may have changed): marching cubes then re-evaluates cell corners instead of
    // gathering them
    sdfk_program* sampled_by = nullptr;
    SampleArgs sampled_args;
    // Rows of `values` / `colors` are pitch() voxels long: nz rounded up to a multiple of 4, so that every row -- and
    // every 4-voxel group a lane of the sampling kernel stores -- is 16-byte aligned whatever nz is.
    int pitch() const { return (nz + 3) & ~3; }
    size_t nvox() const { return (size_t)nx * ny * nz; }          // voxels of the grid (what the host arrays hold)
    size_t nalloc() const { return (size_t)nx * ny * pitch(); }   // voxel slots of the device arrays
    int nxw() const { return (nx + 63) / 64; }
    int nx8() const { return (nx + 7) / 8; }
    int pitch8() const { return (nz + 3) & ~3; }   // bytes per row of bits8
    size_t nbitwords() const { return (size_t)nz * ny * nxw() + 8; }   // k_compact reads 4 words past a row pair
};

struct sdfk_mesh {
    DeviceState* owner = &cur_state();   // the device context the mesh was made in: its accessors work there, whichever thread calls them
    int64_t nv = 0, ni = 0;
    float* vertices = nullptr;
    float* colors = nullptr;
    float* normals = nullptr;
    int32_t* triangles = nullptr;
    float* bounds = nullptr;  // device float[6]
    float h_min[3] = {0, 0, 0}, h_max[3] = {0, 0, 0};
    bool bounds_valid = false;
    int64_t n_active = 0, n_case13 = 0;
    size_t cap_v = 0, cap_i = 0;   // allocated capacity (>= nv, ni)
    // Deferred completion.  The speculative path returns the mesh while its kernels are still
    // queued; the first accessor waits for `done`, checks the size guess against the counters
    // the kernels mirrored to the host and, if the guess was too small, redoes the job exactly.
    sdfk_march_job* pending = nullptr;
    hipEvent_t done = nullptr;
    const sdfk_volume* src = nullptr;   // the volume the job read (kept unchanged until resolved)
    bool owns_src = false;              // temporary volume of sdfk_sample_march / sdfk_march_host
    float iso = 0.0f;
    int step = 1, layer_begin = 0, layer_end = 0;
    int64_t vertex_base = 0;
    uint64_t key = 0;
    int status = 0;                     // sticky error of a failed resolution
    std::string error;
    bool has_colors = true;             // false: the source volume had no colours (Colors are all zero)
    bool colors_valid = true;           // false: has_colors is false AND the device array `colors` was never written (k_vertices skips the
                                        // all-zero colour stores, 12 bytes per vertex): whoever hands device colours out zeroes them first
    bool external = false;              // V / C / N / T are sections of a caller-owned slab payload (not freed here)
    struct GraphJob* graph_job = nullptr;   // the job came from a captured launch graph: `pending` and (while `borrowed`) the buffers are its
    bool borrowed = false;
    void* slab_header = nullptr;        // ... whose header k_triangles writes
    int lane = 0;                       // lane the buffers belong to
    bool used_on_main = false;          // lane-0 work (copies, packing, the caller) may still be reading them
};

struct sdfk_march_job {
    McParams P;
    McCounters c;
    // everything below is owned by the job
    std::vector<void*> owned;
    sdfk_volume* sub = nullptr;    // subsampled copy for step > 1
    int gnx, gny, gnz;             // global voxel dims for Mesh.Transform
    float gmin[3], gmax[3];
    bool finished = false;
    bool empty = false;
    bool have_bits = false;
    uint8_t* bits8 = nullptr;            // byte form of the sign bits (k_signbits8 -> k_bits_transpose), job-owned
    sdfk_program* eval_prog = nullptr;   // corners by re-evaluation (holds a reference)
    SampleArgs eval_args;
    bool colors_elided = false;        // the volume has no colour storage although its program writes colours (SDFK_OPT_ELIDE_VOLUME)
    int slot = -1;                 // index of the pinned result slot (owned until job_release)
    int lane = 0;                  // lane the job's kernels are queued on
    bool* cull_clean = nullptr;        // the source volume's "its culling counters are zero again": set when the count pass is queued
    float* bounds_partial = nullptr;   // per-workgroup AABB partials of k_vertices (allocated once per job: launch_emit is allocation-free after)
    int bounds_blocks = 0;
    uint2* vdesc = nullptr;        // (creator record, edge) per emitted vertex for sdfk_vertex_colors (same rule)
    size_t vdesc_cap = 0;
    size_t rec_first = 0;          // first entry of `owned` that belongs to the record arrays
};


int mesh_resolve(sdfk_mesh* m);
void graph_job_retire(sdfk_mesh* m, bool too_small);
void graph_jobs_destroy_all();
void graph_jobs_forget_volume(const sdfk_volume* v);
void graph_jobs_forget_program(const sdfk_program* p);
void dist_release();   // (dist_rccl.h)
bool dist_active();
void resolve_dependents(const sdfk_volume* v);
void free_mesh_buffers(sdfk_mesh* m);
void drop_source(sdfk_mesh* m);
void job_release(sdfk_march_job* j, bool kernels_may_be_queued = false);
void program_release(sdfk_program* p);
void volume_values_changed(sdfk_volume* v);
void codes_drop_idle();



struct GraphJob;
struct PostCompact { char* out; int64_t out_capacity; unsigned long long* ticket; };

// ---- functions shared by the translation units of the library (defined in lib_*.hip) ------------------------------------------
void codes_trim();
size_t size_class(size_t n);
int dev_alloc(void** p, size_t n);
void dev_free(void* p);
hipStream_t lane_stream(int k);
void phase_token_wait(int kind);
void phase_token_pass(int kind);
void sync_all_lanes();
double chains_us(hipStream_t a, hipStream_t b);
int stream_class(hipStream_t s, const std::vector<hipStream_t>& reps, double serial_us);
void assign_placed_streams();
void place_streams();
void replace_lane0(hipStream_t s0);
hipStream_t placed_exchange_stream();
int prof_name_id(const char* name);
hipEvent_t prof_event();
void prof_drain();
int require_init();
void bind_thread();
int grid_for(size_t work_items, int per_block = 256, int max_blocks = 256 * 8);
void prefault_start(void* p, size_t n);
int stage_reserve();
int copy_to_host(const std::vector<CopyPiece>& pieces, const std::function<void()>& beside = nullptr);
void config_from_env();
void config_from_env_once();
int context_init(int device);
DeviceState* context_claim(int device, bool listed);
void context_unclaim(DeviceState* st);
uint64_t fnv1a64(const std::string& s, uint64_t h);
std::string cache_dir();
bool cache_load(const std::string& path, const std::string& key, std::vector<char>& code);
void cache_store(const std::string& path, const std::string& key, const std::vector<char>& code);
int compile_source(const std::string& src, unsigned mask, std::vector<char>& code, bool use_cache, bool* from_cache = nullptr,
                          bool refresh = false);
void code_unload(ProgCode* c);
ProgCode* code_acquire(std::string&& src);
void code_release(ProgCode* c);
int generate_source(const sdfk_op* ops, int32_t n_ops, const int32_t out_rgbw[4], int32_t writes_color, std::string& src,
                           std::vector<float>* params = nullptr);
int program_fn(const sdfk_program* cp, int k, hipFunction_t* fn);
int job_volume_create(const sdfk_program* p, int nx, int ny, int nz, const float mn[3], const float mx[3], float iso, sdfk_volume** out);
int volume_materialize(sdfk_volume* v);
dim3 transpose_grid(int nz, int ny, int nxw);
dim3 flat_grid(size_t plane, int nx8);
void grid_constants(const sdfk_volume* v, float d[3], float m[3], float* outside);
int sample_impl(const sdfk_program* p, sdfk_volume* v, int32_t clip_to_bounds, float iso_hint);
int acquire_slot();
int alloc_records(sdfk_march_job* j, size_t c);
int launch_classify(sdfk_march_job* j, bool publish);
int wait_counters(sdfk_march_job* j);
int setup_job(const sdfk_volume* v, float iso, int step, int layer_begin, int layer_end, size_t cap_records,
              sdfk_march_job** out);
int alloc_mesh(sdfk_mesh** out, size_t cap_v, size_t cap_i);
int launch_emit(sdfk_march_job* j, sdfk_mesh* m, int64_t vertex_base);
void finalize_mesh(sdfk_march_job* j, sdfk_mesh* m, bool have_bounds);
uint64_t hint_key(const sdfk_volume* v, int step, int layer_begin, int layer_end);
int march_exact(const sdfk_volume* v, float iso, int step, int layer_begin, int layer_end, int64_t vertex_base,
                uint64_t key, sdfk_mesh** out);
bool external_mesh(char* dst, int64_t capacity, bool colors, uint32_t nv_hint, uint32_t ni_hint, sdfk_mesh** out);
int march_range(const sdfk_volume* v, float iso, int step, int layer_begin, int layer_end, int64_t vertex_base, sdfk_mesh** out,
                char* emit_dst = nullptr, int64_t emit_capacity = 0);
void graph_job_destroy(GraphJob* q);
bool graphs_enabled(int64_t nvox);
bool graphs_enabled_slab(int64_t nvox);
int graph_sample_march(const sdfk_program* p, const float mn[3], const float mx[3], int nx, int ny, int nz, int clip, float iso, sdfk_mesh** out);
int launch_post_compact(const PostCompact* pc, const void* plain, int64_t plain_capacity);
int graph_slab_enqueue(const sdfk_program* p, sdfk_volume* slab, int clip, float iso, int lb, int le, void* dst, int64_t capacity, bool* handled,
                       const PostCompact* pc = nullptr);
int slab_enqueue_impl(const sdfk_program* p, sdfk_volume* slab, int32_t clip_to_bounds, float iso_value, int32_t layer_begin,
                             int32_t layer_end, void* dst, int64_t capacity_bytes, int32_t lane, void* wait_hip_event, bool caller_stream_waits,
                             const PostCompact* pc = nullptr);
int slabs_rebase(void* gathered, int32_t world, int64_t stride_bytes, void* headers_mirror);
int eval_points_launch(const sdfk_program* p, const float* points_dev, int64_t n, float* rgbw_dev);
int raymarch_launch(const sdfk_program* p, int32_t width, int32_t height, const float cam[3], const float vpi[16],
                           float nearp, float farp, int32_t iters, float* depth_dev, float* rgb_dev);

// dist_rccl.h / node_local.h (lib_dist.hip)
bool dist_active();
void dist_release();
