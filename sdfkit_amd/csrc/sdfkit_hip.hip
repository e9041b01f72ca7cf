// sdfkit_hip.hip -- C ABI (include/sdfkit_hip.h) and host-side driver of libsdfkit_hip.so.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared ... -lhiprtc
// (see sdfkit_amd/build.py).  gfx950 only; there is no CPU path in this library.
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
#include "mc_kernels.hip"
#include "sample_codegen.h"

using namespace sdfk;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string t_err;
// x rows per wavefront of sdfk_sample_bits (compiled into the programs: 1 and 4 were measured slower)
static constexpr int kSampleRpw = 2;

// Run-time configuration.  The environment supplies DEFAULTS, read ONCE (config_from_env, called by sdfk_init and by the
// device-less sdfk_program_check); afterwards only sdfk_set_option / sdfk_set_cache_dir change it.  Nothing else in this
// file calls getenv.
namespace {
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
Config g_cfg;
}  // namespace
static int fail(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    t_err = buf;
    return code;
}
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

namespace {

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
    bool stream_owned = false;           // (created here, not one of the placed streams of sdfkit_hip.hip)
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
DeviceState g_state0;                                   // the first device's context (storage; `inited` says whether it is in use)
std::atomic<DeviceState*> g_default_state{&g_state0};   // current context of threads that never chose one (written under g_registry_mu,
                                                        // read without it by every call of such a thread: atomic)
std::mutex g_registry_mu;                               // guards g_states / writes of g_default_state / process-wide settings
std::atomic<int> g_contexts_up{0};                      // initialised contexts of the process, listed or private (a node's ranks)
std::vector<DeviceState*> g_states{&g_state0};          // the contexts sdfk_init made, by device (a local node's private ones are not listed)
thread_local DeviceState* t_state = nullptr;
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
HostArena g_arena;
// the calling thread works in context `st` for the lifetime of the scope (accessors of a handle that belongs to another thread's
// context: the mesh a local node hands back, sdfk_node_to_mesh)
struct StateScope {
    DeviceState* saved;
    explicit StateScope(DeviceState* st) : saved(t_state) { if (st) t_state = st; }
    ~StateScope() { t_state = saved; }
};

size_t size_class(size_t n)
{
    size_t c = 256;
    while (c < n) c = (c < (size_t(1) << 26)) ? c * 2 : c + (size_t(1) << 26);  // pow2 up to 64 MiB, then 64 MiB steps
    return c;
}

int dev_alloc(void** p, size_t n)
{
    if (n == 0) n = 1;
    const size_t c = size_class(n);
    auto& pool = g.lanes[g.cur_lane].free_blocks;
    auto it = pool.find(c);
    if (it != pool.end()) {
        *p = it->second;
        pool.erase(it);
        g.live_blocks[*p] = Context::Block{c, g.cur_lane};
        return SDFK_OK;
    }
    // (volumes as UNCACHED device memory, hipDeviceMallocUncached, were measured and dropped: the 512^3 sampling kernel alone
    // 82 -> 75.6 us, the pipelined step unchanged, 1024^3 4 % slower -- DESIGN.md section 5)
    hipError_t e = hipMalloc(p, c);
    if (e != hipSuccess) {
        // drop the caches and retry once (hipFree waits for the device: no block is in use after it)
        for (auto& lane : g.lanes) {
            for (auto& kv : lane.free_blocks) (void)hipFree(kv.second);
            lane.free_blocks.clear();
        }
        e = hipMalloc(p, c);
        if (e != hipSuccess) return fail(SDFK_ERR_NOMEM, "hipMalloc(%zu) failed: %s", c, hipGetErrorString(e));
    }
    g.live_blocks[*p] = Context::Block{c, g.cur_lane};
    return SDFK_OK;
}

void dev_free(void* p)
{
    if (!p) return;
    auto it = g.live_blocks.find(p);
    if (it == g.live_blocks.end()) return;
    g.lanes[it->second.lane].free_blocks.emplace(it->second.size, p);
    g.live_blocks.erase(it);
}

// queue on lane `k` for the lifetime of the scope (allocations included)
// The stream of lane k, created on first use.  Every stream of a process takes one of the runtime's in-order hardware
// queues (shared once there are more streams than GPU_MAX_HW_QUEUES), and this GPU serves about eight queues well: a
// sharded rank uses three lanes, a single-GPU caller three -- the fourth never exists (four are slower than three wherever
// it was measured: tools/slab_chain_probe.py).  (Tried instead, both far worse for
// the pipelined sharded step: lanes on a stream priority of their own, 200 us per step where plain streams reach 55;
// lanes with dedicated queues through hipExtStreamCreateWithCUMask, 110-290 us.)
hipStream_t lane_stream(int k)
{
    if (!g.lanes[k].stream && k > 0) {
        if (hipStreamCreateWithFlags(&g.lanes[k].stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            g.lanes[k].stream = nullptr;
            return g.lanes[0].stream;   // (no stream to be had: the caller's stream -- correct, just not concurrent)
        }
    }
    return g.lanes[k].stream;
}

// Phase tokens.  Left alone, the jobs on the three lanes fall into lock-step (a kernel trace shows pairs of sampling kernels
// starting together, then pairs of k_vertices): two sampling kernels side by side share the HBM write rate and take 182 us
// where one takes 80, two k_vertices share the CUs and take 135-143 us where one takes 57 -- same-phase overlap is a loss
// for exactly these two kernels.  A token per phase keeps them apart: the kernel waits for the event recorded after the
// previous job's kernel of the same kind (on another lane), so that a sampling kernel runs beside the meshing kernels of
// the other jobs instead of beside another sampling kernel.
void phase_token_wait(int kind)
{
    Context::Token& t = g.tokens[kind];
    if (!((g.token_mask >> kind) & 1) || !t.last || t.last_lane == g.cur_lane) return;
    (void)hipStreamWaitEvent(g.stream, t.last, 0);
}
void phase_token_pass(int kind)
{
    Context::Token& t = g.tokens[kind];
    if (!((g.token_mask >> kind) & 1)) return;
    hipEvent_t& e = t.ring[t.next];
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); e = nullptr; return; }
    if (hipEventRecord(e, g.stream) != hipSuccess) { (void)hipGetLastError(); return; }
    t.last = e;
    t.last_lane = g.cur_lane;
    t.next = (t.next + 1) % 8;
}

struct LaneScope {
    int saved;
    explicit LaneScope(int k) : saved(g.cur_lane) { g.cur_lane = k; g.stream = lane_stream(k); }
    ~LaneScope() { g.cur_lane = saved; g.stream = g.lanes[saved].stream; }
};

void sync_all_lanes()
{
    for (auto& lane : g.lanes)
        if (lane.stream) (void)hipStreamSynchronize(lane.stream);
}

// ---------------------------------------------------------------------------
// stream placement
// ---------------------------------------------------------------------------
// Which streams are busy TOGETHER decides everything about overlap on this part (tools/ubench/ub_lanes2.hip: chains of 9 small
// dependent kernels, one captured graph launch each, dealt round-robin over a subset of 8 streams created in order, with
// GPU_MAX_HW_QUEUES = 8): streams 0,1,2,3 -> 16 us per chain (one stream: 61), but streams 0 and 4 ALONE -> 191 us, three
// times slower than one stream: the runtime gives the i-th stream the i-th hardware queue, the queues are dealt over FOUR
// pipes, and a pipe that has two queues with work switches between them at ~15 us a switch.  (And with the runtime's default
// of 4 queues streams 0 and 4 share a QUEUE: in order, the one behind an event wait holds up the other.)  Which queue a stream
// gets depends on how many streams the process -- torch, the host, RCCL -- created before: one more stream in the process
// used to double the time of a sharded step.  So the library does not guess: it creates up to 7 streams when it initialises,
// MEASURES for each whether it runs side by side with the ones it keeps (two interleaved chains of 8-us kernels against the
// same kernels on one stream: side by side 0.5 x, same queue 1 x, same pipe 3 x) and sorts them into classes.  Lanes 1-3
// get one stream each from three classes other than lane 0's (the caller's stream may be busy too); the exchange stream of a
// sharded rank and lane 4 come from lane 0's class (a sharded rank's own stream is idle during steps; four lanes are for
// callers whose own stream is).  ~10 ms at sdfk_init; SDFK_OPT_STREAM_PLACEMENT = 0: streams as they come (round 2's behaviour).
double chains_us(hipStream_t a, hipStream_t b)
{
    double best = 1e30;
    for (int rep = 0; rep < 2; rep++) {
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 6; i++) {
            hipLaunchKernelGGL(k_spin, dim3(32), dim3(64), 0, a, 800, g.spin_sink);
            hipLaunchKernelGGL(k_spin, dim3(32), dim3(64), 0, b, 800, g.spin_sink);
        }
        (void)hipStreamSynchronize(a);
        (void)hipStreamSynchronize(b);
        best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    (void)hipGetLastError();
    return best;
}

// class of stream `s` among the classes whose representatives are `reps` (-1: side by side with all of them)
int stream_class(hipStream_t s, const std::vector<hipStream_t>& reps, double serial_us)
{
    for (size_t c = 0; c < reps.size(); c++)
        if (reps[c] == s || chains_us(s, reps[c]) > 0.75 * serial_us) return (int)c;
    return -1;
}

void assign_placed_streams()
{
    for (auto& q : g.pool) q.user = 0;
    auto take = [&](int user, auto&& ok) -> hipStream_t {
        for (auto& q : g.pool)
            if (!q.user && ok(q)) { q.user = user; return q.s; }
        return nullptr;
    };
    bool used[16] = {};
    if (g.cls_lane0 >= 0 && g.cls_lane0 < 16) used[g.cls_lane0] = true;
    for (int k = 1; k <= 3 && k <= Context::NSIDE; k++) {
        hipStream_t st = take(k, [&](const Context::Placed& q) { return q.cls >= 0 && q.cls < 16 && !used[q.cls]; });
        if (!st) st = take(k, [&](const Context::Placed& q) { return q.cls != g.cls_lane0; });   // (fewer than four classes)
        if (!st) st = take(k, [&](const Context::Placed&) { return true; });
        if (st)
            for (auto& q : g.pool)
                if (q.s == st && q.cls >= 0 && q.cls < 16) used[q.cls] = true;
        g.lanes[k].stream = st;   // (null: lane_stream() creates one on first use)
    }
    // lane 0's class (or, failing that, a class no lane uses): the exchange stream first (a sharded rank), then lane 4
    auto beside_the_lanes = [&](const Context::Placed& q) { return q.cls == g.cls_lane0 || (q.cls >= 0 && q.cls < 16 && !used[q.cls]); };
    (void)take(100, beside_the_lanes);
    if (Context::NSIDE >= 4) g.lanes[4].stream = take(4, beside_the_lanes);
}

void place_streams()
{
    if (!g_cfg.place_streams || g.placed) return;
    g.placed = true;
    if (!g.spin_sink && hipMalloc((void**)&g.spin_sink, sizeof(int)) != hipSuccess) { (void)hipGetLastError(); g.spin_sink = nullptr; }
    hipStream_t s0 = g.lanes[0].stream;
    (void)chains_us(s0, s0);   // (first launches of the probe: code object load, queue creation)
    const double serial = chains_us(s0, s0);
    std::vector<hipStream_t> reps{s0};
    g.cls_lane0 = 0;
    g.pool.clear();
    int have_other = 0, have_same = 0;
    for (int n = 0; n < 7 && !(have_other >= 3 && have_same >= 2); n++) {
        hipStream_t st = nullptr;
        if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
        (void)chains_us(st, st);   // (its queue exists from here on)
        int c = stream_class(st, reps, serial);
        if (c < 0) { reps.push_back(st); c = (int)reps.size() - 1; have_other++; }
        else if (c == 0) have_same++;
        g.pool.push_back(Context::Placed{st, c, 0});
    }
    g.n_classes = (int)reps.size();
    assign_placed_streams();
}

// the caller's stream became lane 0: its class among the pool's classes decides anew who runs where
bool dist_active();   // (dist_rccl.h)
void replace_lane0(hipStream_t s0)
{
    if (!g_cfg.place_streams || !g.placed || g.pool.empty()) return;
    int c = -2;
    if (s0 == g.own_stream) c = 0;
    else {
        auto it = g.foreign_cls.find(s0);
        if (it != g.foreign_cls.end()) c = it->second;
    }
    if (c == -2) {
        std::vector<hipStream_t> reps((size_t)g.n_classes, nullptr);
        reps[0] = g.own_stream;
        for (const auto& q : g.pool)
            if (q.cls > 0 && q.cls < g.n_classes && !reps[q.cls]) reps[q.cls] = q.s;
        for (auto& r : reps) if (!r) r = g.own_stream;
        const double serial = chains_us(g.own_stream, g.own_stream);
        c = stream_class(s0, reps, serial);   // (-1: a class of its own -- every pool stream runs beside it)
        if (g.foreign_cls.size() < 64) g.foreign_cls[s0] = c;
    }
    if (c == g.cls_lane0 || dist_active()) return;   // (a sharded rank keeps its layout: its exchange stream is in use)
    sync_all_lanes();
    g.cls_lane0 = c;
    assign_placed_streams();
}

hipStream_t placed_exchange_stream()
{
    for (const auto& q : g.pool)
        if (q.user == 100) return q.s;
    return nullptr;
}

int prof_name_id(const char* name)
{
    for (size_t i = 0; i < g.prof_names.size(); i++)
        if (g.prof_names[i] == name) return (int)i;
    g.prof_names.push_back(name);
    g.prof_ms.push_back(0.0);
    g.prof_n.push_back(0);
    return (int)g.prof_names.size() - 1;
}

hipEvent_t prof_event()
{
    if (!g.prof_event_pool.empty()) {
        hipEvent_t e = g.prof_event_pool.back();
        g.prof_event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

void prof_drain()
{
    for (auto& s : g.prof_pending) {
        (void)hipEventSynchronize(s.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            g.prof_ms[s.name_id] += ms;
            g.prof_n[s.name_id] += 1;
        }
        g.prof_event_pool.push_back(s.a);
        g.prof_event_pool.push_back(s.b);
    }
    g.prof_pending.clear();
}

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

// The HIP current device is per THREAD, and the C# shim's callers may sit on thread-pool threads:
// every entry point that allocates, launches or copies comes through here, and a thread that has
// not been seen before is bound to the library's device first (otherwise a rank with device != 0
// would allocate and launch on device 0 against streams and modules of device N).
thread_local int t_bound_device = -1;
int require_init()
{
    if (!g.inited) return fail(SDFK_ERR_NO_DEVICE, "sdfk_init() has not been called or no HIP device is available");
    if (t_bound_device != g.device) {
        const hipError_t e = hipSetDevice(g.device);
        if (e != hipSuccess) return fail(SDFK_ERR_HIP, "hipSetDevice(%d): %s", g.device, hipGetErrorString(e));
        t_bound_device = g.device;
    }
    return SDFK_OK;
}

// entry points that cannot fail for lack of a device (frees, accessors of finished objects)
// still bind the calling thread: they may queue work (a deferred mesh is completed, an event is recorded)
void bind_thread()
{
    if (g.inited && t_bound_device != g.device && hipSetDevice(g.device) == hipSuccess) t_bound_device = g.device;
}

int grid_for(size_t work_items, int per_block = 256, int max_blocks = 256 * 8)
{
    size_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (size_t)max_blocks) b = max_blocks;
    return (int)b;
}

}  // namespace

// ---------------------------------------------------------------------------
// host-side copy helpers (sdfk_mesh_copy / sdfk_volume_download)
// ---------------------------------------------------------------------------
// What a managed caller hands over are freshly allocated, pageable arrays (Mesh.cs:10-13 are
// `new Vector3[n]` / `new int[n]`; the Python mirror's are numpy.empty): a device-to-host copy into
// them is dominated by first-touch page faults on ONE thread (512^3 sphere, 33 MB: 9 ms, against
// 0.65 ms into memory that has been touched).  Faults scale with threads, so a small persistent
// pool touches the destination pages (one write per page: the whole range is overwritten right
// after) while the previous array is still on the wire.
namespace {
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
HostPool g_pool;
std::recursive_mutex g_pool_mu;   // one client at a time (the pool is shared by the device contexts of the process)

// Makes the pages of [p, p + n) present and writable before the copy lands in them: slices are
// 2 MiB-aligned so that two threads never fault into the same page-table page (or the same
// transparent huge page).  MADV_POPULATE_WRITE (Linux 5.14+) faults a whole slice in one system call
// and does not modify the pages, so it may cover the partial first / last page; without it, one
// byte per page is written (inside the destination only: the whole range is overwritten right after).
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
void prefault_start(void* p, size_t n)
{
    if (!p || n == 0) { g_pool.start(0, [](int) {}); return; }
    constexpr size_t kSlice = size_t(2) << 20;
    char* base = (char*)p;
    const size_t first = (kSlice - ((uintptr_t)base & (kSlice - 1))) & (kSlice - 1);   // bytes up to the first 2 MiB boundary
    const size_t nslices = 1 + (n > first ? (n - first + kSlice - 1) / kSlice : 0);
    g_pool.start((int)nslices, [=](int t) {
        const size_t a = t == 0 ? 0 : first + (size_t)(t - 1) * kSlice;
        const size_t b = t == 0 ? std::min(first, n) : std::min(a + kSlice, n);
        if (b <= a) return;
        const uintptr_t page = 4096, lo = ((uintptr_t)base + a) & ~(page - 1), hi = ((uintptr_t)base + b + page - 1) & ~(page - 1);
        // SDFK_OPT_PREFAULT_HUGE (off by default): a slice that is a whole, aligned 2 MiB block of the destination is advised
        // MADV_HUGEPAGE first -- one fault instead of 512.  Measured on the bench box (THP and defrag both "madvise"): SLOWER,
        // 3.2 instead of 2.5 ms per 512^3 mesh -- the kernel compacts memory inside the fault to find the huge page.
        if (g_cfg.prefault_huge && t > 0 && b - a == kSlice) (void)madvise((void*)((uintptr_t)base + a), kSlice, MADV_HUGEPAGE);
        static std::atomic<int> have_populate{1};
        if (have_populate.load(std::memory_order_relaxed)) {
            if (madvise((void*)lo, hi - lo, MADV_POPULATE_WRITE) == 0) return;
            if (errno == EINVAL) have_populate.store(0, std::memory_order_relaxed);   // older kernel: touch instead
        }
        for (size_t o = a; o < b; o += page) *(volatile char*)(base + o) = 0;
        *(volatile char*)(base + b - 1) = 0;
    });
}

// Pinned staging RING for device -> pageable host copies: kStageSlots chunks of kStageChunk bytes, reused as their events
// complete (32 MiB of pinned memory whatever the size of the transfer -- a 512^3 volume with colours used to pin 2 GiB
// for good).  Allocated on first use; if hipHostMalloc fails the caller falls back to the runtime's own copy (mode 0).
constexpr size_t kStageChunk = size_t(4) << 20;
constexpr int kStageSlots = 8;
int stage_reserve()
{
    if (g.stage) return SDFK_OK;
    if (hipHostMalloc(&g.stage, kStageChunk * kStageSlots, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        g.stage = nullptr;
        return SDFK_ERR_NOMEM;
    }
    memset(g.stage, 0, kStageChunk * kStageSlots);   // touch it once, here
    g.stage_bytes = kStageChunk * kStageSlots;
    return SDFK_OK;
}

struct CopyPiece { const void* src; void* dst; size_t bytes; };
int64_t g_copy_stats[5] = {0, 0, 0, 0, 0};   // last staged copy: bytes, ns until queued / destination present / done, ns waiting for the DMA

// Device -> caller arrays.  Default (mode 1): everything goes to the pinned staging buffer in 4 MiB
// chunks (plain DMA at the link rate) and the pool copies the chunks that have arrived into the
// destination, which it has pre-faulted while the first chunks were on the wire.  The caller's
// pageable memory is never handed to the HIP runtime: its pin-on-the-fly path measured anywhere
// between 0.8 and 26 ms for the same 33 MB, depending on page size and history (tools/host_io_probe.py).
// mode 0: pre-fault on the pool, then the runtime's own copy; mode 2: the runtime's copy alone.
// `beside`: host work that does not touch the destinations (clearing the colour array of a mesh without colours): done while the
// DMA transfers are in flight where the copy is the runtime's own (resident / pinned destinations), before the copy otherwise
int copy_to_host(const std::vector<CopyPiece>& pieces, const std::function<void()>& beside = nullptr)
{
    int mode = g_cfg.copy_mode;   // SDFK_OPT_COPY_MODE
    size_t total = 0;
    for (auto& p : pieces) total += p.bytes;
    if (total == 0) { if (beside) beside(); return SDFK_OK; }
    bool pinned = true;   // every destination inside a block of the library's pinned arena: plain DMA, nothing to pre-fault
    for (auto& p : pieces) {
        if (!p.bytes) continue;
        std::lock_guard<std::mutex> al(g_arena.mu);
        auto it = g_arena.live.upper_bound(p.dst);
        if (it == g_arena.live.begin()) { pinned = false; break; }
        --it;
        if ((const char*)p.dst + p.bytes > (const char*)it->first + it->second) { pinned = false; break; }
    }
    // Destinations whose pages are all resident (arrays a managed heap has recycled, buffers the caller has used before): the
    // runtime's own copy is the fastest there -- 33 MB in 0.57 ms against 0.8-1.4 ms through the staging ring; it is FRESH
    // pages that it handles badly (2.3-3.0 ms, and anywhere up to 26 ms depending on page size and history), and those
    // take the staged path below.  One mincore() per destination decides (microseconds).
    bool resident = mode == 1 && !pinned && total >= (size_t(1) << 20);
    if (resident) {
        std::vector<unsigned char> vec;
        for (auto& p : pieces) {
            if (!p.bytes) continue;
            const uintptr_t pg = 4096, lo = (uintptr_t)p.dst & ~(pg - 1), hi = ((uintptr_t)p.dst + p.bytes + pg - 1) & ~(pg - 1);
            vec.resize((hi - lo) / pg);
            if (mincore((void*)lo, hi - lo, vec.data()) != 0) { resident = false; break; }
            for (unsigned char c : vec)
                if (!(c & 1)) { resident = false; break; }
            if (!resident) break;
        }
    }
    g_copy_stats[0] = (int64_t)total;
    g_copy_stats[1] = g_copy_stats[2] = g_copy_stats[3] = g_copy_stats[4] = resident ? -1 : 0;
    if (total < (size_t(1) << 20) || mode == 2 || pinned || resident) {   // small: nothing to gain from helpers
        hipError_t e = hipSuccess;
        for (auto& p : pieces)
            if (p.bytes && e == hipSuccess) e = hipMemcpyAsync(p.dst, p.src, p.bytes, hipMemcpyDeviceToHost, g.stream);
        if (beside) beside();   // (while the transfers run)
        if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
        if (e != hipSuccess) return fail(SDFK_ERR_HIP, "device-to-host copy: %s", hipGetErrorString(e));
        return SDFK_OK;
    }
    if (beside) beside();
    // (from here on the shared thread pool works for this copy: one client at a time.  The runtime's own copy above needs no pool:
    // the ranks of a local node copy their slabs side by side, each over its own PCIe link)
    std::lock_guard<std::recursive_mutex> pool_lk(g_pool_mu);
    if (mode == 1 && stage_reserve() != SDFK_OK) mode = 0;   // no pinned memory to be had: the runtime's copy still works
    if (mode == 1) {
        const auto tp0 = std::chrono::steady_clock::now();
        auto since = [&](std::chrono::steady_clock::time_point t) { return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t).count(); };
        // chunk k travels through ring slot k % kStageSlots: DMA into the slot, event; the pool copies the slot into the
        // destination (pre-faulted while the first chunks were on the wire) and only then is the slot's next DMA queued
        struct Chunk { char* dst; const char* src; size_t bytes; };
        std::vector<Chunk> chunks;
        for (auto& p : pieces)
            for (size_t o = 0; o < p.bytes; o += kStageChunk)
                chunks.push_back(Chunk{(char*)p.dst + o, (const char*)p.src + o, std::min(kStageChunk, p.bytes - o)});
        hipEvent_t ev[kStageSlots];
        for (auto& e : ev) e = prof_event();
        hipError_t e = hipSuccess;
        auto issue = [&](size_t k) {
            char* slot = (char*)g.stage + (k % kStageSlots) * kStageChunk;
            hipError_t r = hipMemcpyAsync(slot, chunks[k].src, chunks[k].bytes, hipMemcpyDeviceToHost, g.stream);
            if (r == hipSuccess) r = hipEventRecord(ev[k % kStageSlots], g.stream);
            return r;
        };
        size_t issued = 0;
        for (; issued < chunks.size() && issued < (size_t)kStageSlots && e == hipSuccess; issued++) e = issue(issued);
        if (e == hipSuccess) {
            g_copy_stats[0] = (int64_t)total;
            g_copy_stats[1] = since(tp0);   // chunks queued
            // pre-fault the whole destination while the first chunks travel
            for (auto& p : pieces) { prefault_start(p.dst, p.bytes); g_pool.wait(); }
            g_copy_stats[2] = since(tp0);   // destination present
            g_copy_stats[4] = 0;
            const int nt = g_pool.size() + 1;
            for (size_t k = 0; k < chunks.size() && e == hipSuccess; k++) {
                const auto tw = std::chrono::steady_clock::now();
                e = hipEventSynchronize(ev[k % kStageSlots]);
                g_copy_stats[4] += since(tw);   // waiting for the DMA
                if (e != hipSuccess) break;
                const Chunk c = chunks[k];
                const char* slot = (const char*)g.stage + (k % kStageSlots) * kStageChunk;
                const size_t per = (c.bytes + nt - 1) / nt;
                g_pool.start(nt, [=](int t) {
                    const size_t a = std::min((size_t)t * per, c.bytes), b = std::min(a + per, c.bytes);
                    if (b > a) memcpy(c.dst + a, slot + a, b - a);
                });
                g_pool.wait();
                if (issued < chunks.size()) e = issue(issued++);   // the slot is free again
            }
        }
        g_copy_stats[3] = since(tp0);       // done
        if (e != hipSuccess) (void)hipStreamSynchronize(g.stream);
        for (auto& x : ev) g.prof_event_pool.push_back(x);
        if (e != hipSuccess) return fail(SDFK_ERR_HIP, "device-to-host copy: %s", hipGetErrorString(e));
        return SDFK_OK;
    }
    // mode 0
    size_t k0 = 0;
    while (k0 < pieces.size() && pieces[k0].bytes == 0) k0++;
    prefault_start(pieces[k0].dst, pieces[k0].bytes);
    g_pool.wait();
    for (size_t k = k0; k < pieces.size(); k++) {
        size_t kn = k + 1;
        while (kn < pieces.size() && pieces[kn].bytes == 0) kn++;
        const bool more = kn < pieces.size();
        if (more) prefault_start(pieces[kn].dst, pieces[kn].bytes);
        hipError_t e = hipSuccess;
        if (pieces[k].bytes) {
            e = hipMemcpyAsync(pieces[k].dst, pieces[k].src, pieces[k].bytes, hipMemcpyDeviceToHost, g.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(g.stream);
        }
        if (more) g_pool.wait();
        if (e != hipSuccess) return fail(SDFK_ERR_HIP, "device-to-host copy: %s", hipGetErrorString(e));
        k = kn - 1;
    }
    return SDFK_OK;
}
}  // namespace

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
static bool pk_is_sampler(int k) { return k <= PK_BITS_CLIP_FLAT || k == PK_SIGNS_FLAT || k == PK_CULL || k == PK_EVAL_BLOCKS || (k >= PK_BITS_NC && k <= PK_BITS_NC_CLIP_FLAT); }

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

namespace {
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
}

// ---------------------------------------------------------------------------
// lifetime
// ---------------------------------------------------------------------------
extern "C" int sdfk_abi_version(void) { return SDFK_ABI_VERSION; }

extern "C" const char* sdfk_last_error(void) { return t_err.c_str(); }

// The ONE place the environment is read (see Config): start-up defaults of the options, the cache location, debugging aids.
static void config_from_env_once();
static void config_from_env()
{
    static std::once_flag once;
    std::call_once(once, config_from_env_once);
}
static void config_from_env_once()
{
    g_cfg.loaded = true;
    auto geti = [](const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; };
    auto gets = [](const char* name) { const char* e = getenv(name); return std::string(e ? e : ""); };
    // (a start-up default outside the range sdfk_set_option accepts for that option is ignored: the built-in default stands)
    auto ranged = [&](const char* name, int dflt, int lo, int hi) { const int v = geti(name, dflt); return v >= lo && v <= hi ? v : dflt; };
    g_cfg.lanes = ranged("SDFK_LANES", 3, 0, Context::NSIDE);
    g_cfg.tokens = ranged("SDFK_TOKENS", -1, -1, 3);
    g_cfg.graphs = ranged("SDFK_GRAPHS", 1, 0, 2);
    g_cfg.copy_mode = ranged("SDFK_COPY_MODE", 1, 0, 2);
    g_cfg.copy_threads = ranged("SDFK_COPY_THREADS", 0, 0, 256);
    g_cfg.corner_eval = geti("SDFK_NO_CORNER_EVAL", 0) ? 0 : 1;
    g_cfg.vcolor_eval = geti("SDFK_NO_VCOLOR_EVAL", 0) ? 0 : 1;
    g_cfg.dist_exchange = ranged("SDFK_DIST_EXCHANGE", 0, 0, 3);
    g_cfg.dist_lanes = ranged("SDFK_DIST_LANES", 3, 0, 3);
    g_cfg.dist_index16 = geti("SDFK_DIST_INDEX16", 0) ? 1 : 0;
    g_cfg.code_cache = geti("SDFK_NO_CACHE", 0) ? 0 : 1;
    g_cfg.idle_programs = ranged("SDFK_IDLE_PROGRAMS", 32, 0, 1024);
    g_cfg.elide_volume = ranged("SDFK_ELIDE_VOLUME", 2, 0, 2);
    g_cfg.color_passes = ranged("SDFK_COLOR_PASSES", 0, 0, 2);
    g_cfg.prefault_huge = geti("SDFK_PREFAULT_HUGE", 0) ? 1 : 0;
    g_cfg.place_streams = geti("SDFK_STREAM_PLACEMENT", 1) ? 1 : 0;
    g_cfg.idle_lane = geti("SDFK_IDLE_LANE", 1) ? 1 : 0;
    g_cfg.sample_mode = geti("SDFK_SAMPLE_MODE", -1);
    g_cfg.hw_queues = geti("GPU_MAX_HW_QUEUES", 0);
    g_cfg.env_cache_dir = gets("SDFK_CACHE_DIR");
    g_cfg.env_xdg = gets("XDG_CACHE_HOME");
    g_cfg.env_home = gets("HOME");
    g_cfg.jit_flags = gets("SDFK_JIT_FLAGS");
    g_cfg.dump_source = gets("SDFK_DUMP_SOURCE");
    g_cfg.rccl_lib = gets("SDFK_RCCL_LIB");
}

namespace {
// initialises the calling thread's current context (t_state is set) on `device`
int context_init(int device)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (g.inited) return SDFK_OK;
    // The library's lanes, the caller's stream and the exchange stream must not share hardware queues: the HIP runtime maps
    // all streams of a process onto GPU_MAX_HW_QUEUES (default 4) in-order queues, and a stream that waits for an event (a
    // lane section's end, a collective) then holds up every OTHER stream behind it in the same queue.  Nor may two of them
    // that are busy together sit on queues of the same PIPE ("stream placement" above measures both).
    // The runtime reads the variable when IT initialises, which may be long before this call (torch, the C# host), and a
    // library must not edit its process's environment under the feet of other threads: the HOST BINDINGS export
    // GPU_MAX_HW_QUEUES=8 before their first HIP call (sdfkit_amd/_native.py, shim/SdfKit.Hip/Native.cs, include/SdfKit.hpp);
    // sdfk_get_option(SDFK_OPT_HW_QUEUES) says what the process had when the library came up.
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(SDFK_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(SDFK_ERR_INVALID, "device %d out of range (%d devices)", device, n);
    HIPCHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(SDFK_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    HIPCHK(hipStreamCreateWithFlags(&g.own_stream, hipStreamNonBlocking));
    g.user_stream = g.stream = g.own_stream;
    g.lanes[0].stream = g.own_stream;
    // The lanes sdfk_sample_march rotates over get their streams NOW, placed by measurement (or, with the placement off,
    // simply created right after the library's own stream; a fourth lane's stream is then created when somebody asks for it).
    place_streams();   // (measured: "stream placement" above)
    for (int k = 1; k <= 3 && k <= Context::NSIDE; k++) (void)lane_stream(k);
    g.cur_lane = 0;
    HIPCHK(hipHostMalloc((void**)&g.slots, sizeof(Context::HostSlot) * Context::NSLOTS, hipHostMallocMapped));
    HIPCHK(hipHostGetDevicePointer((void**)&g.slots_dev, g.slots, 0));
    memset(g.slots, 0, sizeof(Context::HostSlot) * Context::NSLOTS);
    g.device = device;
    t_bound_device = device;
    g.inited = true;
    g_contexts_up.fetch_add(1);
    return SDFK_OK;
}

// a context for `device` becomes the calling thread's current one: the one sdfk_init made for that device before, a free one, or a new one
DeviceState* context_claim(int device, bool listed)
{
    std::lock_guard<std::mutex> rl(g_registry_mu);
    DeviceState* st = nullptr;
    if (listed)
        for (DeviceState* q : g_states)
            if (q->listed && q->claimed_device == device) st = q;
    // (a node rank's PRIVATE context never is the object the threads that never chose a context fall back to: they would silently
    // work inside rank 0's context -- they get the uninitialised default instead and fail loudly until somebody calls sdfk_init)
    DeviceState* const dflt = g_default_state.load(std::memory_order_relaxed);
    if (!st)
        for (DeviceState* q : g_states)
            if (q->claimed_device < 0 && (listed || q != dflt)) { st = q; break; }
    if (!st) {
        st = new DeviceState();
        g_states.push_back(st);
    }
    st->claimed_device = device;
    st->listed = listed;
    // the first LISTED context in use serves the threads that never chose (also when the default so far was free, or never came up)
    if (listed && (dflt->claimed_device < 0 || !dflt->listed)) g_default_state.store(st, std::memory_order_release);
    return st;
}

void context_unclaim(DeviceState* st)
{
    std::lock_guard<std::mutex> rl(g_registry_mu);
    st->claimed_device = -1;
    st->listed = true;
    if (g_default_state.load(std::memory_order_relaxed) == st)
        for (DeviceState* q : g_states)
            if (q->listed && q->claimed_device >= 0) { g_default_state.store(q, std::memory_order_release); break; }
}
}  // namespace

// sdfk_init(device): the context of `device` -- created and initialised on first use -- becomes the CALLING THREAD's current
// context (like hipSetDevice).  A process that only ever names one device behaves as in ABI 1-4; naming a second device no
// longer fails: one process may drive several GPUs, one host thread each (or one thread that switches with sdfk_init).
extern "C" int sdfk_init(int device)
{
    config_from_env();
    if (device < 0) return fail(SDFK_ERR_INVALID, "device %d out of range", device);
    DeviceState* prev = t_state;
    DeviceState* st = context_claim(device, true);
    t_state = st;
    const int r = context_init(device);
    if (r) {   // nothing half-made stays behind; the thread keeps the context it had
        bool inited;
        { std::lock_guard<std::recursive_mutex> lk(st->mu); inited = st->ctx.inited; }
        if (!inited) context_unclaim(st);
        t_state = prev;
    }
    return r;
}

extern "C" void sdfk_shutdown(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!g.inited) return;
    while (!g.pending.empty()) (void)mesh_resolve(g.pending.front());
    graph_jobs_destroy_all();
    codes_drop_idle();   // (kernel sets no program uses any more; those of live programs go with their last program)
    if (dist_active()) dist_release();   // (sessions that are still alive read as freed: sdfk_dist_session_free after a shutdown only deletes them)
    (void)hipStreamSynchronize(g.stream);
    prof_drain();
    for (auto e : g.prof_event_pool) (void)hipEventDestroy(e);
    g.prof_event_pool.clear();
    sync_all_lanes();
    for (auto& t : g.tokens) {   // (phase tokens: nothing is queued any more)
        for (auto& e : t.ring) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        t.last = nullptr; t.last_lane = -1; t.next = 0;
    }
    for (auto& lane : g.lanes) {
        for (auto& kv : lane.free_blocks) (void)hipFree(kv.second);
        lane.free_blocks.clear();
        lane.clean_cull_headers.clear();   // (still among the live blocks: freed with them)
    }
    for (auto& kv : g.live_blocks) (void)hipFree(kv.first);
    g.live_blocks.clear();
    for (int k = 1; k <= Context::NSIDE; k++) {
        if (g.lane_done[k]) (void)hipEventDestroy(g.lane_done[k]);
        g.lane_done[k] = nullptr;
        bool pooled = false;
        for (const auto& q : g.pool) pooled = pooled || q.s == g.lanes[k].stream;
        if (g.lanes[k].stream && !pooled) (void)hipStreamDestroy(g.lanes[k].stream);
        g.lanes[k].stream = nullptr;
    }
    for (auto& q : g.pool) (void)hipStreamDestroy(q.s);
    g.pool.clear();
    g.foreign_cls.clear();
    g.placed = false;
    g.cls_lane0 = -1;
    g.n_classes = 0;
    if (g.spin_sink) (void)hipFree(g.spin_sink);
    g.spin_sink = nullptr;
    g.cur_lane = 0;
    g.lanes[0].stream = nullptr;
    for (auto& st : g.slot_state) {
        if (st.dropped) (void)hipEventDestroy(st.dropped);
        st = Context::SlotState();
    }
    if (g.stage) (void)hipHostFree(g.stage);
    g.stage = nullptr;
    g.stage_bytes = 0;
    {   // the pinned arena goes with the LAST context of the process (blocks still in their owners' hands stay mapped -- arrays of the
        // host mirror may outlive the library state --: leaked, not freed)
        bool last = true;
        {
            std::lock_guard<std::mutex> rl(g_registry_mu);
            for (DeviceState* q : g_states) last = last && (q == &cur_state() || q->claimed_device < 0);
        }
        if (last) {
            std::lock_guard<std::mutex> al(g_arena.mu);
            for (auto& kv : g_arena.free_blocks) (void)hipHostFree(kv.second);
            g_arena.free_blocks.clear();
            g_arena.live.clear();
        }
    }
    if (g.slots) (void)hipHostFree(g.slots);
    g.slots = nullptr;
    g.slots_dev = nullptr;
    g.hints.clear();
    if (g.own_stream) (void)hipStreamDestroy(g.own_stream);
    g.own_stream = nullptr;
    g.stream = g.user_stream = nullptr;
    g.inited = false;
    g_contexts_up.fetch_sub(1);
    g.device = -1;
    context_unclaim(&cur_state());   // (the context object stays, free for the next sdfk_init)
}

extern "C" int sdfk_set_stream(void* hip_stream)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (int r = require_init()) return r;
    HIPCHK(hipStreamSynchronize(g.stream));
    g.user_stream = hip_stream ? (hipStream_t)hip_stream : g.own_stream;
    g.lanes[0].stream = g.user_stream;
    replace_lane0(g.user_stream);   // (the lanes keep clear of the queue / pipe the caller's stream sits on)
    g.stream = g.lanes[g.cur_lane].stream;
    return SDFK_OK;
}

// Lane sections: the calls between begin and end are queued on internal stream `lane`.
extern "C" int sdfk_lane_begin(int32_t lane, void* wait_hip_event)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (int r = require_init()) return r;
    if (lane < 1 || lane > Context::NSIDE) return fail(SDFK_ERR_INVALID, "sdfk_lane_begin: lane %d out of range 1..%d", lane, Context::NSIDE);
    if (g.cur_lane != 0) return fail(SDFK_ERR_INVALID, "sdfk_lane_begin: already inside a lane section");
    hipStream_t ls = lane_stream(lane);
    if (wait_hip_event) HIPCHK(hipStreamWaitEvent(ls, (hipEvent_t)wait_hip_event, 0));
    g.cur_lane = lane;
    g.stream = ls;
    return SDFK_OK;
}

extern "C" int sdfk_lane_end(int32_t caller_stream_waits)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (int r = require_init()) return r;
    if (g.cur_lane == 0) return fail(SDFK_ERR_INVALID, "sdfk_lane_end: not inside a lane section");
    const int lane = g.cur_lane;
    g.cur_lane = 0;
    g.stream = g.lanes[0].stream;
    if (caller_stream_waits) {
        hipEvent_t ev = g.lane_done[lane];
        if (!ev) {
            HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            g.lane_done[lane] = ev;
        }
        hipError_t e = hipEventRecord(ev, lane_stream(lane));
        if (e == hipSuccess) e = hipStreamWaitEvent(g.lanes[0].stream, ev, 0);
        if (e != hipSuccess) return fail(SDFK_ERR_HIP, "sdfk_lane_end: %s", hipGetErrorString(e));
    }
    return SDFK_OK;
}

extern "C" int sdfk_synchronize(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (int r = require_init()) return r;
    HIPCHK(hipStreamSynchronize(g.stream));
    sync_all_lanes();
    for (auto& st : g.slot_state) st.drop_pending = false;   // nothing is queued any more: dropped jobs' slots are reusable
    return SDFK_OK;
}

// ---------------------------------------------------------------------------
// options (include/sdfkit_hip.h: sdfk_option)
// ---------------------------------------------------------------------------
namespace { void codes_trim(); }
extern "C" int sdfk_set_option(int32_t key, int64_t value)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    config_from_env();
    auto in = [&](int64_t lo, int64_t hi) { return value >= lo && value <= hi; };
    switch (key) {
    case SDFK_OPT_LANES: if (!in(0, Context::NSIDE)) break; g_cfg.lanes = (int)value; return SDFK_OK;
    case SDFK_OPT_TOKENS: if (!in(-1, 3)) break; g_cfg.tokens = (int)value; return SDFK_OK;
    case SDFK_OPT_GRAPHS: if (!in(0, 2)) break; g_cfg.graphs = (int)value; return SDFK_OK;
    case SDFK_OPT_COPY_MODE: if (!in(0, 2)) break; g_cfg.copy_mode = (int)value; return SDFK_OK;
    case SDFK_OPT_CORNER_EVAL: if (!in(0, 1)) break; g_cfg.corner_eval = (int)value; return SDFK_OK;
    case SDFK_OPT_VCOLOR_EVAL: if (!in(0, 1)) break; g_cfg.vcolor_eval = (int)value; return SDFK_OK;
    case SDFK_OPT_DIST_EXCHANGE: if (!in(0, 3)) break; g_cfg.dist_exchange = (int)value; return SDFK_OK;
    case SDFK_OPT_DIST_LANES: if (!in(0, 3)) break; g_cfg.dist_lanes = (int)value; return SDFK_OK;
    case SDFK_OPT_DIST_INDEX16: if (!in(0, 1)) break; g_cfg.dist_index16 = (int)value; return SDFK_OK;
    case SDFK_OPT_STREAM_PLACEMENT: if (!in(0, 1)) break; g_cfg.place_streams = (int)value; return SDFK_OK;
    case SDFK_OPT_IDLE_LANE: if (!in(0, 1)) break; g_cfg.idle_lane = (int)value; return SDFK_OK;
    case SDFK_OPT_CODE_CACHE: if (!in(0, 1)) break; g_cfg.code_cache = (int)value; return SDFK_OK;
    case SDFK_OPT_IDLE_PROGRAMS: if (!in(0, 1024)) break; g_cfg.idle_programs = (int)value; codes_trim(); return SDFK_OK;
    case SDFK_OPT_ELIDE_VOLUME: if (!in(0, 2)) break; g_cfg.elide_volume = (int)value; return SDFK_OK;
    case SDFK_OPT_COLOR_PASSES: if (!in(0, 2)) break; g_cfg.color_passes = (int)value; return SDFK_OK;
    case SDFK_OPT_PREFAULT_HUGE: if (!in(0, 1)) break; g_cfg.prefault_huge = (int)value; return SDFK_OK;
    case SDFK_OPT_HW_QUEUES: return fail(SDFK_ERR_INVALID, "SDFK_OPT_HW_QUEUES is read-only");
    default: return fail(SDFK_ERR_INVALID, "sdfk_set_option: unknown option %d", key);
    }
    return fail(SDFK_ERR_INVALID, "sdfk_set_option(%d): value %lld out of range", key, (long long)value);
}

extern "C" int sdfk_get_option(int32_t key, int64_t* value)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!value) return fail(SDFK_ERR_INVALID, "sdfk_get_option: null argument");
    config_from_env();
    switch (key) {
    case SDFK_OPT_LANES: *value = g_cfg.lanes; break;
    case SDFK_OPT_TOKENS: *value = g_cfg.tokens; break;
    case SDFK_OPT_GRAPHS: *value = g_cfg.graphs; break;
    case SDFK_OPT_COPY_MODE: *value = g_cfg.copy_mode; break;
    case SDFK_OPT_CORNER_EVAL: *value = g_cfg.corner_eval; break;
    case SDFK_OPT_VCOLOR_EVAL: *value = g_cfg.vcolor_eval; break;
    case SDFK_OPT_DIST_EXCHANGE: *value = g_cfg.dist_exchange; break;
    case SDFK_OPT_DIST_LANES: *value = g_cfg.dist_lanes; break;
    case SDFK_OPT_DIST_INDEX16: *value = g_cfg.dist_index16; break;
    case SDFK_OPT_STREAM_PLACEMENT: *value = g_cfg.place_streams; break;
    case SDFK_OPT_IDLE_LANE: *value = g_cfg.idle_lane; break;
    case SDFK_OPT_CODE_CACHE: *value = g_cfg.code_cache; break;
    case SDFK_OPT_IDLE_PROGRAMS: *value = g_cfg.idle_programs; break;
    case SDFK_OPT_ELIDE_VOLUME: *value = g_cfg.elide_volume; break;
    case SDFK_OPT_COLOR_PASSES: *value = g_cfg.color_passes; break;
    case SDFK_OPT_PREFAULT_HUGE: *value = g_cfg.prefault_huge; break;
    case SDFK_OPT_HW_QUEUES: *value = g_cfg.hw_queues; break;
    default: return fail(SDFK_ERR_INVALID, "sdfk_get_option: unknown option %d", key);
    }
    return SDFK_OK;
}

// Where compiled code objects are kept (NULL: back to $SDFK_CACHE_DIR | $XDG_CACHE_HOME/sdfkit_hip | ~/.cache/sdfkit_hip as
// the process had them at start-up).  SDFK_OPT_CODE_CACHE = 0 switches the cache off.
extern "C" int sdfk_set_cache_dir(const char* path)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    config_from_env();
    g_cfg.cache_dir_set = path != nullptr;
    g_cfg.cache_dir = path ? path : "";
    return SDFK_OK;
}

// ---------------------------------------------------------------------------
// programs (JIT, counterpart of SdfExprCompiler.Compile, SdfExpr.cs:225-273)
// ---------------------------------------------------------------------------
// ---- on-disk cache of compiled code objects ------------------------------------------------
// hiprtc takes 0.3-1 s per program, once per PROCESS without this: the reference's counterpart
// (SdfExprCompiler.Compile, SdfExpr.cs:234-238) is also a JIT, but what an `sdf.ToMesh()` user sees
// is the first-call latency.  A compiled code object is stored under
//   $SDFK_CACHE_DIR | $XDG_CACHE_HOME/sdfkit_hip | $HOME/.cache/sdfkit_hip | /tmp/sdfkit_hip-<uid>
// as <hash of (flags, hiprtc version, source)>.co = { magic, lengths, the key text itself, code }: a
// hit compares the whole key text, so a hash collision cannot return foreign code.  Files appear
// atomically (write to a temporary, rename).  SDFK_NO_CACHE=1 switches it off.
namespace {
struct JitStats { std::atomic<int64_t> compiled{0}, cache_hits{0}; std::atomic<int64_t> compile_us{0}; } g_jit;   // (process-wide: hiprtc may run for two devices at once)

uint64_t fnv1a64(const std::string& s, uint64_t h)
{
    for (unsigned char c : s) { h ^= c; h *= 0x100000001b3ull; }
    return h;
}

// The directory must be OURS: created here with mode 0700, or an existing directory owned by this user that nobody else
// can write to -- another local user who pre-creates /tmp/sdfkit_hip-<uid> (the fallback when HOME is unset) could plant
// code objects otherwise.  Anything else: no cache.
std::string cache_dir()
{
    if (!g_cfg.code_cache) return std::string();
    std::string d;
    if (g_cfg.cache_dir_set) d = g_cfg.cache_dir;
    else if (!g_cfg.env_cache_dir.empty()) d = g_cfg.env_cache_dir;
    else if (!g_cfg.env_xdg.empty()) d = g_cfg.env_xdg + "/sdfkit_hip";
    else if (!g_cfg.env_home.empty()) d = g_cfg.env_home + "/.cache/sdfkit_hip";
    else d = "/tmp/sdfkit_hip-" + std::to_string((long)getuid());
    if (d.empty()) return d;
    // mkdir -p
    for (size_t i = 1; i <= d.size(); i++)
        if (i == d.size() || d[i] == '/') {
            const std::string sub = d.substr(0, i);
            if (mkdir(sub.c_str(), 0700) != 0 && errno != EEXIST) return std::string();
        }
    struct stat st;
    if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode) || st.st_uid != getuid() || (st.st_mode & (S_IWGRP | S_IWOTH)))
        return std::string();
    return d;
}

constexpr uint64_t kCacheMagic = 0x31304f434b464453ull;   // "SDFKCO01"

bool cache_load(const std::string& path, const std::string& key, std::vector<char>& code)
{
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return false;
    FILE* f = fdopen(fd, "rb");
    if (!f) { close(fd); return false; }
    uint64_t hdr[3] = {0, 0, 0};
    bool ok = fread(hdr, sizeof hdr, 1, f) == 1 && hdr[0] == kCacheMagic && hdr[1] == key.size() && hdr[2] > 0 && hdr[2] < (1ull << 31);
    if (ok) {
        std::string k(key.size(), '\0');
        ok = fread(&k[0], 1, k.size(), f) == k.size() && k == key;
    }
    if (ok) {
        code.resize(hdr[2]);
        ok = fread(code.data(), 1, code.size(), f) == code.size() && fgetc(f) == EOF;
    }
    fclose(f);
    if (!ok) code.clear();
    return ok;
}

void cache_store(const std::string& path, const std::string& key, const std::vector<char>& code)
{
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return;
    FILE* f = fdopen(fd, "wb");
    if (!f) { close(fd); (void)remove(tmp.c_str()); return; }
    const uint64_t hdr[3] = {kCacheMagic, key.size(), code.size()};
    bool ok = fwrite(hdr, sizeof hdr, 1, f) == 1 && fwrite(key.data(), 1, key.size(), f) == key.size() &&
              fwrite(code.data(), 1, code.size(), f) == code.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) (void)remove(tmp.c_str());
}
}  // namespace

// hiprtc (or the on-disk cache) for the kernels `mask` of a generated source
static int compile_source(const std::string& src, unsigned mask, std::vector<char>& code, bool use_cache, bool* from_cache = nullptr,
                          bool refresh = false)
{
    if (from_cache) *from_cache = false;
    const std::string dr = "-DSDFK_SAMPLE_RPW=" + std::to_string(kSampleRpw);
    const std::string dk = "-DSDFK_KERNELS=" + std::to_string(mask);
    std::vector<const char*> opts = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-DSDFK_SAMPLE_NT=1", dr.c_str(), dk.c_str()};
    std::vector<std::string> extra;   // SDFK_JIT_FLAGS at start-up: extra hiprtc options, space separated (experiments)
    {
        std::string t;
        for (const char* q = g_cfg.jit_flags.c_str();; q++) {
            if (*q == ' ' || *q == 0) { if (!t.empty()) extra.push_back(t); t.clear(); if (!*q) break; }
            else t += *q;
        }
        for (auto& e : extra) opts.push_back(e.c_str());
    }
    // cache key: everything the code object depends on
    std::string key, path;
    if (use_cache) {
        int vmaj = 0, vmin = 0;
        (void)hiprtcVersion(&vmaj, &vmin);
        key = "sdfkit_hip abi " + std::to_string(SDFK_ABI_VERSION) + " hiprtc " + std::to_string(vmaj) + "." + std::to_string(vmin) + " opts";
        for (const char* o : opts) { key += ' '; key += o; }
        key += '\n';
        key += src;
        const std::string dir = cache_dir();
        if (!dir.empty()) {
            char name[64];
            snprintf(name, sizeof name, "/%016llx%016llx.co", (unsigned long long)fnv1a64(key, 0xcbf29ce484222325ull),
                     (unsigned long long)fnv1a64(key, 0x84222325cbf29ce4ull));
            path = dir + name;
            if (!refresh && cache_load(path, key, code)) {
                g_jit.cache_hits++;
                if (from_cache) *from_cache = true;
                return SDFK_OK;
            }
        }
    }
    const auto t0 = std::chrono::steady_clock::now();
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src.c_str(), "sdfk_sample.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
        return fail(SDFK_ERR_COMPILE, "hiprtcCreateProgram failed");
    hiprtcResult rc = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
    if (rc != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        hiprtcDestroyProgram(&prog);
        return fail(SDFK_ERR_COMPILE, "hiprtc: %s\n%s", hiprtcGetErrorString(rc), log.c_str());
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    code.resize(cs);
    hiprtcGetCode(prog, code.data());
    hiprtcDestroyProgram(&prog);
    g_jit.compiled++;
    g_jit.compile_us += (int64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (!path.empty()) cache_store(path, key, code);
    return SDFK_OK;
}

extern "C" int sdfk_jit_stats(int64_t* n_compiled, int64_t* n_cache_hits, double* compile_ms_total)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (n_compiled) *n_compiled = g_jit.compiled;
    if (n_cache_hits) *n_cache_hits = g_jit.cache_hits;
    if (compile_ms_total) *compile_ms_total = (double)g_jit.compile_us.load() * 1e-3;
    return SDFK_OK;
}

namespace {
// structures with loaded modules, keyed by the generated source (the whole text is the key: no collisions)

void code_unload(ProgCode* c)
{
    if (!c->modules.empty() && g.inited) sync_all_lanes();   // kernels of these modules may still be queued
    for (hipModule_t m : c->modules) (void)hipModuleUnload(m);
    delete c;
}

ProgCode* code_acquire(std::string&& src)
{
    ProgCode*& slot = g_codes[src];
    if (!slot) {
        slot = new ProgCode();
        slot->source = std::move(src);
    }
    slot->refs++;
    slot->last_use = ++g_code_clock;
    return slot;
}

// structures no program uses at the moment stay loaded, up to SDFK_OPT_IDLE_PROGRAMS of them: the one used longest ago goes first
void codes_trim()
{
    for (;;) {
        size_t idle = 0;
        ProgCode* oldest = nullptr;
        for (auto& kv : g_codes)
            if (kv.second->refs == 0) {
                idle++;
                if (!oldest || kv.second->last_use < oldest->last_use) oldest = kv.second;
            }
        if (idle <= (size_t)g_cfg.idle_programs || !oldest) return;
        g_codes.erase(oldest->source);
        code_unload(oldest);
    }
}

void codes_drop_idle()
{
    for (auto it = g_codes.begin(); it != g_codes.end();) {
        if (it->second->refs == 0) { ProgCode* c = it->second; it = g_codes.erase(it); code_unload(c); }
        else ++it;
    }
}

void code_release(ProgCode* c)
{
    if (!c || --c->refs > 0) return;
    c->last_use = ++g_code_clock;
    codes_trim();
}
}  // namespace

static int generate_source(const sdfk_op* ops, int32_t n_ops, const int32_t out_rgbw[4], int32_t writes_color, std::string& src,
                           std::vector<float>* params = nullptr)
{
    std::string err;
    if (!generate_sample_source(ops, n_ops, out_rgbw, writes_color, src, err, params))
        return fail(SDFK_ERR_INVALID, "SDF program: %s", err.c_str());
    if (!g_cfg.dump_source.empty()) {   // debugging aid (SDFK_DUMP_SOURCE at start-up): the generated HIP source of the last program
        if (FILE* f = fopen(g_cfg.dump_source.c_str(), "w")) { fputs(src.c_str(), f); fclose(f); }
    }
    return SDFK_OK;
}

extern "C" int sdfk_program_check(const sdfk_op* ops, int32_t n_ops, const int32_t out_rgbw[4], int32_t writes_color)
{
    if (!ops || !out_rgbw || n_ops <= 0) return fail(SDFK_ERR_INVALID, "sdfk_program_check: null/empty argument");
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    std::string src;
    std::vector<char> code;
    if (int r = generate_source(ops, n_ops, out_rgbw, writes_color, src)) return r;
    config_from_env();
    return compile_source(src, (1u << PK_COUNT) - 1u, code, false);   // every kernel, a real compile: this IS the check
}

extern "C" int sdfk_program_create(const sdfk_op* ops, int32_t n_ops, const int32_t out_rgbw[4],
                                   int32_t writes_color, sdfk_program** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!out || !ops || !out_rgbw || n_ops <= 0) return fail(SDFK_ERR_INVALID, "sdfk_program_create: null/empty argument");
    *out = nullptr;
    if (int r = require_init()) return r;
    sdfk_program* p = new sdfk_program();
    std::string src;
    if (int r = generate_source(ops, n_ops, out_rgbw, writes_color, src, &p->params)) { delete p; return r; }   // validates the op list
    p->writes_color = writes_color;
    p->n_ops = n_ops;
    p->code = code_acquire(std::move(src));
    *out = p;
    return SDFK_OK;
}

namespace {
// The device function of kernel `k` of a program, compiled and loaded on first use.  A sampler instantiation brings
// sdfk_corners_eval along (same module): marching cubes on the volume it sampled re-evaluates cell corners with it.
int program_fn(const sdfk_program* cp, int k, hipFunction_t* fn)
{
    ProgCode* p = cp->code;
    p->last_use = ++g_code_clock;
    if (!p->fn[k]) {
        static const char* const names[PK_COUNT] = {"sdfk_sample_bits", "sdfk_sample_bits_flat", "sdfk_sample_signs", "sdfk_sample_bits_clip",
                                                    "sdfk_sample_bits_clip_flat", "sdfk_vertex_colors", "sdfk_corners_eval", "sdfk_raymarch",
                                                    "sdfk_sample_signs_flat", "sdfk_cull_blocks", "sdfk_eval_blocks", "sdfk_eval_points",
                                                    "sdfk_sample_bits_nc", "sdfk_sample_bits_nc_flat", "sdfk_sample_bits_nc_clip", "sdfk_sample_bits_nc_clip_flat",
                                                    "sdfk_sample_colors"};
        unsigned mask = 1u << k;
        if (k >= PK_BITS_NC && k <= PK_BITS_NC_CLIP_FLAT && !p->fn[PK_COLORS]) mask |= 1u << PK_COLORS;   // (the second pass: same module)
        if (k == PK_CULL || k == PK_EVAL_BLOCKS) mask |= (1u << PK_CULL) | (1u << PK_EVAL_BLOCKS);   // (a pair)
        if (pk_is_sampler(k) && !p->fn[PK_CORNERS]) {   // (and, for a program that writes colours, sdfk_vertex_colors)
            mask |= 1u << PK_CORNERS;
            if (cp->writes_color && !p->fn[PK_VCOLORS]) mask |= 1u << PK_VCOLORS;
        }
        std::vector<char> code;
        bool cached = false;
        if (int r = compile_source(p->source, mask, code, true, &cached)) return r;
        hipModule_t mod = nullptr;
        hipError_t e = hipModuleLoadData(&mod, code.data());
        if (e != hipSuccess && cached) {   // a damaged cache entry: compile again and replace it
            if (int r = compile_source(p->source, mask, code, true, nullptr, true)) return r;
            e = hipModuleLoadData(&mod, code.data());
        }
        if (e != hipSuccess) return fail(SDFK_ERR_HIP, "loading JIT module: %s", hipGetErrorString(e));
        p->modules.push_back(mod);
        for (int q = 0; q < PK_COUNT; q++)
            if ((mask >> q) & 1u) {
                e = hipModuleGetFunction(&p->fn[q], mod, names[q]);
                if (e != hipSuccess) { p->fn[q] = nullptr; return fail(SDFK_ERR_HIP, "JIT module lacks %s: %s", names[q], hipGetErrorString(e)); }
            }
    }
    *fn = p->fn[k];
    return SDFK_OK;
}
}  // namespace

extern "C" const char* sdfk_program_source(const sdfk_program* p) { return p && p->code ? p->code->source.c_str() : ""; }

extern "C" void sdfk_program_destroy(sdfk_program* p)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!p) return;
    // captured jobs of this program can never be asked for again (the key holds the handle): the free ones go now -- and
    // with them their references, volumes and modules --, a borrowed one when its mesh handle is freed (graph_job_retire)
    p->orphaned = true;
    graph_jobs_forget_program(p);
    program_release(p);
}

namespace {
void program_release(sdfk_program* p)
{
    if (!p || --p->refs > 0) return;
    code_release(p->code);   // (the modules stay loaded for the next program of this structure)
    delete p;
}

// `values` no longer are what a program computed / what the cached sign bits describe
void volume_values_changed(sdfk_volume* v)
{
    v->bits_valid = false;
    if (v->sampled_by) program_release(v->sampled_by);
    v->sampled_by = nullptr;
}
}

// ---------------------------------------------------------------------------
// volumes
// ---------------------------------------------------------------------------
extern "C" int sdfk_volume_create_slab(int32_t nx, int32_t ny, int32_t nz_global, const float min[3],
                                       const float max[3], int32_t z0, int32_t nz_local,
                                       int32_t with_colors, sdfk_volume** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!out || !min || !max) return fail(SDFK_ERR_INVALID, "sdfk_volume_create: null argument");
    *out = nullptr;
    if (nx < 1 || ny < 1 || nz_global < 1 || nz_local < 1 || z0 < 0 || z0 + nz_local > nz_global)
        return fail(SDFK_ERR_INVALID, "sdfk_volume_create: bad dimensions %dx%dx%d (slab z0=%d nz=%d)", nx, ny, nz_global, z0, nz_local);
    if ((int64_t)nx * ny * nz_global >= (int64_t(1) << 31))
        return fail(SDFK_ERR_INVALID, "grid exceeds the reference's int32 linear index (Voxels.cs:82)");
    if ((int64_t)nx * ny * ((nz_local + 3) & ~3) >= (int64_t(1) << 31))   // (the kernels index the PITCHED rows with 32-bit plane offsets)
        return fail(SDFK_ERR_INVALID, "grid with rows padded to %d voxels exceeds 2^31 voxel slots", (nz_local + 3) & ~3);
    if (int r = require_init()) return r;
    sdfk_volume* v = new sdfk_volume();
    v->nx = nx; v->ny = ny; v->nz = nz_local; v->nz_global = nz_global; v->z0 = z0;
    memcpy(v->gmin, min, sizeof v->gmin);
    memcpy(v->gmax, max, sizeof v->gmax);
    int r = dev_alloc((void**)&v->values, v->nalloc() * sizeof(float));
    if (!r && with_colors) r = dev_alloc((void**)&v->colors, v->nalloc() * 3 * sizeof(float));
    if (r) { dev_free(v->values); delete v; return r; }
    *out = v;
    return SDFK_OK;
}

extern "C" int sdfk_volume_create(int32_t nx, int32_t ny, int32_t nz, const float min[3], const float max[3],
                                  int32_t with_colors, sdfk_volume** out)
{
    return sdfk_volume_create_slab(nx, ny, nz, min, max, 0, nz, with_colors, out);
}

namespace {
// The temporary volume of a self-contained sample -> mesh job.  With SDFK_OPT_ELIDE_VOLUME (and both re-evaluation paths on) it
// has no Values / Colors storage at all: nobody can ask for them (the volume never leaves the library).
int job_volume_create(const sdfk_program* p, int nx, int ny, int nz, const float mn[3], const float mx[3], float iso, sdfk_volume** out)
{
    if (!out || !mn || !mx) return fail(SDFK_ERR_INVALID, "sdfk_sample_march: null argument");   // (as sdfk_volume_create_slab answers on the stored path)
    // Not elided: a NaN iso value (it never compares equal, so the cached sign bits never match and the sign-bit pass of the
    // meshing job would have to READ the voxels: the stored path returns its empty mesh) and the sampler-only measurement mode
    // (sdfk_profile_enable(2) leaves no valid sign bits behind) -- the option must never change a status code.
    const bool elide = g_cfg.elide_volume && g_cfg.corner_eval && g_cfg.vcolor_eval && !p->no_elide && iso == iso && !g.sampler_only;
    if (!elide) return sdfk_volume_create(nx, ny, nz, mn, mx, p->writes_color ? 1 : 0, out);
    *out = nullptr;
    if (nx < 1 || ny < 1 || nz < 1) return fail(SDFK_ERR_INVALID, "sdfk_sample_march: bad dimensions %dx%dx%d", nx, ny, nz);
    if ((int64_t)nx * ny * ((nz + 3) & ~3) >= (int64_t(1) << 31)) return fail(SDFK_ERR_INVALID, "grid exceeds the reference's int32 linear index (Voxels.cs:82)");
    sdfk_volume* v = new sdfk_volume();
    v->nx = nx; v->ny = ny; v->nz = nz; v->nz_global = nz; v->z0 = 0;
    memcpy(v->gmin, mn, sizeof v->gmin);
    memcpy(v->gmax, mx, sizeof v->gmax);
    v->elided = true;
    v->elided_colors = p->writes_color != 0;
    *out = v;
    return SDFK_OK;
}

// An elided volume gets its storage after all (the rare fall-back: a volume with case-13 sign words needs the dead-cell test of
// k_resolve, which reads neighbouring voxels) and is sampled again, this time with stores.
int sample_impl(const sdfk_program* p, sdfk_volume* v, int32_t clip_to_bounds, float iso_hint);
int volume_materialize(sdfk_volume* v)
{
    if (!v->elided) return SDFK_OK;
    sdfk_program* p = v->sampled_by;
    if (!p) return fail(SDFK_ERR_INVALID, "an elided volume without its program");
    int r = dev_alloc((void**)&v->values, v->nalloc() * sizeof(float));
    if (!r && v->elided_colors) r = dev_alloc((void**)&v->colors, v->nalloc() * 3 * sizeof(float));
    if (r) { dev_free(v->values); v->values = nullptr; return r; }
    v->elided = false;
    p->no_elide = true;
    const int clip = v->sampled_args.clip;
    const float iso = v->bits_iso;
    p->refs++;                       // (sample_impl drops the volume's reference before it takes a new one)
    r = sample_impl(p, v, clip, iso);
    program_release(p);
    return r;
}
}

extern "C" void sdfk_volume_free(sdfk_volume* v)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!v) return;
    graph_jobs_forget_volume(v);   // (captured slab steps write into it)
    resolve_dependents(v);
    volume_values_changed(v);   // (drops the reference to the program that sampled it)
    // no sync: the pool is stream-ordered (every kernel and copy runs on g.stream, so a block
    // handed out again is only touched by work queued after its previous user)
    dev_free(v->values);
    dev_free(v->colors);
    dev_free(v->bits);
    dev_free(v->bits8);
    dev_free(v->cull_list);
    if (v->cull_header) {   // (zero again, in its lane's order: the next volume-less job of that lane takes it as it is)
        if (v->cull_header_clean) g.lanes[v->cull_header_lane].clean_cull_headers.push_back(v->cull_header);
        else dev_free(v->cull_header);
    }
    delete v;
}

extern "C" int sdfk_volume_upload(sdfk_volume* v, const float* values, const float* colors3)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!v || !values) return fail(SDFK_ERR_INVALID, "sdfk_volume_upload: null argument");
    if (int r = require_init()) return r;
    if (colors3 && !v->colors) return fail(SDFK_ERR_INVALID, "sdfk_volume_upload: volume was created without colours");
    resolve_dependents(v);
    volume_values_changed(v);
    if (v->pitch() == v->nz) {
        HIPCHK(hipMemcpyAsync(v->values, values, v->nvox() * sizeof(float), hipMemcpyHostToDevice, g.stream));
        if (colors3) HIPCHK(hipMemcpyAsync(v->colors, colors3, v->nvox() * 3 * sizeof(float), hipMemcpyHostToDevice, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));  // the caller's arrays are not retained
        return SDFK_OK;
    }
    // rows of nz % 4 != 0 voxels: the dense host layout goes to a temporary device array, a kernel spreads the rows out
    const size_t rows = (size_t)v->nx * v->ny;
    float* tmp = nullptr;
    if (int r = dev_alloc((void**)&tmp, v->nvox() * (colors3 ? 3 : 1) * sizeof(float))) return r;
    hipError_t e = hipMemcpyAsync(tmp, values, v->nvox() * sizeof(float), hipMemcpyHostToDevice, g.stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_repitch<true>, dim3(grid_for(v->nvox())), dim3(256), 0, g.stream, tmp, v->values, rows, v->nz, v->pitch());
        if (colors3) {
            e = hipMemcpyAsync(tmp, colors3, v->nvox() * 3 * sizeof(float), hipMemcpyHostToDevice, g.stream);
            if (e == hipSuccess)
                hipLaunchKernelGGL(k_repitch<true>, dim3(grid_for(v->nvox() * 3)), dim3(256), 0, g.stream, tmp, v->colors, rows, v->nz * 3, v->pitch() * 3);
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g.stream);   // the caller's arrays are not retained
    dev_free(tmp);
    if (e != hipSuccess) return fail(SDFK_ERR_HIP, "sdfk_volume_upload: %s", hipGetErrorString(e));
    return SDFK_OK;
}

extern "C" int sdfk_volume_download(const sdfk_volume* v, float* values, float* colors3)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!v) return fail(SDFK_ERR_INVALID, "sdfk_volume_download: null volume");
    if (int r = require_init()) return r;
    std::vector<CopyPiece> pieces;
    float* dense_v = nullptr;   // rows of nz % 4 != 0 voxels: packed into dense temporaries first
    float* dense_c = nullptr;
    if (v->pitch() != v->nz) {
        const size_t rows = (size_t)v->nx * v->ny;
        if (values) {
            if (int r = dev_alloc((void**)&dense_v, v->nvox() * sizeof(float))) return r;
            hipLaunchKernelGGL(k_repitch<false>, dim3(grid_for(v->nvox())), dim3(256), 0, g.stream, v->values, dense_v, rows, v->nz, v->pitch());
        }
        if (colors3 && v->colors) {
            if (int r = dev_alloc((void**)&dense_c, v->nvox() * 3 * sizeof(float))) { dev_free(dense_v); return r; }
            hipLaunchKernelGGL(k_repitch<false>, dim3(grid_for(v->nvox() * 3)), dim3(256), 0, g.stream, v->colors, dense_c, rows, v->nz * 3, v->pitch() * 3);
        }
    }
    struct FreeTmp { float *a, *b; ~FreeTmp() { dev_free(a); dev_free(b); } } free_tmp{dense_v, dense_c};   // (stream-ordered pool; the copies below are synchronous)
    if (values) pieces.push_back({dense_v ? dense_v : v->values, values, v->nvox() * sizeof(float)});
    if (colors3 && v->colors) pieces.push_back({dense_c ? dense_c : v->colors, colors3, v->nvox() * 3 * sizeof(float)});
    if (colors3 && !v->colors) {   // colours that were never written are zero (Voxels.cs:88-92): cleared on the pool
        std::lock_guard<std::recursive_mutex> pool_lk(g_pool_mu);
        const size_t nb = v->nvox() * 3 * sizeof(float), per = size_t(2) << 20;
        char* c = (char*)colors3;
        g_pool.start((int)((nb + per - 1) / per), [=](int t) { const size_t a = (size_t)t * per; memset(c + a, 0, std::min(per, nb - a)); });
        g_pool.wait();
    }
    return copy_to_host(pieces);
}

extern "C" int sdfk_volume_row_pitch(const sdfk_volume* v, int32_t* pitch_voxels)
{
    if (!v || !pitch_voxels) return fail(SDFK_ERR_INVALID, "sdfk_volume_row_pitch: null argument");
    *pitch_voxels = v->pitch();
    return SDFK_OK;
}

extern "C" int sdfk_volume_device_ptrs(const sdfk_volume* v, void** values, void** colors3)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!v) return fail(SDFK_ERR_INVALID, "null volume");
    resolve_dependents(v);   // the caller may write through these pointers: no cached view of
    volume_values_changed(const_cast<sdfk_volume*>(v));   // the values stays valid
    if (values) *values = v->values;
    if (colors3) *colors3 = v->colors;
    return SDFK_OK;
}

// Grid dimensions y and z hold at most 65535 workgroups; extents that may exceed that go to x, or are split over z and y.
static dim3 transpose_grid(int nz, int ny, int nxw)
{
    const unsigned groups = (unsigned)((nxw + 7) / 8);
    return dim3((unsigned)((nz + 127) / 128) * (unsigned)ny, (groups + 65534u) / 65535u, std::min(groups, 65535u));
}
static dim3 flat_grid(size_t plane, int nx8)   // plane-chunk kernels: (chunks of the (y, z) plane, x groups beyond 65535, x groups)
{
    return dim3((unsigned)((plane + 255) / 256), ((unsigned)nx8 + 65534u) / 65535u, std::min((unsigned)nx8, 65535u));
}

// Voxels.cs:32-34,81,139: cell size, first cell centre and ClipToBounds value, in float
static void grid_constants(const sdfk_volume* v, float d[3], float m[3], float* outside)
{
    const int n[3] = {v->nx, v->ny, v->nz_global};
    for (int k = 0; k < 3; k++) {
        d[k] = n[k] >= 1 ? (v->gmax[k] - v->gmin[k]) / (float)n[k] : 0.0f;
        const float h = 0.5f * d[k];
        m[k] = v->gmin[k] + h;
    }
    *outside = (v->gmax[0] - v->gmin[0]) / (float)v->nx;
}

namespace {
int sample_impl(const sdfk_program* p, sdfk_volume* v, int32_t clip_to_bounds, float iso_hint)
{
    SampleArgs A;
    memset(&A, 0, sizeof A);
    float d[3], m[3], outside;
    grid_constants(v, d, m, &outside);
    A.values = v->values;
    A.colors = v->colors;
    A.mx = m[0]; A.my = m[1]; A.mz = m[2];
    A.dx = d[0]; A.dy = d[1]; A.dz = d[2];
    A.nx = v->nx; A.ny = v->ny; A.nz = v->nz;
    A.z0 = v->z0; A.nz_global = v->nz_global;
    A.clip = clip_to_bounds ? 1 : 0;
    A.outside = outside;
    resolve_dependents(v);
    volume_values_changed(v);
    auto sampled = [&]() {   // from here on `values` are this program's output
        sdfk_program* q = const_cast<sdfk_program*>(p);
        q->refs++;
        v->sampled_by = q;
        v->sampled_args = A;
    };
    void* params[] = {&A, p->kargs()};
    {
        // fused sampling + sign bits (iso known or guessed 0): marching cubes then skips its
        // dense pass over the volume
        if (!v->bits) {
            if (int r = dev_alloc((void**)&v->bits, v->nbitwords() * sizeof(uint64_t))) return r;
        }
        if (!v->bits8 && !(v->elided && g_cfg.elide_volume >= 2)) {   // (block culling writes the words directly: no byte form)
            if (int r = dev_alloc((void**)&v->bits8, (size_t)v->ny * v->nx8() * v->pitch8() + 64)) return r;
        }
        A.bits8 = v->bits8;
        A.nx8 = v->nx8();
        A.pitch8 = v->pitch8();
        A.iso = iso_hint;
        {
            const unsigned tpb = 512u / (unsigned)kSampleRpw;
            // 0: z tiles of one y row (nz % 256 == 0); 1: 256-voxel chunks of the (y, z) plane of an x row (any nz:
            // rows are padded to a multiple of 4 voxels)
            const int force = g_cfg.sample_mode;   // (debugging: SDFK_SAMPLE_MODE at start-up)
            int mode = (v->nz % 256) == 0 ? 0 : 1;
            if (force == 0 && (v->nz & 3) == 0) mode = 0;
            if (force == 1 || v->ny > 65535 || v->nx8() > 65535) mode = 1;   // (the row-tiled form has y and x/8 in 16-bit grid dimensions)
            static const char* const names[2][2] = {{"sdfk_sample_bits", "sdfk_sample_bits_flat"}, {"sdfk_sample_bits_clip", "sdfk_sample_bits_clip_flat"}};
            hipFunction_t fn = nullptr, fn_colors = nullptr;   // (compiled on first use)
            // Two passes for a colour volume (sample_codegen.h, "two-pass sampling"): values + sign bytes with this tile's kernel, then the
            // colour array as ONE linear stream.  Worth it when the program is cheap enough to evaluate twice and the grid is large enough
            // for the store rate to matter; SDFK_OPT_COLOR_PASSES = 1 / 2 force one / two passes at any size.
            const bool two_pass = !v->elided && p->writes_color && v->colors &&
                                  (g_cfg.color_passes == 2 || (g_cfg.color_passes == 0 && p->n_ops <= kTwoPassMaxOps && v->nvox() >= kTwoPassMinVoxels));
            const int pk = v->elided ? (mode ? PK_SIGNS_FLAT : PK_SIGNS)
                         : two_pass ? (clip_to_bounds ? PK_BITS_NC_CLIP : PK_BITS_NC) + mode
                                    : (clip_to_bounds ? PK_BITS_CLIP : PK_BITS) + mode;
            if (two_pass)
                if (int r = program_fn(p, PK_COLORS, &fn_colors)) return r;
            if (!(v->elided && g_cfg.elide_volume >= 2))
                if (int r = program_fn(p, pk, &fn)) return r;
            const bool cull = v->elided && g_cfg.elide_volume >= 2;
            // (the name rocprofv3 shows for the entry point launched; the two culling kernels have scopes of their own)
            static const char* const names_nc[2][2] = {{"sdfk_sample_bits_nc", "sdfk_sample_bits_nc_flat"}, {"sdfk_sample_bits_nc_clip", "sdfk_sample_bits_nc_clip_flat"}};
            std::unique_ptr<ProfScope> ps(new ProfScope(cull ? nullptr : (v->elided ? (mode ? "sdfk_sample_signs_flat" : "sdfk_sample_signs")
                                                                                      : (two_pass ? names_nc : names)[clip_to_bounds ? 1 : 0][mode])));
            const size_t plane = (size_t)v->ny * v->pitch();
            phase_token_wait(0);
            if (cull) {
                // block culling: one lane per 64 x 4 x 4 block decides it by interval arithmetic (constant sign words) or lists it; the
                // listed blocks -- those the surface passes through -- are evaluated voxel by voxel; both write the X-packed sign
                // words themselves (sample_codegen.h): no byte form, no transposer
                struct { unsigned long long* bits; unsigned* worklist; unsigned* counter; int nbx, nby, nbz; int cpw; unsigned region; } Cargs;   // (= CullArgs of sample_codegen.h)
                Cargs.bits = (unsigned long long*)v->bits;
                Cargs.nbx = v->nxw(); Cargs.nby = (v->ny + 3) / 4; Cargs.nbz = (v->nz + 3) / 4;
                const size_t nblocks = (size_t)Cargs.nbx * Cargs.nby * Cargs.nbz;
                // 64 sub-lists (SDFK_CULL_LISTS), their counters 128 bytes apart in front: workgroup w appends to sub-list w % 64, whose region
                // holds what its share of the workgroups can list (at most 1024 blocks each)
                constexpr size_t kLists = 64, kHeader = kLists * 32;

                // coarse boxes of 2 x 2 x 2 blocks, cpw of them per wavefront (its one coarse evaluation is the overhead when every box needs
                // the closer look: 1 / cpw): as many as leave >= 4096 wavefronts to the launch, 4 at most.  Measured (sphere / README scene,
                // us) at 512^3, 16 384 coarse boxes: cpw 1: 16.7 / 25.4, 2: 14.1 / 21.0, 4: 13.4 / 18.9, 16: 19.3 / 35.0, 32: 27.2 / 57.6 (few, long
                // wavefronts), without the coarse pass 15.1 / 23.3 at cpw 1; at 1024^3: cpw 4: 44.8 / 59.0, 8: 57.6 / 79.3, 16: 59.1 / 77.4,
                // without the coarse pass 89.3 / 119.1 at cpw 8
                const size_t ncoarse = (size_t)((Cargs.nbx + 1) / 2) * ((Cargs.nby + 1) / 2) * ((Cargs.nbz + 1) / 2);
                static const int cpw_env = [] { const char* e = getenv("SDFK_CULL_CPW"); return e ? atoi(e) : 0; }();   // (experiments)
                Cargs.cpw = cpw_env > 0 ? std::min(cpw_env, 4) : (int)std::min<size_t>(4, std::max<size_t>(1, ncoarse / 4096));   // (<= 4: the kernel's list)
                static const bool coarse_off = [] { const char* e = getenv("SDFK_CULL_COARSE"); return e && atoi(e) == 0; }();
                if (coarse_off) Cargs.cpw = -Cargs.cpw;
                const size_t cull_wgs = (ncoarse + (size_t)std::abs(Cargs.cpw) * 4 - 1) / ((size_t)std::abs(Cargs.cpw) * 4);
                Cargs.region = (unsigned)(((cull_wgs + kLists - 1) / kLists) * (size_t)std::abs(Cargs.cpw) * 32);   // (32 blocks per coarse box of a workgroup's four wavefronts)
                if (!v->cull_list) {   // (a volume's dimensions never change: neither does the size of its regions)
                    // (the blocks' sub-box masks, a byte each, behind the regions)
                    if (int r = dev_alloc((void**)&v->cull_list, kLists * (size_t)Cargs.region * sizeof(uint32_t) + kLists * (size_t)Cargs.region + 64)) return r;
                }
                if (!v->cull_header) {   // the counters: a block this lane knows to be zero, or a new one, cleared once
                    auto& clean = g.lanes[g.cur_lane].clean_cull_headers;
                    if (!clean.empty()) { v->cull_header = clean.back(); clean.pop_back(); }
                    else {
                        if (int r = dev_alloc((void**)&v->cull_header, kHeader * sizeof(uint32_t))) return r;
                        HIPCHK(hipMemsetAsync(v->cull_header, 0, kHeader * sizeof(uint32_t), g.stream));
                    }
                    v->cull_header_lane = g.cur_lane;
                } else if (!v->cull_header_clean || v->cull_header_lane != g.cur_lane) {
                    HIPCHK(hipMemsetAsync(v->cull_header, 0, kHeader * sizeof(uint32_t), g.stream));
                    v->cull_header_lane = g.cur_lane;
                }
                v->cull_header_clean = false;
                Cargs.counter = v->cull_header; Cargs.worklist = v->cull_list;
                hipFunction_t fn_cull = nullptr, fn_eval = nullptr;
                if (int r = program_fn(p, PK_CULL, &fn_cull)) return r;
                if (int r = program_fn(p, PK_EVAL_BLOCKS, &fn_eval)) return r;
                void* cparams[] = {&A, &Cargs, p->kargs()};
                {
                    ProfScope ps2("sdfk_cull_blocks");
                    HIPCHK(hipModuleLaunchKernel(fn_cull, (unsigned)cull_wgs, 1, 1, 256, 1, 1, 0, g.stream, cparams, nullptr));   // (a wavefront per cpw coarse boxes)
                }
                {
                    ProfScope ps2("sdfk_eval_blocks");
                    HIPCHK(hipModuleLaunchKernel(fn_eval, (unsigned)std::min<size_t>((nblocks + 3) / 4, 4096), 1, 1, 256, 1, 1, 0, g.stream, cparams, nullptr));
                }
            } else if (mode == 1) {
                const dim3 fg = flat_grid(plane, v->nx8());
                HIPCHK(hipModuleLaunchKernel(fn, fg.x, fg.y, fg.z, tpb, 1, 1, 0, g.stream, params, nullptr));
            }
            else
                HIPCHK(hipModuleLaunchKernel(fn, (unsigned)((v->nz + 255) / 256), (unsigned)v->ny,
                                             (unsigned)v->nx8(), tpb, 1, 1, 0, g.stream, params, nullptr));
            ps.reset();
            if (two_pass) {   // the colours: 256 consecutive voxels of the padded volume per workgroup, three contiguous KiB each
                ProfScope ps2("sdfk_sample_colors");
                const size_t total = (size_t)v->nx * plane;
                HIPCHK(hipModuleLaunchKernel(fn_colors, (unsigned)((total + 255) / 256), 1, 1, 256, 1, 1, 0, g.stream, params, nullptr));
            }
        }
        phase_token_pass(0);
        if (g.sampler_only) {   // measurement mode (sdfk_profile_enable(2)): the sampling kernel alone, back to back
            v->bits_valid = false;
            return SDFK_OK;
        }
        if (!(v->elided && g_cfg.elide_volume >= 2)) {   // (the block-culling kernels write the words themselves)
            ProfScope ps("k_bits_transpose");
            hipLaunchKernelGGL(k_bits_transpose, transpose_grid(v->nz, v->ny, v->nxw()), dim3(256), 0, g.stream,
                               v->bits8, v->bits, v->nx8(), v->ny, v->nz, v->nxw(), v->pitch8());
            HIPCHK(hipGetLastError());
        }
        v->bits_iso = iso_hint;
        v->bits_valid = true;
        sampled();
        return SDFK_OK;
    }
}
}  // namespace

extern "C" int sdfk_sample(const sdfk_program* p, sdfk_volume* v, int32_t clip_to_bounds)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || !v) return fail(SDFK_ERR_INVALID, "sdfk_sample: null argument");
    if (int r = require_init()) return r;
    return sample_impl(p, v, clip_to_bounds, 0.0f);
}

extern "C" int sdfk_volume_clip_to_bounds(sdfk_volume* v)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!v) return fail(SDFK_ERR_INVALID, "null volume");
    if (int r = require_init()) return r;
    float d[3], m[3], outside;
    grid_constants(v, d, m, &outside);
    resolve_dependents(v);
    // The clip rule is known: the cached views of the values are patched instead of dropped.
    // Sign bits: the six faces become (outside > iso); the sampling program that produced the
    // values (if any) is remembered with clip = 1, so re-evaluated cell corners see the same faces.
    const size_t face = std::max({(size_t)v->ny * v->nz, (size_t)v->nx * v->nz, (size_t)v->nx * v->ny});
    ProfScope ps("k_clip");
    hipLaunchKernelGGL(k_clip, dim3((unsigned)((face + 255) / 256)), dim3(256), 0, g.stream, v->values, v->nx, v->ny, v->nz, v->pitch(), v->z0,
                       v->nz_global, outside);
    if (v->bits && v->bits_valid)
        hipLaunchKernelGGL(k_clip_bits, dim3((unsigned)(((size_t)v->nz * v->ny + 255) / 256)), dim3(256), 0, g.stream, v->bits, v->nx, v->ny,
                           v->nz, v->z0, v->nz_global, v->nxw(), outside > v->bits_iso ? 1 : 0);
    if (v->sampled_by) v->sampled_args.clip = 1;
    HIPCHK(hipGetLastError());
    return SDFK_OK;
}

// ---------------------------------------------------------------------------
// marching cubes driver
// ---------------------------------------------------------------------------
namespace {

template <typename T>
int job_alloc(sdfk_march_job* j, T** p, size_t count)
{
    void* q = nullptr;
    if (int r = dev_alloc(&q, count * sizeof(T))) return r;
    j->owned.push_back(q);
    *p = (T*)q;
    return SDFK_OK;
}

int acquire_slot()
{
    for (int pass = 0; pass < 2; pass++) {
        int oldest = -1;
        for (int i = 0; i < Context::NSLOTS; i++) {
            const int s = (g.slot_next + i) % Context::NSLOTS;
            Context::SlotState& st = g.slot_state[s];
            if (st.busy) continue;
            if (st.drop_pending) {   // the dropped job's kernels may still be queued: has its lane passed them?
                if (hipEventQuery(st.dropped) != hipSuccess) { if (oldest < 0) oldest = s; continue; }
                st.drop_pending = false;
            }
            st.busy = true;
            g.slot_next = (s + 1) % Context::NSLOTS;
            memset(&g.slots[s], 0, sizeof(Context::HostSlot));
            return s;
        }
        if (oldest < 0) break;   // every slot belongs to a live job
        // all free slots still wait for dropped jobs: wait for the one dropped first (slots are
        // handed out round-robin, so the first candidate after slot_next is the oldest) -- one
        // event, not a synchronisation of every lane
        (void)hipEventSynchronize(g.slot_state[oldest].dropped);
    }
    return -1;
}

void job_release(sdfk_march_job* j, bool kernels_may_be_queued)
{
    if (j->slot >= 0) {
        Context::SlotState& st = g.slot_state[j->slot];
        st.busy = false;
        st.drop_pending = false;
        if (kernels_may_be_queued) {
            hipError_t e = hipSuccess;
            if (!st.dropped) e = hipEventCreateWithFlags(&st.dropped, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(st.dropped, lane_stream(j->lane));
            if (e == hipSuccess) st.drop_pending = true;
            else (void)hipStreamSynchronize(lane_stream(j->lane));   // no event: wait here instead
        }
        j->slot = -1;
    }
    for (void* p : j->owned) dev_free(p);
    j->owned.clear();
    if (j->eval_prog) program_release(j->eval_prog);
    j->eval_prog = nullptr;
    if (j->sub) {
        dev_free(j->sub->values);
        dev_free(j->sub->colors);
        delete j->sub;
        j->sub = nullptr;
    }
}

#ifndef SDFK_COMPACT_MASKS
#define SDFK_COMPACT_MASKS 1   // the count pass leaves its activity masks for the write pass (McParams::segmask)
#endif
#ifndef SDFK_COMPACT_STRIDED
#define SDFK_COMPACT_STRIDED 1   // the write pass with interleaved segments (k_compact_write, mc_kernels.hip)
#endif

int alloc_records(sdfk_march_job* j, size_t c)
{
    McParams& P = j->P;
    for (size_t k = j->rec_first; k < j->owned.size(); k++) dev_free(j->owned[k]);
    j->owned.resize(j->rec_first);
    j->bounds_partial = nullptr;   // (allocated after the records: freed with them)
    j->vdesc = nullptr;
    j->vdesc_cap = 0;
    int rr = 0;
    rr = rr ? rr : job_alloc(j, &P.rec_xy, c);
    rr = rr ? rr : job_alloc(j, &P.rec_z, c);
    rr = rr ? rr : job_alloc(j, &P.rec_info, c);
    rr = rr ? rr : job_alloc(j, &P.rec_own, c);
    rr = rr ? rr : job_alloc(j, &P.rec_pre, c);
    rr = rr ? rr : job_alloc(j, &P.rec_corners, c * 8);
    rr = rr ? rr : job_alloc(j, &P.rec_vid, (c / MC_CHUNK + 1) * (size_t)MC_VSTRIDE);
    rr = rr ? rr : job_alloc(j, &P.chunkslots, c / MC_CHUNK + 2);
    rr = rr ? rr : job_alloc(j, &P.chunktot, c / MC_CHUNK + 2);
    rr = rr ? rr : job_alloc(j, &P.chunkpre, c / MC_CHUNK + 2);
    rr = rr ? rr : job_alloc(j, &P.chunkdead, c / MC_CHUNK + 2);
    rr = rr ? rr : job_alloc(j, &P.chunkwin, c / MC_CHUNK + 2);
    rr = rr ? rr : job_alloc(j, &P.chunkwin2, c / MC_CHUNK + 2);
    P.cap_active = (uint32_t)c;
    P.chunkscan = c / MC_CHUNK + 1 > MC_SCAN_CHUNKS ? 1 : 0;   // (k_chunkscan: mc_kernels.hip)
    return rr;
}

// classification: sign bits (unless cached) -> ordered compaction -> corner gather ->
// resolve -> chunk scan.  Launches only; nothing here waits for the GPU.
int launch_classify(sdfk_march_job* j, bool publish)
{
    McParams& P = j->P;
    if (!j->have_bits) {
        uint64_t* bits = const_cast<uint64_t*>(P.bits);
        ProfScope ps("k_signbits");
        const int nx8 = (P.nx + 7) / 8, pitch8 = P.nzp;
        if ((P.nz % 256) != 0 || P.ny > 65535 || nx8 > 65535)   // chunks of the (y, z) plane: rows shorter or longer than a z tile
            hipLaunchKernelGGL(k_signbits8<true>, flat_grid((size_t)P.ny * pitch8, nx8), dim3(256), 0, g.stream,
                               P.values, j->bits8, P.nx, P.ny, P.nz, nx8, pitch8, P.iso);
        else
            hipLaunchKernelGGL(k_signbits8<false>, dim3((P.nz + 255) / 256, P.ny, nx8), dim3(256), 0, g.stream, P.values, j->bits8, P.nx, P.ny,
                               P.nz, nx8, pitch8, P.iso);
        hipLaunchKernelGGL(k_bits_transpose, transpose_grid(P.nz, P.ny, P.nxw), dim3(256), 0, g.stream, j->bits8, bits,
                           nx8, P.ny, P.nz, P.nxw, pitch8);
        HIPCHK(hipGetLastError());
        j->have_bits = true;
    }
    {
        ProfScope ps("k_compact");
        // (a workgroup takes the same 1024 segments of K2_LPB consecutive layers: mc_kernels.hip)
        const int nwg = ((P.lay_list_end - P.lay_count_begin + K2_LPB - 1) / K2_LPB) * P.bpl;
        hipLaunchKernelGGL(k_compact<false>, dim3(nwg), dim3(256), 0, g.stream, P);
        // (the count pass clears the culling kernel's counters: host bookkeeping of a device-side effect, so only once the launch is
        // known to be queued and to have workgroups -- a counter block recorded as clean that is not would make the next volume-less
        // job walk stale work lists)
        HIPCHK(hipGetLastError());
        if (P.zero_cull && j->cull_clean && nwg > 0) *j->cull_clean = true;
        if (P.blockpre) hipLaunchKernelGGL(k_blockscan, dim3(1), dim3(1024), 0, g.stream, P);   // (many blocks: their prefix in one pass)
        if (SDFK_COMPACT_STRIDED && K2_LPB == 1)   // (the write pass with interleaved segments: mc_kernels.hip)
            if (P.segmask) hipLaunchKernelGGL(k_compact_write<true>, dim3((P.lay_list_end - P.lay_count_begin) * P.bpl), dim3(256), 0, g.stream, P);
            else hipLaunchKernelGGL(k_compact_write<false>, dim3((P.lay_list_end - P.lay_count_begin) * P.bpl), dim3(256), 0, g.stream, P);
        else
            hipLaunchKernelGGL(k_compact<true>, dim3(nwg), dim3(256), 0, g.stream, P);
        HIPCHK(hipGetLastError());
    }
    const int nchunks = (int)((P.cap_active + MC_CHUNK - 1u) / MC_CHUNK);
    if (j->eval_prog) {   // the volume still is this program's output: evaluate the corners
        hipFunction_t fn_corners = nullptr;
        if (int r = program_fn(j->eval_prog, PK_CORNERS, &fn_corners)) return r;
        ProfScope ps("sdfk_corners_eval");
        const unsigned* n_active = &P.counters->n_active;
        void* params[] = {&j->eval_args, &P.rec_xy, &P.rec_z, &P.rec_corners, &n_active, &P.cap_active, &P.xbits, j->eval_prog->kargs()};
        HIPCHK(hipModuleLaunchKernel(fn_corners, (unsigned)std::min(nchunks, 256 * 8), 1, 1, 256, 1, 1, 0,
                                     g.stream, params, nullptr));
    } else {
        ProfScope ps("k_gather_corners");
        hipLaunchKernelGGL(k_gather_corners, dim3(std::min(nchunks, 256 * 8)), dim3(256), 0, g.stream, P);
        HIPCHK(hipGetLastError());
    }
    {
        ProfScope ps("k_resolve");
        constexpr int cap = 256 * 12;
        hipLaunchKernelGGL(k_resolve, dim3(std::min(nchunks, cap)), dim3(256), 0, g.stream, P);
        if (P.chunkscan) hipLaunchKernelGGL(k_chunkscan, dim3(1), dim3(1024), 0, g.stream, P);   // (long lists: the chunks' prefix in one pass)
        // totals for the host: workgroup 0 of k_vertices publishes them, unless the caller
        // needs the counts before (or without) emitting
        if (publish) hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, g.stream, P);
        HIPCHK(hipGetLastError());
    }
    return SDFK_OK;
}

// the one host<->device rendezvous of a march: drain the stream, read the mirrored counters
int wait_counters(sdfk_march_job* j)
{
    HIPCHK(hipStreamSynchronize(g.stream));
    j->c = g.slots[j->slot].c;
    return SDFK_OK;
}

// Builds the job for `v` (or for a subsampled copy when step > 1); no kernel except the
// optional subsample is launched here.
int setup_job(const sdfk_volume* v, float iso, int step, int layer_begin, int layer_end, size_t cap_records,
              sdfk_march_job** out)
{
    *out = nullptr;
    if (step < 1) return fail(SDFK_ERR_INVALID, "step must be >= 1");
    const bool slab = (v->z0 != 0 || v->nz != v->nz_global);
    if (step != 1 && (slab || layer_begin != 0)) return fail(SDFK_ERR_UNSUPPORTED, "slab meshing supports step == 1 only");
    sdfk_march_job* j = new sdfk_march_job();
    j->gnx = v->nx; j->gny = v->ny; j->gnz = v->nz_global;
    memcpy(j->gmin, v->gmin, sizeof j->gmin);
    memcpy(j->gmax, v->gmax, sizeof j->gmax);
    j->lane = g.cur_lane;
    j->slot = acquire_slot();
    if (j->slot < 0) { delete j; return fail(SDFK_ERR_NOMEM, "more than %d marching-cubes jobs are alive", Context::NSLOTS); }
    const sdfk_volume* w = v;
    if (step > 1) {
        // MarchingCubes.cs:49-80 touches only voxels at multiples of step
        sdfk_volume* s = new sdfk_volume(*v);
        s->nx = (v->nx - 1) / step + 1; s->ny = (v->ny - 1) / step + 1; s->nz = (v->nz - 1) / step + 1;
        s->nz_global = s->nz;
        s->values = nullptr; s->colors = nullptr; s->bits = nullptr; s->bits8 = nullptr; s->bits_valid = false; s->sampled_by = nullptr;
        j->sub = s;
        int r = dev_alloc((void**)&s->values, s->nalloc() * sizeof(float));
        if (!r && v->colors) r = dev_alloc((void**)&s->colors, s->nalloc() * 3 * sizeof(float));
        if (r) { job_release(j); delete j; return r; }
        ProfScope ps("k_subsample");
        hipLaunchKernelGGL(k_subsample, dim3(grid_for(s->nvox())), dim3(256), 0, g.stream, v->values, v->colors,
                           s->values, s->colors, v->nx, v->ny, v->pitch(), s->nx, s->ny, s->nz, s->pitch(), step);
        w = s;
        layer_end = s->nz - 1;
    }
    McParams& P = j->P;
    memset(&P, 0, sizeof P);
    P.values = w->values; P.colors = w->colors;
    j->colors_elided = w->elided && w->elided_colors;
    // (the counters of the culling kernel that made this volume's sign bits: the count pass clears them for the lane's next job)
    if (w->elided && w->cull_header && w->cull_header_lane == g.cur_lane) { P.zero_cull = w->cull_header; j->cull_clean = &w->cull_header_clean; }
    if (w->elided && (step != 1 || !w->sampled_by || !g_cfg.corner_eval || (w->elided_colors && !g_cfg.vcolor_eval) || !w->bits_valid || w->bits_iso != iso)) {
        job_release(j); delete j;
        return fail(SDFK_ERR_INVALID, "internal: a volume without storage can only be meshed by the program that sampled it (step 1, same iso)");
    }
    P.nx = w->nx; P.ny = w->ny; P.nz = w->nz;
    P.nzp = w->pitch();
    {   // bit split of the packed cell coordinates: 16 + 16 unless one of nx, ny needs more (the other then needs fewer)
        auto bits = [](int n) { int b = 0; while ((1 << b) < n) b++; return b; };
        const int bx = bits(w->nx), by = bits(w->ny);
        P.xbits = by > 16 ? 32 - by : std::max(16, bx);
        P.xmask = P.xbits >= 32 ? 0xffffffffu : ((1u << P.xbits) - 1u);
    }
    P.ncx = w->nx - 1; P.ncy = w->ny - 1; P.ncz = w->nz - 1;
    P.nxw = (w->nx + 63) / 64;
    P.z0 = w->z0;
    P.iso = iso;
    P.step = step;
    const int ncz_global = w->nz_global - 1;
    if (layer_begin < 0 || layer_end > std::max(ncz_global, 0) || layer_begin > layer_end) {
        job_release(j); delete j;
        return fail(SDFK_ERR_INVALID, "layer range [%d,%d) outside [0,%d)", layer_begin, layer_end, ncz_global);
    }
    P.lay_emit_begin = layer_begin - w->z0;
    P.lay_emit_end = layer_end - w->z0;
    P.lay_count_begin = layer_begin > 0 ? P.lay_emit_begin - 1 : P.lay_emit_begin;
    P.lay_list_end = std::min(P.lay_emit_end + 1, P.ncz);   // the layer above feeds seam normals
    j->empty = (P.ncx <= 0 || P.ncy <= 0 || P.ncz <= 0 || layer_begin == layer_end);
    memset(&j->c, 0, sizeof j->c);
    if (j->empty) { *out = j; return SDFK_OK; }
    {   // context planes the slab must hold (see sdfkit_hip.h)
        const int need_lo = std::max(layer_begin - 2, 0), need_hi = std::min(layer_end + 2, w->nz_global);
        if (w->z0 > need_lo || w->z0 + w->nz < need_hi) {
            job_release(j); delete j;
            return fail(SDFK_ERR_INVALID, "slab planes [%d,%d) do not cover the context [%d,%d) of layers [%d,%d)",
                        w->z0, w->z0 + w->nz, need_lo, need_hi, layer_begin, layer_end);
        }
    }
    // logical blocks of k_compact: 1024 consecutive 64-cell segments of one layer
    P.bpl = (int)(((size_t)P.ncy * P.nxw + 1023) / 1024);
    const size_t ncell = (size_t)P.ncx * P.ncy * P.ncz;
    if (cap_records == 0) cap_records = std::max<size_t>(ncell / 12, 1u << 16);
    cap_records = std::min(cap_records, ncell);
    int r = 0;
    if (step == 1 && v->sampled_by && g_cfg.corner_eval) {
        j->eval_prog = v->sampled_by;
        j->eval_prog->refs++;
        j->eval_args = v->sampled_args;
    }
    j->have_bits = (step == 1 && v->bits && v->bits_valid && v->bits_iso == iso);
    if (j->have_bits) P.bits = v->bits;   // written by the fused sampling kernel; owned by the volume
    else {
        uint64_t* bits = nullptr;
        r = r ? r : job_alloc(j, &bits, (size_t)P.nz * P.ny * P.nxw + 8);   // k_compact reads 4 words past a row pair
        P.bits = bits;
        r = r ? r : job_alloc(j, &j->bits8, (size_t)P.ny * ((P.nx + 7) / 8) * ((P.nz + 3) & ~3) + 64);
    }
    r = r ? r : job_alloc(j, &P.blockcnt, (size_t)(P.lay_list_end - P.lay_count_begin) * P.bpl + 1);
    r = r ? r : job_alloc(j, &P.wavecnt, (size_t)(P.lay_list_end - P.lay_count_begin) * P.bpl * 4 + 4);
    P.segmask = nullptr;
    if (SDFK_COMPACT_STRIDED && K2_LPB == 1 && SDFK_COMPACT_MASKS)   // (the write pass reads the count pass's masks: mc_kernels.hip)
        r = r ? r : job_alloc(j, &P.segmask, (size_t)(P.lay_list_end - P.lay_count_begin) * P.bpl * 1024 + 4);
    P.blockpre = nullptr;
    if ((P.lay_list_end - P.lay_count_begin) * P.bpl > MC_SCAN_BLOCKS && SDFK_COMPACT_STRIDED && K2_LPB == 1)   // (k_blockscan: mc_kernels.hip)
        r = r ? r : job_alloc(j, &P.blockpre, (size_t)(P.lay_list_end - P.lay_count_begin) * P.bpl + 1);
    r = r ? r : job_alloc(j, &P.rowstart, (size_t)(P.lay_list_end - P.lay_count_begin) * P.ncy + 2);
    r = r ? r : job_alloc(j, &P.counters, 1);
    P.host_counters = &g.slots_dev[j->slot].c;
    j->rec_first = j->owned.size();
    r = r ? r : alloc_records(j, cap_records);
    if (r) { job_release(j); delete j; return r; }
    *out = j;
    return SDFK_OK;
}

int alloc_mesh(sdfk_mesh** out, size_t cap_v, size_t cap_i)
{
    sdfk_mesh* m = new sdfk_mesh();
    int r = 0;
    r = r ? r : dev_alloc((void**)&m->vertices, std::max<size_t>(cap_v, 1) * 3 * sizeof(float));
    r = r ? r : dev_alloc((void**)&m->colors, std::max<size_t>(cap_v, 1) * 3 * sizeof(float));
    r = r ? r : dev_alloc((void**)&m->normals, std::max<size_t>(cap_v, 1) * 3 * sizeof(float));
    r = r ? r : dev_alloc((void**)&m->triangles, std::max<size_t>(cap_i, 1) * sizeof(int32_t));
    r = r ? r : dev_alloc((void**)&m->bounds, 8 * sizeof(float));
    if (r) { sdfk_mesh_free(m); return r; }
    m->lane = g.cur_lane;
    m->cap_v = cap_v; m->cap_i = cap_i;
    *out = m;
    return SDFK_OK;
}

// emit: vertices (+ AABB partials) then triangles (+ AABB reduction).  Launches only.
int launch_emit(sdfk_march_job* j, sdfk_mesh* m, int64_t vertex_base)
{
    McMeshOut M;
    memset(&M, 0, sizeof M);
    M.vertices = m->vertices; M.colors = m->colors; M.normals = m->normals; M.triangles = m->triangles;
    m->has_colors = j->P.colors != nullptr || j->colors_elided;
    if (!m->has_colors) {   // a .W-only program: every colour is (0,0,0) (Voxels.cs:88-92) -- nothing is stored, sdfk_mesh_copy clears the host array
        M.colors = nullptr;
        m->colors_valid = false;
    }
    M.cap_vertices = (uint32_t)m->cap_v;
    M.cap_indices = m->cap_i;
    M.vertex_base = vertex_base;
    M.slab_header = m->slab_header;
    M.slab_vbytes = m->has_colors ? 36 : 24;
    // MarchingCubes.cs:85-90 (row-vector T*S*T) and Mesh.cs:49-55, all float32
    const int nn[3] = {j->gnx, j->gny, j->gnz};
    for (int k = 0; k < 3; k++) {
        const float size = j->gmax[k] - j->gmin[k];
        const float sum = j->gmin[k] + j->gmax[k];
        const float center = sum * 0.5f;
        const float t1 = (float)(-(nn[k] - 1)) / 2.0f;
        M.sc[k] = size / (float)(nn[k] - 1);
        const float ts = t1 * M.sc[k];
        M.tr[k] = ts + center;
    }
    {
        const float yz = M.sc[1] * M.sc[2], xz = M.sc[0] * M.sc[2], xy = M.sc[0] * M.sc[1];
        const float det = M.sc[0] * yz;
        const float inv_det = 1.0f / det;
        M.inv[0] = yz * inv_det; M.inv[1] = xz * inv_det; M.inv[2] = xy * inv_det;
    }
#ifndef SDFK_KV_GRIDCAP
#define SDFK_KV_GRIDCAP (256 * 8)
#endif
    constexpr int vcap = SDFK_KV_GRIDCAP, tcap = 256 * 8;   // (persistent-workgroup caps were measured: +-2 us, noise)
    const int vgrid = grid_for(j->P.cap_active, (int)MC_CHUNK, vcap);
    if (!j->bounds_partial || j->bounds_blocks != vgrid) {
        if (int rr = job_alloc(j, &j->bounds_partial, (size_t)vgrid * 6)) return rr;
        j->bounds_blocks = vgrid;
    }
    M.bounds_partial = j->bounds_partial;
    M.bounds_blocks = vgrid;
    // Vertex colours of a volume its own program has just sampled: re-evaluated by the program (sdfk_vertex_colors) from
    // the (creator record, edge) descriptors k_vertices leaves, instead of gathered from the colour volume
    const bool no_vcol = !g_cfg.vcolor_eval;   // (SDFK_OPT_VCOLOR_EVAL = 0: the gather path)
    const bool vcol = j->eval_prog && j->eval_prog->writes_color && (j->P.colors || j->colors_elided) && M.colors && j->P.step == 1 && !no_vcol;
    if (vcol) {
        const size_t need = std::max<size_t>(m->cap_v, 1);
        if (!j->vdesc || j->vdesc_cap < need) {
            if (int rr = job_alloc(j, &j->vdesc, need)) return rr;
            j->vdesc_cap = need;
        }
        M.vdesc = j->vdesc;
    }
    M.bounds = m->bounds;
    M.host_bounds = g.slots_dev[j->slot].bounds;
    phase_token_wait(1);
    {
        ProfScope ps("k_vertices");
        uint32_t iso_bits;
        memcpy(&iso_bits, &j->P.iso, 4);
        if (iso_bits == 0u) hipLaunchKernelGGL(k_vertices<true>, dim3(vgrid), dim3(256), 0, g.stream, j->P, M);   // (+0.0: the usual iso value)
        else hipLaunchKernelGGL(k_vertices<false>, dim3(vgrid), dim3(256), 0, g.stream, j->P, M);
        HIPCHK(hipGetLastError());
    }
    phase_token_pass(1);
    if (vcol) {
        struct VColArgs { const uint2* vdesc; const uint32_t* rec_xy; const uint32_t* rec_z; const McCounters* counters; float* colors;
                          uint32_t cap_vertices; int32_t xbits; float iso; } V;   // (= VColArgs of sample_codegen.h)
        V.vdesc = j->vdesc; V.rec_xy = j->P.rec_xy; V.rec_z = j->P.rec_z; V.counters = j->P.counters; V.colors = M.colors;
        V.cap_vertices = M.cap_vertices; V.xbits = j->P.xbits; V.iso = j->P.iso;
        hipFunction_t fn = nullptr;
        if (int rr = program_fn(j->eval_prog, PK_VCOLORS, &fn)) return rr;
        void* params[] = {&j->eval_args, &V, j->eval_prog->kargs()};
        ProfScope ps("sdfk_vertex_colors");
        HIPCHK(hipModuleLaunchKernel(fn, (unsigned)grid_for(std::max<size_t>(m->cap_v, 1), 256, 256 * 8), 1, 1, 256, 1, 1, 0, g.stream, params, nullptr));
    }
    {
        ProfScope ps("k_triangles");
        hipLaunchKernelGGL(k_triangles, dim3(grid_for(j->P.cap_active, (int)MC_CHUNK, tcap)), dim3(256), 0, g.stream, j->P, M);
        HIPCHK(hipGetLastError());
    }
    return SDFK_OK;
}

void finalize_mesh(sdfk_march_job* j, sdfk_mesh* m, bool have_bounds)
{
    const uint32_t nghost = j->c.nghost;
    m->nv = (int64_t)j->c.total_v - (int64_t)nghost;
    m->ni = (int64_t)j->c.total_t * 3;
    m->n_active = j->c.n_emit_cells;
    m->n_case13 = j->c.n_dead;
    if (m->nv == 0) { m->bounds_valid = true; return; }   // Mesh.Measure leaves Min/Max at zero (Mesh.cs:32)
    if (have_bounds) {
        memcpy(m->h_min, g.slots[j->slot].bounds, 12);
        memcpy(m->h_max, g.slots[j->slot].bounds + 3, 12);
        m->bounds_valid = true;
    }
}

uint64_t hint_key(const sdfk_volume* v, int step, int layer_begin, int layer_end)
{
    uint64_t k = ((uint64_t)v->nx << 44) ^ ((uint64_t)v->ny << 24) ^ ((uint64_t)v->nz << 4) ^ (uint64_t)(step & 15);
    return k * 0x9E3779B97F4A7C15ull ^ ((uint64_t)(uint32_t)layer_begin << 32 | (uint32_t)layer_end) ^ ((uint64_t)v->z0 << 17);
}

// Exact path: classify, wait for the counts, size the outputs exactly, emit.
int march_exact(const sdfk_volume* v, float iso, int step, int layer_begin, int layer_end, int64_t vertex_base,
                uint64_t key, sdfk_mesh** out)
{
    *out = nullptr;
    sdfk_march_job* j = nullptr;
    int r = setup_job(v, iso, step, layer_begin, layer_end, 0, &j);
    if (r) return r;
    sdfk_mesh* m = nullptr;
    if (j->empty) {
        r = alloc_mesh(&m, 0, 0);
        if (!r) m->bounds_valid = true;
    } else {
        r = launch_classify(j, true);
        r = r ? r : wait_counters(j);
        if (!r && v->elided && j->c.n_case13 != 0) {
            // case-13 sign words: k_resolve's dead-cell test reads neighbouring VOXELS, and this volume has none
            // (SDFK_OPT_ELIDE_VOLUME): give it its storage, sample again with stores, start over
            job_release(j);
            delete j;
            if (int r2 = volume_materialize(const_cast<sdfk_volume*>(v))) return r2;
            return march_exact(v, iso, step, layer_begin, layer_end, vertex_base, key, out);
        }
        if (!r && j->c.n_active > j->P.cap_active) {   // record list too small: exact size, redo
            r = alloc_records(j, j->c.n_active);
            r = r ? r : launch_classify(j, true);
            r = r ? r : wait_counters(j);
        }
        r = r ? r : alloc_mesh(&m, (size_t)(j->c.total_v - j->c.nghost), (size_t)j->c.total_t * 3);
        if (!r && vertex_base + (int64_t)(j->c.total_v - j->c.nghost) >= (int64_t(1) << 31))
            r = fail(SDFK_ERR_UNSUPPORTED, "vertex index exceeds int32 (Mesh.Triangles is int[])");
        if (!r && j->c.n_active > 0) {
            r = launch_emit(j, m, vertex_base);
            r = r ? r : wait_counters(j);
            if (!r && j->c.overflow) r = fail(SDFK_ERR_HIP, "marching cubes: output capacity exceeded unexpectedly");
        }
        if (!r) {
            finalize_mesh(j, m, j->c.n_active > 0);
            g.hints[key] = Context::Hint{j->c.n_active, (uint32_t)m->nv, (uint32_t)m->ni};
        }
    }
    job_release(j);
    delete j;
    if (r) { if (m) sdfk_mesh_free(m); return r; }
    *out = m;
    return SDFK_OK;
}

void free_mesh_buffers(sdfk_mesh* m)
{
    if (m->borrowed) {   // the buffers belong to a GraphJob
        m->borrowed = false;
        m->vertices = m->colors = m->normals = m->bounds = nullptr;
        m->triangles = nullptr;
        return;
    }
    if (!m->external) {
        dev_free(m->vertices);   // stream-ordered pool: no sync needed
        dev_free(m->colors);
        dev_free(m->normals);
        dev_free(m->triangles);
    }
    m->external = false;
    dev_free(m->bounds);
    m->vertices = m->colors = m->normals = m->bounds = nullptr;
    m->triangles = nullptr;
}

void drop_source(sdfk_mesh* m)
{
    if (m->owns_src && m->src) sdfk_volume_free(const_cast<sdfk_volume*>(m->src));
    m->src = nullptr;
    m->owns_src = false;
}

// Completes a mesh of the speculative path (see sdfk_mesh): no-op for a finished mesh.
int mesh_resolve(sdfk_mesh* m)
{
    if (m->status) { t_err = m->error; return m->status; }
    if (!m->pending) return SDFK_OK;
    LaneScope on_lane(m->lane);   // an exact re-run queues (and allocates) where the first attempt did
    sdfk_march_job* j = m->pending;
    m->pending = nullptr;
    for (auto it = g.pending.begin(); it != g.pending.end(); ++it)
        if (*it == m) { g.pending.erase(it); break; }
    int r = SDFK_OK;
    const hipError_t e = hipEventSynchronize(m->done);
    (void)hipEventDestroy(m->done);
    m->done = nullptr;
    if (e != hipSuccess) r = fail(SDFK_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(e));
    j->c = g.slots[j->slot].c;
    // (an elided volume -- SDFK_OPT_ELIDE_VOLUME -- whose sign words contain case 13 is redone on the exact path, which gives it
    // its storage first: the dead-cell test of k_resolve reads neighbouring voxels)
    const bool needs_voxels = m->src && m->src->elided && j->c.n_case13 != 0;
    const bool fits = !needs_voxels && j->c.n_active <= j->P.cap_active && j->c.overflow == 0 &&
                      (size_t)(j->c.total_v - j->c.nghost) <= m->cap_v && (size_t)j->c.total_t * 3 <= m->cap_i;
    if (!r && fits) {
        if (m->vertex_base + (int64_t)(j->c.total_v - j->c.nghost) >= (int64_t(1) << 31))
            r = fail(SDFK_ERR_UNSUPPORTED, "vertex index exceeds int32 (Mesh.Triangles is int[])");
        else {
            finalize_mesh(j, m, true);
            g.hints[m->key] = Context::Hint{j->c.n_active, (uint32_t)m->nv, (uint32_t)m->ni};
        }
    }
    if (!m->graph_job) {   // (a job of a captured launch graph belongs to its GraphJob and is replayed)
        job_release(j);
        delete j;
    }
    if (!r && !fits) {   // the guess was too small: the exact two-phase path, into the same handle
        sdfk_mesh* x = nullptr;
        r = march_exact(m->src, m->iso, m->step, m->layer_begin, m->layer_end, m->vertex_base, m->key, &x);
        if (m->graph_job) graph_job_retire(m, true);   // its capacities are too small for this scene: rebuilt on a later call
        if (!r) {
            free_mesh_buffers(m);
            m->nv = x->nv; m->ni = x->ni;
            m->vertices = x->vertices; m->colors = x->colors; m->normals = x->normals; m->triangles = x->triangles;
            m->bounds = x->bounds;
            memcpy(m->h_min, x->h_min, sizeof m->h_min);
            memcpy(m->h_max, x->h_max, sizeof m->h_max);
            m->bounds_valid = x->bounds_valid;
            m->n_active = x->n_active; m->n_case13 = x->n_case13;
            m->cap_v = x->cap_v; m->cap_i = x->cap_i;
            m->has_colors = x->has_colors;
            m->colors_valid = x->colors_valid;
            delete x;
        }
    }
    drop_source(m);
    if (r) { m->nv = m->ni = 0; m->status = r; m->error = t_err; }
    return r;
}

// every pending mesh that still depends on the contents of `v` (called before `v` changes or dies)
void resolve_dependents(const sdfk_volume* v)
{
    for (;;) {
        sdfk_mesh* hit = nullptr;
        for (sdfk_mesh* m : g.pending)
            if (m->src == v) { hit = m; break; }
        if (!hit) return;
        (void)mesh_resolve(hit);   // an error stays in the mesh (sticky)
    }
}

// A mesh whose arrays are sections of a slab payload at `dst` (64-byte header, then V | (C) | N | T laid out for the
// capacities): what a sharded step emits into when it writes straight into its all-gather send buffer.  The
// capacities are the size hints scaled up to what `capacity` bytes hold (at most the usual +25 %); returns false when
// not even the hints fit (the caller then takes the ordinary path and packs).
bool external_mesh(char* dst, int64_t capacity, bool colors, uint32_t nv_hint, uint32_t ni_hint, sdfk_mesh** out)
{
    const int64_t vb = colors ? 36 : 24, avail = capacity - SDFK_SLAB_HEADER_BYTES;
    const int64_t min_v = (int64_t)nv_hint + 64, min_i = (int64_t)ni_hint + 192;
    if (avail < vb * min_v + 4 * min_i) return false;
    const double scale = std::min(1.25, (double)avail / (double)(vb * min_v + 4 * min_i));
    int64_t cap_v = std::max<int64_t>(min_v, (int64_t)((double)min_v * scale));
    int64_t cap_i = (avail - vb * cap_v) / 4;
    cap_i = std::min<int64_t>(cap_i, (int64_t)ni_hint + ni_hint / 4 + 12288);
    cap_i -= cap_i % 3;
    if (cap_i < min_i - 2) return false;
    sdfk_mesh* m = new sdfk_mesh();
    if (dev_alloc((void**)&m->bounds, 8 * sizeof(float))) { delete m; return false; }
    char* q = dst + SDFK_SLAB_HEADER_BYTES;
    m->vertices = (float*)q; q += 12 * cap_v;
    if (colors) { m->colors = (float*)q; q += 12 * cap_v; }
    else m->colors = nullptr;       // no colour section: k_vertices skips the (all-zero) colour stores
    m->normals = (float*)q; q += 12 * cap_v;
    m->triangles = (int32_t*)q;
    m->external = true;
    m->slab_header = dst;
    m->has_colors = colors;
    m->lane = g.cur_lane;
    m->cap_v = (size_t)cap_v; m->cap_i = (size_t)cap_i;
    *out = m;
    return true;
}

// MarchingCubes.CreateMesh on the cell layers [layer_begin, layer_end) of a volume / slab.  emit_dst != nullptr (sharded
// step): on the speculative path the mesh is emitted straight into that slab payload (external_mesh).
int march_range(const sdfk_volume* v, float iso, int step, int layer_begin, int layer_end, int64_t vertex_base, sdfk_mesh** out,
                char* emit_dst = nullptr, int64_t emit_capacity = 0)
{
    *out = nullptr;
    const uint64_t key = hint_key(v, step, layer_begin, layer_end);
    auto it = g.hints.find(key);
    if (it != g.hints.end()) {
        // Speculative path: the sizes of the previous mesh of this shape (+25 % and a floor)
        // size every buffer; classification AND emit are queued back to back and the handle
        // is returned without waiting: the host meets the GPU only when a result is read.
        const Context::Hint h = it->second;
        sdfk_march_job* j = nullptr;
        int r = setup_job(v, iso, step, layer_begin, layer_end, (size_t)h.n_active + h.n_active / 4 + 4096, &j);
        if (r) return r;
        if (!j->empty) {
            sdfk_mesh* m = nullptr;
            if (!(emit_dst && external_mesh(emit_dst, emit_capacity, v->colors != nullptr, h.nv, h.ni, &m)))
                r = alloc_mesh(&m, (size_t)h.nv + h.nv / 4 + 4096, (size_t)h.ni + h.ni / 4 + 12288);
            r = r ? r : launch_classify(j, false);
            r = r ? r : launch_emit(j, m, vertex_base);
            if (!r) {
                hipError_t e = hipEventCreateWithFlags(&m->done, hipEventDisableTiming);
                if (e == hipSuccess) e = hipEventRecord(m->done, g.stream);
                if (e != hipSuccess) r = fail(SDFK_ERR_HIP, "hipEventRecord: %s", hipGetErrorString(e));
            }
            if (r) {
                (void)hipStreamSynchronize(g.stream);
                if (m) { if (m->done) (void)hipEventDestroy(m->done); m->done = nullptr; sdfk_mesh_free(m); }
                job_release(j);
                delete j;
                return r;
            }
            m->pending = j;
            m->src = v;
            m->iso = iso; m->step = step; m->layer_begin = layer_begin; m->layer_end = layer_end;
            m->vertex_base = vertex_base; m->key = key;
            g.pending.push_back(m);
            while (g.pending.size() > Context::MAX_PENDING) (void)mesh_resolve(g.pending.front());
            *out = m;
            return SDFK_OK;
        }
        job_release(j);
        delete j;
    }
    return march_exact(v, iso, step, layer_begin, layer_end, vertex_base, key, out);
}

}  // namespace

// ---------------------------------------------------------------------------
// captured launch graphs for repeat sdfk_sample_march jobs
// ---------------------------------------------------------------------------
// On launch-bound grids (<= 2^24 voxels) a job is nine small dependent kernels: queueing them costs the host 23 us, more
// than the GPU needs for the job next to the others in flight (tools/ubench/ub_graph.hip: 9 launches 23.4 us, one
// hipGraphLaunch of the captured chain 6.0 us; the chain takes the GPU the same time either way).  So the speculative
// job of a (program, grid, clip, iso) on a lane is built ONCE -- its own volume, workspace sized from the shape's hints,
// mesh buffers, result slot -- its launches are captured into a hipGraph, and every later call for that key on that lane
// is one hipGraphLaunch.  The mesh handle borrows the GraphJob's buffers until it is freed; a result that does not fit the
// captured capacities is redone on the exact path as always, and the GraphJob is rebuilt with the new hints.
namespace {

struct GraphJob {
    const sdfk_program* prog = nullptr;
    int nx = 0, ny = 0, nz = 0, clip = 0, lane = 0;
    float mn[3], mx[3], iso = 0.0f;
    sdfk_volume* vol = nullptr;
    sdfk_march_job* job = nullptr;
    sdfk_mesh* proto = nullptr;     // owns the mesh buffers and their capacities
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t key = 0;
    sdfk_mesh* borrower = nullptr;  // the live handle whose arrays are proto's (at most one)
    // slab form (sdfk_slab_enqueue): the caller's slab volume and send buffer, nothing borrowed, never "busy"
    bool slab = false;
    sdfk_volume* ext_vol = nullptr;
    void* dst = nullptr;
    void* post_out = nullptr;       // compact payloads: where k_payload_compact (the graph's last node) writes
    int64_t capacity = 0;
    int lb = 0, le = 0;
    Context::Hint hint{};           // the size hints the capacities were derived from (other hints now: rebuild)
    SampleArgs args;                // what sample_impl recorded in the volume (restored on every replay)
    hipEvent_t ran = nullptr;       // after the latest run (a handle dropped unread leaves its run in flight)
    bool busy = false, stale = false;
    size_t bytes = 0;
    uint64_t last_use = 0;
};

void graph_job_destroy(GraphJob* q)
{
    (void)hipStreamSynchronize(lane_stream(q->lane));   // its kernels may still be queued
    if (q->exec) (void)hipGraphExecDestroy(q->exec);
    if (q->graph) (void)hipGraphDestroy(q->graph);
    if (q->ran) (void)hipEventDestroy(q->ran);
    if (sdfk_mesh* b = q->borrower) {   // (only at shutdown: a handle that outlives the library reads as failed)
        b->graph_job = nullptr; b->borrowed = false;
        b->vertices = b->colors = b->normals = b->bounds = nullptr; b->triangles = nullptr;
        b->nv = b->ni = 0; b->pending = nullptr; b->src = nullptr;
        if (!b->status) { b->status = SDFK_ERR_INVALID; b->error = "the library was shut down"; }
    }
    LaneScope on_lane(q->lane);
    if (q->job) { job_release(q->job); delete q->job; }
    if (q->proto) { q->proto->graph_job = nullptr; q->proto->borrowed = false; sdfk_mesh_free(q->proto); }
    if (q->vol) sdfk_volume_free(q->vol);
    if (q->prog) program_release(const_cast<sdfk_program*>(q->prog));
    g.graph_bytes -= std::min(g.graph_bytes, q->bytes);
    for (auto it = g.graph_jobs.begin(); it != g.graph_jobs.end(); ++it)
        if (*it == q) { g.graph_jobs.erase(it); break; }
    delete q;
}

void graph_jobs_destroy_all()
{
    while (!g.graph_jobs.empty()) graph_job_destroy(g.graph_jobs.back());
}

// the mesh no longer needs its GraphJob (freed, or redone on the exact path because the captured capacities were too small)
void graph_job_retire(sdfk_mesh* m, bool too_small)
{
    GraphJob* q = m->graph_job;
    if (!q) return;
    if (too_small) q->stale = true;
    q->busy = false;
    q->borrower = nullptr;
    m->graph_job = nullptr;
    if (q->prog && q->prog->orphaned) {   // its program handle was destroyed while this mesh was out
        if (m->src == q->vol) m->src = nullptr;
        if (m->pending == q->job) m->pending = nullptr;
        graph_job_destroy(q);
    }
}

void graph_jobs_forget_program(const sdfk_program* p)
{
    for (size_t i = 0; i < g.graph_jobs.size();) {
        if (g.graph_jobs[i]->prog == p && !g.graph_jobs[i]->busy) graph_job_destroy(g.graph_jobs[i]);   // (erases the entry)
        else i++;
    }
}

void graph_jobs_forget_volume(const sdfk_volume* v)
{
    for (size_t i = 0; i < g.graph_jobs.size();) {
        if (g.graph_jobs[i]->slab && g.graph_jobs[i]->ext_vol == v) graph_job_destroy(g.graph_jobs[i]);   // (erases the entry)
        else i++;
    }
}

bool graphs_enabled(int64_t nvox)
{
    const int mode = g_cfg.graphs;   // SDFK_OPT_GRAPHS -- 0: never, 1: launch-bound grids, 2: every grid
    return mode == 2 || (mode == 1 && nvox <= (int64_t(1) << 24));   // (measured: 64^3 35 -> 23.5 us per job, 128^3 33 -> 26, 256^3 42 -> 39; 320^3 and up 0-3 % slower)
}

bool graphs_enabled_slab(int64_t nvox)   // (a sharded step also pays for a collective call on the host: graphs pay up to larger slabs)
{
    const int mode = g_cfg.graphs;
    return mode == 2 || (mode == 1 && nvox <= (int64_t(1) << 25));
}

// Queue the job of (p, grid, clip, iso) on the current lane from a captured graph.  *out stays null when graphs do not
// apply (no size hints yet, no free GraphJob and the limits are reached, capture not possible): the caller then takes
// the ordinary path.
int graph_sample_march(const sdfk_program* p, const float mn[3], const float mx[3], int nx, int ny, int nz, int clip, float iso, sdfk_mesh** out)
{
    *out = nullptr;
    const int lane = g.cur_lane;
    if (lane == 0 || g.prof_on || g.sampler_only || !mn || !mx || g_graph_build_failures >= 8) return SDFK_OK;
    GraphJob* q = nullptr;
    size_t alive = 0;
    for (size_t i = 0; i < g.graph_jobs.size();) {
        GraphJob* c = g.graph_jobs[i];
        const bool same = !c->slab && c->prog == p && c->nx == nx && c->ny == ny && c->nz == nz && c->clip == clip && c->lane == lane &&
                          memcmp(&c->iso, &iso, 4) == 0 && memcmp(c->mn, mn, 12) == 0 && memcmp(c->mx, mx, 12) == 0;
        if (same && !c->busy && c->stale) { graph_job_destroy(c); continue; }   // (erases g.graph_jobs[i])
        if (same) alive++;
        if (same && !c->busy && !q) q = c;
        i++;
    }
    if (!q) {
        // build: everything a job needs, allocated up front, sized from the hints of this grid shape
        sdfk_volume probe;
        probe.nx = nx; probe.ny = ny; probe.nz = nz; probe.nz_global = nz; probe.z0 = 0;
        const uint64_t key = hint_key(&probe, 1, 0, std::max(nz - 1, 0));
        auto hit = g.hints.find(key);
        if (hit == g.hints.end() || nx < 2 || ny < 2 || nz < 2) return SDFK_OK;
        {   // A captured job only pays when the IDENTICAL job comes again: it is built on the second sighting of the full key
            // (program, grid, bounds, clip, iso) on this lane, not as soon as the grid shape has hints -- a caller whose
            // program or bounds change per call (an animated SDF) would otherwise pay an un-captured run, a capture, an
            // instantiation and a set of allocations on every call, and an eviction (a stream synchronisation) from the 25th on.
            uint64_t fk = fnv1a64(std::string((const char*)&p, sizeof p), 0xcbf29ce484222325ull);
            const int dims[5] = {nx, ny, nz, clip, lane};
            fk = fnv1a64(std::string((const char*)dims, sizeof dims), fk);
            fk = fnv1a64(std::string((const char*)mn, 12) + std::string((const char*)mx, 12) + std::string((const char*)&iso, 4), fk);
            if (g.graph_sightings.size() > 4096) g.graph_sightings.clear();
            if (g.graph_sightings[fk]++ == 0) return SDFK_OK;
        }
        if (alive >= 3 || g.graph_jobs.size() >= 24 || g.graph_bytes > (size_t(4) << 30)) {
            GraphJob* lru = nullptr;   // make room: the least recently used free one, if any
            for (GraphJob* c : g.graph_jobs)
                if (!c->busy && (!lru || c->last_use < lru->last_use)) lru = c;
            if (!lru || alive >= 3) return SDFK_OK;
            graph_job_destroy(lru);
        }
        const Context::Hint h = hit->second;
        q = new GraphJob();
        q->prog = p; const_cast<sdfk_program*>(p)->refs++;
        q->nx = nx; q->ny = ny; q->nz = nz; q->clip = clip; q->lane = lane; q->iso = iso; q->key = key;
        memcpy(q->mn, mn, 12); memcpy(q->mx, mx, 12);
        g.graph_jobs.push_back(q);
        const size_t before = [] { size_t b = 0; for (auto& kv : g.live_blocks) b += kv.second.size; return b; }();
        int r = sdfk_volume_create(nx, ny, nz, mn, mx, p->writes_color ? 1 : 0, &q->vol);
        // (first run outside the capture: compiles / loads the kernels, allocates the sign-bit arrays, marks the volume as this program's output)
        if (!r) r = sample_impl(p, q->vol, clip, iso);
        if (!r) r = setup_job(q->vol, iso, 1, 0, std::max(nz - 1, 0), (size_t)h.n_active + h.n_active / 4 + 4096, &q->job);
        if (!r && (q->job->empty || !q->job->have_bits)) r = -1;   // (an iso value that never compares equal, NaN, leaves the sign-bit pass to the job)
        if (!r) r = alloc_mesh(&q->proto, (size_t)h.nv + h.nv / 4 + 4096, (size_t)h.ni + h.ni / 4 + 12288);
        if (!r) r = launch_classify(q->job, false);
        if (!r) r = launch_emit(q->job, q->proto, 0);   // (allocates the AABB partials: the captured run below does not allocate)
        if (!r) {
            hipStream_t st = lane_stream(lane);
            hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                int rc = sample_impl(p, q->vol, clip, iso);
                if (!rc) rc = launch_classify(q->job, false);
                if (!rc) rc = launch_emit(q->job, q->proto, 0);
                e = hipStreamEndCapture(st, &q->graph);
                if (e == hipSuccess && rc) e = hipErrorUnknown;
            }
            if (e == hipSuccess) e = hipGraphInstantiate(&q->exec, q->graph, nullptr, nullptr, 0);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&q->ran, hipEventDisableTiming);
            if (e != hipSuccess) { (void)hipGetLastError(); r = -1; }
        }
        if (r) {   // graphs are an optimisation: any failure here means "take the ordinary path"
            g_graph_build_failures++;
            graph_job_destroy(q);
            return SDFK_OK;
        }
        const size_t after = [] { size_t b = 0; for (auto& kv : g.live_blocks) b += kv.second.size; return b; }();
        q->bytes = after > before ? after - before : 0;
        g.graph_bytes += q->bytes;
        // (the un-captured first run above already queued this call's job: no graph launch for it)
    } else {
        if (hipEventQuery(q->ran) != hipSuccess) (void)hipEventSynchronize(q->ran);   // the previous run raises `overflow` in the slot itself
        g.slots[q->job->slot].c.overflow = 0;
        const hipError_t e = hipGraphLaunch(q->exec, lane_stream(lane));
        if (e != hipSuccess) { (void)hipGetLastError(); q->stale = true; return SDFK_OK; }
        g.graph_launches++;
    }
    q->busy = true;
    q->last_use = ++g.graph_clock;
    sdfk_mesh* m = new sdfk_mesh();
    const sdfk_mesh* pr = q->proto;
    m->vertices = pr->vertices; m->colors = pr->colors; m->normals = pr->normals; m->triangles = pr->triangles; m->bounds = pr->bounds;
    m->cap_v = pr->cap_v; m->cap_i = pr->cap_i; m->has_colors = pr->has_colors; m->colors_valid = pr->colors_valid; m->lane = lane;
    m->borrowed = true;
    m->graph_job = q;
    q->borrower = m;
    hipError_t e = hipEventCreateWithFlags(&m->done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(m->done, lane_stream(lane));
    if (e == hipSuccess) e = hipEventRecord(q->ran, lane_stream(lane));
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(lane_stream(lane));
        if (m->done) (void)hipEventDestroy(m->done);
        m->borrowed = false; m->vertices = m->colors = m->normals = m->bounds = nullptr; m->triangles = nullptr;
        q->busy = false;
        q->borrower = nullptr;
        delete m;
        return fail(SDFK_ERR_HIP, "hipEventRecord: %s", hipGetErrorString(e));
    }
    m->pending = q->job;
    m->src = q->vol;
    m->iso = iso; m->step = 1; m->layer_begin = 0; m->layer_end = std::max(nz - 1, 0);
    m->vertex_base = 0; m->key = q->key;
    g.pending.push_back(m);
    while (g.pending.size() > Context::MAX_PENDING) (void)mesh_resolve(g.pending.front());
    *out = m;
    return SDFK_OK;
}

// What a sharded step with compact payloads queues right behind its last kernel: the plain payload at the step's dst ->
// the compact payload at `out` (k_payload_compact, mc_kernels.hip).  Part of the captured step graph.
struct PostCompact { char* out; int64_t out_capacity; unsigned long long* ticket; };
int launch_post_compact(const PostCompact* pc, const void* plain, int64_t plain_capacity)
{
    if (!pc) return SDFK_OK;
    const int64_t words = plain_capacity / 4;
    ProfScope ps("k_payload_compact");
    hipLaunchKernelGGL(k_payload_compact, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((words + 4095) / 4096, 1024))), dim3(256), 0, g.stream,
                       (const char*)plain, pc->out, pc->out_capacity, pc->ticket);
    HIPCHK(hipGetLastError());
    return SDFK_OK;
}

// The same for a sharded step (sdfk_slab_enqueue on a lane): sample the caller's slab, mesh its layers straight into the
// caller's send buffer, header written by the last kernel -- eleven launches as ONE hipGraphLaunch per step.  Key: program,
// slab volume, clip, iso, layer range, destination and capacity, lane; the GraphJob keeps the job's workspace and result
// slot, the mesh arrays ARE sections of the destination.  A step whose capacities were too small says so in its header
// (every rank then redoes it exactly, sdfkit_amd/dist.py), which renews the size hints: a GraphJob built from other hints
// than today's is rebuilt.  *handled stays false when graphs do not apply (the caller then takes the ordinary path).
int graph_slab_enqueue(const sdfk_program* p, sdfk_volume* slab, int clip, float iso, int lb, int le, void* dst, int64_t capacity, bool* handled,
                       const PostCompact* pc = nullptr)
{
    *handled = false;
    const int lane = g.cur_lane;
    if (lane == 0 || g.prof_on || g.sampler_only || g_graph_build_failures >= 8) return SDFK_OK;
    const uint64_t key = hint_key(slab, 1, lb, le);
    auto hit = g.hints.find(key);
    if (hit == g.hints.end()) return SDFK_OK;
    const Context::Hint h = hit->second;
    GraphJob* q = nullptr;
    for (GraphJob* c : g.graph_jobs)
        if (c->slab && c->prog == p && c->ext_vol == slab && c->clip == clip && memcmp(&c->iso, &iso, 4) == 0 && c->lb == lb && c->le == le &&
            c->dst == dst && c->capacity == capacity && c->lane == lane && c->post_out == (pc ? (void*)pc->out : nullptr)) { q = c; break; }
    if (q && memcmp(&q->hint, &h, sizeof h) != 0) { graph_job_destroy(q); q = nullptr; }
    if (q) {
        resolve_dependents(slab);
        volume_values_changed(slab);
        const hipError_t e = hipGraphLaunch(q->exec, lane_stream(lane));
        if (e != hipSuccess) { (void)hipGetLastError(); graph_job_destroy(q); return SDFK_OK; }
        // (what sample_impl leaves in the volume: its values, sign bits and colours are this program's output again)
        sdfk_program* pp = const_cast<sdfk_program*>(p);
        pp->refs++;
        slab->sampled_by = pp;
        slab->sampled_args = q->args;
        slab->bits_iso = iso;
        slab->bits_valid = true;
        g.graph_launches++;
        q->last_use = ++g.graph_clock;
        *handled = true;
        return SDFK_OK;
    }
    // build.  Cheap misfits first (nothing queued yet): they are not failures, the ordinary path packs instead.
    if (slab->nx < 2 || slab->ny < 2 || slab->nz < 2 || lb >= le) return SDFK_OK;
    sdfk_mesh* proto = nullptr;
    if (!external_mesh((char*)dst, capacity, slab->colors != nullptr, h.nv, h.ni, &proto)) return SDFK_OK;
    if (g.graph_jobs.size() >= 24) {
        GraphJob* lru = nullptr;
        for (GraphJob* c : g.graph_jobs)
            if (!c->busy && (!lru || c->last_use < lru->last_use)) lru = c;
        if (!lru) { sdfk_mesh_free(proto); return SDFK_OK; }
        graph_job_destroy(lru);
    }
    q = new GraphJob();
    q->slab = true;
    q->prog = p; const_cast<sdfk_program*>(p)->refs++;
    q->ext_vol = slab; q->clip = clip; q->iso = iso; q->lb = lb; q->le = le; q->dst = dst; q->capacity = capacity; q->lane = lane;
    q->nx = slab->nx; q->ny = slab->ny; q->nz = slab->nz;
    q->key = key; q->hint = h; q->proto = proto;
    q->post_out = pc ? (void*)pc->out : nullptr;
    g.graph_jobs.push_back(q);
    int r = sample_impl(p, slab, clip, iso);   // (outside the capture: loads the kernels, allocates the sign-bit arrays; this call's run)
    if (!r) r = setup_job(slab, iso, 1, lb, le, (size_t)h.n_active + h.n_active / 4 + 4096, &q->job);
    if (!r && (q->job->empty || !q->job->have_bits)) r = -1;
    if (!r) r = launch_classify(q->job, false);
    if (!r) r = launch_emit(q->job, q->proto, 0);
    if (!r) r = launch_post_compact(pc, dst, capacity);
    if (!r) {
        q->args = slab->sampled_args;
        hipStream_t st = lane_stream(lane);
        hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        if (e == hipSuccess) {
            int rc = sample_impl(p, slab, clip, iso);
            if (!rc) rc = launch_classify(q->job, false);
            if (!rc) rc = launch_emit(q->job, q->proto, 0);
            if (!rc) rc = launch_post_compact(pc, dst, capacity);
            e = hipStreamEndCapture(st, &q->graph);
            if (e == hipSuccess && rc) e = hipErrorUnknown;
        }
        if (e == hipSuccess) e = hipGraphInstantiate(&q->exec, q->graph, nullptr, nullptr, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); r = -1; }
    }
    if (r) {   // (whatever was queued is harmless: the ordinary path redoes the step behind it on the same lane)
        g_graph_build_failures++;
        graph_job_destroy(q);
        return SDFK_OK;
    }
    q->last_use = ++g.graph_clock;
    *handled = true;
    return SDFK_OK;
}

}  // namespace

extern "C" int sdfk_graph_stats(int64_t* jobs, int64_t* launches, int64_t* device_bytes)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (jobs) *jobs = (int64_t)g.graph_jobs.size();
    if (launches) *launches = g.graph_launches;
    if (device_bytes) *device_bytes = (int64_t)g.graph_bytes;
    return SDFK_OK;
}

extern "C" int sdfk_march_begin(const sdfk_volume* v, float iso_value, int32_t layer_begin, int32_t layer_end,
                                sdfk_march_job** job, int64_t* n_vertices, int64_t* n_indices)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!v || !job) return fail(SDFK_ERR_INVALID, "sdfk_march_begin: null argument");
    *job = nullptr;
    if (int r = require_init()) return r;
    sdfk_march_job* j = nullptr;
    int r = setup_job(v, iso_value, 1, layer_begin, layer_end, 0, &j);
    if (r) return r;
    if (!j->empty) {
        r = launch_classify(j, true);
        r = r ? r : wait_counters(j);
        if (!r && j->c.n_active > j->P.cap_active) {
            r = alloc_records(j, j->c.n_active);
            r = r ? r : launch_classify(j, true);
            r = r ? r : wait_counters(j);
        }
        if (r) { job_release(j); delete j; return r; }
    }
    if (n_vertices) *n_vertices = (int64_t)j->c.total_v - (int64_t)j->c.nghost;
    if (n_indices) *n_indices = (int64_t)j->c.total_t * 3;
    *job = j;
    return SDFK_OK;
}

extern "C" int sdfk_march_finish(sdfk_march_job* j, int64_t vertex_base, sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!j || !out) return fail(SDFK_ERR_INVALID, "sdfk_march_finish: null argument");
    *out = nullptr;
    if (int r = require_init()) return r;
    if (j->finished) return fail(SDFK_ERR_INVALID, "march job already finished");
    const int64_t nv = (int64_t)j->c.total_v - (int64_t)j->c.nghost;
    if (vertex_base + nv >= (int64_t(1) << 31)) return fail(SDFK_ERR_UNSUPPORTED, "vertex index exceeds int32 (Mesh.Triangles is int[])");
    sdfk_mesh* m = nullptr;
    int r = alloc_mesh(&m, (size_t)nv, (size_t)j->c.total_t * 3);
    if (r) return r;
    j->finished = true;
    const bool work = !j->empty && j->c.n_active > 0;
    if (work) {
        r = launch_emit(j, m, vertex_base);
        r = r ? r : wait_counters(j);
        if (!r && j->c.overflow) r = fail(SDFK_ERR_HIP, "marching cubes: output capacity exceeded unexpectedly");
        if (r) { sdfk_mesh_free(m); return r; }
    }
    finalize_mesh(j, m, work);
    *out = m;
    return SDFK_OK;
}

extern "C" void sdfk_march_job_free(sdfk_march_job* job)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!job) return;
    job_release(job);   // stream-ordered pool: no sync needed
    delete job;
}

extern "C" int sdfk_march(const sdfk_volume* v, float iso_value, int32_t step, sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!v || !out) return fail(SDFK_ERR_INVALID, "sdfk_march: null argument");
    *out = nullptr;
    if (int r = require_init()) return r;
    if (v->z0 != 0 || v->nz != v->nz_global) return fail(SDFK_ERR_INVALID, "sdfk_march needs a whole volume; use sdfk_march_slab for slabs");
    return march_range(v, iso_value, step, 0, std::max(v->nz_global - 1, 0), 0, out);
}

extern "C" int sdfk_march_slab(const sdfk_volume* v, float iso_value, int32_t layer_begin, int32_t layer_end,
                               int64_t vertex_base, sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!v || !out) return fail(SDFK_ERR_INVALID, "sdfk_march_slab: null argument");
    *out = nullptr;
    if (int r = require_init()) return r;
    return march_range(v, iso_value, 1, layer_begin, layer_end, vertex_base, out);
}

extern "C" int sdfk_sample_march_slab(const sdfk_program* p, sdfk_volume* slab, int32_t clip_to_bounds, float iso_value,
                                      int32_t layer_begin, int32_t layer_end, int64_t vertex_base, sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || !slab || !out) return fail(SDFK_ERR_INVALID, "sdfk_sample_march_slab: null argument");
    *out = nullptr;
    if (int r = require_init()) return r;
    if (int r = sample_impl(p, slab, clip_to_bounds, iso_value)) return r;
    return march_range(slab, iso_value, 1, layer_begin, layer_end, vertex_base, out);
}

extern "C" int sdfk_mesh_pack(const sdfk_mesh* m, void* dst, int64_t capacity_bytes, int64_t* needed_bytes)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!m || !dst) return fail(SDFK_ERR_INVALID, "sdfk_mesh_pack: null argument");
    if (int r = require_init()) return r;
    if (m->lane != g.cur_lane) const_cast<sdfk_mesh*>(m)->used_on_main = true;
    if (m->pending && !m->status) {
        // the job is still queued: the device packs from the job's own counters, the host is
        // not involved (needed_bytes is unknown here: -1; the header carries the counts)
        if (capacity_bytes < SDFK_SLAB_HEADER_BYTES) return fail(SDFK_ERR_INVALID, "sdfk_mesh_pack: capacity below the header size");
        PackArgs A;
        A.counters = m->pending->P.counters;
        A.cap_active = m->pending->P.cap_active;
        A.cap_v = (uint32_t)std::min<size_t>(m->cap_v, 0xffffffffu);
        A.cap_i = m->cap_i;
        A.vertices = m->vertices; A.colors = m->colors; A.normals = m->normals; A.triangles = m->triangles;
        A.bounds = m->bounds;
        A.dst = (char*)dst;
        A.capacity = capacity_bytes;
        A.vbytes = m->has_colors ? 36 : 24;
        if (m->lane != g.cur_lane) return fail(SDFK_ERR_INVALID, "sdfk_mesh_pack: queued mesh belongs to another stream");
        ProfScope ps("k_pack");
        hipLaunchKernelGGL(k_pack_pending, dim3(grid_for(m->cap_v * 9 + m->cap_i, 256, 1024)), dim3(256), 0, g.stream, A);
        HIPCHK(hipGetLastError());
        if (needed_bytes) *needed_bytes = -1;
        return SDFK_OK;
    }
    if (int r = mesh_resolve(const_cast<sdfk_mesh*>(m))) return r;
    const int vbytes = m->has_colors ? 36 : 24;
    const int64_t vb = m->nv * 12, need = SDFK_SLAB_HEADER_BYTES + (int64_t)vbytes * m->nv + m->ni * 4;
    if (needed_bytes) *needed_bytes = need;
    if (capacity_bytes < SDFK_SLAB_HEADER_BYTES) return fail(SDFK_ERR_INVALID, "sdfk_mesh_pack: capacity below the header size");
    hipLaunchKernelGGL(k_slab_header, dim3(1), dim3(64), 0, g.stream, (SlabHeader*)dst, (int64_t)m->nv, (int64_t)m->ni,
                       (const float*)m->bounds, vbytes);
    HIPCHK(hipGetLastError());
    if (need > capacity_bytes) return SDFK_OK;
    char* q = (char*)dst + SDFK_SLAB_HEADER_BYTES;
    if (vb) {
        HIPCHK(hipMemcpyAsync(q, m->vertices, vb, hipMemcpyDeviceToDevice, g.stream));
        q += vb;
        if (m->has_colors) {
            HIPCHK(hipMemcpyAsync(q, m->colors, vb, hipMemcpyDeviceToDevice, g.stream));
            q += vb;
        }
        HIPCHK(hipMemcpyAsync(q, m->normals, vb, hipMemcpyDeviceToDevice, g.stream));
        q += vb;
    }
    if (m->ni) HIPCHK(hipMemcpyAsync(q, m->triangles, m->ni * 4, hipMemcpyDeviceToDevice, g.stream));
    return SDFK_OK;
}

// One sharded step of a rank: [lane section] sample the slab -> mesh it into the payload at dst.  caller_stream_waits: the
// caller's stream waits for the section (sdfk_slab_enqueue); the library's own step driver (dist_rccl.h) orders its exchange
// stream with an event of its own instead.
static int slab_enqueue_impl(const sdfk_program* p, sdfk_volume* slab, int32_t clip_to_bounds, float iso_value, int32_t layer_begin,
                             int32_t layer_end, void* dst, int64_t capacity_bytes, int32_t lane, void* wait_hip_event, bool caller_stream_waits,
                             const PostCompact* pc = nullptr)
{
    int r = lane > 0 ? sdfk_lane_begin(lane, wait_hip_event) : SDFK_OK;
    if (r) return r;
    sdfk_mesh* m = nullptr;
    r = require_init();
    if (!r && lane > 0 && graphs_enabled_slab((int64_t)slab->nx * slab->ny * slab->nz)) {   // the repeat step as ONE captured graph launch
        bool handled = false;
        r = graph_slab_enqueue(p, slab, clip_to_bounds ? 1 : 0, iso_value, layer_begin, layer_end, dst, capacity_bytes, &handled, pc);
        if (handled || r) {
            const int r2 = sdfk_lane_end(caller_stream_waits ? 1 : 0);
            return r ? r : r2;
        }
    }
    if (!r) r = sample_impl(p, slab, clip_to_bounds, iso_value);
    // With size hints for this slab shape the mesh is emitted STRAIGHT into the payload at dst (its arrays are the
    // payload's sections, k_triangles writes the header): no pack launch, no second copy of the mesh.
    if (!r) r = march_range(slab, iso_value, 1, layer_begin, layer_end, 0, &m, (char*)dst, capacity_bytes);
    if (!r && !(m->external && m->pending)) {   // first call of a shape (exact path) or hints that do not fit: pack
        int64_t need = 0;
        r = sdfk_mesh_pack(m, dst, capacity_bytes, &need);
    }
    if (m) sdfk_mesh_free(m);   // stream-ordered: the kernels above still use it
    if (!r) r = launch_post_compact(pc, dst, capacity_bytes);
    if (lane > 0) {
        const int r2 = sdfk_lane_end(caller_stream_waits ? 1 : 0);
        if (!r) r = r2;
    }
    return r;
}

extern "C" int sdfk_slab_enqueue(const sdfk_program* p, sdfk_volume* slab, int32_t clip_to_bounds, float iso_value,
                                 int32_t layer_begin, int32_t layer_end, void* dst, int64_t capacity_bytes,
                                 int32_t lane, void* wait_hip_event)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || !slab || !dst) return fail(SDFK_ERR_INVALID, "sdfk_slab_enqueue: null argument");
    if (capacity_bytes < SDFK_SLAB_HEADER_BYTES) return fail(SDFK_ERR_INVALID, "sdfk_slab_enqueue: capacity below the header size");
    return slab_enqueue_impl(p, slab, clip_to_bounds, iso_value, layer_begin, layer_end, dst, capacity_bytes, lane, wait_hip_event, true);
}

static int slabs_rebase(void* gathered, int32_t world, int64_t stride_bytes, void* headers_mirror)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!gathered || world < 1 || stride_bytes < SDFK_SLAB_HEADER_BYTES) return fail(SDFK_ERR_INVALID, "sdfk_slabs_rebase: bad argument");
    if (int r = require_init()) return r;
    if (world == 1 && !headers_mirror) return SDFK_OK;
    hipLaunchKernelGGL(k_slabs_rebase, dim3(64, world), dim3(256), 0, g.stream, (char*)gathered, (int)world, (int64_t)stride_bytes,
                       (SlabHeader*)headers_mirror, 0);
    HIPCHK(hipGetLastError());
    return SDFK_OK;
}

extern "C" int sdfk_slabs_rebase(void* gathered, int32_t world, int64_t stride_bytes)
{
    return slabs_rebase(gathered, world, stride_bytes, nullptr);
}

extern "C" int sdfk_slabs_rebase_mirror(void* gathered, int32_t world, int64_t stride_bytes, void* headers_mirror)
{
    if (!headers_mirror) return fail(SDFK_ERR_INVALID, "sdfk_slabs_rebase_mirror: null mirror");
    return slabs_rebase(gathered, world, stride_bytes, headers_mirror);
}

extern "C" int sdfk_march_host(const float* values, const float* colors3, int32_t nx, int32_t ny, int32_t nz,
                               const float min[3], const float max[3], float iso_value, int32_t step,
                               sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!values || !out) return fail(SDFK_ERR_INVALID, "sdfk_march_host: null argument");
    sdfk_volume* v = nullptr;
    int r = sdfk_volume_create(nx, ny, nz, min, max, colors3 ? 1 : 0, &v);
    if (r) return r;
    r = sdfk_volume_upload(v, values, colors3);
    if (!r) r = sdfk_march(v, iso_value, step, out);
    if (!r && (*out)->pending && (*out)->src == v) (*out)->owns_src = true;   // freed when the mesh is resolved
    else sdfk_volume_free(v);
    return r;
}

extern "C" int sdfk_sample_march(const sdfk_program* p, const float min[3], const float max[3],
                                 int32_t nx, int32_t ny, int32_t nz, int32_t clip_to_bounds,
                                 float iso_value, int32_t step, sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || !out) return fail(SDFK_ERR_INVALID, "sdfk_sample_march: null argument");
    if (int r = require_init()) return r;
    // self-contained job (no input but the program; the output is only read after a host-side
    // wait): consecutive calls alternate between the side lanes and overlap on the GPU
    int lane = g.cur_lane;   // (inside sdfk_lane_begin/end: the caller's lane)
    // SDFK_OPT_LANES, default three: small grids are launch-latency bound (9 dependent launches; 256^3: 39 instead of 48 us
    // per step), and from 512^3 up a third job in flight is worth 4-5 % since the sampler of a .W-only program holds no LDS
    // any more and the meshing kernels of two other jobs fit next to it (512^3 sphere 0.172 -> 0.164 ms, 384^3 0.092 ->
    // 0.087, 1024^3 1.23 -> 1.18; colour scenes unchanged)
    g.side_lanes = std::max(0, std::min(Context::NSIDE, g_cfg.lanes));
    if (g.side_lanes == 1) g.side_lanes = 0;   // one side lane overlaps nothing
    // SDFK_OPT_IDLE_LANE: a fourth lane for launch-bound grids (the captured-graph jobs) while the caller's stream has nothing
    // queued -- the fourth lane's stream sits in the class of the caller's stream ("stream placement": they must not be busy
    // together), which is the one class a job can use when the caller itself is not using it: 256^3 35 instead of 39 us per step
    // ... and for volume-less jobs (no sampling kernel: the same kind of chain of short dependent launches; 512^3 sphere 0.0870 ->
    // 0.0846 ms per pipelined step with the fourth lane, while a job that STORES its volume loses 0.6 % to it)
    const bool chain_only = step == 1 && g_cfg.elide_volume && g_cfg.corner_eval && g_cfg.vcolor_eval && !p->no_elide && iso_value == iso_value &&
                            !g.sampler_only;
    if (g.side_lanes == 3 && g_cfg.idle_lane && Context::NSIDE >= 4 && g.cur_lane == 0 && g.placed && step == 1 &&
        (graphs_enabled((int64_t)nx * ny * nz) || chain_only)) {
        bool placed4 = false;
        for (const auto& q : g.pool) placed4 = placed4 || (q.user == 4 && q.s == g.lanes[4].stream);
        if (placed4) {
            if (hipStreamQuery(g.lanes[0].stream) == hipSuccess) g.side_lanes = 4;
            else (void)hipGetLastError();   // (hipErrorNotReady is an answer, not an error)
        }
    }
    if (g.side_lanes > 0 && g.cur_lane == 0) {
        if (g.next_side >= g.side_lanes) g.next_side = 0;
        lane = 1 + g.next_side;
        g.next_side = (g.next_side + 1) % g.side_lanes;
    }
    LaneScope on_lane(lane);
    *out = nullptr;
    if (step == 1 && graphs_enabled((int64_t)nx * ny * nz)) {   // launch-bound grids: the whole job as one captured graph
        if (int r = graph_sample_march(p, min, max, nx, ny, nz, clip_to_bounds ? 1 : 0, iso_value, out)) return r;
        if (*out) return SDFK_OK;
    }
    sdfk_volume* v = nullptr;
    // (SDFK_OPT_ELIDE_VOLUME applies to grids above the captured-graph limit: a launch-bound grid gains nothing from one launch more
    // and 4 bytes per voxel less, and its captured job -- built on the second sighting of a key -- stores its volume: one kernel
    // set per program structure either way)
    const bool elidable = step == 1 && !graphs_enabled((int64_t)nx * ny * nz);
    int r = elidable ? job_volume_create(p, nx, ny, nz, min, max, iso_value, &v) : sdfk_volume_create(nx, ny, nz, min, max, p->writes_color ? 1 : 0, &v);
    if (r) return r;
    r = require_init();
    {
        // SDFK_OPT_TOKENS -- bit 0: sampling kernels apart, bit 1: k_vertices apart.  Default: the sampling kernels of grids from
        // 2^27 voxels up (512^3 sphere 0.162 -> 0.156 ms per step, 768^3 0.475 -> 0.462, 1024^3 1.12-1.20 -> 1.10, README scene
        // 0.49 -> 0.478; 384^3 and below lose 2-3 %: there a sampling kernel is too short to be worth a cross-stream wait).
        // k_vertices apart costs 2-3 % at every size: its workgroups are long-lived, and a second one fills the first one's tail.
        const int dflt = (int64_t)nx * ny * nz >= (int64_t(1) << 27) ? 1 : 0;
        g.token_mask = (lane > 0 && g.side_lanes > 1) ? (g_cfg.tokens >= 0 ? g_cfg.tokens : dflt) : 0;
    }
    if (!r) r = sample_impl(p, v, clip_to_bounds, step == 1 ? iso_value : 0.0f);
    if (!r) r = sdfk_march(v, iso_value, step, out);
    g.token_mask = 0;
    if (!r && (*out)->pending && (*out)->src == v) (*out)->owns_src = true;   // freed when the mesh is resolved
    else sdfk_volume_free(v);
    return r;
}

// ---------------------------------------------------------------------------
// SdfEx.Sample (Sdf.cs:22-47): the SDF at arbitrary points
// ---------------------------------------------------------------------------
static int eval_points_launch(const sdfk_program* p, const float* points_dev, int64_t n, float* rgbw_dev)
{
    struct { const float* points; float* rgbw; long n; } A{points_dev, rgbw_dev, (long)n};   // (= PointArgs of sample_codegen.h)
    hipFunction_t fn = nullptr;
    if (int r = program_fn(p, PK_POINTS, &fn)) return r;
    void* params[] = {&A, p->kargs()};
    ProfScope ps("sdfk_eval_points");
    HIPCHK(hipModuleLaunchKernel(fn, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, g.stream, params, nullptr));
    return SDFK_OK;
}

extern "C" int sdfk_eval_points_device(const sdfk_program* p, const void* points3_dev, int64_t n, void* rgbw4_dev)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || n < 0 || (n > 0 && (!points3_dev || !rgbw4_dev))) return fail(SDFK_ERR_INVALID, "sdfk_eval_points: null / negative argument");
    if (n >= (int64_t(1) << 31) * 256) return fail(SDFK_ERR_INVALID, "sdfk_eval_points: too many points");
    if (int r = require_init()) return r;
    if (n == 0) return SDFK_OK;
    return eval_points_launch(p, (const float*)points3_dev, n, (float*)rgbw4_dev);
}

extern "C" int sdfk_eval_points(const sdfk_program* p, const float* points3, int64_t n, float* rgbw4)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || n < 0 || (n > 0 && (!points3 || !rgbw4))) return fail(SDFK_ERR_INVALID, "sdfk_eval_points: null / negative argument");
    if (n >= (int64_t(1) << 31) * 256) return fail(SDFK_ERR_INVALID, "sdfk_eval_points: too many points");
    if (int r = require_init()) return r;
    if (n == 0) return SDFK_OK;
    float* pd = nullptr;
    float* od = nullptr;
    int r = dev_alloc((void**)&pd, (size_t)n * 3 * sizeof(float));
    if (!r) r = dev_alloc((void**)&od, (size_t)n * 4 * sizeof(float));
    hipError_t e = hipSuccess;
    if (!r) e = hipMemcpyAsync(pd, points3, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, g.stream);
    // (a program that only assigns .W leaves X, Y, Z of the caller's elements alone, as the reference's delegates do: they travel there and back)
    if (!r && e == hipSuccess && !p->writes_color) e = hipMemcpyAsync(od, rgbw4, (size_t)n * 4 * sizeof(float), hipMemcpyHostToDevice, g.stream);
    if (!r && e == hipSuccess) r = eval_points_launch(p, pd, n, od);
    if (!r && e == hipSuccess) e = hipMemcpyAsync(rgbw4, od, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost, g.stream);
    const hipError_t es = hipStreamSynchronize(g.stream);   // (the caller's arrays are not retained; the pool is stream-ordered)
    dev_free(pd);
    dev_free(od);
    if (r) return r;
    if (e != hipSuccess || es != hipSuccess) return fail(SDFK_ERR_HIP, "sdfk_eval_points: %s", hipGetErrorString(e != hipSuccess ? e : es));
    return SDFK_OK;
}

// ---------------------------------------------------------------------------
// RayMarcher
// ---------------------------------------------------------------------------
static int raymarch_launch(const sdfk_program* p, int32_t width, int32_t height, const float cam[3], const float vpi[16],
                           float nearp, float farp, int32_t iters, float* depth_dev, float* rgb_dev)
{
    RayArgs A;
    memset(&A, 0, sizeof A);
    A.depth = depth_dev; A.rgb = rgb_dev;
    memcpy(A.cam, cam, sizeof A.cam);
    memcpy(A.m, vpi, sizeof A.m);
    A.width = width; A.height = height; A.nearp = nearp; A.farp = farp; A.iters = iters;
    void* params[] = {&A, p->kargs()};
    const size_t n = (size_t)width * height;
    ProfScope ps("sdfk_raymarch");
    hipFunction_t fn_raymarch = nullptr;
    if (int r = program_fn(p, PK_RAYMARCH, &fn_raymarch)) return r;
    HIPCHK(hipModuleLaunchKernel(fn_raymarch, (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, g.stream, params, nullptr));
    return SDFK_OK;
}

extern "C" int sdfk_raymarch_device(const sdfk_program* p, int32_t width, int32_t height, const float camera_position[3],
                                    const float view_projection_inverse[16], float near_plane, float far_plane,
                                    int32_t depth_iterations, void* depth_dev, void* rgb_dev)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || !camera_position || !view_projection_inverse) return fail(SDFK_ERR_INVALID, "sdfk_raymarch: null argument");
    if (width < 1 || height < 1 || depth_iterations < 0 || (size_t)width * height > (size_t(1) << 31))
        return fail(SDFK_ERR_INVALID, "sdfk_raymarch: bad image size %d x %d or iteration count %d", width, height, depth_iterations);
    if (int r = require_init()) return r;
    if (!depth_dev && !rgb_dev) return SDFK_OK;
    return raymarch_launch(p, width, height, camera_position, view_projection_inverse, near_plane, far_plane, depth_iterations,
                           (float*)depth_dev, (float*)rgb_dev);
}

extern "C" int sdfk_raymarch(const sdfk_program* p, int32_t width, int32_t height, const float camera_position[3],
                             const float view_projection_inverse[16], float near_plane, float far_plane,
                             int32_t depth_iterations, float* depth, float* rgb)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || !camera_position || !view_projection_inverse) return fail(SDFK_ERR_INVALID, "sdfk_raymarch: null argument");
    if (width < 1 || height < 1 || depth_iterations < 0 || (size_t)width * height > (size_t(1) << 31))
        return fail(SDFK_ERR_INVALID, "sdfk_raymarch: bad image size %d x %d or iteration count %d", width, height, depth_iterations);
    if (int r = require_init()) return r;
    if (!depth && !rgb) return SDFK_OK;
    const size_t n = (size_t)width * height;
    float* d = nullptr;
    float* c = nullptr;
    int r = SDFK_OK;
    if (depth) r = dev_alloc((void**)&d, n * sizeof(float));
    if (!r && rgb) r = dev_alloc((void**)&c, n * 3 * sizeof(float));
    if (!r) r = raymarch_launch(p, width, height, camera_position, view_projection_inverse, near_plane, far_plane, depth_iterations, d, c);
    if (!r && depth && hipMemcpyAsync(depth, d, n * sizeof(float), hipMemcpyDeviceToHost, g.stream) != hipSuccess) r = fail(SDFK_ERR_HIP, "copy of the depth image failed");
    if (!r && rgb && hipMemcpyAsync(rgb, c, n * 3 * sizeof(float), hipMemcpyDeviceToHost, g.stream) != hipSuccess) r = fail(SDFK_ERR_HIP, "copy of the colour image failed");
    if (hipStreamSynchronize(g.stream) != hipSuccess && !r) r = fail(SDFK_ERR_HIP, "hipStreamSynchronize failed");
    dev_free(d);
    dev_free(c);
    return r;
}

// ---------------------------------------------------------------------------
// pinned host arena
// ---------------------------------------------------------------------------
extern "C" int sdfk_host_alloc(int64_t n_bytes, void** out)
{
    if (!out || n_bytes < 0) return fail(SDFK_ERR_INVALID, "sdfk_host_alloc: bad argument");
    *out = nullptr;
    // (the arena is process-wide and pinned memory belongs to no device context: ANY initialised context of the process will do --
    // also a node's private ones, for a host whose only use of the library is sdfk_node_* and that never called sdfk_init itself)
    if (g_contexts_up.load() <= 0) {
        std::lock_guard<std::recursive_mutex> lk(g_mu);
        if (int r = require_init()) return r;
    }
    const size_t c = size_class((size_t)std::max<int64_t>(n_bytes, 1));
    std::lock_guard<std::mutex> al(g_arena.mu);
    // (the block of this size class that was freed LAST: equal keys keep their insertion order, so it is the one before the upper
    // bound -- its pages are the likeliest to be in the host's caches and TLBs)
    auto it = g_arena.free_blocks.upper_bound(c);
    void* p = nullptr;
    if (it != g_arena.free_blocks.begin() && std::prev(it)->first == c) {
        --it;
        p = it->second;
        g_arena.free_blocks.erase(it);
    } else if (hipHostMalloc(&p, c, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        for (auto& kv : g_arena.free_blocks) (void)hipHostFree(kv.second);   // drop the cache and retry once
        g_arena.free_blocks.clear();
        if (hipHostMalloc(&p, c, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return fail(SDFK_ERR_NOMEM, "hipHostMalloc(%zu) failed", c); }
    }
    g_arena.live[p] = c;
    *out = p;
    return SDFK_OK;
}

extern "C" void sdfk_host_free(void* p)
{
    if (!p) return;
    std::lock_guard<std::mutex> al(g_arena.mu);
    auto it = g_arena.live.find(p);
    if (it == g_arena.live.end()) return;
    g_arena.free_blocks.emplace(it->second, p);   // back to the arena
    g_arena.live.erase(it);
}

// Makes [p, p + n_bytes) of the caller's (pageable) memory present and writable on the library's thread pool: what
// sdfk_mesh_copy / sdfk_volume_download do to their destinations anyway, offered separately so that a host can do it WHILE
// the GPU is still computing the mesh (sdfk_mesh_size_hint tells how large the arrays will be).
// Phases of the last staged device -> pageable-host copy (SDFK_OPT_COPY_MODE 1): stats[5] = { bytes, ns until every chunk was
// queued, ns until the destination pages were present, ns until done, ns of that spent waiting for the DMA } (measurement).
extern "C" int sdfk_stream_placement(int32_t out[8])
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!out) return fail(SDFK_ERR_INVALID, "sdfk_stream_placement: null argument");
    if (int r = require_init()) return r;
    out[0] = g.placed && !g.pool.empty() ? 1 : 0;
    out[1] = g.n_classes;
    out[2] = g.cls_lane0;
    for (int k = 1; k <= 4; k++) {
        out[2 + k] = -1;
        for (const auto& q : g.pool)
            if (k <= Context::NSIDE && q.s == g.lanes[k].stream) out[2 + k] = q.cls;
    }
    out[7] = -1;
    for (const auto& q : g.pool)
        if (q.user == 100) out[7] = q.cls;
    return SDFK_OK;
}

extern "C" int sdfk_copy_stats(int64_t stats[5])
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!stats) return fail(SDFK_ERR_INVALID, "sdfk_copy_stats: null argument");
    memcpy(stats, g_copy_stats, sizeof g_copy_stats);
    return SDFK_OK;
}

extern "C" int sdfk_host_prefault(void* p, int64_t n_bytes)
{
    if (n_bytes < 0) return fail(SDFK_ERR_INVALID, "sdfk_host_prefault: bad size");
    if (!p || n_bytes == 0) return SDFK_OK;
    std::lock_guard<std::recursive_mutex> pool_lk(g_pool_mu);
    config_from_env();
    prefault_start(p, (size_t)n_bytes);
    g_pool.wait();
    return SDFK_OK;
}

// ---------------------------------------------------------------------------
// meshes
// ---------------------------------------------------------------------------
// How large will the mesh be?  Without waiting: for a mesh whose job is still queued, the sizes of the previous mesh of the
// same grid shape (what its buffers were sized from; exact whenever the scene repeats) -- the host can allocate and
// pre-fault its arrays while the GPU works and only re-allocates if sdfk_mesh_counts says otherwise; for a finished mesh,
// its counts.  *exact = 1 in the second case.
extern "C" int sdfk_mesh_size_hint(const sdfk_mesh* m, int64_t* n_vertices, int64_t* n_indices, int32_t* exact)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!m) return fail(SDFK_ERR_INVALID, "null mesh");
    int64_t nv = m->nv, ni = m->ni;
    int ex = 1;
    if (m->pending && !m->status) {
        ex = 0;
        auto it = g.hints.find(m->key);
        if (it != g.hints.end()) { nv = it->second.nv; ni = it->second.ni; }
        else { nv = (int64_t)m->cap_v; ni = (int64_t)m->cap_i; }
    }
    if (n_vertices) *n_vertices = nv;
    if (n_indices) *n_indices = ni;
    if (exact) *exact = ex;
    return SDFK_OK;
}

extern "C" int sdfk_mesh_counts(const sdfk_mesh* m, int64_t* n_vertices, int64_t* n_indices)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!m) return fail(SDFK_ERR_INVALID, "null mesh");
    if (int r = mesh_resolve(const_cast<sdfk_mesh*>(m))) return r;
    if (n_vertices) *n_vertices = m->nv;
    if (n_indices) *n_indices = m->ni;
    return SDFK_OK;
}

extern "C" int sdfk_mesh_stats(const sdfk_mesh* m, int64_t* n_active_cells, int64_t* n_case13_cells)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!m) return fail(SDFK_ERR_INVALID, "null mesh");
    if (int r = mesh_resolve(const_cast<sdfk_mesh*>(m))) return r;
    if (n_active_cells) *n_active_cells = m->n_active;
    if (n_case13_cells) *n_case13_cells = m->n_case13;
    return SDFK_OK;
}

extern "C" int sdfk_mesh_bounds(const sdfk_mesh* mc, float min[3], float max[3])
{
    StateScope in_owner_context(mc ? mc->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    sdfk_mesh* m = const_cast<sdfk_mesh*>(mc);
    if (!m) return fail(SDFK_ERR_INVALID, "null mesh");
    if (int r = require_init()) return r;
    if (int r = mesh_resolve(m)) return r;
    if (!m->bounds_valid) {
        float hb[6];
        HIPCHK(hipMemcpyAsync(hb, m->bounds, 6 * sizeof(float), hipMemcpyDeviceToHost, g.stream));
        HIPCHK(hipStreamSynchronize(g.stream));
        if (m->nv > 0) { memcpy(m->h_min, hb, 12); memcpy(m->h_max, hb + 3, 12); }
        m->bounds_valid = true;
    }
    if (min) memcpy(min, m->h_min, 12);
    if (max) memcpy(max, m->h_max, 12);
    return SDFK_OK;
}

extern "C" int sdfk_mesh_copy(const sdfk_mesh* m, float* vertices3, float* colors3, float* normals3, int32_t* triangles)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!m) return fail(SDFK_ERR_INVALID, "null mesh");
    if (int r = require_init()) return r;
    if (int r = mesh_resolve(const_cast<sdfk_mesh*>(m))) return r;
    const size_t vb = (size_t)m->nv * 3 * sizeof(float);
    std::vector<CopyPiece> pieces;
    if (vertices3 && vb) pieces.push_back({m->vertices, vertices3, vb});
    if (colors3 && vb) {
        // (a volume without colours has all-zero mesh colours: nothing to move, the pool clears the array)
        if (m->has_colors) pieces.push_back({m->colors, colors3, vb});
    }
    if (normals3 && vb) pieces.push_back({m->normals, normals3, vb});
    if (triangles && m->ni) pieces.push_back({m->triangles, triangles, (size_t)m->ni * sizeof(int32_t)});
    if (m->lane != g.cur_lane) const_cast<sdfk_mesh*>(m)->used_on_main = true;
    std::function<void()> clear_colors;
    if (colors3 && vb && !m->has_colors)   // zero-fill (and first touch) on the pool, beside the transfers of the other arrays
        clear_colors = [=]() {
            std::lock_guard<std::recursive_mutex> pool_lk(g_pool_mu);
            const size_t per = size_t(2) << 20, nt = (vb + per - 1) / per;
            char* c = (char*)colors3;
            g_pool.start((int)nt, [=](int t) { const size_t a = (size_t)t * per; memset(c + a, 0, std::min(per, vb - a)); });
            g_pool.wait();
        };
    return copy_to_host(pieces, clear_colors);
}

extern "C" int sdfk_mesh_copy_device(const sdfk_mesh* m, void* vertices3, void* colors3, void* normals3, void* triangles)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!m) return fail(SDFK_ERR_INVALID, "null mesh");
    if (int r = require_init()) return r;
    if (int r = mesh_resolve(const_cast<sdfk_mesh*>(m))) return r;
    const_cast<sdfk_mesh*>(m)->used_on_main = true;
    const size_t vb = (size_t)m->nv * 3 * sizeof(float);
    if (vertices3 && vb) HIPCHK(hipMemcpyAsync(vertices3, m->vertices, vb, hipMemcpyDeviceToDevice, g.stream));
    if (colors3 && vb) {
        if (m->has_colors || m->colors_valid) HIPCHK(hipMemcpyAsync(colors3, m->colors, vb, hipMemcpyDeviceToDevice, g.stream));
        else HIPCHK(hipMemsetAsync(colors3, 0, vb, g.stream));   // (all zero, never stored: colors_valid)
    }
    if (normals3 && vb) HIPCHK(hipMemcpyAsync(normals3, m->normals, vb, hipMemcpyDeviceToDevice, g.stream));
    if (triangles && m->ni) HIPCHK(hipMemcpyAsync(triangles, m->triangles, (size_t)m->ni * sizeof(int32_t), hipMemcpyDeviceToDevice, g.stream));
    return SDFK_OK;
}

extern "C" int sdfk_mesh_device_ptrs(const sdfk_mesh* m, void** vertices3, void** colors3, void** normals3, void** triangles)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!m) return fail(SDFK_ERR_INVALID, "null mesh");
    if (int r = mesh_resolve(const_cast<sdfk_mesh*>(m))) return r;   // (the buffers may be replaced by an exact re-run)
    const_cast<sdfk_mesh*>(m)->used_on_main = true;
    if (colors3 && !m->has_colors && !m->colors_valid && m->colors && m->nv > 0) {   // the all-zero colours were never stored: now they are asked for
        if (int r = require_init()) return r;
        HIPCHK(hipMemsetAsync(m->colors, 0, (size_t)m->nv * 3 * sizeof(float), g.stream));
        const_cast<sdfk_mesh*>(m)->colors_valid = true;
    }
    if (vertices3) *vertices3 = m->vertices;
    if (colors3) *colors3 = m->colors;
    if (normals3) *normals3 = m->normals;
    if (triangles) *triangles = m->triangles;
    return SDFK_OK;
}

// Mesh.Transform(Matrix4x4) (Mesh.cs:47-64) on the device-resident mesh, in place.
extern "C" int sdfk_mesh_transform(sdfk_mesh* m, const float matrix[16], const float normal_matrix[16])
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!m || !matrix || !normal_matrix) return fail(SDFK_ERR_INVALID, "sdfk_mesh_transform: null argument");
    if (int r = require_init()) return r;
    if (int r = mesh_resolve(m)) return r;
    // (a mesh whose arrays are a captured job's -- every repeat sdfk_sample_march of a launch-bound grid -- is transformed in
    // place like any other: the job stays busy, its buffers untouched by later launches, until this handle is freed)
    if (m->external) return fail(SDFK_ERR_UNSUPPORTED, "sdfk_mesh_transform: the mesh arrays are sections of a slab payload (extract the mesh with sdfk_dist_mesh first)");
    if (m->nv == 0) return SDFK_OK;   // (Mesh.Measure leaves Min / Max alone, Mesh.cs:32)
    if (m->lane != g.cur_lane) m->used_on_main = true;
    const int grid = grid_for((size_t)m->nv, 256, 1024);
    float* partial = nullptr;
    if (int r = dev_alloc((void**)&partial, (size_t)grid * 6 * sizeof(float))) return r;
    XformArgs A;
    A.vertices = m->vertices; A.normals = m->normals; A.n = m->nv; A.partial = partial;
    memcpy(A.m, matrix, sizeof A.m);
    memcpy(A.nm, normal_matrix, sizeof A.nm);
    hipLaunchKernelGGL(k_mesh_transform, dim3(grid), dim3(256), 0, g.stream, A);
    hipLaunchKernelGGL(k_bounds_reduce, dim3(1), dim3(256), 0, g.stream, (const float*)partial, grid, m->bounds);
    const hipError_t e = hipGetLastError();
    dev_free(partial);   // (stream-ordered)
    if (e != hipSuccess) return fail(SDFK_ERR_HIP, "sdfk_mesh_transform: %s", hipGetErrorString(e));
    m->bounds_valid = false;   // sdfk_mesh_bounds reads the new AABB from the device
    return SDFK_OK;
}

extern "C" void sdfk_mesh_free(sdfk_mesh* m)
{
    StateScope in_owner_context(m ? m->owner : nullptr);
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!m) return;
    if (m->pending) {   // never read: drop the queued job's workspace (stream-ordered, no wait)
        for (auto it = g.pending.begin(); it != g.pending.end(); ++it)
            if (*it == m) { g.pending.erase(it); break; }
        if (!m->graph_job) {
            job_release(m->pending, true);   // its kernels may still be queued (and will still write the result slot)
            delete m->pending;
        }
        m->pending = nullptr;
        if (m->done) (void)hipEventDestroy(m->done);
    }
    if (m->graph_job) { m->src = nullptr; graph_job_retire(m, false); }   // (the GraphJob is free for the next call on its lane: stream order protects its buffers)
    drop_source(m);
    if (m->lane != 0 && m->used_on_main && g.inited) {
        // lane-0 work may still read the buffers: their lane must not reuse them before that
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess) {
            (void)hipEventRecord(ev, g.lanes[0].stream);
            (void)hipStreamWaitEvent(lane_stream(m->lane), ev, 0);
            (void)hipEventDestroy(ev);
        }
    }
    free_mesh_buffers(m);
    delete m;
}

// ---------------------------------------------------------------------------
// measurement hooks
// ---------------------------------------------------------------------------
extern "C" int sdfk_profile_enable(int32_t on)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (int r = require_init()) return r;
    if (!on) prof_drain();
    g.prof_on = on == 1;
    g.sampler_only = on == 2;
    return SDFK_OK;
}

extern "C" int sdfk_profile_reset(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    prof_drain();
    std::fill(g.prof_ms.begin(), g.prof_ms.end(), 0.0);
    std::fill(g.prof_n.begin(), g.prof_n.end(), 0);
    return SDFK_OK;
}

extern "C" int sdfk_profile_count(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    prof_drain();
    return (int)g.prof_names.size();
}

extern "C" int sdfk_profile_get(int32_t i, const char** name, double* total_ms, int64_t* launches)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    prof_drain();
    if (i < 0 || i >= (int)g.prof_names.size()) return fail(SDFK_ERR_INVALID, "profile index out of range");
    if (name) *name = g.prof_names[i].c_str();
    if (total_ms) *total_ms = g.prof_ms[i];
    if (launches) *launches = g.prof_n[i];
    return SDFK_OK;
}

// ---------------------------------------------------------------------------
// Z-slab sharding: sdfk_dist_* (RCCL called by the library itself)
// ---------------------------------------------------------------------------
#include "dist_rccl.h"
#include "node_local.h"
