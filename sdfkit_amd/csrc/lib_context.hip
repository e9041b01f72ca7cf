// lib_context.hip -- errors, configuration, device contexts (streams, lanes, pools, phase tokens, stream placement), options, the pinned
// host arena, the copy helpers and the measurement hooks of libsdfkit_hip.so.
#include "lib_internal.h"

// ---- the process-wide state declared in lib_internal.h ----------------------------------------------------------------------------
thread_local std::string t_err;
Config g_cfg;
DeviceState g_state0;
std::atomic<DeviceState*> g_default_state{&g_state0};
std::mutex g_registry_mu;
std::atomic<int> g_contexts_up{0};
std::vector<DeviceState*> g_states{&g_state0};
thread_local DeviceState* t_state = nullptr;
HostArena g_arena;
HostPool g_pool;
std::recursive_mutex g_pool_mu;   // one client at a time (the pool is shared by the device contexts of the process)

int fail(int code, const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    t_err = buf;
    return code;
}

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

int grid_for(size_t work_items, int per_block, int max_blocks)
{
    size_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > (size_t)max_blocks) b = max_blocks;
    return (int)b;
}



// ---------------------------------------------------------------------------
// host-side copy helpers (sdfk_mesh_copy / sdfk_volume_download)
// ---------------------------------------------------------------------------
// What a managed caller hands over are freshly allocated, pageable arrays (Mesh.cs:10-13 are
// `new Vector3[n]` / `new int[n]`; the Python mirror's are numpy.empty): a device-to-host copy into
// them is dominated by first-touch page faults on ONE thread (512^3 sphere, 33 MB: 9 ms, against
// 0.65 ms into memory that has been touched).  Faults scale with threads, so a small persistent
// pool touches the destination pages (one write per page: the whole range is overwritten right
// after) while the previous array is still on the wire.


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

int64_t g_copy_stats[5] = {0, 0, 0, 0, 0};   // last staged copy: bytes, ns until queued / destination present / done, ns waiting for the DMA

// Device -> caller arrays.  Default (mode 1): everything goes to the pinned staging buffer in 4 MiB
// chunks (plain DMA at the link rate) and the pool copies the chunks that have arrived into the
// destination, which it has pre-faulted while the first chunks were on the wire.  The caller's
// pageable memory is never handed to the HIP runtime: its pin-on-the-fly path measured anywhere
// between 0.8 and 26 ms for the same 33 MB, depending on page size and history (tools/host_io_probe.py).
// mode 0: pre-fault on the pool, then the runtime's own copy; mode 2: the runtime's copy alone.
// `beside`: host work that does not touch the destinations (clearing the colour array of a mesh without colours): done while the
// DMA transfers are in flight where the copy is the runtime's own (resident / pinned destinations), before the copy otherwise
int copy_to_host(const std::vector<CopyPiece>& pieces, const std::function<void()>& beside)
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


// ---------------------------------------------------------------------------
// lifetime
// ---------------------------------------------------------------------------
extern "C" int sdfk_abi_version(void) { return SDFK_ABI_VERSION; }

extern "C" const char* sdfk_last_error(void) { return t_err.c_str(); }

// The ONE place the environment is read (see Config): start-up defaults of the options, the cache location, debugging aids.
void config_from_env_once();
void config_from_env()
{
    static std::once_flag once;
    std::call_once(once, config_from_env_once);
}
void config_from_env_once()
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
void codes_trim();
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
