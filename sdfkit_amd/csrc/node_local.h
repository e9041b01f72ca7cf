// node_local.h -- several GPUs from ONE process (include/sdfkit_hip.h, "one process, several GPUs"): sdfk_node_*.
//
// The reference is a library that a single .NET process calls (Sdf.cs:59-63: `sdf.ToMesh(...)`); the Z-slab sharded step of
// dist_rccl.h wants one RANK per GPU.  A node gives every GPU a device context of its own (lib_internal.h, "device contexts") and a
// host THREAD of the library's own that is the rank: the threads join one RCCL communicator per device (ncclCommInitRank on a shared
// id, exactly what one process per GPU does) and run the very same sharded step -- sdfk_dist_session_* -- side by side; the calling
// thread posts a command and gets rank 0's whole mesh back as an ordinary sdfk_mesh (its accessors work from any thread: the mesh
// remembers its context).  Nothing else is new: partition, exchange, rebase and extraction are dist_rccl.h's.
// Ranks that share a device (a device listed twice: tests on a one-GPU box, RCCL refuses two ranks on one device) exchange through
// host memory between the threads (the library's host transport, sdfk_dist_init_host).
// Included by lib_dist.hip after dist_rccl.h.
#pragma once
#ifndef SDFK_LIB_DIST_TU
#error "node_local.h holds definitions: it is part of lib_dist.hip, not a header to include elsewhere (declarations: lib_internal.h)"
#endif
#include <condition_variable>
#include <thread>


// A barrier that can be ABORTED: a rank thread whose command failed in a rank-local spot never reaches the next rendezvous of that
// command; it aborts the barrier, which releases everybody who waits there (and everybody who arrives later) with `false` -- they
// fail the command too instead of waiting for ever.  The poster re-arms it before the next command, when every worker is idle.
struct NodeBarrier {
    std::mutex m;
    std::condition_variable cv;
    int n = 1, count = 0;
    uint64_t gen = 0;
    bool aborted = false;
    bool wait()
    {
        std::unique_lock<std::mutex> lk(m);
        if (aborted) return false;
        const uint64_t my = gen;
        if (++count == n) { count = 0; gen++; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return gen != my || aborted; });
        return gen != my;   // (released by the last arrival, not by an abort)
    }
    void abort()
    {
        std::lock_guard<std::mutex> lk(m);
        aborted = true;
        cv.notify_all();
    }
    void rearm()
    {
        std::lock_guard<std::mutex> lk(m);
        aborted = false;
        count = 0;
    }
};


struct sdfk_node {
    struct Worker {
        sdfk_node* node = nullptr;
        std::thread th;
        int device = 0, rank = 0;
        DeviceState* st = nullptr;
        // the scene the rank holds a session for (a host that meshes frame after frame asks for the same one again)
        std::string key;
        sdfk_program* prog = nullptr;
        sdfk_dist_session* sess = nullptr;
        int status = SDFK_OK;
        std::string error;
        bool released = false;   // the status is "another rank failed and released me": the caller is told about THAT rank first
        std::string transport_error;   // what the host transport between the rank threads had to say (its caller only knows "it failed")
    };
    int world = 1;
    bool host_transport = false;
    std::vector<Worker> workers;
    std::mutex mu;
    std::condition_variable cv_cmd, cv_done;
    uint64_t cmd_gen = 0;
    int cmd = 0;      // 1: to_mesh, 2: stop
    int done = 0;
    // command
    std::vector<sdfk_op> ops;
    int32_t out_rgbw[4] = {0, 0, 0, 0};
    int32_t writes_color = 0, nx = 0, ny = 0, nz = 0, clip = 0;
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0}, iso = 0.0f;
    sdfk_mesh* mesh0 = nullptr;
    // host-array form (sdfk_node_mesh_begin / sdfk_node_mesh_copy): totals of the step in flight, the caller's arrays
    int64_t total_v = 0, total_i = 0;
    int32_t has_colors = 0;
    float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
    bool step_open = false;
    float *dst_v = nullptr, *dst_c = nullptr, *dst_n = nullptr;
    int32_t* dst_t = nullptr;
    // start-up rendezvous
    unsigned char uid[SDFK_DIST_ID_BYTES] = {};
    NodeBarrier bar;
    std::vector<const void*> send_ptrs;   // host transport: every rank's send buffer of the all-gather in progress
    std::mutex call_mu;                   // one sdfk_node_to_mesh at a time
    // fault injection for the tests (SDFK_NODE_FAULT_RANK=k at sdfk_node_open): rank k fails the next command's rank-local
    // part ONCE -- what an allocation failing on one device looks like to the others.  SDFK_NODE_FAULT_STAGE says where:
    // 0 (default) while it makes its session, 1 in the rank-local part of the step (the protocol's consensus point between
    // enqueue / run_exact and the first collective), 2 inside the exchange of the host transport, between two rendezvous of the
    // ranks (nobody can agree there: the failed rank aborts the barrier and the others are released)
    std::atomic<int> fault_rank{-1};
    int fault_stage = 0;
};



// the host transport between the threads of a node: everybody publishes its send buffer, then copies everybody's
int node_allgather(void* ctx, const void* send, void* recv, int64_t bytes)
{
    sdfk_node::Worker* w = (sdfk_node::Worker*)ctx;
    sdfk_node* n = w->node;
    if (n->fault_stage == 2 && n->fault_rank.load() == w->rank) {   // (injected: this rank never reaches the rendezvous)
        n->fault_rank.store(-1);
        w->transport_error = "SDFK_NODE_FAULT_RANK: injected failure inside the exchange";
        return -1;
    }
    auto released = [&] {   // the barrier was aborted: a rank failed elsewhere in this command and will never arrive
        w->released = true;
        w->transport_error = "another rank failed and left the command";
        return -1;
    };
    n->send_ptrs[(size_t)w->rank] = send;
    if (!n->bar.wait()) return released();
    for (int q = 0; q < n->world; q++) memcpy((char*)recv + (size_t)q * (size_t)bytes, n->send_ptrs[(size_t)q], (size_t)bytes);
    if (!n->bar.wait()) return released();   // (nobody overwrites its send buffer before everybody has read it)
    return 0;
}

void node_worker_release(sdfk_node::Worker* w)
{
    if (w->sess) sdfk_dist_session_free(w->sess);
    w->sess = nullptr;
    if (w->prog) sdfk_program_destroy(w->prog);
    w->prog = nullptr;
    w->key.clear();
}

// the rank's session for the scene of the current command (kept while the scene stays the same)
int node_worker_session(sdfk_node::Worker* w);

// Before the ranks enter a COLLECTIVE part of a command they agree that every one of them got there: a rank-local failure (an
// allocation on ONE device) must fail the command on every rank -- the others would wait in the exchange for ever.  Returns the
// first failing rank's status (the same on every rank), its message in t_err.
int node_agree(sdfk_node::Worker* w, int mine, int stage = 0)
{
    sdfk_node* n = w->node;
    if (!mine && n->fault_stage == stage && n->fault_rank.load() == w->rank) {
        n->fault_rank.store(-1);
        mine = fail(SDFK_ERR_NOMEM, "SDFK_NODE_FAULT_RANK: injected failure%s", stage ? " in the rank-local part of the step" : "");
    }
    w->status = mine;
    w->error = mine ? t_err : std::string();
    w->released = false;
    auto released = [&] {   // the barrier was aborted: some rank left this command in a spot where nobody could agree
        w->released = !mine;
        return mine ? mine : fail(SDFK_ERR_HIP, "another rank failed and left the command");
    };
    if (!n->bar.wait()) return released();
    int r = SDFK_OK;
    for (auto& q : n->workers)
        if (q.status && !r) { r = q.status; if (&q != w) { t_err = "rank " + std::to_string(q.rank) + " failed: " + q.error; w->released = true; } }
    if (!n->bar.wait()) return released();   // (nobody changes its status before everybody has read it)
    return r;
}

// SlabOps::consensus of a node's rank (dist_rccl.h): the agreement between the rank-local part of a step and its first collective
int node_consensus(void* ctx, int mine) { return node_agree((sdfk_node::Worker*)ctx, mine, 1); }

int node_worker_to_mesh(sdfk_node::Worker* w, sdfk_mesh** out)
{
    *out = nullptr;
    if (int r = node_agree(w, node_worker_session(w))) return r;
    int r = sdfk_dist_submit(w->sess);
    if (!r) r = sdfk_dist_collect(w->sess, nullptr, nullptr);
    // (the node's sessions leave the mesh sharded -- exchange mode 3 --: asking for the WHOLE mesh gathers the payloads of this step,
    // which is collective; rank 0's copy is the one handed out)
    sdfk_mesh* m = nullptr;
    if (!r) r = sdfk_dist_mesh(w->sess, &m);
    if (w->rank == 0) *out = m;
    else if (m) sdfk_mesh_free(m);
    return r;
}

// host-array form, phase 1: the step; every rank then knows every slab's counts (the headers) -- rank 0 publishes the totals
int node_worker_begin(sdfk_node::Worker* w)
{
    sdfk_node* n = w->node;
    if (int r = node_agree(w, node_worker_session(w))) return r;
    int r = sdfk_dist_submit(w->sess);
    if (!r) r = sdfk_dist_collect(w->sess, nullptr, nullptr);
    if (r || w->rank != 0) return r;
    const int64_t* h = w->sess->proto.last_headers();
    if (!h) return fail(SDFK_ERR_INVALID, "sdfk_node_mesh_begin: no collected step");
    n->total_v = n->total_i = 0;
    bool any = false;
    for (int q = 0; q < n->world; q++) {
        const int64_t* hq = h + q * sdfk::kSlabHeaderWords;
        if (hq[0] < 0 || hq[1] < 0) return fail(SDFK_ERR_INVALID, "sdfk_node_mesh_begin: the step has an unresolved slab");
        n->total_v += hq[0];
        n->total_i += hq[1];
        const float* b = reinterpret_cast<const float*>(hq + 2);
        if (hq[0] > 0) {   // Mesh.Measure (Mesh.cs:30-45) over the slabs that have vertices
            for (int c = 0; c < 3; c++) {
                n->bmin[c] = any ? std::min(n->bmin[c], b[c]) : b[c];
                n->bmax[c] = any ? std::max(n->bmax[c], b[3 + c]) : b[3 + c];
            }
            any = true;
        }
    }
    if (!any) for (int c = 0; c < 3; c++) n->bmin[c] = n->bmax[c] = 0.0f;
    if (n->total_v >= (int64_t(1) << 31)) return fail(SDFK_ERR_UNSUPPORTED, "vertex index exceeds int32 (Mesh.Triangles is int[])");
    n->has_colors = w->sess->vbytes == 36;
    return SDFK_OK;
}

// phase 2: every rank copies ITS slab into its slice of the caller's arrays -- over its own PCIe link, all ranks at once, no exchange
// between the GPUs at all (the mesh stayed sharded)
int node_worker_copy(sdfk_node::Worker* w)
{
    sdfk_node* n = w->node;
    const int64_t* h = w->sess ? w->sess->proto.last_headers() : nullptr;
    if (!h) return fail(SDFK_ERR_INVALID, "sdfk_node_mesh_copy: no step to copy (call sdfk_node_mesh_begin first)");
    int64_t vb = 0, ib = 0;
    for (int q = 0; q < w->rank; q++) { vb += h[q * sdfk::kSlabHeaderWords]; ib += h[q * sdfk::kSlabHeaderWords + 1]; }
    sdfk_mesh* m = nullptr;
    if (int r = sdfk_dist_slab_mesh(w->sess, &m)) return r;
    const int r = sdfk_mesh_copy(m, n->dst_v ? n->dst_v + 3 * vb : nullptr, n->dst_c ? n->dst_c + 3 * vb : nullptr,
                                 n->dst_n ? n->dst_n + 3 * vb : nullptr, n->dst_t ? n->dst_t + ib : nullptr);
    sdfk_mesh_free(m);
    return r;
}

int node_worker_session(sdfk_node::Worker* w)
{
    sdfk_node* n = w->node;
    std::string key((const char*)n->ops.data(), n->ops.size() * sizeof(sdfk_op));
    key.append((const char*)n->out_rgbw, sizeof n->out_rgbw);
    const int32_t dims[5] = {n->writes_color, n->nx, n->ny, n->nz, n->clip};
    key.append((const char*)dims, sizeof dims);
    key.append((const char*)n->mn, 12);
    key.append((const char*)n->mx, 12);
    key.append((const char*)&n->iso, 4);
    if (key != w->key || !w->sess) {
        node_worker_release(w);
        if (int r = sdfk_program_create(n->ops.data(), (int32_t)n->ops.size(), n->out_rgbw, n->writes_color, &w->prog)) return r;
        if (int r = sdfk_dist_session_create(w->prog, n->mn, n->mx, n->nx, n->ny, n->nz, n->clip, n->iso, 1, &w->sess)) return r;
        w->sess->exchange_mode = 3;   // the mesh stays sharded: a step moves the headers only, payloads travel when (and where) they are asked for
        w->key = key;
    }
    return SDFK_OK;
}

void node_worker_main(sdfk_node::Worker* w)
{
    sdfk_node* n = w->node;
    // ---- start-up: a private context on the device, then the communicator (all ranks together)
    w->st = context_claim(w->device, false);
    t_state = w->st;
    w->status = context_init(w->device);
    if (w->status) w->error = t_err;
    n->bar.wait();                                   // (1) every context is up (or has failed)
    bool ok = true;
    for (auto& q : n->workers) ok = ok && q.status == SDFK_OK;
    if (ok && !n->host_transport && w->rank == 0) {
        w->status = sdfk_dist_unique_id(n->uid);
        if (w->status) w->error = t_err;
    }
    n->bar.wait();                                   // (2) the id is there
    ok = true;
    for (auto& q : n->workers) ok = ok && q.status == SDFK_OK;
    // the rank-local half of sdfk_dist_init first (the library, the exchange stream, the agreement buffers) ...
    bool prepared = false;
    if (ok) {
        w->status = n->host_transport ? sdfk_dist_init_host(n->world, w->rank, node_allgather, w) : dist_prepare_rccl(n->world, w->rank);
        if (w->status) w->error = t_err;
        prepared = !w->status;
        if (!w->status && n->fault_stage == 3 && n->fault_rank.load() == w->rank) {   // (injected: this rank's preparation fails)
            n->fault_rank.store(-1);
            w->status = fail(SDFK_ERR_NOMEM, "SDFK_NODE_FAULT_RANK: injected failure before ncclCommInitRank");
            w->error = t_err;
        }
    }
    n->bar.wait();                                   // (3) every rank is prepared (or has failed)
    ok = true;
    for (auto& q : n->workers) ok = ok && q.status == SDFK_OK;
    // ... and only when EVERY rank got that far the collective half: ncclCommInitRank returns when all ranks have made the call
    if (ok && !n->host_transport) {
        w->status = dist_join_rccl(n->uid);
        if (w->status) w->error = t_err;
    } else if (!ok && prepared) {
        std::lock_guard<std::recursive_mutex> lk(g_mu);
        dist_release();                              // (prepared, but the node will not come up)
    }
    if (ok && !w->status) { gd.consensus_fn = node_consensus; gd.consensus_ctx = w; }
    n->bar.wait();                                   // (4) the communicator is up (or somebody failed: the opener reads the statuses)
    ok = true;
    for (auto& q : n->workers) ok = ok && q.status == SDFK_OK;
    // ---- commands
    uint64_t seen = 0;
    for (;;) {
        int cmd;
        {
            std::unique_lock<std::mutex> lk(n->mu);
            n->cv_cmd.wait(lk, [&] { return n->cmd_gen != seen; });
            seen = n->cmd_gen;
            cmd = n->cmd;
        }
        if (cmd == 1 && ok) {
            sdfk_mesh* m = nullptr;
            w->released = false;
            w->transport_error.clear();
            w->status = node_worker_to_mesh(w, &m);
            w->error = w->status ? t_err : std::string();
            if (w->rank == 0) n->mesh0 = m;
            else if (m) sdfk_mesh_free(m);
        }
        if ((cmd == 3 || cmd == 4) && ok) {
            w->released = false;
            w->transport_error.clear();
            w->status = cmd == 3 ? node_worker_begin(w) : node_worker_copy(w);
            w->error = w->status ? t_err : std::string();
        }
        if (w->status && !w->transport_error.empty()) w->error += ": " + w->transport_error;
        // a rank that leaves a collective command with an error may have left it between two rendezvous: whoever waits for it there
        // (the host transport's all-gather, an agreement) is released and fails the command too
        if ((cmd == 1 || cmd == 3) && ok && w->status) {
            n->bar.abort();
            // (the session's protocol state no longer matches the other ranks': the next command makes a new one on every rank --
            // the ranks that were released drop theirs as well)
            node_worker_release(w);
        }
        if (cmd == 2) {
            node_worker_release(w);
            sdfk_dist_shutdown();
            bool up;
            { std::lock_guard<std::recursive_mutex> lk(w->st->mu); up = w->st->ctx.inited; }
            if (up) sdfk_shutdown();        // (this thread's context: frees everything it holds and gives the context back -- ONCE: a second
                                            // unclaim could hit a context some other thread has claimed in between)
            else context_unclaim(w->st);    // (the context never came up: sdfk_shutdown has nothing to do, the claim is still ours)
            t_state = nullptr;
        }
        {
            std::lock_guard<std::mutex> lk(n->mu);
            if (++n->done == n->world) n->cv_done.notify_all();
        }
        if (cmd == 2) return;
    }
}

int node_post(sdfk_node* n, int cmd)
{
    std::unique_lock<std::mutex> lk(n->mu);
    if (cmd != 0) n->bar.rearm();   // (every worker is idle between commands; the start-up rendezvous -- command 0 -- is under way)
    n->cmd = cmd;
    n->done = 0;
    n->cmd_gen++;
    n->cv_cmd.notify_all();
    n->cv_done.wait(lk, [&] { return n->done == n->world; });
    return SDFK_OK;
}




void node_set_scene(sdfk_node* n, const sdfk_op* ops, int32_t n_ops, const int32_t out_rgbw[4], int32_t writes_color, const float min[3],
                    const float max[3], int32_t nx, int32_t ny, int32_t nz, int32_t clip_to_bounds, float iso_value)
{
    n->ops.assign(ops, ops + n_ops);
    memcpy(n->out_rgbw, out_rgbw, sizeof n->out_rgbw);
    n->writes_color = writes_color;
    memcpy(n->mn, min, 12);
    memcpy(n->mx, max, 12);
    n->nx = nx; n->ny = ny; n->nz = nz; n->clip = clip_to_bounds ? 1 : 0; n->iso = iso_value;
}
int node_first_error(sdfk_node* n, const char* what)
{
    // the rank that failed by itself first; a rank that only reports "another rank failed" when nobody else has anything better
    for (int pass = 0; pass < 2; pass++)
        for (auto& w : n->workers)
            if (w.status && (pass || !w.released)) return fail(w.status, "%s: rank %d (device %d): %s", what, w.rank, w.device, w.error.c_str());
    return SDFK_OK;
}


extern "C" int sdfk_node_open(const int32_t* devices, int32_t n_devices, sdfk_node** out)
{
    if (!out) return fail(SDFK_ERR_INVALID, "sdfk_node_open: null argument");
    *out = nullptr;
    config_from_env();
    int have = 0;
    const hipError_t e = hipGetDeviceCount(&have);
    if (e != hipSuccess || have <= 0) return fail(SDFK_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
    std::vector<int> devs;
    if (!devices || n_devices <= 0)
        for (int d = 0; d < have; d++) devs.push_back(d);
    else
        for (int i = 0; i < n_devices; i++) devs.push_back(devices[i]);
    if (devs.size() > 64) return fail(SDFK_ERR_INVALID, "sdfk_node_open: more than 64 ranks");
    bool shared = false;
    for (size_t i = 0; i < devs.size(); i++) {
        if (devs[i] < 0 || devs[i] >= have) return fail(SDFK_ERR_INVALID, "sdfk_node_open: device %d out of range (%d devices)", devs[i], have);
        for (size_t k = 0; k < i; k++) shared = shared || devs[k] == devs[i];
    }
    sdfk_node* n = new sdfk_node();
    n->world = (int)devs.size();
    n->host_transport = shared;   // (RCCL refuses two ranks on one device: ranks that share one exchange through the host)
    n->workers.resize(devs.size());
    n->send_ptrs.assign(devs.size(), nullptr);
    n->bar.n = n->world;
    if (const char* f = getenv("SDFK_NODE_FAULT_RANK")) n->fault_rank.store(atoi(f));
    if (const char* f = getenv("SDFK_NODE_FAULT_STAGE")) n->fault_stage = atoi(f);
    for (size_t i = 0; i < devs.size(); i++) {
        n->workers[i].node = n;
        n->workers[i].device = devs[i];
        n->workers[i].rank = (int)i;
    }
    for (auto& w : n->workers) w.th = std::thread(node_worker_main, &w);
    // the workers rendezvous among themselves; a no-op command tells when all of them are through their start-up
    node_post(n, 0);
    for (auto& w : n->workers)
        if (w.status) {
            const int r = w.status;
            const std::string why = "sdfk_node_open: rank " + std::to_string(w.rank) + " (device " + std::to_string(w.device) + "): " + w.error;
            node_post(n, 2);
            for (auto& q : n->workers) q.th.join();
            delete n;
            return fail(r, "%s", why.c_str());
        }
    *out = n;
    return SDFK_OK;
}

extern "C" int sdfk_node_info(const sdfk_node* n, int32_t* world, int32_t* backend)
{
    if (!n) return fail(SDFK_ERR_INVALID, "sdfk_node_info: null node");
    if (world) *world = n->world;
    if (backend) *backend = n->host_transport ? 2 : 1;
    return SDFK_OK;
}

extern "C" int sdfk_node_to_mesh(sdfk_node* n, const sdfk_op* ops, int32_t n_ops, const int32_t out_rgbw[4], int32_t writes_color,
                                 const float min[3], const float max[3], int32_t nx, int32_t ny, int32_t nz,
                                 int32_t clip_to_bounds, float iso_value, sdfk_mesh** out)
{
    if (!n || !ops || n_ops <= 0 || !out_rgbw || !min || !max || !out) return fail(SDFK_ERR_INVALID, "sdfk_node_to_mesh: null/empty argument");
    *out = nullptr;
    std::lock_guard<std::mutex> one(n->call_mu);
    node_set_scene(n, ops, n_ops, out_rgbw, writes_color, min, max, nx, ny, nz, clip_to_bounds, iso_value);
    n->mesh0 = nullptr;
    n->step_open = false;
    node_post(n, 1);
    if (int r = node_first_error(n, "sdfk_node_to_mesh")) {
        if (n->mesh0) sdfk_mesh_free(n->mesh0);
        n->mesh0 = nullptr;
        return r;
    }
    *out = n->mesh0;
    n->mesh0 = nullptr;
    return SDFK_OK;
}


// The mesh on the HOST, in two phases like every managed caller needs it (Mesh.cs:10-13: four exact-length arrays): begin runs the
// sharded step and returns the totals; copy fills the caller's arrays -- every rank copies its own slab into its slice over its own
// PCIe link, all at once; the slabs never cross xGMI.
extern "C" int sdfk_node_mesh_begin(sdfk_node* n, const sdfk_op* ops, int32_t n_ops, const int32_t out_rgbw[4], int32_t writes_color,
                                    const float min[3], const float max[3], int32_t nx, int32_t ny, int32_t nz, int32_t clip_to_bounds,
                                    float iso_value, int64_t* n_vertices, int64_t* n_indices, int32_t* has_colors)
{
    if (!n || !ops || n_ops <= 0 || !out_rgbw || !min || !max) return fail(SDFK_ERR_INVALID, "sdfk_node_mesh_begin: null/empty argument");
    std::lock_guard<std::mutex> one(n->call_mu);
    node_set_scene(n, ops, n_ops, out_rgbw, writes_color, min, max, nx, ny, nz, clip_to_bounds, iso_value);
    n->step_open = false;
    node_post(n, 3);
    if (int r = node_first_error(n, "sdfk_node_mesh_begin")) return r;
    n->step_open = true;
    if (n_vertices) *n_vertices = n->total_v;
    if (n_indices) *n_indices = n->total_i;
    if (has_colors) *has_colors = n->has_colors;
    return SDFK_OK;
}

extern "C" int sdfk_node_mesh_copy(sdfk_node* n, float* vertices3, float* colors3, float* normals3, int32_t* triangles, float min[3], float max[3])
{
    if (!n) return fail(SDFK_ERR_INVALID, "sdfk_node_mesh_copy: null node");
    std::lock_guard<std::mutex> one(n->call_mu);
    if (!n->step_open) return fail(SDFK_ERR_INVALID, "sdfk_node_mesh_copy: no step to copy (call sdfk_node_mesh_begin first)");
    n->dst_v = vertices3; n->dst_c = colors3; n->dst_n = normals3; n->dst_t = triangles;
    node_post(n, 4);
    n->dst_v = n->dst_c = n->dst_n = nullptr; n->dst_t = nullptr;
    if (int r = node_first_error(n, "sdfk_node_mesh_copy")) return r;
    if (min) memcpy(min, n->bmin, 12);
    if (max) memcpy(max, n->bmax, 12);
    return SDFK_OK;
}

extern "C" void sdfk_node_close(sdfk_node* n)
{
    if (!n) return;
    {
        std::lock_guard<std::mutex> one(n->call_mu);
        node_post(n, 2);
    }
    for (auto& w : n->workers) w.th.join();
    delete n;
}
