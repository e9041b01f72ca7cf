// dist_rccl.h -- the Z-slab sharded step behind the C ABI (include/sdfkit_hip.h, "Z-slab sharding"): the GPU backend of
// the protocol in slab_protocol.h.  Included by lib_dist.hip (it uses the library's context, lanes, allocator and the
// slab form of the captured step graphs).
//
// Streams of a sharded rank: the library's own stream (exact steps, mesh extraction), lanes 1..3 (the steps' kernel
// chains, rotating per step, one captured hipGraphLaunch each) and ONE exchange stream that carries, per step,
//   wait(step packed) -> ncclAllGather | grouped ncclSend/ncclRecv -> k_slabs_rebase (+ header mirror) -> record(ready)
// so the exchange of step i travels while the kernels of step i+1 run, and nothing on the host waits inside a step.
// RCCL is loaded with dlopen on first use (librccl.so.1; SDFK_RCCL_LIB at start-up names another file): a process that never
// shards never loads it.  The DEFAULT exchange is the plainest collective there is: ncclAllGather of int32-index payloads,
// in place (SDFK_OPT_DIST_EXCHANGE = 0) -- the one configuration whose every part has the library's own tests behind it on
// real RCCL.  xGMI is point-to-point -- 7 links x 76.8 GB/s per direction per GPU --, so sending every peer its copy directly
// over the link between the two (mode 1: one grouped ncclSend / ncclRecv launch) or to rank 0 only (mode 2), 16-bit-index
// payloads (SDFK_OPT_DIST_INDEX16) and the tuner that measures them against each other (sdfk_dist_tune) exist -- as explicit
// opt-ins: none of them has run between two GPUs yet (one GPU per development box).
#pragma once
#ifndef SDFK_LIB_DIST_TU
#error "dist_rccl.h holds definitions: it is part of lib_dist.hip, not a header to include elsewhere (declarations: lib_internal.h)"
#endif
#include <dlfcn.h>

#include "slab_protocol.h"



// (struct RcclApi, struct DistContext and the per-context `gd`: lib_internal.h, "device contexts")

int rccl_load()
{
    RcclApi& A = gd.nccl;
    if (A.handle) return SDFK_OK;
    const char* names[] = {g_cfg.rccl_lib.empty() ? nullptr : g_cfg.rccl_lib.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n) continue;
        A.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (A.handle) break;
    }
    if (!A.handle) return fail(SDFK_ERR_UNSUPPORTED, "RCCL is not available: %s", dlerror());
    bool ok = true;
    auto sym = [&](const char* n) { void* p = dlsym(A.handle, n); if (!p) ok = false; return p; };
    A.GetUniqueId = (decltype(A.GetUniqueId))sym("ncclGetUniqueId");
    A.CommInitRank = (decltype(A.CommInitRank))sym("ncclCommInitRank");
    A.CommDestroy = (decltype(A.CommDestroy))sym("ncclCommDestroy");
    A.AllGather = (decltype(A.AllGather))sym("ncclAllGather");
    A.Send = (decltype(A.Send))sym("ncclSend");
    A.Recv = (decltype(A.Recv))sym("ncclRecv");
    A.GroupStart = (decltype(A.GroupStart))sym("ncclGroupStart");
    A.GroupEnd = (decltype(A.GroupEnd))sym("ncclGroupEnd");
    A.GetErrorString = (decltype(A.GetErrorString))sym("ncclGetErrorString");
    if (!ok) { dlclose(A.handle); A = RcclApi(); return fail(SDFK_ERR_UNSUPPORTED, "the RCCL library lacks an entry point this library needs"); }
    return SDFK_OK;
}

#define NCCLCHK(expr)                                                                                                 \
    do {                                                                                                              \
        ncclResult_t r_ = (expr);                                                                                     \
        if (r_ != ncclSuccess) return fail(SDFK_ERR_HIP, "%s: %s (%s:%d)", #expr, gd.nccl.GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

void dist_release();

int dist_common_init(int world, int rank)
{
    if (int r = require_init()) return r;
    if (gd.backend) return fail(SDFK_ERR_INVALID, "sdfk_dist_init: already initialised (world %d, rank %d)", gd.world, gd.rank);
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(SDFK_ERR_INVALID, "sdfk_dist_init: bad world %d / rank %d", world, rank);
    // the exchange stream: the one the stream placement keeps for it (lane 0's class: never busy beside a lane's queue or pipe)
    hipError_t e = hipSuccess;
    gd.stream = placed_exchange_stream();
    gd.stream_owned = gd.stream == nullptr;
    if (!gd.stream) e = hipStreamCreateWithFlags(&gd.stream, hipStreamNonBlocking);
    int r = e == hipSuccess ? dev_alloc((void**)&gd.agree_dev, sizeof(int64_t) * (1 + world)) : fail(SDFK_ERR_HIP, "exchange stream: %s", hipGetErrorString(e));
    if (!r && hipHostMalloc((void**)&gd.agree_host, sizeof(int64_t) * (1 + world), hipHostMallocDefault) != hipSuccess)
        r = fail(SDFK_ERR_NOMEM, "sdfk_dist_init: pinned memory for the stride agreement");
    if (r) {   // (nothing half-made stays behind: a later sdfk_dist_init starts from scratch)
        const std::string keep_err = t_err;
        dist_release();
        t_err = keep_err;
        return r;
    }
    gd.world = world;
    gd.rank = rank;
    return SDFK_OK;
}

bool dist_active() { return gd.backend != 0 || gd.stream != nullptr; }

void dist_release()
{
    if (gd.comm && gd.nccl.CommDestroy) (void)gd.nccl.CommDestroy(gd.comm);
    gd.comm = nullptr;
    if (gd.stream) { (void)hipStreamSynchronize(gd.stream); if (gd.stream_owned) (void)hipStreamDestroy(gd.stream); }
    gd.stream = nullptr;
    dev_free(gd.agree_dev);
    gd.agree_dev = nullptr;
    if (gd.agree_host) (void)hipHostFree(gd.agree_host);
    gd.agree_host = nullptr;
    if (gd.stage) (void)hipHostFree(gd.stage);
    gd.stage = nullptr;
    gd.stage_stride = 0;
    gd.backend = 0;
    gd.world = 1;
    gd.rank = 0;
    gd.host_fn = nullptr;
    gd.host_ctx = nullptr;
    gd.consensus_fn = nullptr;
    gd.consensus_ctx = nullptr;
}

// The whole mesh out of a gather buffer: the sections of slab r go to the vertex / index offsets the headers give
// (exclusive prefix of the counts of slabs 0..r-1); indices were rebased by the step's own kernel.  blockIdx.y = slab.
struct ConcatArgs {
    const char* gathered;
    int world;
    int64_t stride;
    float* vertices;
    float* colors;
    float* normals;
    int32_t* triangles;
    float* bounds;   // device float[6]
    const int32_t* decoded;   // compact steps: the whole mesh's indices as the step decoded them (k_slabs_decode16), or null
};

__global__ __launch_bounds__(256) void k_slabs_concat(ConcatArgs A)
{
    using sdfk::SlabHeader;
    const int r = blockIdx.y;
    int64_t vbase = 0, ibase = 0;
    for (int q = 0; q < r; q++) {
        const SlabHeader* h = reinterpret_cast<const SlabHeader*>(A.gathered + (size_t)q * A.stride);
        vbase += h->nv;
        ibase += h->ni;
    }
    const SlabHeader* h = reinterpret_cast<const SlabHeader*>(A.gathered + (size_t)r * A.stride);
    const int64_t nv = h->nv, ni = h->ni, capv = h->cap_v > 0 ? (int64_t)h->cap_v : nv;
    const bool wc = h->vbytes == 36;
    const char* sec = A.gathered + (size_t)r * A.stride + sizeof(SlabHeader);
    const uint32_t* sv = reinterpret_cast<const uint32_t*>(sec);
    const uint32_t* sc = reinterpret_cast<const uint32_t*>(sec + 12 * capv);
    const uint32_t* sn = reinterpret_cast<const uint32_t*>(sec + (wc ? 24 : 12) * capv);
    const uint32_t* st = reinterpret_cast<const uint32_t*>(sec + (int64_t)h->vbytes * capv);
    // compact slab (k_payload_compact): uint16 offsets against one int32 base per 1024 indices, slab-LOCAL ids (not rebased)
    const bool c16 = h->idx_bits == 16;
    const uint16_t* t16 = reinterpret_cast<const uint16_t*>(st);
    const int32_t* bases = reinterpret_cast<const int32_t*>(sec + (int64_t)h->vbytes * capv + ((2 * ni + 3) & ~int64_t(3)));
    uint32_t* dv = reinterpret_cast<uint32_t*>(A.vertices) + 3 * vbase;
    uint32_t* dc = reinterpret_cast<uint32_t*>(A.colors) + 3 * vbase;
    uint32_t* dn = reinterpret_cast<uint32_t*>(A.normals) + 3 * vbase;
    uint32_t* dt = reinterpret_cast<uint32_t*>(A.triangles) + ibase;
    const int64_t nf = 3 * nv, total = 3 * nf + ni;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        if (i < nf) dv[i] = sv[i];
        else if (i < 2 * nf) dc[i - nf] = wc ? sc[i - nf] : 0u;   // (a volume without colours: all zero, Voxels.cs:88-92)
        else if (i < 3 * nf) dn[i - 2 * nf] = sn[i - 2 * nf];
        else {
            const int64_t k = i - 3 * nf;
            dt[k] = !c16 ? st[k] : (A.decoded ? (uint32_t)A.decoded[ibase + k] : (uint32_t)(bases[k / sdfk::SLAB_IDX_BLOCK] + (int32_t)t16[k] + (int32_t)vbase));
        }
    }
    if (r == 0 && blockIdx.x == 0 && threadIdx.x == 0) {   // Mesh.Measure (Mesh.cs:30-45) over the slabs that have vertices
        float b[6] = {0, 0, 0, 0, 0, 0};
        bool any = false;
        for (int q = 0; q < A.world; q++) {
            const SlabHeader* hq = reinterpret_cast<const SlabHeader*>(A.gathered + (size_t)q * A.stride);
            if (hq->nv <= 0) continue;
            for (int k = 0; k < 3; k++) {
                b[k] = any ? fminf(b[k], hq->bmin[k]) : hq->bmin[k];
                b[3 + k] = any ? fmaxf(b[3 + k], hq->bmax[k]) : hq->bmax[k];
            }
            any = true;
        }
        for (int k = 0; k < 6; k++) A.bounds[k] = b[k];
    }
}

// ONE slab of a gather buffer as a mesh of its own: sections -> dense arrays, indices + ibias (the vertex counts of the slabs before
// it when the step left them slab-local -- exchange mode 3 --, 0 when the step's own kernel has rebased them already).
struct SlabExtractArgs {
    const char* section;   // the slab's payload (header first)
    float* vertices;
    float* colors;
    float* normals;
    int32_t* triangles;
    float* bounds;
    int32_t ibias;
};

__global__ __launch_bounds__(256) void k_slab_extract(SlabExtractArgs A)
{
    using sdfk::SlabHeader;
    const SlabHeader* h = reinterpret_cast<const SlabHeader*>(A.section);
    const int64_t nv = h->nv, ni = h->ni, capv = h->cap_v > 0 ? (int64_t)h->cap_v : nv;
    const bool wc = h->vbytes == 36;
    const char* sec = A.section + sizeof(SlabHeader);
    const uint32_t* sv = reinterpret_cast<const uint32_t*>(sec);
    const uint32_t* sc = reinterpret_cast<const uint32_t*>(sec + 12 * capv);
    const uint32_t* sn = reinterpret_cast<const uint32_t*>(sec + (wc ? 24 : 12) * capv);
    const int32_t* st = reinterpret_cast<const int32_t*>(sec + (int64_t)h->vbytes * capv);
    const bool c16 = h->idx_bits == 16;
    const uint16_t* t16 = reinterpret_cast<const uint16_t*>(st);
    const int32_t* bases = reinterpret_cast<const int32_t*>(sec + (int64_t)h->vbytes * capv + ((2 * ni + 3) & ~int64_t(3)));
    uint32_t* dv = reinterpret_cast<uint32_t*>(A.vertices);
    uint32_t* dc = reinterpret_cast<uint32_t*>(A.colors);
    uint32_t* dn = reinterpret_cast<uint32_t*>(A.normals);
    const int64_t nf = 3 * nv, total = 3 * nf + ni;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        if (i < nf) dv[i] = sv[i];
        else if (i < 2 * nf) dc[i - nf] = wc ? sc[i - nf] : 0u;
        else if (i < 3 * nf) dn[i - 2 * nf] = sn[i - 2 * nf];
        else {
            const int64_t k = i - 3 * nf;
            A.triangles[k] = (c16 ? bases[k / sdfk::SLAB_IDX_BLOCK] + (int32_t)t16[k] : st[k]) + A.ibias;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 6) A.bounds[threadIdx.x] = nv > 0 ? (threadIdx.x < 3 ? h->bmin[threadIdx.x] : h->bmax[threadIdx.x - 3]) : 0.0f;
}



// ---------------------------------------------------------------------------------------------------------------------
// the session: SlabOps on HIP + RCCL (or the host transport)
// ---------------------------------------------------------------------------------------------------------------------
struct sdfk_dist_session final : sdfk::SlabOps {
    sdfk_program* prog = nullptr;
    float mn[3], mx[3], iso = 0.0f;
    int nx = 0, ny = 0, nz = 0, clip = 0;
    int lb = 0, le = 0, z0 = 0, nzl = 0;
    int vbytes = 24;
    int exchange_mode = 0;
    int lanes = 3;
    bool idx16 = false;            // SDFK_OPT_DIST_INDEX16: compact payloads (k_payload_compact); falls back when a slab does not fit
    int64_t idx16_fallbacks = 0;
    uint64_t nsub = 0;
    int64_t stride = 0;
    int64_t host_ns_submit = 0, host_ns_collect = 0, steps = 0;
    struct Slot {
        sdfk_volume* vol = nullptr;
        char* gathered = nullptr;         // world x stride; this rank's section is the step's send buffer (in-place exchange)
        int64_t* hdr_host = nullptr;      // pinned + mapped: world x 8 words, written by the rebase kernel
        void* hdr_dev = nullptr;          // its device address
        hipEvent_t packed = nullptr;      // the step's kernels have written the send buffer
        hipEvent_t ready = nullptr;       // exchange + rebase + header mirror done
        hipEvent_t read = nullptr;        // sdfk_dist_mesh has read the gather buffer
        char* stage = nullptr;            // idx16: the step emits its plain payload here, k_payload_compact encodes it into the gather buffer
        int64_t stage_bytes = 0;
        unsigned long long* ticket = nullptr;   // arrival counter of k_payload_compact (zero between launches)
        int32_t* decoded = nullptr;       // idx16: the whole mesh's int32 indices, decoded by the step (k_slabs_decode16)
        int64_t decoded_cap = 0;
        bool ready_valid = false, read_valid = false;
        bool payloads_gathered = false;   // the last exchange of this slot brought THIS rank the other ranks' slab payloads (false: headers
                                          // only -- exchange mode 3, or a rank other than 0 under gather-to-root); set once the exchange is queued
        bool own_rebased = false;         // this rank's own section holds GLOBAL indices: the rebase kernel ran over it (never with headers
                                          // only, never in the compact form) -- what sdfk_dist_slab_mesh must know to bias or not
        sdfk_mesh* exact = nullptr;       // mesh of run_exact until pack_exact
    };
    std::vector<Slot> slots;
    sdfk::SlabProtocol proto;
    std::string err;

    sdfk_dist_session(int depth) : slots((size_t)depth), proto(this, depth, 1.0 / 32) {}

    char* send_buf(Slot& s) const { return s.gathered + (size_t)gd.rank * stride; }
    int64_t need_bytes(int64_t nv, int64_t ni) const
    {
        if (!idx16) return SDFK_SLAB_HEADER_BYTES + (int64_t)vbytes * nv + 4 * ni;
        return SDFK_SLAB_HEADER_BYTES + (int64_t)vbytes * nv + ((2 * ni + 3) & ~int64_t(3)) + 4 * ((ni + sdfk::SLAB_IDX_BLOCK - 1) / sdfk::SLAB_IDX_BLOCK);
    }
    int64_t payload_bytes(const int64_t* hdr) const override
    {
        const int64_t vb = hdr[5] & 0xffffffffll, bits = hdr[6] & 0xffffffffll;
        if (bits != 16) return SDFK_SLAB_HEADER_BYTES + vb * hdr[0] + 4 * hdr[1];
        return SDFK_SLAB_HEADER_BYTES + vb * hdr[0] + ((2 * hdr[1] + 3) & ~int64_t(3)) + 4 * ((hdr[1] + sdfk::SLAB_IDX_BLOCK - 1) / sdfk::SLAB_IDX_BLOCK);
    }
    // plain payload in the slot's stage buffer -> compact payload in this rank's section of the gather buffer (queued on `st`)
    int compact(Slot& s, hipStream_t st)
    {
        const int64_t words = s.stage_bytes / 4;
        hipLaunchKernelGGL(sdfk::k_payload_compact, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((words + 4095) / 4096, 1024))), dim3(256), 0, st,
                           (const char*)s.stage, send_buf(s), stride, s.ticket);
        if (hipGetLastError() != hipSuccess) return keep(fail(SDFK_ERR_HIP, "k_payload_compact launch failed"));
        return SDFK_OK;
    }
    int keep(int r) { if (r) err = t_err; return r; }

    int world() const override { return gd.world; }
    int rank() const override { return gd.rank; }
    const char* last_error() const override { return err.c_str(); }
    // (slab_protocol.h: before a collective, after the rank-local part.  Only a node's rank threads have an agreement that costs nothing.)
    int consensus(int mine) override
    {
        if (!gd.consensus_fn) return mine;
        if (mine && err.empty()) err = t_err;
        const int r = gd.consensus_fn(gd.consensus_ctx, mine);
        if (r && !mine) err = t_err;   // (another rank's failure, named by the hook)
        return r;
    }

    int run_exact(int k, int64_t* nv, int64_t* ni, int64_t* need) override
    {
        Slot& s = slots[k];
        if (s.exact) { sdfk_mesh_free(s.exact); s.exact = nullptr; }
        if (int r = sdfk_sample_march_slab(prog, s.vol, clip, iso, lb, le, 0, &s.exact)) return keep(r);
        if (int r = sdfk_mesh_counts(s.exact, nv, ni)) return keep(r);
        *need = need_bytes(*nv, *ni);
        return SDFK_OK;
    }

    int agree_max(int64_t mine, int64_t* mx_all) override
    {
        const int w = gd.world;
        if (gd.backend == 2) {
            gd.agree_host[0] = mine;
            if (gd.host_fn(gd.host_ctx, gd.agree_host, gd.agree_host + 1, sizeof(int64_t)) != 0)
                return keep(fail(SDFK_ERR_HIP, "the host transport's all-gather failed"));
        } else {
            gd.agree_host[0] = mine;
            hipError_t e = hipMemcpyAsync(gd.agree_dev, gd.agree_host, sizeof(int64_t), hipMemcpyHostToDevice, gd.stream);
            if (e == hipSuccess) {
                const ncclResult_t nr = gd.nccl.AllGather(gd.agree_dev, gd.agree_dev + 1, sizeof(int64_t), ncclChar, gd.comm, gd.stream);
                if (nr != ncclSuccess)
                    return keep(fail(SDFK_ERR_HIP, "ncclAllGather (stride agreement): %s (rank %d of %d, device %d)", gd.nccl.GetErrorString(nr), gd.rank, w, g.device));
                e = hipMemcpyAsync(gd.agree_host + 1, gd.agree_dev + 1, sizeof(int64_t) * w, hipMemcpyDeviceToHost, gd.stream);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(gd.stream);
            if (e != hipSuccess) return keep(fail(SDFK_ERR_HIP, "stride agreement: %s", hipGetErrorString(e)));
        }
        int64_t m = mine;
        for (int q = 0; q < w; q++) m = std::max(m, gd.agree_host[1 + q]);
        *mx_all = m;
        return SDFK_OK;
    }

    void free_buffers()
    {
        for (Slot& s : slots) {
            if (s.vol) graph_jobs_forget_volume(s.vol);   // (captured steps write into the old send buffer)
            dev_free(s.gathered);
            s.gathered = nullptr;
            dev_free(s.stage);
            s.stage = nullptr;
            s.stage_bytes = 0;
            dev_free(s.ticket);
            s.ticket = nullptr;
            dev_free(s.decoded);
            s.decoded = nullptr;
            s.decoded_cap = 0;
            if (s.hdr_host) (void)hipHostFree(s.hdr_host);
            s.hdr_host = nullptr;
            s.hdr_dev = nullptr;
            s.ready_valid = s.read_valid = false;
        }
    }

    // All-or-nothing: `stride` is committed only when every slot has every buffer; a failure leaves the session without
    // buffers and with stride 0 (the protocol then bootstraps again instead of enqueueing into a null send buffer).
    int resize(int64_t new_stride) override
    {
        free_buffers();
        stride = 0;
        const int r = resize_alloc(new_stride);
        if (r) { const std::string keep_err = err; free_buffers(); err = keep_err; return r; }
        stride = new_stride;
        return SDFK_OK;
    }

    int resize_alloc(int64_t ns)
    {
        const size_t total = (size_t)gd.world * (size_t)ns;
        for (Slot& s : slots) {
            if (int r = dev_alloc((void**)&s.gathered, total)) return keep(r);
            hipError_t e = hipMemsetAsync(s.gathered, 0, total, g.stream);
            if (idx16) {   // (a plain payload is at most twice its compact form: indices 4 instead of 2 bytes)
                s.stage_bytes = 2 * ns;
                if (int r = dev_alloc((void**)&s.stage, (size_t)s.stage_bytes)) return keep(r);
                if (int r = dev_alloc((void**)&s.ticket, sizeof(unsigned long long))) return keep(r);
                s.decoded_cap = (int64_t)gd.world * (ns / 2);   // (an index takes at least 2 bytes of a payload)
                if (int r = dev_alloc((void**)&s.decoded, (size_t)s.decoded_cap * sizeof(int32_t))) return keep(r);
                if (e == hipSuccess) e = hipMemsetAsync(s.ticket, 0, sizeof(unsigned long long), g.stream);
                if (e == hipSuccess) e = hipMemsetAsync(s.stage, 0, SDFK_SLAB_HEADER_BYTES, g.stream);
            }
            if (e == hipSuccess) e = hipHostMalloc((void**)&s.hdr_host, (size_t)gd.world * SDFK_SLAB_HEADER_BYTES, hipHostMallocMapped);
            if (e == hipSuccess) { memset(s.hdr_host, 0, (size_t)gd.world * SDFK_SLAB_HEADER_BYTES); e = hipHostGetDevicePointer(&s.hdr_dev, s.hdr_host, 0); }
            if (e != hipSuccess) return keep(fail(SDFK_ERR_HIP, "gather buffers: %s", hipGetErrorString(e)));
        }
        if (gd.backend == 2 && gd.stage_stride < ns) {
            if (gd.stage) (void)hipHostFree(gd.stage);
            gd.stage = nullptr;
            gd.stage_stride = 0;
            if (hipHostMalloc((void**)&gd.stage, (size_t)(1 + gd.world) * (size_t)ns, hipHostMallocDefault) != hipSuccess)
                return keep(fail(SDFK_ERR_NOMEM, "pinned staging for the host transport (%lld bytes)", (long long)((1 + gd.world) * ns)));
            gd.stage_stride = ns;
        }
        if (hipStreamSynchronize(g.stream) != hipSuccess) return keep(fail(SDFK_ERR_HIP, "gather buffers: memset failed"));
        return SDFK_OK;
    }

    // a slot whose buffers are gone (a failed regrowth) is never written through
    int have_buffers(const Slot& s, const char* what)
    {
        if (stride > 0 && s.gathered && s.hdr_host && (!idx16 || (s.stage && s.ticket && s.decoded))) return SDFK_OK;
        return keep(fail(SDFK_ERR_INVALID, "%s: the session has no gather buffers (an earlier regrowth failed): submit() bootstraps again", what));
    }

    int pack_exact(int k) override
    {
        Slot& s = slots[k];
        if (!s.exact) return keep(fail(SDFK_ERR_INVALID, "pack_exact without an exact mesh"));
        if (int r = have_buffers(s, "pack_exact")) return r;
        int64_t need = 0;
        int r = idx16 ? sdfk_mesh_pack(s.exact, s.stage, s.stage_bytes, &need) : sdfk_mesh_pack(s.exact, send_buf(s), stride, &need);
        sdfk_mesh_free(s.exact);   // (stream-ordered)
        s.exact = nullptr;
        if (!r && idx16) r = compact(s, g.stream);
        if (r) return keep(r);
        if (hipEventRecord(s.packed, g.stream) != hipSuccess) return keep(fail(SDFK_ERR_HIP, "hipEventRecord failed"));
        return SDFK_OK;
    }

    // the step's kernels: lane section (the lanes in rotation), ONE captured graph launch from the second use of a (slot, lane) on
    int enqueue(int k) override
    {
        Slot& s = slots[k];
        if (int r = have_buffers(s, "enqueue")) return r;
        const int lane = lanes ? 1 + (int)(++nsub % (uint64_t)lanes) : 0;
        hipStream_t st = lane ? lane_stream(lane) : g.stream;
        // the send buffer is a section of the gather buffer: its last exchange must have finished, and so must a mesh
        // extraction that reads it (on lane 0 that one is ordered by the stream itself)
        if (s.read_valid && lane) { if (hipStreamWaitEvent(st, s.read, 0) != hipSuccess) return keep(fail(SDFK_ERR_HIP, "hipStreamWaitEvent failed")); }
        if (s.ready_valid && !lane) { if (hipStreamWaitEvent(st, s.ready, 0) != hipSuccess) return keep(fail(SDFK_ERR_HIP, "hipStreamWaitEvent failed")); }
        s.read_valid = false;
        const PostCompact pc{send_buf(s), stride, s.ticket};   // (compact payloads: the encoder is the last node of the captured step)
        if (int r = slab_enqueue_impl(prog, s.vol, clip, iso, lb, le, idx16 ? s.stage : send_buf(s), idx16 ? s.stage_bytes : stride, lane,
                                      (lane && s.ready_valid) ? s.ready : nullptr, false, idx16 ? &pc : nullptr)) return keep(r);
        if (hipEventRecord(s.packed, st) != hipSuccess) return keep(fail(SDFK_ERR_HIP, "hipEventRecord failed"));
        return SDFK_OK;
    }

    int exchange(int k) override { return exchange_as(k, exchange_mode); }

    // mode 3 ("the mesh stays sharded"): only the 64-byte HEADERS travel -- every decision of the protocol needs them, nothing else
    // does -- and the slab payloads stay where they were emitted until somebody asks for the whole mesh (sdfk_dist_mesh then runs
    // the payload exchange of that one step, exchange_as(k, 0)); sdfk_dist_slab_mesh hands out a rank's own slab without any exchange.
    // What every rank has to RECEIVE per step drops from (world - 1) slab meshes to (world - 1) x 64 bytes: the step is no longer
    // bound by the fabric ("the fabric bound").
    int exchange_as(int k, int mode)
    {
        Slot& s = slots[k];
        if (int r = have_buffers(s, "exchange")) return r;
        const int w = gd.world, me = gd.rank;
        hipStream_t cs = gd.stream;
        // what THIS rank receives: headers only in mode 3, and under gather-to-root (mode 2) on every rank but 0 -- with either transport
        const bool headers_only = w > 1 && ((mode == 2 && me != 0) || mode == 3);
        // (host bookkeeping of device-side effects: both flags are set at the END, once everything of this exchange is queued -- a
        // failed on-demand payload exchange must not look done to the next sdfk_dist_mesh, and an own section nobody rebased must
        // not be handed out as global)
        s.payloads_gathered = false;
        s.own_rebased = false;
        if (gd.backend == 2) {
            // host transport: device -> pinned, the host's own all-gather, pinned -> device (synchronous: a test / bring-up path)
            char* hs = gd.stage;
            char* hr = gd.stage + stride;
            const int64_t piece = mode == 3 ? (int64_t)SDFK_SLAB_HEADER_BYTES : stride;   // (mode 3: the headers only)
            hipError_t e = hipEventSynchronize(s.packed);
            if (e == hipSuccess) e = hipMemcpyAsync(hs, send_buf(s), (size_t)piece, hipMemcpyDeviceToHost, cs);
            if (e == hipSuccess) e = hipStreamSynchronize(cs);
            if (e != hipSuccess) return keep(fail(SDFK_ERR_HIP, "host transport: %s", hipGetErrorString(e)));
            if (gd.host_fn(gd.host_ctx, hs, hr, piece) != 0) return keep(fail(SDFK_ERR_HIP, "the host transport's all-gather failed"));
            if (headers_only) {   // header q -> the head of section q (this rank's own section keeps its payload); mode 2: like RCCL's
                                  // gather-to-root, only rank 0 takes the payloads (the host's all-gather moved them anyway: a test path)
                for (int q = 0; q < w && e == hipSuccess; q++)
                    if (q != me) e = hipMemcpyAsync(s.gathered + (size_t)q * stride, hr + (size_t)q * piece, (size_t)SDFK_SLAB_HEADER_BYTES, hipMemcpyHostToDevice, cs);
            } else
                e = hipMemcpyAsync(s.gathered, hr, (size_t)w * (size_t)stride, hipMemcpyHostToDevice, cs);
            if (e == hipSuccess) e = hipStreamSynchronize(cs);   // (the staging block is reused by the next exchange)
            if (e != hipSuccess) return keep(fail(SDFK_ERR_HIP, "host transport: %s", hipGetErrorString(e)));
        } else {
            if (hipStreamWaitEvent(cs, s.packed, 0) != hipSuccess) return keep(fail(SDFK_ERR_HIP, "hipStreamWaitEvent failed"));
            const RcclApi& N = gd.nccl;
            ncclResult_t nr = ncclSuccess;
            const int exchange_mode = mode;   // (the mode of THIS exchange)
            if (exchange_mode == 0 || w == 1) {
                nr = N.AllGather(send_buf(s), s.gathered, (size_t)stride, ncclChar, gd.comm, cs);   // in place
            } else {
                // every peer gets this rank's payload over the link between the two, all links at once: ONE grouped launch.
                // mode 2: only rank 0 receives payloads; the others get the 64-byte headers (the protocol's decisions need them).
                nr = N.GroupStart();
                for (int q = 0; q < w && nr == ncclSuccess; q++) {
                    if (q == me) continue;
                    const bool full_to_q = exchange_mode == 1 || (exchange_mode == 2 && q == 0), full_from_q = exchange_mode == 1 || (exchange_mode == 2 && me == 0);
                    nr = N.Send(send_buf(s), full_to_q ? (size_t)stride : (size_t)SDFK_SLAB_HEADER_BYTES, ncclChar, q, gd.comm, cs);
                    if (nr == ncclSuccess)
                        nr = N.Recv(s.gathered + (size_t)q * stride, full_from_q ? (size_t)stride : (size_t)SDFK_SLAB_HEADER_BYTES, ncclChar, q, gd.comm, cs);
                }
                const ncclResult_t ne = N.GroupEnd();
                if (nr == ncclSuccess) nr = ne;
            }
            if (nr != ncclSuccess)
                return keep(fail(SDFK_ERR_HIP, "RCCL exchange: %s (rank %d of %d, device %d, exchange mode %d, %s payloads, stride %lld bytes, slot %d)",
                                 N.GetErrorString(nr), me, w, g.device, exchange_mode, idx16 ? "16-bit-index" : "plain", (long long)stride, k));
        }
        // indices of slab r += vertices of slabs 0..r-1; the headers land in pinned host memory (one event, no copy).
        // (mode 2 on a rank other than 0: there are no foreign payloads to rebase -- the kernel sees header-only slabs)
        if (idx16 && !headers_only)   // compact slabs: decode into the whole mesh's int32 index array (+ the header mirror)
            hipLaunchKernelGGL(sdfk::k_slabs_decode16, dim3(64, w), dim3(256), 0, cs, (const char*)s.gathered, w, stride, (sdfk::SlabHeader*)s.hdr_dev,
                               s.decoded, s.decoded_cap);
        else
            hipLaunchKernelGGL(sdfk::k_slabs_rebase, dim3(headers_only ? 1 : 64, w), dim3(256), 0, cs, s.gathered, w, stride, (sdfk::SlabHeader*)s.hdr_dev,
                               headers_only ? 1 : 0);
        if (hipGetLastError() != hipSuccess || hipEventRecord(s.ready, cs) != hipSuccess) return keep(fail(SDFK_ERR_HIP, "rebase launch failed"));
        s.ready_valid = true;
        s.payloads_gathered = !headers_only;
        s.own_rebased = !headers_only && !idx16;   // (k_slabs_rebase returns early for mirror_only and leaves 16-bit sections alone)
        return SDFK_OK;
    }

    int headers(int k, const int64_t** hdr) override
    {
        Slot& s = slots[k];
        if (!s.ready_valid) return keep(fail(SDFK_ERR_INVALID, "headers of a slot that was never exchanged"));
        const hipError_t e = hipEventSynchronize(s.ready);
        if (e != hipSuccess) return keep(fail(SDFK_ERR_HIP, "waiting for the exchange: %s", hipGetErrorString(e)));
        if (idx16)   // some rank's indices did not fit 16 bits (flags bit 0): every rank sees it here and goes back to int32 indices;
            for (int q = 0; q < gd.world; q++)   // the step carries -1 / -1 and is redone exactly, which re-agrees (and regrows) the stride
                if ((s.hdr_host[q * sdfk::kSlabHeaderWords + 6] >> 32) & 1) { idx16 = false; idx16_fallbacks++; break; }
        *hdr = s.hdr_host;
        return SDFK_OK;
    }

    int quiesce() override
    {
        hipError_t e = hipStreamSynchronize(g.stream);
        sync_all_lanes();
        if (e == hipSuccess && gd.stream) e = hipStreamSynchronize(gd.stream);
        if (e != hipSuccess) return keep(fail(SDFK_ERR_HIP, "quiesce: %s", hipGetErrorString(e)));
        return SDFK_OK;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int sdfk_dist_unique_id(void* id_out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!id_out) return fail(SDFK_ERR_INVALID, "sdfk_dist_unique_id: null argument");
    config_from_env();
    if (int r = rccl_load()) return r;
    static_assert(sizeof(ncclUniqueId) == SDFK_DIST_ID_BYTES, "SDFK_DIST_ID_BYTES");
    ncclUniqueId id;
    NCCLCHK(gd.nccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return SDFK_OK;
}

// sdfk_dist_init in its two halves: everything that can fail on ONE rank (the library, the exchange stream, the agreement buffers) ...
int dist_prepare_rccl(int32_t world, int32_t rank)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (int r = rccl_load()) return r;
    return dist_common_init(world, rank);
}

// ... and the collective ncclCommInitRank, which returns when EVERY rank has made the call: ranks that can (a node's threads) agree
// between the two that all of them are prepared, so that nobody waits for a rank that never joins
int dist_join_rccl(const void* id)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    const int world = gd.world, rank = gd.rank;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    const ncclResult_t nr = gd.nccl.CommInitRank(&gd.comm, world, uid, rank);
    if (nr != ncclSuccess) {
        const int r = fail(SDFK_ERR_HIP, "ncclCommInitRank(world %d, rank %d): %s", world, rank, gd.nccl.GetErrorString(nr));
        gd.comm = nullptr;
        dist_release();
        return r;
    }
    gd.backend = 1;
    return SDFK_OK;
}

extern "C" int sdfk_dist_init(int32_t world, int32_t rank, const void* id)
{
    if (!id) return fail(SDFK_ERR_INVALID, "sdfk_dist_init: null id");
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (int r = dist_prepare_rccl(world, rank)) return r;
    return dist_join_rccl(id);
}

extern "C" int sdfk_dist_init_host(int32_t world, int32_t rank, sdfk_allgather_fn allgather, void* ctx)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!allgather) return fail(SDFK_ERR_INVALID, "sdfk_dist_init_host: null transport");
    if (int r = dist_common_init(world, rank)) return r;
    gd.host_fn = allgather;
    gd.host_ctx = ctx;
    gd.backend = 2;
    return SDFK_OK;
}

extern "C" int sdfk_dist_info(int32_t* world, int32_t* rank, int32_t* backend)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (world) *world = gd.world;
    if (rank) *rank = gd.rank;
    if (backend) *backend = gd.backend;
    return SDFK_OK;
}

extern "C" void sdfk_dist_shutdown(void)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!gd.backend || !g.inited) return;
    dist_release();
}

extern "C" int sdfk_dist_slab(int32_t nz, int32_t world, int32_t rank, int32_t* layer_begin, int32_t* layer_end, int32_t* z0, int32_t* nz_local)
{
    if (nz < 1 || world < 1 || rank < 0 || rank >= world) return fail(SDFK_ERR_INVALID, "sdfk_dist_slab: bad argument");
    int lb, le, a, n;
    sdfk::slab_layers(nz - 1, world, rank, &lb, &le);
    sdfk::slab_planes(lb, le, nz, &a, &n);
    if (layer_begin) *layer_begin = lb;
    if (layer_end) *layer_end = le;
    if (z0) *z0 = a;
    if (nz_local) *nz_local = n;
    return SDFK_OK;
}

extern "C" void sdfk_dist_session_free(sdfk_dist_session* s)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    bind_thread();
    if (!s) return;
    if (g.inited) {
        (void)s->proto.drain();
        (void)s->quiesce();
        for (auto& sl : s->slots) {
            if (sl.exact) sdfk_mesh_free(sl.exact);
            sl.exact = nullptr;
        }
        s->free_buffers();
        for (auto& sl : s->slots) {
            if (sl.vol) sdfk_volume_free(sl.vol);
            for (hipEvent_t e : {sl.packed, sl.ready, sl.read})
                if (e) (void)hipEventDestroy(e);
        }
        if (s->prog) program_release(s->prog);
    }
    gd.sessions = std::max(0, gd.sessions - 1);
    delete s;
}

extern "C" int sdfk_dist_session_create(const sdfk_program* p, const float min[3], const float max[3], int32_t nx, int32_t ny, int32_t nz,
                                        int32_t clip_to_bounds, float iso_value, int32_t depth, sdfk_dist_session** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!p || !min || !max || !out) return fail(SDFK_ERR_INVALID, "sdfk_dist_session_create: null argument");
    *out = nullptr;
    if (int r = require_init()) return r;
    if (!gd.backend) return fail(SDFK_ERR_INVALID, "sdfk_dist_session_create: call sdfk_dist_init first");
    if (depth < 1 || depth > 8) return fail(SDFK_ERR_INVALID, "sdfk_dist_session_create: depth %d outside 1..8", depth);
    if (nx < 1 || ny < 1 || nz < 1) return fail(SDFK_ERR_INVALID, "sdfk_dist_session_create: bad grid %dx%dx%d", nx, ny, nz);
    sdfk_dist_session* s = new sdfk_dist_session(depth);
    s->prog = const_cast<sdfk_program*>(p);
    s->prog->refs++;
    memcpy(s->mn, min, 12);
    memcpy(s->mx, max, 12);
    s->nx = nx; s->ny = ny; s->nz = nz; s->clip = clip_to_bounds ? 1 : 0; s->iso = iso_value;
    sdfk::slab_layers(nz - 1, gd.world, gd.rank, &s->lb, &s->le);
    sdfk::slab_planes(s->lb, s->le, nz, &s->z0, &s->nzl);
    s->vbytes = p->writes_color ? 36 : 24;
    s->exchange_mode = g_cfg.dist_exchange;
    s->lanes = g_cfg.dist_lanes;
    s->idx16 = g_cfg.dist_index16 != 0;
    gd.sessions++;
    int r = SDFK_OK;
    for (auto& sl : s->slots) {
        r = r ? r : sdfk_volume_create_slab(nx, ny, nz, min, max, s->z0, std::max(s->nzl, 1), p->writes_color ? 1 : 0, &sl.vol);
        for (hipEvent_t* e : {&sl.packed, &sl.ready, &sl.read})
            if (!r && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) r = fail(SDFK_ERR_HIP, "hipEventCreate failed");
    }
    if (r) { sdfk_dist_session_free(s); return r; }
    *out = s;
    return SDFK_OK;
}

int dist_fail(sdfk_dist_session* s, int r)
{
    t_err = s->proto.error().empty() ? s->err : s->proto.error();
    return r == 1 ? SDFK_ERR_INVALID : r;   // (1: a misuse the protocol itself reports)
}

extern "C" int sdfk_dist_submit(sdfk_dist_session* s)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!s) return fail(SDFK_ERR_INVALID, "sdfk_dist_submit: null session");
    if (int r = require_init()) return r;
    const auto t0 = std::chrono::steady_clock::now();
    const int r = s->proto.submit();
    s->host_ns_submit += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    s->steps++;
    return r ? dist_fail(s, r) : SDFK_OK;
}

extern "C" int sdfk_dist_collect(sdfk_dist_session* s, int64_t* n_vertices_mine, int64_t* n_indices_mine)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!s) return fail(SDFK_ERR_INVALID, "sdfk_dist_collect: null session");
    if (int r = require_init()) return r;
    const auto t0 = std::chrono::steady_clock::now();
    const int r = s->proto.collect(n_vertices_mine, n_indices_mine);
    s->host_ns_collect += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
    return r ? dist_fail(s, r) : SDFK_OK;
}

extern "C" int sdfk_dist_counts(const sdfk_dist_session* s, int64_t* counts)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!s || !counts) return fail(SDFK_ERR_INVALID, "sdfk_dist_counts: null argument");
    const int64_t* h = s->proto.last_headers();
    if (!h) return fail(SDFK_ERR_INVALID, "sdfk_dist_counts: no collected step (or its slot has been resubmitted)");
    for (int q = 0; q < gd.world; q++) {
        counts[2 * q] = h[q * sdfk::kSlabHeaderWords];
        counts[2 * q + 1] = h[q * sdfk::kSlabHeaderWords + 1];
    }
    return SDFK_OK;
}

extern "C" int sdfk_dist_gathered(const sdfk_dist_session* s, void** device_ptr, int64_t* stride_bytes)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!s) return fail(SDFK_ERR_INVALID, "sdfk_dist_gathered: null session");
    const int k = s->proto.last_slot();
    if (k < 0) return fail(SDFK_ERR_INVALID, "sdfk_dist_gathered: no collected step (or its slot has been resubmitted)");
    // headers only (mode 3; mode 2 on a rank other than 0): the foreign sections hold this step's 64-byte header followed by stale
    // bytes -- a consumer that decoded them would get a wrong mesh without noticing
    if (!s->slots[k].payloads_gathered)
        return fail(SDFK_ERR_UNSUPPORTED, "sdfk_dist_gathered: this rank received the headers only (SDFK_OPT_DIST_EXCHANGE = %d): "
                                          "sdfk_dist_slab_mesh gives its own slab, sdfk_dist_mesh gathers the payloads on demand", s->exchange_mode);
    if (device_ptr) *device_ptr = s->slots[k].gathered;
    if (stride_bytes) *stride_bytes = s->stride;
    return SDFK_OK;
}

extern "C" int sdfk_dist_stats(const sdfk_dist_session* s, int64_t stats[8])
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!s || !stats) return fail(SDFK_ERR_INVALID, "sdfk_dist_stats: null argument");
    stats[0] = s->stride; stats[1] = s->steps; stats[2] = s->proto.redone(); stats[3] = s->proto.grown();
    stats[4] = gd.backend == 2 ? -1 : s->exchange_mode; stats[5] = s->host_ns_submit; stats[6] = s->host_ns_collect;
    stats[7] = s->proto.depth() | (s->idx16 ? 0x100 : 0) | (s->idx16_fallbacks << 16);
    return SDFK_OK;
}

extern "C" int sdfk_dist_enqueue_only(sdfk_dist_session* s)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!s) return fail(SDFK_ERR_INVALID, "sdfk_dist_enqueue_only: null session");
    if (int r = require_init()) return r;
    if (s->proto.in_flight() != 0 || s->stride == 0) return fail(SDFK_ERR_INVALID, "sdfk_dist_enqueue_only: collect every queued step first (and run one)");
    if (int r = s->enqueue(0)) { t_err = s->err; return r; }
    return SDFK_OK;
}

// Which exchange is faster on THIS node's fabric is a measurement, not a constant: RCCL's all-gather (rings / trees over
// the xGMI mesh, its own protocol choice per size) against direct grouped sends to every peer.  Runs `steps_per_mode`
// pipelined steps with each, takes the slowest rank's time per mode (agree_max: the same numbers on every rank), keeps the
// faster mode for the session.  Collective; nothing may be in flight.  ns_per_mode[2] (may be NULL) = the agreed times.
extern "C" int sdfk_dist_tune(sdfk_dist_session* s, int32_t steps_per_mode, int64_t* ns_per_config)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!s || steps_per_mode < 1) return fail(SDFK_ERR_INVALID, "sdfk_dist_tune: bad argument");
    if (int r = require_init()) return r;
    if (s->proto.in_flight() != 0) return fail(SDFK_ERR_INVALID, "sdfk_dist_tune: collect every queued step first");
    if (ns_per_config) for (int k = 0; k < 4; k++) ns_per_config[k] = 0;
    if (gd.backend != 1) return SDFK_OK;   // the host transport has one exchange only
    // the tuner chooses between exchanges that leave the WHOLE mesh on every rank (0 and 1); a session that keeps the mesh sharded
    // (3) or gathers it to rank 0 (2) has another contract, which a measurement must not replace behind the caller's back
    if (s->exchange_mode >= 2)
        return fail(SDFK_ERR_UNSUPPORTED, "sdfk_dist_tune: the session's exchange mode %d is a contract (who holds the mesh), not a candidate: "
                                          "tune a session created with SDFK_OPT_DIST_EXCHANGE = 0 or 1", s->exchange_mode);
    const int mode_before = s->exchange_mode;
    auto run = [&](int n) {
        for (int i = 0; i < n; i++) {
            if (s->proto.in_flight() == s->proto.depth())
                if (int r = s->proto.collect(nullptr, nullptr)) return r;
            if (int r = s->proto.submit()) return r;
        }
        return s->proto.drain();
    };
    auto set_form = [&](bool idx) -> int {   // the payload form belongs to the buffers: a new bootstrap (every rank alike)
        if (s->idx16 == idx) return SDFK_OK;
        if (int r = s->quiesce()) return r;
        s->idx16 = idx;
        if (s->proto.reset()) { s->err = s->proto.error(); return SDFK_ERR_INVALID; }
        return SDFK_OK;
    };
    const int64_t kNever = INT64_MAX;
    int64_t agreed[4] = {kNever, kNever, kNever, kNever};   // index = mode + 2 * (16-bit indices)
    for (int idx = 0; idx < 2; idx++) {
        if (int r = set_form(idx != 0)) { s->exchange_mode = mode_before; return dist_fail(s, r); }
        for (int mode = 1; mode >= 0; mode--) {
            s->exchange_mode = mode;
            int r = run(2 * s->proto.depth() + 2);   // (captured step graphs of every slot and lane exist after this)
            if (!r) r = s->quiesce();
            const auto t0 = std::chrono::steady_clock::now();
            if (!r) r = run(steps_per_mode);
            if (!r) r = s->quiesce();
            const int64_t ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            if (r) { s->exchange_mode = mode_before; return dist_fail(s, r); }
            if (int r2 = s->agree_max(ns, &agreed[mode + 2 * idx])) { s->exchange_mode = mode_before; t_err = s->err; return r2; }
            // (a slab that does not fit 16-bit offsets sent the session back to int32 indices -- on every rank, seen in the
            // same step's headers: the compact form is not a candidate for this scene)
            if (idx && !s->idx16) agreed[mode + 2 * idx] = kNever;
        }
    }
    int best = 0;   // (ties go to the default: ncclAllGather, plain payloads)
    for (int k = 0; k < 4; k++)
        if (agreed[k] < agreed[best]) best = k;
    if (int r = set_form(best >= 2)) return dist_fail(s, r);
    s->exchange_mode = best & 1;
    if (ns_per_config) for (int k = 0; k < 4; k++) ns_per_config[k] = agreed[k] == kNever ? -1 : agreed[k];
    return SDFK_OK;
}

extern "C" int sdfk_dist_mesh(sdfk_dist_session* s, sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!s || !out) return fail(SDFK_ERR_INVALID, "sdfk_dist_mesh: null argument");
    *out = nullptr;
    if (int r = require_init()) return r;
    const int k = s->proto.last_slot();
    const int64_t* h = s->proto.last_headers();
    if (k < 0 || !h) return fail(SDFK_ERR_INVALID, "sdfk_dist_mesh: no collected step (or its slot has been resubmitted)");
    if (s->exchange_mode == 2 && gd.rank != 0 && gd.world > 1)
        return fail(SDFK_ERR_UNSUPPORTED, "sdfk_dist_mesh: with SDFK_OPT_DIST_EXCHANGE = 2 only rank 0 holds the mesh");
    int64_t nv = 0, ni = 0;
    float bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
    bool any = false;
    for (int q = 0; q < gd.world; q++) {
        const int64_t* hq = h + q * sdfk::kSlabHeaderWords;
        if (hq[0] < 0 || hq[1] < 0) return fail(SDFK_ERR_INVALID, "sdfk_dist_mesh: the collected step has an unresolved slab");
        nv += hq[0];
        ni += hq[1];
        const float* b = reinterpret_cast<const float*>(hq + 2);
        if (hq[0] > 0) {
            for (int c = 0; c < 3; c++) {
                bmin[c] = any ? std::min(bmin[c], b[c]) : b[c];
                bmax[c] = any ? std::max(bmax[c], b[3 + c]) : b[3 + c];
            }
            any = true;
        }
    }
    if (nv >= (int64_t(1) << 31)) return fail(SDFK_ERR_UNSUPPORTED, "vertex index exceeds int32 (Mesh.Triangles is int[])");
    sdfk_dist_session::Slot& sl = s->slots[k];
    if (!sl.payloads_gathered) {
        // exchange mode 3: the step moved the headers only.  The whole mesh is asked for now: the payload exchange of THIS step
        // (collective: every rank makes this call), then the usual extraction
        int r = s->exchange_as(k, 0);
        if (!r && hipEventSynchronize(sl.ready) != hipSuccess) r = fail(SDFK_ERR_HIP, "sdfk_dist_mesh: waiting for the payload exchange failed");
        if (r) { if (!s->err.empty()) t_err = s->err; return r; }
    }
    sdfk_mesh* m = nullptr;
    if (int r = alloc_mesh(&m, (size_t)nv, (size_t)ni)) return r;
    ConcatArgs A{sl.gathered, gd.world, s->stride, m->vertices, m->colors, m->normals, m->triangles, m->bounds,
                 (s->idx16 && ni <= sl.decoded_cap) ? sl.decoded : nullptr};
    // (the slot's exchange has completed: collect waited for its `ready` event)
    const int64_t words = 9 * nv + ni;
    hipLaunchKernelGGL(k_slabs_concat, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((words / gd.world + 1023) / 1024, 512)), gd.world), dim3(256), 0,
                       g.stream, A);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(sl.read, g.stream);
    if (e != hipSuccess) { sdfk_mesh_free(m); return fail(SDFK_ERR_HIP, "sdfk_dist_mesh: %s", hipGetErrorString(e)); }
    sl.read_valid = true;
    m->nv = nv; m->ni = ni;
    memcpy(m->h_min, bmin, 12);
    memcpy(m->h_max, bmax, 12);
    m->bounds_valid = true;
    m->has_colors = s->vbytes == 36;
    *out = m;
    return SDFK_OK;
}

// This rank's OWN slab of the step collected last as a mesh of its own, indices global (+ the vertex counts of the slabs before it,
// from the gathered headers): what a host assembles the whole mesh from without any payload exchange (exchange mode 3) -- every
// rank copies its slab to the host over its own PCIe link, to the offsets sdfk_dist_counts gives.  Not collective.
extern "C" int sdfk_dist_slab_mesh(sdfk_dist_session* s, sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!s || !out) return fail(SDFK_ERR_INVALID, "sdfk_dist_slab_mesh: null argument");
    *out = nullptr;
    if (int r = require_init()) return r;
    const int k = s->proto.last_slot();
    const int64_t* h = s->proto.last_headers();
    if (k < 0 || !h) return fail(SDFK_ERR_INVALID, "sdfk_dist_slab_mesh: no collected step (or its slot has been resubmitted)");
    int64_t vbase = 0;
    for (int q = 0; q < gd.world; q++) {
        const int64_t* hq = h + q * sdfk::kSlabHeaderWords;
        if (hq[0] < 0 || hq[1] < 0) return fail(SDFK_ERR_INVALID, "sdfk_dist_slab_mesh: the collected step has an unresolved slab");
        if (q < gd.rank) vbase += hq[0];
    }
    const int64_t* me = h + gd.rank * sdfk::kSlabHeaderWords;
    const int64_t nv = me[0], ni = me[1];
    if (vbase + nv >= (int64_t(1) << 31)) return fail(SDFK_ERR_UNSUPPORTED, "vertex index exceeds int32 (Mesh.Triangles is int[])");
    sdfk_mesh* m = nullptr;
    if (int r = alloc_mesh(&m, (size_t)nv, (size_t)ni)) return r;
    sdfk_dist_session::Slot& sl = s->slots[k];
    // (indices in this rank's section: still slab-local after a headers-only step -- mode 3, or mode 2 on a rank other than 0 -- and
    // in the compact form; rebased in place otherwise: the slot remembers whether the rebase kernel ran over THIS section)
    const bool local_ids = !sl.own_rebased;
    SlabExtractArgs A{s->send_buf(sl), m->vertices, m->colors, m->normals, m->triangles, m->bounds, local_ids ? (int32_t)vbase : 0};
    const int64_t words = 9 * nv + ni;
    hipLaunchKernelGGL(k_slab_extract, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((words + 1023) / 1024, 1024))), dim3(256), 0, g.stream, A);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipEventRecord(sl.read, g.stream);
    if (e != hipSuccess) { sdfk_mesh_free(m); return fail(SDFK_ERR_HIP, "sdfk_dist_slab_mesh: %s", hipGetErrorString(e)); }
    sl.read_valid = true;
    m->nv = nv; m->ni = ni;
    const float* b = reinterpret_cast<const float*>(me + 2);
    if (nv > 0) { memcpy(m->h_min, b, 12); memcpy(m->h_max, b + 3, 12); }
    m->bounds_valid = true;
    m->has_colors = s->vbytes == 36;
    *out = m;
    return SDFK_OK;
}

extern "C" int sdfk_dist_to_mesh(const sdfk_program* p, const float min[3], const float max[3], int32_t nx, int32_t ny, int32_t nz,
                                 int32_t clip_to_bounds, float iso_value, sdfk_mesh** out)
{
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!out) return fail(SDFK_ERR_INVALID, "sdfk_dist_to_mesh: null argument");
    *out = nullptr;
    sdfk_dist_session* s = nullptr;
    int r = sdfk_dist_session_create(p, min, max, nx, ny, nz, clip_to_bounds, iso_value, 1, &s);
    if (r) return r;
    r = sdfk_dist_submit(s);
    if (!r) r = sdfk_dist_collect(s, nullptr, nullptr);
    if (!r) r = sdfk_dist_mesh(s, out);
    if (!r && hipStreamSynchronize(g.stream) != hipSuccess) r = fail(SDFK_ERR_HIP, "sdfk_dist_to_mesh: synchronisation failed");
    const std::string keep_err = t_err;
    sdfk_dist_session_free(s);
    if (r) { if (*out) sdfk_mesh_free(*out); *out = nullptr; t_err = keep_err; }
    return r;
}
