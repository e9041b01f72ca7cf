// slab_protocol.h -- the per-rank driver of the Z-slab sharded step (the reference has no distributed
// path: this serves SdfEx.ToMesh, Sdf.cs:59-63, over the GPUs of one node).  Plain C++, no HIP: what a rank computes and
// how bytes travel is behind SlabOps, so the SAME protocol code runs in libsdfkit_hip.so (HIP kernels + RCCL, dist_rccl.h)
// and in the CPU tests (tests/cpp/protocol_host.cpp: fixture workers + a gloo all-gather).
//
// A step, per rank, is queued without waiting for the GPU: sample + mesh the slab with slab-local ids into the rank's
// section of the gather buffer (buffers sized from the previous step), ONE exchange of the padded payloads, one kernel
// that rebases the gathered indices from the headers and mirrors the `world` 64-byte headers to the host.  submit() queues
// a step into the next of `depth` slots; collect() waits for the OLDEST queued step only and returns this rank's counts.
//
// Decisions every rank must take identically (the collectives have to stay matched) are taken from data every rank
// sees: the gathered headers, or the result of agree_max().
//   * bootstrap (first step): exact, synchronising path; the payload stride = max over ranks of the bytes needed
//     (+ head-room) -- one agree_max;
//   * a step whose speculative capacities were too small on ANY rank carries nv = ni = -1 in that rank's header: every
//     rank sees it after the exchange and all of them redo that step on the exact path;
//   * a payload that outgrew the stride shows in its header too (the counts say what it needs): same redo, and the exact
//     path re-agrees the stride (every exact step ends with an agree_max), growing every slot's buffers when needed.
#pragma once
#include <stdint.h>

#include <deque>
#include <string>

namespace sdfk {

constexpr int kSlabHeaderBytes = 64;   // SDFK_SLAB_HEADER_BYTES
constexpr int kSlabHeaderWords = 8;    // int64 words per header: nv, ni, 3 x (bounds), vbytes | cap_v << 32, idx_bits | flags << 32, pad

struct SlabOps {
    virtual ~SlabOps() {}
    virtual int world() const = 0;
    virtual int rank() const = 0;
    // Exact, synchronising form of this rank's step into slot's worker: counts and the payload bytes they need.
    virtual int run_exact(int slot, int64_t* nv, int64_t* ni, int64_t* need_bytes) = 0;
    // Collective: max of `mine` over the ranks.
    virtual int agree_max(int64_t mine, int64_t* max_all) = 0;
    // (Re)allocate every slot's send / gather buffers for `stride` bytes per rank; nothing is in flight.
    virtual int resize(int64_t stride) = 0;
    // The exact mesh of run_exact(slot) -> slot's send buffer (dense sections).
    virtual int pack_exact(int slot) = 0;
    // Speculative step into slot's send buffer, asynchronous; the payload header will carry the counts (or -1, -1).
    virtual int enqueue(int slot) = 0;
    // Asynchronous: all ranks' send buffers -> slot's gather buffer, indices rebased, headers on their way to the host.
    virtual int exchange(int slot) = 0;
    // Waits for slot's exchange; *hdr = world x kSlabHeaderWords int64 (valid until the slot is resubmitted).
    virtual int headers(int slot, const int64_t** hdr) = 0;
    // Waits for everything this rank has queued.
    virtual int quiesce() = 0;
    virtual const char* last_error() const = 0;
    // A rank-local operation (run_exact, resize, pack_exact, enqueue: allocations, launches) has returned `mine` and a COLLECTIVE
    // comes next.  A rank that failed returns from the step; the others would wait in the collective for ever.  A backend that can
    // agree cheaply (ranks that are threads of one process: dist_rccl.h + node_local.h) makes every rank return the first failing
    // rank's status here, so that all leave the step together; the default -- one process per GPU, where an agreement would be a
    // host-synchronising collective in every pipelined step and a dead rank is the launcher's business -- is no agreement.
    // Called by every rank at the same points of the protocol, in the same order.
    virtual int consensus(int mine) { return mine; }
    // Bytes the payload described by one gathered header needs (the protocol compares it with the stride).  Default: the
    // plain layout, header + vertex_bytes per vertex + 4 bytes per index; a backend with another encoding overrides it.
    virtual int64_t payload_bytes(const int64_t* hdr) const
    {
        const int64_t vbytes = hdr[5] & 0xffffffffll;
        return kSlabHeaderBytes + vbytes * hdr[0] + 4 * hdr[1];
    }
};

class SlabProtocol {
public:
    SlabProtocol(SlabOps* ops, int depth, double headroom) : ops_(ops), depth_(depth < 1 ? 1 : depth), headroom_(headroom) {}

    int depth() const { return depth_; }
    int in_flight() const { return (int)queue_.size(); }
    int64_t stride() const { return stride_; }
    int64_t redone() const { return redone_; }
    int64_t grown() const { return grown_; }
    int last_slot() const { return last_slot_; }              // slot whose gather buffer holds the step collected last (-1: none)
    const int64_t* last_headers() const { return last_hdr_; } // its headers: world x kSlabHeaderWords
    const std::string& error() const { return err_; }

    // Queue one step into the next slot.  Returns 0, or an error code (message in error()).
    int submit()
    {
        if ((int)queue_.size() == depth_) return fail("all slots are in flight: collect() first");
        const int slot = next_slot_;
        next_slot_ = (slot + 1) % depth_;
        if (last_slot_ == slot) { last_slot_ = -1; last_hdr_ = nullptr; }   // its gather buffer is about to be rewritten
        if (stride_ == 0) {   // bootstrap: sizes, stride, hints (every rank takes this branch together)
            Entry e{slot, true, 0, 0};
            if (int r = exact_step(slot, &e.nv, &e.ni)) return r;
            queue_.push_back(e);
            return 0;
        }
        if (int r = ops_->consensus(ops_->enqueue(slot))) return ops_fail(r);
        if (int r = ops_->exchange(slot)) return ops_fail(r);
        queue_.push_back(Entry{slot, false, 0, 0});
        return 0;
    }

    // Wait for the oldest queued step; this rank's (n_vertices, n_indices).
    int collect(int64_t* nv, int64_t* ni)
    {
        if (queue_.empty()) return fail("collect() without a queued step");
        Entry e = queue_.front();
        queue_.pop_front();
        const int64_t* hdr = nullptr;
        if (int r = ops_->headers(e.slot, &hdr)) return ops_fail(r);
        if (!e.exact) {
            bool redo = false;
            const int w = ops_->world();
            for (int q = 0; q < w && !redo; q++) {
                const int64_t qv = hdr[q * kSlabHeaderWords], qi = hdr[q * kSlabHeaderWords + 1];
                if (qv < 0 || qi < 0) { redo = true; break; }   // some rank's guess was too small
                if (ops_->payload_bytes(hdr + q * kSlabHeaderWords) > stride_) redo = true;   // a payload outgrew the stride
            }
            if (redo) {   // everybody redoes the step exactly (same decision on every rank: same headers)
                redone_++;
                if (int r = exact_step(e.slot, &e.nv, &e.ni)) return r;
                if (int r = ops_->headers(e.slot, &hdr)) return ops_fail(r);
            } else {
                e.nv = hdr[ops_->rank() * kSlabHeaderWords];
                e.ni = hdr[ops_->rank() * kSlabHeaderWords + 1];
            }
        }
        last_slot_ = e.slot;
        last_hdr_ = hdr;
        if (nv) *nv = e.nv;
        if (ni) *ni = e.ni;
        return 0;
    }

    int drain()
    {
        while (!queue_.empty())
            if (int r = collect(nullptr, nullptr)) return r;
        return 0;
    }

    // Forget the agreed stride: the next submit() bootstraps again (exact step, stride agreement, new buffers) -- for a
    // backend that changes its payload form.  Every rank calls it at the same point, with nothing in flight.
    int reset()
    {
        if (!queue_.empty()) return fail("reset() with steps in flight");
        stride_ = 0;
        next_slot_ = 0;
        last_slot_ = -1;
        last_hdr_ = nullptr;
        return 0;
    }

private:
    struct Entry { int slot; bool exact; int64_t nv, ni; };

    int fail(const char* msg) { err_ = msg; return 1; }
    int ops_fail(int r) { err_ = ops_->last_error(); return r; }

    // Synchronous, exact form of a step (first step, and the redo of a failed one).  Ends with the stride agreement:
    // every rank is here together, so the max of the bytes needed is known to all, and all grow their buffers -- or none.
    int exact_step(int slot, int64_t* nv, int64_t* ni)
    {
        // An exact step can itself come back unresolved when the backend has to change its payload ENCODING (16-bit
        // indices that do not fit: flagged in the header, seen by every rank, the backend switches to int32 in headers()):
        // it is then simply done again -- same decision on every rank -- and a second failure is an error.
        for (int attempt = 0;; attempt++) {
            if (int r = exact_step_once(slot, nv, ni)) return r;
            const int64_t* hdr = nullptr;
            if (int r = ops_->headers(slot, &hdr)) return ops_fail(r);
            bool unresolved = false;
            for (int q = 0; q < ops_->world(); q++)
                if (hdr[q * kSlabHeaderWords] < 0 || hdr[q * kSlabHeaderWords + 1] < 0) unresolved = true;
            if (!unresolved) return 0;
            if (attempt == 1) return fail("an exact step came back unresolved twice");
        }
    }

    int exact_step_once(int slot, int64_t* nv, int64_t* ni)
    {
        int64_t need = 0, mx = 0;
        if (int r = ops_->consensus(ops_->run_exact(slot, nv, ni, &need))) return ops_fail(r);
        if (int r = ops_->agree_max(need, &mx)) return ops_fail(r);
        if (mx > stride_) {
            // every rank sends `stride` bytes in every step, used or not: keep the head-room modest
            const int64_t want = ((mx + (int64_t)((double)mx * headroom_) + 4096) + 255) / 256 * 256;
            const bool regrow = stride_ != 0;
            if (regrow) {
                // Steps queued after this one hold buffers of the old stride: they finish (their collectives are matched
                // on every rank), their results are dropped with the buffers ...
                if (int r = ops_->quiesce()) return ops_fail(r);
                grown_++;
                last_slot_ = -1;
                last_hdr_ = nullptr;
            }
            if (int r = ops_->consensus(ops_->resize(want))) {   // (a rank whose own resize succeeded drops its buffers with the others:
                // the next bootstrap allocates them again)
                // no buffers any more (the backend has released what it had): nothing queued can complete into them, and the
                // next submit() must bootstrap -- never enqueue into buffers that do not exist
                stride_ = 0;
                queue_.clear();
                next_slot_ = 0;
                last_slot_ = -1;
                last_hdr_ = nullptr;
                return ops_fail(r);
            }
            stride_ = want;
            // ... and they are queued again -- enqueue + exchange, in submission order -- into the new buffers, here, where
            // every rank does the same (the agreed maximum is what brought them all to this branch)
            if (regrow)
                for (auto& q : queue_) {
                    q.exact = false;
                    if (int r = ops_->consensus(ops_->enqueue(q.slot))) return ops_fail(r);
                    if (int r = ops_->exchange(q.slot)) return ops_fail(r);
                }
        }
        if (int r = ops_->consensus(ops_->pack_exact(slot))) return ops_fail(r);
        if (int r = ops_->exchange(slot)) return ops_fail(r);
        const int64_t* hdr = nullptr;
        if (int r = ops_->headers(slot, &hdr)) return ops_fail(r);   // (the exact path is synchronous)
        return 0;
    }

    SlabOps* ops_;
    int depth_;
    double headroom_;
    int64_t stride_ = 0, redone_ = 0, grown_ = 0;
    int next_slot_ = 0, last_slot_ = -1;
    const int64_t* last_hdr_ = nullptr;
    std::deque<Entry> queue_;
    std::string err_;
};

// ---- partition ---------------------------------------------------------------------------------------------------
// Balanced contiguous split of `n_layers` cell layers of the serial z sweep (MarchingCubes.cs:53-82): rank r owns
// [*lb, *le); the global vertex / triangle order is the concatenation of the slabs in rank order.
inline void slab_layers(int n_layers, int world, int rank, int* lb, int* le)
{
    if (n_layers < 0) n_layers = 0;
    const int base = n_layers / world, rem = n_layers % world;
    *lb = rank * base + (rank < rem ? rank : rem);
    *le = *lb + base + (rank < rem ? 1 : 0);
}

// Voxel planes [*z0, *z0 + *n) a slab holds for layers [lb, le): the context the marching-cubes job asks for
// ([lb-2, le+2) clipped to the grid -- two planes below to recount which vertices layer lb-1 creates, one above for the
// normals of the top face), widened to a multiple of 4 planes where the grid allows (16-byte stores of the sampler).
inline void slab_planes(int lb, int le, int nz, int* z0, int* n)
{
    int a = lb - 2 > 0 ? lb - 2 : 0;
    int b = le + 2 < nz ? le + 2 : nz;
    const int pad = (4 - (b - a) % 4) % 4;
    const int up = pad < nz - b ? pad : nz - b;
    b += up;
    const int down = (pad - up) < a ? (pad - up) : a;
    a -= down;
    *z0 = a;
    *n = b - a;
}

}  // namespace sdfk
