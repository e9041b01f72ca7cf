// protocol_host.cpp -- test shim (g++ -shared, no HIP): the product's sharded-step protocol, sdfk::SlabProtocol
// (sdfkit_amd/csrc/slab_protocol.h, the class libsdfkit_hip.so drives with HIP kernels + RCCL), driven with callbacks so
// that tests/test_dist_gloo.py can run it on CPU: workers built from oracle fixtures, a gloo all-gather as the transport.
// What is under test is the protocol code itself -- slot rotation, bootstrap, stride agreement and regrowth, "a rank's
// guess was too small: everybody redoes the step", matched collectives -- not a Python re-statement of it.
#include <stdint.h>

#include <string>

#include "../../sdfkit_amd/csrc/slab_protocol.h"

extern "C" {

struct proto_callbacks {
    void* ctx;
    int32_t world, rank;
    int (*run_exact)(void* ctx, int32_t slot, int64_t* nv, int64_t* ni, int64_t* need_bytes);
    int (*agree_max)(void* ctx, int64_t mine, int64_t* max_all);
    int (*resize)(void* ctx, int64_t stride);
    int (*pack_exact)(void* ctx, int32_t slot);
    int (*enqueue)(void* ctx, int32_t slot);
    int (*exchange)(void* ctx, int32_t slot);
    int (*headers)(void* ctx, int32_t slot, const int64_t** hdr);
    int (*quiesce)(void* ctx);
    int (*consensus)(void* ctx, int32_t mine);   // may be NULL: SlabOps's default (no agreement)
};

}  // extern "C"

namespace {

struct CallbackOps final : sdfk::SlabOps {
    proto_callbacks cb;
    std::string err = "a callback failed";
    int world() const override { return cb.world; }
    int rank() const override { return cb.rank; }
    int run_exact(int k, int64_t* nv, int64_t* ni, int64_t* need) override { return cb.run_exact(cb.ctx, k, nv, ni, need); }
    int agree_max(int64_t mine, int64_t* mx) override { return cb.agree_max(cb.ctx, mine, mx); }
    int resize(int64_t stride) override { return cb.resize(cb.ctx, stride); }
    int pack_exact(int k) override { return cb.pack_exact(cb.ctx, k); }
    int enqueue(int k) override { return cb.enqueue(cb.ctx, k); }
    int exchange(int k) override { return cb.exchange(cb.ctx, k); }
    int headers(int k, const int64_t** h) override { return cb.headers(cb.ctx, k, h); }
    int quiesce() override { return cb.quiesce(cb.ctx); }
    int consensus(int mine) override { return cb.consensus ? cb.consensus(cb.ctx, mine) : mine; }
    const char* last_error() const override { return err.c_str(); }
};

struct Proto {
    CallbackOps ops;
    sdfk::SlabProtocol p;
    Proto(const proto_callbacks& cb, int depth, double headroom) : p(&ops, depth, headroom) { ops.cb = cb; }
};

}  // namespace

extern "C" {

void* proto_create(const proto_callbacks* cb, int32_t depth, double headroom) { return new Proto(*cb, depth, headroom); }
void proto_free(void* h) { delete static_cast<Proto*>(h); }
int proto_submit(void* h) { return static_cast<Proto*>(h)->p.submit(); }
int proto_collect(void* h, int64_t* nv, int64_t* ni) { return static_cast<Proto*>(h)->p.collect(nv, ni); }
int proto_drain(void* h) { return static_cast<Proto*>(h)->p.drain(); }
int proto_reset(void* h) { return static_cast<Proto*>(h)->p.reset(); }
int proto_in_flight(void* h) { return static_cast<Proto*>(h)->p.in_flight(); }
int proto_last_slot(void* h) { return static_cast<Proto*>(h)->p.last_slot(); }
int64_t proto_stride(void* h) { return static_cast<Proto*>(h)->p.stride(); }
int64_t proto_redone(void* h) { return static_cast<Proto*>(h)->p.redone(); }
int64_t proto_grown(void* h) { return static_cast<Proto*>(h)->p.grown(); }
const char* proto_error(void* h) { return static_cast<Proto*>(h)->p.error().c_str(); }
void proto_slab_layers(int32_t n_layers, int32_t world, int32_t rank, int32_t* lb, int32_t* le) { sdfk::slab_layers(n_layers, world, rank, lb, le); }
void proto_slab_planes(int32_t lb, int32_t le, int32_t nz, int32_t* z0, int32_t* n) { sdfk::slab_planes(lb, le, nz, z0, n); }

}  // extern "C"
