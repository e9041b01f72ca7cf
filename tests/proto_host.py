"""Test harness: the product's C++ step protocol (sdfkit_amd/csrc/slab_protocol.h, sdfk::SlabProtocol -- the class
libsdfkit_hip.so drives with HIP + RCCL) built with g++ into a small shim (tests/cpp/protocol_host.cpp) and driven from
Python callbacks: workers that serve slices of an oracle mesh, a torch.distributed (gloo) all-gather as the transport.
No GPU, no HIP: what runs here is the protocol code itself."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None

_i32, _i64 = C.c_int32, C.c_int64
_p64 = C.POINTER(C.c_int64)


class Callbacks(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("world", _i32), ("rank", _i32),
                ("run_exact", C.CFUNCTYPE(C.c_int, C.c_void_p, _i32, _p64, _p64, _p64)),
                ("agree_max", C.CFUNCTYPE(C.c_int, C.c_void_p, _i64, _p64)),
                ("resize", C.CFUNCTYPE(C.c_int, C.c_void_p, _i64)),
                ("pack_exact", C.CFUNCTYPE(C.c_int, C.c_void_p, _i32)),
                ("enqueue", C.CFUNCTYPE(C.c_int, C.c_void_p, _i32)),
                ("exchange", C.CFUNCTYPE(C.c_int, C.c_void_p, _i32)),
                ("headers", C.CFUNCTYPE(C.c_int, C.c_void_p, _i32, C.POINTER(_p64))),
                ("quiesce", C.CFUNCTYPE(C.c_int, C.c_void_p)),
                ("consensus", C.CFUNCTYPE(C.c_int, C.c_void_p, _i32))]


def lib():
    """Builds (once per source change) and loads tests/cpp/_build/libproto_host.so."""
    global _lib
    if _lib is None:
        import fcntl
        src = os.path.join(ROOT, "tests", "cpp", "protocol_host.cpp")
        hdr = os.path.join(ROOT, "sdfkit_amd", "csrc", "slab_protocol.h")
        out_dir = os.path.join(ROOT, "tests", "cpp", "_build")
        os.makedirs(out_dir, exist_ok=True)
        so = os.path.join(out_dir, "libproto_host.so")
        with open(so + ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
                tmp = f"{so}.tmp.{os.getpid()}"
                subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-fPIC", "-shared", src, "-o", tmp])
                os.replace(tmp, so)
        L = C.CDLL(so)
        L.proto_create.restype = C.c_void_p
        L.proto_create.argtypes = [C.POINTER(Callbacks), _i32, C.c_double]
        L.proto_free.argtypes = [C.c_void_p]
        L.proto_free.restype = None
        for name in ("proto_submit", "proto_drain", "proto_reset", "proto_in_flight", "proto_last_slot"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = C.c_int
        L.proto_collect.argtypes = [C.c_void_p, _p64, _p64]
        for name in ("proto_stride", "proto_redone", "proto_grown"):
            getattr(L, name).argtypes = [C.c_void_p]
            getattr(L, name).restype = _i64
        L.proto_error.argtypes = [C.c_void_p]
        L.proto_error.restype = C.c_char_p
        L.proto_slab_layers.argtypes = [_i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]
        L.proto_slab_layers.restype = None
        L.proto_slab_planes.argtypes = [_i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32)]
        L.proto_slab_planes.restype = None
        _lib = L
    return _lib


def slab_layers(n_layers, world, rank):
    a, b = _i32(), _i32()
    lib().proto_slab_layers(n_layers, world, rank, C.byref(a), C.byref(b))
    return a.value, b.value


def slab_planes(lb, le, nz):
    a, b = _i32(), _i32()
    lib().proto_slab_planes(lb, le, nz, C.byref(a), C.byref(b))
    return a.value, b.value


def rebase_host(gathered, world, stride):
    """numpy twin of k_slabs_rebase (test fixture only): indices of slab r += vertices of slabs 0..r-1."""
    g = gathered.numpy()
    nvs = [int(g[r, :8].view(np.int64)[0]) for r in range(world)]
    if min(nvs) < 0:
        return
    base = 0
    for r in range(world):
        ni = int(g[r, 8:16].view(np.int64)[0])
        o = 64 + int(g[r, 40:44].view(np.int32)[0]) * nvs[r]
        if o + 4 * ni <= stride and base:
            g[r, o:o + 4 * ni].view(np.int32)[:] += base
        base += nvs[r]


class ProtoSession:
    """sdfk::SlabProtocol over `make_worker(slot)` workers (run_local() -> (nv, ni); pack_self_describing(buf);
    enqueue(buf); vertex_bytes) and the default process group's all_gather."""

    def __init__(self, make_worker, depth, group=None, headroom=0.125, agree_on_failures=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.depth = depth
        self.workers = [make_worker(k) for k in range(depth)]
        self.buf, self.gathered, self.hdr = [None] * depth, [None] * depth, [None] * depth
        self.stride = 0
        self.log = []
        self.error = None

        def guard(fn):
            def wrapped(*a):
                try:
                    return fn(*a) or 0
                except Exception as e:   # (never unwind through the C++ caller)
                    import traceback
                    traceback.print_exc()
                    self.error = e
                    return 3
            return wrapped

        def run_exact(ctx, slot, nv, ni, need):
            a, b = self.workers[slot].run_local()
            nv[0], ni[0] = a, b
            need[0] = 64 + getattr(self.workers[slot], "vertex_bytes", 36) * a + 4 * b
            self.log.append(("exact", slot))

        def agree_max(ctx, mine, out):
            t = torch.tensor([mine], dtype=torch.int64)
            parts = [torch.empty_like(t) for _ in range(self.world)]
            dist.all_gather(parts, t, group=group)
            out[0] = max(int(x.item()) for x in parts)

        def resize(ctx, stride):
            self.stride = stride
            for k in range(depth):
                self.buf[k] = torch.zeros(stride, dtype=torch.uint8)
                self.gathered[k] = torch.zeros((self.world, stride), dtype=torch.uint8)
                self.hdr[k] = np.zeros((self.world, 8), np.int64)
            self.log.append(("resize", stride))

        def pack_exact(ctx, slot):
            self.workers[slot].pack_self_describing(self.buf[slot])

        def enqueue(ctx, slot):
            self.workers[slot].enqueue(self.buf[slot])
            self.log.append(("enqueue", slot))

        def exchange(ctx, slot):
            g = self.gathered[slot]
            dist.all_gather([g[r] for r in range(self.world)], self.buf[slot], group=group)
            rebase_host(g, self.world, self.stride)
            self.hdr[slot][:] = g[:, :64].contiguous().numpy().view(np.int64)

        def headers(ctx, slot, out):
            out[0] = self.hdr[slot].ctypes.data_as(_p64)

        def quiesce(ctx):
            pass

        def consensus(ctx, mine):
            # SlabOps::consensus with a backend that CAN agree (what a node's rank threads do with a thread barrier): the
            # largest status over the ranks, so that a rank-local failure takes every rank out of the step before the collective
            t = torch.tensor([mine], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            self.log.append(("consensus", int(mine), int(t.item())))
            return int(t.item())

        self.cb = Callbacks()
        self.cb.ctx, self.cb.world, self.cb.rank = None, self.world, self.rank
        fns = dict(run_exact=run_exact, agree_max=agree_max, resize=resize, pack_exact=pack_exact, enqueue=enqueue,
                   exchange=exchange, headers=headers, quiesce=quiesce, consensus=consensus)
        for name, ftype in Callbacks._fields_[3:]:
            if name == "consensus" and not agree_on_failures:
                continue                      # (NULL: the protocol's default, no agreement)
            setattr(self.cb, name, ftype(guard(fns[name])))
        self.L = lib()
        self.h = C.c_void_p(self.L.proto_create(C.byref(self.cb), depth, headroom))

    def _check(self, r):
        if r:
            raise RuntimeError(f"protocol error {r}: {self.L.proto_error(self.h).decode()} ({self.error!r})")

    def submit(self):
        self._check(self.L.proto_submit(self.h))

    def collect(self):
        a, b = _i64(), _i64()
        self._check(self.L.proto_collect(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def reset(self):
        """Forget the agreed stride (nothing in flight): the next submit bootstraps again."""
        self._check(self.L.proto_reset(self.h))

    @property
    def agreed_stride(self):
        return int(self.L.proto_stride(self.h))

    @property
    def in_flight(self):
        return self.L.proto_in_flight(self.h)

    @property
    def redone(self):
        return int(self.L.proto_redone(self.h))

    @property
    def grown(self):
        return int(self.L.proto_grown(self.h))

    def mesh(self):
        """Arrays of the step collected last (rebased gather buffer of its slot)."""
        from sdfkit_amd import dist as D
        slot = self.L.proto_last_slot(self.h)
        assert slot >= 0
        return D.unpack_self_describing(self.gathered[slot].numpy())

    def close(self):
        if self.h:
            self._check(self.L.proto_drain(self.h))
            self.L.proto_free(self.h)
            self.h = None
