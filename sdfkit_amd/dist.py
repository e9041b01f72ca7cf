"""Z-slab sharding of the sample -> mesh path over the GPUs of one node: the Python binding of sdfk_dist_*.

The reference has no distributed path; this is the multi-GPU form of SdfEx.ToMesh
(Sdf.cs:59-63).  Everything that matters lives behind the C ABI (include/sdfkit_hip.h, "Z-slab sharding";
csrc/slab_protocol.h = the per-rank step protocol, csrc/dist_rccl.h = HIP + RCCL): the partition, the
speculative slab step emitted straight into the gather buffer, the exchange -- the library calls RCCL itself, on
its own stream --, the rebase kernel, the header mirror, the stride agreement and the "some rank's buffers were
too small: everybody redoes this step exactly" rule.  What is left here is what a host has to do in ITS language:
hand the 128-byte RCCL id from rank 0 to the other ranks (torch.distributed's store does that below; a C# host
would use its own launcher), and wrap the handles.

    import torch.distributed as dist          # any process group: it is only used to pass the id around
    from sdfkit_amd import dist as D
    D.init(group=None)                          # RCCL; D.init_host(group) = exchange through the group itself (gloo)
    mesh = D.sharded_to_mesh(sdf, mn, mx, nx, ny, nz)        # one-off: the whole mesh on every rank
    ses = D.SlabSession(sdf, mn, mx, nx, ny, nz, depth=4)    # repeated: up to `depth` steps in flight
"""
import ctypes as C

import numpy as np

from . import _native as N

SLAB_HEADER_BYTES = 64   # SDFK_SLAB_HEADER_BYTES
ID_BYTES = 128           # SDFK_DIST_ID_BYTES

_host_transport = None   # keeps the ctypes callback (and the group it closes over) alive


def slab(nz, world, rank):
    """(layer_begin, layer_end, z0, nz_local) of rank `rank`: sdfk_dist_slab (no device needed)."""
    lb, le, z0, n = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    N.check(N.lib().sdfk_dist_slab(nz, world, rank, C.byref(lb), C.byref(le), C.byref(z0), C.byref(n)))
    return lb.value, le.value, z0.value, n.value


def info():
    """(world, rank, backend) of this process's sharding context; backend 0 = none, 1 = RCCL, 2 = host transport."""
    w, r, b = C.c_int32(), C.c_int32(), C.c_int32()
    N.check(N.lib().sdfk_dist_info(C.byref(w), C.byref(r), C.byref(b)))
    return w.value, r.value, b.value


def init(group=None, device=None):
    """RCCL: rank 0 makes the id, the process group's store hands it round, every rank joins the communicator."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    N.init(device)
    box = [None]
    if rank == 0:
        buf = (C.c_ubyte * ID_BYTES)()
        N.check(N.lib().sdfk_dist_unique_id(buf))
        box[0] = bytes(buf)
    if world > 1:
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    N.check(N.lib().sdfk_dist_init(world, rank, (C.c_ubyte * ID_BYTES).from_buffer_copy(box[0])))


def init_host(group=None, device=None):
    """The exchange goes through `group` itself (any backend that gathers CPU tensors: gloo) -- ranks that share a
    GPU (RCCL refuses two ranks on one device), bring-up, tests.  Same protocol, same kernels."""
    global _host_transport
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    N.init(device)

    def allgather(ctx, send, recv, nbytes):
        try:
            s = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_ubyte)), (nbytes,)))
            r = torch.from_numpy(np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_ubyte)), (world * nbytes,)))
            dist.all_gather([r[q * nbytes:(q + 1) * nbytes] for q in range(world)], s, group=group)
            return 0
        except Exception:   # (an exception must not unwind through the C caller)
            import traceback
            traceback.print_exc()
            return 1

    cb = N.ALLGATHER_FN(allgather)
    N.check(N.lib().sdfk_dist_init_host(world, rank, C.cast(cb, C.c_void_p), None))
    _host_transport = cb


def shutdown():
    global _host_transport
    N.lib().sdfk_dist_shutdown()
    _host_transport = None


def sharded_to_mesh(sdf, mn, mx, nx, ny, nz, clip_to_bounds=True, iso=0.0):
    """SdfEx.ToMesh over all ranks (one GPU each); every rank returns the full Mesh.  Collective."""
    from .api import Mesh
    h = C.c_void_p()
    N.check(N.lib().sdfk_dist_to_mesh(sdf.program(), N.f3(mn), N.f3(mx), nx, ny, nz, 1 if clip_to_bounds else 0,
                                      C.c_float(iso), C.byref(h)))
    return Mesh._from_handle(h)


class SlabSession:
    """Repeated sharded sample -> mesh of the same grid (what bench.py --gpus N times): sdfk_dist_session_*.

    submit() queues a step without waiting for the GPU (from the second one on); collect() waits for the OLDEST
    queued step and returns this rank's (vertices, indices); up to `depth` steps are in flight.  mesh() = the whole
    mesh of the step collected last (every rank has it), valid until that slot is resubmitted."""

    def __init__(self, sdf, mn, mx, nx, ny, nz, clip_to_bounds=True, iso=0.0, depth=1):
        self.L = N.lib()
        self.sdf = sdf   # (keeps the program alive)
        self.depth = int(depth)
        self.world, self.rank, self.backend = info()
        h = C.c_void_p()
        N.check(self.L.sdfk_dist_session_create(sdf.program(), N.f3(mn), N.f3(mx), nx, ny, nz, 1 if clip_to_bounds else 0,
                                                C.c_float(iso), self.depth, C.byref(h)))
        self.h = h
        self.in_flight = 0
        self._nv, self._ni = C.c_int64(), C.c_int64()
        self._refs = (C.byref(self._nv), C.byref(self._ni))

    def submit(self):
        N.check(self.L.sdfk_dist_submit(self.h))
        self.in_flight += 1

    def collect(self):
        N.check(self.L.sdfk_dist_collect(self.h, *self._refs))
        self.in_flight -= 1
        return self._nv.value, self._ni.value

    def step(self):
        self.submit()
        return self.collect()

    def drain(self):
        out = None
        while self.in_flight:
            out = self.collect()
        return out

    def counts(self):
        """[(vertices, indices)] per rank of the step collected last."""
        a = (C.c_int64 * (2 * self.world))()
        N.check(self.L.sdfk_dist_counts(self.h, a))
        return [(int(a[2 * q]), int(a[2 * q + 1])) for q in range(self.world)]

    def stats(self):
        a = (C.c_int64 * 8)()
        N.check(self.L.sdfk_dist_stats(self.h, a))
        keys = ("stride_bytes", "steps", "redone", "regrown", "exchange_mode", "host_ns_submit", "host_ns_collect")
        d = dict(zip(keys, (int(x) for x in a)))
        d["depth"], d["index16"], d["index16_fallbacks"] = int(a[7]) & 0xff, bool(int(a[7]) & 0x100), int(a[7]) >> 16
        return d

    def tune(self, steps_per_mode=20):
        """sdfk_dist_tune: measure both exchanges, with plain and with compact payloads, on this fabric and keep the fastest;
        {(mode, compact): agreed ns, -1 = the scene does not fit the compact form}."""
        a = (C.c_int64 * 4)()
        N.check(self.L.sdfk_dist_tune(self.h, steps_per_mode, a))
        return {(k & 1, bool(k >> 1)): int(a[k]) for k in range(4)}

    def enqueue_only(self):
        """This rank's slab kernels without the exchange (measurement)."""
        N.check(self.L.sdfk_dist_enqueue_only(self.h))

    def gathered(self):
        """(device pointer, stride in bytes) of the gather buffer of the step collected last (sdfk_dist_gathered): `world` self-describing
        slab payloads.  Refused (SDFK_ERR_UNSUPPORTED) on a rank that received headers only -- exchange mode 3, mode 2 off rank 0."""
        ptr, stride = C.c_void_p(), C.c_int64()
        N.check(self.L.sdfk_dist_gathered(self.h, C.byref(ptr), C.byref(stride)))
        return ptr.value, stride.value

    def slab_mesh(self):
        """Host copy of THIS rank's slab of the step collected last, indices global (sdfk_dist_slab_mesh: no payload exchange --
        with SDFK_OPT_DIST_EXCHANGE = 3 the whole mesh is the ranks' slabs in rank order)."""
        from .api import Mesh
        m = C.c_void_p()
        N.check(self.L.sdfk_dist_slab_mesh(self.h, C.byref(m)))
        return Mesh._from_handle(m)

    def mesh(self):
        """Host copy of the whole mesh of the step collected last."""
        from .api import Mesh
        m = C.c_void_p()
        N.check(self.L.sdfk_dist_mesh(self.h, C.byref(m)))
        return Mesh._from_handle(m)

    def close(self):
        if self.h is not None:
            self.L.sdfk_dist_session_free(self.h)
            self.h = None
            self.in_flight = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def unpack_self_describing(g):
    """g: [world, stride] uint8 array of rebased slab payloads (sdfk_dist_gathered copied to the host, or a fixture) ->
    concatenated arrays.  Header (64 bytes): int64 nv, ni; float min[3], max[3]; int32 vertex_bytes (36, or 24: colours,
    all zero, left out); int32 cap_v = vertex slots the V / (C) / N sections are laid out for (0 = dense: nv); int32 idx_bits
    (16: compact indices, decoded and rebased here; else int32, already rebased by the step)."""
    V, Cc, Nn, T, mins, maxs = [], [], [], [], [], []
    for row in g:
        nv, ni = (int(x) for x in row[:16].view(np.int64))
        b = row[16:40].view(np.float32)
        vbytes, cap_v = (int(x) for x in row[40:48].view(np.int32))
        sec = 12 * (cap_v if cap_v > 0 else nv)      # bytes per section
        o, vb = SLAB_HEADER_BYTES, nv * 12
        V.append(row[o:o + vb].view(np.float32).reshape(-1, 3))
        o += sec
        if vbytes == 36:
            Cc.append(row[o:o + vb].view(np.float32).reshape(-1, 3))
            o += sec
        else:
            Cc.append(np.zeros((nv, 3), np.float32))
        Nn.append(row[o:o + vb].view(np.float32).reshape(-1, 3))
        o += sec
        idx_bits = int(row[48:52].view(np.int32)[0])
        if idx_bits == 16:     # compact form (k_payload_compact): uint16 offsets + one int32 base per 1024 indices, slab-local ids
            t16 = row[o:o + 2 * ni].view(np.uint16).astype(np.int64)
            ob = o + ((2 * ni + 3) & ~3)
            bases = row[ob:ob + 4 * ((ni + 1023) // 1024)].view(np.int32).astype(np.int64)
            vbase = sum(len(v) for v in V[:-1])
            T.append((np.repeat(bases, 1024)[:ni] + t16 + vbase).astype(np.int32))
        else:
            T.append(row[o:o + 4 * ni].view(np.int32))
        if nv:
            mins.append(b[0:3]); maxs.append(b[3:6])
    mn = np.min(np.stack(mins), axis=0) if mins else np.zeros(3, np.float32)
    mx = np.max(np.stack(maxs), axis=0) if maxs else np.zeros(3, np.float32)
    return (np.concatenate(V), np.concatenate(Cc), np.concatenate(Nn), np.concatenate(T),
            mn.astype(np.float32), mx.astype(np.float32))


class Node:
    """Several GPUs from ONE process (sdfk_node_*, include/sdfkit_hip.h): every listed device gets a context of its own and a host
    thread of the library's own that is its rank; `to_mesh` is SdfEx.ToMesh (Sdf.cs:59-63) over all of them, called from one
    thread -- what a managed host, which is one process, needs (the one-process-per-GPU form is `init` + `sharded_to_mesh`).

        node = D.Node()                      # every GPU of the process; D.Node([0, 0]) = two ranks sharing GPU 0 (host transport)
        mesh = node.to_mesh(sdf, mn, mx, nx, ny, nz)
        node.close()
    """

    def __init__(self, devices=None):
        self._h = C.c_void_p()
        L = N.lib()
        if devices is None:
            N.check(L.sdfk_node_open(None, 0, C.byref(self._h)))
        else:
            arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            N.check(L.sdfk_node_open(arr, len(devices), C.byref(self._h)))
        w, b = C.c_int32(), C.c_int32()
        N.check(L.sdfk_node_info(self._h, C.byref(w), C.byref(b)))
        self.world, self.backend = w.value, b.value

    def to_mesh_handle(self, sdf, mn, mx, nx, ny, nz, clipToBounds=True, isoValue=0.0):
        arr, n, out = sdf.ir()
        h = C.c_void_p()
        N.check(N.lib().sdfk_node_to_mesh(self._h, arr, n, out, int(sdf.writes_color), N.f3(mn), N.f3(mx), nx, ny, nz,
                                          1 if clipToBounds else 0, C.c_float(isoValue), C.byref(h)))
        return h

    def to_mesh(self, sdf, mn, mx, nx, ny, nz, clipToBounds=True, isoValue=0.0):
        from .api import Mesh
        return Mesh._from_handle(self.to_mesh_handle(sdf, mn, mx, nx, ny, nz, clipToBounds, isoValue))

    def to_mesh_host(self, sdf, mn, mx, nx, ny, nz, clipToBounds=True, isoValue=0.0):
        """The same mesh assembled on the HOST: sdfk_node_mesh_begin (the sharded step, totals) + sdfk_node_mesh_copy (every rank copies
        its own slab into its slice of the four exact-length arrays, all PCIe links at once; the slabs never cross xGMI)."""
        from .api import Mesh, MeshArrayPool
        import numpy as np
        arr, n, out = sdf.ir()
        nv, ni, hc = C.c_int64(), C.c_int64(), C.c_int32()
        N.check(N.lib().sdfk_node_mesh_begin(self._h, arr, n, out, int(sdf.writes_color), N.f3(mn), N.f3(mx), nx, ny, nz,
                                             1 if clipToBounds else 0, C.c_float(isoValue), C.byref(nv), C.byref(ni), C.byref(hc)))
        if Mesh.Pool is None:
            Mesh.Pool = MeshArrayPool()
        pool = Mesh.Pool
        v, c, nn = (pool.rent((nv.value, 3), np.float32) for _ in range(3))
        t = pool.rent((ni.value,), np.int32)
        lo, hi = (C.c_float * 3)(), (C.c_float * 3)()
        N.check(N.lib().sdfk_node_mesh_copy(self._h, v.ctypes.data, c.ctypes.data, nn.ctypes.data, t.ctypes.data, lo, hi))
        m = Mesh(v, c, nn, t, np.array(lo[:], np.float32), np.array(hi[:], np.float32))
        m._pool = pool
        return m

    def close(self):
        if self._h is not None and self._h.value:
            N.lib().sdfk_node_close(self._h)
        self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
