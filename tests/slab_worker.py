"""Test helper: one rank's slab pieces through the C ABI, one call at a time (sdfk_volume_create_slab, sdfk_sample,
sdfk_march_begin / finish, sdfk_sample_march_slab, sdfk_slab_enqueue, sdfk_mesh_pack) -- what the library's own step
driver (sdfk_dist_*, csrc/dist_rccl.h) strings together.  The parity tests use it to mesh several slabs on ONE GPU and
compare their concatenation with the whole volume."""
import ctypes as C

from sdfkit_amd import dist as D

HEADER_BYTES = 32  # 6 float32 bounds + 2 pad (the old two-collective payload of pack_into)


def slab_layers(n_layers, world, rank):
    return D.slab(n_layers + 1, world, rank)[:2]


def slab_planes(lb, le, nz):
    from tests import proto_host as P
    return P.slab_planes(lb, le, nz)


class GpuSlabWorker:
    """Product worker: samples and meshes one Z slab on this process's GPU."""

    def __init__(self, sdf, mn, mx, nx, ny, nz, rank, world, clip_to_bounds=True, iso=0.0):
        from sdfkit_amd import _native as N
        self.N = N
        N.init()
        self.sdf, self.mn, self.mx = sdf, mn, mx
        self.nx, self.ny, self.nz = nx, ny, nz
        self.clip, self.iso = clip_to_bounds, iso
        self.lb, self.le, self.z0, self.nzl = D.slab(nz, world, rank)
        self.vol = C.c_void_p()
        N.check(N.lib().sdfk_volume_create_slab(nx, ny, nz, N.f3(mn), N.f3(mx), self.z0, max(self.nzl, 1),
                                                1 if sdf.writes_color else 0, C.byref(self.vol)))
        self.prog = sdf.program()
        self.vertex_bytes = 36 if sdf.writes_color else 24   # payload bytes per vertex (sdfk_mesh_pack)
        self.job = None
        self.mesh = None
        self._clip_i = None

    def begin(self):
        N = self.N
        self.release()
        N.check(N.lib().sdfk_sample(self.prog, self.vol, 1 if self.clip else 0))
        job, nv, ni = C.c_void_p(), C.c_int64(), C.c_int64()
        N.check(N.lib().sdfk_march_begin(self.vol, C.c_float(self.iso), self.lb, self.le, C.byref(job),
                                         C.byref(nv), C.byref(ni)))
        self.job = job
        return nv.value, ni.value

    def finish(self, vertex_base):
        N = self.N
        m = C.c_void_p()
        N.check(N.lib().sdfk_march_finish(self.job, vertex_base, C.byref(m)))
        self.mesh = m
        return m

    def pack_into(self, buf, nv, ni):
        """Pack [bounds | V | C | N | T] of the finished slab mesh into the uint8 torch
        tensor `buf` (on this GPU), device to device."""
        N = self.N
        mn, mx = (C.c_float * 3)(), (C.c_float * 3)()
        N.check(N.lib().sdfk_mesh_bounds(self.mesh, mn, mx))
        import torch
        hdr = torch.tensor(list(mn) + list(mx) + [0.0, 0.0], dtype=torch.float32)
        buf[:HEADER_BYTES].copy_(hdr.view(torch.uint8).to(buf.device, non_blocking=True))
        p = buf.data_ptr() + HEADER_BYTES
        vb = nv * 12
        N.check(N.lib().sdfk_mesh_copy_device(self.mesh, p, p + vb, p + 2 * vb, p + 3 * vb))

    def run_local(self):
        """One-call form: sample the slab and mesh its layers with slab-LOCAL vertex ids
        (sdfk_sample_march_slab, one host sync).  Returns (n_vertices, n_indices)."""
        N = self.N
        self.release()
        m = C.c_void_p()
        N.check(N.lib().sdfk_sample_march_slab(self.prog, self.vol, 1 if self.clip else 0, C.c_float(self.iso),
                                               self.lb, self.le, 0, C.byref(m)))
        self.mesh = m
        nv, ni = C.c_int64(), C.c_int64()
        N.check(N.lib().sdfk_mesh_counts(m, C.byref(nv), C.byref(ni)))
        return nv.value, ni.value

    def enqueue(self, buf, lane=0, wait_event=None):
        """Asynchronous form of run_local + pack_self_describing: queues sample + mesh, EMITTED STRAIGHT INTO the
        uint8 torch tensor `buf` (the mesh arrays are sections of the payload, laid out for the guessed capacities;
        the last kernel writes the header), and returns without waiting; the counts are in the payload header
        (-1 = this job's speculative capacities were too small).
        lane > 0: inside a lane section of the library (sdfk_lane_begin/end), after `wait_event`.
        One foreign call (sdfk_slab_enqueue) instead of five."""
        N = self.N
        self.release()
        if self._clip_i is None:
            self._clip_i, self._iso_f = (1 if self.clip else 0), C.c_float(self.iso)
        N.check(N.lib().sdfk_slab_enqueue(self.prog, self.vol, self._clip_i, self._iso_f, self.lb, self.le,
                                          buf.data_ptr(), buf.numel(), lane, wait_event))

    def pack_self_describing(self, buf):
        """sdfk_mesh_pack into the uint8 torch tensor `buf`; returns the bytes needed."""
        need = C.c_int64()
        self.N.check(self.N.lib().sdfk_mesh_pack(self.mesh, C.c_void_p(buf.data_ptr()), buf.numel(), C.byref(need)))
        return need.value

    def release(self):
        N = self.N
        if self.mesh is not None:
            N.lib().sdfk_mesh_free(self.mesh)
            self.mesh = None
        if self.job is not None:
            N.lib().sdfk_march_job_free(self.job)
            self.job = None

    def close(self):
        self.release()
        if self.vol is not None:
            self.N.lib().sdfk_volume_free(self.vol)
            self.vol = None
