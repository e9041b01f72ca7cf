"""Z-slab sharding of the sample -> mesh path over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).  The
reference has no distributed path at all; this is new design:

* Rank r owns the contiguous CELL layers [lb, le) of the serial z sweep
  (MarchingCubes.cs:53-82), so the global vertex/triangle order is the concatenation of
  the slabs in rank order.
* Sampling is a pure function of the voxel index (Voxels.cs:99-108), so no halo is
  EXCHANGED: each rank samples its own planes plus context planes [lb-2, le+2).
  The two planes below let it recount which vertices the layer lb-1 creates (they are
  referenced by layer lb) including the "earlier cell emits nothing" corner case; the plane
  above feeds the normals of vertices on its top face.
* Exchange #1 (tiny): all-gather of (vertex count, index count); exclusive prefix = global
  vertex base of each slab.  Triangle indices are emitted already rebased.
* Exchange #2: one padded all-gather of the packed slab meshes [bounds | V | C | N | T].

The compute backend is a "slab worker" with begin()/finish(); GpuSlabWorker is the product
one (C ABI, HIP).  The exchange code only touches torch tensors, so it is exercised on CPU
with the gloo backend in tests/ (with a worker built from fixtures).
"""
import ctypes as C

import numpy as np

HEADER_BYTES = 32  # 6 float32 bounds + 2 pad


def slab_layers(n_layers, world, rank):
    """Balanced contiguous split of `n_layers` cell layers: returns [lb, le)."""
    base, rem = divmod(max(n_layers, 0), world)
    lb = rank * base + min(rank, rem)
    return lb, lb + base + (1 if rank < rem else 0)


def slab_planes(lb, le, nz):
    """Voxel planes [z0, z0+n) a slab holds for layers [lb, le): the context the C ABI asks for
    ([lb-2, le+2) clipped to the grid, include/sdfkit_hip.h), widened to a multiple of 4 planes
    where the grid allows -- the fused sampling kernel (one 16-byte store per 4 z) needs that, and
    a few extra planes cost far less than falling back to the one-voxel-per-lane kernel."""
    z0 = max(lb - 2, 0)
    z1 = min(le + 2, nz)
    pad = (-(z1 - z0)) % 4
    up = min(pad, nz - z1)
    z1 += up
    z0 -= min(pad - up, z0)
    return z0, z1 - z0


def exclusive_prefix(counts):
    out, acc = [], 0
    for c in counts:
        out.append(acc)
        acc += int(c)
    return out, acc


class GpuSlabWorker:
    """Product worker: samples and meshes one Z slab on this process's GPU."""

    def __init__(self, sdf, mn, mx, nx, ny, nz, rank, world, clip_to_bounds=True, iso=0.0):
        from . import _native as N
        self.N = N
        N.init()
        self.sdf, self.mn, self.mx = sdf, mn, mx
        self.nx, self.ny, self.nz = nx, ny, nz
        self.clip, self.iso = clip_to_bounds, iso
        self.lb, self.le = slab_layers(nz - 1, world, rank)
        self.z0, self.nzl = slab_planes(self.lb, self.le, nz)
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


def payload_bytes(nv, ni):
    return HEADER_BYTES + 36 * int(nv) + 4 * int(ni)


def exchange_counts(nv, ni, group=None, device="cpu"):
    """All-gather of (nv, ni).  Returns (list of nv, list of ni)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    mine = torch.tensor([nv, ni], dtype=torch.int64, device=device)
    out = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    host = torch.stack(out).cpu().numpy()
    return [int(x) for x in host[:, 0]], [int(x) for x in host[:, 1]]


def gather_payloads(buf, group=None):
    """One padded all-gather of the packed slab meshes.  `buf`: uint8 tensor, same length
    on every rank.  Returns a [world, len] uint8 tensor."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world, buf.numel()), dtype=torch.uint8, device=buf.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out.view(-1), buf, group=group)
    else:
        parts = [out[r] for r in range(world)]
        dist.all_gather(parts, buf, group=group)
    return out


def unpack(gathered, nvs, nis):
    """Concatenate slab payloads in rank order -> (V, C, N, T, min, max) numpy arrays."""
    g = gathered.cpu().numpy()
    V, Cc, Nn, T, mins, maxs = [], [], [], [], [], []
    for r, (nv, ni) in enumerate(zip(nvs, nis)):
        row = g[r]
        hdr = row[:HEADER_BYTES].view(np.float32)
        o = HEADER_BYTES
        vb = nv * 12
        V.append(row[o:o + vb].view(np.float32).reshape(-1, 3))
        Cc.append(row[o + vb:o + 2 * vb].view(np.float32).reshape(-1, 3))
        Nn.append(row[o + 2 * vb:o + 3 * vb].view(np.float32).reshape(-1, 3))
        T.append(row[o + 3 * vb:o + 3 * vb + 4 * ni].view(np.int32))
        if nv > 0:
            mins.append(hdr[0:3])
            maxs.append(hdr[3:6])
    mn = np.min(np.stack(mins), axis=0) if mins else np.zeros(3, np.float32)
    mx = np.max(np.stack(maxs), axis=0) if maxs else np.zeros(3, np.float32)
    return (np.concatenate(V), np.concatenate(Cc), np.concatenate(Nn), np.concatenate(T),
            mn.astype(np.float32), mx.astype(np.float32))


def sharded_step(worker, group=None, device="cpu", make_buffer=None):
    """One sample -> mesh pass of this rank's slab plus the two exchanges.  Returns
    (gathered uint8 [world, L], nvs, nis).  `make_buffer(nbytes)` returns the uint8 send
    buffer (on `device`); the worker packs its slab mesh into it."""
    import torch
    nv, ni = worker.begin()
    nvs, nis = exchange_counts(nv, ni, group, device)
    bases, _ = exclusive_prefix(nvs)
    import torch.distributed as dist
    rank = dist.get_rank(group)
    worker.finish(bases[rank])
    nbytes = max(payload_bytes(a, b) for a, b in zip(nvs, nis))
    nbytes = (nbytes + 255) // 256 * 256
    buf = make_buffer(nbytes) if make_buffer else torch.zeros(nbytes, dtype=torch.uint8, device=device)
    worker.pack_into(buf, nv, ni)
    return gather_payloads(buf, group), nvs, nis


def sharded_to_mesh(sdf, mn, mx, nx, ny, nz, clip_to_bounds=True, iso=0.0, group=None):
    """SdfEx.ToMesh over all ranks of `group` (one GPU each); every rank returns the full Mesh."""
    import torch
    import torch.distributed as dist
    from . import _native as N
    from .api import Mesh
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device())
    N.init(torch.cuda.current_device())
    N.bind_torch_stream(dev)
    w = GpuSlabWorker(sdf, mn, mx, nx, ny, nz, rank, world, clip_to_bounds, iso)
    try:
        gathered, nvs, nis = sharded_step(w, group, dev)
        torch.cuda.current_stream().synchronize()
        V, Cc, Nn, T, bmin, bmax = unpack(gathered, nvs, nis)
    finally:
        w.close()
    return Mesh(V, Cc, Nn, T, bmin, bmax)


# ---------------------------------------------------------------------------
# steady-state form: ONE collective per step, NO host wait inside a step
# ---------------------------------------------------------------------------
SLAB_HEADER_BYTES = 64  # SDFK_SLAB_HEADER_BYTES


class SlabSession:
    """Repeated sharded sample -> mesh of the same grid (what bench.py --gpus N times).

    A step, per rank, is queued without waiting for the GPU: sample + mesh the slab with
    slab-local ids (buffers sized from the previous step), pack a self-describing payload ON THE
    DEVICE (header = counts + bounds, written from the job's own counters), ONE padded all-gather
    over RCCL -- launched asynchronously, so that it travels while the NEXT step's kernels run --
    one kernel that rebases the gathered indices from the headers, and an asynchronous copy of
    the `world` headers to pinned host memory.  `submit()` queues a step into the next of
    `depth` slots (own slab volume, send and gather buffers each); `collect()` waits for the
    OLDEST queued step only, reads its headers and returns this rank's counts -- so with
    depth > 1 the host never idles the GPU or the links between steps.  `step()` = submit +
    collect (one step in flight).

    The payload stride is agreed once, on the first step, with a count all-gather (+`headroom`,
    12.5 % by default: the padding travels too); a later slab that outgrows it raises (make a new
    session, or use sharded_to_mesh for one-off meshes).
    A step whose speculative buffers were too small on ANY rank is marked in that rank's
    header; every rank sees it after the gather and all of them redo that step on the exact
    (synchronising) path -- same decision everywhere, so the collectives stay matched.

    `make_worker(slot)` returns the compute backend of a slot: `run_local() -> (nv, ni)`
    (synchronous, exact), `pack_self_describing(buf)`, `enqueue(buf)` (asynchronous form of the
    two) and `close()`.  `rebase(gathered, world, stride)` adds the vertex bases to the gathered
    indices in place.  The defaults are the product ones (GpuSlabWorker, sdfk_slabs_rebase);
    tests/test_dist_gloo.py drives the same protocol on CPU tensors over gloo with fixtures."""

    def __init__(self, sdf=None, mn=None, mx=None, nx=0, ny=0, nz=0, clip_to_bounds=True, iso=0.0, group=None,
                 device=None, depth=1, make_worker=None, rebase=None, headroom=0.125):
        import torch
        import torch.distributed as dist
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        if make_worker is None:
            def make_worker(slot):
                return GpuSlabWorker(sdf, mn, mx, nx, ny, nz, self.rank, self.world, clip_to_bounds, iso)
        self.mirror_headers = rebase is None and self.on_gpu   # product path: the rebase kernel also mirrors the headers to the host
        if self.mirror_headers:
            from . import _native as N
            N.init(self.device.index if self.device.index is not None else None)
            N.bind_torch_stream(self.device)   # the collectives must be ordered against the library's kernels
        if rebase is None:
            from . import _native as N

            def rebase(gathered, world, stride, mirror=None):
                if mirror is not None:
                    N.check(N.lib().sdfk_slabs_rebase_mirror(C.c_void_p(gathered.data_ptr()), world, stride, C.c_void_p(mirror.data_ptr())))
                else:
                    N.check(N.lib().sdfk_slabs_rebase(C.c_void_p(gathered.data_ptr()), world, stride))
        self.rebase = rebase
        self.depth = max(int(depth), 1)
        self.headroom = float(headroom)   # payload stride = largest first-step payload * (1 + headroom): the padding travels too
        self.workers = [make_worker(k) for k in range(self.depth)]
        self.worker = self.workers[0]
        self.stride = None
        self.buf = [None] * self.depth
        self.gathered_slots = [None] * self.depth
        self.hdr_host = [None] * self.depth
        self.ready = [None] * self.depth
        self.work = [None] * self.depth
        self.buf_free = [None] * self.depth
        # per-step host time matters (at 8 ranks a step is bound by it): everything that can be made once
        # per slot is -- events, the flat view of the gather buffer, an int64 view of the pinned headers
        self.slot_event = [None] * self.depth
        self.gathered_flat = [None] * self.depth
        self.hdr_np = [None] * self.depth
        self._nccl = None
        self._cur_stream = None
        self.nsub = 0
        import os
        self.lanes = 2 if (self.mirror_headers and os.environ.get("SDFK_LANES", "2") != "0") else 0
        self.unfinished = None   # slot whose all-gather has been launched but not yet waited for / rebased
        self.copy_stream = None   # (created on first use: every stream of the process takes a hardware queue)
        self.queue = []          # slots in submission order
        self.next_slot = 0
        self.gathered = None     # gather buffer of the step collected last
        self.redone = 0          # steps that had to be redone on the exact path

    # -- helpers ----------------------------------------------------------------
    def _start_gather(self, slot):
        """Launch the all-gather of a slot WITHOUT making the compute stream wait for it: the
        next step's kernels are queued behind the pack of this one, not behind its exchange."""
        import torch.distributed as dist
        g, b = self.gathered_slots[slot], self.buf[slot]
        if self._nccl is None:
            self._nccl = dist.get_backend(self.group) == "nccl"
        if self._nccl:
            self.work[slot] = dist.all_gather_into_tensor(self.gathered_flat[slot], b, group=self.group, async_op=True)
        else:
            self.work[slot] = dist.all_gather([g[r] for r in range(self.world)], b, group=self.group, async_op=True)
        self.unfinished = slot

    def _finish_gather(self, slot=None):
        """Second half of an exchange (default: the newest one): the compute stream waits for it,
        indices are rebased, the headers start their way to the host."""
        if slot is None:
            slot = self.unfinished
        if slot is None or self.work[slot] is None:
            return
        if self.unfinished == slot:
            self.unfinished = None
        self.work[slot].wait()
        self.work[slot] = None
        if self.mirror_headers:
            # the rebase kernel writes the headers straight into pinned host memory: one event, no
            # copy; the same event says "this slot's send buffer has been read by the collective"
            import torch
            self.rebase(self.gathered_slots[slot], self.world, self.stride, self.hdr_host[slot])
            ev = self.slot_event[slot]   # (a slot is resubmitted only after it has been collected: its event is free again)
            if ev is None:
                ev = self.slot_event[slot] = torch.cuda.Event()
            if self._cur_stream is None:
                self._cur_stream = torch.cuda.current_stream(self.device)
            ev.record(self._cur_stream)
            self.ready[slot] = self.buf_free[slot] = ev
            return
        self.rebase(self.gathered_slots[slot], self.world, self.stride)
        if self.on_gpu:
            import torch
            cur = torch.cuda.current_stream(self.device)
            done = torch.cuda.Event()
            done.record(cur)
            if self.copy_stream is None:
                self.copy_stream = torch.cuda.Stream(self.device)
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(done)
                self.hdr_host[slot].copy_(self.gathered_slots[slot][:, :SLAB_HEADER_BYTES], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            self.ready[slot] = ev

    def _agree_stride(self, need):
        import torch
        import torch.distributed as dist
        t = torch.tensor([need], dtype=torch.int64, device=self.device)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        mx = max(int(x.item()) for x in out)
        # every rank sends `stride` bytes in every step, used or not: keep the head-room modest
        self.stride = ((mx + int(mx * self.headroom) + 4096) + 255) // 256 * 256
        for k in range(self.depth):
            self.buf[k] = torch.zeros(self.stride, dtype=torch.uint8, device=self.device)
            self.gathered_slots[k] = torch.zeros((self.world, self.stride), dtype=torch.uint8, device=self.device)
            h = torch.zeros((self.world, SLAB_HEADER_BYTES), dtype=torch.uint8)
            self.hdr_host[k] = h.pin_memory() if self.on_gpu else h
            self.gathered_flat[k] = self.gathered_slots[k].view(-1)
            self.hdr_np[k] = self.hdr_host[k].numpy().view(np.int64) if self.on_gpu else None   # [world, 8]: nv, ni first

    def _exact_step(self, slot):
        """Synchronous, exact form of a step (first step, and the redo of a failed one)."""
        self._finish_gather()
        w = self.workers[slot]
        nv, ni = w.run_local()
        need = SLAB_HEADER_BYTES + getattr(w, "vertex_bytes", 36) * nv + 4 * ni
        if self.stride is None:   # first step only (every rank takes this branch together)
            self._agree_stride(need)
        if need > self.stride:
            raise RuntimeError(f"slab payload grew to {need} B (> agreed stride {self.stride} B)")
        w.pack_self_describing(self.buf[slot])
        self._start_gather(slot)
        self._finish_gather()
        return nv, ni

    def _headers(self, slot):
        if self.on_gpu:
            self.ready[slot].synchronize()
            return self.hdr_np[slot][:, :2]   # (read in place: the slot is not rewritten before it is resubmitted)
        h = self.gathered_slots[slot][:, :SLAB_HEADER_BYTES].contiguous()
        return h.numpy()[:, :16].copy().view(np.int64)   # [world, 2] = (nv, ni)

    # -- pipeline ---------------------------------------------------------------
    def submit(self):
        """Queue one step into the next slot."""
        if len(self.queue) == self.depth:
            raise RuntimeError("all slots are in flight: collect() first")
        slot = self.next_slot
        self.next_slot = (slot + 1) % self.depth
        if self.stride is None:
            counts = self._exact_step(slot)        # bootstrap: sizes, stride, hints
            self.queue.append((slot, counts))
            return
        # compute of this step first, then the second half of the PREVIOUS step's exchange: the
        # all-gather of step i travels while the kernels of step i+1 run
        if self.lanes:
            # the step's kernels go to an internal stream of the library (alternating between two):
            # the launch-latency-bound kernel chains of consecutive steps overlap on the GPU.  The
            # lane only waits for the collective that read this slot's send buffer last time.
            ev = self.buf_free[slot]
            self.nsub += 1   # (lanes alternate per step, not per slot: consecutive steps never share one)
            self.workers[slot].enqueue(self.buf[slot], 1 + self.nsub % self.lanes, ev.cuda_event if ev is not None else None)
        else:
            self.workers[slot].enqueue(self.buf[slot])
        prev = self.unfinished
        self._start_gather(slot)
        self._finish_gather(prev)
        self.queue.append((slot, None))

    def collect(self):
        """Wait for the oldest queued step; returns this rank's (n_vertices, n_indices)."""
        slot, counts = self.queue.pop(0)
        if counts is None:
            self._finish_gather(slot)
            hdr = self._headers(slot).tolist()      # [[nv, ni]] * world (plain ints: a handful of numpy calls cost more)
            if any(nv < 0 or ni < 0 for nv, ni in hdr):   # some rank's guess was too small: everybody redoes the step
                self.redone += 1
                counts = self._exact_step(slot)
            else:
                vb = getattr(self.workers[slot], "vertex_bytes", 36)
                need = max(SLAB_HEADER_BYTES + vb * nv + 4 * ni for nv, ni in hdr)
                if need > self.stride:
                    raise RuntimeError(f"slab payload grew to {need} B (> agreed stride {self.stride} B)")
                counts = (hdr[self.rank][0], hdr[self.rank][1])
        self.gathered = self.gathered_slots[slot]
        return counts

    def step(self):
        self.submit()
        return self.collect()

    def drain(self):
        out = None
        while self.queue:
            out = self.collect()
        return out

    def mesh(self):
        """Host copy of the mesh of the step collected last (synchronises)."""
        from .api import Mesh
        if self.on_gpu:
            import torch
            torch.cuda.current_stream(self.device).synchronize()
        V, Cc, Nn, T, mn, mx = unpack_self_describing(self.gathered.cpu().numpy())
        return Mesh(V, Cc, Nn, T, mn, mx)

    def close(self):
        self.drain()
        for w in self.workers:
            w.close()


def unpack_self_describing(g):
    """g: [world, stride] uint8 array of rebased slab payloads -> concatenated arrays.  Header (64 bytes): int64 nv, ni;
    float min[3], max[3]; int32 vertex_bytes (36, or 24: colours, all zero, left out); int32 cap_v = vertex slots the
    V / (C) / N sections are laid out for (0 = dense: nv) -- a step that emits straight into its send buffer lays the
    sections out for the capacities it guessed."""
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
        T.append(row[o:o + 4 * ni].view(np.int32))
        if nv:
            mins.append(b[0:3]); maxs.append(b[3:6])
    mn = np.min(np.stack(mins), axis=0) if mins else np.zeros(3, np.float32)
    mx = np.max(np.stack(maxs), axis=0) if maxs else np.zeros(3, np.float32)
    return (np.concatenate(V), np.concatenate(Cc), np.concatenate(Nn), np.concatenate(T),
            mn.astype(np.float32), mx.astype(np.float32))
