"""World-size-2 (and 3) CPU tests of the Z-slab exchange protocol (sdfkit_amd/dist.py) over
the gloo backend.  The compute backend is replaced by a worker that serves slices of a mesh
computed by the CPU oracle (tests may use the oracle; the product worker is GpuSlabWorker and
is covered by the -m gpu slab tests) -- what is exercised here is everything torch.distributed
touches: the count all-gather, the exclusive-prefix vertex bases, the packed padded payload,
the all-gather and the reassembly in rank order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class FixtureSlabWorker:
    """Serves the part of a reference mesh that belongs to cell layers [lb, le)."""

    def __init__(self, mesh, nx, ny, nz, rank, world):
        from sdfkit_amd import dist as D
        self.m = mesh
        self.lb, self.le = D.slab_layers(nz - 1, world, rank)
        cz = mesh.cells[:, 0] // (nx * ny)           # layer of every active cell, sweep order
        nt = mesh.cells[:, 3]
        tri_start = np.concatenate([[0], np.cumsum(nt * 3)])
        i0, i1 = np.searchsorted(cz, self.lb, "left"), np.searchsorted(cz, self.le, "left")  # cz is sorted
        self.t0, self.t1 = int(tri_start[i0]), int(tri_start[i1])
        # vertices are numbered in order of first reference: those first referenced by my triangles
        first = np.full(len(mesh.vertices), len(mesh.triangles), np.int64)
        np.minimum.at(first, mesh.triangles, np.arange(len(mesh.triangles)))
        self.v0, self.v1 = int(np.count_nonzero(first < self.t0)), int(np.count_nonzero(first < self.t1))
        mine = np.nonzero((first >= self.t0) & (first < self.t1))[0]
        assert len(mine) == self.v1 - self.v0 and (len(mine) == 0 or (mine[0] == self.v0 and mine[-1] == self.v1 - 1))
        self.base_seen = None

    def begin(self):
        return self.v1 - self.v0, self.t1 - self.t0

    def finish(self, vertex_base):
        self.base_seen = vertex_base
        assert vertex_base == self.v0  # exclusive prefix of the lower slabs' vertex counts

    def pack_into(self, buf, nv, ni):
        from sdfkit_amd import dist as D
        m = self.m
        v = m.vertices[self.v0:self.v1]
        hdr = np.zeros(8, np.float32)
        if nv:
            hdr[0:3] = v.min(axis=0)
            hdr[3:6] = v.max(axis=0)
        parts = [hdr.view(np.uint8), np.ascontiguousarray(v).view(np.uint8).ravel(),
                 np.ascontiguousarray(m.colors[self.v0:self.v1]).view(np.uint8).ravel(),
                 np.ascontiguousarray(m.normals[self.v0:self.v1]).view(np.uint8).ravel(),
                 np.ascontiguousarray(m.triangles[self.t0:self.t1]).view(np.uint8).ravel()]
        raw = np.concatenate(parts)
        assert len(raw) == D.payload_bytes(nv, ni)
        buf[:len(raw)] = torch.from_numpy(raw.copy())


def _worker(rank, world, port, dims, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from sdfkit_amd import dist as D
        from tests import scenes as S
        scene, _ = S.readme_repeat_xy()
        mn, mx = [-2.8125] * 3, [2.8125] * 3
        v, c = O.sample(scene, mn, mx, *dims)
        O.clip_to_bounds(v, mn, mx)
        ref = O.march(v, c, mn, mx)
        w = FixtureSlabWorker(ref, *dims, rank, world)
        gathered, nvs, nis = D.sharded_step(w, None, "cpu")
        V, Cc, Nn, T, bmin, bmax = D.unpack(gathered, nvs, nis)
        ok = (np.array_equal(V, ref.vertices) and np.array_equal(Cc, ref.colors) and
              np.array_equal(Nn, ref.normals, equal_nan=True) and np.array_equal(T, ref.triangles) and
              np.array_equal(bmin, ref.min) and np.array_equal(bmax, ref.max) and sum(nvs) == len(ref.vertices))
        out_q.put((rank, bool(ok), len(V), len(T)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_exchange_over_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    dims = (20, 18, 23)
    procs = [ctx.Process(target=_worker, args=(r, world, port, dims, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    assert len({(nv, nt) for _, _, nv, nt in res}) == 1  # every rank holds the same full mesh


class FixtureSessionWorker:
    """SlabSession backend over fixtures: the self-describing payload (64-byte header, slab-LOCAL
    indices) of this rank's part of a reference mesh; `fail_at` = enqueue calls that pretend the
    speculative buffers were too small (header counts -1, no arrays)."""

    def __init__(self, mesh, nx, ny, nz, rank, world, fail_at=()):
        self.f = FixtureSlabWorker(mesh, nx, ny, nz, rank, world)
        self.calls, self.fail_at = 0, set(fail_at)

    def _payload(self):
        f, m = self.f, self.f.m
        nv, ni = f.v1 - f.v0, f.t1 - f.t0
        v = m.vertices[f.v0:f.v1]
        hdr = np.zeros(64, np.uint8)
        hdr[:16] = np.array([nv, ni], np.int64).view(np.uint8)
        hdr[40:44] = np.array([36], np.int32).view(np.uint8)      # bytes per vertex: V, C, N
        if nv:
            hdr[16:28] = v.min(axis=0).astype(np.float32).view(np.uint8)
            hdr[28:40] = v.max(axis=0).astype(np.float32).view(np.uint8)
        tri = (m.triangles[f.t0:f.t1].astype(np.int64) - f.v0).astype(np.int32)   # slab-local ids (may be negative: seam)
        parts = [hdr, np.ascontiguousarray(v).view(np.uint8).ravel(),
                 np.ascontiguousarray(m.colors[f.v0:f.v1]).view(np.uint8).ravel(),
                 np.ascontiguousarray(m.normals[f.v0:f.v1]).view(np.uint8).ravel(), tri.view(np.uint8).ravel()]
        return nv, ni, np.concatenate(parts)

    def run_local(self):
        nv, ni, _ = self._payload()
        return nv, ni

    def pack_self_describing(self, buf):
        _, _, raw = self._payload()
        buf[:len(raw)] = torch.from_numpy(raw.copy())

    def enqueue(self, buf):
        self.calls += 1
        if self.calls in self.fail_at:
            buf[:16] = torch.from_numpy(np.array([-1, -1], np.int64).view(np.uint8).copy())
            return
        self.pack_self_describing(buf)

    def close(self):
        pass


def _rebase_host(gathered, world, stride):
    """numpy twin of sdfk_slabs_rebase (test fixture only)."""
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


def _session_worker(rank, world, port, dims, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle as O
        from sdfkit_amd import dist as D
        from tests import scenes as S
        scene, _ = S.union8()
        mn, mx = [-2.8125] * 3, [2.8125] * 3
        v, c = O.sample(scene, mn, mx, *dims)
        O.clip_to_bounds(v, mn, mx)
        ref = O.march(v, c, mn, mx)
        # rank 1 "fails" its 3rd queued step: every rank must redo exactly that step
        fails = (3,) if rank == 1 else ()
        workers = []

        def make_worker(slot):
            w = FixtureSessionWorker(ref, *dims, rank, world, fails if slot == 1 else ())
            workers.append(w)
            return w

        ses = D.SlabSession(group=None, device="cpu", depth=3, make_worker=make_worker, rebase=_rebase_host)
        ok, steps = True, 0
        for it in range(9):            # keep the pipeline full: submit ahead, collect the oldest
            if len(ses.queue) == ses.depth:
                nv, ni = ses.collect()
                steps += 1
                m = ses.mesh()
                ok &= (nv, ni) == workers[0].run_local()
                ok &= (np.array_equal(m.Vertices, ref.vertices) and np.array_equal(m.Triangles, ref.triangles) and
                       np.array_equal(m.Normals, ref.normals, equal_nan=True) and np.array_equal(m.Colors, ref.colors) and
                       np.array_equal(m.Min, ref.min) and np.array_equal(m.Max, ref.max))
            ses.submit()
        while ses.queue:
            ses.collect()
            steps += 1
            m = ses.mesh()
            ok &= np.array_equal(m.Vertices, ref.vertices) and np.array_equal(m.Triangles, ref.triangles)
        out_q.put((rank, bool(ok), steps, ses.redone))
        ses.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_session_over_gloo(world):
    """SlabSession with three steps in flight: payload headers carry the counts, a step one rank
    marks as failed is redone by all ranks, collectives stay matched."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    dims = (22, 20, 19)
    procs = [ctx.Process(target=_session_worker, args=(r, world, port, dims, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res
    assert all(steps == 9 for _, _, steps, _ in res), res
    assert all(redone == 1 for _, _, _, redone in res), res


def test_slab_partition_covers_all_layers():
    from sdfkit_amd import dist as D
    for n_layers in (0, 1, 7, 8, 511, 1023):
        for world in (1, 2, 3, 8):
            got, prev = [], 0
            for r in range(world):
                lb, le = D.slab_layers(n_layers, world, r)
                assert lb == prev and le >= lb
                got.append(le - lb)
                prev = le
            assert prev == n_layers and max(got) - min(got) <= 1
    assert D.slab_planes(0, 64, 512) == (0, 68)       # context [0, 66) widened upwards to a multiple of 4 planes
    assert D.slab_planes(64, 128, 512) == (62, 68)    # two context planes below, two above: already a multiple of 4
    assert D.slab_planes(448, 511, 512) == (444, 68)  # clipped at the last plane: widened downwards
    assert D.slab_planes(0, 4, 5) == (0, 5)           # nowhere to widen to: stays as it is
    for nz in (5, 64, 511, 512, 1024):
        for world in (1, 2, 3, 8):
            for r in range(world):
                lb, le = D.slab_layers(nz - 1, world, r)
                z0, n = D.slab_planes(lb, le, nz)
                assert z0 <= max(lb - 2, 0) and z0 + n >= min(le + 2, nz) and z0 >= 0 and z0 + n <= nz
                assert n % 4 == 0 or n == nz or nz < 8
    assert D.exclusive_prefix([3, 0, 5]) == ([0, 3, 3], 8)
