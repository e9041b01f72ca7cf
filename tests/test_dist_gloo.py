"""World-size-2 (and 3) CPU tests of the Z-slab step protocol over the gloo backend.  The protocol under test is the
PRODUCT's C++ class (sdfkit_amd/csrc/slab_protocol.h, sdfk::SlabProtocol -- what libsdfkit_hip.so drives with HIP kernels
and RCCL), built into a g++ shim and driven through callbacks (tests/proto_host.py): the compute backend is a worker
that serves slices of a mesh computed by the CPU oracle, the transport is torch.distributed's all_gather.  Exercised:
bootstrap and stride agreement, pipelined slots, "a rank's speculative buffers were too small -> every rank redoes that
step exactly", a payload that outgrows the stride -> every rank regrows its buffers, matched collectives throughout.  (The
GPU half -- sdfk_dist_* with the same class -- is covered by tests/test_gpu_multirank.py.)"""
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
    """The part of a reference mesh that belongs to cell layers [lb, le)."""

    def __init__(self, mesh, nx, ny, nz, rank, world):
        from tests import proto_host as P
        self.m = mesh
        self.lb, self.le = P.slab_layers(nz - 1, world, rank)
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


class FixtureSessionWorker:
    """Protocol backend over fixtures: the self-describing payload (64-byte header, slab-LOCAL indices) of this rank's
    part of a reference mesh; `fail_at` = enqueue calls that pretend the speculative buffers were too small (header
    counts -1, no arrays); `grow_at` = enqueue call from which the worker serves `bigger` (a larger mesh) instead."""

    vertex_bytes = 36

    def __init__(self, mesh, nx, ny, nz, rank, world, fail_at=(), bigger=None, grow_at=None):
        self.f = FixtureSlabWorker(mesh, nx, ny, nz, rank, world)
        self.big = FixtureSlabWorker(bigger[0], *bigger[1], rank, world) if bigger else None
        self.calls, self.fail_at, self.grow_at = 0, set(fail_at), grow_at

    def _cur(self):
        return self.big if (self.big is not None and self.grow_at is not None and self.calls >= self.grow_at) else self.f

    def _payload(self):
        f = self._cur()
        m = f.m
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
        if len(raw) > buf.numel():     # does not fit: header only (what sdfk_mesh_pack does)
            raw = raw[:64]
        buf[:len(raw)] = torch.from_numpy(raw.copy())

    def enqueue(self, buf):
        self.calls += 1
        if self.calls in self.fail_at:
            buf[:16] = torch.from_numpy(np.array([-1, -1], np.int64).view(np.uint8).copy())
            return
        self.pack_self_describing(buf)


def _same(arrs, ref):
    V, Cc, Nn, T, mn, mx = arrs
    return (np.array_equal(V, ref.vertices) and np.array_equal(Cc, ref.colors) and np.array_equal(Nn, ref.normals, equal_nan=True) and
            np.array_equal(T, ref.triangles) and np.array_equal(mn, ref.min) and np.array_equal(mx, ref.max))


def _reference(scene_fn, dims):
    from oracle import oracle as O
    scene, _ = scene_fn()
    mn, mx = [-2.8125] * 3, [2.8125] * 3
    v, c = O.sample(scene, mn, mx, *dims)
    O.clip_to_bounds(v, mn, mx)
    return O.march(v, c, mn, mx)


def _one_off_worker(rank, world, port, dims, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests import proto_host as P
        from tests import scenes as S
        ref = _reference(S.readme_repeat_xy, dims)
        ses = P.ProtoSession(lambda slot: FixtureSessionWorker(ref, *dims, rank, world), depth=1)
        ses.submit()                       # the one-off form (sdfk_dist_to_mesh): bootstrap = exact step + stride agreement
        nv, ni = ses.collect()
        arrs = ses.mesh()
        ok = _same(arrs, ref) and (nv, ni) == ses.workers[0].run_local()
        out_q.put((rank, bool(ok), len(arrs[0]), len(arrs[3])))
        ses.close()
    finally:
        dist.destroy_process_group()


def _spawn(target, world, args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port) + args + (q,)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world", [2, 3])
def test_slab_exchange_over_gloo(world):
    res = _spawn(_one_off_worker, world, ((20, 18, 23),))
    assert all(ok for _, ok, _, _ in res), res
    assert len({(nv, nt) for _, _, nv, nt in res}) == 1  # every rank holds the same full mesh


def _session_worker(rank, world, port, dims, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests import proto_host as P
        from tests import scenes as S
        ref = _reference(S.union8, dims)
        # rank 1 "fails" its 3rd queued step: every rank must redo exactly that step
        fails = (3,) if rank == 1 else ()
        ses = P.ProtoSession(lambda slot: FixtureSessionWorker(ref, *dims, rank, world, fails if slot == 1 else ()), depth=3)
        ok, steps = True, 0
        for it in range(9):            # keep the pipeline full: submit ahead, collect the oldest
            if ses.in_flight == ses.depth:
                nv, ni = ses.collect()
                steps += 1
                ok &= (nv, ni) == ses.workers[0].run_local()
                ok &= _same(ses.mesh(), ref)
            ses.submit()
        while ses.in_flight:
            ses.collect()
            steps += 1
            ok &= _same(ses.mesh(), ref)
        out_q.put((rank, bool(ok), steps, ses.redone, ses.grown))
        ses.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_pipelined_session_over_gloo(world):
    """Three steps in flight: payload headers carry the counts, a step one rank marks as failed is redone by all ranks,
    collectives stay matched."""
    res = _spawn(_session_worker, world, ((22, 20, 19),))
    assert all(ok for _, ok, _, _, _ in res), res
    assert all(steps == 9 for _, _, steps, _, _ in res), res
    assert all(redone == 1 and grown == 0 for _, _, _, redone, grown in res), res


def _growing_worker(rank, world, port, dims, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests import proto_host as P
        from tests import scenes as S
        small = _reference(lambda: S.sphere_w(0.6), dims)          # one sphere ...
        big = _reference(S.union8, dims)              # ... then eight primitives: every slab payload outgrows the stride
        workers = []

        def make_worker(slot):
            # the scene changes between the 2nd and the 3rd queued step of every slot (the same steps on every rank)
            w = FixtureSessionWorker(small, *dims, rank, world, bigger=(big, dims), grow_at=2)
            workers.append(w)
            return w

        ses = P.ProtoSession(make_worker, depth=2, headroom=1.0 / 32)
        ok, seen = True, []
        for it in range(10):
            if ses.in_flight == ses.depth:
                ses.collect()
                arrs = ses.mesh()
                seen.append("small" if _same(arrs, small) else ("big" if _same(arrs, big) else "?"))
            ses.submit()
        while ses.in_flight:
            ses.collect()
            arrs = ses.mesh()
            seen.append("small" if _same(arrs, small) else ("big" if _same(arrs, big) else "?"))
        ok &= "?" not in seen and seen[0] == "small" and seen[-1] == "big" and seen == sorted(seen, key=lambda s: s != "small")
        out_q.put((rank, bool(ok), seen, ses.redone, ses.grown, ses.stride))
        ses.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_payload_outgrowing_the_stride_regrows_on_every_rank(world):
    """The scene gets bigger mid-session: the payloads no longer fit the agreed stride.  The headers say so, every rank
    redoes the step exactly, re-agrees the stride and regrows its buffers; steps that were already queued are re-run."""
    res = _spawn(_growing_worker, world, ((30, 28, 31),))
    assert all(ok for _, ok, *_ in res), res
    assert len({tuple(seen) for _, _, seen, *_ in res}) == 1, res       # every rank saw the same sequence of meshes
    assert all(grown >= 1 and redone >= 1 for _, _, _, redone, grown, _ in res), res
    assert len({stride for *_, stride in res}) == 1, res                # ... and agreed on the same new stride


def _reset_worker(rank, world, port, dims, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests import proto_host as P
        from tests import scenes as S
        ref = _reference(S.readme_repeat_xy, dims)
        ses = P.ProtoSession(lambda slot: FixtureSessionWorker(ref, *dims, rank, world), depth=2)
        ok, strides = True, []
        for phase in range(3):
            for it in range(5):
                if ses.in_flight == ses.depth:
                    ses.collect()
                    ok &= _same(ses.mesh(), ref)
                ses.submit()
            while ses.in_flight:
                ses.collect()
                ok &= _same(ses.mesh(), ref)
            strides.append(ses.agreed_stride)
            try:
                ses.submit()
                ses.reset()                 # refused with a step in flight
                ok = False
            except RuntimeError:
                pass
            ses.collect()
            ses.reset()                     # (what sdfk_dist_tune does when it changes the payload form)
            ok &= ses.agreed_stride == 0
        out_q.put((rank, bool(ok), strides, ses.redone, ses.grown))
        ses.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_reset_bootstraps_again_on_every_rank(world):
    """SlabProtocol::reset (nothing in flight, every rank at the same point): the next step is an exact one again and agrees
    the stride anew; results unchanged, no step counted as redone or regrown."""
    res = _spawn(_reset_worker, world, ((22, 20, 19),))
    assert all(ok for _, ok, *_ in res), res
    assert all(len(set(strides)) == 1 and strides[0] > 0 and redone == 0 and grown == 0 for _, _, strides, redone, grown in res), res


class _BreaksAt(FixtureSessionWorker):
    """A worker whose `at`-th enqueue fails for real on this rank (an allocation that fails on ONE device): an exception, which the
    harness turns into a non-zero status of the callback."""

    def __init__(self, *a, at=None, **k):
        super().__init__(*a, **k)
        self.at = at

    def enqueue(self, buf):
        if self.at is not None and self.calls + 1 == self.at:
            self.calls += 1
            raise MemoryError("injected: this rank could not queue its step")
        super().enqueue(buf)


def _rank_local_failure_worker(rank, world, port, dims, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        from tests import proto_host as P
        from tests import scenes as S
        ref = _reference(S.union8, dims)
        # the LAST rank cannot queue the 2nd speculative step of slot 0 (the 5th submit with two slots: the first is the bootstrap's
        # exact step): with an agreement between the rank-local
        # part and the exchange every rank's submit() fails there -- nobody is left waiting in the all-gather
        ses = P.ProtoSession(lambda slot: _BreaksAt(ref, *dims, rank, world, at=(2 if (rank == world - 1 and slot == 0) else None)),
                             depth=2, agree_on_failures=True)
        done, failed_at = 0, None
        for it in range(6):
            try:
                if ses.in_flight == ses.depth:
                    ses.collect()
                    done += 1
                ses.submit()
            except RuntimeError:
                failed_at = it
                break
        agreed = [e for e in ses.log if e[0] == "consensus" and e[2] != 0]
        out_q.put((rank, failed_at, done, len(agreed), agreed[0][1] if agreed else None))
        ses.h = None      # (no drain: the step that failed is not in the queue, and the ranks stop here together)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_a_rank_local_failure_takes_every_rank_out_of_the_step(world):
    """SlabOps::consensus (slab_protocol.h): between the rank-local part of a step (enqueue / run_exact / resize / pack_exact) and
    the collective that follows, ranks that can agree cheaply do -- a rank that could not queue its step must not leave the
    others in the exchange for ever (the round-5 advisor's finding for sdfk_node_*, whose ranks are threads).  Here the agreement
    is a gloo all-reduce; every rank's submit() fails at the SAME step and no rank hangs."""
    res = _spawn(_rank_local_failure_worker, world, ((22, 20, 19),))
    assert {f for _, f, *_ in res} == {4}, res                       # the fifth submit, on every rank
    assert all(n == 1 for *_, n, _ in res), res                      # exactly one agreement came back non-zero
    own = {r: mine for r, _, _, _, mine in res}
    assert own[world - 1] != 0 and all(own[r] == 0 for r in range(world - 1)), res   # ... and only the last rank had failed itself


def test_slab_partition_covers_all_layers():
    from tests import proto_host as P
    for n_layers in (0, 1, 7, 8, 511, 1023):
        for world in (1, 2, 3, 8):
            got, prev = [], 0
            for r in range(world):
                lb, le = P.slab_layers(n_layers, world, r)
                assert lb == prev and le >= lb
                got.append(le - lb)
                prev = le
            assert prev == n_layers and max(got) - min(got) <= 1
    assert P.slab_planes(0, 64, 512) == (0, 68)       # context [0, 66) widened upwards to a multiple of 4 planes
    assert P.slab_planes(64, 128, 512) == (62, 68)    # two context planes below, two above: already a multiple of 4
    assert P.slab_planes(448, 511, 512) == (444, 68)  # clipped at the last plane: widened downwards
    assert P.slab_planes(0, 4, 5) == (0, 5)           # nowhere to widen to: stays as it is
    for nz in (5, 64, 511, 512, 1024):
        for world in (1, 2, 3, 8):
            for r in range(world):
                lb, le = P.slab_layers(nz - 1, world, r)
                z0, n = P.slab_planes(lb, le, nz)
                assert z0 <= max(lb - 2, 0) and z0 + n >= min(le + 2, nz) and z0 >= 0 and z0 + n <= nz
                assert n % 4 == 0 or n == nz or nz < 8


def test_library_partition_matches_the_protocol_header():
    """sdfk_dist_slab (the exported form, no device needed) = slab_layers + slab_planes of slab_protocol.h."""
    from sdfkit_amd import dist as D
    from tests import proto_host as P
    for nz in (2, 5, 64, 511, 512, 1024):
        for world in (1, 2, 3, 8):
            for r in range(world):
                lb, le = P.slab_layers(nz - 1, world, r)
                assert D.slab(nz, world, r) == (lb, le) + P.slab_planes(lb, le, nz)
