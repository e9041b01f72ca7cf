"""Several GPUs from ONE process: sdfk_node_* (include/sdfkit_hip.h, csrc/node_local.h) and the per-thread device contexts behind it.

The reference is a library one .NET process calls (Sdf.cs:59-63); a node gives every listed device a context of its own and a host
thread of the library's own that is its Z-slab rank.  On a one-GPU box: one rank over real RCCL (ncclCommInitRank at world 1), and
2-3 ranks that SHARE the GPU and exchange through host memory between the threads -- the whole protocol, several device
contexts alive in one process, every mesh the oracle's bit for bit.  With two or more GPUs the same calls run over RCCL."""
import ctypes as C
import threading

import numpy as np
import pytest

from sdfkit_amd import SdfExprs, Sdfs, Vec3
from sdfkit_amd import _native as N
from sdfkit_amd import dist as D
from sdfkit_amd.api import Mesh
from oracle import oracle as O
from tests import scenes as S
from tests.test_gpu_parity import assert_mesh_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    N.init(0)
    return True


def _oracle(scene, mn, mx, dims, clip=True, iso=0.0):
    ov, oc = O.sample(scene, mn, mx, *dims)
    if clip:
        O.clip_to_bounds(ov, mn, mx)
    return O.march(ov, oc, mn, mx, iso=iso)


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_node_meshes_are_the_oracles(gpu, devices):
    """One process, len(devices) ranks (threads of the library), each with its own device context: sphere and the README scene
    (colours), repeated calls (the second is one sharded step of the kept session), a change of scene in between."""
    mn, mx, dims = [-1.5] * 3, [1.5] * 3, (72, 64, 80)
    scene, sdf = S.sphere_w(1.0)
    ref = _oracle(scene, mn, mx, dims, clip=False)
    scene2, sdf2 = S.CATALOGUE["readme_repeat_xy"]()
    mn2, mx2 = [-2.8125] * 3, [2.8125] * 3
    ref2 = _oracle(scene2, mn2, mx2, dims, clip=True)
    with D.Node(devices) as node:
        assert node.world == len(devices)
        assert node.backend == (1 if len(set(devices)) == len(devices) else 2)
        for rep in range(3):
            assert_mesh_equal(node.to_mesh(sdf, mn, mx, *dims, clipToBounds=False), ref)
        for rep in range(2):
            assert_mesh_equal(node.to_mesh(sdf2, mn2, mx2, *dims, clipToBounds=True), ref2)
        assert_mesh_equal(node.to_mesh(sdf, mn, mx, *dims, clipToBounds=False), ref)
        # the host-array form: begin (the step, totals) + copy (every rank its own slab into its slice; nothing crosses between the GPUs)
        for rep in range(2):
            assert_mesh_equal(node.to_mesh_host(sdf2, mn2, mx2, *dims, clipToBounds=True), ref2)
            assert_mesh_equal(node.to_mesh_host(sdf, mn, mx, *dims, clipToBounds=False), ref)
        # the calling thread's own context is untouched by the node's private ones
        assert_mesh_equal(sdf.ToMesh(mn, mx, *dims, clipToBounds=False), ref)
        # a mesh handle of the node is an ordinary handle: accessors from this thread, device-side transform included
        h = node.to_mesh_handle(sdf, mn, mx, *dims, clipToBounds=False)
        a, b = C.c_int64(), C.c_int64()
        N.check(N.lib().sdfk_mesh_counts(h, C.byref(a), C.byref(b)))
        assert (a.value, b.value) == (len(ref.vertices), len(ref.triangles))
        assert_mesh_equal(Mesh._from_handle(h), ref)


def test_node_errors_come_back_to_the_caller(gpu):
    L = N.lib()
    h = C.c_void_p()
    bad = (C.c_int32 * 1)(99)
    assert L.sdfk_node_open(bad, 1, C.byref(h)) == N.ERR_INVALID and not h.value
    assert b"out of range" in L.sdfk_last_error()
    with D.Node([0, 0]) as node:
        _, sdf = S.sphere_w(1.0)
        arr, n, out = sdf.ir()
        m = C.c_void_p()
        # a grid the reference's int32 linear index cannot hold (Voxels.cs:82): every rank refuses, the caller gets the message
        r = L.sdfk_node_to_mesh(node._h, arr, n, out, 0, N.f3([-1] * 3), N.f3([1] * 3), 2048, 2048, 2048, 0, C.c_float(0.0), C.byref(m))
        assert r != 0 and not m.value and b"rank" in L.sdfk_last_error()
        # and the node still works
        mn, mx, dims = [-1.5] * 3, [1.5] * 3, (40, 40, 40)
        scene, _ = S.sphere_w(1.0)
        assert_mesh_equal(node.to_mesh(sdf, mn, mx, *dims, clipToBounds=False), _oracle(scene, mn, mx, dims, clip=False))


@pytest.mark.parametrize("stage,what", [(0, "injected failure$"), (1, "rank-local part of the step"), (2, "inside the exchange")])
def test_one_ranks_failure_fails_the_command_on_every_rank(gpu, monkeypatch, stage, what):
    """A rank-local failure (an allocation on ONE device) must not leave the other ranks waiting in a collective -- wherever it
    happens.  SDFK_NODE_FAULT_RANK injects one such failure on rank 1, SDFK_NODE_FAULT_STAGE says where:
      0  while the rank makes its session: the ranks agree before any of them enters the step (node_agree);
      1  in the rank-local part of the step, after the session stage (the round-5 advisor's finding): the protocol's consensus point
         between enqueue / run_exact and the first collective (SlabOps::consensus = a thread barrier for a node's ranks);
      2  INSIDE the exchange, between two rendezvous of the ranks, where nobody can agree: the failed rank aborts the node's
         barrier, which releases the ranks that wait for it there.
    In every case the caller gets the failing rank's own message, the node is not stuck (the call returns, and so does
    sdfk_node_close), and the next command is served."""
    mn, mx, dims = [-1.5] * 3, [1.5] * 3, (40, 40, 40)
    scene, sdf = S.sphere_w(1.0)
    ref = _oracle(scene, mn, mx, dims, clip=False)
    for form in ("handle", "arrays"):
        monkeypatch.setenv("SDFK_NODE_FAULT_RANK", "1")
        monkeypatch.setenv("SDFK_NODE_FAULT_STAGE", str(stage))
        with D.Node([0, 0, 0]) as node:
            monkeypatch.delenv("SDFK_NODE_FAULT_RANK")
            monkeypatch.delenv("SDFK_NODE_FAULT_STAGE")
            with pytest.raises(Exception, match="rank 1 .*injected failure") as ei:
                if form == "handle":
                    node.to_mesh(sdf, mn, mx, *dims, clipToBounds=False)
                else:
                    node.to_mesh_host(sdf, mn, mx, *dims, clipToBounds=False)
            import re
            assert re.search(what, str(ei.value)), str(ei.value)
            # the fault was for one command: the next one is served
            assert_mesh_equal(node.to_mesh(sdf, mn, mx, *dims, clipToBounds=False), ref)
            assert_mesh_equal(node.to_mesh_host(sdf, mn, mx, *dims, clipToBounds=False), ref)


def test_a_rank_that_cannot_prepare_fails_the_opening_of_the_node(gpu, monkeypatch):
    """Start-up has the same shape: ncclCommInitRank returns when EVERY rank has called it, so the ranks first do the rank-local half
    of sdfk_dist_init (library, exchange stream, agreement buffers), agree that all of them are prepared, and only then join.
    SDFK_NODE_FAULT_STAGE=3 fails rank 1's preparation: sdfk_node_open returns its message, nothing hangs, the next node opens."""
    monkeypatch.setenv("SDFK_NODE_FAULT_RANK", "1")
    monkeypatch.setenv("SDFK_NODE_FAULT_STAGE", "3")
    with pytest.raises(Exception, match="rank 1 .*injected failure before ncclCommInitRank"):
        D.Node([0, 0])
    monkeypatch.delenv("SDFK_NODE_FAULT_RANK")
    monkeypatch.delenv("SDFK_NODE_FAULT_STAGE")
    mn, mx, dims = [-1.5] * 3, [1.5] * 3, (40, 40, 40)
    scene, sdf = S.sphere_w(1.0)
    with D.Node([0, 0]) as node:
        assert_mesh_equal(node.to_mesh(sdf, mn, mx, *dims, clipToBounds=False), _oracle(scene, mn, mx, dims, clip=False))


def test_threads_have_their_own_current_context(gpu):
    """sdfk_init is per thread (like hipSetDevice): a second thread that initialises device 0 shares that device's context with the
    first; a thread that never called sdfk_init works in the process's first context; meshes made on one thread are read on another."""
    mn, mx, dims = [-1.5] * 3, [1.5] * 3, (48, 48, 48)
    scene, sdf = S.sphere_w(1.0)
    ref = _oracle(scene, mn, mx, dims, clip=False)
    out, errs = {}, []

    def worker(k, call_init):
        try:
            if call_init:
                N.check(N.lib().sdfk_init(0))
            _, mine = S.sphere_w(1.0)
            out[k] = mine.ToMesh(mn, mx, *dims, clipToBounds=False)
        except Exception as e:   # noqa: BLE001
            errs.append((k, repr(e)))

    ts = [threading.Thread(target=worker, args=(k, k % 2 == 0)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for k in range(4):
        assert_mesh_equal(out[k], ref)
    # a device that does not exist is an error that leaves the thread's context as it was
    n = C.c_int()
    assert N.lib().sdfk_init(63) == N.ERR_INVALID
    assert_mesh_equal(sdf.ToMesh(mn, mx, *dims, clipToBounds=False), ref)


_NODE_BEFORE_INIT = r"""
import ctypes as C, sys, threading
sys.path.insert(0, %r)
from sdfkit_amd import _native as N
from sdfkit_amd import dist as D
from tests import scenes as S
L = N.lib()
mn, mx, dims = [-1.5] * 3, [1.5] * 3, (40, 40, 40)
_, sdf = S.sphere_w(1.0)
node = D.Node([0])                          # a node BEFORE any sdfk_init of the process
ref = node.to_mesh(sdf, mn, mx, *dims, clipToBounds=False)
out = {}
def never_chose():                          # a thread that never called sdfk_init: must not fall into rank 0's private context
    v = C.c_void_p()
    out["status"] = L.sdfk_volume_create(8, 8, 8, N.f3(mn), N.f3(mx), 0, C.byref(v))
    out["msg"] = L.sdfk_last_error().decode()
t = threading.Thread(target=never_chose); t.start(); t.join()
assert out["status"] == N.ERR_NO_DEVICE and "sdfk_init" in out["msg"], out
N.init(0)                                   # now a listed context exists: it becomes the default of the threads that never chose
def after_init():
    v = C.c_void_p()
    out["status2"] = L.sdfk_volume_create(8, 8, 8, N.f3(mn), N.f3(mx), 0, C.byref(v))
    if v.value: L.sdfk_volume_free(v)
t = threading.Thread(target=after_init); t.start(); t.join()
assert out["status2"] == 0, out
mine = sdf.ToMesh(mn, mx, *dims, clipToBounds=False)
import numpy as np
assert np.array_equal(mine.Vertices, ref.Vertices) and np.array_equal(mine.Triangles, ref.Triangles)
node.close()
again = sdf.ToMesh(mn, mx, *dims, clipToBounds=False)      # the listed context survives the node's private ones
assert np.array_equal(again.Vertices, ref.Vertices)
print("node-before-init ok")
"""


def test_a_node_opened_before_sdfk_init_keeps_its_contexts_private():
    """Round-5 advisor: a node opened before any sdfk_init took the process's DEFAULT context object for rank 0, and threads that never
    called sdfk_init then silently worked inside rank 0's private context.  An unlisted claim never takes the default's object: such a
    thread fails loudly (SDFK_ERR_NO_DEVICE) until somebody calls sdfk_init, whose context then becomes the default."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", _NODE_BEFORE_INIT % root], capture_output=True, text=True, timeout=300, cwd=root)
    assert p.returncode == 0 and "node-before-init ok" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
