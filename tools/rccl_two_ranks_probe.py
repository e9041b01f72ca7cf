#!/usr/bin/env python3
"""GPU probe: can TWO ranks on ONE GPU join an RCCL communicator through sdfk_dist_init?  (NCCL proper refuses duplicate
devices; if RCCL on this stack does too, the multi-rank exchange can only be exercised through the host transport on a
one-GPU box, which is what the tests do.)  Usage: rccl_two_ranks_probe.py [world]"""
import ctypes as C
import multiprocessing as mp
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, conn, mode):
    from sdfkit_amd import _native as N, Sdfs
    from sdfkit_amd import dist as D
    N.init(0)
    L = N.lib()
    if rank == 0:
        buf = (C.c_ubyte * 128)()
        N.check(L.sdfk_dist_unique_id(buf))
        for c in conn:
            c.send(bytes(buf))
        idb = bytes(buf)
    else:
        idb = conn[0].recv()
    r = L.sdfk_dist_init(world, rank, (C.c_ubyte * 128).from_buffer_copy(idb))
    print(f"rank {rank}: sdfk_dist_init -> {r} {L.sdfk_last_error().decode() if r else ''}", flush=True)
    if r:
        return
    N.set_option(N.OPT_DIST_EXCHANGE, mode)
    sdf = Sdfs.Sphere(1.0)
    whole = None
    m = D.sharded_to_mesh(sdf, [-1.5] * 3, [1.5] * 3, 64, 64, 64, False)
    print(f"rank {rank}: mode {mode}: {len(m.Vertices)} vertices, {len(m.Triangles)} indices", flush=True)
    ses = D.SlabSession(sdf, [-1.5] * 3, [1.5] * 3, 64, 64, 64, False, 0.0, depth=3)
    for it in range(12):
        if ses.in_flight == ses.depth:
            ses.collect()
        ses.submit()
    ses.drain()
    print(f"rank {rank}: session ok {ses.counts()} {ses.stats()}", flush=True)
    ses.close()
    D.shutdown()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ctx = mp.get_context("spawn")
    for mode in (0, 1):
        pipes = [ctx.Pipe() for _ in range(world - 1)]
        procs = [ctx.Process(target=worker, args=(0, world, [p[0] for p in pipes], mode))]
        procs += [ctx.Process(target=worker, args=(r, world, [pipes[r - 1][1]], mode)) for r in range(1, world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                print("timeout: killing", p.pid, flush=True)
                p.kill()
