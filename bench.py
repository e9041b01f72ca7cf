#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sample -> mesh hot path on MI355X.

Metric (BASELINE.json): Mvoxels/s (+ Mtris/s) of Voxels.SampleSdf -> MarchingCubes.CreateMesh
on a synthetic 512^3 sphere SDF (`Sdfs.Sphere(1)`, bounds -1.5..1.5, no clip, iso 0) -- config
"C3s" of BASELINE.md.  A step = one pass of the path over that grid: the JIT sampling kernel
writes the volume to HBM, the marching-cubes pipeline reads it and leaves the indexed mesh
(vertices, colours, normals, triangles) in HBM; inputs and outputs stay device-resident.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

N > 1: the SAME 512^3 grid is sharded by Z slab over the ranks (strong scaling); the exchange of the slab
meshes is done by the LIBRARY, which calls RCCL itself on its own stream (sdfk_dist_*: include/sdfkit_hip.h,
csrc/dist_rccl.h) -- torch.distributed here only hands the RCCL id round and provides the barrier / max-over-ranks
of the timing contract (a gloo group: no second communicator, no torch streams on the GPU).  Rank 0 prints ONE JSON line.
Started as plain `python bench.py --gpus N` (no WORLD_SIZE in the environment) it launches
`python -m torch.distributed.run --nproc-per-node N bench.py ...` itself, as a CHILD process,
before anything in this process has touched the GPU, and relays rank 0's JSON line.
Every rank a launcher starts is a SUPERVISOR that never touches HIP: the measuring process is its child, with a time
limit; the supervisors agree (gloo) whether all workers finished and otherwise ALL start the next, more conservative
configuration (RANK_TRIES) -- a collective that hangs on first contact with a node costs one time limit, not the run.

Beside the headline, on the same JSON line (never as `value`):
  N = 1   BASELINE's other configs -- c2_sphere_256, c3_repeatxy, c4_union8_1024 (the whole 1024^3 grid on one GPU), c5_raymarch --, the
          product default (`elided`: the temporary volume of SdfEx.ToMesh is not stored) with a bound of its own, the controls for the
          colour sampler, the host hand-off figures, the CPU baseline (oracle/, the reference algorithm restated in C).
  N > 1   after the headline is safe with the supervisor, under watchdogs of their own: BASELINE C4 (1024^3 union of 8 primitives)
          sharded, with the default exchange and with the mesh left sharded; the CONTENT of every sharded mesh (SHA-256 of the four
          arrays against the same grid meshed on rank 0's GPU alone -- a mismatch fails the run with exit code 4, line on stdout);
          last, the tuner.  SDFK_BENCH_STACKS_AFTER_S=t: every rank dumps its Python stacks after t seconds (where a hung pass sits).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# RCCL / device-memory sharing between the ranks of a node needs dmabuf IPC on this driver stack; the
# launcher's environment normally carries this already
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# one hardware queue per stream in play (the library's lanes, torch's stream, RCCL's): with the runtime's default of 4 a
# stream that waits for an event holds up other streams in the same queue (sdfk_init says more); read at HIP initialisation
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)


WORKLOADS = {
    "sphere": "Sdfs.Sphere(1), bounds -1.5..1.5, no clip",
    "repeatxy": "SdfExprs.Sphere(0.5).RepeatXY(1.125,1.125,colour), bounds -2.8125..2.8125, clipToBounds",
    "union8": "nested SdfExprs.Union of 8 primitives (sphere/box/cylinder) at the octant centres of [-2,2]^3, bounds -2..2, clipToBounds",
}


def _run_ranks(cmd, env, limit_s, graceful=False):
    """Runs a child command in a process group of its own; (exit code, stdout).  On a timeout the whole group is killed and the
    exit code is 124.  graceful: SIGTERM to the child first (torch.distributed.run then stops ITS children, the rank
    supervisors, which stop their workers), SIGKILL to the group a little later.  Nobody is left behind: the child dies with
    this process (PR_SET_PDEATHSIG), a SIGTERM to this process is passed down first, and a process whose own parent has
    disappeared (a supervisor whose launcher was killed) takes its child down and leaves."""
    import signal
    import subprocess
    import time

    def die_with_parent():   # (in the child, before exec)
        try:
            C.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL)   # PR_SET_PDEATHSIG
        except OSError:
            pass

    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True, preexec_fn=die_with_parent)

    def kill_group():
        if graceful and p.poll() is None:
            p.terminate()
            t0 = time.time()
            while p.poll() is None and time.time() - t0 < 10:
                time.sleep(0.1)
        try:
            os.killpg(p.pid, signal.SIGKILL)     # (the exact group this function started)
        except ProcessLookupError:
            pass

    def on_term(signum, frame):   # the launcher is taking this process down: the child's whole group goes first
        kill_group()
        raise SystemExit(128 + signum)

    old_term = signal.signal(signal.SIGTERM, on_term)
    parent = os.getppid()
    t_end = time.time() + limit_s
    try:
        while True:
            try:
                out, _ = p.communicate(timeout=1.0)
                return p.returncode, out
            except subprocess.TimeoutExpired:
                if os.getppid() != parent:   # orphaned: whoever started this process is gone
                    kill_group()
                    raise SystemExit(1)
                if time.time() >= t_end:
                    kill_group()
                    out, _ = p.communicate()
                    return 124, out or ""
    finally:
        signal.signal(signal.SIGTERM, old_term)


# What a multi-rank run falls back to when the pipelined form fails (or hangs) on a node: first RCCL's own all-gather instead of
# the grouped point-to-point sends, one step in flight, no exchange tuning; then, on top, no internal lanes at all (same
# protocol, same kernels).  The bench line of a retry says so (`note`).
RANK_TRIES = [
    ({}, ""),   # (the library's defaults: ncclAllGather of plain payloads, four steps in flight on three lanes; the tuner runs AFTER the headline)
    ({"SDFK_BENCH_DEPTH": "1", "SDFK_DIST_EXCHANGE": "0", "SDFK_BENCH_NO_TUNE": "1", "SDFK_BENCH_INDEX16": "0", "SDFK_BENCH_NO_SINGLE": "1"},
     "retry: one step in flight, plain ncclAllGather of plain payloads, no tuned pass"),
    ({"SDFK_BENCH_DEPTH": "1", "SDFK_LANES": "0", "SDFK_DIST_LANES": "0", "SDFK_DIST_EXCHANGE": "0", "SDFK_BENCH_NO_TUNE": "1",
      "SDFK_BENCH_INDEX16": "0", "SDFK_BENCH_NO_SINGLE": "1", "SDFK_GRAPHS": "0"},
     "retry: one step in flight, no lanes, no captured graphs, plain ncclAllGather of plain payloads, no tuned pass"),
]


def attempt_limit_s():
    return float(os.environ.get("SDFK_BENCH_RANKS_TIMEOUT_S", "300"))   # (a hung collective must not eat the caller's whole budget)


def _content_mismatch(line):
    """True when a bench line says that a sharded mesh differs from the single-GPU mesh (sharded.every_mesh_equals_single_gpu)"""
    try:
        return json.loads(line).get("sharded", {}).get("every_mesh_equals_single_gpu") is False
    except (ValueError, AttributeError):
        return False


def supervise_rank(argv):
    """One rank of a multi-rank run as its launcher started it (`torch.distributed.run ... bench.py --gpus N`: RANK /
    WORLD_SIZE in the environment).  This process is a SUPERVISOR: it never touches HIP.  The measuring process is a child
    (SDFK_BENCH_WORKER=1) with a time limit; the supervisors agree over a gloo group of their own whether every worker
    finished, and if one did not -- died, or hung in a collective and was killed -- ALL of them start the next, more
    conservative configuration of RANK_TRIES.  Rank 0 relays its worker's JSON line."""
    import datetime
    import socket
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # (the container's hostname may not resolve)
    limit = attempt_limit_s()
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=limit + 180))
    import tempfile
    rc = 1
    for attempt, (extra, note) in enumerate(RANK_TRIES):
        port = [0, ""]
        if rank == 0:   # a rendezvous port of their own for the workers of this attempt (one node: the contract of bench.py)
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            port[0] = sk.getsockname()[1]
            sk.close()
            # where rank 0's worker leaves the finished headline line and every worker its "headline done" marker BEFORE the
            # tuned pass starts (one node: a path every rank can see)
            port[1] = os.path.join(tempfile.gettempdir(), f"sdfk_bench_{os.getpid()}_{attempt}")
        dist.broadcast_object_list(port, src=0)
        base = port[1]
        # (without the launcher's TORCHELASTIC_* variables: the workers' rank 0 hosts the store of THEIR rendezvous itself)
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
        env.update(extra)
        env.update({"SDFK_BENCH_WORKER": "1", "MASTER_PORT": str(port[0]), "SDFK_BENCH_RESULT_BASE": base})
        if note:
            env["SDFK_BENCH_NOTE"] = note
            if rank == 0:
                print(f"bench.py: multi-rank run failed (rc {rc}); {note}", file=sys.stderr, flush=True)
        # (SDFK_BENCH_WORKER_SCRIPT: tests/test_bench_host.py puts a stub in the worker's place to exercise this ladder on CPU)
        rc, out = _run_ranks([sys.executable, os.environ.get("SDFK_BENCH_WORKER_SCRIPT", os.path.abspath(__file__))] + argv, env, limit)
        if rc == 124:
            print(f"bench.py: rank {rank} did not finish within {limit:.0f} s", file=sys.stderr, flush=True)
        line = next((l for l in reversed(out.splitlines()) if l.startswith("{")), None)
        # A worker whose HEADLINE pass completed counts as finished even if it died or hung later, in the tuned pass (which tries
        # the exchanges that have never run between two GPUs): its marker is there, and rank 0's line is in the result file.
        headline_done = os.path.exists(f"{base}.rank{rank}.done")
        if rank == 0 and line is None and headline_done and os.path.exists(base + ".json"):
            with open(base + ".json") as f:
                line = f.read().strip() or None
            if line:
                print(f"bench.py: the tuned pass of attempt {attempt} did not finish (rc {rc}); reporting the headline pass", file=sys.stderr, flush=True)
        ok = (rc == 0 or headline_done) and (rank != 0 or line is not None)
        for suffix in (f".rank{rank}.done",) + ((".json",) if rank == 0 else ()):
            try:
                os.remove(base + suffix)
            except OSError:
                pass
        failed = torch.tensor([0 if ok else 1])
        dist.all_reduce(failed, op=dist.ReduceOp.MAX)
        if int(failed.item()) == 0:
            # a run whose sharded mesh DIFFERS from the single-GPU mesh of the same grid is reported -- the line says where -- and fails:
            # every rank leaves with exit code 4 (no retry: a more conservative exchange would hide a wrong mesh, not fix it)
            mismatch = torch.tensor([1 if (rank == 0 and _content_mismatch(line)) else 0])
            dist.broadcast(mismatch, src=0)
            if rank == 0:
                sys.stdout.write(line + "\n")
                sys.stdout.flush()
            dist.destroy_process_group()
            return 4 if int(mismatch.item()) else 0
        sys.stderr.write(out)
        rc = rc or 1
    dist.destroy_process_group()
    return rc or 1


def launch_ranks(argv, n):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU under
    torch.distributed.run as a child process (never exec: see the module docstring), forward its
    output and return its exit code.  Nothing here imports torch or touches HIP.  (The ranks supervise themselves:
    supervise_rank.)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    limit = len(RANK_TRIES) * (attempt_limit_s() + 30) + 240
    rc, out = _run_ranks(cmd, env, limit, graceful=True)
    if rc == 124:
        print(f"bench.py: the ranks did not finish within {limit:.0f} s", file=sys.stderr, flush=True)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    wrong_mesh = any(_content_mismatch(l) for l in lines)
    if (rc == 0 or wrong_mesh) and lines:
        sys.stdout.write(out)
        sys.stdout.flush()
        return 4 if wrong_mesh else 0     # (the line is on stdout either way; a wrong sharded mesh is an error of the run)
    sys.stderr.write(out)
    return rc or 1


def scene_for(name, other_constants=False):
    """other_constants: the same scene STRUCTURE with another radius (the next frame of an animation): no hiprtc compile"""
    from sdfkit_amd import SdfExprs, Sdfs, Vec3
    dr = -0.03125 if other_constants else 0.0
    if name == "sphere":
        return Sdfs.Sphere(1.0 + dr), [-1.5] * 3, [1.5] * 3, False
    if name == "repeatxy":  # BASELINE C3 (README scene), clipToBounds = true
        sdf = SdfExprs.Sphere(0.5 + dr).RepeatXY(1.125, 1.125,
                                            lambda i, p, d: 0.9 * Vec3.of(p.x.b, 1.0) - Vec3.Abs(i) / 6.0).ToSdf()
        return sdf, [-2.8125] * 3, [2.8125] * 3, True
    if name == "union8":    # BASELINE C4: nested Union of 8 primitives at the octant centres of [-2,2]^3
        prims, k = [], 0
        for sx in (-1, 1):
            for sy in (-1, 1):
                for sz in (-1, 1):
                    q = (SdfExprs.Sphere(0.6 + dr), SdfExprs.Box(0.5), SdfExprs.Cylinder(0.4, 0.6))[k % 3]
                    prims.append(q.Translate(sx, sy, sz))
                    k += 1
        prod = prims[0]
        for q in prims[1:]:
            prod = SdfExprs.Union(prod, q)
        return prod.ToSdf(), [-2.0] * 3, [2.0] * 3, True
    if name == "sphere_color":   # (probes only: the colour sampler's kernel shape with next to no arithmetic -- one primitive, a constant colour)
        from sdfkit_amd import SdfFuncs
        return SdfFuncs.Sphere(0.5 + dr).WithColor(1.0, 0.2, 0.3).ToSdf(), [-2.8125] * 3, [2.8125] * 3, True
    if name == "two_spheres_color":   # (probes only: MarchingCubesTests.ColoredSpheres' scene -- two primitives with constant colours, one Union)
        from sdfkit_amd import SdfFuncs
        return SdfFuncs.Union(SdfFuncs.Sphere(0.4).WithColor(1.0, 0.2, 0.3).Translate(-1, 0, 0),
                              SdfFuncs.Sphere(0.2 + dr).WithColor(0.1, 1.0, 0.3).Translate(1, 0, 0)).ToSdf(), [-3.0] * 3, [3.0] * 3, False
    raise SystemExit(f"unknown scene {name}")


def cpu_baseline(scene, n, timed=3):
    """Reference-algorithm CPU baseline: the C restatement of the reference (oracle/), sampler
    multi-threaded over 2048-point batches on all host cores, marching cubes single-threaded
    exactly like the reference.  Same scene and (by default) the same grid as the GPU run;
    1 warm-up + `timed` timed passes, mean of the timed ones (Perf/Program.cs:43-62 convention)."""
    from oracle import oracle as O
    s = O.Scene()
    if scene == "sphere":
        s.sphere_w(1.0)
        mn, mx, clip = [-1.5] * 3, [1.5] * 3, False
    elif scene == "union8":
        nodes, k = [], 0
        for sx in (-1, 1):
            for sy in (-1, 1):
                for sz in (-1, 1):
                    prim = (lambda: s.f_sphere(0.6), lambda: s.f_box(0.5), lambda: s.f_cylinder(0.4, 0.6))[k % 3]()
                    nodes.append(s.f_translate(prim, sx, sy, sz))
                    k += 1
        root = nodes[0]
        for node in nodes[1:]:
            root = s.f_union(root, node)
        s.root = root
        mn, mx, clip = [-2.0] * 3, [2.0] * 3, True
    else:
        s.f_repeat_xy_idx(s.f_sphere(0.5), 1.125, 1.125, O.CF_README)
        mn, mx, clip = [-2.8125] * 3, [2.8125] * 3, True
    cores = O.hardware_threads()
    tot = smp = mc = 0.0
    tris = 0
    for it in range(1 + timed):
        t0 = time.perf_counter()
        v, c = O.sample(s, mn, mx, n, n, n, threads=cores)
        if clip:
            O.clip_to_bounds(v, mn, mx)
        t1 = time.perf_counter()
        m = O.march(v, c, mn, mx)
        t2 = time.perf_counter()
        tris = len(m.triangles) // 3
        del v, c, m
        if it:   # (pass 0 is the warm-up)
            tot += t2 - t0
            smp += t1 - t0
            mc += t2 - t1
    tot, smp, mc = tot / timed, smp / timed, mc / timed
    return {"value": round(n ** 3 / tot / 1e6, 3), "unit": "Mvoxels/s", "cores": cores, "kind": "port",
            "sample": f"{n}^3 grid of the same scene, 1 warm-up + {timed} timed passes, mean "
                      f"(sample {smp:.2f} s on {cores} threads, marching cubes {mc:.2f} s on 1 thread)",
            "mtris_per_s": round(tris / tot / 1e6, 3)}


def load_pmc_traffic(kernel, scene, n):
    """HBM bytes per launch from committed rocprofv3 PMC passes (profiles/), if present."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(p):
        return None
    try:
        d = json.load(open(p))
        return d.get(f"{kernel}@{scene}@{n}")  # bytes per launch, or None when that kernel/scene/size was not profiled
    except Exception:
        return None


def mesh_digest(V, Cc, Nn, T):
    """SHA-256 of the bytes of a mesh's four arrays (numpy): what the sharded content check compares"""
    import hashlib
    import numpy as np
    return {name: hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes() if a.size else b"").hexdigest()
            for name, a in (("Vertices", V), ("Colors", Cc), ("Normals", Nn), ("Triangles", T))}


def compare_sharded_with_single(single, whole, slab_digests, counts, flip_one_index=False):
    """The comparison of the sharded content check, pure: `single` / `whole` = meshes (.Vertices / .Colors / .Normals / .Triangles numpy arrays) of
    the same grid from one GPU / gathered from the ranks; slab_digests[q] = (mesh_digest of rank q's own slab, (vertices, indices)) or None;
    counts[q] = (vertices, indices) of slab q from the gathered headers.  flip_one_index: fault injection -- one index of the gathered mesh
    is flipped before it is hashed (the check must then say false).  tests/test_bench_host.py exercises it without a GPU."""
    T_g = whole.Triangles
    if flip_one_index and len(T_g):
        T_g = T_g.copy()
        T_g[len(T_g) // 2] ^= 1
    d_g = mesh_digest(whole.Vertices, whole.Colors, whole.Normals, T_g)
    d_s = mesh_digest(single.Vertices, single.Colors, single.Normals, single.Triangles)
    slab_ok, vb, ib = True, 0, 0
    for q, (nvq, niq) in enumerate(counts):
        want = mesh_digest(single.Vertices[vb:vb + nvq], single.Colors[vb:vb + nvq], single.Normals[vb:vb + nvq], single.Triangles[ib:ib + niq])
        got = slab_digests[q] if q < len(slab_digests) else None
        slab_ok = slab_ok and got is not None and tuple(got[1]) == (nvq, niq) and got[0] == want
        vb, ib = vb + nvq, ib + niq
    slab_ok = slab_ok and (vb, ib) == (len(single.Vertices), len(single.Triangles))
    return {"mesh_equals_single_gpu": bool(d_g == d_s and len(whole.Vertices) == len(single.Vertices)),
            "every_ranks_slab_equals_its_slice": bool(slab_ok),
            "vertices": len(single.Vertices), "indices": len(single.Triangles), "sha256_single_gpu": d_s,
            "sha256_sharded": d_g if d_g != d_s else "identical"}


XGMI_LINKS = 7
XGMI_LINK_GBS_PER_DIRECTION = 76.8   # 153.6 GB/s per link, both directions together


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", "--grid", dest="n", type=int, default=512, help="grid edge (--grid under torch.distributed.run, whose own parser claims --n)")
    ap.add_argument("--scene", default="sphere", choices=["sphere", "repeatxy", "union8"])
    ap.add_argument("--cpu-n", type=int, default=0, help="grid edge of the CPU-baseline run (default: the same grid as the GPU run)")
    ap.add_argument("--cpu-passes", type=int, default=3, help="timed passes of the CPU baseline (after 1 warm-up)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--minimal", action="store_true", help="timed loop + per-kernel pass only (for rocprofv3 counter passes: no cold-call, "
                    "latency, host-copy or host-Values legs, whose launches would mix into the per-kernel means)")
    ap.add_argument("--elide", action="store_true", help="PROFILING ONLY (with --minimal): the timed loop runs the product default -- the volume is not "
                    "stored (SDFK_OPT_ELIDE_VOLUME = 2) -- so that a rocprofv3 pass sees the volume-less step's kernels; such a line is not the contract's step")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing above has imported torch or
        # called HIP; the ranks are CHILD processes (a process that has touched the GPU must never exec).
        argv = ["--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--grid", str(args.n),
                "--scene", args.scene, "--cpu-n", str(args.cpu_n), "--cpu-passes", str(args.cpu_passes)] + (["--no-cpu"] if args.no_cpu else []) + \
               (["--minimal"] if args.minimal else [])
        sys.exit(launch_ranks(argv, args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("SDFK_BENCH_WORKER") != "1":
        # a rank as the launcher (the driver's torch.distributed.run, or launch_ranks above) started it: supervise a worker
        sys.exit(supervise_rank(sys.argv[1:]))

    # SDFK_BENCH_STACKS_AFTER_S=t: every thread's Python stack goes to stderr t seconds from now (and every t seconds after) -- where a
    # rank sits when a pass hangs on first contact with a node (the supervisor's time limit only says THAT it hung)
    if os.environ.get("SDFK_BENCH_STACKS_AFTER_S"):
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["SDFK_BENCH_STACKS_AFTER_S"]), repeat=True, file=sys.stderr)
    import numpy as np
    import torch
    import torch.distributed as dist
    from sdfkit_amd import _native as N
    from sdfkit_amd import dist as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # SDFK_BENCH_ONE_GPU=1 (testing only): every rank uses GPU 0 and the exchange goes over gloo --
    # lets the whole multi-rank path run on a single-GPU box (RCCL refuses two ranks on one device)
    one_gpu = os.environ.get("SDFK_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
        os.environ["LOCAL_RANK"] = "0"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("SDFK_BENCH_FORCE_DIST") == "1"   # exercise the sharded path on one rank (real RCCL, world 1)
    sharded = world > 1 or force_dist
    # Control plane: a gloo group (barrier, max-over-ranks of the timings, handing the RCCL id round).  Data plane: the
    # library's own RCCL communicator and exchange stream (sdfk_dist_init) -- or, with every rank on ONE GPU (testing), the
    # library's host transport over the same gloo group.  No torch NCCL group: it would be a second communicator and two
    # more streams competing for the hardware queues (see GPU_MAX_HW_QUEUES above).
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # (the container's hostname may not resolve)
            # RCCL's bootstrap (the rendezvous behind ncclCommInitRank) picks a network interface by itself; all ranks of this
            # bench are on ONE node, where the loopback always works -- whatever else the box has or lacks (the data goes over xGMI)
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        import datetime
        # (a short collective timeout: a rank that died must fail the run -- and let the launcher retry -- instead of hanging it)
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    N.init(local_rank)
    L = N.lib()
    N.set_option(N.OPT_ELIDE_VOLUME, 2 if (args.elide and args.minimal) else 0)   # the contract's step STORES the volume, whatever the environment's default; the elided form has a leg of its own
    # torch events (the roofline pass) and the library's lane 0 on one explicit stream.  NOT for a sharded run: it uses no
    # torch GPU work at all, and one more stream in the process changes which of them share a hardware queue (a sharded step
    # on a small slab: 94 us with the extra stream, 39 us without -- tools/slab_step_trace.py)
    stream = None if sharded else N.bind_torch_stream(dev)
    if sharded:
        if one_gpu and world > 1:
            D.init_host(device=local_rank)
        else:
            D.init(device=local_rank)
    backend = {0: None, 1: "RCCL, called by the library (librccl via dlopen, own exchange stream)", 2: "host transport (gloo all-gather; ranks share one GPU)"}[D.info()[2]]

    n = args.n
    sdf, mn, mx, clip = scene_for(args.scene)

    prog = None

    def sample_march_once():
        m = C.c_void_p()
        N.check(L.sdfk_sample_march(prog, N.f3(mn), N.f3(mx), n, n, n, 1 if clip else 0, C.c_float(0.0), 1, C.byref(m)))
        a, b = C.c_int64(), C.c_int64()
        N.check(L.sdfk_mesh_counts(m, C.byref(a), C.byref(b)))
        L.sdfk_mesh_free(m)
        return a.value, b.value

    # ---- what a caller sees on the FIRST call: program creation (op-list validation + source generation), then the
    # first sample -> mesh of a shape: its kernels are compiled on demand (hiprtc, or the on-disk code-object cache of
    # an earlier process) and the exact two-phase path runs (no size hints yet: two host syncs)
    def jit_stats():
        a, b, c = C.c_int64(), C.c_int64(), C.c_double()
        L.sdfk_jit_stats(C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    # the first call when the machine has never seen the program: code-object cache off, and the structure's modules unloaded
    # with the program afterwards (SDFK_OPT_IDLE_PROGRAMS = 0) so that the legs below start where a fresh process starts
    cold_first_call_ms = None
    if world == 1 and not args.minimal:
        with N.option(N.OPT_CODE_CACHE, 0), N.option(N.OPT_IDLE_PROGRAMS, 0):
            cold_sdf = scene_for(args.scene)[0]
            t0 = time.perf_counter()
            cold_sdf.ToMesh(mn, mx, 8, 8, n, clipToBounds=clip)     # same sampler instantiation (same row length), tiny grid
            cold_first_call_ms = round((time.perf_counter() - t0) * 1e3, 1)
            del cold_sdf
    j0 = jit_stats()
    t0 = time.perf_counter()
    prog = sdf.program()
    t_prog = time.perf_counter() - t0
    first = {"program_ms": round(t_prog * 1e3, 2)}
    nv = ni = 0
    first_call_new_constants_ms = None
    if world == 1:
        t0 = time.perf_counter()
        nv, ni = sample_march_once()
        first["first_mesh_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
        j1 = jit_stats()
        first["kernels_from"] = ("hiprtc compile" if j1[0] > j0[0] else "on-disk code-object cache") + \
            f" ({j1[0] - j0[0]} module(s) compiled in {j1[2] - j0[2]:.0f} ms, {j1[1] - j0[1]} loaded from the cache)"
        first["what"] = ("sdfk_program_create, then the first sdfk_sample_march + sdfk_mesh_counts of this shape: the sampler instantiation "
                         "the grid needs + sdfk_corners_eval are compiled / loaded here, then the exact two-phase path runs with buffers "
                         "allocated from the driver")
        if cold_first_call_ms is not None:
            first["cold_first_call_ms"] = cold_first_call_ms
        # The same scene with OTHER constants (a new radius: the next frame of an animation, a parameter sweep).  The constants of
        # a program are kernel arguments and the compiled kernels belong to the program's STRUCTURE, so this is program creation
        # (tracing in the Python mirror + sdfk_program_create) + one sample -> mesh + the counts: no hiprtc, no cache file.
        # In the reference Sdfs.Sphere(radius) is a closure (Sdf.cs:202-214).
        sdf2 = scene_for(args.scene, other_constants=True)[0]
        t0 = time.perf_counter()
        m2 = C.c_void_p()
        N.check(L.sdfk_sample_march(sdf2.program(), N.f3(mn), N.f3(mx), n, n, n, 1 if clip else 0, C.c_float(0.0), 1, C.byref(m2)))
        a2, b2 = C.c_int64(), C.c_int64()
        N.check(L.sdfk_mesh_counts(m2, C.byref(a2), C.byref(b2)))
        first_call_new_constants_ms = round((time.perf_counter() - t0) * 1e3, 3)
        L.sdfk_mesh_free(m2)
        j2 = jit_stats()
        first["new_constants"] = {"ms": first_call_new_constants_ms, "modules_compiled": j2[0] - j1[0], "loaded_from_cache": j2[1] - j1[1],
                                  "vertices": a2.value,
                                  "what": "a fresh Sdf of the same structure with another radius: tracing + sdfk_program_create + sdfk_sample_march + sdfk_mesh_counts"}
        parameterised = "K.k[0]" in sdf.source()     # (programs with more than 28 constants keep literals: their constants ARE their structure)
        first["new_constants"]["constants_are_arguments"] = parameterised
        assert not parameterised or (j2[0] == j1[0] and j2[1] == j1[1]), "a program with other constants must not compile or load anything"
        del sdf2
    first_call_ms = round(first["program_ms"] + first.get("first_mesh_ms", 0.0), 2)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # steps in flight: the GPU then holds ~1 ms of queued work, so that a hiccup of the host (an interrupt, the interpreter)
    # inside the 3 ms timed region of the driver's 20-step command does not drain the pipeline (3 and 6 in flight give the
    # same steady state; with 3 one run in five came out 5-10 % slow)
    depth_env = int(os.environ.get("SDFK_BENCH_DEPTH", "0"))   # 0 = default: 5 single-GPU (the library itself holds at most 6 unread jobs), 4 sharded

    prog0, mn0, mx0, clip0 = prog, mn, mx, clip   # (the headline scene: defaults of single_gpu_stepper)

    def single_gpu_stepper(depth_n, prog=None, mn=None, mx=None, clip=None, n=n):
        prog = prog if prog is not None else prog0
        mn, mx, clip = (mn0 if mn is None else mn), (mx0 if mx is None else mx), (clip0 if clip is None else clip)
        # sdfk_sample_march returns its mesh handle while the kernels are still queued (the
        # sizes are a guess from the previous mesh of this shape; the first accessor waits and
        # verifies).  Steps are therefore enqueued `depth` ahead of the one whose counts are read
        # back: every step is still checked, but the host never idles the GPU in between, and
        # consecutive steps overlap on the library's three internal streams.
        depth = [depth_n]
        inflight, last = [], [0, 0]

        # (the foreign-call arguments are built once: on launch-bound grids a step is ~25 us, and the interpreter's share counts)
        mn_c, mx_c, iso_c, clip_c = N.f3(mn), N.f3(mx), C.c_float(0.0), 1 if clip else 0
        cnt_a, cnt_b = C.c_int64(), C.c_int64()
        ref_a, ref_b = C.byref(cnt_a), C.byref(cnt_b)
        f_march, f_counts, f_free, check = L.sdfk_sample_march, L.sdfk_mesh_counts, L.sdfk_mesh_free, N.check

        def retire(m):
            check(f_counts(m, ref_a, ref_b))
            f_free(m)
            a, b = cnt_a.value, cnt_b.value
            if last[0] and (a != last[0] or b != last[1]):
                raise SystemExit(f"mesh changed between steps: {last} -> {(a, b)}")
            last[0], last[1] = a, b

        def step():
            m = C.c_void_p()
            check(f_march(prog, mn_c, mx_c, n, n, n, clip_c, iso_c, 1, C.byref(m)))
            inflight.append(m)
            while len(inflight) > depth[0]:
                retire(inflight.pop(0))
            return last

        def drain():
            while inflight:
                retire(inflight.pop(0))
            return tuple(last)
        return step, drain, depth

    if not sharded:
        step, drain, depth = single_gpu_stepper(depth_env or 5)
    else:
        # four steps in flight on three internal streams (measured: 128^3 at world 1 27.7 us per step; 3 in flight on 2 streams 39.7),
        # one exchange per step issued by the library on its own stream, no host wait inside a step
        # (sdfk_dist_session_*: csrc/slab_protocol.h + csrc/dist_rccl.h)
        # plain or compact payloads (indices as 16-bit offsets: 48 -> 36 bytes per vertex received from every peer, an encode and
        # a decode pass more): the tuned pass after the headline measures both on this node's fabric; SDFK_BENCH_INDEX16=0/1
        # sets the form of the headline pass (default: plain)
        idx16_env = os.environ.get("SDFK_BENCH_INDEX16")
        N.set_option(N.OPT_DIST_INDEX16, int(idx16_env) if idx16_env is not None else 0)
        headline_mode = N.get_option(N.OPT_DIST_EXCHANGE)
        worker = D.SlabSession(sdf, mn, mx, n, n, n, clip, 0.0, depth=depth_env or 4)
        last = [0, 0]
        tuned = None

        def step():
            if worker.in_flight == worker.depth:
                last[0], last[1] = worker.collect()
            worker.submit()
            return tuple(last)

        def drain():
            while worker.in_flight:
                last[0], last[1] = worker.collect()
            return tuple(last)

    n_warm = max(args.warmup, 1) + 4   # (the extra steps fill the allocator's pool: untimed set-up)
    import gc
    gc.collect()
    gc.disable()   # (no collector pause inside the timed region; collected HERE, before the warm-up: a collection takes
                   # milliseconds during which the GPU idles, and an idle GPU drops its clocks)
    barrier()
    for _ in range(n_warm):
        nv, ni = step()
    nv, ni = drain()
    barrier()
    # (The headline pass of a sharded run uses the configuration with the fewest parts that have never met a second GPU: the
    # library's default exchange -- ncclAllGather, int32 indices, in place.  Which exchange and payload form are FASTEST on this
    # node is measured afterwards, in a pass of its own that is reported separately -- `sharded.tuned_pass` -- and whose failure
    # cannot cost the headline: see "tuned pass" below.)
    # (what a step costs once the pools are filled: a second, short batch -- the first one contains one-off costs, device
    # allocations of half a GB each for one, and on a box that had never run the program before they made the estimate
    # 40 times too large and the clock warm-up below 9 steps long)
    n_est = 8
    t0 = time.perf_counter()
    for _ in range(n_est):
        nv, ni = step()
    nv, ni = drain()
    barrier()
    t_est = time.perf_counter() - t0
    # Clock warm-up, untimed: after an idle period the GPU needs 10-15 ms of load to reach its sustained clocks
    # (tools/step_transient_probe.py, 512^3 sphere: 0.20 ms per step over the first 20 steps after >= 20 ms of idling,
    # 0.169 ms from the 80th step on) -- the W warm-up steps of a default run are 1 ms.  So the same step keeps running
    # for SDFK_BENCH_CLOCK_WARM_MS (default 80 ms; 0 = off) before the timed region; the count is derived from the
    # slowest rank's warm-up time so that every rank queues the same number of steps (matched collectives).
    warm_ms = float(os.environ.get("SDFK_BENCH_CLOCK_WARM_MS", "80"))
    per_warm = max_over_ranks(t_est / n_est)
    n_clock = 0 if warm_ms <= 0 else min(4000, int(warm_ms * 1e-3 / max(per_warm, 1e-6)) + 1)
    for _ in range(n_clock):
        nv, ni = step()
    nv, ni = drain()
    # The timed region, R times over: every block is EXACTLY K steps bracketed by barrier + torch.cuda.synchronize() on both sides (the
    # contract's measurement), max over ranks; `value` is the MEDIAN block, min / max / all blocks are on the line (`blocks`).  One
    # block is 3 ms at 512^3: a single sample says little (box to box 0.147-0.154 ms in round 4).  SDFK_BENCH_BLOCKS=1: one block.
    n_blocks = max(1, int(os.environ.get("SDFK_BENCH_BLOCKS", "11")))
    block_dts = []
    for _ in range(n_blocks):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            nv, ni = step()
        nv, ni = drain()   # every queued step has completed and been checked before the clock stops
        barrier()
        block_dts.append(max_over_ranks(time.perf_counter() - t0))
    dt = sorted(block_dts)[len(block_dts) // 2]
    gc.enable()
    # The same K steps with COLD clocks: the GPU idles long enough to drop its clocks, then W warm-up steps and K timed ones
    # -- what the driver's command measures without the clock warm-up above (the headline `value` has it, and says so).
    time.sleep(0.25)
    barrier()
    for _ in range(max(args.warmup, 1)):
        step()
    drain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    barrier()
    dt_cold = max_over_ranks(time.perf_counter() - t0)
    per_rank = None
    if sharded:   # per-rank counts -> totals of the whole mesh (from the gathered headers of the last step)
        per_rank = [list(c) for c in worker.counts()]
        nv, ni = sum(p[0] for p in per_rank), sum(p[1] for p in per_rank)
    ms_step = dt / args.steps * 1e3   # (each block already is the max over ranks)

    # ---- CONTENT of a sharded mesh (not just its counts): rank 0 meshes the same grid on ONE GPU through the ordinary single-GPU call
    # and compares SHA-256 digests of Vertices / Colors / Normals / Triangles with the mesh the sharded step left (sdfk_dist_mesh: the
    # gathered slabs concatenated; collective with exchange mode 3, where it runs the payload exchange of that step on demand), and
    # every rank's OWN slab (sdfk_dist_slab_mesh, global indices) with its slice of the single-GPU mesh.  First contact with a node
    # of several GPUs then yields correctness evidence, not just a time: `mesh_equals_single_gpu: false` makes the run exit non-zero.
    def content_check(ses, sdf_x, mn_x, mx_x, clip_x, nn, mode_x):
        # (mode_x: the SDFK_OPT_DIST_EXCHANGE the session was created with -- its stats report -1 for the host transport, whatever the mode)
        # A failure of the CHECK on one rank (an allocation, a copy) must cost neither the line nor the other ranks: rank-local parts are
        # guarded, the collectives (the on-demand payload exchange of mode 3, the gathering of the digests, the barrier) are always entered.
        problems = []

        def guarded(what, fn):
            try:
                return fn()
            except BaseException as e:   # noqa: B902 -- reported on the line
                problems.append(f"rank {rank}: {what}: {type(e).__name__}: {e}")
                return None

        counts_x = guarded("counts", ses.counts)

        def own_slab():                  # (first: with exchange mode 3 no payload has moved yet)
            own = ses.slab_mesh()
            return mesh_digest(own.Vertices, own.Colors, own.Normals, own.Triangles), (len(own.Vertices), len(own.Triangles))
        own_dn = guarded("sdfk_dist_slab_mesh", own_slab)
        whole = None
        if mode_x == 3 or rank == 0:     # (mode 3: collective -- the payload exchange of this step happens here; mode 2: only rank 0 holds it)
            whole = guarded("sdfk_dist_mesh", ses.mesh)
        slabs = [None] * world
        if world > 1:
            dist.all_gather_object(slabs, (own_dn, list(problems)))
        else:
            slabs = [(own_dn, list(problems))]
        res = None
        if rank == 0:
            def compare():
                # (sdfk_sample_march itself: in a process that has joined a sharding context sdf.ToMesh IS the sharded, collective call)
                from sdfkit_amd.api import Mesh as HostMesh
                with N.option(N.OPT_ELIDE_VOLUME, 0):
                    h1 = C.c_void_p()
                    N.check(L.sdfk_sample_march(sdf_x.program(), N.f3(mn_x), N.f3(mx_x), nn, nn, nn, 1 if clip_x else 0, C.c_float(0.0), 1, C.byref(h1)))
                    single = HostMesh._from_handle(h1)
                # (SDFK_BENCH_FAULT_FLIP_INDEX=1, tests: one flipped index must turn the check false)
                return compare_sharded_with_single(single, whole, [sl[0] if sl else None for sl in slabs], counts_x,
                                                   flip_one_index=os.environ.get("SDFK_BENCH_FAULT_FLIP_INDEX") == "1")
            everybody = [p for q in range(world) for p in (slabs[q][1] if slabs[q] else [f"rank {q}: nothing gathered"])]
            res = (guarded("comparison", compare) if (whole is not None and counts_x is not None) else None) or \
                {"mesh_equals_single_gpu": None, "every_ranks_slab_equals_its_slice": None}
            everybody += [p for p in problems if p not in everybody]
            if everybody:      # the check itself could not be completed: neither true nor false, and said so
                res["check_failed"] = everybody
            res["what"] = ("SHA-256 of Vertices / Colors / Normals / Triangles: the mesh of the sharded step collected last (sdfk_dist_mesh) against "
                           "sdfk_sample_march of the same grid on rank 0's GPU alone; every rank's own slab (sdfk_dist_slab_mesh) against its slice")
        del whole
        if world > 1:
            dist.barrier()
        return res

    # ---- the sharded step without its exchange: this rank's slab kernels alone, queued back to back
    # (what "kernel-only" means at N > 1); max over ranks
    dist_extra = {}
    headline_content = None
    if sharded:
        headline_content = content_check(worker, sdf, mn, mx, clip, n, headline_mode)   # (before enqueue_only reuses slot 0's send buffer)
        st = worker.stats()
        for _ in range(3):
            worker.enqueue_only()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            worker.enqueue_only()
        torch.cuda.synchronize()
        tk = max_over_ranks(time.perf_counter() - t0)
        barrier()
        stride = st["stride_bytes"]
        mode = st["exchange_mode"]
        recv = (world - 1) * stride
        # one GPU's pipelined step of the SAME grid, measured in this run on rank 0 while the other ranks wait: the numerator of
        # the speed-up ceiling below (and what the driver's own N = 1 run measures)
        single_ms = None
        if world > 1 and not (one_gpu and world > 1) and os.environ.get("SDFK_BENCH_NO_SINGLE") != "1":
            if rank == 0:
                s_step, s_drain, _ = single_gpu_stepper(5)
                for _ in range(12):
                    s_step()
                s_drain()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(max(args.steps, 20)):
                    s_step()
                s_drain()
                single_ms = (time.perf_counter() - t0) / max(args.steps, 20) * 1e3
            barrier()
        links = max(1, min(world - 1, XGMI_LINKS))
        recv_bound_ms = recv / (links * XGMI_LINK_GBS_PER_DIRECTION * 1e9) * 1e3 if world > 1 else 0.0
        ring_bound_ms = recv / (XGMI_LINK_GBS_PER_DIRECTION * 1e9) * 1e3 if world > 1 else 0.0
        dist_extra = {"world": world, "backend": backend,
                      "exchange": {-1: "host transport", 0: "ncclAllGather (in place)", 1: "grouped ncclSend / ncclRecv to every peer (all xGMI links at once)",
                                   2: "grouped sends to rank 0 only (headers to everybody)", 3: "headers only: the mesh stays sharded (payloads on demand)"}[mode],
                      "exchange_tuned_ns": tuned,
                      "payload": ("compact: indices as 16-bit offsets against one int32 base per 1024 indices, decoded by the step into the whole "
                                  "mesh's int32 index array" if st["index16"] else "plain: int32 indices, rebased in place by the step") +
                                 (f" (fell back from the compact form {st['index16_fallbacks']} time(s): a slab did not fit 16 bits)" if st["index16_fallbacks"] else ""),
                      "per_rank_vertices_indices": per_rank,
                      "slab_kernels_only_ms": round(tk / args.steps * 1e3, 4),
                      "end_to_end_ms": round(ms_step, 4),
                      "host_us_per_step": {"submit": round(st["host_ns_submit"] / max(st["steps"], 1) / 1e3, 2),
                                           "collect": round(st["host_ns_collect"] / max(st["steps"], 1) / 1e3, 2),
                                           "what": "host time inside sdfk_dist_submit (queueing a step: one captured graph launch + the exchange + the "
                                                   "rebase kernel) and sdfk_dist_collect (waiting for the oldest step), averaged over every step of "
                                                   "this run; collect includes waiting for the GPU"},
                      "gather_stride_bytes_per_rank": int(stride), "gather_bytes_received_per_rank": int(recv),
                      "xgmi": {"links_per_gpu": XGMI_LINKS, "gbs_per_link_per_direction": XGMI_LINK_GBS_PER_DIRECTION,
                               "receive_bound_ms": round(recv_bound_ms, 4),
                               "one_link_ring_bound_ms": round(ring_bound_ms, 4),
                               "what": "every rank must RECEIVE the other ranks' slab meshes each step: bytes received / (links used x per-direction "
                                       "link rate) is a floor for the step time whatever the kernels do -- reached only by an exchange that drives "
                                       "every link at once; a ring that passes every slab over ONE link per hop is bound by one_link_ring_bound_ms"},
                      # The all-gather contract (every rank ends up with the WHOLE mesh) caps the speed-up whatever the kernels do:
                      # single-GPU step / receive bound.  512^3 sphere at 8 ranks: 26 MB of mesh, 7/8 of it received over 7 links
                      # = 33-44 us against a 0.15-0.16 ms single-GPU step: about 3.6x (plain payloads) to 4.8x (compact ones) -- BASELINE.json's
                      # >= 6x at 8 GPUs is above this ceiling for this mesh size.
                      "speedup_ceiling": None if not (single_ms and recv_bound_ms) else round(single_ms / recv_bound_ms, 2),
                      "speedup_ceiling_is": "single_gpu_ms_per_step / xgmi.receive_bound_ms: the most the z-slab + all-gather contract allows on this "
                                            "grid with this payload form, reached only if the slab kernels hide entirely behind an exchange that runs "
                                            "at the links' peak",
                      # ... and what lifts it: SDFK_OPT_DIST_EXCHANGE = 3, the mesh stays sharded (a step moves the 64-byte headers only;
                      # sdfk_dist_slab_mesh / sdfk_dist_mesh fetch payloads on demand): the step is then bound by the slowest rank's slab
                      # kernels, measured above without the exchange
                      "speedup_ceiling_mesh_stays_sharded": None if not (single_ms and tk) else round(single_ms / (tk / args.steps * 1e3), 2),
                      "single_gpu_ms_per_step": None if single_ms is None else round(single_ms, 4),
                      "speedup_measured": None if not single_ms else round(single_ms / ms_step, 3),
                      "steps_redone_on_the_exact_path": st["redone"], "stride_regrowths": st["regrown"],
                      "mesh_equals_single_gpu": None if headline_content is None else headline_content["mesh_equals_single_gpu"],
                      "content_check": headline_content}
        if os.environ.get("SDFK_BENCH_NOTE"):
            dist_extra["note"] = os.environ["SDFK_BENCH_NOTE"]

    # per-kernel durations: HIP events on the launch stream, same K steps again (events around
    # every launch perturb the un-instrumented timing above, so they get their own pass).  The
    # jobs stay queued ahead (no host bubbles) but on ONE in-order stream: a kernel's roofline is
    # about the kernel having the GPU to itself, not about how it shares the chip with the
    # previous step's kernels on the other internal stream.
    lanes_before = N.get_option(N.OPT_LANES)
    N.set_option(N.OPT_LANES, 0)      # every job on ONE in-order stream
    N.check(L.sdfk_profile_reset())
    N.check(L.sdfk_profile_enable(1))
    for _ in range(args.steps):
        step()
    drain()
    barrier()
    N.check(L.sdfk_profile_enable(0))
    prof = N.profile_snapshot()
    kern = {k: {"avg_us": round(v[0] / max(v[1], 1) * 1e3, 2), "launches": v[1]} for k, v in prof.items() if v[1]}

    # ---- latency of ONE call, nothing else in flight, everything on one stream: sdfk_sample_march,
    # then the accessor that waits for it (what a caller that needs each mesh before it builds the
    # next one gets; the headline `value` is the pipelined steady state)
    latency_ms = latency_default_ms = None
    if not sharded and not args.minimal:
        for _ in range(3):
            sample_march_once()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sample_march_once()
        latency_ms = (time.perf_counter() - t0) / args.steps * 1e3
        # ... and the same with the PRODUCT default (SDFK_OPT_ELIDE_VOLUME = 2: the temporary volume is not stored, blocks away from the
        # surface are not evaluated): what one synchronous sdf.ToMesh costs a caller who changes nothing
        if n ** 3 > (1 << 24):
            with N.option(N.OPT_ELIDE_VOLUME, 2):
                for _ in range(3):
                    sample_march_once()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    sample_march_once()
                latency_default_ms = (time.perf_counter() - t0) / args.steps * 1e3
    N.set_option(N.OPT_LANES, lanes_before)

    # The roofline kernel on its own: K back-to-back launches of the fused sampling kernel into a
    # resident volume (sdfk_sample), HIP event pairs around each launch on the launch stream.  In
    # the pipeline pass above the event pair of a job's FIRST kernel also contains the time the
    # stream sat idle waiting for the (event-laden, hence slower) host to queue the job.
    roof_us = None
    if not sharded and not args.minimal:
        from sdfkit_amd.api import Voxels
        # (four volumes in turn, as the pipeline does: re-writing the SAME 512 MiB back to back is
        # 30 % slower than writing a buffer that was last touched a few launches ago)
        vols = [Voxels(mn, mx, n, n, n) for _ in range(4)]
        for k in range(8):
            vols[k % 4]._sample(sdf, clip=clip)
        N.check(L.sdfk_profile_enable(2))            # the sampling kernel alone
        t_w = time.perf_counter()
        k = 0
        while k < 4 or (time.perf_counter() - t_w) * 1e3 < warm_ms:   # (sustained clocks, as for the timed region above)
            vols[k % 4]._sample(sdf, clip=clip)
            k += 1
            if k % 64 == 0:
                torch.cuda.synchronize()    # (the host must not run far ahead: wall time stands for GPU time here)
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(args.steps):
            vols[k % 4]._sample(sdf, clip=clip)
        e1.record(stream)
        barrier()
        N.check(L.sdfk_profile_enable(0))
        roof_us = e0.elapsed_time(e1) * 1e3 / args.steps
        for vol in vols:
            vol._free()

    # ---- SDFK_OPT_ELIDE_VOLUME (opt-in, never `value`): the same pipelined steps without STORING the volume -- sdfk_sample_march
    # never hands its Voxels out (SdfEx.ToMesh, Sdf.cs:59-63: a temporary), and with the sign bits, re-evaluated corners and
    # re-evaluated vertex colours the meshing chain does not read it.  Meshes are bit-identical (tests/test_gpu_elide_volume.py).
    # The contract's step includes the stores, so the headline keeps them.
    def timed_elided(step_fn, drain_fn, k, mode=2):
        with N.option(N.OPT_ELIDE_VOLUME, mode):
            for _ in range(8):
                step_fn()
            drain_fn()
            t_w = time.perf_counter()
            while (time.perf_counter() - t_w) * 1e3 < warm_ms * 0.5:
                for _ in range(8):
                    step_fn()
                drain_fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k):
                step_fn()
            drain_fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / k

    elided_ms = elided1_ms = None
    elided_kern = None
    if not sharded and not args.minimal and n ** 3 > (1 << 24) and os.environ.get("SDFK_BENCH_NO_ELIDED") != "1":
        elided1_ms = timed_elided(step, drain, args.steps, 1) * 1e3     # every voxel evaluated, nothing stored
        elided_ms = timed_elided(step, drain, args.steps, 2) * 1e3      # + blocks decided by interval arithmetic
        # the volume-less step's kernels, one in-order stream, event pairs per launch (like `kernels_us` for the stored step)
        with N.option(N.OPT_ELIDE_VOLUME, 2):
            lanes_b = N.get_option(N.OPT_LANES)
            N.set_option(N.OPT_LANES, 0)
            for _ in range(4):
                step()
            drain()
            N.check(L.sdfk_profile_reset())
            N.check(L.sdfk_profile_enable(1))
            for _ in range(args.steps):
                step()
            drain()
            torch.cuda.synchronize()
            N.check(L.sdfk_profile_enable(0))
            elided_kern = {k: {"avg_us": round(v[0] / max(v[1], 1) * 1e3, 2), "launches": v[1]} for k, v in N.profile_snapshot().items() if v[1]}
            N.check(L.sdfk_profile_reset())
            N.set_option(N.OPT_LANES, lanes_b)

    # ---- the other BASELINE configs that fit one GPU, timed in THIS run next to the headline -- never as `value`: C2 (256^3 sphere,
    # launch-bound), C3 ("HBM roofline run": the README's RepeatXY scene with colours, 512^3, clipToBounds, /root/reference
    # README.md:24-30) and C4 (1024^3 union of 8 primitives, the grid BASELINE shards over 8 GPUs, here whole on one).  Each: k pipelined
    # steps bracketed by synchronisation with the volume STORED, the same with the product default (volume-less above 2^24 voxels),
    # and its sampling kernel alone, back to back ((4 | 16) B/voxel stored + 1/8 B/voxel of sign bits).
    def side_config(scene_name, n_side, k_side, depth_side, label, sampler_launches=None):
        from sdfkit_amd.api import Voxels
        sdf_s, mn_s, mx_s, clip_s = scene_for(scene_name)
        t0 = time.perf_counter()
        prog_s = sdf_s.program()
        step_s_, drain_s, _ = single_gpu_stepper(depth_side, prog_s, mn_s, mx_s, clip_s, n_side)
        nv_s, ni_s = step_s_()
        nv_s, ni_s = drain_s()
        first_ms = (time.perf_counter() - t0) * 1e3          # program creation + the first sample -> mesh of this scene and shape
        for _ in range(7):
            step_s_()
        nv_s, ni_s = drain_s()
        t_w = time.perf_counter()
        while (time.perf_counter() - t_w) * 1e3 < warm_ms:      # sustained clocks, as for the headline
            for _ in range(depth_side + 3):
                step_s_()
            drain_s()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k_side):
            nv_s, ni_s = step_s_()
        nv_s, ni_s = drain_s()
        torch.cuda.synchronize()
        s_stored = (time.perf_counter() - t0) / k_side
        colors_s = bool(sdf_s.writes_color)
        bpv = 16 if colors_s else 4
        ks = sampler_launches or k_side
        vols = [Voxels(mn_s, mx_s, n_side, n_side, n_side) for _ in range(2)]
        for k in range(4):
            vols[k % 2]._sample(sdf_s, clip=clip_s)
        N.check(L.sdfk_profile_enable(2))            # the sampling kernel alone
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(ks):
            vols[k % 2]._sample(sdf_s, clip=clip_s)
        e1.record(stream)
        torch.cuda.synchronize()
        N.check(L.sdfk_profile_enable(0))
        samp_us = e0.elapsed_time(e1) * 1e3 / ks
        for vol in vols:
            vol._free()
        e2 = e1_ = None
        elidable = n_side ** 3 > (1 << 24)           # (grids up to the captured-graph limit are never elided: DESIGN.md section 5)
        if os.environ.get("SDFK_BENCH_NO_ELIDED") != "1" and elidable:
            e1_ = timed_elided(step_s_, drain_s, k_side, 1)
            e2 = timed_elided(step_s_, drain_s, k_side, 2)
        nvox = n_side ** 3
        bytes_s = nvox * bpv + nvox // 8
        meas = load_pmc_traffic("pipeline_step", scene_name, n_side)
        mesh_b = 36 * nv_s + 4 * ni_s
        out_s = {"workload": WORKLOADS[scene_name] + f", {n_side}^3", "config": label, "steps": k_side, "ms_per_step": round(s_stored * 1e3, 4),
                 "mvoxels_per_s": round(nvox / s_stored / 1e6, 1), "mtris_per_s": round(ni_s / 3 / s_stored / 1e6, 2),
                 "vertices": nv_s, "triangles": ni_s // 3,
                 "first_call_ms": round(first_ms, 1),
                 "frac_design_bytes": round((bytes_s + mesh_b) / s_stored / 1e9 / HBM_PEAK_GBS, 4),
                 "frac_contract_bytes": None if colors_s else round((nvox * 8 + mesh_b) / s_stored / 1e9 / HBM_PEAK_GBS, 4),
                 "frac_measured_bytes": None if not meas else round(meas / s_stored / 1e9 / HBM_PEAK_GBS, 4),
                 # the PRODUCT DEFAULT (SDFK_OPT_ELIDE_VOLUME = 2): the temporary volume of SdfEx.ToMesh is not stored
                 "product_default_ms_per_step": round((e2 if e2 is not None else s_stored) * 1e3, 4),
                 "product_default_is": "volume-less (cull + eval of the blocks the surface passes through)" if e2 is not None else
                                       ("the stored step: grids of at most 2^24 voxels are one captured graph launch and are never elided" if not elidable else "not measured"),
                 "elided_volume_ms_per_step": None if e2 is None else round(e2 * 1e3, 4),
                 "elided_volume_no_culling_ms_per_step": None if e1_ is None else round(e1_ * 1e3, 4),
                 "sampler_us_back_to_back": round(samp_us, 1),
                 "sampler_frac": round(nvox * bpv / (samp_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),            # algorithmic: (4 | 16) B/voxel stored
                 "sampler_frac_design_bytes": round(bytes_s / (samp_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),   # + 1/8 B/voxel of sign bytes
                 "what": f"BASELINE config {label} in this run: {k_side} pipelined sample -> mesh steps ({depth_side} in flight, volume stored: {bpv} B/voxel), "
                         f"the same with the product default, then its sampling kernel alone ({ks} launches back to back into two resident volumes); "
                         "fractions = bytes / time / 8 TB/s (design: (4 | 16) + 1/8 B per voxel + the mesh; contract: SURVEY 8(d)'s 8 B/voxel for a "
                         "distance-only volume; measured: profiles/pmc_traffic.json)"}
        del sdf_s
        return out_s

    side_ok = not sharded and not args.minimal and args.scene == "sphere" and n == 512
    c3 = c2 = c4 = None

    def side_leg(*a, **k):
        # (a leg beside the headline must never cost the headline: whatever goes wrong in it -- 17 GB volumes on a box with less free memory --
        # is reported in its place)
        try:
            return side_config(*a, **k)
        except Exception as e:   # noqa: BLE001
            print(f"bench.py: the {a[-1][:2]} leg failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            try:
                N.set_option(N.OPT_ELIDE_VOLUME, 0)
                N.check(L.sdfk_profile_enable(0))
                torch.cuda.synchronize()
            except Exception:   # noqa: BLE001
                pass
            return {"failed": f"{type(e).__name__}: {e}"}

    if side_ok and os.environ.get("SDFK_BENCH_NO_C3") != "1":
        c3 = side_leg("repeatxy", 512, 10, 5, "C3")
    if side_ok and os.environ.get("SDFK_BENCH_NO_C2") != "1":
        c2 = side_leg("sphere", 256, 40, 5, "C2")
    if side_ok and os.environ.get("SDFK_BENCH_NO_C4") != "1":
        c4 = side_leg("union8", 1024, 5, 3, "C4 (the whole 1024^3 grid on ONE GPU; the 8-GPU Z-slab form is `bench.py --gpus N`'s c4_union8_1024 leg)", sampler_launches=4)

    # ---- a CONTROL for the colour sampler (C3 / C4 run at ~0.70 of peak where the distance-only sampler reaches 0.82): (a) a plain
    # fill of the same bytes and duration -- one launch writes 2 GiB and lasts ~350 us, where hbm_measured's fill writes 512 MiB in
    # 80 us -- and (b) the SAME sampling kernel with next to no arithmetic (a sphere with a constant colour: one square root per voxel,
    # 16 B/voxel stored).  If both drop to the colour sampler's rate, long launches at this store rate are what the part sustains
    # (clock / power management) and nothing is left in the kernel; if they hold the short fill's rate, the SDF's arithmetic is.
    long_fill = None

    def control_leg():
        from sdfkit_amd import SdfFuncs
        from sdfkit_amd.api import Voxels
        bufs = [torch.empty(2 << 30, dtype=torch.uint8, device=dev) for _ in range(2)]
        for b_ in bufs:
            b_.fill_(1)
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        k = 0
        while (time.perf_counter() - t_w) * 1e3 < warm_ms:
            bufs[k % 2].fill_(0)
            k += 1
            if k % 16 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(10):
            bufs[k % 2].fill_(0)
        e1.record()
        torch.cuda.synchronize()
        fill_us = e0.elapsed_time(e1) * 1e3 / 10
        fill_gbs = (2 << 30) / (fill_us * 1e-6) / 1e9
        del bufs
        triv = SdfFuncs.Sphere(0.5).WithColor(1.0, 0.2, 0.3).ToSdf()
        mn3, mx3 = [-2.8125] * 3, [2.8125] * 3
        vols = [Voxels(mn3, mx3, 512, 512, 512) for _ in range(2)]

        def triv_sampler_us(passes):
            # (SDFK_OPT_COLOR_PASSES: 1 = the ONE fused kernel whose shape this controls for; 2 = values + sign bytes, then the colour array as one
            # linear stream -- what the library does by default for programs this small)
            with N.option(N.OPT_COLOR_PASSES, passes):
                for k in range(4):
                    vols[k % 2]._sample(triv, clip=True)
                torch.cuda.synchronize()
                t_w = time.perf_counter()
                k = 0
                while (time.perf_counter() - t_w) * 1e3 < warm_ms:
                    vols[k % 2]._sample(triv, clip=True)
                    k += 1
                    if k % 16 == 0:
                        torch.cuda.synchronize()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for k in range(10):
                    vols[k % 2]._sample(triv, clip=True)
                e1.record(stream)
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / 10

        N.check(L.sdfk_profile_enable(2))            # (the sampling kernels alone: no transposer behind them)
        triv_us = triv_sampler_us(1)
        triv2_us = triv_sampler_us(2)
        N.check(L.sdfk_profile_enable(0))
        for vol in vols:
            vol._free()
        triv_gbs = 512 ** 3 * 16 / (triv_us * 1e-6) / 1e9
        c3_gbs = 512 ** 3 * 16 / (c3["sampler_us_back_to_back"] * 1e-6) / 1e9
        res = {"fill_2gib_us": round(fill_us, 1), "fill_2gib_gbs": round(fill_gbs, 1), "fill_2gib_frac_of_peak": round(fill_gbs / HBM_PEAK_GBS, 4),
                     "trivial_colour_sampler_us": round(triv_us, 1), "trivial_colour_sampler_gbs": round(triv_gbs, 1),
                     "trivial_colour_sampler_frac_of_peak": round(triv_gbs / HBM_PEAK_GBS, 4),
                     "trivial_colour_sampler_two_passes_us": round(triv2_us, 1),
                     "trivial_colour_sampler_two_passes_frac_of_peak": round(512 ** 3 * 16 / (triv2_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "what": "controls for the colour sampler: torch fill_ of 2 GiB (two buffers in turn, 10 launches after the clock warm-up: a plain "
                             "fill as LONG as the colour sampler's launch) and sdfk_sample_bits_clip on a sphere with a constant colour (the same "
                             "kernel shape and 16 B/voxel of stores, one square root of arithmetic; SDFK_OPT_COLOR_PASSES = 1), 512^3, 10 launches back to back; "
                             "two_passes = the same program as the library samples it by default (values + sign bytes, then the colour array as one linear "
                             "stream: profiles/r06_ab_color_passes.txt)"}
        c3["sampler_frac_of_long_fill"] = round(c3_gbs / fill_gbs, 4)
        c3["sampler_frac_of_trivial_colour_sampler"] = round(c3_gbs / triv_gbs, 4)
        return res

    if side_ok and c3 is not None and "failed" not in c3 and os.environ.get("SDFK_BENCH_NO_CONTROL") != "1":
        try:
            long_fill = control_leg()
        except Exception as e:   # noqa: BLE001 -- a leg beside the headline never costs the headline
            print(f"bench.py: the colour-sampler control failed: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            long_fill = {"failed": f"{type(e).__name__}: {e}"}

    # ---- BASELINE config C5 (stretch: RayMarcher.Render, RayMarcher.cs:45-211) in THIS run: 1920 x 1080, 256 depth iterations, the README
    # scene, camera (-2, 2, 4) -> origin (Perf/Program.cs:54-58), images device-resident; never `value`.  ALU-bound (no HBM traffic to
    # speak of): the roofline figure is the share of the vector-issue time, from the committed SQ counter pass (profiles/pmc_traffic.json)
    c5 = None
    if not sharded and not args.minimal and args.scene == "sphere" and n == 512 and os.environ.get("SDFK_BENCH_NO_C5") != "1":
        from sdfkit_amd import Matrix4x4, RayMarcher
        w5, h5, it5, k5 = 1920, 1080, 256, 10
        sdf5 = scene_for("repeatxy")[0]
        rm = RayMarcher(w5, h5, sdf5)
        rm.DepthIterations = it5
        rm.ViewTransform = Matrix4x4.CreateLookAt((-2, 2, 4), (0, 0, 0), (0, 1, 0))
        pos5, vpi5 = rm.camera()
        rgb5 = torch.empty((h5, w5, 3), dtype=torch.float32, device=dev)
        a5 = (sdf5.program(), w5, h5, N.f3(pos5), (C.c_float * 16)(*[float(x) for x in vpi5.ravel()]), C.c_float(1.0), C.c_float(100.0), it5,
              None, C.c_void_p(rgb5.data_ptr()))
        for _ in range(3):          # (Perf/Program.cs:43-62: the first loop is discarded; here also the clocks)
            N.check(L.sdfk_raymarch_device(*a5))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(k5):
            N.check(L.sdfk_raymarch_device(*a5))
        e1.record(stream)
        torch.cuda.synchronize()
        f5 = e0.elapsed_time(e1) * 1e-3 / k5
        evals5 = w5 * h5 * (it5 + 6)
        sq5 = load_pmc_traffic("sdfk_raymarch:SQ_ACTIVE_INST_VALU", "c5", 1080)     # quad-cycles of vector issue per launch, summed over the chip
        iv5 = load_pmc_traffic("sdfk_raymarch:SQ_INSTS_VALU", "c5", 1080)           # wave-level vector instructions per launch
        c5 = {"workload": "RayMarcher.Render 1920 x 1080, 256 depth iterations + 6 evaluations for the normal, README RepeatXY scene, camera (-2,2,4) -> origin",
              "frames": k5, "ms_per_frame": round(f5 * 1e3, 4), "mrays_per_s": round(w5 * h5 / f5 / 1e6, 1), "gevals_per_s": round(evals5 / f5 / 1e9, 2),
              # share of the vector ALUs' peak issue rate: a SIMD issues one wave64 vector instruction per 2 cycles at full rate (157 TFLOP/s
              # fp32 = 1024 SIMDs x 32 FMA lanes x 2.4 GHz), so wave-instructions x 2 cycles / (1024 SIMDs x 2.4 GHz x frame time) -- a LOWER
              # bound: the quarter-rate instructions of the divisions and square roots count as full-rate ones, and the clock the chip
              # holds under this load is below 2.4 GHz (SQ_BUSY_CYCLES of the same pass: ~2.2 GHz)
              "valu_issue_frac": None if not iv5 else round(iv5 * 2 / 1024 / 2.4e9 / f5, 4),
              # wavefront-cycles with a vector instruction in flight / SIMD-cycles (can exceed 1: two wavefronts of a SIMD overlap in the pipeline)
              "valu_active_frac": None if not sq5 else round(sq5 * 4 / 1024 / 2.4e9 / f5, 4),
              "valu_insts_per_eval": None if not iv5 else round(iv5 * 64 / evals5, 1),
              "checksum": float(torch.nan_to_num(rgb5.double()).sum().item()),
              "what": "BASELINE config C5 in this run: K frames back to back, one HIP event pair on the launch stream; bound = vector ALU issue (the scene costs "
                      "two IEEE divisions and a square root per evaluation); valu_issue_frac from the committed SQ counter pass of tools/bench_raymarch.py "
                      "(profiles/pmc_traffic.json: sdfk_raymarch:SQ_ACTIVE_INST_VALU@c5@1080), not re-measured here; depth and colour images are "
                      "bit-identical to the oracle's (tests/test_raymarch.py)"}
        del rgb5, sdf5

    # context figures, outside the timed region, rank 0 only: what this box
    # reaches with a plain device fill / copy, and one step including the mesh copy to the host
    extra = {}
    if rank == 0 and not args.minimal:
        x = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
        y = torch.empty_like(x)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        x.fill_(1); y.copy_(x)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            x.fill_(0)
        e1.record()
        for _ in range(10):
            y.copy_(x)
        e2.record()
        torch.cuda.synchronize()
        extra["hbm_measured"] = {"fill_gbs": round(x.numel() * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1),
                                 "copy_gbs_read_plus_write": round(2 * x.numel() * 10 / (e1.elapsed_time(e2) * 1e-3) / 1e9, 1),
                                 "what": "torch fill_ / copy_ of 512 MiB, 10 launches each"}
        del x, y
        if not sharded:
            # What a host-array caller gets per call: sample + mesh + counts + the four mesh arrays copied to the
            # host + bounds.  Three kinds of destination:
            #   pinned   the Python mirror's Mesh (arrays from the library's pinned arena, sdfk_host_alloc): plain DMA
            #   managed  FRESH private anonymous 4 KiB-page memory nobody has touched -- what `new Vector3[n]` /
            #            `new int[n]` of the C# shim are (Mesh.cs:10-13); the library pre-faults it on its thread pool
            #   numpy    fresh numpy.empty arrays (numpy advises transparent huge pages for them)
            import mmap
            from sdfkit_amd.api import Mesh, MeshArrayPool

            def fresh_managed(nbytes):
                mm = mmap.mmap(-1, max(nbytes, 1), flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
                try:
                    mm.madvise(mmap.MADV_NOHUGEPAGE)
                except (AttributeError, OSError):
                    pass
                return mm, np.frombuffer(mm, dtype=np.uint8, count=nbytes)

            recycled = {}

            def managed_alloc(shape, dtype):   # what GC.AllocateUninitializedArray hands a C# host on a pool miss: memory nobody has touched
                dt = np.dtype(dtype)
                mm, a = fresh_managed(int(np.prod(shape)) * dt.itemsize)
                return a.view(dt).reshape(shape)

            pooled = MeshArrayPool(alloc=managed_alloc)

            def one_call(kind):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                m = C.c_void_p()
                N.check(L.sdfk_sample_march(prog, N.f3(mn), N.f3(mx), n, n, n, 1 if clip else 0, C.c_float(0.0), 1, C.byref(m)))
                if kind == "pinned":
                    hm = Mesh._from_handle(m, pool=MeshArrayPool())   # counts + the four arrays + bounds + stats, frees the handle
                    dt1 = time.perf_counter() - t1
                    del hm
                    return dt1 * 1e3
                if kind == "pooled":
                    # the mirror of the shim's Mesh.FromNative: the four exact-length arrays are RENTED from a pool of arrays
                    # that earlier meshes of this size handed back (Mesh.Recycle); the first call misses and allocates fresh ones
                    hm = Mesh._from_handle(m, pool=pooled)
                    dt1 = time.perf_counter() - t1
                    if (len(hm.Vertices), len(hm.Triangles)) != (nv, ni):
                        raise SystemExit(f"pooled hand-off: mesh differs: {(len(hm.Vertices), len(hm.Triangles))} vs {(nv, ni)}")
                    hm.Recycle()
                    return dt1 * 1e3
                a, b = C.c_int64(), C.c_int64()
                N.check(L.sdfk_mesh_counts(m, C.byref(a), C.byref(b)))
                if kind == "managed":
                    keep = [fresh_managed(a.value * 12) for _ in range(3)] + [fresh_managed(b.value * 4)]
                    arrs = [x for _, x in keep]
                elif kind == "managed_recycled":
                    if recycled.get("shape") != (a.value, b.value):
                        recycled["keep"] = [fresh_managed(a.value * 12) for _ in range(3)] + [fresh_managed(b.value * 4)]
                        recycled["shape"] = (a.value, b.value)
                    arrs = [x for _, x in recycled["keep"]]
                else:
                    arrs = [np.empty(a.value * 12, np.uint8) for _ in range(3)] + [np.empty(b.value * 4, np.uint8)]
                # (the Colors of a program that only writes .W are all zero -- Voxels.cs:88-92 -- and a managed runtime's new
                # array IS zero: the shim passes no colour destination then, and the library has nothing to clear)
                col = arrs[1].ctypes.data if (colors_written or kind == "numpy") else None
                N.check(L.sdfk_mesh_copy(m, arrs[0].ctypes.data, col, arrs[2].ctypes.data, arrs[3].ctypes.data))
                lo, hi = (C.c_float * 3)(), (C.c_float * 3)()
                N.check(L.sdfk_mesh_bounds(m, lo, hi))
                L.sdfk_mesh_free(m)
                return (time.perf_counter() - t1) * 1e3

            # The opaque-delegate route (a user `Sdf` the shim cannot lower: it samples on the CPU with the reference's
            # own code and hands the managed Values array over): sdfk_march_host = upload 4 B/voxel + sign bits from
            # the uploaded volume (k_signbits8) + gathered corners + the same meshing chain, one call at a time
            if n ** 3 * 4 <= (2 << 30):
                from sdfkit_amd.api import Voxels
                hv = Voxels(mn, mx, n, n, n)
                hv._sample(sdf, clip=clip)
                host_values = np.empty((n, n, n), np.float32)
                host_colors = np.empty((n, n, n, 3), np.float32) if sdf.writes_color else None
                N.check(L.sdfk_volume_download(hv._h, host_values.ctypes.data, host_colors.ctypes.data if host_colors is not None else None))
                hv._free()
                ts = []
                for _ in range(4):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    m = C.c_void_p()
                    N.check(L.sdfk_march_host(host_values.ctypes.data, host_colors.ctypes.data if host_colors is not None else None,
                                              n, n, n, N.f3(mn), N.f3(mx), C.c_float(0.0), 1, C.byref(m)))
                    a, b = C.c_int64(), C.c_int64()
                    N.check(L.sdfk_mesh_counts(m, C.byref(a), C.byref(b)))
                    L.sdfk_mesh_free(m)
                    ts.append((time.perf_counter() - t1) * 1e3)
                    if (a.value, b.value) != (nv, ni):
                        raise SystemExit(f"sdfk_march_host mesh differs: {(a.value, b.value)} vs {(nv, ni)}")
                extra["host_values_march_ms"] = round(sorted(ts[1:])[1], 3)
                extra["host_values_march_what"] = ("sdfk_march_host on a host Values array (the route of an opaque delegate sampled on the CPU by the shim): "
                                                   f"upload of {n ** 3 * (16 if host_colors is not None else 4) / 1e6:.0f} MB + meshing + counts; median of 3 after a warm-up; "
                                                   f"all four: {[round(t, 3) for t in ts]}")
                del host_values, host_colors
            d2h = {}
            colors_written = bool(sdf.writes_color)

            def pipelined_handoff(k):
                """Hand-off THROUGHPUT: the next job is queued before the previous mesh is copied out, so the copy of mesh i (DMA + the
                host's share) overlaps the kernels of job i + 1 on the library's lanes -- what a host that meshes frame after frame
                gets per mesh with the reference-shaped calls in this order, pooled arrays (Mesh.Recycle)."""
                def submit():
                    m = C.c_void_p()
                    N.check(L.sdfk_sample_march(prog, N.f3(mn), N.f3(mx), n, n, n, 1 if clip else 0, C.c_float(0.0), 1, C.byref(m)))
                    return m
                prev = submit()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(k):
                    nxt = submit()
                    hm = Mesh._from_handle(prev, pool=pooled)
                    hm.Recycle()
                    prev = nxt
                Mesh._from_handle(prev, pool=pooled).Recycle()
                return (time.perf_counter() - t1) / (k + 1) * 1e3
            for kind in ("pooled", "pinned", "managed", "managed_recycled", "numpy"):
                ts = [one_call(kind) for _ in range(8)]
                d2h[kind] = {"median_ms": round(sorted(ts[1:])[len(ts[1:]) // 2], 3), "all_ms": [round(t, 3) for t in ts]}
            d2h["pooled"]["pool"] = {"hits": pooled.hits, "misses": pooled.misses}
            pipelined_handoff(3)
            extra["pipelined_handoff_ms_per_mesh"] = round(pipelined_handoff(10), 3)
            extra["pipelined_handoff_is"] = ("sample -> mesh -> the four arrays on the host, per mesh, when the next job is queued BEFORE the previous mesh is copied "
                                             "out (the copy overlaps the next job's kernels): 10 meshes, pooled exact-length arrays; "
                                             f"{(36 * nv + 4 * ni if colors_written else 24 * nv + 4 * ni) / 1e6:.1f} MB over PCIe per mesh")
            # (the key every round has carried = the DEFAULT API path: a caller of the reference-shaped API never recycles, so its four
            # arrays are fresh managed memory; the pooled figure -- the opt-in Mesh.Recycle(), not in the reference -- has its own key)
            extra["one_step_incl_mesh_d2h_ms"] = d2h["managed"]["median_ms"]
            extra["one_step_incl_mesh_d2h_pooled_ms"] = d2h["pooled"]["median_ms"]
            # ... and the same calls as a host that changes NOTHING gets them: the product default (SDFK_OPT_ELIDE_VOLUME = 2, the
            # temporary volume of SdfEx.ToMesh is not stored) -- the figures above keep the contract's stored volume like `value`
            if n ** 3 > (1 << 24):
                with N.option(N.OPT_ELIDE_VOLUME, 2):
                    for kind in ("pooled", "managed"):
                        ts = [one_call(kind) for _ in range(8)]
                        d2h[kind + "_product_default"] = {"median_ms": round(sorted(ts[1:])[len(ts[1:]) // 2], 3), "all_ms": [round(t, 3) for t in ts]}
                    pipelined_handoff(3)
                    extra["pipelined_handoff_product_default_ms_per_mesh"] = round(pipelined_handoff(10), 3)
                extra["one_step_incl_mesh_d2h_product_default_ms"] = d2h["managed_product_default"]["median_ms"]
                extra["one_step_incl_mesh_d2h_product_default_pooled_ms"] = d2h["pooled_product_default"]["median_ms"]
            extra["one_step_incl_mesh_d2h"] = {
                "what": "sdfk_sample_march + sdfk_mesh_counts + sdfk_mesh_copy of V/C/N/T to the host + sdfk_mesh_bounds, one call at a time; "
                        "median of 7 calls after a warm-up call (the second copy into an address is still slow -- the runtime maps the pages "
                        "for its first direct copy -- from the third on it is the steady state).  one_step_incl_mesh_d2h_ms = managed (the default API "
                        "path, comparable with rounds 1-3); one_step_incl_mesh_d2h_pooled_ms = pooled (what a host that meshes a grid shape "
                        "repeatedly and calls the opt-in Mesh.Recycle() gets) = the four exact-length arrays rented from a pool of arrays that earlier meshes of the "
                        "same size handed back (Mesh.Recycle; shim: MeshArrayPool), a miss -- the warm-up call here -- allocating "
                        "untouched memory; incl. all four arrays, bounds and stats, and the colour array cleared by the library.  "
                        "managed (the worst case of a C# caller, Mesh.cs:10-13) "
                        "= freshly mapped, never touched 4 KiB-page memory, as the new arrays of a GROWING managed heap are (the library "
                        "pre-faults it on its thread pool: the page faults are most of the time); managed_recycled = the same kind of memory "
                        "already touched, as arrays a managed heap hands out again are (resident pages: the library lets the runtime copy "
                        "straight into them); pinned = destination arrays from the library's pinned host arena (the Python mirror's Mesh, "
                        "Span<T> accessors of a shim); numpy = fresh numpy.empty arrays (huge-page advised)",
                **d2h}
    if rank == 0:
        nvox_rank = n * n * (D.slab(n, world, rank)[3] if world > 1 else n)
        colors = bool(sdf.writes_color)
        # ALGORITHMIC bytes per launch of the dominant kernel = SURVEY.md section 8(d)'s per-unit figure x the voxels of one launch: the
        # sampling kernel STORES 4 B/voxel of distance (+ 12 B/voxel of colour).  The 1/8 B/voxel of sign bytes the fused kernel also
        # leaves are this design's own (design_bytes_per_launch, and what the PMC `traffic` contains): not counted in `achieved`
        cands, design = {}, {}
        for k in kern:  # sdfk_sample_bits[_clip][_flat]: whichever entry point of the fused sampler ran
            if k.startswith("sdfk_sample_bits"):
                cands[k] = nvox_rank * (16 if colors else 4)
                design[k] = cands[k] + nvox_rank // 8
        dom = max(cands, key=lambda k: kern[k]["avg_us"]) if cands else None
        roof = None
        if dom:
            own = roof_us is not None
            us = roof_us if own else kern[dom]["avg_us"]
            if "sdfk_sample_colors" in kern and dom.startswith("sdfk_sample_bits_nc"):
                # a colour volume sampled in two passes (SDFK_OPT_COLOR_PASSES): the pair of kernels writes the 16 B/voxel between them
                if not own:
                    us += kern["sdfk_sample_colors"]["avg_us"]
                cands[dom + " + sdfk_sample_colors"] = cands.pop(dom)
                design[dom + " + sdfk_sample_colors"] = design.pop(dom)
                dom = dom + " + sdfk_sample_colors"
            ach = cands[dom] / (us * 1e-6) / 1e9
            roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": load_pmc_traffic(dom, args.scene, n),
                    "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of this command, committed; not re-measured in this run)",
                    "algorithmic_bytes_per_launch": cands[dom], "design_bytes_per_launch": design[dom],
                    "frac_design_bytes_per_launch": round(design[dom] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_us": round(us, 2),
                    "method": ("one HIP event pair on the launch stream around K back-to-back launches of the kernel alone (four resident "
                               "volumes in turn), divided by K" if own else
                               "hipEvent pairs around each launch on the launch stream, K-step pipeline pass on ONE in-order stream")
                              + "; the timed pass overlaps consecutive steps on three streams, which stretches every kernel"}
        step_s = dt / args.steps
        mesh_bytes = 36 * nv + 4 * ni
        contract_bytes = n ** 3 * (32 if colors else 8) + mesh_bytes
        # what THIS design has to move per step: the volume is stored once (4 B distance, +12 B colour) with 1/8 B of sign
        # bits per voxel, never re-read; plus the mesh
        design_bytes = n ** 3 * (16 if colors else 4) + n ** 3 // 8 + mesh_bytes
        measured_hbm = load_pmc_traffic("pipeline_step", args.scene, n)
        # the headline fraction: the contract's model (8 B/voxel: 4 stored by sampling + 4 loaded by meshing) for a distance-only
        # volume; for colour volumes that model (32 B/voxel) counts 16 B/voxel of loads the fused path never does -- fractions
        # above 1 -- so the design model stands there
        model_bytes = design_bytes if colors else contract_bytes

        def frac(nbytes, seconds):
            return None if not nbytes or not seconds else round(nbytes / seconds / 1e9 / HBM_PEAK_GBS, 4)

        lat_s = None if latency_ms is None else latency_ms * 1e-3
        # ---- the PRODUCT DEFAULT path (volume-less) with a bound of its own: its longest kernel is k_vertices, which neither moves many
        # bytes nor waits for them -- it is bound by VECTOR ISSUE (fp64 cell math the reference dictates); the fraction is the share of
        # the SIMDs' cycles in which a vector instruction was in flight, from the committed SQ counter pass of the volume-less run
        # (profiles/pmc_traffic.json, tools/gpu_elided_counters.sh), over the kernel's event-timed duration in THIS run
        elided_obj = None
        if elided_ms is not None and elided_kern:
            chain = [k for k in elided_kern if not k.startswith(("sdfk_cull", "sdfk_eval_blocks"))]
            kv = next((k for k in elided_kern if k.startswith("k_vertices")), None)
            act = load_pmc_traffic("k_vertices<true>:SQ_ACTIVE_INST_VALU", args.scene + "_elided", n)
            ins = load_pmc_traffic("k_vertices<true>:SQ_INSTS_VALU", args.scene + "_elided", n)
            busy = load_pmc_traffic("k_vertices<true>:SQ_BUSY_CYCLES", args.scene + "_elided", n)
            meas_e = load_pmc_traffic("pipeline_step", args.scene + "_elided", n)
            kv_s = None if not kv else elided_kern[kv]["avg_us"] * 1e-6
            serial_us = sum(v["avg_us"] for v in elided_kern.values())
            elided_obj = {
                "ms_per_step": round(elided_ms, 4), "mvoxels_per_s": round(n ** 3 / (elided_ms * 1e-3) / 1e6, 1),
                "latency_ms_single_call": None if latency_default_ms is None else round(latency_default_ms, 4),
                "kernels_us": elided_kern, "kernels_serial_sum_us": round(serial_us, 1),
                "chain_serial_us": round(sum(elided_kern[k]["avg_us"] for k in chain), 1),
                "measured_hbm_bytes": meas_e,
                "frac_measured_bytes": None if not meas_e else round(meas_e / (elided_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "roofline": None if not (kv and act) else {
                    "bound": "valu", "kernel": kv, "avg_launch_us": elided_kern[kv]["avg_us"],
                    # wavefront quad-cycles with a vector instruction in flight x 4 / (1024 SIMDs x 2.4 GHz x the launch): the share of
                    # the vector ALUs' time this kernel keeps busy (fp64 instructions occupy the pipe twice as long as fp32 ones)
                    "frac": round(act * 4 / 1024 / 2.4e9 / kv_s, 4),
                    "achieved": round(act * 4 / 1024 / kv_s / 1e9, 3), "peak": 2.4, "unit": "G busy SIMD-cycles/s per SIMD",
                    "valu_insts_per_launch": ins, "sq_busy_cycles_per_launch": busy,
                    "source": "SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU of k_vertices in a volume-less run (committed: profiles/pmc_traffic.json, "
                              "`...@sphere_elided@512`), duration event-timed in this run on one in-order stream"},
                "what": "the product default (SDFK_OPT_ELIDE_VOLUME = 2) on the headline grid: pipelined ms/step, one synchronous call, every kernel of "
                        "the volume-less step on one in-order stream (cull + eval of the surface blocks, then the meshing chain), and the bound of its "
                        "longest kernel.  The step as a whole is LATENCY-bound: seven dependent launches of 7-40 us each, ~50 MB of algorithmic bytes"}
        out = {
            "metric": "Mvoxels/s, 512^3 sphere SDF sample->mesh" if (n == 512 and args.scene == "sphere")
                      else f"Mvoxels/s, {n}^3 {args.scene} SDF sample->mesh",
            "value": round(n ** 3 / step_s / 1e6, 1),
            "unit": "Mvoxels/s",
            # (first among the extras: what the driver's command measures WITHOUT the clock warm-up `value` has -- the GPU idles, then
            # W warm-up steps and K timed ones)
            "value_cold_clocks": round(n ** 3 / (dt_cold / args.steps) / 1e6, 1),
            "ms_per_step_cold_clocks": round(dt_cold / args.steps * 1e3, 4),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            **({"NOT_THE_CONTRACT_STEP": "--elide: the volume is not stored (a profiling run of the product default)"} if (args.elide and args.minimal) else {}),
            "ms_per_step": round(ms_step, 4),
            "blocks": {"n": len(block_dts), "ms_per_step_median": round(ms_step, 4),
                       "ms_per_step_min": round(min(block_dts) / args.steps * 1e3, 4), "ms_per_step_max": round(max(block_dts) / args.steps * 1e3, 4),
                       "ms_per_step_all": [round(b / args.steps * 1e3, 4) for b in block_dts],
                       "what": f"{len(block_dts)} timed blocks of K = {args.steps} steps, each bracketed by barrier + synchronize on both sides (max over "
                               "ranks); `value` / `ms_per_step` = the median block"},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32 (sampling) / f64 (cell math)",
            "data": "synthetic",
            "config": {"workload": f"{WORKLOADS[args.scene]}"
                                   f", {n}^3 voxels, iso 0, step 1; Voxels.SampleSdf -> MarchingCubes.CreateMesh, device-resident",
                       "grid": [n, n, n], "vertices": nv, "triangles": ni // 3,
                       "parallelism": "1 GPU" if world == 1 else
                                      f"z-slab x{world}, slab meshes exchanged by the library over " + ("RCCL" if D.info()[2] == 1 else "its host transport (gloo; ranks share one GPU)")},
            "value_is": ("throughput of the pipelined steady state with warm clocks, median of the timed blocks (`blocks`): " +
                         ("five identical jobs in flight on the library's three internal streams" if not sharded else "three sharded steps in flight") +
                         ", buffers sized from the previous mesh of the shape; value_cold_clocks = the same K steps after the GPU idled (no clock "
                         "warm-up); latency_ms_single_stream / frac_single_call = one call at a time; first_call_ms = the first call"),
            "untimed_steps_before_timing": {"warmup": n_warm + n_est, "clock_warmup": n_clock,
                                            "why": "W + 4 steps fill the allocator's pools; then the same step runs for ~80 ms so that the GPU is at "
                                                   "its sustained clocks when the K timed steps start (after idling it needs 10-15 ms of load: "
                                                   "0.20 -> 0.169 ms per step at 512^3, tools/step_transient_probe.py); SDFK_BENCH_CLOCK_WARM_MS=0 switches it off; "
                                                   "value_cold_clocks is the figure without it"},
            "mtris_per_s": round(ni / 3 / step_s / 1e6, 2),
            "latency_ms_single_stream": None if latency_ms is None else round(latency_ms, 4),
            "latency_ms_single_stream_product_default": None if latency_default_ms is None else round(latency_default_ms, 4),
            "first_call_ms": first_call_ms,
            "first_call_new_constants_ms": first_call_new_constants_ms,   # the same structure, another radius: no compile (first_call.new_constants)
            "first_call": first,
            "stream_placement": N.stream_placement(),   # classes of lanes 0..4 / the exchange stream as measured at sdfk_init
            "pipeline_algorithmic_gbs": round(model_bytes / step_s / 1e9, 1),
            "pipeline_frac_of_hbm_peak": frac(model_bytes, step_s),
            # (N > 1: the same bytes of the WHOLE grid over the step time, against the N GPUs' HBM together -- BASELINE: "as fraction of the HBM roofline")
            "pipeline_frac_of_aggregate_hbm_peak": None if not frac(model_bytes, step_s) else round(frac(model_bytes, step_s) / world, 4),
            "pipeline_frac_is": ("contract model: 8 B/voxel (4 stored by sampling + 4 loaded by meshing) + 36 B/vertex + 4 B/index, divided by the step "
                                 "time and by 8 TB/s -- the fused path never re-reads the volume, so this is NOT achieved bandwidth: see "
                                 "frac_design_bytes / frac_measured_bytes" if not colors else
                                 "design model (colour volume): 16 B/voxel stored once + 1/8 B/voxel of sign bits + 36 B/vertex + 4 B/index; the "
                                 "contract's 32 B/voxel would count 16 B/voxel of loads this fused path never does (fractions above 1)"),
            "frac_design_bytes": frac(design_bytes, step_s),
            "frac_measured_bytes": frac(measured_hbm, step_s),
            "frac_single_call": frac(model_bytes, lat_s),
            "frac_single_call_measured_bytes": frac(measured_hbm, lat_s),
            "fracs_are": "bytes / time / 8 TB/s -- design = what this fused design must move per step ((4 | 16) + 1/8 B per voxel + the mesh); measured = "
                         "HBM bytes of one step from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json); single_call = the same byte "
                         "models over latency_ms_single_stream (one synchronous call, nothing else in flight) instead of the pipelined step",
            "pipeline_measured_hbm_bytes": measured_hbm,
            "pipeline_measured_hbm_gbs": None if not measured_hbm else round(measured_hbm / step_s / 1e9, 1),
            "kernels_us": kern,
            "roofline": roof,
            "c2_sphere_256": c2,
            "c3_repeatxy": c3,
            "c4_union8_1024": c4,
            "colour_sampler_control": long_fill,
            "c5_raymarch": c5,
            "elided_volume_ms_per_step": None if elided_ms is None else round(elided_ms, 4),
            "elided_volume_no_culling_ms_per_step": None if elided1_ms is None else round(elided1_ms, 4),
            "elided": elided_obj,
            "elided_volume_is": "SDFK_OPT_ELIDE_VOLUME (opt-in): the same K pipelined steps with a volume that is never stored -- the sampler "
                                "leaves sign bits only, corners and vertex colours are re-evaluated; meshes bit-identical.  = 2 (the first figure): "
                                "64 x 4 x 4 blocks whose values provably lie on one side of the iso value -- the program evaluated in interval "
                                "arithmetic over the block -- get constant sign words, only the blocks the surface passes through are evaluated "
                                "voxel by voxel; = 1 (no_culling): every voxel evaluated.  Not the contract's step (which evaluates and stores "
                                "every voxel): reported next to `value`, never as it",
        }
        out.update(extra)
        if dist_extra:
            out["sharded"] = dist_extra
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.scene, args.cpu_n or n, max(args.cpu_passes, 1))
    # ---- after the headline (N > 1).  The headline above used the plainest exchange.  The line is now SAFE: rank 0 leaves it in the file
    # its supervisor named, every rank leaves a marker -- a worker that dies or hangs from here on still counts as a finished headline
    # attempt (supervise_rank) -- and rank 0 rewrites the file after every further pass, so that whatever finished is reported.  Order =
    # least new ground first: (1) BASELINE C4 (1024^3 union of 8 primitives, the config that can scale) with the SAME exchange as the
    # headline; (2) the passes that leave the mesh sharded (exchange mode 3: headers only), headline grid and C4; (3) last, the tuner, which
    # tries the exchanges that have never run between two GPUs.  Every pass has a watchdog of its own.
    result_base = os.environ.get("SDFK_BENCH_RESULT_BASE")
    multi = sharded and world > 1
    after_on = multi and os.environ.get("SDFK_BENCH_NO_TUNE") != "1"
    tune_it = after_on and D.info()[2] == 1

    def save_line():
        if multi and result_base and rank == 0:
            with open(result_base + ".json.tmp", "w") as f:
                f.write(json.dumps(out) + "\n")
            os.replace(result_base + ".json.tmp", result_base + ".json")

    if multi:
        if rank == 0:
            out["sharded"]["tuned_pass"] = "did not finish (the headline pass above is unaffected)" if tune_it else "off (SDFK_BENCH_NO_TUNE / host transport)"
            out["sharded"]["mesh_stays_sharded_pass"] = "did not finish" if after_on else "off"
            out["sharded"]["c4_union8_1024"] = "did not finish" if after_on else "off"
        save_line()
        if result_base:
            open(f"{result_base}.rank{rank}.done", "w").close()

    import threading

    def watchdog(what, seconds):
        dog = threading.Timer(seconds, lambda: (print(f"bench.py: rank {rank}: {what} did not finish within {seconds:.0f} s", file=sys.stderr, flush=True), os._exit(3)))
        dog.daemon = True
        dog.start()
        return dog

    def timed_session_steps(ses, k):
        """W + 8 untimed steps, then EXACTLY k steps bracketed by barrier + synchronize, max over ranks (seconds)"""
        def step_():
            if ses.in_flight == ses.depth:
                ses.collect()
            ses.submit()
        for _ in range(max(args.warmup, 1) + 8):
            step_()
        ses.drain()
        barrier()
        t0 = time.perf_counter()
        for _ in range(k):
            step_()
        ses.drain()
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    # ---- (1) + (2b) BASELINE config C4 sharded: the only configuration of BASELINE.json where Z slabs can approach the >= 6x target --
    # a 1024^3 step is 3.7 ms on one GPU (0.97 with the product default), its slab meshes ~150 MB: the slabs' compute dominates the
    # exchange.  Grid: 1024 beside the 512^3 headline (SDFK_BENCH_C4_GRID overrides; 2 x --grid otherwise).
    c4_grid = int(os.environ.get("SDFK_BENCH_C4_GRID", "0")) or (1024 if n == 512 else 2 * n)
    c4_leg = None
    c4_state = {}

    def c4_pass(mode_key, mode):
        """K_c timed steps of the C4 grid in one exchange mode, its counts, stats and content check -> dict (rank 0; None elsewhere)"""
        k_c = int(os.environ.get("SDFK_BENCH_C4_STEPS", "10"))
        if "scene" not in c4_state:
            c4_state["scene"] = scene_for("union8")
        sdf_c, mn_c, mx_c, clip_c = c4_state["scene"]
        with N.option(N.OPT_DIST_EXCHANGE, mode):
            ses = D.SlabSession(sdf_c, mn_c, mx_c, c4_grid, c4_grid, c4_grid, clip_c, 0.0, depth=min(depth_env or 4, 3))
        try:
            dt_c = timed_session_steps(ses, k_c)
            ses.submit()
            ses.collect()
            counts_c = [list(c) for c in ses.counts()]
            st_c = ses.stats()
            chk = content_check(ses, sdf_c, mn_c, mx_c, clip_c, c4_grid, mode)
            # rank 0's single-GPU step of the SAME grid (stored volume, like the sharded step, and the product default), measured here
            # while the other ranks wait: the numerator of speedup_measured
            if "single_ms" not in c4_state and os.environ.get("SDFK_BENCH_NO_SINGLE") != "1":
                single = [None, None]
                if rank == 0:
                    s_step, s_drain, _ = single_gpu_stepper(3, sdf_c.program(), mn_c, mx_c, clip_c, c4_grid)
                    for which, mode_e in ((0, 0), (1, 2)):
                        with N.option(N.OPT_ELIDE_VOLUME, mode_e):
                            for _ in range(6):
                                s_step()
                            s_drain()
                            torch.cuda.synchronize()
                            t0 = time.perf_counter()
                            for _ in range(k_c):
                                s_step()
                            s_drain()
                            torch.cuda.synchronize()
                            single[which] = (time.perf_counter() - t0) / k_c * 1e3
                barrier()
                c4_state["single_ms"] = single
        finally:
            ses.close()
        if rank != 0:
            return None
        single = c4_state.get("single_ms", [None, None])
        ms_c = dt_c / k_c * 1e3
        return {"exchange": {-1: "host transport", 0: "ncclAllGather (in place)", 1: "grouped ncclSend / ncclRecv to every peer",
                             2: "gather to rank 0", 3: "headers only: the mesh stays sharded (payloads on demand)"}[st_c["exchange_mode"]] +
                            (" -- mesh stays sharded (SDFK_OPT_DIST_EXCHANGE = 3)" if mode == 3 and st_c["exchange_mode"] == -1 else ""),
                "steps": k_c, "ms_per_step": round(ms_c, 4), "mvoxels_per_s": round(c4_grid ** 3 / (ms_c * 1e-3) / 1e6, 1),
                "vertices": sum(c[0] for c in counts_c), "triangles": sum(c[1] for c in counts_c) // 3, "per_rank_vertices_indices": counts_c,
                "gather_stride_bytes_per_rank": int(st_c["stride_bytes"]),
                "bytes_received_per_rank": int((world - 1) * (st_c["stride_bytes"] if mode != 3 else 64)),
                "steps_redone": st_c["redone"], "stride_regrowths": st_c["regrown"],
                "single_gpu_ms_per_step": None if single[0] is None else round(single[0], 4),
                "single_gpu_product_default_ms_per_step": None if single[1] is None else round(single[1], 4),
                "speedup_measured": None if single[0] is None else round(single[0] / ms_c, 3),
                "speedup_vs_single_gpu_product_default": None if single[1] is None else round(single[1] / ms_c, 3),
                "mesh_equals_single_gpu": None if chk is None else chk["mesh_equals_single_gpu"], "content_check": chk}

    if after_on and os.environ.get("SDFK_BENCH_NO_C4") != "1":
        dog = watchdog("the C4 pass (default exchange)", float(os.environ.get("SDFK_BENCH_C4_TIMEOUT_S", "150")))
        try:
            barrier()
            leg = c4_pass("all_gather", N.get_option(N.OPT_DIST_EXCHANGE))
            if rank == 0:
                c4_leg = {"workload": WORKLOADS["union8"] + f", {c4_grid}^3, z-slab x{world}", "grid": c4_grid, "all_gather": leg,
                          "mesh_stays_sharded": "did not finish",
                          "what": "BASELINE config C4 sharded like the headline: K timed steps (same barriers, max over ranks) with the default "
                                  "exchange, then with exchange mode 3; speedup_measured = rank 0's single-GPU step of the same grid (volume "
                                  "stored, as in the slabs) / the sharded step; reported here only, never as `value`"}
                out["sharded"]["c4_union8_1024"] = c4_leg
        except BaseException as e:   # noqa: B902 -- reported; the headline is already safe
            if rank == 0:
                out["sharded"]["c4_union8_1024"] = {"failed": f"{type(e).__name__}: {e}"}
            print(f"bench.py: rank {rank} (device {local_rank}): C4 pass failed: {e}", file=sys.stderr, flush=True)
        dog.cancel()
        save_line()

    # ---- (2) "the mesh stays sharded" pass: the same K steps with SDFK_OPT_DIST_EXCHANGE = 3 -- a step moves the 64-byte headers
    # only, the slab meshes stay on their GPUs (sdfk_dist_slab_mesh / sdfk_dist_mesh fetch payloads on demand).  NOT the contract's step
    # (which all-gathers the slab meshes): reported beside it, after the headline is safe, under a watchdog of its own.
    sharded_result_pass = None
    if after_on and os.environ.get("SDFK_BENCH_NO_SHARDED_RESULT") != "1":
        dog = watchdog("the sharded-result pass", float(os.environ.get("SDFK_BENCH_TUNE_TIMEOUT_S", "90")))
        w3 = None
        try:
            barrier()
            with N.option(N.OPT_DIST_EXCHANGE, 3):
                w3 = D.SlabSession(sdf, mn, mx, n, n, n, clip, 0.0, depth=depth_env or 4)
            dt_c = timed_session_steps(w3, args.steps)
            w3.submit()
            w3.collect()
            c3_counts = [list(c) for c in w3.counts()]
            ok3 = (sum(c[0] for c in c3_counts), sum(c[1] for c in c3_counts)) == (nv, ni)
            chk3 = content_check(w3, sdf, mn, mx, clip, n, 3)
            sharded_result_pass = {"exchange": "headers only (SDFK_OPT_DIST_EXCHANGE = 3): the mesh stays sharded, payloads on demand",
                                   "ms_per_step": round(dt_c / args.steps * 1e3, 4), "value": round(n ** 3 / (dt_c / args.steps) / 1e6, 1),
                                   "counts_equal_the_headline_mesh": ok3,
                                   "mesh_equals_single_gpu": None if chk3 is None else chk3["mesh_equals_single_gpu"], "content_check": chk3,
                                   "speedup_measured": None if not dist_extra.get("single_gpu_ms_per_step") else
                                   round(dist_extra["single_gpu_ms_per_step"] / (dt_c / args.steps * 1e3), 3),
                                   "what": "K timed steps (same barriers, max over ranks) in which every rank meshes its slab and only the 64-byte "
                                           "headers cross the fabric; reported here only, never as `value`"}
        except BaseException as e:   # noqa: B902 -- reported; the headline is already safe
            sharded_result_pass = {"failed": f"{type(e).__name__}: {e}"}
            print(f"bench.py: rank {rank} (device {local_rank}): sharded-result pass failed: {e}", file=sys.stderr, flush=True)
        dog.cancel()
        try:
            if w3 is not None:
                w3.close()
        except BaseException:   # noqa: B902
            pass
        if rank == 0:
            out["sharded"]["mesh_stays_sharded_pass"] = sharded_result_pass
        save_line()
        if os.environ.get("SDFK_BENCH_NO_C4") != "1" and (rank != 0 or isinstance(c4_leg, dict)):
            dog = watchdog("the C4 pass (mesh stays sharded)", float(os.environ.get("SDFK_BENCH_C4_TIMEOUT_S", "150")))
            try:
                barrier()
                leg3 = c4_pass("mesh_stays_sharded", 3)
                if rank == 0:
                    c4_leg["mesh_stays_sharded"] = leg3
            except BaseException as e:   # noqa: B902
                if rank == 0:
                    c4_leg["mesh_stays_sharded"] = {"failed": f"{type(e).__name__}: {e}"}
                print(f"bench.py: rank {rank} (device {local_rank}): C4 pass (mode 3) failed: {e}", file=sys.stderr, flush=True)
            dog.cancel()
            save_line()

    # ---- (3) tuned pass (N > 1 over RCCL): the tuner (ncclAllGather against direct grouped sends, int32 against 16-bit indices, 20
    # pipelined steps each) and K timed steps in the configuration it keeps.
    tuned_pass = None
    if tune_it:
        dog = watchdog(f"the tuned pass (exchange mode {worker.stats()['exchange_mode']}, stride {worker.stats()['stride_bytes']} bytes)",
                       float(os.environ.get("SDFK_BENCH_TUNE_TIMEOUT_S", "90")))
        try:
            barrier()
            tuned = {f"mode{m}_{'compact' if c else 'plain'}": ns for (m, c), ns in worker.tune(20).items()}
            barrier()
            for _ in range(max(args.warmup, 1) + 8):
                step()
            drain()
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            drain()
            barrier()
            dt_b = max_over_ranks(time.perf_counter() - t0)
            st_b = worker.stats()
            step()
            drain()
            chk_b = content_check(worker, sdf, mn, mx, clip, n, st_b["exchange_mode"])
            tuned_pass = {"tuner_ns_per_20_steps": tuned,
                          "exchange": {0: "ncclAllGather (in place)", 1: "grouped ncclSend / ncclRecv to every peer", 2: "gather to rank 0"}[st_b["exchange_mode"]],
                          "payload": "compact (16-bit index offsets)" if st_b["index16"] else "plain (int32 indices)",
                          "ms_per_step": round(dt_b / args.steps * 1e3, 4), "value": round(n ** 3 / (dt_b / args.steps) / 1e6, 1),
                          "gather_stride_bytes_per_rank": int(st_b["stride_bytes"]),
                          "mesh_equals_single_gpu": None if chk_b is None else chk_b["mesh_equals_single_gpu"],
                          "what": "K timed steps (same barriers, max over ranks) in the exchange / payload form sdfk_dist_tune kept; reported "
                                  "here only, never as `value`"}
        except BaseException as e:   # noqa: B902 -- reported; the headline is already safe
            tuned_pass = {"failed": f"{type(e).__name__}: {e}"}
            print(f"bench.py: rank {rank} (device {local_rank}): tuned pass failed: {e}", file=sys.stderr, flush=True)
        dog.cancel()
        if rank == 0:
            out["sharded"]["tuned_pass"] = tuned_pass
        save_line()
    content_ok = True
    if rank == 0:
        if sharded:
            # every content check of this run: the headline's, the C4 passes', the mesh-stays-sharded pass's, the tuned pass's
            checks = {"headline": out["sharded"].get("mesh_equals_single_gpu")}
            for key in ("mesh_stays_sharded_pass", "tuned_pass"):
                if isinstance(out["sharded"].get(key), dict):
                    checks[key] = out["sharded"][key].get("mesh_equals_single_gpu")
            if isinstance(out["sharded"].get("c4_union8_1024"), dict):
                for key in ("all_gather", "mesh_stays_sharded"):
                    if isinstance(out["sharded"]["c4_union8_1024"].get(key), dict):
                        checks["c4_" + key] = out["sharded"]["c4_union8_1024"][key].get("mesh_equals_single_gpu")
            out["sharded"]["content_checks"] = checks
            content_ok = all(v is not False for v in checks.values())
            out["sharded"]["every_mesh_equals_single_gpu"] = content_ok
            save_line()
        print(json.dumps(out), flush=True)
    if sharded:
        worker.close()
        dist.barrier()
        D.shutdown()
        dist.destroy_process_group()
    if not content_ok:
        print("bench.py: a sharded mesh DIFFERS from the single-GPU mesh of the same grid (sharded.content_checks)", file=sys.stderr, flush=True)
        sys.exit(4)


def _worker_post_mortem(e):
    """A rank's measuring process is going down: say on stderr who it was and in what configuration -- rank, device, exchange,
    payload form, lanes, the library's last error (an RCCL failure carries RCCL's own error string) -- before the supervisors
    agree on the next, more conservative attempt."""
    try:
        from sdfkit_amd import _native as N
        opts = {k: N.get_option(getattr(N, k)) for k in ("OPT_DIST_EXCHANGE", "OPT_DIST_INDEX16", "OPT_DIST_LANES", "OPT_LANES", "OPT_GRAPHS", "OPT_HW_QUEUES")}
        err = N.lib().sdfk_last_error().decode("utf-8", "replace")
    except Exception as e2:   # (the library never came up)
        opts, err = {}, f"(library state unavailable: {e2})"
    print(f"bench.py: rank {os.environ.get('RANK', '0')} of {os.environ.get('WORLD_SIZE', '1')} (device {os.environ.get('LOCAL_RANK', '0')}) failed: "
          f"{type(e).__name__}: {e}; options {opts}; depth {os.environ.get('SDFK_BENCH_DEPTH', 'default')}; "
          f"attempt: {os.environ.get('SDFK_BENCH_NOTE', 'first (default exchange: ncclAllGather, plain payloads)')}; library: {err}",
          file=sys.stderr, flush=True)


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:
        if os.environ.get("SDFK_BENCH_WORKER") == "1":
            _worker_post_mortem(e)
        raise
