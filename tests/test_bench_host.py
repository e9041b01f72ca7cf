"""Host-side logic of bench.py that needs no GPU: the `--gpus N` self-launcher (the driver starts the
multi-GPU leg as plain `python bench.py --gpus N`) and the CPU-baseline leg (oracle, bounded sample)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_cpu_baseline_keeps_the_grid_edge_union8():
    # (the node ids of the 8-primitive union must not overwrite the grid edge)
    r = bench.cpu_baseline("union8", 32, timed=1)
    assert r["sample"].startswith("32^3"), r["sample"]
    assert r["kind"] == "port" and r["cores"] >= 1 and r["value"] > 0 and r["mtris_per_s"] > 0


@pytest.mark.parametrize("scene", ["sphere", "repeatxy"])
def test_cpu_baseline_scenes(scene):
    r = bench.cpu_baseline(scene, 24, timed=2)
    assert r["sample"].startswith("24^3") and "2 timed passes" in r["sample"]


def test_launcher_spawns_torchrun_as_a_child(monkeypatch):
    """WORLD_SIZE unset + --gpus 4: one child `python -m torch.distributed.run --nproc-per-node 4 bench.py ...`
    whose JSON line is relayed (the retries are the ranks' own business: supervise_rank)."""
    calls = []

    def fake_run(cmd, env, limit_s, graceful=False):
        calls.append((cmd, env, limit_s))
        return 0, '{"metric": "x", "n_gpus": 4}\n'

    monkeypatch.setattr(bench, "_run_ranks", fake_run)
    rc = bench.launch_ranks(["--gpus", "4", "--steps", "5"], 4)
    assert rc == 0 and len(calls) == 1
    cmd, env, limit = calls[0]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "5"] and cmd[-5].endswith("bench.py")
    assert limit >= len(bench.RANK_TRIES) * bench.attempt_limit_s()   # room for every attempt of the ranks' ladder


def _launch_with_stub(first_attempt, timeout_s, world=2):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SDFK_BENCH_NOTE", "SDFK_BENCH_WORKER"):
        env.pop(k, None)
    env.update({"SDFK_BENCH_WORKER_SCRIPT": os.path.join(ROOT, "tests", "bench_worker_stub.py"), "STUB_FIRST_ATTEMPT": first_attempt,
                "SDFK_BENCH_RANKS_TIMEOUT_S": str(timeout_s)})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0", "--no-cpu", "--grid", "32"],
                          env=env, capture_output=True, text=True, timeout=900)


@pytest.mark.parametrize("first_attempt", ["ok", "die", "hang"])
def test_rank_supervisors_agree_on_the_retry_ladder(first_attempt):
    """bench.py --gpus 2 -> torch.distributed.run -> two SUPERVISOR ranks (no HIP) -> two workers with a rendezvous of their
    own.  A worker that dies, or hangs and is killed at its time limit, makes BOTH supervisors start the next configuration
    of RANK_TRIES; rank 0 relays exactly one JSON line, from the attempt that succeeded."""
    import json
    p = _launch_with_stub(first_attempt, 25 if first_attempt == "hang" else 120)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["argv"][:2] == ["--gpus", "2"]
    if first_attempt == "ok":
        assert d["note"] == "" and d["depth"] == "" and d["tuned_pass"] == "done"
    else:
        assert d["note"].startswith("retry: one step in flight, plain ncclAllGather") and d["depth"] == "1"
        assert "multi-rank run failed" in p.stderr
        if first_attempt == "hang":
            assert "did not finish within 25 s" in p.stderr


@pytest.mark.parametrize("first_attempt", ["tune_die", "tune_hang"])
def test_a_failed_tuned_pass_does_not_cost_the_headline(first_attempt):
    """The headline pass of a multi-rank run uses the plainest exchange; the tuner -- which tries the exchanges that have never
    run between two GPUs -- comes AFTER rank 0 has left the finished line with its supervisor and every rank its marker.  A
    worker that dies or hangs in the tuned pass therefore still counts as a finished attempt: no retry, exactly one line, the
    headline's, saying that the tuned pass did not finish."""
    import json
    p = _launch_with_stub(first_attempt, 20 if first_attempt == "tune_hang" else 120)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["note"] == "" and d["depth"] == ""          # the FIRST attempt's line: nobody retried
    assert d["tuned_pass"] == "did not finish"
    assert "reporting the headline pass" in p.stderr and "multi-rank run failed" not in p.stderr


def test_a_wrong_sharded_mesh_fails_the_run_but_keeps_the_line():
    """`sharded.every_mesh_equals_single_gpu: false` (bench.py's content check: SHA-256 of the gathered mesh against the same grid meshed on
    one GPU): the line still reaches stdout -- it says which pass differed -- but the run exits non-zero, on every rank and in the
    launcher, and nobody retries with a more conservative exchange (that would hide a wrong mesh, not fix it)."""
    import json
    p = _launch_with_stub("wrong_mesh", 120)
    assert p.returncode == 4, (p.returncode, p.stderr[-2000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["sharded"]["every_mesh_equals_single_gpu"] is False and d["note"] == ""      # the first attempt's line: no retry
    assert bench._content_mismatch(lines[0]) and not bench._content_mismatch('{"sharded": {"every_mesh_equals_single_gpu": true}}')
    assert not bench._content_mismatch('{"metric": "x"}') and not bench._content_mismatch("not json")


def test_plain_python_with_gpus_gt_1_launches_before_touching_torch():
    """`python bench.py --gpus 2` must reach the launcher without importing torch in the parent: run it
    with a poisoned `torch` on the path of the PARENT only -- the launcher hands the children the same
    environment, so they fail, which is what this box (no GPU) would do anyway; the parent must exit with
    the children's failure, not with an ImportError of its own."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu", "--grid", "32"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0                      # no GPU here: the ranks refuse to run
    assert "bench.py needs an MI355X" in p.stderr and "retry" in p.stderr   # (every configuration of the ladder was tried)
    assert "launch with torch.distributed.run" not in p.stderr


def test_killing_the_launcher_takes_the_workers_down():
    """SIGTERM to `python bench.py --gpus 2` while its workers hang: the launcher kills torch.distributed.run's process group,
    the supervisors die with it, and every worker -- a session of its own -- dies with its supervisor (PR_SET_PDEATHSIG):
    nothing is left behind on the GPU."""
    import signal
    import time
    import psutil
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SDFK_BENCH_NOTE", "SDFK_BENCH_WORKER"):
        env.pop(k, None)
    env.update({"SDFK_BENCH_WORKER_SCRIPT": os.path.join(ROOT, "tests", "bench_worker_stub.py"), "STUB_FIRST_ATTEMPT": "hang",
                "SDFK_BENCH_RANKS_TIMEOUT_S": "600"})
    marker = "7919"
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", marker, "--warmup", "0", "--no-cpu", "--grid", "32"],
                         env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    def workers():
        found = []
        for q in psutil.process_iter(["cmdline"]):
            c = q.info["cmdline"] or []
            if any("bench_worker_stub.py" in a for a in c) and marker in c:
                found.append(q.pid)
        return found

    try:
        t0 = time.time()
        while len(workers()) < 2 and time.time() - t0 < 180:
            time.sleep(0.5)
        assert len(workers()) == 2, "the stub workers never came up"
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=60)
        t0 = time.time()
        while workers() and time.time() - t0 < 30:
            time.sleep(0.5)
        assert workers() == []
    finally:
        if p.poll() is None:
            p.kill()
        for pid in workers():
            os.kill(pid, signal.SIGKILL)     # (exact pids found above)


def test_sharded_content_comparison_is_right_without_a_gpu():
    """bench.compare_sharded_with_single -- what `bench.py --gpus N` decides `sharded.mesh_equals_single_gpu` with on first contact with a
    node of several GPUs -- on meshes from the CPU oracle: identical meshes and slabs say true / true; ONE flipped index, one flipped
    normal bit, a slab with slab-LOCAL indices (the rebase bug the round-5 advisor found), a missing slab and counts that do not add up
    each turn exactly the verdict they must."""
    import types
    import numpy as np
    from oracle import oracle as O
    s = O.Scene()
    s.sphere_w(1.0)
    mn, mx = [-1.5] * 3, [1.5] * 3
    v, c = O.sample(s, mn, mx, 24, 24, 24)
    m = O.march(v, c, mn, mx)

    def mesh(V, Cc, Nn, T):
        return types.SimpleNamespace(Vertices=np.ascontiguousarray(V, np.float32), Colors=np.ascontiguousarray(Cc, np.float32),
                                     Normals=np.ascontiguousarray(Nn, np.float32), Triangles=np.ascontiguousarray(T, np.int32))

    single = mesh(m.vertices, m.colors, m.normals, m.triangles)
    nv, ni = len(single.Vertices), len(single.Triangles)
    cuts_v, cuts_i = [0, nv // 3, nv // 3, nv], [0, (ni // 9) * 3, (ni // 9) * 3, ni]     # three slabs, the middle one empty
    counts = [(cuts_v[q + 1] - cuts_v[q], cuts_i[q + 1] - cuts_i[q]) for q in range(3)]

    def slabs_of(mesh_):
        out = []
        for q in range(3):
            a, b, i, j = cuts_v[q], cuts_v[q + 1], cuts_i[q], cuts_i[q + 1]
            out.append((bench.mesh_digest(mesh_.Vertices[a:b], mesh_.Colors[a:b], mesh_.Normals[a:b], mesh_.Triangles[i:j]), (b - a, j - i)))
        return out

    good = bench.compare_sharded_with_single(single, mesh(m.vertices, m.colors, m.normals, m.triangles), slabs_of(single), counts)
    assert good["mesh_equals_single_gpu"] is True and good["every_ranks_slab_equals_its_slice"] is True and good["sha256_sharded"] == "identical"
    assert (good["vertices"], good["indices"]) == (nv, ni) and len(good["sha256_single_gpu"]["Normals"]) == 64
    # the fault injection of the GPU test: one flipped index of the gathered mesh
    bad = bench.compare_sharded_with_single(single, mesh(m.vertices, m.colors, m.normals, m.triangles), slabs_of(single), counts, flip_one_index=True)
    assert bad["mesh_equals_single_gpu"] is False and bad["every_ranks_slab_equals_its_slice"] is True
    assert bad["sha256_sharded"]["Triangles"] != bad["sha256_single_gpu"]["Triangles"] and bad["sha256_sharded"]["Vertices"] == bad["sha256_single_gpu"]["Vertices"]
    # one bit of one normal
    nrm = np.ascontiguousarray(m.normals, np.float32).copy()
    nrm.view(np.uint32)[nv // 2, 1] ^= 1
    assert bench.compare_sharded_with_single(single, mesh(m.vertices, m.colors, nrm, m.triangles), slabs_of(single), counts)["mesh_equals_single_gpu"] is False
    # a slab handed out with slab-LOCAL indices (its own section never rebased): the whole mesh is right, that rank's slab is not
    local = slabs_of(single)
    t2 = single.Triangles[cuts_i[2]:cuts_i[3]] - cuts_v[2]
    local[2] = (bench.mesh_digest(single.Vertices[cuts_v[2]:], single.Colors[cuts_v[2]:], single.Normals[cuts_v[2]:], t2), local[2][1])
    r = bench.compare_sharded_with_single(single, mesh(m.vertices, m.colors, m.normals, m.triangles), local, counts)
    assert r["mesh_equals_single_gpu"] is True and r["every_ranks_slab_equals_its_slice"] is False
    # a rank whose slab never arrived; counts that do not add up to the single-GPU mesh
    missing = slabs_of(single)
    missing[1] = None
    assert bench.compare_sharded_with_single(single, single, missing, counts)["every_ranks_slab_equals_its_slice"] is False
    short = [counts[0], counts[1], (counts[2][0] - 1, counts[2][1])]
    assert bench.compare_sharded_with_single(single, single, slabs_of(single), short)["every_ranks_slab_equals_its_slice"] is False
    # a gathered mesh with a vertex too many is not the single-GPU mesh
    more = mesh(np.vstack([m.vertices, m.vertices[:1]]), np.vstack([m.colors, m.colors[:1]]), np.vstack([m.normals, m.normals[:1]]), m.triangles)
    assert bench.compare_sharded_with_single(single, more, slabs_of(single), counts)["mesh_equals_single_gpu"] is False
