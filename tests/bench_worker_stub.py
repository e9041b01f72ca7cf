"""Stand-in for the measuring process of bench.py (SDFK_BENCH_WORKER_SCRIPT), for tests/test_bench_host.py: joins the
workers' own gloo rendezvous, fails or hangs on request in the FIRST attempt, and lets rank 0 print a JSON line that says
which attempt it was.  No GPU, no library."""
import datetime
import json
import os
import sys
import time

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ.get("SDFK_BENCH_WORKER") == "1"
assert not any(k.startswith("TORCHELASTIC_") for k in os.environ)
first = "SDFK_BENCH_NOTE" not in os.environ
mode = os.environ.get("STUB_FIRST_ATTEMPT", "ok")
if first and mode == "die" and rank == world - 1:
    sys.exit(3)
if first and mode == "hang" and rank == world - 1:
    time.sleep(3600)
dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
t = torch.tensor([rank + 1])
dist.all_reduce(t)
assert int(t.item()) == world * (world + 1) // 2
line = {"metric": "stub", "n_gpus": world, "note": os.environ.get("SDFK_BENCH_NOTE", ""), "argv": sys.argv[1:],
        "depth": os.environ.get("SDFK_BENCH_DEPTH", "")}
# like the real worker: the finished headline goes to the supervisor's file, every rank leaves its marker, THEN the tuned pass
base = os.environ["SDFK_BENCH_RESULT_BASE"]
if rank == 0:
    with open(base + ".json", "w") as f:
        f.write(json.dumps(dict(line, tuned_pass="did not finish")) + "\n")
open(f"{base}.rank{rank}.done", "w").close()
dist.barrier()
if first and mode == "tune_die" and rank == world - 1:
    os._exit(3)
if first and mode == "tune_hang":      # (one rank hangs in the tuner's collective, the other waits for it: both are killed at the limit)
    time.sleep(3600)
if first and mode == "tune_die":       # (rank 0 waits for the dead rank in the next collective and fails)
    dist.barrier()
if mode == "wrong_mesh":               # (the real worker's content check found a sharded mesh that differs from the single-GPU one)
    line["sharded"] = {"every_mesh_equals_single_gpu": False, "content_checks": {"headline": False}}
if rank == 0:
    print(json.dumps(dict(line, tuned_pass="done")), flush=True)
dist.destroy_process_group()
if mode == "wrong_mesh":
    sys.exit(4)
