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
if rank == 0:
    print(json.dumps({"metric": "stub", "n_gpus": world, "note": os.environ.get("SDFK_BENCH_NOTE", ""), "argv": sys.argv[1:],
                      "depth": os.environ.get("SDFK_BENCH_DEPTH", "")}), flush=True)
dist.destroy_process_group()
