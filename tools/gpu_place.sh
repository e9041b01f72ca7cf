#!/bin/bash
# stream placement: what it finds in different process contexts, and what it is worth
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/place; mkdir -p $O
for pl in 0 1 1; do SDFK_STREAM_PLACEMENT=$pl python3 - <<'PY'
import time, os
from sdfkit_amd import _native as N
N.lib()
t0 = time.perf_counter(); N.init(0); print("placement", os.environ["SDFK_STREAM_PLACEMENT"], "sdfk_init ms", round((time.perf_counter() - t0) * 1e3, 1))
PY
done
python3 - <<'PY'
import time, torch
from sdfkit_amd import _native as N
t0 = time.perf_counter(); N.init(0); print("init ms", round((time.perf_counter() - t0) * 1e3, 1), "placement", N.stream_placement())
s = N.bind_torch_stream(torch.device("cuda", 0)); print("after binding torch's stream", N.stream_placement())
PY
timeout 600 python3 -m pytest tests/test_gpu_async.py tests/test_gpu_graphs.py -x -q 2>&1 | tail -3
for pl in 1 0; do for l in 3 4; do for g in 512 256; do
  echo "placement $pl lanes $l grid $g: $(SDFK_STREAM_PLACEMENT=$pl SDFK_LANES=$l timeout 300 python3 bench.py --steps 40 --warmup 5 --no-cpu --minimal --grid $g 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d.get('value_cold_clocks'))")"
done; done; done
for pl in 1 0; do for g in 128 512; do
  echo "placement $pl sharded world 1 grid $g: $(SDFK_STREAM_PLACEMENT=$pl SDFK_BENCH_FORCE_DIST=1 timeout 300 python3 bench.py --steps 200 --warmup 5 --no-cpu --minimal --grid $g 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'], d['sharded']['host_us_per_step']['submit'], d['sharded']['slab_kernels_only_ms'])")"
done; done
for pl in 1 0; do echo "placement $pl"; SDFK_STREAM_PLACEMENT=$pl PROBE_LANES=3,4,2 timeout 300 python3 tools/slab_chain_probe.py 2>&1 | grep "us per step"; done
