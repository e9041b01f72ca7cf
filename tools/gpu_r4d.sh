#!/bin/bash
# round 4: elided-volume tests + suite under SDFK_ELIDE_VOLUME=1, default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_elide_volume.py tests/test_mesh_pool.py tests/test_mesh_transform.py -m gpu -x -q > $O/pytest_elide.log 2>&1; echo "pytest rc $?" >> $O/pytest_elide.log
tail -12 $O/pytest_elide.log
SDFK_ELIDE_VOLUME=1 timeout 1700 python3 -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_alt_elide.log 2>&1; echo "pytest rc $?" >> $O/pytest_alt_elide.log
tail -5 $O/pytest_alt_elide.log
timeout 900 python3 bench.py --steps 20 --warmup 5 --cpu-n 128 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -3 $O/bench.err
python3 - <<'PY'
import json
ls=[l for l in open("gpurun_out/r04d/bench.json") if l.startswith("{")]
d=json.loads(ls[-1])
for k in ("value","ms_per_step","latency_ms_single_stream","first_call_ms","first_call_new_constants_ms","one_step_incl_mesh_d2h_ms","frac_measured_bytes","elided_volume_ms_per_step","c3_repeatxy"):
    print(k, d.get(k))
PY
