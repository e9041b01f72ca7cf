#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/r04h; mkdir -p $O
timeout 1700 python3 -m pytest tests/test_gpu_elide_volume.py tests/test_gpu_param_programs.py -m gpu -x -q > $O/pytest_elide.log 2>&1; echo "pytest rc $?" >> $O/pytest_elide.log
tail -6 $O/pytest_elide.log
timeout 600 python3 tools/elide_kernels_probe.py 2>/dev/null | tee $O/elide_kernels.txt
timeout 900 python3 bench.py --steps 20 --warmup 5 --no-cpu > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 - <<'PY'
import json
ls=[l for l in open("gpurun_out/r04h/bench.json") if l.startswith("{")]
d=json.loads(ls[-1])
for k in ("value","ms_per_step","elided_volume_ms_per_step","elided_volume_no_culling_ms_per_step"):
    print(k, d.get(k))
c=d["c3_repeatxy"]; print({k:c[k] for k in ("ms_per_step","elided_volume_ms_per_step","elided_volume_no_culling_ms_per_step")})
PY
