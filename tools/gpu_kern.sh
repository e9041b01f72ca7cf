#!/bin/bash
# per-kernel serial table (bench --minimal) for the sphere and the README scene at 512^3 and the sphere at 256^3, + a parity subset
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"; export TMPDIR=/tmp
O=gpurun_out/kern; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -x -k "not c4_union8_1024 and not c3_repeat" 2>&1 | tail -3
for cfg in "sphere 512" "repeatxy 512" "sphere 256"; do
  set -- $cfg
  timeout 300 python3 bench.py --steps 40 --warmup 5 --no-cpu --minimal --scene $1 --grid $2 > $O/b_$1_$2.json 2> $O/b_$1_$2.err
  python3 - "$O/b_$1_$2.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k=d["kernels_us"]
chain=sum(v["avg_us"] for n,v in k.items() if not n.startswith("sdfk_sample"))
print(sys.argv[1], "ms/step", d["ms_per_step"], "cold", d["ms_per_step_cold_clocks"], "| chain", round(chain,1), "|", " ".join(f"{n.replace('sdfk_','').replace('k_','')}={v['avg_us']}" for n,v in k.items()))
PY
done
