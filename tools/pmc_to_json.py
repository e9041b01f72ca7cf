#!/usr/bin/env python3
"""HBM bytes per launch (and per whole step) from the two rocprofv3 PMC passes (--pmc WRITE_SIZE, --pmc FETCH_SIZE)
-> profiles/pmc_traffic.json, the file bench.py reads `roofline.traffic` and `pipeline_measured_hbm_bytes` from.
Counters are in KiB; bytes = WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (MI355X_MICROARCH.md, HBM / rocprofv3 section:
on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads).
usage: pmc_to_json.py <write.csv> <fetch.csv> <scene> <grid edge> [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise  # noqa: E402

w, f, scene, n = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "pmc_traffic.json")
res = summarise([w, f])
d = json.load(open(out)) if os.path.exists(out) else {}
step = 0.0
for k, c in res.items():
    if "sdfk" not in k:          # torch's own kernels (the fill / copy context figures)
        continue
    b = c.get("WRITE_SIZE", 0.0) * 1024 + 2 * c.get("FETCH_SIZE", 0.0) * 1024
    name = k.replace("void ", "").replace("sdfk::", "").split("(")[0].strip()
    d[f"{name}@{scene}@{n}"] = int(round(b))
    # the kernels of a steady-state step of the fused path (k_publish, k_signbits8, k_gather_corners, k_clip* only run on
    # the exact / host-array / explicit-clip routes)
    if name.startswith(("sdfk_sample_bits", "k_bits_transpose", "k_compact", "sdfk_corners_eval", "sdfk_vertex_colors", "k_resolve", "k_vertices", "k_triangles")):
        step += b
d[f"pipeline_step@{scene}@{n}"] = int(round(step))
json.dump(d, open(out, "w"), indent=1, sort_keys=True)
print(f"pipeline_step@{scene}@{n} = {step / 1e6:.1f} MB")
