"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol
include/sdfkit_hip.h declares; SDF programs lower and JIT-compile for gfx950; the product
fails loudly without a GPU.  No compute calls are made here."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from sdfkit_amd import _native as N
from sdfkit_amd import Sdfs, SdfExprs, Vec3
from sdfkit_amd.api import _fmt_single
from tests import scenes as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "sdfkit_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdfk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = N.lib()
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/sdfkit_hip.h but not exported"
    assert set(syms) == set(N.SIGNATURES), set(syms) ^ set(N.SIGNATURES)
    # the dynamic symbol table holds EXACTLY the header's entry points: no mangled C++ members, no template
    # instantiations, no kernel handles (-fvisibility=hidden + csrc/exports.map)
    out = subprocess.check_output(["nm", "-D", "--defined-only", N.library_path()], text=True)
    rows = [l.split() for l in out.splitlines() if l.strip()]
    assert all(len(r) == 3 and r[1] == "T" for r in rows), [r for r in rows if len(r) != 3 or r[1] != "T"]
    assert sorted(r[2] for r in rows) == syms
    assert lib.sdfk_abi_version() == 6


def test_graft_entry_version_check_follows_the_header():
    """__graft_entry__.build() compares the built library with include/sdfkit_hip.h, not with a number of its own
    (an ABI bump must not break the driver's build check)."""
    import inspect
    import __graft_entry__ as G
    src = inspect.getsource(G.build)
    assert "SDFK_ABI_VERSION" in src and not re.search(r"sdfk_abi_version\(\)\s*==\s*\d", src)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sdfkit_hip.h")) as f:
        declared = int(re.search(r"#define\s+SDFK_ABI_VERSION\s+(\d+)", f.read()).group(1))
    assert N.lib().sdfk_abi_version() == declared


def test_library_contains_gfx950_code_object(tmp_path):
    # (--offloading extracts the bundles next to its input: work on a copy outside the tree)
    import shutil
    copy = shutil.copy(N.library_path(), str(tmp_path / "lib.so"))
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", copy],
                         capture_output=True, text=True, cwd=str(tmp_path)).stdout
    assert "gfx950" in out


@pytest.mark.parametrize("name", sorted(S.CATALOGUE))
def test_catalogue_lowers_and_compiles(name):
    _, sdf = S.CATALOGUE[name]()
    sdf.check()  # IR validation + hiprtc for gfx950, no device needed


def test_bad_program_is_rejected():
    import ctypes as C
    ops = (N.Op * 2)()
    ops[0].opcode, ops[0].a = 10, 5  # sqrt of a value that does not exist yet
    ops[1].opcode = 1
    out = (C.c_int32 * 4)(0, 0, 0, 1)
    st = N.lib().sdfk_program_check(ops, 2, out, 1)
    assert st == 1 and b"earlier value" in N.lib().sdfk_last_error()
    ops[0].opcode = 99
    assert N.lib().sdfk_program_check(ops, 2, out, 1) == 1


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_fails_loudly_without_gpu():
    with pytest.raises(N.SdfKitNativeError) as e:
        Sdfs.Sphere(1.0).ToMesh([-1] * 3, [1] * 3, 8, 8, 8)
    assert e.value.status == 2  # SDFK_ERR_NO_DEVICE: there is no CPU fallback
    # the same for the several-GPUs-from-one-process form: no device, no node (and no worker thread left behind)
    import ctypes as C
    import threading
    before = threading.active_count()
    h = C.c_void_p()
    assert N.lib().sdfk_node_open(None, 0, C.byref(h)) == 2 and not h.value
    assert threading.active_count() == before


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "sdfkit_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dp, f)).read()
                assert "oracle" not in text.lower().replace("no oracle", ""), f"{f} mentions the oracle"


def test_lut_blob_matches_manifest():
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "luts_manifest.json")))
    for path, prefix in (("oracle/lewiner_luts.h", "OLUT_"), ("sdfkit_amd/csrc/mc_luts.h", "MCLUT_")):
        text = open(os.path.join(ROOT, path)).read()
        body = text.split(prefix + "BLOB_VALUES")[1].split("#endif")[0]
        vals = [int(x) for x in re.findall(r"-?\d+", body)]
        assert len(vals) == man["_total"] == 13452
        for name, meta in man.items():
            if name.startswith("_"):
                continue
            n = int(np.prod(meta["shape"]))
            h = 0xCBF29CE484222325
            for v in vals[meta["offset"]:meta["offset"] + n]:
                h ^= v & 0xFF
                h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
            assert f"{h:016x}" == meta["fnv1a64"], name


def test_lut_invariant_edges_equal_sign_changes():
    """every tiling row references exactly the cube edges whose two
    corners differ in sign -- the property the parallel vertex-creation rule rests on."""
    man = json.load(open(os.path.join(ROOT, "tests", "golden", "luts_manifest.json")))
    text = open(os.path.join(ROOT, "sdfkit_amd/csrc/mc_luts.h")).read()
    vals = [int(x) for x in re.findall(r"-?\d+", text.split("MCLUT_BLOB_VALUES")[1].split("#endif")[0])]
    T = {k: np.array(vals[m["offset"]:m["offset"] + int(np.prod(m["shape"]))]).reshape(m["shape"])
         for k, m in man.items() if not k.startswith("_")}
    ends = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
    variants = {1: ["tiling1"], 2: ["tiling2"], 3: ["tiling3_1", "tiling3_2"], 4: ["tiling4_1", "tiling4_2"],
                5: ["tiling5"], 6: ["tiling6_1_1", "tiling6_1_2", "tiling6_2"],
                7: ["tiling7_1", "tiling7_2", "tiling7_3", "tiling7_4_1", "tiling7_4_2"], 8: ["tiling8"],
                9: ["tiling9"], 10: ["tiling10_1_1", "tiling10_1_1_", "tiling10_1_2", "tiling10_2", "tiling10_2_"],
                11: ["tiling11"], 12: ["tiling12_1_1", "tiling12_1_1_", "tiling12_1_2", "tiling12_2", "tiling12_2_"],
                13: ["tiling13_1", "tiling13_1_", "tiling13_2", "tiling13_2_", "tiling13_3", "tiling13_3_",
                     "tiling13_4", "tiling13_5_1", "tiling13_5_2"], 14: ["tiling14"]}
    checked = 0
    for index in range(256):
        cas, cfg = T["cases"][index]
        if cas == 0:
            continue
        want = {e for e, (a, b) in enumerate(ends) if ((index >> a) & 1) != ((index >> b) & 1)}
        for name in variants[int(cas)]:
            rows = T[name][cfg]
            rows = rows.reshape(-1, rows.shape[-1]) if rows.ndim > 1 else rows[None]
            for row in rows:
                assert {int(e) for e in row if e != 12} == want, (index, name)
                tri = row.reshape(-1, 3)
                assert all(len(set(t)) == 3 for t in tri.tolist())
                checked += 1
    assert checked == 728


def test_obj_number_formatting():
    assert _fmt_single(0.5) == "0.5" and _fmt_single(-1.0) == "-1" and _fmt_single(0.0) == "0"
    assert _fmt_single(np.float32(0.1)) == "0.1" and _fmt_single(np.float32(1e-5)) == "1E-05"
    assert _fmt_single(np.float32(123456.79)) == "123456.79" and _fmt_single(np.float32(1e20)) == "1E+20"
