#!/usr/bin/env python3
"""Extract the numeric Lewiner-MC33 lookup tables into flat int8 blobs.

Runs ONLY in the build container (it reads /root/reference/SdfKit/Luts.cs, which
does not exist on the GPU box).  The outputs are committed:

  oracle/lewiner_luts.h            -- used by the CPU oracle (test infrastructure)
  sdfkit_amd/csrc/mc_luts.h        -- used by the HIP product path (__constant__)
  tests/golden/luts_manifest.json  -- name -> shape/offset/fnv1a, checked by tests

What is taken from the reference is DATA only (table values of Luts.cs:26-28 and
Luts.cs:53-2054; `casesClassic`, Luts.cs:2069, is dead in the reference and is
skipped).  The blob layout (one packed int8 array + an offset table) is ours.
"""
import json
import os
import re
import sys

SRC = "/root/reference/SdfKit/Luts.cs"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SKIP = {"casesClassic"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return text


def parse_tables(text):
    text = strip_comments(text)
    tables = []
    for m in re.finditer(r"public\s+static\s+sbyte\[([,]*)\]\s+(\w+)\s*=\s*\{", text):
        rank = len(m.group(1)) + 1
        name = m.group(2)
        # find the matching closing brace
        i = m.end() - 1
        depth = 0
        j = i
        while True:
            c = text[j]
            if c == "{":
                depth += 1
            elif c == "}":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        body = text[i:j + 1]
        shape = infer_shape(body, rank)
        vals = [int(v) for v in re.findall(r"-?\d+", body)]
        n = 1
        for s in shape:
            n *= s
        if n != len(vals):
            raise SystemExit(f"{name}: shape {shape} != {len(vals)} values (ragged?)")
        if any(v < -128 or v > 127 for v in vals):
            raise SystemExit(f"{name}: value out of int8 range")
        tables.append((name, shape, vals))
    return tables


def infer_shape(body, rank):
    """Shape of a rectangular nested-brace initialiser."""
    def parse(s, pos):
        # s[pos] == '{' ; returns (tree, newpos)
        assert s[pos] == "{"
        pos += 1
        items = []
        num = ""
        while True:
            c = s[pos]
            if c == "{":
                sub, pos = parse(s, pos)
                items.append(sub)
                continue
            if c == "}":
                if num.strip():
                    items.append(int(num))
                return items, pos + 1
            if c == ",":
                if num.strip():
                    items.append(int(num))
                num = ""
            else:
                num += c
            pos += 1
    tree, _ = parse(body, 0)
    shape = []
    t = tree
    while isinstance(t, list):
        shape.append(len(t))
        # rectangular check
        if t and isinstance(t[0], list):
            ln = {len(x) for x in t}
            if len(ln) != 1:
                raise SystemExit("ragged table")
        t = t[0] if t else None
    assert len(shape) == rank, (shape, rank)
    return shape


def fnv1a(vals):
    h = 0xCBF29CE484222325
    for v in vals:
        h ^= (v & 0xFF)
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


TILING_NT = {  # triangles per row of every tiling table (MarchingCubes.cs:94-371)
    "tiling1": 1, "tiling2": 2, "tiling3_1": 2, "tiling3_2": 4, "tiling4_1": 2, "tiling4_2": 6, "tiling5": 3,
    "tiling6_1_1": 3, "tiling6_1_2": 9, "tiling6_2": 5, "tiling7_1": 3, "tiling7_2": 5, "tiling7_3": 9,
    "tiling7_4_1": 5, "tiling7_4_2": 9, "tiling8": 2, "tiling9": 4, "tiling10_1_1": 4, "tiling10_1_1_": 4,
    "tiling10_1_2": 8, "tiling10_2": 8, "tiling10_2_": 8, "tiling11": 4, "tiling12_1_1": 4, "tiling12_1_1_": 4,
    "tiling12_1_2": 8, "tiling12_2": 8, "tiling12_2_": 8, "tiling13_1": 4, "tiling13_1_": 4, "tiling13_2": 6,
    "tiling13_2_": 6, "tiling13_3": 10, "tiling13_3_": 10, "tiling13_4": 12, "tiling13_5_1": 6, "tiling13_5_2": 10,
    "tiling14": 4,
}


def row_table(tables):
    """Derived index over all triangle rows: row id -> (blob offset, nt, occurrence word).
    occurrence word: 4 bits per vertex id 0..12 = how often the row references it."""
    rows, bases = [], {}
    off = 0
    for name, shape, vals in tables:
        if name in TILING_NT:
            nt = TILING_NT[name]
            rowlen = shape[-1]
            assert rowlen == 3 * nt, (name, shape, nt)
            nrows = len(vals) // rowlen
            bases[name] = len(rows)
            for r in range(nrows):
                row = vals[r * rowlen:(r + 1) * rowlen]
                occ = 0
                for e in range(13):
                    c = row.count(e)
                    assert c < 16
                    occ |= c << (4 * e)
                # distinct vertex ids in the order of their first reference (= creation order of the
                # vertices a cell owns, Cell.cs:272-359): 4 bits each from bit 0, their number in bits 60..63
                order = []
                for e in row:
                    if e not in order:
                        order.append(e)
                assert len(order) <= 13
                ord_word = len(order) << 60
                for k, e in enumerate(order):
                    ord_word |= e << (4 * k)
                rows.append((off + r * rowlen, nt, occ, ord_word))
        off += len(vals)
    return rows, bases


def emit_header(path, guard, prefix, tables, note):
    off = 0
    lines = []
    lines.append(f"/* GENERATED by tools/gen_luts.py -- do not edit.  {note}")
    lines.append(" * Values: Lewiner MC33 lookup tables as carried by the reference")
    lines.append(" * (SdfKit/Luts.cs:26-28, :53-2054).  Layout (single packed int8 blob +")
    lines.append(" * per-table offset/stride macros) is this repository's own. */")
    lines.append(f"#ifndef {guard}")
    lines.append(f"#define {guard}")
    lines.append("#include <stdint.h>")
    blob = []
    for name, shape, vals in tables:
        lines.append(f"#define {prefix}OFF_{name} {off}")
        for d, s in enumerate(shape):
            lines.append(f"#define {prefix}DIM{d}_{name} {s}")
        off += len(vals)
        blob.extend(vals)
    lines.append(f"#define {prefix}BLOB_SIZE {off}")
    rows, bases = row_table(tables)
    for name, b in bases.items():
        lines.append(f"#define {prefix}ROWBASE_{name} {b}")
    lines.append(f"#define {prefix}NROWS {len(rows)}")
    lines.append(f"#define {prefix}ROWOFF_VALUES " + ",".join(str(r[0]) for r in rows))
    lines.append(f"#define {prefix}ROWNT_VALUES " + ",".join(str(r[1]) for r in rows))
    lines.append(f"#define {prefix}ROWOCC_VALUES " + ",".join("0x%xull" % r[2] for r in rows))
    lines.append(f"#define {prefix}ROWORD_VALUES " + ",".join("0x%xull" % r[3] for r in rows))
    lines.append(f"#define {prefix}BLOB_VALUES \\")
    row = []
    chunks = []
    for v in blob:
        row.append(str(v))
        if len(row) == 32:
            chunks.append(",".join(row))
            row = []
    if row:
        chunks.append(",".join(row))
    lines.append(", \\\n".join("  " + c for c in chunks))
    lines.append(f"#endif /* {guard} */")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return off


def main():
    if not os.path.exists(SRC):
        sys.exit("reference not present: run in the build container")
    tables = [t for t in parse_tables(open(SRC).read()) if t[0] not in SKIP]
    n1 = emit_header(os.path.join(ROOT, "oracle", "lewiner_luts.h"),
                     "SDFK_ORACLE_LEWINER_LUTS_H", "OLUT_", tables,
                     "CPU oracle copy (test infrastructure).")
    n2 = emit_header(os.path.join(ROOT, "sdfkit_amd", "csrc", "mc_luts.h"),
                     "SDFK_MC_LUTS_H", "MCLUT_", tables,
                     "HIP product copy (__constant__ blob).")
    assert n1 == n2
    manifest = {}
    off = 0
    for name, shape, vals in tables:
        manifest[name] = {"shape": shape, "offset": off, "fnv1a64": f"{fnv1a(vals):016x}"}
        off += len(vals)
    manifest["_total"] = off
    gp = os.path.join(ROOT, "tests", "golden", "luts_manifest.json")
    os.makedirs(os.path.dirname(gp), exist_ok=True)
    with open(gp, "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(f"{len(tables)} tables, {off} bytes")


if __name__ == "__main__":
    main()
