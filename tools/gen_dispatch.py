#!/usr/bin/env python3
"""Extract the CONTROL FLOW of the reference's Lewiner case dispatcher into a manifest (sibling of gen_luts.py, which
extracts the table VALUES).

Runs ONLY in the build container (it reads /root/reference/SdfKit/MarchingCubes.cs, which does not exist on the GPU
box).  Output, committed: tests/golden/dispatch_manifest.json

What is taken from the reference is structure, mechanically parsed -- no C# text is stored:
  * "switch":  the statement tree of TheBigSwitch (MarchingCubes.cs:94-371) as nested lists
                 ["if", cond, then, else] / ["add", tiling table, sub index | null, nt] / ["inc" | "set" | "map", ...] / ["print"]
                 with cond = ["eq", var, n] | ["face", test table, column | null] | ["internal", test table, column | null]
  * "face_corners":  TestFace's face -> (A, B, C, D) corner numbers (MarchingCubes.cs:386-398)
  * "internal_edge_source":  which table entry gives TestInternal its reference edge per case (MarchingCubes.cs:432-436)
  * "internal_edges":  the 12 reference-edge rows (MarchingCubes.cs:438-511): t = v[a] / (v[a] - v[b] + eps),
                 Bt = v[B0] + (v[B1] - v[B0]) * t, Ct, Dt likewise -> [a, b, B0, B1, C0, C1, D0, D1]
  * "internal_result":  test value 0..15 -> "s>0" | "s<0" | ["det<eps" | "det>=eps", "s>0", fall-through] (MarchingCubes.cs:526-545)
  * "internal_case_4_10":  the corner numbers of the closed-form branch (MarchingCubes.cs:424-431), as parsed token lists

tests/test_dispatch_manifest.py interprets this manifest (a third, table-driven evaluator) and checks that the oracle's
orc_resolve_tiling and a host-compiled instantiation of csrc/mc_device.h take exactly these branches.
"""
import json
import os
import re
import sys

SRC = "/root/reference/SdfKit/MarchingCubes.cs"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def function_body(text, signature):
    i = text.index(signature)
    i = text.index("{", i)
    depth, j = 0, i
    while True:
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                return text[i + 1:j]
        j += 1


TOKEN = re.compile(r"\s*(==|\+=|>=|<=|&&|\|\||[A-Za-z_][A-Za-z_0-9.]*|-?\d+|\"[^\"]*\"|[{}()\[\],;=<>*+\-/])")


def tokenize(s):
    out, pos = [], 0
    s = s.strip()
    while pos < len(s):
        m = TOKEN.match(s, pos)
        if not m:
            raise SystemExit(f"cannot tokenize at: {s[pos:pos + 40]!r}")
        out.append(m.group(1))
        pos = m.end()
    return out


class Parser:
    def __init__(self, toks):
        self.t, self.i = toks, 0

    def peek(self, k=0):
        return self.t[self.i + k] if self.i + k < len(self.t) else None

    def eat(self, tok=None):
        cur = self.t[self.i]
        if tok is not None and cur != tok:
            raise SystemExit(f"expected {tok!r}, got {cur!r} at token {self.i}: {' '.join(self.t[max(0, self.i - 8):self.i + 8])}")
        self.i += 1
        return cur

    def block_or_stmt(self):
        if self.peek() == "{":
            self.eat("{")
            out = []
            while self.peek() != "}":
                out.extend(self.stmt())
            self.eat("}")
            return out
        return self.stmt()

    def lut_ref(self):
        """Luts.NAME[config] | Luts.NAME[config, C] -> (NAME, C | None)"""
        name = self.eat()
        assert name.startswith("Luts."), name
        self.eat("[")
        self.eat("config")
        col = None
        if self.peek() == ",":
            self.eat(",")
            col = int(self.eat())
        self.eat("]")
        return name[5:], col

    def cond(self):
        t = self.eat()
        if t in ("cas", "subconfig"):
            self.eat("==")
            return ["eq", t, int(self.eat())]
        if t == "TestFace":
            self.eat("(")
            self.eat("cell")
            self.eat(",")
            name, col = self.lut_ref()
            self.eat(")")
            return ["face", name, col]
        if t == "TestInternal":
            self.eat("(")
            for x in ("cell", ",", "cas", ",", "config", ",", "subconfig", ","):
                self.eat(x)
            name, col = self.lut_ref()
            self.eat(")")
            return ["internal", name, col]
        raise SystemExit(f"unknown condition starting with {t!r}")

    def stmt(self):
        t = self.peek()
        if t == "if":
            self.eat("if")
            self.eat("(")
            c = self.cond()
            self.eat(")")
            then = self.block_or_stmt()
            els = []
            if self.peek() == "else":
                self.eat("else")
                els = self.block_or_stmt()
            return [["if", c, then, els]]
        if t == "int":                      # int subconfig = 0;
            self.eat("int")
            self.eat("subconfig")
            self.eat("=")
            v = int(self.eat())
            self.eat(";")
            return [["set", "subconfig", v]]
        if t == "subconfig":
            self.eat("subconfig")
            op = self.eat()
            if op == "+=":
                v = int(self.eat())
                self.eat(";")
                return [["inc", "subconfig", v]]
            assert op == "=", op
            if self.peek().startswith("Luts."):
                name = self.eat()[5:]
                self.eat("[")
                self.eat("subconfig")
                self.eat("]")
                self.eat(";")
                return [["map", "subconfig", name]]
            v = int(self.eat())
            self.eat(";")
            return [["set", "subconfig", v]]
        if t in ("cell.AddTriangles", "cell.AddTriangles2"):
            self.eat()
            self.eat("(")
            name = self.eat()[5:]
            self.eat(",")
            self.eat("config")
            self.eat(",")
            idx = None
            if t.endswith("2"):
                idx = int(self.eat())
                self.eat(",")
            nt = int(self.eat())
            self.eat(")")
            self.eat(";")
            return [["add", name, idx, nt]]
        if t == "Console.WriteLine":
            while self.eat() != ";":
                pass
            return [["print"]]
        raise SystemExit(f"unknown statement starting with {t!r} at token {self.i}")


def parse_switch(text):
    body = function_body(text, "static void TheBigSwitch(")
    p = Parser(tokenize(body))
    out = []
    while p.peek() is not None:
        out.extend(p.stmt())
    return out


def parse_test_face(text):
    body = function_body(text, "static bool TestFace(")
    rows = {}
    for m in re.finditer(r"absFace\s*==\s*(\d)\)\s*\{\s*\(A,\s*B,\s*C,\s*D\)\s*=\s*\(cell\.v(\d),\s*cell\.v(\d),\s*cell\.v(\d),\s*cell\.v(\d)\)", body):
        rows[int(m.group(1))] = [int(m.group(k)) for k in (2, 3, 4, 5)]
    assert sorted(rows) == [1, 2, 3, 4, 5, 6], rows
    # the sign rule, as token lists (checked by the test against the evaluator's transcription)
    m = re.search(r"double AC_BD = (.*?);\s*if \((.*?)\) \{\s*return (.*?);\s*\} else \{\s*return (.*?);", body, flags=re.S)
    rule = {"det": tokenize(m.group(1)), "near_zero": tokenize(m.group(2)), "near_zero_result": tokenize(m.group(3)), "result": tokenize(m.group(4))}
    return rows, rule


def parse_test_internal(text):
    body = function_body(text, "static bool TestInternal(")
    src = {}
    for m in re.finditer(r"cas == (\d+)\) \{ edge = Luts\.(\w+)\[config,\s*(\w+)(?:,\s*(\d+))?\];", body):
        cas, name, a, b = int(m.group(1)), m.group(2), m.group(3), m.group(4)
        src[cas] = [name, int(a)] if b is None else [name, a, int(b)]
    assert sorted(src) == [6, 7, 12, 13], src
    v = r"cell\.v(\d)"
    edges = {}
    pat = (r"edge==(\d+)\) \{\s*t\s*=\s*" + v + r" / \( " + v + r" - " + v + r" \+ FLT_EPSILON \);\s*At = 0;\s*"
           r"Bt = " + v + r" \+ \( " + v + r" - " + v + r" \) \* t;\s*Ct = " + v + r" \+ \( " + v + r" - " + v + r" \) \* t;\s*"
           r"Dt = " + v + r" \+ \( " + v + r" - " + v + r" \) \* t;")
    for m in re.finditer(pat, body):
        g = [int(x) for x in m.groups()]
        e, a, a2, b, b0, b1, b0b, c0, c1, c0b, d0, d1, d0b = g
        assert a == a2 and b0 == b0b and c0 == c0b and d0 == d0b, g
        edges[e] = [a, b, b0, b1, c0, c1, d0, d1]
    assert sorted(edges) == list(range(12)), sorted(edges)
    # closed-form branch for cases 4 and 10
    m = re.search(r"if \(cas==4 \|\| cas==10\) \{(.*?)\} else if", body, flags=re.S)
    closed = {}
    for name, expr in re.findall(r"(\w+)\s*=\s*(.*?);", m.group(1)):
        closed[name] = tokenize(expr)
    guard = re.search(r"if \((t<0 \|\| t>1)\) return (s>0);", m.group(1))
    closed["out_of_range"] = [guard.group(1).replace(" ", ""), guard.group(2)]
    # result table
    res = {}
    tail = body[body.index("if (At >= 0) test += 1;"):]
    bits = re.findall(r"if \((\w)t >= 0\) test \+= (\d+);", tail)
    assert bits == [("A", "1"), ("B", "2"), ("C", "4"), ("D", "8")], bits
    for m in re.finditer(r"test==(\d+)\) \{\s*(?:return (s[<>]0);|if \(At \* Ct - Bt \* Dt\s*(<|>=)\s*FLT_EPSILON\) return (s[<>]0);)", tail):
        k = int(m.group(1))
        res[k] = m.group(2) if m.group(2) else ["det" + m.group(3) + "eps", m.group(4)]
    assert sorted(res) == list(range(16)), sorted(res)
    final = re.search(r"\}\s*return (s[<>]0);\s*$", tail.strip())
    return src, edges, closed, res, final.group(1)


def leaves(stmts, out):
    for s in stmts:
        if s[0] == "if":
            leaves(s[2], out)
            leaves(s[3], out)
        elif s[0] == "add":
            out.append(s)
    return out


def main():
    text = strip_comments(open(SRC).read())
    eps = re.search(r"FLT_EPSILON\s*=\s*([0-9.eE+\-]+)", text).group(1)
    tree = parse_switch(text)
    face, face_rule = parse_test_face(text)
    src, edges, closed, res, final = parse_test_internal(text)
    man = {
        "source": "SdfKit/MarchingCubes.cs (TheBigSwitch :94-371, TestFace :376-407, TestInternal :412-546)",
        "eps_literal": eps,
        "switch": tree,
        "n_add_leaves": len(leaves(tree, [])),
        "face_corners": {str(k): v for k, v in sorted(face.items())},
        "face_rule": face_rule,
        "internal_edge_source": {str(k): v for k, v in sorted(src.items())},
        "internal_edges": [edges[e] for e in range(12)],
        "internal_case_4_10": closed,
        "internal_bits": ["At", "Bt", "Ct", "Dt"],
        "internal_result": [res[k] for k in range(16)],
        "internal_fallthrough": final,
    }
    out = os.path.join(ROOT, "tests", "golden", "dispatch_manifest.json")
    with open(out, "w") as f:
        json.dump(man, f, sort_keys=True, separators=(",", ":"))
        f.write("\n")
    print(f"{out}: {man['n_add_leaves']} AddTriangles leaves, {len(edges)} edge rows, {len(face)} faces, eps {eps}")


if __name__ == "__main__":
    sys.exit(main())
