#!/usr/bin/env python3
"""A tree-walking interpreter for the small imperative subset of C# that the reference's marching-cubes stage is written in
(SdfKit/MarchingCubes.cs, SdfKit/Cell.cs: classes with fields, properties and methods; int / double / float / bool locals;
arrays, List<T>, System.Numerics.Vector3; if / while / for; tuple assignment; casts).

Why: the image has no .NET, so the reference cannot be RUN here -- but its source can be EXECUTED: the generators
tools/gen_reference_vectors.py (MarchingCubes.CreateMesh on seeded volumes), gen_reference_sdf_vectors.py (the SdfFuncs
catalogue at seeded points: lambdas, closures, extension methods, overloads) and gen_reference_path_vectors.py (Voxels ->
SampleSdf -> ClipToBounds -> CreateMesh) parse the files where they lie under /root/reference, run them through this
interpreter and commit the inputs and outputs as golden vectors (tests/golden/reference_*.npz).  Nothing of the reference's
text is stored; this file contains no reference code.  Build container only (the GPU box has no /root/reference).

Numeric semantics (ours, as documented for the oracle in oracle/sdfk_oracle.h): int = Python int with C# truncating division;
double = Python float; float = numpy.float32, one rounding per operation; binary numeric promotion int < float < double;
implicit conversions on assignment to a typed variable / field / parameter; (float) of a double rounds to nearest even.
System.Numerics.Vector3 (BCL, not reference code): componentwise float32 arithmetic; Normalize(v) = v / Length(v) with
Length = sqrt((x x + y y) + z z) in float32 -- the restatement the oracle uses, see SURVEY.md section 8(c).
"""
import re

import numpy as np

F32 = np.float32

# ---------------------------------------------------------------------------------------------------------------------
# lexer
# ---------------------------------------------------------------------------------------------------------------------
TOKEN = re.compile(r"""
    (?P<ws>\s+|//[^\n]*|/\*.*?\*/)
  | (?P<num>\d+\.\d*(?:[eE][+-]?\d+)?[fFdD]?|\.\d+(?:[eE][+-]?\d+)?[fFdD]?|\d+(?:[eE][+-]?\d+)?[fFdD]?)
  | (?P<id>[A-Za-z_]\w*)
  | (?P<str>"(?:[^"\\]|\\.)*")
  | (?P<op>=>|==|!=|<=|>=|&&|\|\||\+\+|--|\+=|-=|\*=|/=|\?\.|[-+*/%<>=!?.,;:(){}\[\]])
""", re.S | re.X)


def lex(text):
    out, pos = [], 0
    while pos < len(text):
        m = TOKEN.match(text, pos)
        if not m:
            raise SyntaxError(f"cannot lex at {text[pos:pos + 40]!r}")
        pos = m.end()
        if m.lastgroup == "ws":
            continue
        out.append((m.lastgroup, m.group(m.lastgroup)))
    out.append(("eof", ""))
    return out


MODIFIERS = {"public", "private", "internal", "protected", "static", "readonly", "const", "sealed", "partial", "unsafe", "override", "virtual"}


# ---------------------------------------------------------------------------------------------------------------------
# parser -> nested tuples
# ---------------------------------------------------------------------------------------------------------------------
class Parser:
    def __init__(self, text):
        self.t = lex(text)
        self.i = 0

    # -- helpers
    def peek(self, k=0):
        return self.t[min(self.i + k, len(self.t) - 1)]

    def at(self, val, k=0):
        return self.peek(k)[1] == val and self.peek(k)[0] in ("op", "id")

    def eat(self, val=None):
        tok = self.t[self.i]
        if val is not None and tok[1] != val:
            ctx = " ".join(v for _, v in self.t[max(0, self.i - 8):self.i + 6])
            raise SyntaxError(f"expected {val!r}, got {tok[1]!r} near: {ctx}")
        self.i += 1
        return tok[1]

    def ident(self):
        kind, v = self.t[self.i]
        if kind != "id":
            raise SyntaxError(f"identifier expected, got {v!r}")
        self.i += 1
        return v

    # -- types: Name[<T,...>][?]([,*])*
    def try_type(self):
        save = self.i
        if self.peek()[0] != "id":
            return None
        name = self.ident()
        while self.at("."):
            if self.peek(1)[0] != "id":
                break
            self.eat(".")
            name += "." + self.ident()
        if self.at("<"):
            depth, j = 0, self.i
            while True:   # generic argument list: identifiers, commas, nested <>, ?, []
                kind, v = self.t[j]
                if v == "<":
                    depth += 1
                elif v == ">":
                    depth -= 1
                    if depth == 0:
                        break
                elif not (kind == "id" or v in (",", "?", "[", "]", ".")):
                    self.i = save
                    return None
                j += 1
            name += "<" + "".join(v for _, v in self.t[self.i + 1:j]) + ">"
            self.i = j + 1
        if self.at("?"):
            self.eat("?")
        rank = []
        while self.at("[") and (self.peek(1)[1] == "]" or self.peek(1)[1] == ","):
            self.eat("[")
            r = 1
            while self.at(","):
                self.eat(",")
                r += 1
            self.eat("]")
            rank.append(r)
        return ("type", name, tuple(rank))

    # -- file
    def parse_file(self):
        classes = {}
        while self.peek()[0] != "eof":
            if self.at("namespace"):
                self.eat()
                while not self.at(";") and not self.at("{"):
                    self.eat()
                self.eat()
                continue
            if self.at("using"):
                while not self.at(";"):
                    self.eat()
                self.eat(";")
                continue
            while self.peek()[1] in MODIFIERS:
                self.eat()
            if self.at("class") or self.at("struct"):
                self.eat()
                name = self.ident()
                if self.at(":"):
                    while not self.at("{"):
                        self.eat()
                classes[name] = self.parse_class_body(name)
                continue
            if self.at("}"):
                self.eat()
                continue
            raise SyntaxError(f"unexpected top-level token {self.peek()[1]!r}")
        return classes

    def parse_class_body(self, cname):
        self.eat("{")
        fields, methods, props = [], {}, {}
        while not self.at("}"):
            mods = set()
            while self.peek()[1] in MODIFIERS:
                mods.add(self.eat())
            if self.peek()[1] == cname and self.peek(1)[1] == "(":   # constructor
                self.eat()
                params = self.parse_params()
                chain = None
                if self.at(":"):   # : this(args) -- the other constructor runs first
                    self.eat(":")
                    self.eat("this")
                    self.eat("(")
                    chain = self.parse_args()
                body = self.parse_block()
                methods.setdefault("$all:.ctor", []).append(("method", ".ctor", params, body, mods, ("type", "void", ()), chain))
                methods.setdefault(".ctor", ("method", ".ctor", params, body, mods))
                continue
            ty = self.try_type()
            if ty is None:
                raise SyntaxError(f"member type expected in {cname}, got {self.peek()[1]!r}")
            name = self.ident()
            if self.at("("):
                params = self.parse_params()
                if self.at("=>"):
                    self.eat("=>")
                    e = self.parse_expr()
                    self.eat(";")
                    body = ("block", [("expr", e)] if ty[1] == "void" else [("return", e)])
                else:
                    body = self.parse_block()
                ext = bool(params) and params[0][0] == "$ext"
                if ext:
                    params = params[1:]
                    mods = mods | {"$ext"}
                methods.setdefault("$all:" + name, []).append(("method", name, params, body, mods, ty))
                methods.setdefault(name, ("method", name, params, body, mods, ty))
            elif self.at("=>"):
                self.eat("=>")
                e = self.parse_expr()
                self.eat(";")
                props[name] = ("block", [("return", e)])
            elif self.at("{") and self.peek(1)[1] == "get" and self.peek(2)[1] == ";":   # auto-property: a field
                while not self.at("}"):
                    self.eat()
                self.eat("}")
                fields.append((name, ty, None, "static" in mods))
            elif self.at("{"):
                self.eat("{")
                self.eat("get")
                props[name] = self.parse_block()
                if self.at("set"):
                    raise SyntaxError("property setters are not in the subset")
                self.eat("}")
            else:
                while True:
                    init = None
                    if self.at("="):
                        self.eat("=")
                        init = self.parse_expr()
                    fields.append((name, ty, init, "static" in mods or "const" in mods))
                    if self.at(","):
                        self.eat(",")
                        name = self.ident()
                        continue
                    break
                self.eat(";")
        self.eat("}")
        return {"fields": fields, "methods": methods, "props": props}

    def parse_params(self):
        self.eat("(")
        params = []
        while not self.at(")"):
            if self.at("this"):   # extension method: the receiver
                self.eat()
                params.append(("$ext", None))
            ty = self.try_type()
            name = self.ident()
            if self.at("="):   # default value
                self.eat("=")
                self.parse_expr()
            params.append((name, ty))
            if self.at(","):
                self.eat(",")
        self.eat(")")
        return params

    # -- statements
    def parse_block(self):
        self.eat("{")
        stmts = []
        while not self.at("}"):
            stmts.append(self.parse_stmt())
        self.eat("}")
        return ("block", stmts)

    def try_decl(self):
        """type name [= e] (, name [= e])* ;  -- or None (position restored)"""
        save = self.i
        ty = self.try_type()
        if ty is None or self.peek()[0] != "id" or self.peek(1)[1] not in ("=", ",", ";"):
            self.i = save
            return None
        if ty[1] in ("return", "new", "else"):
            self.i = save
            return None
        decls = []
        while True:
            name = self.ident()
            init = None
            if self.at("="):
                self.eat("=")
                init = self.parse_expr()
            decls.append((name, init))
            if self.at(","):
                self.eat(",")
                continue
            break
        return ("decl", ty, decls)

    def try_tuple_assign(self):
        save = self.i
        if not self.at("("):
            return None
        try:
            self.eat("(")
            lhs = [self.parse_unary()]
            while self.at(","):
                self.eat(",")
                lhs.append(self.parse_unary())
            if len(lhs) < 2 or not self.at(")") or self.peek(1)[1] != "=":
                raise SyntaxError("not a tuple assignment")
            self.eat(")")
            self.eat("=")
            self.eat("(")
            rhs = [self.parse_expr()]
            while self.at(","):
                self.eat(",")
                rhs.append(self.parse_expr())
            self.eat(")")
            if len(lhs) != len(rhs):
                raise SyntaxError("tuple arity")
            return ("tuple_assign", lhs, rhs)
        except SyntaxError:
            self.i = save
            return None

    def parse_stmt(self):
        if self.at("{"):
            return self.parse_block()
        if self.at(";"):
            self.eat(";")
            return ("block", [])
        if self.at("if"):
            self.eat()
            self.eat("(")
            c = self.parse_expr()
            self.eat(")")
            then = self.parse_stmt()
            els = None
            if self.at("else"):
                self.eat()
                els = self.parse_stmt()
            return ("if", c, then, els)
        if self.at("while"):
            self.eat()
            self.eat("(")
            c = self.parse_expr()
            self.eat(")")
            return ("while", c, self.parse_stmt())
        if self.at("for"):
            self.eat()
            self.eat("(")
            init = self.try_decl()
            if init is None and not self.at(";"):
                init = ("expr", self.parse_expr())
            self.eat(";")
            cond = None if self.at(";") else self.parse_expr()
            self.eat(";")
            incr = None if self.at(")") else self.parse_expr()
            self.eat(")")
            return ("for", init, cond, incr, self.parse_stmt())
        if self.at("return"):
            self.eat()
            e = None if self.at(";") else self.parse_expr()
            self.eat(";")
            return ("return", e)
        if self.at("var") and self.peek(1)[1] == "(":
            self.eat()
            self.eat("(")
            names = [self.ident()]
            while self.at(","):
                self.eat(",")
                names.append(self.ident())
            self.eat(")")
            self.eat("=")
            e = self.parse_expr()
            self.eat(";")
            return ("decl_tuple", names, e)
        d = self.try_decl()
        if d is not None:
            self.eat(";")
            return d
        t = self.try_tuple_assign()
        if t is not None:
            self.eat(";")
            return t
        e = self.parse_expr()
        self.eat(";")
        return ("expr", e)

    # -- expressions
    def parse_expr(self):
        lam = self.try_lambda()
        if lam is not None:
            return lam
        lhs = self.parse_or()
        if self.at("?") :
            self.eat("?")
            a = self.parse_expr()
            self.eat(":")
            b = self.parse_expr()
            return ("cond", lhs, a, b)
        if self.peek()[1] in ("=", "+=", "-=", "*=", "/="):
            op = self.eat()
            rhs = self.parse_expr()
            return ("assign", op, lhs, rhs)
        return lhs

    def try_lambda(self):
        """p => body | (p, q) => body, body an expression or a block"""
        save = self.i
        names = None
        if self.peek()[0] == "id" and self.peek(1)[1] == "=>" and self.peek()[1] not in ("new", "return"):
            names = [self.ident()]
        elif self.at("("):
            j, names2 = self.i + 1, []
            while self.t[j][0] == "id" and self.t[j + 1][1] in (",", ")"):
                names2.append(self.t[j][1])
                j += 2
                if self.t[j - 1][1] == ")":
                    break
            if names2 and self.t[j - 1][1] == ")" and self.t[j][1] == "=>":
                self.i = j
                names = names2
            elif self.t[self.i + 1][1] == ")" and self.t[self.i + 2][1] == "=>":
                self.i += 2
                names = []
        if names is None:
            self.i = save
            return None
        self.eat("=>")
        if self.at("{"):
            return ("lambda", names, self.parse_block())
        return ("lambda", names, ("block", [("return", self.parse_expr())]))

    def _binary(self, ops, sub):
        e = sub()
        while self.peek()[0] == "op" and self.peek()[1] in ops:
            op = self.eat()
            e = ("bin", op, e, sub())
        return e

    def parse_or(self):
        return self._binary(("||",), self.parse_and)

    def parse_and(self):
        return self._binary(("&&",), self.parse_eq)

    def parse_eq(self):
        return self._binary(("==", "!="), self.parse_rel)

    def parse_rel(self):
        return self._binary(("<", ">", "<=", ">="), self.parse_add)

    def parse_add(self):
        return self._binary(("+", "-"), self.parse_mul)

    def parse_mul(self):
        return self._binary(("*", "/", "%"), self.parse_unary)

    def parse_unary(self):
        if self.at("-"):
            self.eat()
            return ("neg", self.parse_unary())
        if self.at("+"):
            self.eat()
            return self.parse_unary()
        if self.at("!"):
            self.eat()
            return ("not", self.parse_unary())
        if self.at("++") or self.at("--"):
            op = self.eat()
            return ("preinc", op, self.parse_unary())
        if self.at("(") and self.peek(1)[1] in ("float", "double", "int") and self.peek(2)[1] == ")":
            self.eat("(")
            ty = self.eat()
            self.eat(")")
            return ("cast", ty, self.parse_unary())
        return self.parse_postfix()

    def parse_args(self, close=")"):
        args = []
        while not self.at(close):
            if self.at("out"):   # out var name | out name: the callee's result lands in that variable
                self.eat()
                declare = False
                if self.at("var") or (self.peek()[0] == "id" and self.peek(1)[0] == "id"):
                    self.eat()
                    declare = True
                args.append(("outarg", self.ident(), declare))
            else:
                args.append(self.parse_expr())
            if self.at(","):
                self.eat(",")
        self.eat(close)
        return args

    def parse_postfix(self):
        e = self.parse_primary()
        while True:
            if self.at("."):
                self.eat(".")
                e = ("member", e, self.ident())
                if self.at("<"):   # explicit generic arguments, if that is what they are: only type-ish tokens up to '>' '('
                    j, depth = self.i, 0
                    while True:
                        kind, v = self.t[j]
                        if v == "<":
                            depth += 1
                        elif v == ">":
                            depth -= 1
                            if depth == 0:
                                break
                        elif not (kind == "id" or v in (",", "(", ")", "[", "]", ".", "?")):
                            j = -1
                            break
                        j += 1
                    if j > 0 and self.t[j + 1][1] == "(":
                        self.i = j + 1
            elif self.at("?."):
                self.eat("?.")
                name = self.ident()
                self.eat("(")
                e = ("nullcall", e, name, self.parse_args())
            elif self.at("("):
                self.eat("(")
                e = ("call", e, self.parse_args())
            elif self.at("["):
                self.eat("[")
                e = ("index", e, self.parse_args("]"))
            elif self.at("++") or self.at("--"):
                e = ("postinc", self.eat(), e)
            else:
                return e

    def parse_primary(self):
        kind, v = self.peek()
        if kind == "num":
            self.eat()
            if v[-1] in "fF":
                return ("lit", F32(v[:-1]))
            if v[-1] in "dD":
                return ("lit", float(v[:-1]))
            if "." in v or "e" in v or "E" in v:
                return ("lit", float(v))
            return ("lit", int(v))
        if kind == "str":
            self.eat()
            return ("lit", v[1:-1])
        if v == "(":
            self.eat("(")
            e = self.parse_expr()
            if self.at(","):   # tuple value (only as the right side of a deconstruction; handled there)
                items = [e]
                while self.at(","):
                    self.eat(",")
                    items.append(self.parse_expr())
                self.eat(")")
                return ("tuple", items)
            self.eat(")")
            return ("paren", e)
        if v == "typeof":
            self.eat()
            self.eat("(")
            ty = self.try_type()
            self.eat(")")
            return ("lit", ("typeof", ty[1]))
        if v == "new" and self.peek(1)[1] == "[" and self.peek(2)[1] == "]":   # new[] { a, b }: an implicitly typed array
            self.eat()
            self.eat("[")
            self.eat("]")
            self.eat("{")
            return ("arraylit", self.parse_args("}"))
        if v == "new":
            self.eat()
            if self.at("("):   # target-typed new(capacity): an empty List<T>
                self.eat("(")
                self.parse_args()
                return ("newlist",)
            ty = self.try_type_for_new()
            if self.at("{"):   # object initialiser
                self.eat("{")
                inits = []
                while not self.at("}"):
                    name = self.ident()
                    self.eat("=")
                    inits.append((name, self.parse_expr()))
                    if self.at(","):
                        self.eat(",")
                self.eat("}")
                return ("newinit", ty, inits)
            if self.at("["):
                self.eat("[")
                dims = self.parse_args("]")
                return ("newarray", ty, dims[0]) if len(dims) == 1 else ("newarray_nd", ty, dims)
            self.eat("(")
            return ("new", ty, self.parse_args())
        if v in ("true", "false"):
            self.eat()
            return ("lit", v == "true")
        if v == "null":
            self.eat()
            return ("lit", None)
        if v == "this":
            self.eat()
            return ("this",)
        if kind == "id":
            self.eat()
            return ("name", v)
        raise SyntaxError(f"unexpected token {v!r}")

    def try_type_for_new(self):
        name = self.ident()
        while self.at(".") and self.peek(1)[0] == "id":
            self.eat(".")
            name = self.ident()   # (the last component names the class)
        if self.at("<"):
            self.eat("<")
            depth = 1
            while depth:
                v = self.eat()
                depth += (v == "<") - (v == ">")
        return name


# ---------------------------------------------------------------------------------------------------------------------
# values
# ---------------------------------------------------------------------------------------------------------------------
class Vec3:
    """System.Numerics.Vector3: three float32, value semantics."""
    __slots__ = ("X", "Y", "Z")

    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.X, self.Y, self.Z = F32(x), F32(y), F32(z)

    def __repr__(self):
        return f"<{self.X}, {self.Y}, {self.Z}>"

    def Length(self):   # BCL: sqrt(Dot(v, v)), the products summed left to right in float32
        with np.errstate(all="ignore"):
            return F32(np.sqrt(F32(F32(F32(self.X * self.X) + F32(self.Y * self.Y)) + F32(self.Z * self.Z))))


class Vec4:
    """System.Numerics.Vector4: four float32, value semantics."""
    __slots__ = ("X", "Y", "Z", "W")

    def __init__(self, *a):
        if len(a) == 2:   # (Vector3, w)
            a = (a[0].X, a[0].Y, a[0].Z, a[1])
        self.X, self.Y, self.Z, self.W = (F32(q) for q in a)


class Closure:
    def __init__(self, interp, names, body, fr):
        self.interp, self.names, self.body = interp, names, body
        self.fr = {"this": fr["this"], "cname": fr["cname"], "vars": list(fr["vars"]), "types": list(fr["types"])}

    def __call__(self, *args):
        fr = {"this": self.fr["this"], "cname": self.fr["cname"], "vars": self.fr["vars"] + [dict(zip(self.names, args))],
              "types": self.fr["types"] + [{}]}
        try:
            self.interp.exec(self.body, fr)
        except ReturnEx as r:
            return r.v
        return None


class CsList(list):
    pass


class SpanHost:
    """Span<T> over a slice of a Python list: element access goes to the underlying storage."""
    def __init__(self, store, start, n):
        self.store, self.start, self.Length = store, start, n

    def __getitem__(self, i):
        if not 0 <= i < self.Length:
            raise IndexError(i)
        return self.store[self.start + i]

    def __setitem__(self, i, v):
        if not 0 <= i < self.Length:
            raise IndexError(i)
        self.store[self.start + i] = v


class MemoryHost:
    """Memory<T>: Slice, Length, Span"""
    def __init__(self, store, start, n):
        self.store, self.start, self.Length = store, start, n

    def Slice(self, start, n):
        if start < 0 or n < 0 or start + n > self.Length:
            raise IndexError((start, n))
        return MemoryHost(self.store, self.start + start, n)

    @property
    def Span(self):
        return SpanHost(self.store, self.start, self.Length)


class Opaque:
    """Host stand-in whose every operation yields itself (the matrices after the point where the vectors are taken)."""
    def __getattr__(self, name):
        return self


class Instance:
    def __init__(self, cls, cname):
        self.cls, self.cname, self.f, self.ftype = cls, cname, {}, {}


class ReturnEx(Exception):
    def __init__(self, v):
        self.v = v


def is_f64(v):
    return isinstance(v, float) and not isinstance(v, F32)


def value_copy(v):
    """Vector3 / Vector4 are structs: assignment copies."""
    if isinstance(v, Vec3):
        return Vec3(v.X, v.Y, v.Z)
    if isinstance(v, Vec4):
        return Vec4(v.X, v.Y, v.Z, v.W)
    if not isinstance(v, Opaque) and hasattr(v, "cs_copy"):   # a host struct (Matrix4x4)
        return v.cs_copy()
    return v


class OutRef:
    """an `out` argument: the host callee assigns .value"""
    def __init__(self):
        self.value = None


def coerce(v, ty):
    """Implicit conversion on assignment to a variable of declared type `ty` (None / var: as is)."""
    v = value_copy(v)
    if ty is None or v is None:
        return v
    name, rank = ty[1], ty[2]
    if rank:
        return v
    if name == "double" and isinstance(v, (int, float, F32)) and not isinstance(v, bool):
        return float(v)
    if name == "float" and isinstance(v, (int, F32)) and not isinstance(v, bool):
        return F32(v)
    if name == "float" and is_f64(v):
        raise TypeError("implicit double -> float conversion")
    if name in ("int", "sbyte") and isinstance(v, (np.integer,)):
        return int(v)
    return v


def default_of(ty):
    name, rank = ty[1], ty[2]
    if rank:
        return None
    return {"int": 0, "sbyte": 0, "double": 0.0, "float": F32(0), "bool": False, "Vector3": Vec3()}.get(name)


def arith(op, a, b):
    if isinstance(a, Opaque) or isinstance(b, Opaque):
        return a if isinstance(a, Opaque) else b
    if op == "*" and hasattr(a, "cs_mul"):
        return a.cs_mul(b)
    if isinstance(a, Vec3) or isinstance(b, Vec3):
        if isinstance(a, Vec3) and isinstance(b, Vec3):
            if op == "+":
                return Vec3(a.X + b.X, a.Y + b.Y, a.Z + b.Z)
            if op == "-":
                return Vec3(a.X - b.X, a.Y - b.Y, a.Z - b.Z)
            if op == "*":
                return Vec3(a.X * b.X, a.Y * b.Y, a.Z * b.Z)
        if op in ("*", "/"):
            v, s = (a, b) if isinstance(a, Vec3) else (b, a)
            if is_f64(s):
                raise TypeError("Vector3 with a double")
            s = F32(s)
            if op == "*":
                return Vec3(v.X * s, v.Y * s, v.Z * s)
            if isinstance(a, Vec3):
                return Vec3(v.X / s, v.Y / s, v.Z / s)
        raise TypeError(f"Vector3 {op}")
    for v in (a, b):
        if isinstance(v, bool) or not isinstance(v, (int, float, F32, np.integer)):
            raise TypeError(f"arithmetic on {type(a).__name__} {op} {type(b).__name__}")
    if is_f64(a) or is_f64(b):
        a, b = float(a), float(b)
        with np.errstate(all="ignore"):
            if op == "+":
                return a + b
            if op == "-":
                return a - b
            if op == "*":
                return a * b
            if op == "/":
                return float(np.float64(a) / np.float64(b))   # (IEEE: inf / nan instead of ZeroDivisionError)
            if op == "%":
                return float(np.fmod(a, b))
    if isinstance(a, F32) or isinstance(b, F32):
        a, b = F32(a), F32(b)
        with np.errstate(all="ignore"):
            if op == "+":
                return F32(a + b)
            if op == "-":
                return F32(a - b)
            if op == "*":
                return F32(a * b)
            if op == "/":
                return F32(a / b)
            if op == "%":
                return F32(np.fmod(a, b))
    a, b = int(a), int(b)
    if op == "+":
        return a + b
    if op == "-":
        return a - b
    if op == "*":
        return a * b
    if op == "/":
        q = abs(a) // abs(b)
        return q if (a >= 0) == (b >= 0) else -q
    if op == "%":
        q = abs(a) // abs(b)
        q = q if (a >= 0) == (b >= 0) else -q
        return a - q * b
    raise TypeError(op)


def compare(op, a, b):
    if is_f64(a) or is_f64(b):
        a, b = float(a), float(b)
    elif isinstance(a, F32) or isinstance(b, F32):
        a, b = F32(a), F32(b)
    return bool({"<": a < b, ">": a > b, "<=": a <= b, ">=": a >= b, "==": a == b, "!=": a != b}[op])


# ---------------------------------------------------------------------------------------------------------------------
# evaluator
# ---------------------------------------------------------------------------------------------------------------------
class Interp:
    def __init__(self, classes, hosts):
        self.classes = classes          # name -> parsed class
        self.hosts = hosts              # name -> Python object (static stand-ins: Luts, Math, Console, Vector3, Matrix4x4, Mesh ...)
        self.statics = {}               # class name -> {field: value}
        self.static_imports = []        # `using static` classes: their methods are callable by bare name
        self.console = []
        for cname, c in classes.items():
            st = {}
            for name, ty, init, is_static in c["fields"]:
                if is_static:
                    st[name] = coerce(self.eval(init, {"this": None, "cname": cname, "vars": [{}], "types": [{}]}), ty) if init is not None else default_of(ty)
            self.statics[cname] = st

    # -- objects
    def new(self, cname, args):
        c = self.classes[cname]
        o = Instance(c, cname)
        frame = {"this": o, "cname": cname, "vars": [{}], "types": [{}]}
        for name, ty, init, is_static in c["fields"]:
            if is_static:
                continue
            o.ftype[name] = ty
            o.f[name] = coerce(self.eval(init, frame), ty) if init is not None else default_of(ty)
            if init is not None and init[0] == "newlist":
                o.f[name] = CsList()
        ctors = c["methods"].get("$all:.ctor", [])
        if ctors:
            self.construct(o, cname, ctors, args)
        hook = getattr(self, "on_new", {}).get(cname)
        if hook:
            hook(o)
        return o

    def construct(self, o, cname, ctors, args):
        cands = [m for m in ctors if len(m[2]) == len(args)]
        if len(cands) > 1:
            cands = [m for m in cands if all(self._matches(ty, v) for (_, ty), v in zip(m[2], args))] or cands
        if not cands:
            raise TypeError(f"no constructor of {cname} takes {len(args)} arguments")
        m = cands[0]
        if len(m) > 6 and m[6] is not None:   # : this(...) -- evaluated with the parameters in scope
            frame = {"this": o, "cname": cname, "vars": [{name: coerce(v, ty) for (name, ty), v in zip(m[2], args)}], "types": [{}]}
            self.construct(o, cname, ctors, [self.eval(a, frame) for a in m[6]])
        self.invoke(o, cname, m, args)

    def invoke(self, this, cname, method, args):
        params, body = method[2], method[3]
        if len(params) != len(args):
            raise TypeError(f"{cname}.{method[1]}: {len(args)} arguments for {len(params)} parameters")
        frame = {"this": this, "cname": cname, "vars": [{}], "types": [{}]}
        for (name, ty), v in zip(params, args):
            frame["vars"][0][name] = coerce(v, ty)
            frame["types"][0][name] = ty
        try:
            self.exec(body, frame)
        except ReturnEx as r:
            return coerce(r.v, method[5]) if len(method) > 5 and method[5][1] != "void" else r.v
        return None

    @staticmethod
    def _matches(ty, v):
        if ty is None:
            return True
        name = ty[1]
        if name in ("Vector3", "SdfInput", "SdfColor", "SdfIndex"):
            return isinstance(v, Vec3)
        if name in ("Vector4", "SdfOutput"):
            return isinstance(v, Vec4)
        if name in ("float", "double", "int"):
            return isinstance(v, (int, float, F32)) and not isinstance(v, bool)
        return True

    def pick(self, cname, mname, args, ext=False):
        """the overload of cname.mname for these arguments (arity, then argument kinds); None if the class has none"""
        c = self.classes.get(cname)
        if c is None:
            return None
        cands = [m for m in c["methods"].get("$all:" + mname, []) if len(m[2]) == len(args) and (("$ext" in m[4]) or not ext)]
        if len(cands) > 1:
            cands = [m for m in cands if all(self._matches(ty, v) for (_, ty), v in zip(m[2], args))] or cands
        return cands[0] if cands else None

    def call_static(self, cname, mname, args):
        m = self.pick(cname, mname, args)
        if m is None:
            raise AttributeError(f"{cname}.{mname}/{len(args)}")
        return self.invoke(None, cname, m, args)

    def call_extension(self, recv, mname, args):
        for cname in self.classes:
            m = self.pick(cname, mname, [recv] + args, ext=True)
            if m is not None and "$ext" in m[4]:
                return self.invoke(None, cname, m, [recv] + args)
        raise AttributeError(f"no extension method {mname} for {type(recv).__name__}")

    # -- statements
    def exec(self, s, fr):
        k = s[0]
        if k == "block":
            fr["vars"].append({})
            fr["types"].append({})
            try:
                for q in s[1]:
                    self.exec(q, fr)
            finally:
                fr["vars"].pop()
                fr["types"].pop()
        elif k == "expr":
            self.eval(s[1], fr)
        elif k == "decl":
            ty = None if s[1][1] == "var" else s[1]
            for name, init in s[2]:
                v = self.eval(init, fr) if init is not None else (default_of(ty) if ty else None)
                fr["vars"][-1][name] = coerce(v, ty)
                fr["types"][-1][name] = ty
        elif k == "decl_tuple":
            vals = self.eval(s[2], fr)
            for name, v in zip(s[1], vals):
                fr["vars"][-1][name] = v
                fr["types"][-1][name] = None
        elif k == "tuple_assign":
            vals = [self.eval(e, fr) for e in s[2]]
            for lv, v in zip(s[1], vals):
                self.store(lv, v, fr)
        elif k == "if":
            if self.truth(self.eval(s[1], fr)):
                self.exec(s[2], fr)
            elif s[3] is not None:
                self.exec(s[3], fr)
        elif k == "while":
            while self.truth(self.eval(s[1], fr)):
                self.exec(s[2], fr)
        elif k == "for":
            fr["vars"].append({})
            fr["types"].append({})
            try:
                if s[1] is not None:
                    self.exec(s[1], fr)
                while s[2] is None or self.truth(self.eval(s[2], fr)):
                    self.exec(s[4], fr)
                    if s[3] is not None:
                        self.eval(s[3], fr)
            finally:
                fr["vars"].pop()
                fr["types"].pop()
        elif k == "return":
            raise ReturnEx(self.eval(s[1], fr) if s[1] is not None else None)
        else:
            raise NotImplementedError(k)

    @staticmethod
    def truth(v):
        if not isinstance(v, bool):
            raise TypeError("condition is not a bool")
        return v

    # -- names
    def lookup(self, name, fr):
        for scope in reversed(fr["vars"]):
            if name in scope:
                return scope[name]
        this = fr["this"]
        if this is not None:
            if name in this.f:
                return this.f[name]
            if name in this.cls["props"]:
                return self.get_prop(this, name)
        if name in self.statics.get(fr["cname"], {}):
            return self.statics[fr["cname"]][name]
        if name in self.classes:
            return ("class", name)
        if name in self.hosts:
            return self.hosts[name]
        raise NameError(name)

    def get_prop(self, obj, name):
        frame = {"this": obj, "cname": obj.cname, "vars": [{}], "types": [{}]}
        try:
            self.exec(obj.cls["props"][name], frame)
        except ReturnEx as r:
            return r.v
        return None

    def store(self, lv, v, fr):
        k = lv[0]
        if k == "paren":
            return self.store(lv[1], v, fr)
        if k == "name":
            name = lv[1]
            for scope, types in zip(reversed(fr["vars"]), reversed(fr["types"])):
                if name in scope:
                    scope[name] = coerce(v, types.get(name))
                    return
            this = fr["this"]
            if this is not None and name in this.f:
                this.f[name] = coerce(v, this.ftype.get(name))
                return
            raise NameError(name)
        if k == "member":
            obj = self.eval(lv[1], fr)
            if isinstance(obj, Instance):
                obj.f[lv[2]] = coerce(v, obj.ftype.get(lv[2]))
                return
            if isinstance(obj, (Vec3, Vec4)) and lv[2] in ("X", "Y", "Z", "W"):
                if is_f64(v):
                    raise TypeError("double into a float component")
                setattr(obj, lv[2], F32(v))   # (obj is the variable's / the array element's own struct: mutated in place)
                return
            setattr(obj, lv[2], v)            # (a host object's property)
            return
        if k == "index":
            arr = self.eval(lv[1], fr)
            idx = [self.eval(e, fr) for e in lv[2]]
            if len(idx) != 1:
                if arr.dtype == np.float32:
                    if is_f64(v):
                        raise TypeError("double into a float[,,]")
                    arr[tuple(idx)] = F32(v)
                else:
                    arr[tuple(idx)] = value_copy(v)
                return
            old = arr[idx[0]]
            v = value_copy(v)
            if is_f64(old) and not isinstance(old, bool):
                v = float(v)
            elif isinstance(old, int) and not isinstance(old, bool):
                v = int(v)
            arr[idx[0]] = v
            return
        raise TypeError(f"not assignable: {k}")

    def eval_args(self, arg_exprs, fr):
        """argument values; `out` arguments become OutRef objects, bound to their variables by finish_outs()"""
        vals, outs = [], []
        for a in arg_exprs:
            if a[0] == "outarg":
                r = OutRef()
                outs.append((a, r))
                vals.append(r)
            else:
                vals.append(self.eval(a, fr))
        return vals, outs

    def finish_outs(self, outs, fr):
        for (_, name, declare), r in outs:
            if declare:
                fr["vars"][-1][name] = value_copy(r.value)
                fr["types"][-1][name] = None
            else:
                self.store(("name", name), r.value, fr)

    # -- expressions
    def eval(self, e, fr):
        k = e[0]
        if k == "lit":
            return e[1]
        if k == "paren":
            return self.eval(e[1], fr)
        if k == "name":
            return self.lookup(e[1], fr)
        if k == "this":
            return fr["this"]
        if k == "bin":
            op = e[1]
            if op == "&&":
                return self.truth(self.eval(e[2], fr)) and self.truth(self.eval(e[3], fr))
            if op == "||":
                return self.truth(self.eval(e[2], fr)) or self.truth(self.eval(e[3], fr))
            a, b = self.eval(e[2], fr), self.eval(e[3], fr)
            if op in ("<", ">", "<=", ">=", "==", "!="):
                return compare(op, a, b)
            return arith(op, a, b)
        if k == "neg":
            v = self.eval(e[1], fr)
            if isinstance(v, Vec3):
                return Vec3(-v.X, -v.Y, -v.Z)
            if isinstance(v, Opaque):
                return v
            return -v
        if k == "not":
            return not self.truth(self.eval(e[1], fr))
        if k == "cast":
            v = self.eval(e[2], fr)
            if e[1] == "float":
                with np.errstate(all="ignore"):
                    return F32(v)
            if e[1] == "double":
                return float(v)
            return int(v)
        if k == "assign":
            op, lv = e[1], e[2]
            v = self.eval(e[3], fr)
            if op != "=":
                v = arith(op[0], self.eval(lv, fr), v)
            self.store(lv, v, fr)
            return v
        if k in ("postinc", "preinc"):
            lv = e[2]
            old = self.eval(lv, fr)
            new = arith("+" if e[1] == "++" else "-", old, 1)
            self.store(lv, new, fr)
            return old if k == "postinc" else new
        if k == "member":
            obj = self.eval(e[1], fr)
            name = e[2]
            if isinstance(obj, Instance):
                if name in obj.f:
                    return obj.f[name]
                if name in obj.cls["props"]:
                    return self.get_prop(obj, name)
                raise AttributeError(f"{obj.cname}.{name}")
            if isinstance(obj, tuple) and obj and obj[0] == "class":
                if name in self.statics[obj[1]]:
                    return self.statics[obj[1]][name]
                return ("method", obj[1], name)
            if isinstance(obj, (list, CsList)) and name in ("Count", "Length"):
                return len(obj)
            if isinstance(obj, np.ndarray) and name == "Length":
                return int(obj.size)
            if isinstance(obj, np.ndarray) and name == "GetLength":
                return lambda k: int(obj.shape[k])
            if isinstance(obj, (Vec3, Vec4)) and name in ("X", "Y", "Z", "W"):
                return getattr(obj, name)
            return getattr(obj, name)
        if k == "index":
            arr = self.eval(e[1], fr)
            idx = [self.eval(q, fr) for q in e[2]]
            v = arr[idx[0]] if len(idx) == 1 else arr[tuple(idx)]
            return int(v) if isinstance(v, np.integer) else v
        if k == "call":
            f = e[1]
            if f[0] == "member":   # obj.Method(args): an interpreted instance, a class (static), a List<T>, a host object
                obj = self.eval(f[1], fr)
                name = f[2]
                args, outs = self.eval_args(e[2], fr)
                if outs:
                    r = getattr(obj, name)(*args)
                    self.finish_outs(outs, fr)
                    return r
                if isinstance(obj, Instance):
                    m = self.pick(obj.cname, name, args)
                    if m is not None:
                        return self.invoke(obj, obj.cname, m, args)
                    return self.call_extension(obj, name, args)
                if isinstance(obj, tuple) and obj and obj[0] == "class":
                    return self.call_static(obj[1], name, args)
                if isinstance(obj, Closure):
                    return self.call_extension(obj, name, args)
                if isinstance(obj, np.ndarray) and name == "GetLength":
                    return int(obj.shape[args[0]])
                if isinstance(obj, list) and not isinstance(obj, CsList) and name == "AsMemory":
                    return MemoryHost(obj, 0, len(obj))
                if isinstance(obj, CsList):
                    if name == "Add":
                        obj.append(args[0])
                        return None
                    if name == "ToArray":
                        return list(obj)
                if not isinstance(obj, Opaque) and not hasattr(obj, name):   # an extension method on a value / host type
                    return self.call_extension(obj, name, args)
                return getattr(obj, name)(*args)
            args = [self.eval(a, fr) for a in e[2]]
            if f[0] == "name":
                local = any(f[1] in scope for scope in fr["vars"])   # (a delegate held in a variable shadows a method name)
                for cname in ([] if local else [fr["cname"]] + self.static_imports):
                    m = self.pick(cname, f[1], args)
                    if m is not None:
                        return self.invoke(None if "static" in m[4] else fr["this"], cname, m, args)
                target = self.lookup(f[1], fr)
            else:
                target = self.eval(f, fr)
            if callable(target):
                return target(*args)
            raise TypeError(f"not callable: {target!r}")
        if k == "nullcall":
            obj = self.eval(e[1], fr)
            if obj is None:
                return None
            return getattr(obj, e[2])(*[self.eval(a, fr) for a in e[3]])
        if k == "new":
            args = [self.eval(a, fr) for a in e[2]]
            if e[1] in ("Vector3", "Vector4"):
                if any(is_f64(a) for a in args):
                    raise TypeError(f"new {e[1]} with a double argument")
                if e[1] == "Vector4":
                    return Vec4(*args)
                return Vec3(*(args * 3 if len(args) == 1 else args))
            if e[1] in self.classes:
                return self.new(e[1], args)
            return self.hosts[e[1]](*args)
        if k == "arraylit":
            return [self.eval(q, fr) for q in e[1]]
        if k == "lambda":
            return Closure(self, e[1], e[2], fr)
        if k == "cond":
            return self.eval(e[2], fr) if self.truth(self.eval(e[1], fr)) else self.eval(e[3], fr)
        if k == "newinit":
            if e[1] in self.classes:
                o = self.new(e[1], [])
                for name, ex in e[2]:
                    o.f[name] = self.eval(ex, fr)
                return o
            o = self.hosts[e[1]]()
            for name, ex in e[2]:
                setattr(o, name, self.eval(ex, fr))
            return o
        if k == "newarray_nd":
            dims = [self.eval(q, fr) for q in e[2]]
            if e[1] == "float":
                return np.zeros(dims, dtype=np.float32)
            a = np.empty(dims, dtype=object)
            for ix in np.ndindex(*dims):
                a[ix] = Vec3() if e[1] == "Vector3" else None
            return a
        if k == "newarray":
            n = self.eval(e[2], fr)
            d = {"int": 0, "double": 0.0, "float": F32(0)}
            if e[1] == "Vector3":
                return [Vec3() for _ in range(n)]
            if e[1] == "Vector4":
                return [Vec4(0, 0, 0, 0) for _ in range(n)]
            return [d[e[1]]] * n
        if k == "newlist":
            return CsList()
        if k == "tuple":
            return tuple(self.eval(q, fr) for q in e[1])
        raise NotImplementedError(k)


def parse(text):
    return Parser(text).parse_file()
