"""numpy float32 interpreter of the SDF program IR (include/sdfkit_hip.h: sdfk_op) -- TEST
INFRASTRUCTURE (a second oracle, for the JIT code generator): every op is one IEEE binary32
operation on arrays, exactly what the generated HIP code must compute per voxel.  Sample points
follow Voxels.SampleSdf (Voxels.cs:81,104-106): p = (min + 0.5*D) + (float)i * D."""
import numpy as np

f32 = np.float32
(CONST, X, Y, Z, ADD, SUB, MUL, DIV, NEG, ABS, SQRT, FLOOR, MIN_SEL, MAX_SEL, MIN_IEEE, MAX_IEEE, SEL_LT) = range(17)


def _max_ieee(a, b):   # Math.Max: NaN if either is NaN, -0 < +0
    with np.errstate(all="ignore"):
        return np.where(a != b, np.where(np.isnan(a), a, np.where(b < a, a, b)), np.where(np.signbit(b), a, b)).astype(f32)


def _min_ieee(a, b):
    with np.errstate(all="ignore"):
        return np.where(a != b, np.where(np.isnan(a), a, np.where(a < b, a, b)), np.where(np.signbit(a), a, b)).astype(f32)


def _eval(ops, px, py, pz):
    """every value of the program at the points (px, py, pz) (float32 arrays of one shape)"""
    v = []
    with np.errstate(all="ignore"):
        for (op, a, b, c, dd, imm) in ops:
            if op == CONST: r = np.full(px.shape, f32(imm), f32)
            elif op == X: r = px
            elif op == Y: r = py
            elif op == Z: r = pz
            elif op == ADD: r = v[a] + v[b]
            elif op == SUB: r = v[a] - v[b]
            elif op == MUL: r = v[a] * v[b]
            elif op == DIV: r = v[a] / v[b]
            elif op == NEG: r = -v[a]
            elif op == ABS: r = np.abs(v[a])
            elif op == SQRT: r = np.sqrt(v[a])
            elif op == FLOOR: r = np.floor(v[a])
            elif op == MIN_SEL: r = np.where(v[a] < v[b], v[a], v[b])
            elif op == MAX_SEL: r = np.where(v[a] > v[b], v[a], v[b])
            elif op == MIN_IEEE: r = _min_ieee(v[a], v[b])
            elif op == MAX_IEEE: r = _max_ieee(v[a], v[b])
            elif op == SEL_LT: r = np.where(v[a] < v[b], v[c], v[dd])
            else: raise ValueError(op)
            v.append(np.asarray(r, f32))
    return v


def sample(ops, out_rgbw, writes_color, mn, mx, nx, ny, nz):
    """ops: list of (opcode, a, b, c, d, imm).  Returns (values[nx,ny,nz], colors[nx,ny,nz,3])."""
    mn, mx = np.asarray(mn, f32), np.asarray(mx, f32)
    d = (mx - mn) / np.array([nx, ny, nz], f32)
    m = mn + f32(0.5) * d
    px = (m[0] + np.arange(nx, dtype=f32) * d[0])[:, None, None] + np.zeros((nx, ny, nz), f32)
    py = (m[1] + np.arange(ny, dtype=f32) * d[1])[None, :, None] + np.zeros((nx, ny, nz), f32)
    pz = (m[2] + np.arange(nz, dtype=f32) * d[2])[None, None, :] + np.zeros((nx, ny, nz), f32)
    v = _eval(ops, px, py, pz)
    values = v[out_rgbw[3]]
    colors = np.stack([v[out_rgbw[k]] for k in range(3)], axis=-1) if writes_color else np.zeros((nx, ny, nz, 3), f32)
    return values, colors


def run(ops, out_rgbw, points):
    """The program at arbitrary points [n, 3] (float32): (r, g, b, w) arrays, None for an output id < 0."""
    pts = np.asarray(points, f32)
    v = _eval(ops, np.ascontiguousarray(pts[:, 0]), np.ascontiguousarray(pts[:, 1]), np.ascontiguousarray(pts[:, 2]))
    return [v[k] if k >= 0 else None for k in out_rgbw]


def random_program(seed, n_ops=48):
    """A random DAG over the 17 opcodes (constants include 0, -0, huge and tiny values, so NaN, inf,
    signed zeros and denormals all flow through)."""
    rng = np.random.default_rng(seed)
    ops = [(X, -1, -1, -1, -1, 0.0), (Y, -1, -1, -1, -1, 0.0), (Z, -1, -1, -1, -1, 0.0)]
    consts = [0.0, -0.0, 1.0, -1.5, 0.25, 3.0, 1e-30, 1e30, 1e-42, 0.1]
    for c in rng.choice(consts, 4, replace=False):
        ops.append((CONST, -1, -1, -1, -1, float(f32(c))))
    while len(ops) < n_ops:
        op = int(rng.choice([ADD, SUB, MUL, MUL, ADD, DIV, NEG, ABS, SQRT, FLOOR, MIN_SEL, MAX_SEL, MIN_IEEE, MAX_IEEE, SEL_LT]))
        k = len(ops)
        a, b, c, d = (int(rng.integers(max(0, k - 12), k)) for _ in range(4))
        if op in (NEG, ABS, SQRT, FLOOR):
            ops.append((op, a, -1, -1, -1, 0.0))
        elif op == SEL_LT:
            ops.append((op, a, b, c, d, 0.0))
        else:
            ops.append((op, a, b, -1, -1, 0.0))
    k = len(ops)
    out = [int(rng.integers(k - 10, k)) for _ in range(4)]
    return ops, out
