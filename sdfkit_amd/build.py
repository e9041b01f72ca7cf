"""Builds libsdfkit_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

The library is seven translation units (csrc/lib_*.hip + csrc/mc_kernels.hip) compiled in parallel -- objects under csrc/_obj/, rebuilt
only when a file they include is newer -- and linked with the version script that exports exactly include/sdfkit_hip.h's symbols."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libsdfkit_hip.so")
# translation units: context / streams / options / copies; JIT + code cache; volumes + sampling; the marching-cubes driver + graphs;
# mesh accessors; the sharded step; the marching-cubes kernels
SOURCES = ["lib_context.hip", "lib_jit.hip", "lib_volume.hip", "lib_march.hip", "lib_mesh.hip", "lib_dist.hip", "mc_kernels.hip"]
HEADERS = ["lib_internal.h", "mc_kernels.h", "mc_device.h", "mc_decide.h", "mc_params.h", "mc_luts.h", "sample_codegen.h", "dist_rccl.h", "node_local.h",
           "slab_protocol.h", os.path.join("..", "..", "include", "sdfkit_hip.h")]
DEPS = ["exports.map"] + SOURCES + HEADERS
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fno-fast-math", "-fvisibility=hidden", "-Wall",
          "-Wno-unused-function"]
EXTRA = os.environ.get("SDFKIT_HIP_CFLAGS", "").split()   # (experiment builds: tools/variants.sh)


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def _includes(path, seen):
    """The csrc / include files `path` includes, transitively (quoted includes only)."""
    if path in seen or not os.path.exists(path):
        return
    seen.add(path)
    with open(path) as f:
        for m in re.finditer(r'^\s*#include\s+"([^"]+)"', f.read(), re.M):
            _includes(os.path.normpath(os.path.join(os.path.dirname(path), m.group(1))), seen)


def _object_for(src, hipcc, verbose, out_dir=None, extra=()):
    out_dir = out_dir or OBJ
    os.makedirs(out_dir, exist_ok=True)
    s = os.path.join(CSRC, src)
    o = os.path.join(out_dir, src.replace(".hip", ".o"))
    deps = set()
    _includes(s, deps)
    if not extra and os.path.exists(o) and all(os.path.getmtime(d) <= os.path.getmtime(o) for d in deps) and \
            os.path.getmtime(os.path.abspath(__file__)) <= os.path.getmtime(o):
        return o
    cmd = [hipcc] + CFLAGS + list(extra) + ["-c", s, "-o", o]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return o


def build(force=False, verbose=False, out=None, extra=()):
    """out / extra: an experiment build of the same sources with extra -D flags into another file (tools/variants.sh)."""
    target = out or LIB
    extra = list(extra) + EXTRA
    if not out and not force and not needs_build():
        return LIB
    import fcntl
    # one builder at a time (the ranks of a multi-process launch all come through here), and the
    # library appears atomically: nobody ever loads a half-written file
    with open(target + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not out and not force and not needs_build():   # another process built it while we waited
            return LIB
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        obj_dir = OBJ if not extra else os.path.join(OBJ, "variant_" + re.sub(r"\W+", "_", os.path.basename(target)))
        with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
            objs = list(ex.map(lambda s: _object_for(s, hipcc, verbose, obj_dir, extra), SOURCES))
        tmp = f"{target}.tmp.{os.getpid()}"
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", tmp] + objs + \
              ["-lhiprtc", "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, target)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return target


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    if len(args) >= 1:      # python -m sdfkit_amd.build OUT.so -DFLAG ...
        build(force=True, verbose=True, out=os.path.abspath(args[0]), extra=args[1:])
    else:
        build(force="--force" in sys.argv, verbose=True)
