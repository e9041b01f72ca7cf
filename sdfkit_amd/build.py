"""Builds libsdfkit_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsdfkit_hip.so")
SOURCES = ["sdfkit_hip.hip"]
DEPS = ["exports.map", "sdfkit_hip.hip", "mc_kernels.hip", "mc_device.h", "mc_decide.h", "mc_params.h", "mc_luts.h", "sample_codegen.h", "dist_rccl.h", "node_local.h", "slab_protocol.h",
        os.path.join("..", "..", "include", "sdfkit_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fno-fast-math", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    import fcntl
    # one builder at a time (the ranks of a multi-process launch all come through here), and the
    # library appears atomically: nobody ever loads a half-written file
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build():   # another process built it while we waited
            return LIB
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        tmp = f"{LIB}.tmp.{os.getpid()}"
        cmd = ([hipcc] + FLAGS + ["-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", tmp] +
               [os.path.join(CSRC, s) for s in SOURCES] + ["-lhiprtc", "-ldl"])
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        try:
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
