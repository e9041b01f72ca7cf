"""ctypes binding of libsdfkit_hip.so (include/sdfkit_hip.h).

There is no fallback of any kind: if the shared library is missing it is built with
hipcc; if that fails, or if no gfx950 device is present when a compute entry point is
used, an exception is raised.
"""
import ctypes as C
import os
import sys

from . import build as _build

# Streams that wait for events must not share hardware queues with the library's lanes (sdfk_init, csrc/lib_context.hip, says
# why): the HIP runtime reads this when it initialises -- with torch in the process that is the first CUDA call, not the
# import -- so it is set as early as this module is imported, unless the user chose a value.  The library itself never
# edits the environment (a host binding's job: this module, shim/SdfKit.Hip/Native.cs, include/SdfKit.hpp);
# sdfk_get_option(SDFK_OPT_HW_QUEUES) reports what the process had when the library came up.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None
_inited_device = None


class SdfKitNativeError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"sdfkit_hip status {status}: {message}")
        self.status = status


class Op(C.Structure):
    """struct sdfk_op"""
    _fields_ = [("opcode", C.c_int32), ("a", C.c_int32), ("b", C.c_int32), ("c", C.c_int32),
                ("d", C.c_int32), ("imm", C.c_float)]


# every symbol declared in include/sdfkit_hip.h: name -> (restype, argtypes)
_vp, _i32, _i64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_float
_fp, _vpp = C.POINTER(C.c_float), C.POINTER(C.c_void_p)
SIGNATURES = {
    "sdfk_abi_version": (C.c_int, []),
    "sdfk_init": (C.c_int, [C.c_int]),
    "sdfk_shutdown": (None, []),
    "sdfk_set_stream": (C.c_int, [_vp]),
    "sdfk_synchronize": (C.c_int, []),
    "sdfk_last_error": (C.c_char_p, []),
    "sdfk_program_create": (C.c_int, [C.POINTER(Op), _i32, C.POINTER(_i32), _i32, _vpp]),
    "sdfk_program_check": (C.c_int, [C.POINTER(Op), _i32, C.POINTER(_i32), _i32]),
    "sdfk_program_source": (C.c_char_p, [_vp]),
    "sdfk_program_destroy": (None, [_vp]),
    "sdfk_jit_stats": (C.c_int, [C.POINTER(_i64), C.POINTER(_i64), C.POINTER(C.c_double)]),
    "sdfk_graph_stats": (C.c_int, [C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "sdfk_volume_create": (C.c_int, [_i32, _i32, _i32, _fp, _fp, _i32, _vpp]),
    "sdfk_volume_create_slab": (C.c_int, [_i32, _i32, _i32, _fp, _fp, _i32, _i32, _i32, _vpp]),
    "sdfk_volume_upload": (C.c_int, [_vp, _vp, _vp]),
    "sdfk_volume_download": (C.c_int, [_vp, _vp, _vp]),
    "sdfk_volume_device_ptrs": (C.c_int, [_vp, _vpp, _vpp]),
    "sdfk_volume_row_pitch": (C.c_int, [_vp, C.POINTER(_i32)]),
    "sdfk_volume_free": (None, [_vp]),
    "sdfk_sample": (C.c_int, [_vp, _vp, _i32]),
    "sdfk_volume_clip_to_bounds": (C.c_int, [_vp]),
    "sdfk_march": (C.c_int, [_vp, _f, _i32, _vpp]),
    "sdfk_march_host": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _fp, _fp, _f, _i32, _vpp]),
    "sdfk_sample_march": (C.c_int, [_vp, _fp, _fp, _i32, _i32, _i32, _i32, _f, _i32, _vpp]),
    "sdfk_march_begin": (C.c_int, [_vp, _f, _i32, _i32, _vpp, C.POINTER(_i64), C.POINTER(_i64)]),
    "sdfk_march_finish": (C.c_int, [_vp, _i64, _vpp]),
    "sdfk_march_job_free": (None, [_vp]),
    "sdfk_march_slab": (C.c_int, [_vp, _f, _i32, _i32, _i64, _vpp]),
    "sdfk_sample_march_slab": (C.c_int, [_vp, _vp, _i32, _f, _i32, _i32, _i64, _vpp]),
    "sdfk_mesh_pack": (C.c_int, [_vp, _vp, _i64, C.POINTER(_i64)]),
    "sdfk_slab_enqueue": (C.c_int, [_vp, _vp, _i32, _f, _i32, _i32, _vp, _i64, _i32, _vp]),
    "sdfk_slabs_rebase": (C.c_int, [_vp, _i32, _i64]),
    "sdfk_slabs_rebase_mirror": (C.c_int, [_vp, _i32, _i64, _vp]),
    "sdfk_host_alloc": (C.c_int, [_i64, _vpp]),
    "sdfk_host_free": (None, [_vp]),
    "sdfk_host_prefault": (C.c_int, [_vp, _i64]),
    "sdfk_copy_stats": (C.c_int, [C.POINTER(_i64)]),
    "sdfk_stream_placement": (C.c_int, [C.POINTER(_i32)]),
    "sdfk_mesh_size_hint": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32)]),
    "sdfk_mesh_counts": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "sdfk_mesh_bounds": (C.c_int, [_vp, _fp, _fp]),
    "sdfk_mesh_copy": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "sdfk_mesh_device_ptrs": (C.c_int, [_vp, _vpp, _vpp, _vpp, _vpp]),
    "sdfk_mesh_copy_device": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "sdfk_mesh_transform": (C.c_int, [_vp, _fp, _fp]),
    "sdfk_mesh_stats": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "sdfk_mesh_free": (None, [_vp]),
    "sdfk_profile_enable": (C.c_int, [_i32]),
    "sdfk_profile_reset": (C.c_int, []),
    "sdfk_lane_begin": (C.c_int, [_i32, _vp]),
    "sdfk_lane_end": (C.c_int, [_i32]),
    "sdfk_raymarch": (C.c_int, [_vp, _i32, _i32, _fp, _fp, C.c_float, C.c_float, _i32, _vp, _vp]),
    "sdfk_raymarch_device": (C.c_int, [_vp, _i32, _i32, _fp, _fp, C.c_float, C.c_float, _i32, _vp, _vp]),
    "sdfk_set_option": (C.c_int, [_i32, _i64]),
    "sdfk_get_option": (C.c_int, [_i32, C.POINTER(_i64)]),
    "sdfk_set_cache_dir": (C.c_int, [C.c_char_p]),
    "sdfk_dist_unique_id": (C.c_int, [_vp]),
    "sdfk_dist_init": (C.c_int, [_i32, _i32, _vp]),
    "sdfk_dist_init_host": (C.c_int, [_i32, _i32, _vp, _vp]),
    "sdfk_dist_info": (C.c_int, [C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "sdfk_dist_shutdown": (None, []),
    "sdfk_dist_slab": (C.c_int, [_i32, _i32, _i32, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "sdfk_dist_session_create": (C.c_int, [_vp, _fp, _fp, _i32, _i32, _i32, _i32, _f, _i32, _vpp]),
    "sdfk_dist_submit": (C.c_int, [_vp]),
    "sdfk_dist_collect": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "sdfk_dist_counts": (C.c_int, [_vp, C.POINTER(_i64)]),
    "sdfk_dist_mesh": (C.c_int, [_vp, _vpp]),
    "sdfk_dist_slab_mesh": (C.c_int, [_vp, _vpp]),
    "sdfk_dist_gathered": (C.c_int, [_vp, _vpp, C.POINTER(_i64)]),
    "sdfk_dist_stats": (C.c_int, [_vp, C.POINTER(_i64)]),
    "sdfk_dist_enqueue_only": (C.c_int, [_vp]),
    "sdfk_dist_tune": (C.c_int, [_vp, _i32, C.POINTER(_i64)]),
    "sdfk_dist_session_free": (None, [_vp]),
    "sdfk_dist_to_mesh": (C.c_int, [_vp, _fp, _fp, _i32, _i32, _i32, _i32, _f, _vpp]),
    "sdfk_eval_points": (C.c_int, [_vp, _vp, _i64, _vp]),
    "sdfk_eval_points_device": (C.c_int, [_vp, _vp, _i64, _vp]),
    "sdfk_node_open": (C.c_int, [C.POINTER(_i32), _i32, _vpp]),
    "sdfk_node_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "sdfk_node_to_mesh": (C.c_int, [_vp, _vp, _i32, C.POINTER(_i32), _i32, _fp, _fp, _i32, _i32, _i32, _i32, _f, _vpp]),
    "sdfk_node_mesh_begin": (C.c_int, [_vp, _vp, _i32, C.POINTER(_i32), _i32, _fp, _fp, _i32, _i32, _i32, _i32, _f, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32)]),
    "sdfk_node_mesh_copy": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _fp, _fp]),
    "sdfk_node_close": (None, [_vp]),
    "sdfk_profile_count": (C.c_int, []),
    "sdfk_profile_get": (C.c_int, [_i32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(_i64)]),
}


# sdfk_allgather_fn: int (*)(void* ctx, const void* send, void* recv, int64_t bytes_per_rank)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)

# enum sdfk_status
OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_COMPILE, ERR_NOMEM, ERR_UNSUPPORTED = 0, 1, 2, 3, 4, 5, 6

# enum sdfk_option
OPT_LANES, OPT_TOKENS, OPT_GRAPHS, OPT_COPY_MODE, OPT_CORNER_EVAL, OPT_VCOLOR_EVAL = 1, 2, 3, 4, 5, 6
OPT_DIST_EXCHANGE, OPT_DIST_LANES, OPT_HW_QUEUES, OPT_CODE_CACHE, OPT_PREFAULT_HUGE, OPT_DIST_INDEX16, OPT_STREAM_PLACEMENT, OPT_IDLE_LANE = 7, 8, 9, 10, 11, 12, 13, 14
OPT_IDLE_PROGRAMS = 15
OPT_ELIDE_VOLUME = 16
OPT_COLOR_PASSES = 17


def library_path():
    # SDFKIT_HIP_LIBRARY: an experiment build of the same sources (tools/variants.sh), never a different product
    return os.environ.get("SDFKIT_HIP_LIBRARY") or os.path.join(_HERE, "libsdfkit_hip.so")


def lib():
    """Load (building if needed) the shared library.  Raises if it cannot be produced."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.environ.get("SDFKIT_HIP_LIBRARY") and (not os.path.exists(path) or _build.needs_build()):
            # sources or the header are newer than the library (or it is missing): rebuild, and never
            # run a stale binary against the current ctypes table -- a failed build is an error
            try:
                _build.build()
            except Exception as e:
                what = "is missing" if not os.path.exists(path) else "is older than its sources"
                raise RuntimeError(f"libsdfkit_hip.so {what} and could not be built with hipcc: {e}") from e
        # PyTorch's ROCm wheels bundle their own HIP runtime, and the runtime a process loads
        # FIRST is the one that owns the GPU: loading this library before torch leaves torch
        # without a device ("No HIP GPUs are available").  Where torch is installed, load it
        # first so that torch tensors / streams and this library share one runtime.
        if "torch" not in sys.modules and os.environ.get("SDFKIT_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != 0:
        raise SdfKitNativeError(status, lib().sdfk_last_error().decode("utf-8", "replace"))


def init(device=None):
    """sdfk_init on `device` (default: LOCAL_RANK or 0).  Raises without a gfx950 GPU."""
    global _inited_device
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if _inited_device is None:
        check(lib().sdfk_init(device))
        _inited_device = device
    elif _inited_device != device:
        raise RuntimeError(f"sdfkit_hip already initialised on device {_inited_device}")
    return _inited_device


def bind_torch_stream(device=None):
    """Put torch and this library on ONE explicit stream and return it.

    torch's default stream has the null handle, which sdfk_set_stream reads as "the library's own
    stream" -- a non-blocking stream that nothing orders against the null stream.  Code that mixes
    torch work (collectives, copies, events) with library calls must not rely on that pair: this
    creates a stream (unless torch's current one is already a real one), makes it torch's current
    stream and hands it to the library."""
    import torch
    s = torch.cuda.current_stream(device)
    if s.cuda_stream == 0:
        s = torch.cuda.Stream(device)
        torch.cuda.set_stream(s)
    check(lib().sdfk_set_stream(C.c_void_p(s.cuda_stream)))
    return s


class _PinnedBlock:
    """A block of the library's pinned host arena, exposed to numpy through the array interface; goes
    back to the arena when the last array viewing it is collected."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        check(lib().sdfk_host_alloc(int(nbytes), C.byref(p)))
        self.ptr, self.nbytes = p.value, int(nbytes)

    @property
    def __array_interface__(self):
        return {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False), "version": 3}

    def __del__(self):
        try:
            if self.ptr and _lib is not None and _inited_device is not None:
                _lib.sdfk_host_free(C.c_void_p(self.ptr))
        except Exception:
            pass
        self.ptr = 0


def pinned_empty(shape, dtype):
    """numpy.empty in pinned host memory (sdfk_host_alloc): device-to-host copies into it are plain DMA.
    SDFK_PINNED_ARRAYS=0 falls back to numpy's own allocator (pageable memory, as a managed runtime hands out)."""
    import numpy as np
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    if n == 0 or os.environ.get("SDFK_PINNED_ARRAYS") == "0":
        return np.empty(shape, dt)
    return np.asarray(_PinnedBlock(n)).view(dt).reshape(shape)


def stream_placement():
    """sdfk_stream_placement: {'measured', 'classes', 'lanes': [class of lane 0..4], 'exchange'} (-1: no placed stream)."""
    a = (C.c_int32 * 8)()
    check(lib().sdfk_stream_placement(a))
    return {"measured": bool(a[0]), "classes": int(a[1]), "lanes": [int(a[2 + k]) for k in range(5)], "exchange": int(a[7])}


def set_option(key, value):
    check(lib().sdfk_set_option(key, int(value)))


def get_option(key):
    v = C.c_int64()
    check(lib().sdfk_get_option(key, C.byref(v)))
    return v.value


class option:
    """with N.option(N.OPT_LANES, 0): ...   -- an option changed for a block and restored after it."""

    def __init__(self, key, value):
        self.key, self.value = key, value

    def __enter__(self):
        self.old = get_option(self.key)
        set_option(self.key, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.key, self.old)
        return False


def shutdown():
    global _inited_device
    if _lib is not None:
        _lib.sdfk_shutdown()
    _inited_device = None


def f3(v):
    import numpy as np
    return (C.c_float * 3)(*[float(np.float32(x)) for x in v])


def profile_snapshot():
    """{kernel name: (total_ms, launches)} recorded since the last reset."""
    L = lib()
    out = {}
    for i in range(L.sdfk_profile_count()):
        name, ms, n = C.c_char_p(), C.c_double(), C.c_int64()
        check(L.sdfk_profile_get(i, C.byref(name), C.byref(ms), C.byref(n)))
        out[name.value.decode()] = (ms.value, n.value)
    return out
