"""ctypes binding of libxfeat_hip.so (include/xfeat_hip.h).

The shared library is the product; this module only declares its C ABI for the Python
host-side mirror, the tests and bench.py.  It never falls back to a CPU implementation:
if the library is missing it raises, and every compute call fails loudly when no HIP
device is present (XFH_ERR_NO_DEVICE).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
# XFEAT_HIP_LIB: another build of the same ABI (the harness only: tests/workers/knob_worker.py and tools/ load the debug build with the test knobs,
# libxfeat_hip_knobs.so, this way -- the library itself reads no such variable)
LIB_PATH = os.environ.get("XFEAT_HIP_LIB") or os.path.join(_DIR, "libxfeat_hip.so")
KNOBS_LIB_PATH = os.path.join(_DIR, "libxfeat_hip_knobs.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

OK = 0
STATUS = {0: "OK", 1: "INVALID_ARG", 2: "EMPTY_IMAGE", 3: "BAD_SIZE", 4: "NO_WEIGHTS", 5: "BAD_WEIGHTS",
          6: "HIP", 7: "NO_DEVICE", 8: "OUT_OF_MEMORY", 9: "BATCH_TOO_LARGE", 10: "IO", 11: "COMM"}
ERR_EMPTY_IMAGE = 2
ERR_NO_DEVICE = 7
ERR_COMM = 11

K = dict(NONE=0, MNN_GEMM=1, CONV_MFMA=2, CONV_DIRECT=3, NMS=4, SELECT=5, DESC=6, HEADS=7, DIST_I32=8, PREPROC=9, BEST2=10, DISTINCTIVE=11, MNN_GEMM_SEG=12)
T = dict(X=0, XSTAT=1, SKIP_POOL=2, FEATS=6, H1=8, K1H=9, RAW0=16, STAT0=48, SEL=80)


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_height", C.c_int32), ("max_width", C.c_int32),
                ("nfeatures", C.c_int32), ("max_batch", C.c_int32), ("bn_mode", C.c_int32),
                ("nms_threshold", C.c_float), ("flags", C.c_int32), ("reserved", C.c_int32 * 7)]


FLAG_RESCALE_KEYPOINTS = 1
FLAG_SERIAL_BRANCH = 2


# every symbol include/xfeat_hip.h declares: (name, restype, argtypes)
_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
_pi = C.POINTER(C.c_int)
SYMBOLS = [
    ("xfh_config_default", None, [C.POINTER(Config)]),
    ("xfh_create", _i, [C.POINTER(Config), C.POINTER(_vp)]),
    ("xfh_destroy", _i, [_vp]),
    ("xfh_load_weights", _i, [_vp, _vp, _sz]),
    ("xfh_load_weights_file", _i, [_vp, C.c_char_p]),
    ("xfh_extract", _i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _pi, _pi]),
    ("xfh_extract_submit", _i, [_vp, _vp, _i, _i, _i, _i, _i]),
    ("xfh_extract_collect", _i, [_vp, _vp, _vp, _pi, _pi]),
    ("xfh_detect_and_compute", _i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _pi, _pi]),
    ("xfh_record_bytes", _sz, [_i]),
    ("xfh_record_kps_offset", _sz, []),
    ("xfh_record_desc_offset", _sz, [_i]),
    ("xfh_extract_batch", _i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    ("xfh_extract_batch_submit", _i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    ("xfh_extract_batch_wait", _i, [_vp]),
    ("xfh_extract_batch_drain", _i, [_vp]),
    ("xfh_pipeline_lanes", _i, [_vp, _i]),
    ("xfh_host_alloc", _i, [C.POINTER(_vp), _sz]),
    ("xfh_host_free", _i, [_vp]),
    ("xfh_host_register", _i, [_vp, _sz]),
    ("xfh_host_unregister", _i, [_vp]),
    ("xfh_extract_batch_device", _i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    ("xfh_extract_batch_device_images", _i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    ("xfh_match_mnn", _i, [_vp, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _pi]),
    ("xfh_match_mnn_device", _i, [_vp, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    ("xfh_match_image_bytes", _sz, [_i]),
    ("xfh_match_prepare_device", _i, [_vp, _vp, _i, _vp]),
    ("xfh_match_mnn_prepared_device", _i, [_vp, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    ("xfh_match_mnn_prepared_batch_device", _i, [_vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    ("xfh_match_records_device", _i, [_vp, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp]),
    ("xfh_descriptor_distance", _i, [_vp, _vp]),
    ("xfh_distance_i32", _i, [_vp, _vp, _i, _vp, _i, _vp]),
    ("xfh_distance_i32_device", _i, [_vp, _vp, _i, _vp, _i, _vp]),
    ("xfh_best2_csr", _i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    ("xfh_best2_csr_device", _i, [_vp, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    ("xfh_distinctive_csr", _i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp]),
    ("xfh_distinctive_csr_device", _i, [_vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp]),
    ("xfh_comm_unique_id", _i, [_vp]),
    ("xfh_comm_library", C.c_char_p, []),
    ("xfh_comm_create", _i, [_vp, _vp, _i, _i]),
    ("xfh_comm_destroy", _i, [_vp]),
    ("xfh_comm_rank", _i, [_vp]),
    ("xfh_comm_world", _i, [_vp]),
    ("xfh_allgather_records", _i, [_vp, _vp, _i, _vp, _i]),
    ("xfh_gather_records_root", _i, [_vp, _vp, _i, _vp, _i, _i]),
    ("xfh_compact_bytes_max", _sz, [_i, _i]),
    ("xfh_gather_compact_root", _i, [_vp, _vp, _i, _vp, C.POINTER(_sz), _i, _i]),
    ("xfh_unpack_compact", _i, [_vp, _sz, _i, _i, _vp, _vp, _pi, _pi]),
    ("xfh_allgather_bytes", _i, [_vp, _vp, _sz, _vp, _i]),
    ("xfh_comm_fence", _i, [_vp, _i]),
    ("xfh_comm_wait_ctx", _i, [_vp, _vp]),
    ("xfh_comm_fence_ctx", _i, [_vp, _vp, _i]),
    ("xfh_comm_synchronize", _i, [_vp]),
    ("xfh_synchronize", _i, [_vp]),
    ("xfh_set_stream", _i, [_vp, _vp]),
    ("xfh_strerror", C.c_char_p, [_i]),
    ("xfh_last_hip_error", C.c_char_p, [_vp]),
    ("xfh_version", C.c_char_p, []),
    ("xfh_device_count", _i, []),
    ("xfh_dev_alloc", _i, [C.POINTER(_vp), _sz]),
    ("xfh_dev_free", _i, [_vp]),
    ("xfh_memcpy_h2d", _i, [_vp, _vp, _sz]),
    ("xfh_memcpy_d2h", _i, [_vp, _vp, _sz]),
    ("xfh_timing_enable", _i, [_vp, _i, C.c_uint]),
    ("xfh_timing_read", _i, [_vp, _pi, C.POINTER(C.c_double)]),
    ("xfh_bench_mnn_gemm", _i, [_vp, _vp, _i, _vp, _i, _i, C.POINTER(C.c_double)]),
    ("xfh_debug_match_plan", _i, [_i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("xfh_bench_sclk", _i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("xfh_bench_mnn_gemm_batch", _i, [_vp, _i, _vp, _vp, _vp, _vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("xfh_bench_match_batch", _i, [_vp, _i, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp, _vp, _vp, _i, C.POINTER(C.c_double)]),
    ("xfh_bench_match_prepared", _i, [_vp, _vp, _i, _vp, _i, C.c_float, _vp, _vp, _vp, _vp, _i, C.POINTER(C.c_double)]),
    ("xfh_bench_match_raw", _i, [_vp, _vp, _i, _vp, _i, C.c_float, _vp, _vp, _vp, _vp, _i, C.POINTER(C.c_double)]),
    ("xfh_bench_calib", _i, [_vp, _i, _sz, _i]),
    ("xfh_kernel_name", C.c_char_p, [_i]),
    ("xfh_debug_tensor", _i, [_vp, _i, _i, _vp, _sz, C.POINTER(_sz)]),
    ("xfh_debug_select", _i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
]

_lib = None


def lib():
    """Load libxfeat_hip.so (raises if it has not been built: there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950); this package has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)          # AttributeError if the ABI symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class XfhError(RuntimeError):
    def __init__(self, status: int, detail: str = ""):
        self.status = status
        msg = lib().xfh_strerror(status).decode()
        super().__init__(f"xfeat_hip status {status} ({STATUS.get(status, '?')}: {msg}) {detail}")


def check(status: int, ctx=None):
    if status != OK:
        detail = ""
        if ctx is not None and status in (6, 11):
            detail = lib().xfh_last_hip_error(ctx).decode()
        raise XfhError(status, detail)


class HostBuffer:
    """pinned host allocation through the C ABI (xfh_host_alloc), viewed as a numpy array"""

    def __init__(self, nbytes: int):
        p = C.c_void_p()
        check(lib().xfh_host_alloc(C.byref(p), nbytes))
        self.ptr = p.value
        self.nbytes = nbytes
        self.array = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self.ptr))

    def free(self):
        if self.ptr:
            self.array = None
            lib().xfh_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """HBM allocation owned through the C ABI (no torch needed)."""

    def __init__(self, nbytes: int):
        p = C.c_void_p()
        check(lib().xfh_dev_alloc(C.byref(p), nbytes))
        self.ptr = p.value
        self.nbytes = nbytes

    def upload(self, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        check(lib().xfh_memcpy_h2d(self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, dtype, count: int, offset: int = 0) -> np.ndarray:
        out = np.empty(count, dtype)
        check(lib().xfh_memcpy_d2h(out.ctypes.data, self.ptr + offset, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().xfh_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
