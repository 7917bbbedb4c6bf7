"""ctypes binding of libblaze_hip.so (include/blaze_hip.h).  The library is the product; this module
only declares signatures and turns return codes into the reference's error enum (src/error.rs:6-32).
There is no CPU path: if the shared library is missing, or no HIP device is visible, calls raise."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLAZE_HIP_LIB: an A/B build of the same library (development aid; csrc/Makefile)
LIB_PATH = os.environ.get("BLAZE_HIP_LIB") or os.path.join(_HERE, "lib", "libblaze_hip.so")


class DriverClientError(Exception):
    """src/error.rs:6-32.  `variant` is the reference enum variant name."""

    VARIANTS = {
        1: "WriteError",
        2: "ReadError",
        3: "HBICAPNotReady",
        4: "InvalidPrimitiveParam",
        5: "CsvError",
        6: "LoadFailed",
        7: "FileError",
        8: "Unknown",
    }

    def __init__(self, code: int, message: str):
        self.code = code
        self.variant = self.VARIANTS.get(code, "Unknown")
        super().__init__(f"{self.variant}: {message}")


_lib = None
_aux = None

# every exported symbol of include/blaze_hip.h: name -> (restype, argtypes)
_u8p = C.c_void_p
_SIGS = {
    "blz_last_error_message": (C.c_char_p, []),
    "blz_device_count": (C.c_int, []),
    "blz_point_size": (C.c_size_t, [C.c_int]),
    "blz_result_size": (C.c_size_t, [C.c_int]),
    "blz_msm_new": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "blz_msm_free": (None, [C.c_void_p]),
    "blz_msm_loaded_binary_parameters": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "blz_msm_initialize": (C.c_int, [C.c_void_p, C.c_uint32, C.c_int, C.c_uint64, C.c_uint64]),
    "blz_msm_start_process": (C.c_int, [C.c_void_p]),
    "blz_msm_set_data": (C.c_int, [C.c_void_p, _u8p, C.c_size_t, _u8p, C.c_size_t, C.c_uint32, C.c_int, C.c_uint64, C.c_uint64]),
    "blz_msm_set_data_device": (C.c_int, [C.c_void_p, _u8p, C.c_size_t, _u8p, C.c_size_t, C.c_uint32, C.c_int, C.c_uint64, C.c_uint64]),
    "blz_msm_wait_result": (C.c_int, [C.c_void_p]),
    "blz_msm_result": (C.c_int, [C.c_void_p, _u8p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]),
    "blz_msm_load_data_to_hbm": (C.c_int, [C.c_void_p, _u8p, C.c_size_t, C.c_uint64, C.c_uint64]),
    "blz_msm_load_data_to_hbm_device": (C.c_int, [C.c_void_p, _u8p, C.c_size_t, C.c_uint64, C.c_uint64]),
    "blz_msm_get_data_from_hbm": (C.c_int, [C.c_void_p, _u8p, C.c_size_t, C.c_uint64, C.c_uint64]),
    "blz_arena_release": (C.c_int, [C.c_int]),
    "blz_arena_set_policy": (C.c_int, [C.c_int, C.c_uint32]),
    "blz_arena_export": (C.c_int, [C.c_int, C.c_char_p]),
    "blz_arena_attach": (C.c_int, [C.c_int, C.c_char_p]),
    "blz_msm_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "blz_msm_task_label": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "blz_msm_nof_elements": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "blz_msm_is_engine_ready": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "blz_msm_stream_progress": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "blz_msm_reset": (C.c_int, [C.c_void_p]),
    "blz_msm_last_timings": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "blz_comm_unique_id": (C.c_int, [_u8p]),
    "blz_msm_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _u8p]),
    "blz_msm_all_gather_combine": (C.c_int, [C.c_void_p, _u8p, _u8p, C.c_size_t]),
    "blz_msm_comm_init_all": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "blz_msm_all_gather_combine_all": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, _u8p, _u8p, C.c_size_t]),
    "blz_msm_comm_free": (C.c_int, [C.c_void_p]),
    "blz_msm_precompute_bases_device": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]),
    "blz_msm_combine_partials": (C.c_int, [C.c_void_p, _u8p, C.c_size_t, _u8p, C.c_size_t]),
    "blz_msm_last_sort_hidden": (C.c_int, [C.c_void_p, C.POINTER(C.c_int)]),
    "blz_msm_memory_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "blz_msm_set_window_table": (C.c_int, [C.c_void_p, C.c_int]),
    "blz_msm_prepare_window_table": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_int)]),
    "blz_msm_set_precompute_plan": (C.c_int, [C.c_void_p, C.c_int]),
    "blz_msm_prepare_precompute_plan": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(C.c_int)]),
    "blz_msm_precompute_plan_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "blz_msm_set_scalar_range": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "blz_msm_shard_layout": (C.c_int, [C.c_int, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32)]),
    "blz_msm_shard_layout_ex": (C.c_int, [C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_uint32)]),
    "blz_msm_shard_layout_candidate": (C.c_int, [C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]),
    "blz_msm_window_table_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "blz_msm_plan": (C.c_int, [C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)]),
    "blz_ntt_new": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "blz_ntt_new_ex": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "blz_ntt_new_field": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "blz_ntt_new_ex2": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]),
    "blz_ntt_new_ex3": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "blz_ntt_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "blz_ntt_exchange": (C.c_int, [C.c_void_p, C.c_size_t, _u8p, C.c_size_t, _u8p, C.c_size_t]),
    "blz_ntt_free": (None, [C.c_void_p]),
    "blz_ntt_initialize": (C.c_int, [C.c_void_p]),
    "blz_ntt_set_data": (C.c_int, [C.c_void_p, C.c_size_t, _u8p, C.c_size_t]),
    "blz_ntt_set_data_device": (C.c_int, [C.c_void_p, C.c_size_t, _u8p, C.c_size_t]),
    "blz_ntt_start_process": (C.c_int, [C.c_void_p, C.c_size_t]),
    "blz_ntt_wait_result": (C.c_int, [C.c_void_p]),
    "blz_ntt_result": (C.c_int, [C.c_void_p, C.c_size_t, _u8p, C.c_size_t]),
    "blz_ntt_result_device": (C.c_int, [C.c_void_p, C.c_size_t, _u8p, C.c_size_t]),
    "blz_ntt_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "blz_ntt_reset": (C.c_int, [C.c_void_p]),
    "blz_ntt_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "blz_ntt_banks_preprocess_device": (C.c_int, [C.c_void_p, _u8p, _u8p]),
    "blz_ntt_banks_postprocess_device": (C.c_int, [C.c_void_p, _u8p, _u8p]),
    "blz_device_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "blz_device_free": (C.c_int, [C.c_int, C.c_void_p]),
    "blz_host_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]),
    "blz_host_free": (C.c_int, [C.c_void_p]),
    "blz_memcpy_h2d": (C.c_int, [C.c_int, C.c_void_p, _u8p, C.c_size_t]),
    "blz_memcpy_d2h": (C.c_int, [C.c_int, _u8p, C.c_void_p, C.c_size_t]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)
# libblaze_hip_aux.so (include/blaze_hip_aux.h): test / bench scaffolding, not part of the product library
_AUX_SIGS = {
    "blz_synth_scalars": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64]),
    "blz_synth_scalars_at": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]),
    "blz_synth_points": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64]),
    "blz_synth_field_elements": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64, C.c_uint64]),
    "blz_calib_mad_rate": (C.c_int, [C.c_int, C.c_uint32, C.POINTER(C.c_double)]),
    "blz_test_msm_stall": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "blz_test_ntt_stall": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "blz_test_stall_release": (C.c_int, [C.c_void_p]),
    "blz_test_field_op": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _u8p, _u8p, C.c_size_t]),
    "blz_test_ec_op": (C.c_int, [C.c_int, C.c_int, C.c_int, _u8p, _u8p, _u8p, _u8p, _u8p, C.c_size_t]),
}
AUX_EXPORTED_SYMBOLS = tuple(_AUX_SIGS)


def lib():
    """Load libblaze_hip.so.  Fails loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C blaze_amd/csrc` "
                "(or __graft_entry__.build()); blaze_amd has no CPU fallback"
            )
        L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def aux():
    """Load libblaze_hip_aux.so - synthetic inputs, calibration, element-wise test hooks, stall kernels - next to the product
    library.  Used by tests/, bench.py and tools/ only."""
    global _aux
    if _aux is None:
        lib()   # the product library first: the aux library links against it (by soname)
        path = LIB_PATH[:-3] + "_aux.so"
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `make -C blaze_amd/csrc`")
        A = C.CDLL(path)
        for name, (res, args) in _AUX_SIGS.items():
            fn = getattr(A, name)
            fn.restype = res
            fn.argtypes = args
        _aux = A
    return _aux


def check(rc: int):
    if rc != 0:
        msg = lib().blz_last_error_message()
        raise DriverClientError(rc, msg.decode() if msg else "")


def buf_ptr(buf):
    """(void*, nbytes, keepalive) for bytes / bytearray / memoryview / numpy arrays; no copy."""
    if buf is None:
        return None, 0, None
    if isinstance(buf, bytes):
        return C.cast(C.c_char_p(buf), C.c_void_p), len(buf), buf
    if isinstance(buf, bytearray):
        arr = (C.c_char * len(buf)).from_buffer(buf)
        return C.cast(arr, C.c_void_p), len(buf), arr
    if hasattr(buf, "ctypes") and hasattr(buf, "nbytes"):  # numpy
        if not buf.flags["C_CONTIGUOUS"]:
            # (data, nbytes) of a strided view would name bytes that are not the view's: the caller decides whether to copy
            raise ValueError("numpy array is not C-contiguous: pass np.ascontiguousarray(a) (inputs) or a contiguous buffer (outputs)")
        return C.c_void_p(buf.ctypes.data), int(buf.nbytes), buf
    mv = memoryview(buf)
    if mv.readonly:
        b = mv.tobytes()
        return C.cast(C.c_char_p(b), C.c_void_p), len(b), b
    arr = (C.c_char * mv.nbytes).from_buffer(mv)
    return C.cast(arr, C.c_void_p), mv.nbytes, arr


class DeviceBuffer:
    """A device allocation owned by the library (blz_device_malloc)."""

    def __init__(self, device_id: int, nbytes: int):
        self.device_id = device_id
        self.nbytes = nbytes
        p = C.c_void_p()
        check(lib().blz_device_malloc(device_id, nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, data):
        p, n, _keep = buf_ptr(data)
        assert n <= self.nbytes
        check(lib().blz_memcpy_h2d(self.device_id, self.ptr, p, n))

    def download(self, nbytes: int | None = None, offset: int = 0) -> bytearray:
        n = self.nbytes - offset if nbytes is None else nbytes
        out = bytearray(n)
        p, _, _keep = buf_ptr(out)
        check(lib().blz_memcpy_d2h(self.device_id, p, self.ptr + offset, n))
        return out

    def free(self):
        if self.ptr:
            lib().blz_device_free(self.device_id, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HostBuffer:
    """Page-locked host memory owned by the library (blz_host_malloc): a writable buffer (memoryview / numpy via
    `array()`) that the runtime copies from and to without staging."""

    def __init__(self, device_id: int, nbytes: int):
        self.nbytes = nbytes
        p = C.c_void_p()
        check(lib().blz_host_malloc(device_id, nbytes, C.byref(p)))
        self.ptr = p.value
        self._arr = (C.c_uint8 * nbytes).from_address(self.ptr)

    def array(self):
        import numpy as np

        return np.frombuffer(self._arr, dtype=np.uint8)

    def free(self):
        if self.ptr:
            self._arr = None
            lib().blz_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
