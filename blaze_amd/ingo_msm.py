"""Host-side mirror of src/ingo_msm (msm_api.rs, msm_cfg.rs): same type names, fields and call
sequence; every method forwards 1:1 to the C ABI (include/blaze_hip.h)."""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass
from typing import Optional, Tuple

from ._lib import DeviceBuffer, buf_ptr, check, lib
from .driver_client import DriverClient, DriverPrimitive

PRECOMPUTE_FACTOR_BASE = 1  # msm_api.rs:39
PRECOMPUTE_FACTOR = 8  # msm_api.rs:40


class Curve(enum.IntEnum):  # msm_cfg.rs:4-8 (declaration order)
    BLS377 = 0
    BLS381 = 1
    BN254 = 2


class PointMemoryType(enum.IntEnum):  # msm_cfg.rs:11-14
    HBM = 0
    DMA = 1


@dataclass
class MSMInit:  # msm_api.rs:16-20
    mem_type: PointMemoryType
    is_precompute: bool
    curve: Curve


@dataclass
class MSMParams:  # msm_api.rs:23-26
    nof_elements: int
    hbm_point_addr: Optional[Tuple[int, int]] = None


@dataclass
class MSMInput:  # msm_api.rs:28-32
    points: Optional[object]  # bytes-like, or DeviceBuffer for HBM-resident data
    scalars: object
    params: MSMParams


@dataclass
class MSMResult:  # msm_api.rs:33-37
    result: bytes
    result_label: int


@dataclass(frozen=True)
class MSMConfig:  # msm_cfg.rs:17-30 (sizes only: the DMA FIFO addresses have no GPU meaning)
    result_point_size: int
    point_size: int
    scalar_size: int

    @staticmethod
    def msm_cfg(curve: Curve, _mem: PointMemoryType) -> "MSMConfig":  # msm_cfg.rs:32-41
        return MSMConfig(int(lib().blz_result_size(int(curve))), int(lib().blz_point_size(int(curve))), 32)


def _hbm(params: MSMParams):
    if params.hbm_point_addr is None:
        return 0, 0, 0
    return 1, int(params.hbm_point_addr[0]), int(params.hbm_point_addr[1])


def _field_msb_first(p: int, lo: int, hi: int) -> int:
    v = 0
    for b in range(lo, hi + 1):      # bit `lo` of the word is the field's most significant bit
        v = (v << 1) | ((p >> b) & 1)
    return v


def _put_msb_first(v: int, lo: int, hi: int) -> int:
    w = 0
    for k, b in enumerate(range(hi, lo - 1, -1)):   # the field's bit k sits at word bit hi - k
        w |= ((v >> k) & 1) << b
    return w


def pack_image_params(curve_code: int, ec_adders: int, buckets_mem_addr_width: int, segments: int, is_stub: int = 0) -> int:
    """The word MSMImageParametrs.parse_image_params decodes (what blz_msm_loaded_binary_parameters emits)."""
    return (_put_msb_first(is_stub, 28, 31) | _put_msb_first((curve_code << 2) & 0xFF, 20, 27) | _put_msb_first(ec_adders, 16, 19)
            | _put_msb_first(buckets_mem_addr_width, 8, 15) | _put_msb_first(segments, 4, 7))


@dataclass
class MSMImageParametrs:  # msm_api.rs:333-347 (packed_struct, msb0, after params.reverse_bits(): :350-354)
    hif2cpu_c_is_stub: int
    hif2_cpu_c_curve: int
    hif2_cpu_c_number_of_ec_adders: int
    hif2_cpu_c_buckets_mem_addr_width: int
    hif2_cpu_c_number_of_segments: int
    hif2_cpu_c_place_holder: int

    @staticmethod
    def parse_image_params(params: int) -> "MSMImageParametrs":
        return MSMImageParametrs(_field_msb_first(params, 28, 31), _field_msb_first(params, 20, 27), _field_msb_first(params, 16, 19),
                                 _field_msb_first(params, 8, 15), _field_msb_first(params, 4, 7), _field_msb_first(params, 0, 3))

    def curve_name(self) -> str:
        """msm_api.rs:359-364 intends 0 / 1 / 2 = BLS12_377 / BN254 / BLS12_381 above the low two flag bits."""
        return {0: "BLS12_377", 1: "BN254", 2: "BLS12_381"}.get(self.hif2_cpu_c_curve >> 2, "UNKNOWN")

    def debug_information(self) -> str:
        return (f"Is Stub: {self.hif2cpu_c_is_stub}; curve: {self.curve_name()}; EC adders (x16 CUs): "
                f"{self.hif2_cpu_c_number_of_ec_adders}; buckets memory address width: {self.hif2_cpu_c_buckets_mem_addr_width}; "
                f"segments (XCDs): {self.hif2_cpu_c_number_of_segments}")


class MSMClient(DriverPrimitive[MSMInit, MSMParams, MSMInput, MSMResult]):
    """msm_api.rs:8-14, 42-331."""

    def __init__(self, init: MSMInit, dclient: DriverClient):
        self.mem_type = init.mem_type
        self.precompute_factor = PRECOMPUTE_FACTOR if init.is_precompute else PRECOMPUTE_FACTOR_BASE
        self.curve = Curve(init.curve)
        self.msm_cfg = MSMConfig.msm_cfg(self.curve, init.mem_type)
        self.driver_client = dclient
        h = C.c_void_p()
        check(lib().blz_msm_new(dclient.id, int(init.mem_type), int(bool(init.is_precompute)), int(init.curve), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().blz_msm_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- DriverPrimitive
    def loaded_binary_parameters(self) -> list[int]:  # msm_api.rs:57-70
        out = (C.c_uint32 * 2)()
        check(lib().blz_msm_loaded_binary_parameters(self._h, out))
        return [int(out[0]), int(out[1])]

    def initialize(self, params: MSMParams) -> None:  # msm_api.rs:72-111
        has, addr, off = _hbm(params)
        check(lib().blz_msm_initialize(self._h, params.nof_elements, has, addr, off))

    def start_process(self, _param: Optional[int] = None) -> None:  # msm_api.rs:113-120
        check(lib().blz_msm_start_process(self._h))

    def set_data(self, data: MSMInput) -> None:  # msm_api.rs:155-220
        has, addr, off = _hbm(data.params)
        on_dev = isinstance(data.scalars, DeviceBuffer)
        if on_dev:
            pp = data.points.ptr if data.points is not None else None
            pn = data.points.nbytes if data.points is not None else 0
            check(lib().blz_msm_set_data_device(self._h, pp, pn, data.scalars.ptr, data.scalars.nbytes,
                                                data.params.nof_elements, has, addr, off))
            return
        pp, pn, _k1 = buf_ptr(data.points)
        sp, sn, _k2 = buf_ptr(data.scalars)
        check(lib().blz_msm_set_data(self._h, pp, pn, sp, sn, data.params.nof_elements, has, addr, off))

    def wait_result(self) -> None:  # msm_api.rs:222-238
        check(lib().blz_msm_wait_result(self._h))

    def result(self, _param: Optional[int] = None) -> Optional[MSMResult]:  # msm_api.rs:240-274
        out = C.create_string_buffer(self.msm_cfg.result_point_size)
        n = C.c_size_t()
        label = C.c_uint32()
        check(lib().blz_msm_result(self._h, C.cast(out, C.c_void_p), len(out), C.byref(n), C.byref(label)))
        return MSMResult(out.raw[: n.value], int(label.value))

    # ---- extras (msm_api.rs:277-331)
    def task_label(self) -> int:
        v = C.c_uint32()
        check(lib().blz_msm_task_label(self._h, C.byref(v)))
        return int(v.value)

    def nof_elements(self) -> int:
        v = C.c_uint32()
        check(lib().blz_msm_nof_elements(self._h, C.byref(v)))
        return int(v.value)

    def is_msm_engine_ready(self) -> int:
        v = C.c_uint32()
        check(lib().blz_msm_is_engine_ready(self._h, C.byref(v)))
        return int(v.value)

    def stream_progress(self) -> Tuple[int, int]:
        """A task fed by several set_data calls (blaze_hip.h "STREAMED TASKS": with a task queued, a set_data whose
        params.nof_elements is smaller than what the task still lacks is its next slice, msm_api.rs:155-202 /
        msm_hw_code.rs:18-19): (elements received so far, elements of the queued task)."""
        v = (C.c_uint32 * 2)()
        check(lib().blz_msm_stream_progress(self._h, v))
        return int(v[0]), int(v[1])

    def load_data_to_hbm(self, points, addr: int, offset: int) -> None:
        if isinstance(points, DeviceBuffer):
            check(lib().blz_msm_load_data_to_hbm_device(self._h, points.ptr, points.nbytes, addr, offset))
            return
        p, n, _k = buf_ptr(points)
        check(lib().blz_msm_load_data_to_hbm(self._h, p, n, addr, offset))

    def get_data_from_hbm(self, data_len: int, addr: int, offset: int) -> bytes:
        out = bytearray(data_len)
        p, _, _k = buf_ptr(out)
        check(lib().blz_msm_get_data_from_hbm(self._h, p, data_len, addr, offset))
        return bytes(out)

    def get_api(self) -> dict:
        """msm_api.rs:324-330 dumps every register; here: the phase timers of the last task."""
        t = (C.c_float * 8)()
        check(lib().blz_msm_last_timings(self._h, t))
        keys = ["total_ms", "accumulate_kernel_ms", "sort_ms", "phase1_accumulate_ms", "phase2_reduce_ms", "phase3_final_ms",
                "window_bits", "windows"]
        api = dict(zip(keys, [float(x) for x in t]))
        hid = C.c_int(0)
        check(lib().blz_msm_last_sort_hidden(self._h, C.byref(hid)))
        api["sort_hidden"] = int(hid.value)
        api["device_memory"] = self.memory_info()
        return api

    def memory_info(self) -> dict:
        """Device bytes behind this client (include/blaze_hip.h blz_msm_memory_info); the arena figures are per device."""
        m = (C.c_uint64 * 6)()
        check(lib().blz_msm_memory_info(self._h, m))
        keys = ["workspace", "staging", "arena_raw", "arena_montgomery", "arena_window_tables", "total"]
        return dict(zip(keys, [int(x) for x in m]))

    def reset(self) -> None:
        check(lib().blz_msm_reset(self._h))

    # ---- multi-GPU exchange inside the library (include/blaze_hip.h, blz_msm_comm_*): RCCL over xGMI
    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0 makes the communicator id; the host ships these 128 bytes to the other ranks."""
        out = C.create_string_buffer(128)
        check(lib().blz_comm_unique_id(C.cast(out, C.c_void_p)))
        return out.raw

    def set_window_table(self, enable) -> None:
        """Opt in to the resident-base window table (include/blaze_hip.h blz_msm_set_window_table): pf = 1 handles whose
        bases live in the arena; built beside the tasks (prepare_window_table), W x the memory of the bases, fewer bucket additions.
        False / 0 off, True / 1 where it pays (the BLS curves), 2 always."""
        check(lib().blz_msm_set_window_table(self._h, int(enable)))

    def prepare_window_table(self, nof_elements: int, hbm_addr=(0, 0), wait_ms: int = -1) -> bool:
        """Enqueue the table's build for the bases at hbm_addr - it runs beside the tasks, which take the plain path until it
        is there - and wait up to wait_ms for it (0: not at all; < 0: the library's wait deadline).  True: the table is in place."""
        ready = C.c_int(0)
        check(lib().blz_msm_prepare_window_table(self._h, nof_elements, hbm_addr[0], hbm_addr[1], wait_ms, C.byref(ready)))
        return bool(ready.value)

    def set_precompute_plan(self, enable) -> None:
        """Opt a precompute client (MSMInit.is_precompute) in to the checked-table plan (include/blaze_hip.h
        blz_msm_set_precompute_plan): resident x8 tables are checked once per load against precompute_base_*
        (tests/msm/mod.rs:360-380) and, if consistent, served as 4n even bases with 64-bit chunks - 12 bucket additions per
        element instead of 16, identical result bytes.  A table that fails the check keeps the exact path."""
        check(lib().blz_msm_set_precompute_plan(self._h, int(bool(enable))))

    def prepare_precompute_plan(self, nof_elements: int, hbm_addr=(0, 0)) -> bool:
        """Run the table check (and build the even-base copy) now; True: tasks over these bases take the plan."""
        ok = C.c_int(0)
        check(lib().blz_msm_prepare_precompute_plan(self._h, nof_elements, hbm_addr[0], hbm_addr[1], C.byref(ok)))
        return bool(ok.value)

    def precompute_plan_info(self) -> dict:
        """Of the last HBM task: did it take the plan, what the check of its bases said, what the check cost."""
        out = (C.c_uint64 * 4)()
        check(lib().blz_msm_precompute_plan_info(self._h, out))
        return {"used": bool(out[0]), "check": ["unchecked", "consistent", "refuted"][int(out[1])], "check_ms": int(out[2]) / 1000.0,
                "even_copy_bytes": int(out[3])}

    def set_scalar_range(self, bit_lo: int, bit_hi: int) -> None:
        """This client's tasks sum only bits [bit_lo, bit_hi) of every scalar and return 2^bit_lo x that sum: one shard of a
        job split by scalar chunk (include/blaze_hip.h blz_msm_set_scalar_range; blaze_amd.multi_gpu.shard_layout).
        (0, 256) = the whole scalar."""
        check(lib().blz_msm_set_scalar_range(self._h, bit_lo, bit_hi))

    def window_table_info(self) -> dict:
        """Of the table the last HBM task used (all zero: it took the plain path)."""
        out = (C.c_uint64 * 4)()
        check(lib().blz_msm_window_table_info(self._h, out))
        return {"bytes": int(out[0]), "window_bits": int(out[1]), "windows": int(out[2]), "build_ms": int(out[3]) / 1000.0}

    def comm_init(self, rank: int, nranks: int, comm_id: bytes) -> None:
        p, _n, _k = buf_ptr(comm_id)
        check(lib().blz_msm_comm_init(self._h, rank, nranks, p))

    def all_gather_combine(self, partial: bytes) -> bytes:
        """All-gather of the ranks' partial results + rank-ordered add on the device: the full result, identical
        bytes on every rank."""
        out = C.create_string_buffer(self.msm_cfg.result_point_size)
        p, _n, _k = buf_ptr(partial)
        check(lib().blz_msm_all_gather_combine(self._h, p, C.cast(out, C.c_void_p), len(out)))
        return out.raw

    @staticmethod
    def comm_init_all(clients) -> None:
        """One process, one thread, one client per device: all communicator ranks as ONE RCCL group (rank i =
        clients[i]).  The per-rank comm_init is a blocking rendezvous and cannot be called in sequence from one thread."""
        hs = (C.c_void_p * len(clients))(*[c._h for c in clients])
        check(lib().blz_msm_comm_init_all(hs, len(clients)))

    @staticmethod
    def all_gather_combine_all(clients, partials) -> list:
        """The exchange for the clients of comm_init_all: partials[i] is client i's result; every client's sum comes back."""
        rs = clients[0].msm_cfg.result_point_size
        assert len(partials) == len(clients) and all(len(p) == rs for p in partials)
        hs = (C.c_void_p * len(clients))(*[c._h for c in clients])
        flat = b"".join(bytes(p) for p in partials)
        out = C.create_string_buffer(rs * len(clients))
        p, _n, _k = buf_ptr(flat)
        check(lib().blz_msm_all_gather_combine_all(hs, len(clients), p, C.cast(out, C.c_void_p), len(out)))
        return [out.raw[i * rs: (i + 1) * rs] for i in range(len(clients))]

    def comm_free(self) -> None:
        check(lib().blz_msm_comm_free(self._h))

    def combine_partials(self, partials: bytes, count: int) -> bytes:
        """Multi-GPU: rank-ordered sum of `count` partial results."""
        out = C.create_string_buffer(self.msm_cfg.result_point_size)
        p, _n, _k = buf_ptr(partials)
        check(lib().blz_msm_combine_partials(self._h, p, count, C.cast(out, C.c_void_p), len(out)))
        return out.raw
