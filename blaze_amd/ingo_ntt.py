"""Host-side mirror of src/ingo_ntt (ntt_api.rs): same names and call sequence over the C ABI."""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass
from typing import Optional

from ._lib import DeviceBuffer, buf_ptr, check, lib
from .driver_client import DriverClient, DriverPrimitive

NTT_LOG_SIZE = 27  # ntt_data.rs:65: NTT_SIZE = 2^27
NTT_WORD_SIZE = 32  # ntt_data.rs:66


class NTT(enum.Enum):  # ntt_api.rs:8-10
    Ntt = 0


@dataclass
class NttInit:  # ntt_api.rs:17
    pass


@dataclass
class NTTInput:  # ntt_api.rs:19-23
    buf_host: int
    data: object  # bytes-like of 2^log_size * 32 bytes, or a DeviceBuffer


class NTTClient(DriverPrimitive[NTT, NttInit, NTTInput, bytes]):
    """ntt_api.rs:12-15, 25-125.  `log_size` defaults to the reference's fixed 2^27; smaller
    transforms exist for tests (the reference has no such knob).  `field` names the curve whose
    scalar field the transform is over ("BLS381" by default, "BLS377", "BN254")."""

    _FIELDS = {"BLS377": 0, "BLS381": 1, "BN254": 2}  # enum blz_curve

    NO_FACTOR_TABLE = 1  # include/blaze_hip.h BLZ_NTT_NO_FACTOR_TABLE
    INVERSE = 2          # BLZ_NTT_INVERSE
    BITREV_INPUT = 4     # BLZ_NTT_BITREV_INPUT
    BITREV_OUTPUT = 8    # BLZ_NTT_BITREV_OUTPUT

    def __init__(self, _ptype: NTT, dclient: DriverClient, log_size: int = NTT_LOG_SIZE, inverse: bool = False,
                 field: str = "BLS381", flags: int = 0, root: Optional[int] = None):
        """root / BITREV_* flags: the transform's convention, which the reference leaves unstated (NttInit {} is empty,
        ntt_api.rs:8-23) - any primitive 2^log_size-th root of unity (an int, checked on the device) instead of this
        build's g^((r - 1) / 2^log_size), input and / or output in bit-reversed order (include/blaze_hip.h blz_ntt_new_ex3)."""
        self.driver_client = dclient
        self.log_size = log_size
        self.inverse = inverse or bool(flags & self.INVERSE)
        self.field = field
        self.nbytes = NTT_WORD_SIZE << log_size
        h = C.c_void_p()
        rb = None if root is None else int(root).to_bytes(32, "little")
        check(lib().blz_ntt_new_ex3(dclient.id, self._FIELDS[field], log_size, int(flags) | (self.INVERSE if inverse else 0), rb, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib().blz_ntt_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def loaded_binary_parameters(self) -> list[int]:
        raise NotImplementedError("todo!() in the reference too (ntt_api.rs:33-35)")

    def initialize(self, _param: NttInit = NttInit()) -> None:  # ntt_api.rs:37-56
        check(lib().blz_ntt_initialize(self._h))

    def start_process(self, buf_kernel: Optional[int] = None) -> None:  # ntt_api.rs:58-70
        if buf_kernel is None:
            raise TypeError("buf_kernel is required (the reference unwraps it: ntt_api.rs:62)")
        check(lib().blz_ntt_start_process(self._h, buf_kernel))

    def set_data(self, input: NTTInput) -> None:  # ntt_api.rs:72-87
        if isinstance(input.data, DeviceBuffer):
            check(lib().blz_ntt_set_data_device(self._h, input.buf_host, input.data.ptr, input.data.nbytes))
            return
        p, n, _k = buf_ptr(input.data)
        check(lib().blz_ntt_set_data(self._h, input.buf_host, p, n))

    def wait_result(self) -> None:  # ntt_api.rs:89-108
        check(lib().blz_ntt_wait_result(self._h))

    def result(self, buf_num: Optional[int] = None) -> Optional[bytes]:  # ntt_api.rs:110-124
        if buf_num is None:
            raise TypeError("buf_num is required (the reference unwraps it: ntt_api.rs:113)")
        out = bytearray(self.nbytes)
        p, _, _k = buf_ptr(out)
        check(lib().blz_ntt_result(self._h, buf_num, p, self.nbytes))
        return out

    def result_into(self, buf_num: int, out) -> None:
        """result() into a caller-owned writable buffer (bytearray, numpy array ...) of 32 * n bytes: a host that
        keeps its output vector between transforms does not pay the first-touch page faults of a fresh 4 GiB
        allocation on every call (the C ABI takes the caller's pointer anyway)."""
        p, nb, _k = buf_ptr(out)
        if nb != self.nbytes:
            raise ValueError(f"result buffer holds {nb} bytes, the transform has {self.nbytes}")
        check(lib().blz_ntt_result(self._h, buf_num, p, self.nbytes))

    def exchange(self, buf_host: int, next_input, out) -> None:
        """result(buf_host) into `out` and set_data(NTTInput(buf_host, next_input)) as one full-duplex call
        (include/blaze_hip.h blz_ntt_exchange): what a cycle of ntt_parallel_test_correctness does on the buffer the
        kernel is not using (tests/integration_ntt.rs:102-136), with both directions of the link busy at once."""
        pi, ni, _k1 = buf_ptr(next_input)
        po, no, _k2 = buf_ptr(out)
        if no < self.nbytes:
            raise ValueError(f"result buffer holds {no} bytes, the transform has {self.nbytes}")
        check(lib().blz_ntt_exchange(self._h, buf_host, pi, ni, po, no))

    def info(self) -> dict:
        """Device bytes this client holds and which pass-2 kernel it runs (include/blaze_hip.h blz_ntt_info)."""
        v = (C.c_uint64 * 4)()
        check(lib().blz_ntt_info(self._h, v))
        return {"device_bytes": int(v[0]), "pass2_factor_table": bool(v[1]), "pass1_boundary_table": bool(v[2]), "log_size": int(v[3])}

    def result_device(self, buf_num: int, dst: DeviceBuffer) -> None:
        check(lib().blz_ntt_result_device(self._h, buf_num, dst.ptr, dst.nbytes))

    def reset(self) -> None:
        check(lib().blz_ntt_reset(self._h))

    # NTTBanks::preprocess / postprocess (ntt_data.rs:80-156) on device buffers
    def banks_preprocess(self, d_in: DeviceBuffer, d_banks: DeviceBuffer) -> None:
        check(lib().blz_ntt_banks_preprocess_device(self._h, d_in.ptr, d_banks.ptr))

    def banks_postprocess(self, d_banks: DeviceBuffer, d_out: DeviceBuffer) -> None:
        check(lib().blz_ntt_banks_postprocess_device(self._h, d_banks.ptr, d_out.ptr))

    def last_kernel_ms(self) -> float:
        v = C.c_float()
        check(lib().blz_ntt_last_kernel_ms(self._h, C.byref(v)))
        return float(v.value)
