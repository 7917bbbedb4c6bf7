"""TEST INFRASTRUCTURE ONLY: ctypes loader for oracle/liboracle.so (the plain-C CPU restatement,
oracle/blz_oracle.c) plus the pure-Python big-int reference (oracle/pyref.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
PARITY UNPINNED against the reference at byte level (see blz_oracle.c header)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")

CURVE_ID = {"BLS377": 0, "BLS381": 1, "BN254": 2}


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "blz_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        u8p, u64 = C.c_char_p, C.c_uint64
        L.orc_point_bytes.argtypes = [C.c_int]
        L.orc_result_bytes.argtypes = [C.c_int]
        L.orc_decode_result.argtypes = [C.c_int, u8p, u8p]
        L.orc_is_on_curve.argtypes = [C.c_int, u8p]
        L.orc_generator_mul.argtypes = [C.c_int, u8p, u8p]
        L.orc_point_mul.argtypes = [C.c_int, u8p, u8p, u8p]
        L.orc_point_add.argtypes = [C.c_int, u8p, u8p, C.c_int, u8p]
        L.orc_precompute_base.argtypes = [C.c_int, u8p, C.c_int, u8p]
        L.orc_msm_naive.argtypes = [C.c_int, C.c_void_p, C.c_void_p, u64, C.c_int, u8p]
        L.orc_input_generator.argtypes = [C.c_int, u64, C.c_int, u64, C.c_void_p, C.c_void_p, u8p]
        L.orc_input_tile.argtypes = [C.c_int, u64, C.c_int, u64, C.c_void_p, C.c_void_p, u8p]
        L.orc_index_weighted_sum.argtypes = [C.c_int, C.c_void_p, u64, u64, u8p]
        L.orc_index_weighted_sum_mt.argtypes = [C.c_int, C.c_void_p, u64, u64, C.c_int, u8p]
        L.orc_msm_pippenger.argtypes = [C.c_int, C.c_void_p, C.c_void_p, u64, C.c_int, C.c_int, C.c_int, u8p]
        L.orc_omega.argtypes = [C.c_int, C.c_int, u8p]
        L.orc_ntt.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_ntt_ex.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
        L.orc_bitrev_permute.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.orc_ntt_eval_at.argtypes = [C.c_int, C.c_void_p, C.c_int, u64, u8p]
        L.orc_ntt_eval_at_mt.argtypes = [C.c_int, C.c_void_p, C.c_int, u64, C.c_int, u8p]
        L.orc_dft_naive.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_ntt_preprocess.argtypes = [C.c_void_p, C.c_void_p, u64]
        L.orc_ntt_postprocess.argtypes = [C.c_void_p, C.c_void_p, u64, u64]
        _lib = L
    return _lib


def _cid(curve) -> int:
    return CURVE_ID[curve] if isinstance(curve, str) else int(curve)


def _ptr(buf):
    """bytes / bytearray / numpy array -> void* (no copy for writable buffers)."""
    if isinstance(buf, bytes):
        return C.cast(C.c_char_p(buf), C.c_void_p)
    if isinstance(buf, bytearray):
        return C.cast((C.c_char * len(buf)).from_buffer(buf), C.c_void_p)
    return C.c_void_p(buf.ctypes.data)  # numpy


def point_bytes(curve) -> int:
    return lib().orc_point_bytes(_cid(curve))


def result_bytes(curve) -> int:
    return lib().orc_result_bytes(_cid(curve))


def decode_result(curve, res: bytes):
    """-> (affine x||y bytes or None for infinity, on_curve)."""
    out = C.create_string_buffer(point_bytes(curve))
    fl = lib().orc_decode_result(_cid(curve), bytes(res), out)
    if fl & 2:
        return None, True
    return out.raw, bool(fl & 1)


def is_on_curve(curve, xy: bytes) -> bool:
    return bool(lib().orc_is_on_curve(_cid(curve), bytes(xy)))


def generator_mul(curve, k: int):
    out = C.create_string_buffer(point_bytes(curve))
    inf = lib().orc_generator_mul(_cid(curve), int(k).to_bytes(32, "little"), out)
    return None if inf else out.raw


def point_mul(curve, xy: bytes, k: int):
    out = C.create_string_buffer(point_bytes(curve))
    inf = lib().orc_point_mul(_cid(curve), bytes(xy), int(k).to_bytes(32, "little"), out)
    return None if inf else out.raw


def point_add(curve, p, q):
    pb = point_bytes(curve)
    out = C.create_string_buffer(pb)
    fl = (1 if p is None else 0) | (2 if q is None else 0)
    z = b"\0" * pb
    inf = lib().orc_point_add(_cid(curve), z if p is None else bytes(p), z if q is None else bytes(q), fl, out)
    return None if inf else out.raw


def precompute_base(curve, xy: bytes, pf: int) -> bytes:
    out = C.create_string_buffer(point_bytes(curve) * pf)
    lib().orc_precompute_base(_cid(curve), bytes(xy), pf, out)
    return out.raw


def msm_naive(curve, points, scalars, n: int, pf: int = 1) -> bytes:
    out = C.create_string_buffer(result_bytes(curve))
    rc = lib().orc_msm_naive(_cid(curve), _ptr(points), _ptr(scalars), n, pf, out)
    assert rc == 0
    return out.raw


def msm_pippenger(curve, points, scalars, n: int, pf: int = 1, threads: int = 1, cbits: int = 0) -> bytes:
    out = C.create_string_buffer(result_bytes(curve))
    rc = lib().orc_msm_pippenger(_cid(curve), _ptr(points), _ptr(scalars), n, pf, threads, cbits, out)
    assert rc == 0
    return out.raw


def input_generator(curve, n: int, pf: int, seed: int):
    """-> (points bytearray, scalars bytearray, expected result bytes) per tests/msm/mod.rs:297-358."""
    pts = bytearray(n * pf * point_bytes(curve))
    sc = bytearray(n * 32)
    exp = C.create_string_buffer(result_bytes(curve))
    rc = lib().orc_input_generator(_cid(curve), n, pf, seed, _ptr(pts), _ptr(sc), exp)
    assert rc == 0
    return pts, sc, exp.raw


def input_tile(curve, n: int, pf: int, seed: int):
    """The generator's <= 256-element tile only (points, scalars) and the expected result for all n elements
    (same stream as input_generator: input_generator(...) == tile repeated)."""
    t = min(n, 256)
    pts = bytearray(t * pf * point_bytes(curve))
    sc = bytearray(t * 32)
    exp = C.create_string_buffer(result_bytes(curve))
    rc = lib().orc_input_tile(_cid(curve), n, pf, seed, _ptr(pts), _ptr(sc), exp)
    assert rc == 0
    return pts, sc, exp.raw


def index_weighted_sum(curve, scalars, n: int, start: int = 0, threads: int = 1) -> int:
    out = C.create_string_buffer(32)
    if threads > 1:
        rc = lib().orc_index_weighted_sum_mt(_cid(curve), _ptr(scalars), n, start, threads, out)
    else:
        rc = lib().orc_index_weighted_sum(_cid(curve), _ptr(scalars), n, start, out)
    assert rc == 0
    return int.from_bytes(out.raw, "little")


def result_from_affine(curve, xy) -> bytes:
    """Canonical device encoding Z=1|y|x (or Z=0|Y=1|X=0 for infinity)."""
    fb = point_bytes(curve) // 2
    if xy is None:
        return b"\0" * fb + (1).to_bytes(fb, "little") + b"\0" * fb
    return (1).to_bytes(fb, "little") + xy[fb:] + xy[:fb]


def omega(curve, logn: int) -> int:
    out = C.create_string_buffer(32)
    rc = lib().orc_omega(_cid(curve), logn, out)
    assert rc == 0
    return int.from_bytes(out.raw, "little")


def ntt(curve, data, logn: int, inverse: bool = False, threads: int = 1, root: int | None = None, bitrev_in: bool = False,
        bitrev_out: bool = False) -> bytearray:
    """root: any primitive 2^logn-th root of unity instead of the generator's; bitrev_*: the buffer is in bit-reversed order
    (the conventions the device's blz_ntt_new_ex3 offers; the reference states none)."""
    out = bytearray(32 << logn)
    flags = int(inverse) | (2 if bitrev_in else 0) | (4 if bitrev_out else 0)
    rb = None if root is None else C.c_char_p(int(root).to_bytes(32, "little"))
    rc = lib().orc_ntt_ex(_cid(curve), _ptr(data), _ptr(out), logn, flags, rb, threads)
    assert rc == 0, rc
    return out


def bitrev_permute(data, logn: int, threads: int = 1) -> bytearray:
    """out[bitrev(i)] = in[i] over logn bits, 32-byte elements."""
    out = bytearray(32 << logn)
    rc = lib().orc_bitrev_permute(_ptr(data), _ptr(out), logn, threads)
    assert rc == 0
    return out


def ntt_eval_at(curve, data, logn: int, k: int, threads: int = 1) -> int:
    out = C.create_string_buffer(32)
    if threads > 1:
        rc = lib().orc_ntt_eval_at_mt(_cid(curve), _ptr(data), logn, k, threads, out)
    else:
        rc = lib().orc_ntt_eval_at(_cid(curve), _ptr(data), logn, k, out)
    assert rc == 0
    return int.from_bytes(out.raw, "little")


def dft_naive(curve, data, logn: int) -> bytearray:
    out = bytearray(32 << logn)
    rc = lib().orc_dft_naive(_cid(curve), _ptr(data), _ptr(out), logn)
    assert rc == 0
    return out


def ntt_preprocess(data, n: int) -> bytearray:
    out = bytearray(32 * n)
    rc = lib().orc_ntt_preprocess(_ptr(data), _ptr(out), n)
    assert rc == 0
    return out


def ntt_postprocess(banks, n: int, groups: int) -> bytearray:
    out = bytearray(32 * n)
    rc = lib().orc_ntt_postprocess(_ptr(banks), _ptr(out), n, groups)
    assert rc == 0
    return out
