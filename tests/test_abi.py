"""The C-ABI boundary without a GPU: the library loads, exports every symbol include/blaze_hip.h
declares, and fails loudly (FileError, no CPU fallback) when no device is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

import blaze_amd
from blaze_amd import DriverClientError
from blaze_amd._lib import AUX_EXPORTED_SYMBOLS, EXPORTED_SYMBOLS, LIB_PATH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="blaze_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(blz_[a-z0-9_]+)\s*\(", src)))


def _dynamic_symbols(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("blz_")}


def test_library_exports_every_declared_symbol():
    L = blaze_amd.lib()
    declared = _declared_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/blaze_hip.h but not exported"
    assert sorted(EXPORTED_SYMBOLS) == declared, "python binding table out of sync with the header"


def test_scaffolding_lives_in_the_aux_library_only():
    """Synthetic inputs, the calibration kernel, element-wise test hooks and the stall kernels are test / bench scaffolding:
    declared in include/blaze_hip_aux.h, exported by libblaze_hip_aux.so, and absent from the product library - a stall kernel
    a host can enqueue on a production handle is not a product feature."""
    A = blaze_amd.aux()
    declared = _declared_symbols("blaze_hip_aux.h")
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(A, name), f"{name} declared in include/blaze_hip_aux.h but not exported by the aux library"
    assert sorted(AUX_EXPORTED_SYMBOLS) == declared
    assert not set(declared) & set(_declared_symbols())
    product = _dynamic_symbols(LIB_PATH)
    assert not [s for s in product if s.startswith(("blz_test_", "blz_synth_", "blz_calib_"))]
    assert set(_declared_symbols()) <= product
    assert set(declared) <= _dynamic_symbols(LIB_PATH[:-3] + "_aux.so")


def test_sizes_match_reference_config():
    L = blaze_amd.lib()
    # src/ingo_msm/msm_cfg.rs:44-92
    assert (L.blz_point_size(0), L.blz_result_size(0)) == (96, 144)
    assert (L.blz_point_size(1), L.blz_result_size(1)) == (96, 144)
    assert (L.blz_point_size(2), L.blz_result_size(2)) == (64, 96)


def test_error_codes_follow_reference_enum():
    # src/error.rs:6-32 declaration order
    assert DriverClientError.VARIANTS == {
        1: "WriteError", 2: "ReadError", 3: "HBICAPNotReady", 4: "InvalidPrimitiveParam",
        5: "CsvError", 6: "LoadFailed", 7: "FileError", 8: "Unknown"}
    hdr = open(os.path.join(ROOT, "include", "blaze_hip.h")).read()
    for code, name in [(1, "BLZ_ERR_WRITE"), (2, "BLZ_ERR_READ"), (4, "BLZ_ERR_INVALID_PARAM"), (7, "BLZ_ERR_FILE"), (8, "BLZ_ERR_UNKNOWN")]:
        assert re.search(rf"{name}\s*=\s*{code}\b", hdr)


@pytest.mark.skipif(blaze_amd.lib().blz_device_count() > 0, reason="a GPU is present")
def test_no_gpu_means_loud_failure_not_fallback():
    L = blaze_amd.lib()
    h = C.c_void_p()
    rc = L.blz_msm_new(0, 1, 0, 1, C.byref(h))
    assert rc == 7 and not h.value            # FileError, like the failed open() of /dev/xdma0_*
    assert b"no CPU path" in L.blz_last_error_message()
    rc = L.blz_ntt_new(0, 10, C.byref(h))
    assert rc == 7
    with pytest.raises(DriverClientError) as ei:
        blaze_amd.driver_client.DriverClient(0)
    assert ei.value.variant == "FileError"


def test_invalid_arguments_are_rejected_before_touching_a_device():
    L = blaze_amd.lib()
    h = C.c_void_p()
    assert L.blz_msm_new(0, 1, 0, 7, C.byref(h)) == 4      # unknown curve -> InvalidPrimitiveParam
    assert L.blz_msm_new(0, 5, 0, 1, C.byref(h)) == 4      # unknown mem type
    assert L.blz_ntt_new(0, 28, C.byref(h)) == 4           # > 2^27
    assert L.blz_msm_wait_result(None) == 4
    assert L.blz_msm_initialize(None, 1, 0, 0, 0) == 4
    assert L.blz_msm_set_scalar_range(None, 0, 64) == 4
    assert L.blz_msm_set_window_table(None, 1) == 4
    out = (C.c_uint32 * 4)()
    assert L.blz_msm_shard_layout(1, 1 << 20, 0, 0, out) == 4      # no ranks
    assert L.blz_msm_shard_layout(1, 1 << 20, 4, 4, out) == 4      # rank out of range
    assert L.blz_msm_shard_layout(9, 1 << 20, 4, 0, out) == 4      # unknown curve
    assert L.blz_msm_shard_layout(1, 1 << 20, 4, 3, None) == 4
    assert L.blz_msm_shard_layout(1, 1 << 20, 1, 0, out) == 0 and list(out) == [0, 1 << 20, 0, 256]


def _build_cpp_example(tmp_path):
    import subprocess

    exe = str(tmp_path / "host_example")
    libdir = os.path.join(ROOT, "blaze_amd", "lib")
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "host_example.cpp"), "-L" + libdir, "-lblaze_hip", "-Wl,-rpath," + libdir,
           "-L/opt/rocm/lib", "-lamdhip64", "-o", exe]
    subprocess.check_call(cmd)
    return exe


@pytest.mark.skipif(blaze_amd.lib().blz_device_count() > 0, reason="a GPU is present")
def test_cpp_host_mirror_compiles_and_fails_loudly_without_gpu(tmp_path):
    """include/blaze.hpp (the C++ mirror of DriverPrimitive / MSMClient / NTTClient) builds against
    the C ABI; without a device it reports FileError (kind 7), it does not fall back."""
    import subprocess

    exe = _build_cpp_example(tmp_path)
    p = subprocess.run([exe, "a", "b", "1", "c"], capture_output=True, text=True)
    assert p.returncode == 1 and "kind 7" in p.stderr


def test_shard_layout_candidates_and_plan_override_are_host_side():
    """blz_msm_shard_layout_ex / _candidate need no device: argument checks, and the estimates scale as documented."""
    import ctypes as C

    import blaze_amd
    L = blaze_amd.lib()
    out = (C.c_uint32 * 8)()
    assert L.blz_msm_shard_layout_ex(1, 1 << 20, 0, 0, 0, out) == 4          # no ranks
    assert L.blz_msm_shard_layout_ex(1, 1 << 20, 4, 0, 0, None) == 4
    assert L.blz_msm_shard_layout_candidate(1, 1 << 20, 4, 0, 0, 3, out) == 4   # 3 ranges do not divide 4 ranks
    assert L.blz_msm_shard_layout_candidate(1, 1 << 20, 4, 0, 0, 8, out) == 4
    assert L.blz_msm_shard_layout_candidate(1, 1 << 26, 8, 5, 1, 2, out) == 0
    first, count, lo, hi, R, comp_us, link_us, mib = list(out)
    assert (first, count, lo, hi, R) == (2 * (1 << 24), 1 << 24, 128, 256, 2)     # rank 5 = chunk 2, range 1
    assert abs(link_us - (1 << 24) * 32 / 56.3e3) < 2 and mib == (1 << 24) * (96 + 128 + 32) >> 20 and comp_us > 0
    # more ranks than elements: the ranks' rectangles still tile the job
    area = 0
    for r in range(8):
        assert L.blz_msm_shard_layout_ex(0, 3, 8, r, 0, out) == 0
        area += out[1] * (out[3] - out[2])
    assert area == 3 * 256
