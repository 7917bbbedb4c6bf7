"""Host-side mirror of the reference's types (no GPU needed)."""
import dataclasses

from blaze_amd.driver_client import CardType, DriverConfig, DriverPrimitive
from blaze_amd.ingo_msm import (PRECOMPUTE_FACTOR, PRECOMPUTE_FACTOR_BASE, Curve, MSMClient, MSMConfig, MSMInit,
                                MSMInput, MSMParams, MSMResult, PointMemoryType)
from blaze_amd.ingo_ntt import NTT, NTT_LOG_SIZE, NTTClient, NTTInput, NttInit


def test_enum_orders_match_reference():
    assert [c.name for c in Curve] == ["BLS377", "BLS381", "BN254"]          # msm_cfg.rs:4-8
    assert [m.name for m in PointMemoryType] == ["HBM", "DMA"]               # msm_cfg.rs:11-14
    assert (PRECOMPUTE_FACTOR_BASE, PRECOMPUTE_FACTOR) == (1, 8)             # msm_api.rs:39-40
    assert NTT_LOG_SIZE == 27                                                # ntt_data.rs:65


def test_type_fields_match_reference():
    assert [f.name for f in dataclasses.fields(MSMInit)] == ["mem_type", "is_precompute", "curve"]
    assert [f.name for f in dataclasses.fields(MSMParams)] == ["nof_elements", "hbm_point_addr"]
    assert [f.name for f in dataclasses.fields(MSMInput)] == ["points", "scalars", "params"]
    assert [f.name for f in dataclasses.fields(MSMResult)] == ["result", "result_label"]
    assert [f.name for f in dataclasses.fields(NTTInput)] == ["buf_host", "data"]
    assert dataclasses.fields(NttInit) == ()
    assert NTT.Ntt is not None


def test_clients_implement_the_seven_method_trait():
    methods = {"__init__", "loaded_binary_parameters", "initialize", "set_data", "start_process", "wait_result", "result"}
    assert methods <= set(DriverPrimitive.__abstractmethods__)                # dclient.rs:28-46
    for cls in (MSMClient, NTTClient):
        assert issubclass(cls, DriverPrimitive)
        assert not getattr(cls, "__abstractmethods__", None)
    for extra in ("task_label", "nof_elements", "is_msm_engine_ready", "load_data_to_hbm", "get_data_from_hbm", "get_api"):
        assert hasattr(MSMClient, extra)                                      # msm_api.rs:277-331


def test_msm_config_sizes():
    assert MSMConfig.msm_cfg(Curve.BLS381, PointMemoryType.DMA) == MSMConfig(144, 96, 32)
    assert MSMConfig.msm_cfg(Curve.BN254, PointMemoryType.HBM) == MSMConfig(96, 64, 32)
    assert DriverConfig.driver_client_cfg(CardType.MI355X).card is CardType.MI355X


def test_window_plan_invariants():
    """make_plan through the C ABI (host code, no device): the window widths cover the scalar width + 1
    for signed digits, use at most two adjacent widths below the top window, the workspace bounds hold
    and the choice grows with the input."""
    import ctypes as C
    import blaze_amd
    L = blaze_amd.lib()
    prev_c = {}
    for curve in (0, 1, 2):
        for pf in (0, 1):
            for logn in range(0, 27):
                n = 1 << logn
                if pf and n * 8 >= 1 << 31:
                    continue
                out = (C.c_uint32 * 4)()
                wd = (C.c_uint8 * 96)()
                assert L.blz_msm_plan(curve, n, pf, out, wd) == 0, L.blz_last_error_message()
                c, W, unit, G = list(out)
                widths = list(wd)[:W]
                sbits = 32 if pf else 256
                assert all(3 <= x <= 23 for x in widths) and max(widths[:-1] or widths) == c
                assert sum(widths) >= sbits + 1
                assert sum(widths[:-1]) < sbits + 1 + min(widths)          # no window beyond the carry
                lower = widths[:-1]
                assert not lower or max(lower) - min(lower) <= 1            # two adjacent widths
                assert lower == sorted(lower, reverse=True) and widths[-1] >= min(widths)
                assert G == sum(1 << (x - 1) for x in widths) and G <= 1 << 26
                assert G % (1 << (min(widths) - 1)) == 0                    # whole virtual windows
                assert 16 <= unit <= 256 and unit & (unit - 1) == 0
                assert n * (8 if pf else 1) * W < 1 << 32
                key = (curve, pf)
                if logn >= 8:
                    assert c + 1 >= prev_c.get(key, 0)       # grows with n (a dip of one bit between W steps is fine)
                    prev_c[key] = max(prev_c.get(key, 0), c)
    out = (C.c_uint32 * 4)()
    assert L.blz_msm_plan(1, 1 << 26, 0, out, None) == 0 and out[0] in (20, 21, 22)
    assert L.blz_msm_plan(7, 10, 0, out, None) != 0
    assert L.blz_msm_plan(1, 0, 0, out, None) != 0


def test_readme_switch_table_is_what_the_sources_read():
    """at most eight runtime switches, and README.md's table is exactly the set of environment variables
    the shipped sources read (everything else sits behind exp_knob(), which only an experiment build turns into getenv)."""
    import glob
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    readme = open(os.path.join(root, "README.md")).read()
    table = readme[readme.index("<!-- switches:begin -->"): readme.index("<!-- switches:end -->")]
    documented = set(re.findall(r"^\| `(BLAZE_[A-Z0-9_]+)`", table, flags=re.M))
    read = set()
    for path in glob.glob(os.path.join(root, "blaze_amd", "csrc", "*")):
        if not path.endswith((".hip", ".hpp", ".h", ".inc")):
            continue
        src = open(path).read()
        read |= set(re.findall(r'(?<![a-z_])getenv\("(BLAZE_[A-Z0-9_]+)"\)', src))
        read |= set(re.findall(r'(?<![a-z_])env_int\("(BLAZE_[A-Z0-9_]+)"', src))
    for path in glob.glob(os.path.join(root, "blaze_amd", "*.py")):
        read |= set(re.findall(r'environ(?:\.get\(|\[)"(BLAZE_[A-Z0-9_]+)"', open(path).read()))
    assert read == documented, (sorted(read - documented), sorted(documented - read))
    assert len(documented) <= 8
    # ... and no experiment knob reaches the environment in the shipped build
    hpp = open(os.path.join(root, "blaze_amd", "csrc", "common.hpp")).read()
    assert "#ifdef BLZ_EXPERIMENT_KNOBS" in hpp and "inline int exp_knob(const char*, int dflt) { return dflt; }" in hpp
    assert "BLZ_EXPERIMENT_KNOBS" not in open(os.path.join(root, "blaze_amd", "csrc", "Makefile")).read().replace("EXTRA=-DBLZ_EXPERIMENT_KNOBS", "")


def test_buffers_handed_to_the_c_abi():
    """blaze_amd._lib.buf_ptr turns host buffers into (pointer, length) without copying: bytes, bytearray, memoryview, contiguous numpy
    arrays - and refuses a strided numpy view, whose (data, nbytes) would name bytes that are not the view's."""
    import numpy as np
    import pytest

    from blaze_amd._lib import buf_ptr

    assert buf_ptr(None)[:2] == (None, 0)
    for b in (b"abcd", bytearray(b"abcd"), memoryview(bytearray(b"abcd")), np.frombuffer(b"abcd", dtype=np.uint8), memoryview(b"abcd")):
        p, n, _keep = buf_ptr(b)
        assert n == 4 and p
    assert buf_ptr(bytearray())[1] == 0 and buf_ptr(b"")[1] == 0
    a = np.arange(64, dtype=np.uint8).reshape(8, 8)
    assert buf_ptr(a)[1] == 64 and buf_ptr(a[2:4])[1] == 16          # row slices stay contiguous
    with pytest.raises(ValueError):
        buf_ptr(a[:, :4])                                             # a strided view is refused
    with pytest.raises(ValueError):
        buf_ptr(a[::2])


def test_only_the_checkers_touch_the_oracle():
    """The oracle is test infrastructure: only tests/ (its probes included), __graft_entry__.smoke() and bench.py with its extra legs
    (bench_extras.py: result checks and the cpu_baseline legs) may import it; nothing under blaze_amd/, include/, rust/ or tools/ does, and the product library links
    nothing of it."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    allowed = {"bench.py", "bench_extras.py", "__graft_entry__.py"}
    pat = re.compile(r"^\s*(import|from)\s+([\w., ]*\boracle\b)", re.M)
    offenders = []
    for base, dirs, files in os.walk(root):
        rel = os.path.relpath(base, root)
        dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out", "build", "__pycache__") and not (rel == "." and d in ("tests", "oracle"))]
        for f in files:
            p = os.path.relpath(os.path.join(base, f), root)
            if f.endswith((".py", ".sh")) and p not in allowed and pat.search(open(os.path.join(base, f), errors="replace").read()):
                offenders.append(p)
    assert offenders == [], offenders
    mk = open(os.path.join(root, "blaze_amd", "csrc", "Makefile")).read()
    assert "oracle" not in mk
    for base, _dirs, files in os.walk(os.path.join(root, "blaze_amd", "csrc")):
        for f in files:
            if f.endswith((".hip", ".hpp", ".h", ".inc")):
                assert "blz_oracle" not in open(os.path.join(base, f), errors="replace").read(), f
