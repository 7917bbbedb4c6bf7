"""The Rust binding (rust/src/driver_client/hip_ffi.rs) cannot be compiled in this image (no rustc), so its
extern "C" block is checked against include/blaze_hip.h textually: every declared function exists in the header
with the same number of parameters and compatible parameter / return types; the error-code mapping of
dclient.rs covers every variant of the header's enum; and the image-parameter word the library emits is decoded
by a Python transcription of MSMImageParametrs::parse_image_params (msm_api.rs:350-354)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _c_decls():
    txt = open(os.path.join(ROOT, "include", "blaze_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", " ", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", " ", txt)
    txt = re.sub(r"^\s*#[^\n]*", " ", txt, flags=re.M)     # preprocessor lines
    decls = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(blz_\w+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        decls[name] = (ret, params)
    return decls


def _rust_decls():
    txt = open(os.path.join(ROOT, "rust", "src", "driver_client", "hip_ffi.rs")).read()
    txt = re.sub(r"//[^\n]*", " ", txt)
    block = txt[txt.index('extern "C" {'):]
    decls = {}
    for m in re.finditer(r"pub fn (blz_\w+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "()").strip()
        params = [a.strip() for a in args.split(",") if a.strip()]
        decls[name] = (ret, [p.split(":", 1)[1].strip() for p in params])
    return decls


def _c_kind(t):
    t = t.strip()
    t = re.sub(r"\b\w+$", "", t).strip() if not t.endswith("*") and " " in t else t   # drop the parameter name
    t = re.sub(r"\[[^\]]*\]", "*", t)          # arrays decay
    t = t.replace("const", "").strip()
    if "*" in t:
        return "ptr"
    t = re.sub(r"\s+", " ", t)
    return {"int": "i32", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "float": "f32", "void": "void"}.get(t, t)


def _c_param_kind(p):
    if "*" in p or "[" in p:
        return "ptr"
    toks = p.replace("const", "").split()
    return _c_kind(" ".join(toks[:-1]) if len(toks) > 1 else toks[0])


def _rust_kind(t):
    t = t.strip()
    if t.startswith("*"):
        return "ptr"
    return {"c_int": "i32", "u32": "u32", "u64": "u64", "usize": "usize", "f32": "f32", "()": "void"}.get(t, t)


def test_every_rust_extern_matches_the_header():
    c, r = _c_decls(), _rust_decls()
    assert len(r) >= 35
    for name, (rret, rparams) in r.items():
        assert name in c, f"{name} is not declared in include/blaze_hip.h"
        cret, cparams = c[name]
        assert len(cparams) == len(rparams), f"{name}: {len(rparams)} parameters in Rust, {len(cparams)} in the header"
        assert _rust_kind(rret) == ("ptr" if "*" in cret else _c_kind(cret)), f"{name}: return type {rret} vs {cret}"
        for i, (cp, rp) in enumerate(zip(cparams, rparams)):
            assert _rust_kind(rp) == _c_param_kind(cp), f"{name} parameter {i}: {rp} vs {cp}"
    # the DriverPrimitive methods of both primitives are all bound
    for need in ("blz_msm_new", "blz_msm_initialize", "blz_msm_set_data", "blz_msm_start_process", "blz_msm_wait_result",
                 "blz_msm_result", "blz_msm_loaded_binary_parameters", "blz_ntt_new", "blz_ntt_initialize",
                 "blz_ntt_set_data", "blz_ntt_start_process", "blz_ntt_wait_result", "blz_ntt_result"):
        assert need in r


def test_error_codes_are_all_mapped():
    hdr = open(os.path.join(ROOT, "include", "blaze_hip.h")).read()
    codes = {int(v): k for k, v in re.findall(r"(BLZ_ERR_\w+)\s*=\s*(\d+)", hdr)}
    assert sorted(codes) == list(range(1, 9))
    # the crate maps codes in one place (error.rs from_code / code) and check() in dclient.rs goes through it
    err = open(os.path.join(ROOT, "rust", "src", "error.rs")).read()
    arms = dict(re.findall(r"^\s*(\d+) => Self::(\w+)", err, flags=re.M))
    back = {v: k for k, v in re.findall(r"^\s*Self::(\w+)[^=\n]*=> (\d+),", err, flags=re.M)}
    want = {1: "WriteError", 2: "ReadError", 3: "HBICAPNotReady", 4: "InvalidPrimitiveParam", 5: "CsvError", 6: "LoadFailed", 7: "FileError"}
    for code, variant in want.items():
        assert arms.get(str(code)) == variant, (code, arms.get(str(code)))
        assert back.get(str(code)) == variant, (code, back.get(str(code)))
    assert "_ => Self::Unknown" in err and back.get("8") == "Unknown"
    dcl = open(os.path.join(ROOT, "rust", "src", "driver_client", "dclient.rs")).read()
    assert "DriverClientError::from_code(" in dcl
    # the Python mirror agrees
    from blaze_amd._lib import DriverClientError
    assert DriverClientError.VARIANTS == {**want, 8: "Unknown"}


def test_image_parameter_word_layout():
    from blaze_amd.ingo_msm import MSMImageParametrs, pack_image_params

    # transcription of the reference decode: params.reverse_bits().to_be_bytes(), packed_struct msb0 bit ranges
    def reference_decode(p):
        rev = int(f"{p:032b}"[::-1], 2)
        bits = f"{rev:032b}"                      # msb0 string of the big-endian buffer
        f = lambda lo, hi: int(bits[lo:hi + 1], 2)   # noqa: E731
        return dict(is_stub=f(28, 31), curve=f(20, 27), adders=f(16, 19), width=f(8, 15), segments=f(4, 7), place_holder=f(0, 3))

    for curve_code in (0, 1, 2):
        for adders, width, segs in ((15, 21, 8), (1, 0, 0), (7, 255, 15)):
            w = pack_image_params(curve_code, adders, width, segs)
            d = reference_decode(w)
            assert d == dict(is_stub=0, curve=curve_code << 2, adders=adders, width=width, segments=segs, place_holder=0)
            m = MSMImageParametrs.parse_image_params(w)
            assert (m.hif2cpu_c_is_stub, m.hif2_cpu_c_curve, m.hif2_cpu_c_number_of_ec_adders,
                    m.hif2_cpu_c_buckets_mem_addr_width, m.hif2_cpu_c_number_of_segments, m.hif2_cpu_c_place_holder) == \
                   (0, curve_code << 2, adders, width, segs, 0)
            assert m.curve_name() == ["BLS12_377", "BN254", "BLS12_381"][curve_code]


def test_rust_sources_only_use_what_is_declared():
    """More of what a compiler would have said: every blz_* function a Rust source CALLS is declared in hip_ffi.rs, and every
    method the ported integration tests / benches call on a client is defined in msm_api.rs / ntt_api.rs / dclient.rs."""
    import glob

    rust = os.path.join(ROOT, "rust")
    declared = set(_rust_decls())
    used = set()
    for f in glob.glob(os.path.join(rust, "src", "**", "*.rs"), recursive=True):
        if f.endswith("hip_ffi.rs"):
            continue
        used |= set(re.findall(r"\b(blz_\w+)\s*\(", open(f).read()))
    assert used and not used - declared, sorted(used - declared)
    methods = set()
    for rel in ("src/ingo_msm/msm_api.rs", "src/ingo_ntt/ntt_api.rs", "src/driver_client/dclient.rs"):
        methods |= set(re.findall(r"fn (\w+)\s*[<(]", open(os.path.join(rust, rel)).read()))
    calls = set()
    for f in glob.glob(os.path.join(rust, "tests", "*.rs")) + glob.glob(os.path.join(rust, "benches", "*.rs")):
        calls |= set(re.findall(r"\b(?:driver|second|dclient|client)\.(\w+)\(", open(f).read()))
    assert calls and not calls - methods, sorted(calls - methods)
    # the round-5 entry points are bound
    for need in ("blz_msm_set_precompute_plan", "blz_msm_prepare_precompute_plan", "blz_msm_precompute_plan_info", "blz_msm_memory_info",
                 "blz_ntt_exchange", "blz_ntt_info", "blz_ntt_new_ex2", "blz_arena_set_policy", "blz_host_malloc"):
        assert need in declared, need
