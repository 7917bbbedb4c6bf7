"""bench.py prices the two dominant kernels against the chip's integer multiply issue rate with multiply-add counts that
used to be hand-counted constants.  These tests count the v_mad_u64_u32 of the SHIPPED gfx950 code
object: the hot loop of k_accumulate<Fq_BLS381> and the three passes of the 2^27 NTT kernel."""
import os
import re

import pytest

import blaze_amd
from isa_util import count, disassemble_library, function_instructions, loops, tools_available

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def disasm():
    if not tools_available():
        pytest.skip("ROCm LLVM tools not installed")
    lib = os.environ.get("BLAZE_HIP_LIB") or os.path.join(ROOT, "blaze_amd", "lib", "libblaze_hip.so")
    text = disassemble_library(lib)
    assert "v_mad_u64_u32" in text
    return text


def _bench_constant(pattern: str) -> int:
    src = open(os.path.join(ROOT, "bench.py")).read()
    m = re.search(pattern, src)
    assert m, pattern
    return int(m.group(1))


def test_accumulate_loop_multiply_adds(disasm):
    """One bucket addition (mixed XYZZ add on 14 x 28-bit limbs) = 6 products x 392 + 2 squarings x 301 + one fused sum of two
    products x 588 = 3542 multiply-adds (ec_rr.hip.hpp): the loop of k_accumulate<Fq_BLS381> that walks a unit's run holds exactly
    that many, and bench.py's roofline.integer_issue multiplies by the same number."""
    assert 6 * 392 + 2 * 301 + 588 == 3542
    # (BLS12-377's q is 1 mod 2^28: its quotient digits are negations, 14 multiply-adds less in each of the 9 reductions)
    for curve, per_add in (("Fq_BLS381", 3542), ("Fq_BLS377", 3542 - 9 * 14)):
        ins = function_instructions(disasm, f"_ZN3blz12k_accumulateINS_9{curve}EEEvPKjS3_S3_S3_S3_S3_S3_jPj")
        per_loop = sorted(count(body, "v_mad_u64_u32") for _, _, body in loops(ins))
        assert per_add in per_loop, (curve, per_loop)    # the steady-state iteration: acc += point
        assert count(ins, "v_mad_u64_u32") >= per_add + 1800   # ... and the run's first addition, affine + affine, beside it
    assert _bench_constant(r"mads = n_loc \* occupied \* (\d+)") == 3542


def test_ntt_pass_multiply_adds(disasm):
    """The 2^27 NTT's per-lane multiply-adds: 37 / 37 / 29 field products per pass at 143 (Shoup, table
    twiddles) or 153 (Montgomery: pass 2's boundary factors, read from the per-element table) plus a 9-multiply-add quotient
    reduction per un-twiddled output.  Pass 2's table kernel has no cold alternative path, so its STATIC count is its per-lane
    count; passes 1 and 3 also carry the no-boundary-table chain and the inverse transform's closing products, so theirs bound
    it from above; pass 2 without the table (smaller transforms) steps its factors: 36 Shoup + 10 Montgomery products."""
    want = [37 * 143 + 2 * 9, 29 * 143 + 8 * 153 + 2 * 9, 29 * 143 + 10 * 9]
    assert sum(want) == 14935
    got = []
    for p, tab in ((1, 0), (2, 1), (3, 0)):
        ins = function_instructions(disasm, f"_ZN3blz11k_ntt512_rrINS_9Fr_BLS381ELi{p}ELb{tab}EEEvPKjPjNS_7NttGeomENS_11NttTablesRRE")
        got.append(count(ins, "v_mad_u64_u32"))
    assert abs(got[1] - want[1]) <= 8, (got, want)
    assert want[0] <= got[0] <= want[0] + 19 * 153 + 8 * 143, (got, want)     # + the stepping chain of a transform without the boundary table
    assert want[2] <= got[2] <= want[2] + 8 * 153, (got, want)                # + the inverse transform's n^-1 products
    stepped = function_instructions(disasm, "_ZN3blz11k_ntt512_rrINS_9Fr_BLS381ELi2ELb0EEEvPKjPjNS_7NttGeomENS_11NttTablesRRE")
    assert abs(count(stepped, "v_mad_u64_u32") - (36 * 143 + 10 * 153 + 2 * 9)) <= 8
    src = open(os.path.join(ROOT, "bench_extras.py")).read()
    # bench.py's NTT leg (bench_extras.py) prices the transform with these very counts, picking pass 2's by what blz_ntt_info reports for the handle it times
    assert "pass2_table, pass2_stepped = 29 * 143 + 8 * 153 + 2 * 9, 36 * 143 + 10 * 153 + 2 * 9" in src
    assert '(37 * 143 + 2 * 9) + (pass2_table if ninfo["pass2_factor_table"] else pass2_stepped) + (29 * 143 + 10 * 9)' in src


def test_table_check_doubling_multiply_adds(disasm):
    """The checked-table plan's device check walks chains of 32 Jacobian doublings (ec_rr.hip.hpp ptrr_jdbl): 3 squarings x 301 +
    2 products x 392 + one fused sum of two products x 588 = 2275 multiply-adds on 14 x 28 bits; 969 on 9 x 29
    (945 in the six products, the rest in the two one-digit quotient reductions): the doubling loop of the shipped kernels holds
    that many."""
    assert 3 * 301 + 2 * 392 + 588 == 2275
    for curve, want in (("9Fq_BLS381", 2275), ("9Fq_BLS377", 2275 - 6 * 14), ("8Fq_BN254", 969)):
        ins = function_instructions(disasm, f"_ZN3blz18k_check_precomputeINS_{curve}EEEvPKjmPj")
        per_loop = [count(body, "v_mad_u64_u32") for _, _, body in loops(ins)]
        assert want in per_loop, (curve, per_loop)
