"""CPU model checks of the reduced-radix arithmetic the device code relies on (field_rr.hip.hpp), over the constants
tools/gen_constants.py emits: the borrow-form multiples of m, the column-sum and value bounds of the three DFT
steps' types, and the one-digit quotient estimate of rr_reduce2m.  No GPU, no oracle: plain integers."""
import importlib.util
import os
import random

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("gen_constants", os.path.join(HERE, "..", "tools", "gen_constants.py"))
gc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gc)

FIELDS = []
for name, cid, q, r, b, gx, gy, nq, s, g in gc.CURVES:
    if f"Fq_{name}" in gc.RR:
        FIELDS.append((f"Fq_{name}", q) + gc.RR[f"Fq_{name}"])
    if f"Fr_{name}" in gc.RR:
        FIELDS.append((f"Fr_{name}", r) + gc.RR[f"Fr_{name}"])


def limbs_of(v, B, NL):
    out = [(v >> (B * i)) & ((1 << B) - 1) for i in range(NL - 1)]
    out.append(v >> (B * (NL - 1)))
    return out


@pytest.mark.parametrize("name,m,B,NL", FIELDS)
def test_reduce2m_quotient_digit(name, m, B, NL):
    """q = floor(x_top MU / 2^32), MU = floor(2^32 / (m_top + 1)): never above floor(x / m), at most 1 below, for every
    x < V m the kernels feed it (V up to the field's value head-room), exact multiples of m and their neighbours first."""
    head = B * NL - m.bit_length()
    m_top = m >> (B * (NL - 1))
    MU = (1 << 32) // (m_top + 1)
    vmax = min(1 << head, 64)
    # the static_asserts of rr_reduce2m, evaluated for the largest V used
    if not (4 * (vmax + 1) <= m_top and (vmax + 1) * (m_top + 1) <= 1 << 30):
        vmax = max(v for v in range(2, vmax + 1) if 4 * (v + 1) <= m_top and (v + 1) * (m_top + 1) <= 1 << 30)
    rng = random.Random(hash(name) & 0xFFFF)
    cases = []
    for k in range(vmax + 1):
        for d in (-2, -1, 0, 1, 2):
            x = k * m + d
            if 0 <= x < vmax * m:
                cases.append(x)
    cases += [rng.randrange(vmax * m) for _ in range(20000)]
    cases += [vmax * m - 1, 0, (1 << (B * (NL - 1))) - 1, 1 << (B * (NL - 1))]
    for x in cases:
        if x >= vmax * m:
            continue
        x_top = x >> (B * (NL - 1))
        assert x_top < 1 << 32
        qd = (x_top * MU) >> 32
        qt = x // m
        assert qt - 1 <= qd <= qt, (name, hex(x), qd, qt)
        y = x - qd * m
        assert 0 <= y < 2 * m
        # the limb-wise subtraction with a borrow chain reproduces y (top limb keeps the rest)
        a = limbs_of(x, B, NL)
        t = limbs_of(qd * m, B, NL)
        borrow, out = 0, []
        for i in range(NL - 1):
            d = a[i] - t[i] - borrow
            borrow = 1 if d < 0 else 0
            out.append(d & ((1 << B) - 1))
        out.append(a[NL - 1] - t[NL - 1] - borrow)
        assert out[NL - 1] >= 0
        assert sum(v << (B * i) for i, v in enumerate(out)) == y


@pytest.mark.parametrize("name,m,B,NL", FIELDS)
def test_borrow_form_multiples(name, m, B, NL):
    """KM[j-1] = 2^j m with every limb but the top raised by 2^B and the next lowered by 1: the same integer, every
    lower limb >= 2^B - 1 (so u + (KM - v) never borrows for a normalised v) and the top limb >= that of any v < 2^(j-1) m."""
    nk = min(14, B * NL - m.bit_length() - 1)
    mask = (1 << B) - 1
    for j in range(1, nk + 1):
        v = m << j
        d = [(v >> (B * i)) & mask for i in range(NL)]
        d[NL - 1] = v >> (B * (NL - 1))
        e = [d[i] + ((1 << B) if i < NL - 1 else 0) - (1 if i > 0 else 0) for i in range(NL)]
        assert sum(x << (B * i) for i, x in enumerate(e)) == v
        assert all(mask <= x < 1 << 32 for x in e[:-1])
        assert e[NL - 1] >= ((m << (j - 1)) >> (B * (NL - 1)))
        assert e[NL - 1] < 1 << 32


@pytest.mark.parametrize("name,m,B,NL", FIELDS)
def test_column_sums_fit_64_bits(name, m, B, NL):
    """rr_cols_ok: (sum of Fa Fb + 1) NL + 1 <= 2^(64 - 2B) is exactly "no column of the product scanning can overflow":
    worst-case limbs, the reduction's own products and the carried-in high part included."""
    lim = 1 << (64 - 2 * B)
    fmax = max(f for f in range(1, 4096) if (f + 1) * NL + 1 <= lim)
    # worst case for one column: NL products of limbs < fa 2^B and < fb 2^B with fa fb = fmax, NL products q_i m_j
    # (both < 2^B), plus the carry of the previous column (< 2^(64 - B))
    worst = NL * fmax * ((1 << B) - 1) ** 2 + NL * ((1 << B) - 1) ** 2 + (1 << (64 - B))
    assert worst < 1 << 64, (name, fmax)
    # and the bound is not vacuous: one more unit of limb slack can overflow
    over = NL * (fmax + 2) * ((1 << B) - 1) ** 2 + NL * ((1 << B) - 1) ** 2 + (1 << (64 - B))
    assert over >= (1 << 64) or fmax > 1000
