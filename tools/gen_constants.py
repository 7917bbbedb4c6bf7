#!/usr/bin/env python3
"""Emit blaze_amd/csrc/curve_constants.h: 32-bit-limb Montgomery constants for the three
curves' base fields (Fq) and scalar fields (Fr).  Everything is derived here from the published
moduli / generators by big-int arithmetic; tests/test_constants.py re-checks the emitted values
against the literals of SURVEY.md appendix B."""
import os

CURVES = [
    ("BLS377", 0,
     0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001,
     0x12AB655E9A2CA55660B44D1E5C37B00159AA76FED00000010A11800000000001, 1,
     0x008848DEFE740A67C8FC6225BF87FF5485951E2CAA9D41BB188282C8BD37CB5CD5481512FFCD394EEAB9B16EB21BE9EF,
     0x01914A69C5102EFF1F674F5D30AFEEC4BD7FB348CA3E52D96D182AD44FB82305C2FE3D3634A9591AFD82DE55559C8EA6,
     12, 47, 22),
    ("BLS381", 1,
     0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
     0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001, 4,
     0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
     0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
     12, 32, 7),
    ("BN254", 2,
     0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47,
     0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001, 3, 1, 2,
     8, 28, 5),
]


def limbs(v, n):
    return ", ".join("0x%08xu" % ((v >> (32 * i)) & 0xFFFFFFFF) for i in range(n))


def field_struct(name, m, n, extra=""):
    R = 1 << (32 * n)
    lazy = 4 * m < R  # values may live in [0,2m) and Montgomery needs no final subtraction
    n0 = (-pow(m, -1, 1 << 32)) % (1 << 32)
    out = []
    out.append(f"struct {name} {{")
    out.append(f"    static constexpr int N = {n};")
    out.append(f"    static constexpr int BITS = {m.bit_length()};")
    out.append(f"    static constexpr bool LAZY = {'true' if lazy else 'false'};  // R > 4m: redundant [0,2m) form allowed")
    out.append(f"    static constexpr uint32_t N0 = 0x{n0:08x}u;  // -m^-1 mod 2^32")
    out.append(f"    static constexpr uint32_t MOD[{n}] = {{{limbs(m, n)}}};")
    out.append(f"    static constexpr uint32_t MOD2[{n}] = {{{limbs(2 * m, n)}}};  // 2m")
    out.append(f"    static constexpr uint32_t R1[{n}] = {{{limbs(R % m, n)}}};  // R mod m")
    out.append(f"    static constexpr uint32_t R2[{n}] = {{{limbs(R * R % m, n)}}};  // R^2 mod m")
    out.append(f"    static constexpr uint32_t MODM2[{n}] = {{{limbs(m - 2, n)}}};  // m-2 (Fermat exponent)")
    if extra:
        out.append(extra)
    out.append("};")
    return "\n".join(out)


def rr_struct(name, m, n32, B, NL):
    """Reduced-radix twin of a field: NL limbs of B bits in 32-bit registers, Montgomery radix
    Rrr = 2^(B NL).  Limbs leave 32 - B spare bits (carry-free add / sub) and the 64-bit column sums of
    v_mad_u64_u32 never overflow, so a 32x32 multiply-add is ONE instruction (field_rr.hip.hpp)."""
    Rrr = 1 << (B * NL)
    assert Rrr > (m << 5), "value head-room of the lazy range"
    nk = min(14, B * NL - m.bit_length() - 1)  # multiples 2^j m that still fit NL limbs
    mask = (1 << B) - 1

    def L(v, cnt=NL):
        assert v >> (B * cnt) == 0
        return ", ".join("0x%08xu" % ((v >> (B * i)) & mask) for i in range(cnt))

    n0 = (-pow(m, -1, 1 << B)) % (1 << B)
    out = [f"struct {name} {{"]
    out.append(f"    static constexpr int B = {B};    // bits per limb")
    out.append(f"    static constexpr int NL = {NL};  // limbs")
    out.append(f"    static constexpr int N32 = {n32};  // limbs of the 32-bit twin")
    out.append(f"    static constexpr int BITS = {m.bit_length()};")
    out.append(f"    static constexpr uint32_t MASK = 0x{mask:08x}u;")
    out.append(f"    static constexpr uint32_t N0 = 0x{n0:08x}u;    // -m^-1 mod 2^B")
    out.append(f"    static constexpr uint32_t MINV = 0x{pow(m, -1, 1 << B):08x}u;  // m^-1 mod 2^B")
    out.append(f"    static constexpr uint32_t MOD[{NL}] = {{{L(m)}}};")
    out.append(f"    static constexpr uint32_t ONE[{NL}] = {{{L(Rrr % m)}}};  // Rrr mod m")
    out.append(f"    static constexpr uint32_t RR2[{NL}] = {{{L(Rrr * Rrr % m)}}};  // Rrr^2 mod m")
    out.append(f"    static constexpr uint32_t TO32[{NL}] = {{{L((1 << (32 * n32)) % m)}}};  // 2^(32 N32) mod m, plain: x Rrr -> x R32")
    out.append(f"    static constexpr uint32_t FROM32[{NL}] = {{{L(Rrr * Rrr * pow(1 << (32 * n32), -1, m) % m)}}};  // Rrr^2 / R32: x R32 -> x Rrr")
    out.append(f"    static constexpr uint32_t MBAR[{NL}] = {{{L(Rrr - m)}}};  // Rrr - m: the Shoup product's x w - q m as a sum (field_rr.hip.hpp)")
    out.append(f"    static constexpr uint32_t T2M = 0x{(2 * m) >> (B * (NL - 1)):08x}u;  // top limb of 2m: a value whose top limb is below it is < 2m")
    # K m in "borrow form" for carry-free subtraction a - b + K m: limb i gains 2^B, limb i + 1 loses 1,
    # so every limb but the top is >= 2^B - 1 >= any normalised limb of b, and the top limb is
    # (K m)_top - 1 >= b_top whenever b < (K / 2) m.
    rows = []
    for j in range(1, nk + 1):
        v = m << j
        d = [(v >> (B * i)) & mask for i in range(NL)]
        d[NL - 1] = v >> (B * (NL - 1))
        e = [0] * NL
        for i in range(NL):
            e[i] = d[i] + ((1 << B) if i < NL - 1 else 0) - (1 if i > 0 else 0)
        assert sum(e[i] << (B * i) for i in range(NL)) == v and all(0 <= x < (1 << 32) for x in e)
        assert e[NL - 1] >= ((m << (j - 1)) >> (B * (NL - 1)))
        rows.append("{" + ", ".join("0x%08xu" % x for x in e) + "}")
    out.append(f"    static constexpr int NKM = {nk};")
    out.append(f"    static constexpr uint32_t KM[{nk}][{NL}] = {{  // KM[j-1] = 2^j m, borrow form")
    out.append("        " + ",\n        ".join(rows) + "};")
    out.append("};")
    return "\n".join(out)


# reduced-radix twins: (field, B, NL)
# BLS base fields: 14 x 28 bits (392 product multiply-adds, 4 spare bits per limb, 11+ bits of value head-room).
# Every 254 / 255-bit field - the three scalar fields (2^27 NTT) and, from round 3 on, BN254's base field (bucket
# accumulation) - takes 9 x 29 bits = 261 bits: 162 multiply-adds per product (10 x 27 bits: 200; 8 x 32 bits with a
# carry word: 128 pairs = 256 instructions).  The price is a tight lazy range - 3 spare bits per limb (limbs < 8 x 2^29),
# 6-8 bits of value head-room (values < 64 m ... 128 m) and 64-bit column sums that admit only sum(Fa Fb) <= 6 - so
# lazy operands are carry-propagated before a product and accumulator coordinates are kept below 2m with the one-digit
# quotient reduction (rr_reduce2m); all of it is checked at compile time by the bounds in the Frr type.
# History: round 2 tried BN254's base field on 10 x 27 bits and measured the same 60 ms for its accumulation at 2^26
# as on 8 x 32, and blamed the gathers; round 3 measured the gathers out (profiles/r03_bn254_gather_split.txt: 5 %) -
# 200 multiply-adds simply cost what 128 multiply-add / add-carry pairs cost.
RR = {"Fq_BLS377": (28, 14), "Fq_BLS381": (28, 14), "Fq_BN254": (29, 9), "Fr_BLS377": (29, 9), "Fr_BLS381": (29, 9), "Fr_BN254": (29, 9)}


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "..", "blaze_amd", "csrc", "curve_constants.h")
    o = []
    o.append("// GENERATED by tools/gen_constants.py - do not edit.")
    o.append("// 32-bit-limb little-endian Montgomery constants, R = 2^(32 N).")
    o.append("#pragma once")
    o.append("#include <cstdint>")
    o.append("namespace blz {")
    for name, cid, q, r, b, gx, gy, nq, s, g in CURVES:
        Rq = 1 << (32 * nq)
        extra = []
        extra.append(f"    static constexpr uint32_t CURVE_B[{nq}] = {{{limbs(b * Rq % q, nq)}}};  // b, Montgomery")
        extra.append(f"    static constexpr uint32_t GX[{nq}] = {{{limbs(gx * Rq % q, nq)}}};  // generator x, Montgomery")
        extra.append(f"    static constexpr uint32_t GY[{nq}] = {{{limbs(gy * Rq % q, nq)}}};  // generator y, Montgomery")
        if f"Fq_{name}" in RR:
            o.append(rr_struct(f"Fq_{name}_RR", q, nq, *RR[f"Fq_{name}"]))
            extra.append(f"    using RR = Fq_{name}_RR;  // reduced-radix twin (field_rr.hip.hpp)")
        else:
            extra.append("    using RR = void;")
        o.append(field_struct(f"Fq_{name}", q, nq, "\n".join(extra)))
        Rr = 1 << 256
        root = pow(g, (r - 1) >> s, r)
        assert pow(root, 1 << s, r) == 1 and pow(root, 1 << (s - 1), r) == r - 1
        extra = []
        extra.append(f"    static constexpr int TWO_ADICITY = {s};")
        extra.append(f"    static constexpr uint32_t ROOT[8] = {{{limbs(root * Rr % r, 8)}}};  // primitive 2^{s}-th root, Montgomery")
        extra.append(f"    static constexpr uint32_t ROOT_INV[8] = {{{limbs(pow(root, -1, r) * Rr % r, 8)}}};  // its inverse, Montgomery")
        if f"Fr_{name}" in RR:
            o.append(rr_struct(f"Fr_{name}_RR", r, 8, *RR[f"Fr_{name}"]))
            extra.append(f"    using RR = Fr_{name}_RR;  // reduced-radix twin (field_rr.hip.hpp), used by the 2^27 NTT")
        else:
            extra.append("    using RR = void;")
        o.append(field_struct(f"Fr_{name}", r, 8, "\n".join(extra)))
        o.append(f"struct Curve_{name} {{ using Fq = Fq_{name}; using Fr = Fr_{name}; static constexpr int ID = {cid}; }};")
    o.append("}  // namespace blz")
    with open(dst, "w") as f:
        f.write("\n".join(o) + "\n")
    print("wrote", os.path.normpath(dst))


if __name__ == "__main__":
    main()
