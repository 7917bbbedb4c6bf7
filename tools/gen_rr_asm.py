#!/usr/bin/env python3
"""Emit blaze_amd/csrc/rr_gen.inc: inline-asm column batches of the reduced-radix multiplier
(field_rr.hip.hpp).  One function = one asm statement = all products of one kind in one column, chained on the
64-bit column accumulator with v_mad_u64_u32 (its carry-out, which the column bound rules out, goes to vcc):

    rr_ab<NL, K>(acc, a, b)      acc += sum_{i+j=K} a_i b_j
    rr_sq<NL, K>(acc, a, a2)     acc += sum_{i<j, i+j=K} a_i a2_j  [+ a_(K/2)^2]     (a2 = 2 a limbwise)
    rr_qm<NL, K>(acc, q, m...)   acc += sum_{i+j=K, i<K or K>=NL} q_i m_j           (m_j wave-uniform: SGPR operands)
    rr_abqm<NL, K>(acc, a, b, q, m)   both of the above in one statement (columns whose operands fit: rr_fused_ok)

Why asm: left to itself hipcc (a) re-associates each column into "products first, carry-in last" and closes it
with an extra v_lshl_add_u64, (b) hoists products of later columns, which in the fused a b + c d form costs
190 extra live registers.  hipcc pads every asm statement that consumes a just-written VGPR with one s_nop,
hence whole columns per statement (an asm statement takes at most 30 operands: a column of 14 + 14 operands
plus the accumulator just fits)."""
import os
import sys


def emit_ab(NL, K):
    ilo = 0 if K < NL else K - NL + 1
    ihi = K if K < NL else NL - 1
    idx = list(range(ilo, ihi + 1))
    lines = [f'"v_mad_u64_u32 %[acc], vcc, %[a{i}], %[b{K - i}], %[acc]\\n\\t"' for i in idx]
    ins = [f'[a{i}] "v"(a[{i}])' for i in idx] + [f'[b{K - i}] "v"(b[{K - i}])' for i in idx]
    return (f"template <> BLZ_DEV void rr_ab<{NL}, {K}>(uint64_t& acc, const uint32_t (&a)[{NL}], const uint32_t (&b)[{NL}]) {{\n"
            f"    asm({chr(10).join('        ' + l for l in lines).lstrip()}\n"
            f"        : [acc] \"+v\"(acc)\n        : {', '.join(ins)}\n        : \"vcc\");\n}}\n")


def emit_sq(NL, K):
    ilo = 0 if K < NL else K - NL + 1
    ihi = K if K < NL else NL - 1
    pairs = [(i, K - i) for i in range(ilo, ihi + 1) if i < K - i]
    lines = [f'"v_mad_u64_u32 %[acc], vcc, %[a{i}], %[d{j}], %[acc]\\n\\t"' for i, j in pairs]
    ins = [f'[a{i}] "v"(a[{i}])' for i, _ in pairs] + [f'[d{j}] "v"(a2[{j}])' for _, j in pairs]
    if K % 2 == 0:
        h = K // 2
        lines.append(f'"v_mad_u64_u32 %[acc], vcc, %[h], %[h], %[acc]\\n\\t"')
        ins.append(f'[h] "v"(a[{h}])')
    return (f"template <> BLZ_DEV void rr_sq<{NL}, {K}>(uint64_t& acc, const uint32_t (&a)[{NL}], const uint32_t (&a2)[{NL}]) {{\n"
            f"    asm({chr(10).join('        ' + l for l in lines).lstrip()}\n"
            f"        : [acc] \"+v\"(acc)\n        : {', '.join(ins)}\n        : \"vcc\");\n}}\n")


def emit_qm(NL, K):
    if K < NL:
        idx = list(range(0, K))          # q_K m_0 follows once q_K exists
    else:
        idx = list(range(K - NL + 1, NL))
    if not idx:
        return (f"template <> BLZ_DEV void rr_qm<{NL}, {K}>(uint64_t& acc, const uint32_t (&q)[{NL}], const uint32_t (&m)[{NL}]) {{\n"
                f"    (void)acc; (void)q; (void)m;\n}}\n")
    lines = [f'"v_mad_u64_u32 %[acc], vcc, %[q{i}], %[m{K - i}], %[acc]\\n\\t"' for i in idx]
    ins = [f'[q{i}] "v"(q[{i}])' for i in idx] + [f'[m{K - i}] "s"(m[{K - i}])' for i in idx]
    return (f"template <> BLZ_DEV void rr_qm<{NL}, {K}>(uint64_t& acc, const uint32_t (&q)[{NL}], const uint32_t (&m)[{NL}]) {{\n"
            f"    asm({chr(10).join('        ' + l for l in lines).lstrip()}\n"
            f"        : [acc] \"+v\"(acc)\n        : {', '.join(ins)}\n        : \"vcc\");\n}}\n")


def emit_as(NL, K):
    """full column of a (VGPRs) times a wave-uniform constant s (SGPR operands): the q * (2^(B NL) - m) half of the
    Shoup product (field_rr.hip.hpp rr_mul_shoup)"""
    ilo = 0 if K < NL else K - NL + 1
    ihi = K if K < NL else NL - 1
    idx = list(range(ilo, ihi + 1))
    lines = [f'"v_mad_u64_u32 %[acc], vcc, %[a{i}], %[s{K - i}], %[acc]\\n\\t"' for i in idx]
    ins = [f'[a{i}] "v"(a[{i}])' for i in idx] + [f'[s{K - i}] "s"(s[{K - i}])' for i in idx]
    return (f"template <> BLZ_DEV void rr_as<{NL}, {K}>(uint64_t& acc, const uint32_t (&a)[{NL}], const uint32_t (&s)[{NL}]) {{\n"
            f"    asm({chr(10).join('        ' + l for l in lines).lstrip()}\n"
            f"        : [acc] \"+v\"(acc)\n        : {', '.join(ins)}\n        : \"vcc\");\n}}\n")


def emit_lo2(NL, K, uniform):
    """Shoup product, low half, column K < NL in ONE statement: a (VGPRs) times b (VGPRs; `uniform`: SGPRs) plus q (VGPRs)
    times the constant s (SGPRs).  4 (K + 1) + 1 operands: K <= 6."""
    idx = list(range(0, K + 1))
    lines = [f'"v_mad_u64_u32 %[acc], vcc, %[a{i}], %[b{K - i}], %[acc]\\n\\t"' for i in idx]
    lines += [f'"v_mad_u64_u32 %[acc], vcc, %[q{i}], %[s{K - i}], %[acc]\\n\\t"' for i in idx]
    bc = "s" if uniform else "v"
    ins = ([f'[a{i}] "v"(a[{i}])' for i in idx] + [f'[b{K - i}] "{bc}"(b[{K - i}])' for i in idx] +
           [f'[q{i}] "v"(q[{i}])' for i in idx] + [f'[s{K - i}] "s"(s[{K - i}])' for i in idx])
    name = "rr_asqs" if uniform else "rr_abqs"
    return (f"template <> BLZ_DEV void {name}<{NL}, {K}>(uint64_t& acc, const uint32_t (&a)[{NL}], const uint32_t (&b)[{NL}], const uint32_t (&q)[{NL}], const uint32_t (&s)[{NL}]) {{\n"
            f"    asm({chr(10).join('        ' + l for l in lines).lstrip()}\n"
            f"        : [acc] \"+v\"(acc)\n        : {', '.join(ins)}\n        : \"vcc\");\n}}\n")


def fused_ok(NL, K):
    """Can the caller's products and the reduction products of column K share one asm statement (30 operands)?"""
    n_ab = K + 1 if K < NL else 2 * NL - 1 - K
    n_qm = K if K < NL else 2 * NL - 1 - K
    return n_qm > 0 and 2 * n_ab + 2 * n_qm + 1 <= 30


def emit_abqm(NL, K):
    ilo = 0 if K < NL else K - NL + 1
    ihi = K if K < NL else NL - 1
    ab = list(range(ilo, ihi + 1))
    qm = list(range(0, K)) if K < NL else list(range(K - NL + 1, NL))
    lines = [f'"v_mad_u64_u32 %[acc], vcc, %[a{i}], %[b{K - i}], %[acc]\\n\\t"' for i in ab]
    lines += [f'"v_mad_u64_u32 %[acc], vcc, %[q{i}], %[m{K - i}], %[acc]\\n\\t"' for i in qm]
    ins = ([f'[a{i}] "v"(a[{i}])' for i in ab] + [f'[b{K - i}] "v"(b[{K - i}])' for i in ab] +
           [f'[q{i}] "v"(q[{i}])' for i in qm] + [f'[m{K - i}] "s"(m[{K - i}])' for i in qm])
    return (f"template <> BLZ_DEV void rr_abqm<{NL}, {K}>(uint64_t& acc, const uint32_t (&a)[{NL}], const uint32_t (&b)[{NL}], "
            f"const uint32_t (&q)[{NL}], const uint32_t (&m)[{NL}]) {{\n"
            f"    asm({chr(10).join('        ' + l for l in lines).lstrip()}\n"
            f"        : [acc] \"+v\"(acc)\n        : {', '.join(ins)}\n        : \"vcc\");\n}}\n")


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "..", "blaze_amd", "csrc", "rr_gen.inc")
    out = ["// GENERATED by tools/gen_rr_asm.py - do not edit.  Included by field_rr.hip.hpp inside namespace blz.\n",
           "template <int NL, int K> BLZ_DEV void rr_ab(uint64_t& acc, const uint32_t (&a)[NL], const uint32_t (&b)[NL]);",
           "template <int NL, int K> BLZ_DEV void rr_sq(uint64_t& acc, const uint32_t (&a)[NL], const uint32_t (&a2)[NL]);",
           "template <int NL, int K> BLZ_DEV void rr_qm(uint64_t& acc, const uint32_t (&q)[NL], const uint32_t (&m)[NL]);",
           "// rr_ab and rr_qm of one column in ONE statement, where its 30 operands allow (hipcc pads every statement whose",
           "// result is read next with an s_nop: one per column instead of two)",
           "constexpr bool rr_fused_ok(int NL, int K) {",
           "    const int n_ab = K < NL ? K + 1 : 2 * NL - 1 - K, n_qm = K < NL ? K : 2 * NL - 1 - K;",
           "    return n_qm > 0 && 2 * n_ab + 2 * n_qm + 1 <= 30;",
           "}",
           "template <int NL, int K> BLZ_DEV void rr_abqm(uint64_t& acc, const uint32_t (&a)[NL], const uint32_t (&b)[NL], const uint32_t (&q)[NL], const uint32_t (&m)[NL]);",
           "// full column of a times a wave-uniform constant (the low half of the Shoup product)",
           "template <int NL, int K> BLZ_DEV void rr_as(uint64_t& acc, const uint32_t (&a)[NL], const uint32_t (&s)[NL]);",
           "// a low column of the Shoup product in one statement (K <= 6: 30 operands): a b + q s, b in VGPRs / in SGPRs",
           "constexpr bool rr_lo2_ok(int K) { return 4 * (K + 1) + 1 <= 30; }",
           "template <int NL, int K> BLZ_DEV void rr_abqs(uint64_t& acc, const uint32_t (&a)[NL], const uint32_t (&b)[NL], const uint32_t (&q)[NL], const uint32_t (&s)[NL]);",
           "template <int NL, int K> BLZ_DEV void rr_asqs(uint64_t& acc, const uint32_t (&a)[NL], const uint32_t (&b)[NL], const uint32_t (&q)[NL], const uint32_t (&s)[NL]);\n"]
    for NL in (9, 14):
        for K in range(2 * NL - 1):
            out.append(emit_ab(NL, K))
            out.append(emit_sq(NL, K))
            out.append(emit_qm(NL, K))
            if fused_ok(NL, K):
                out.append(emit_abqm(NL, K))
            if NL == 9 and K < NL and 4 * (K + 1) + 1 <= 30:
                out.append(emit_lo2(NL, K, False))
                out.append(emit_lo2(NL, K, True))
            if NL == 9:   # K < NL: the q * (R - m) half of the Shoup product; every K: products by a wave-uniform twiddle
                out.append(emit_as(NL, K))
    with open(dst, "w") as f:
        f.write("\n".join(out))
    print("wrote", os.path.normpath(dst))


if __name__ == "__main__":
    main()
