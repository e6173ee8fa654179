#!/usr/bin/env python3
"""Emits hb_mm8w_body.inc: the MFMA phase of k_mm8w (hb_mfma_wide.hip) as one inline-asm block.

k_mm8w is the matrix-core mat-vec for FULL-SIZE matrix entries (any residue mod p: inverse Vandermonde
matrices at omega-power points, Vandermonde matrices whose powers outgrow 2^127, arbitrary hb_matrix
operands).  Entries are cut into 32 base-256 digits M_b, inputs into their 32 bytes X_a:

    S = sum_l M[l] x[l] = sum_c 2^(8c) col_c,   col_c = sum_l sum_b M_b[l] X_{c-b}[l],   c < 63.

One v_mfma_i32_16x16x64_i8 contracts 4 terms x 16 digits, so a column needs two digit groups G (b in
[16G, 16G + 16)) per term block: with s = c - 15 - 16G the B operand is bytes [s, s + 15] of the element,
i.e. dwords q .. q+3 of the element shifted right by rho bytes (s = 4q + rho).  Every window s in
[-15, 31] therefore feeds TWO MFMAs (group 0 -> column s + 15, group 1 -> column s + 31): 94 per term
block, against 18 preparation ops per (term block, rho) -- the phase is matrix-pipe bound, which is the
point: the int8 pipe does the 1024 byte products of a 256 x 256-bit multiplication in 16 cycles per 16 x 16
outputs, the VALU needs 81 half-rate v_mad_u64_u32 per lane.

Register files and the schedule are those of gen_mm8.py (two EA sets for even q, EB one register apart for
odd q, the next group's shifts built while the current group's MFMAs issue), with four shifts rho = 0..3 per
term block instead of two per half, all 63 accumulators live (AGPR operands of the asm statement: the kernel
runs one wave per SIMD with the 512-register budget), the matrix digits streamed from L2 (two dwordx4 per lane
per term block, one block ahead) and a LOOP over pairs of term blocks, so that the code does not grow with
the inner dimension.  The first pair is peeled: its first touch of every column starts from the inline
constant 0.

Operands: %0..%62 accumulators; %63 LDS byte address of the lane's element slot (advanced here), %64 per-lane
byte offset into the digit image (advanced here), %65 loop count = nkb / 2 - 1 (consumed), %66 digit image
base of this row tile (SGPR pair).
"""
import os

NC = 63
XB = 190                       # 8 dwords: LDS prefetch of the next term block's element
ABUF = [[198, 202], [206, 210]]  # [term block parity][digit group]: 4 dwords each
EA_SETS = [214, 228]           # 14 registers each, k = -4 .. 9
EA_KMIN = -4
EB0, EB_KMIN = 242, -3         # 14 registers, k = -3 .. 10
CLOBBER_LO, CLOBBER_HI = 190, 255
RHOS = (0, 1, 2, 3)
OP_XA, OP_VA, OP_CNT, OP_SB = "%63", "%64", "%65", "%66"


def ea(s, k):
    assert -4 <= k <= 9
    return EA_SETS[s] + k - EA_KMIN


def eb(k):
    assert -3 <= k <= 10
    return EB0 + k - EB_KMIN


def windows(rho):
    return [q for q in range(-4, 8) if -15 <= 4 * q + rho <= 31]


def loads(par):
    """element and digits of the NEXT term block -> XB, ABUF[par]; both cursors move on by one block"""
    a0, a1 = ABUF[par]
    return [
        f"ds_read_b128 v[{XB}:{XB + 3}], {OP_XA}",
        f"ds_read_b128 v[{XB + 4}:{XB + 7}], {OP_XA} offset:1024",
        f"v_add_u32 {OP_XA}, 0x800, {OP_XA}",
        f"global_load_dwordx4 v[{a0}:{a0 + 3}], {OP_VA}, {OP_SB}",
        f"global_load_dwordx4 v[{a1}:{a1 + 3}], {OP_VA}, {OP_SB} offset:1024",
        f"v_add_u32 {OP_VA}, 0x800, {OP_VA}",
    ]


def interleave(mfmas, ops):
    """one MFMA, then a share of ops; every op ends up before the last MFMA"""
    out = []
    if not mfmas:
        return list(ops)
    n = len(mfmas)
    per = (len(ops) + max(n - 1, 1) - 1) // max(n - 1, 1)
    pi = 0
    for i, m in enumerate(mfmas):
        if i == n - 1:
            out += ops[pi:]
            pi = len(ops)
        out.append(m)
        if i < n - 1:
            out += ops[pi:pi + per]
            pi += per
    return out


def prep_a(gi):
    """fill EA set gi & 1 for group gi = (term block, rho): from XB for rho = 0, else one more byte of shift"""
    rho = RHOS[gi % 4]
    s = gi & 1
    ops = []
    if rho == 0:
        ops.append("s_waitcnt lgkmcnt(0)")
        for k in range(8):
            ops.append(f"v_xor_b32 v{ea(s, k)}, 0x80808080, v{XB + k}")
        ops.append(f"v_mov_b32 v{ea(s, -1)}, 0")
    else:
        for k in range(-1, 8):
            ops.append(f"v_alignbyte_b32 v{ea(s, k)}, v{ea(1 - s, k + 1)}, v{ea(1 - s, k)}, 1")
    return ops


def prep_b(gi):
    s = gi & 1
    return [f"v_mov_b32 v{eb(k)}, v{ea(s, k)}" for k in range(-1, 8)]


def mfmas(gi, parity, par, seen):
    """MFMAs of group gi whose window starts on an even (parity 0: EA) / odd (1: EB) dword; `seen` = columns already
    started (None: accumulate always)"""
    rho = RHOS[gi % 4]
    s = gi & 1
    out = []
    for q in windows(rho):
        if q % 2 != parity:
            continue
        r = ea(s, q) if parity == 0 else eb(q)
        assert r % 2 == 0
        for grp in (0, 1):
            c = 4 * q + rho + 15 + 16 * grp
            assert 0 <= c < NC
            ab = ABUF[par][grp]
            cin = f"%{c}"
            if seen is not None and c not in seen:
                cin = "0"
                seen.add(c)
            out.append(f"v_mfma_i32_16x16x64_i8 %{c}, v[{ab}:{ab + 3}], v[{r}:{r + 3}], {cin}")
    return out


def pair(first):
    """two term blocks (8 groups): digits of the even one in ABUF[0], of the odd one in ABUF[1].  On entry the first
    group's EA / EB files are ready, XB has been consumed and the even block's digits are in flight.
    The odd block of the LAST pair must not prefetch: nothing waits for a load issued there, and one that lands after the
    asm statement would overwrite registers the compiler has taken back (it did, once per ~10^4 launches, when L2 was cold)."""
    L = []
    seen = set() if first else None
    tag = "p" if first else "l"
    for gi in range(8):
        par = gi // 4
        if gi % 4 == 0:
            # the digits of this block were requested one block ago; every MFMA reading the other buffer has been issued
            L.append("s_waitcnt vmcnt(0)")
            if gi == 4:
                # remaining pairs after this one: the counter itself in the peeled pair, counter - 1 in the loop body
                L += [f"s_cmp_eq_u32 {OP_CNT}, {0 if first else 1}", f"s_cbranch_scc1 .Lmm8w_nopf_{tag}_%="]
            L += loads(1 - par)
            if gi == 4:
                L.append(f".Lmm8w_nopf_{tag}_%=:")
        L += interleave(mfmas(gi, 1, par, seen), prep_a(gi + 1))
        L += interleave(mfmas(gi, 0, par, seen), prep_b(gi + 1))
        L.append("s_nop 0")
    if first:
        assert seen == set(range(NC))
    return L


def asm_lines():
    L = []
    # positions that are read but never written stay zero: k <= -2 and k >= 8
    for s in range(2):
        for k in (-4, -3, -2, 8, 9):
            L.append(f"v_mov_b32 v{ea(s, k)}, 0")
    for k in (-3, -2, 8, 9, 10):
        L.append(f"v_mov_b32 v{eb(k)}, 0")
    L += loads(0)                                   # term block 0
    L += prep_a(0) + prep_b(0) + ["s_nop 1"]
    L += pair(True)
    L += [f"s_cmp_eq_u32 {OP_CNT}, 0", "s_cbranch_scc1 .Lmm8w_end_%="]
    L.append(".Lmm8w_loop_%=:")
    L += pair(False)
    L += [f"s_sub_u32 {OP_CNT}, {OP_CNT}, 1", f"s_cmp_lg_u32 {OP_CNT}, 0", "s_cbranch_scc1 .Lmm8w_loop_%="]
    L.append(".Lmm8w_end_%=:")
    L += ["s_nop 7", "s_nop 7"]
    return L


def emit():
    out = ["// GENERATED by gen_mm8w.py -- do not edit", ""]
    clob = ", ".join(f'"v{r}"' for r in range(CLOBBER_LO, CLOBBER_HI + 1))
    lines = asm_lines()
    out.append("static __device__ __forceinline__ void mm8w_phase(v4i (&acc)[63], uint32_t &xa, uint32_t &va, uint32_t &cnt, uint64_t abase) {")
    out.append("    asm volatile(")
    for ln in lines:
        out.append(f'        "{ln}\\n\\t"')
    outs = ", ".join(f'"=&a"(acc[{i}])' for i in range(NC))
    out.append(f"        : {outs}, \"+v\"(xa), \"+v\"(va), \"+s\"(cnt)")
    out.append('        : "s"(abase)')
    out.append(f'        : {clob}, "scc", "memory");')
    out.append("}")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    text = emit()
    path = os.path.join(here, "hb_mm8w_body.inc")
    old = open(path).read() if os.path.exists(path) else None
    if old != text:
        open(path, "w").write(text)
    print(path)
