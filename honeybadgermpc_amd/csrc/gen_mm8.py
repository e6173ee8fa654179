#!/usr/bin/env python3
"""Emits hb_mm8_body.inc: the MFMA phases of k_mm8 (hb_mfma.hip) as inline-asm blocks.

Column c of an output is an int8 dot product between matrix digits (A operand) and a window of input bytes
(B operand): col_c = sum_l sum_b M_b[l] X_{c-b}[l].  One v_mfma_i32_16x16x64_i8 contracts 64 products per
(row, chunk); a lane's 16 B-operand bytes are cut as

    8 TERMS x 8 DIGITS per K-block:  lane (n, g) feeds terms t0 = 8 kb + 2 g and t1 = t0 + 1 of chunk n, the 16 digits of
    an entry are two groups G (b in [8 G, 8 G + 8)), and with s = c - 7 - 8 G = 4 q + rho the operand is the 8-byte
    windows [s, s + 7] of both elements: dwords q, q + 1 of each element shifted right by rho bytes.

Why 8 x 8 and not 4 terms x 16 digits (the first version of this kernel): a 16-byte window per element slides over 47
positions per term block (47 MFMAs per 4 terms = 11.75 per term, 68 % of the K slots useful); 8-byte windows take
39 positions x 2 groups per 8 terms = 9.75 per term -- 17 % fewer MFMAs -- and, more important for a kernel bound by
VALU issue, the operand files shrink to one: with the register file laid out [dword k][element e] the operand of window q
is registers 2 q .. 2 q + 3, ALWAYS even-aligned, so the second copy of the file one register apart (9 v_mov per group)
is gone, and a lane prepares two elements per group for eight terms instead of one for four: 270 preparation ops per wave
pass instead of 520.

The columns are produced in two halves BY SHIFT: half 0 = rho in {0, 1} (c = 3, 0 mod 4; 23 columns), half 1 = rho in
{2, 3} (c = 1, 2 mod 4; 24).  Within a half each (K-block, rho) group covers every window of that shift, so its shifted
dwords SH_rho[k], k = -1 .. 7, are computed once per element:
    F[k][e] = X_e[k] ^ 0x80808080                       (rho = 0: also the copy out of the LDS prefetch registers)
    F[k][e] = alignbyte(F[k+1][e], F[k][e], 2)          (rho = 2: in place on that copy)
    F'[k][e] = alignbyte(F[k+1][e], F[k][e], 1)         (rho 0 -> 1, 2 -> 3: into the OTHER of two file sets, built while the
                                                          MFMAs of the current group still read this one)
Registers v180..v255 are reserved (clobbers): XB (LDS prefetch of the next K-block's two elements), two pairs of A-operand
buffers, two file sets.  The accumulators are ordinary "=&v" outputs.

Hazards handled here (nothing inside an asm string is padded by the compiler): VALU write -> MFMA operand needs 2 wait
states (every preparation op of a group precedes the last MFMA of the previous one, plus s_nop 0); a file set is only
rewritten after every MFMA that reads it has been issued; MFMA result -> VALU read after the block (s_nop 7 x2).
"""
import os

NC = 47
MAXKB = 4                   # K-blocks of 8 terms: d <= 32
XB = 180                    # 16 dwords: element e dword k at XB + 8 e + k
ABUF = [[196, 200], [204, 208]]   # [K-block parity][digit group], 4 dwords each
F_SETS = [212, 234]         # 22 registers each: dword k = -2 .. 8 of element e at base + 2 (k + 2) + e
F_KMIN = -2
CLOBBER_LO, CLOBBER_HI = 180, 255
HALF_RHOS = [(0, 1), (2, 3)]


def f(s, k, e):
    assert -2 <= k <= 8 and e in (0, 1)
    return F_SETS[s] + 2 * (k - F_KMIN) + e


def windows(rho):
    """dword offsets q of the windows s = 4 q + rho in [-7, 31]"""
    return [q for q in range(-2, 8) if -7 <= 4 * q + rho <= 31]


def cols_of(rho):
    return sorted({4 * q + rho + 7 + 8 * g for q in windows(rho) for g in (0, 1)})


def half_columns(half):
    return sorted(c for rho in HALF_RHOS[half] for c in cols_of(rho))


def lds_loads(kb, xs_op, as_op):
    """K-block kb: both elements of the lane (2 x 32 bytes) and both digit groups"""
    a0, a1 = ABUF[kb & 1]
    out = []
    for e in (0, 1):
        for h in (0, 1):
            out.append(f"ds_read_b128 v[{XB + 8 * e + 4 * h}:{XB + 8 * e + 4 * h + 3}], {xs_op} offset:{((kb * 2 + e) * 2 + h) * 1024}")
    out.append(f"ds_read_b128 v[{a0}:{a0 + 3}], {as_op} offset:{(kb * 2) * 1024}")
    out.append(f"ds_read_b128 v[{a1}:{a1 + 3}], {as_op} offset:{(kb * 2 + 1) * 1024}")
    return out


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


def asm_half(half, nkb, skip=False):
    """skip: the matrix has no digit above the eighth in its first eight terms (Vandermonde matrices at small points: x^l < 2^56
    for l < 8), so the MFMAs of digit group 1 in K-block 0 would add zeros -- they are left out (39 of 78 n_kb), and the columns only
    group 1 reaches start from the bias in K-block 1 instead"""
    assert not skip or nkb >= 2
    cols = half_columns(half)
    pos = {c: i for i, c in enumerate(cols)}
    ncol = len(cols)
    xs_op, as_op, bias = f"%{ncol}", f"%{ncol + 1}", f"%{ncol + 2}"
    L = []
    # positions that are read but never written stay zero: k = -2 and k = 8
    for s in range(2):
        for k in (-2, 8):
            for e in (0, 1):
                L.append(f"v_mov_b32 v{f(s, k, e)}, 0")
    L += lds_loads(0, xs_op, as_op)[:(5 if skip else 6)]          # (skip: group 1's digits of K-block 0 are not needed)
    groups = [(kb, rho) for kb in range(nkb) for rho in HALF_RHOS[half]]

    def prep(gi):
        """fill file set gi & 1 for group gi"""
        kb, rho = groups[gi]
        s = gi & 1
        ops = []
        if rho == HALF_RHOS[half][0]:
            ops.append("s_waitcnt lgkmcnt(0)")
            for e in (0, 1):
                for k in range(8):
                    ops.append(f"v_xor_b32 v{f(s, k, e)}, 0x80808080, v{XB + 8 * e + k}")
                ops.append(f"v_mov_b32 v{f(s, -1, e)}, 0")
            if rho != 0:                                    # half 1 starts at shift 2: in place, nobody reads this set yet
                for e in (0, 1):
                    for k in range(-1, 8):
                        ops.append(f"v_alignbyte_b32 v{f(s, k, e)}, v{f(s, k + 1, e)}, v{f(s, k, e)}, {rho}")
        else:
            for e in (0, 1):
                for k in range(-1, 8):
                    ops.append(f"v_alignbyte_b32 v{f(s, k, e)}, v{f(1 - s, k + 1, e)}, v{f(1 - s, k, e)}, 1")
        return ops

    started = set()

    def mfmas(gi):
        kb, rho = groups[gi]
        s = gi & 1
        out = []
        for q in windows(rho):
            r = f(s, q, 0)
            assert r % 2 == 0
            for grp in (0, 1):
                if skip and kb == 0 and grp == 1:
                    continue
                c = 4 * q + rho + 7 + 8 * grp
                ab = ABUF[kb & 1][grp]
                cin = f"%{pos[c]}"
                if c not in started:
                    assert kb == 0 or (skip and kb == 1)
                    cin = bias
                    started.add(c)
                out.append(f"v_mfma_i32_16x16x64_i8 %{pos[c]}, v[{ab}:{ab + 3}], v[{r}:{r + 3}], {cin}")
        return out

    L += prep(0) + ["s_nop 1"]
    for gi in range(len(groups)):
        nxt = gi + 1 < len(groups)
        kb, rho = groups[gi]
        if rho == HALF_RHOS[half][0] and kb + 1 < nkb:
            # XB was consumed by this group's preparation and every MFMA of K-block kb - 1 (the last reader of the other A
            # buffers) has been issued: fetch K-block kb + 1
            L += lds_loads(kb + 1, xs_op, as_op)
        L += interleave(mfmas(gi), prep(gi + 1) if nxt else [])
        if nxt:
            L.append("s_nop 0")
    assert started == set(cols)
    L += ["s_nop 7", "s_nop 7"]
    return L, ncol


def emit():
    out = ["// GENERATED by gen_mm8.py -- do not edit", ""]
    out.append("// column held by accumulator i of each half (half 0: rho in {0,1}; half 1: rho in {2,3})")
    for half in range(2):
        cols = half_columns(half)
        out.append(f"// half {half}: {cols}")
    assert half_columns(0) == [c for c in range(NC) if c % 4 in (0, 3)] and half_columns(1) == [c for c in range(NC) if c % 4 in (1, 2)]
    out.append("template <int NKB, int HALF, bool SKIP = false> struct Mm8Phase;")
    clob = ", ".join(f'"v{r}"' for r in range(CLOBBER_LO, CLOBBER_HI + 1))
    for nkb, skip in [(k, False) for k in range(1, MAXKB + 1)] + [(k, True) for k in range(2, MAXKB + 1)]:
        for half in range(2):
            lines, ncol = asm_half(half, nkb, skip)
            out.append(f"template <> struct Mm8Phase<{nkb}, {half}, {'true' if skip else 'false'}> {{")
            out.append("    static __device__ __forceinline__ void run(v4i (&acc)[24], uint32_t xs_addr, uint32_t as_addr, v4i biasv) {")
            out.append("        asm volatile(")
            for ln in lines:
                out.append(f'            "{ln}\\n\\t"')
            outs = ", ".join(f'"=&v"(acc[{i}])' for i in range(ncol))
            out.append(f"            : {outs}")
            out.append('            : "v"(xs_addr), "v"(as_addr), "v"(biasv)')
            out.append(f'            : {clob}, "memory");')
            out.append("    }")
            out.append("};")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    text = emit()
    path = os.path.join(here, "hb_mm8_body.inc")
    old = open(path).read() if os.path.exists(path) else None
    if old != text:
        open(path, "w").write(text)
    print(path)
