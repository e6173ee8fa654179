#!/usr/bin/env python3
"""Emits hb_mm8_body.inc: the MFMA phases of k_mm8 (hb_mfma.hip) as inline-asm blocks.

Column c of an output is an int8 dot product between the matrix digits (A operand, 16 digits of 4
terms per K-block) and bytes [c-15, c] of every input element (B operand).  With c - 15 = 4q + rho the
B operand is dwords q .. q+3 of the element shifted right by rho bytes.

The columns are produced in two halves BY SHIFT: half 0 = the columns with rho in {0, 1} (c = 3, 0 mod
4; 23 of them), half 1 = rho in {2, 3} (c = 1, 2 mod 4; 24).  Within a half each (term block kb, rho)
group covers the whole element (q = -4 .. 7), so its nine shifted dwords SH_rho[k], k = -1 .. 7, are
computed exactly once, and consecutive shifts are derived from each other IN PLACE:
    EA[k] = X[k] ^ 0x80808080                 (rho = 0, or the base of half 1; doubles as the copy out of
                                               the LDS prefetch buffer)
    EA'[k] = alignbyte(EA[k+1], EA[k], s)     s = 1 (rho 0 -> 1, rho 2 -> 3) or 2 (base -> rho 2)
(the next shift is written into the OTHER of two EA sets, so that it can be built while the MFMAs of
the current group are still reading this one).  A 128-bit MFMA operand must start on an even VGPR, so
the file exists twice, one register apart: EA (k = -4 .. 9) serves the even q, EB (k = -3 .. 10) the odd
q; EB is refreshed by 9 v_mov per group.  Every window is then an aligned sub-range of a file and no
operand is ever assembled per MFMA -- which is what hipcc could not be talked into (3-4 v_mov per
MFMA), hence the asm.

Schedule of one group: its odd-q MFMAs (EB) interleaved with building the next group's EA set, then its
even-q MFMAs (EA) interleaved with copying that set into EB (free by then).  Registers v198..v255 are
reserved for this (clobbers): XB (LDS prefetch of the next term block's element), two A-operand
buffers, two EA sets, EB.  The accumulators are ordinary "=&v" outputs.

Hazards handled here (nothing inside an asm string is padded by the compiler): VALU write -> MFMA
operand needs 2 wait states (every preparation op of a group precedes the last MFMA of the previous
one, plus s_nop 0); a file is only rewritten after every MFMA that reads it has been issued, the
registers rewritten first being the ones read by the earliest of those MFMAs; MFMA result -> VALU read
after the block (s_nop 7 x2 at the end).
"""
import os

NC = 47
MAXKB = 8
XB = 198                   # 8 dwords: LDS prefetch of the next term block's element
ABUF = [206, 210]          # 4 dwords each
EA_SETS = [214, 228]       # 14 registers each, k = -4 .. 9; alternate between consecutive groups
EA_KMIN = -4
EB0, EB_KMIN = 242, -3     # 14 registers, k = -3 .. 10
CLOBBER_LO, CLOBBER_HI = 198, 255
HALF_RHOS = [(0, 1), (2, 3)]


def ea(s, k):
    assert -4 <= k <= 9
    return EA_SETS[s] + k - EA_KMIN


def eb(k):
    assert -3 <= k <= 10
    return EB0 + k - EB_KMIN


def cols_of(rho):
    return [c for c in range(NC) if (c - 15) % 4 == rho]


def half_columns(half):
    return sorted(c for rho in HALF_RHOS[half] for c in cols_of(rho))


def lds_loads(kb, xs_op, as_op):
    ab = ABUF[kb & 1]
    return [
        f"ds_read_b128 v[{XB}:{XB + 3}], {xs_op} offset:{(2 * kb) * 1024}",
        f"ds_read_b128 v[{XB + 4}:{XB + 7}], {xs_op} offset:{(2 * kb + 1) * 1024}",
        f"ds_read_b128 v[{ab}:{ab + 3}], {as_op} offset:{kb * 1024}",
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


def asm_half(half, nkb):
    cols = half_columns(half)
    pos = {c: i for i, c in enumerate(cols)}
    ncol = len(cols)
    xs_op, as_op, bias = f"%{ncol}", f"%{ncol + 1}", f"%{ncol + 2}"
    L = []
    # positions that are read but never written stay zero: k <= -2 and k >= 8
    for s in range(2):
        for k in (-4, -3, -2, 8, 9):
            L.append(f"v_mov_b32 v{ea(s, k)}, 0")
    for k in (-3, -2, 8, 9, 10):
        L.append(f"v_mov_b32 v{eb(k)}, 0")
    L += lds_loads(0, xs_op, as_op)
    groups = [(kb, rho) for kb in range(nkb) for rho in HALF_RHOS[half]]

    def prep_a(gi):
        """fill EA set gi & 1 for group gi (reads the other set, or XB for the first shift of a term block)"""
        kb, rho = groups[gi]
        s = gi & 1
        ops = []
        first = rho == HALF_RHOS[half][0]
        if first:
            ops.append("s_waitcnt lgkmcnt(0)")
            for k in range(8):
                ops.append(f"v_xor_b32 v{ea(s, k)}, 0x80808080, v{XB + k}")
            ops.append(f"v_mov_b32 v{ea(s, -1)}, 0")
            if rho != 0:                                    # half 1 starts at shift 2: in place, nobody reads this set yet
                for k in range(-1, 8):
                    ops.append(f"v_alignbyte_b32 v{ea(s, k)}, v{ea(s, k + 1)}, v{ea(s, k)}, {rho}")
        else:
            prev = HALF_RHOS[half][0]
            for k in range(-1, 8):
                hi = f"v{ea(1 - s, k + 1)}"
                ops.append(f"v_alignbyte_b32 v{ea(s, k)}, {hi}, v{ea(1 - s, k)}, {rho - prev}")
        return ops

    def prep_b(gi):
        s = gi & 1
        return [f"v_mov_b32 v{eb(k)}, v{ea(s, k)}" for k in range(-1, 8)]

    def mfmas(gi, parity):
        kb, rho = groups[gi]
        s, ab = gi & 1, ABUF[kb & 1]
        out = []
        for c in cols_of(rho):
            q = (c - 15 - rho) // 4
            if q % 2 != parity:
                continue
            r = ea(s, q) if parity == 0 else eb(q)
            assert r % 2 == 0
            cin = bias if kb == 0 else f"%{pos[c]}"
            out.append(f"v_mfma_i32_16x16x64_i8 %{pos[c]}, v[{ab}:{ab + 3}], v[{r}:{r + 3}], {cin}")
        return out

    # group 0 has nothing to hide behind
    L += prep_a(0) + prep_b(0) + ["s_nop 1"]
    for gi in range(len(groups)):
        nxt = gi + 1 < len(groups)
        kb, rho = groups[gi]
        if rho == HALF_RHOS[half][0] and kb + 1 < nkb:
            # XB was consumed by this group's preparation and every MFMA of term block kb - 1 (the last reader of
            # the other A buffer) has been issued: fetch term block kb + 1
            L += lds_loads(kb + 1, xs_op, as_op)
        # odd windows (EB) first; meanwhile the next group's EA set is built from this one's
        L += interleave(mfmas(gi, 1), prep_a(gi + 1) if nxt else [])
        # then the even windows (EA); EB is free now and is refreshed for the next group
        L += interleave(mfmas(gi, 0), prep_b(gi + 1) if nxt else [])
        if nxt:
            L.append("s_nop 0")
    L += ["s_nop 7", "s_nop 7"]
    return L, ncol


def emit():
    out = ["// GENERATED by gen_mm8.py -- do not edit", ""]
    out.append("// column held by accumulator i of each half (half 0: rho in {0,1}; half 1: rho in {2,3})")
    for half in range(2):
        cols = half_columns(half)
        out.append(f"// half {half}: {cols}")
    out.append("template <int NKB, int HALF> struct Mm8Phase;")
    clob = ", ".join(f'"v{r}"' for r in range(CLOBBER_LO, CLOBBER_HI + 1))
    for nkb in range(1, MAXKB + 1):
        for half in range(2):
            lines, ncol = asm_half(half, nkb)
            out.append(f"template <> struct Mm8Phase<{nkb}, {half}> {{")
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
