#!/usr/bin/env python3
"""Emits hb_mm8_body.inc: the MFMA phases of k_mm8 (hb_mfma.hip) as inline-asm blocks.

Column c of an output is an int8 dot product between the matrix digits (A operand, 16 digits of 4
terms per K-block) and bytes [c-15, c] of every input element (B operand).  With c - 15 = 4q + rho the
B operand is dwords q .. q+3 of the element shifted right by rho bytes.  A 128-bit MFMA operand
must start on an even VGPR, so the shifted dwords SH_rho[k] = alignbyte(X[k+1], X[k], rho) of one
(term block kb, shift rho) group are written twice, into two small register files one register
apart: EA serves the even q, EB the odd q.  Every window is then an aligned sub-range of a file and
no operand is ever copied -- which is what hipcc could not be talked into (it assembles each B
operand with 3-4 v_mov), hence the asm.

Registers v196..v255 are reserved for this (clobbers): two input-element buffers, two A-operand
buffers (LDS reads for term block kb+1 are issued while kb computes) and two E-file sets that
alternate between consecutive groups, so that the v_alignbyte of group g+1 interleave with the MFMAs
of group g.  The accumulators are ordinary "=&v" outputs placed by the compiler.

Hazards handled here (nothing inside an asm string is padded by the compiler): VALU write -> MFMA
operand needs 2 wait states (the last write of a group is always followed by an MFMA of the previous
group plus s_nop 0); MFMA result -> VALU read after the block (s_nop 7 x2 at the end).
"""
import os

NC = 47
MAXKB = 8
HALVES = [(0, 24), (24, NC)]
XBUF = [196, 204]          # 8 dwords each
ABUF = [212, 216]          # 4 dwords each
EA = [220, 238]            # 10 registers each, k = ka0 .. ka0 + 9
EB = [230, 248]            # 8 registers each,  k = kb0 .. kb0 + 7
KA0 = [-4, 2]
KB0 = [-3, 3]
CLOBBER_LO, CLOBBER_HI = 196, 255


def groups(half):
    c0, c1 = HALVES[half]
    out = []
    for rho in range(4):
        cols = [c for c in range(c0, c1) if (c - 15) % 4 == rho]
        qs = [(c - 15 - rho) // 4 for c in cols]
        used = sorted({k for q in qs for k in range(q, q + 4) if -1 <= k <= 7})   # k <= -2 and k >= 8 are zero
        out.append((rho, cols, qs, used))
    return out


def prep_ops(half, kb, rho, used, eset):
    """instructions that fill the E files of set `eset` for group (kb, rho)"""
    ops = []
    xb = XBUF[kb & 1]
    for k in used:
        lo = f"v{xb + k}" if 0 <= k <= 7 else "0"
        hi = f"v{xb + k + 1}" if 0 <= k + 1 <= 7 else "0"
        dsts = []
        ia = k - KA0[half]
        if 0 <= ia < 10:
            dsts.append(EA[eset] + ia)
        ib = k - KB0[half]
        if 0 <= ib < 8:
            dsts.append(EB[eset] + ib)
        for dreg in dsts:
            if rho == 0:
                ops.append(f"v_mov_b32 v{dreg}, {lo}")
            elif lo == "0" and hi == "0":
                ops.append(f"v_mov_b32 v{dreg}, 0")
            else:
                ops.append(f"v_alignbyte_b32 v{dreg}, {hi}, {lo}, {rho}")
    return ops


def mfma_ops(half, kb, rho, cols, qs, eset):
    c0, _ = HALVES[half]
    ops = []
    ab = ABUF[kb & 1]
    for c, q in zip(cols, qs):
        if q % 2 == 0:
            r = EA[eset] + (q - KA0[half])
        else:
            r = EB[eset] + (q - KB0[half])
        assert r % 2 == 0
        cin = "%{BIAS}" if kb == 0 else f"%{c - c0}"
        ops.append(f"v_mfma_i32_16x16x64_i8 %{c - c0}, v[{ab}:{ab + 3}], v[{r}:{r + 3}], {cin}")
    return ops


def lds_loads(kb, xs_op, as_op):
    xb, ab = XBUF[kb & 1], ABUF[kb & 1]
    return [
        f"ds_read_b128 v[{xb}:{xb + 3}], {xs_op} offset:{(2 * kb) * 1024}",
        f"ds_read_b128 v[{xb + 4}:{xb + 7}], {xs_op} offset:{(2 * kb + 1) * 1024}",
        f"ds_read_b128 v[{ab}:{ab + 3}], {as_op} offset:{kb * 1024}",
    ]


def asm_half(half, nkb):
    c0, c1 = HALVES[half]
    ncol = c1 - c0
    xs_op, as_op, bias = f"%{ncol}", f"%{ncol + 1}", f"{ncol + 2}"
    lines = []
    # the positions of the E files that are read but never written (k <= -2, k >= 8) are zero
    grp = groups(half)
    read_k = {k for (_, _, qs, _) in grp for q in qs for k in range(q, q + 4)}
    for eset in range(2):
        for k in sorted(read_k):
            if -1 <= k <= 7:
                continue
            if 0 <= k - KA0[half] < 10:
                lines.append(f"v_mov_b32 v{EA[eset] + k - KA0[half]}, 0")
            if 0 <= k - KB0[half] < 8:
                lines.append(f"v_mov_b32 v{EB[eset] + k - KB0[half]}, 0")
    lines += lds_loads(0, xs_op, as_op)
    seq = [(kb, g) for kb in range(nkb) for g in grp]       # g = (rho, cols, qs, used)
    pending_mfma = []                                         # MFMAs of the previous group, to interleave with this prep
    for idx, (kb, (rho, cols, qs, used)) in enumerate(seq):
        eset = idx & 1
        pre = []
        if rho == 1 and kb + 1 < nkb:
            # the other buffers were last read by the MFMAs of (kb - 1, rho = 3), all issued by now
            pre += lds_loads(kb + 1, xs_op, as_op)
        if rho == 0:
            pre.append("s_waitcnt lgkmcnt(0)")
            xb = XBUF[kb & 1]
            need = range(0, 7) if half == 0 else range(2, 8)
            pre += [f"v_xor_b32 v{xb + k}, 0x80808080, v{xb + k}" for k in need]
        prep = pre + prep_ops(half, kb, rho, used, eset)
        # interleave: one MFMA of the previous group, then a share of this group's preparation
        if pending_mfma:
            per = (len(prep) + len(pending_mfma) - 1) // len(pending_mfma)
            pi = 0
            for mi, m in enumerate(pending_mfma):
                if mi == len(pending_mfma) - 1:
                    # keep the last MFMA of the previous group after all of this group's writes
                    lines += prep[pi:]
                    pi = len(prep)
                    lines.append(m)
                else:
                    lines.append(m)
                    lines += prep[pi:pi + per]
                    pi += per
            lines.append("s_nop 0")
        else:
            lines += prep
            lines.append("s_nop 1")
        pending_mfma = [m.replace("%{BIAS}", "%" + bias) for m in mfma_ops(half, kb, rho, cols, qs, eset)]
    lines += pending_mfma
    lines += ["s_nop 7", "s_nop 7"]
    return lines, ncol


def emit():
    out = ["// GENERATED by gen_mm8.py -- do not edit", ""]
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
