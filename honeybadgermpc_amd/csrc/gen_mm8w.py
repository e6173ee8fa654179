#!/usr/bin/env python3
"""Emits hb_mm8w_body.inc: the passes of k_mm8w (hb_mfma_wide.hip) as inline-asm statements.

k_mm8w is the matrix-core mat-vec for FULL-SIZE matrix entries (any residue mod p: inverse Vandermonde
matrices at omega-power points, Vandermonde matrices whose powers outgrow 2^127, the fused decode + validate
matrices of hb_open.hip, arbitrary hb_matrix operands).  Entries are cut into 32 base-256 digits M_b, inputs
into their 32 bytes X_a:

    S = sum_l M[l] x[l] = sum_c 2^(8c) col_c,   col_c = sum_l sum_b M_b[l] X_{c-b}[l],   c < 63.

One v_mfma_i32_16x16x64_i8 contracts 64 products per (row, chunk); a lane's 16 operand bytes are cut as in gen_mm8.py:
8 TERMS x 8 DIGITS per K-block -- lane (n, g) feeds terms 8 kb + 2 g and + 1 of chunk n, the 32 digits of an entry are four
groups G (b in [8 G, 8 G + 8)), and with s = c - 7 - 8 G = 4 q + rho the B operand is the 8-byte windows [s, s + 7] of both
elements: dwords q, q + 1 of each shifted right by rho bytes.  Every window s in [-7, 31] feeds FOUR MFMAs (group G -> column
s + 7 + 8 G): 156 per K-block = 78 per 4 terms (the first version cut 4 terms x 16 digits: 94), and with the register file laid
out [dword k][element e] the operand of window q is registers 2 q .. 2 q + 3, always even-aligned: one file, 18 preparation ops
per (K-block, rho) for eight terms (the first version: 18 per four, half of them copies into a second file one register apart).
The int8 pipe does the 1024 byte products of a 256 x 256-bit multiplication in 16 cycles per 16 x 16 outputs, the VALU needs 81
half-rate v_mad_u64_u32 per lane.

A pass = one asm statement, SOFTWARE-PIPELINED over passes: the kernel runs one wave per SIMD (all 63 accumulators of
16 x 16 outputs live in AGPRs a0..a251), so nothing else could hide the reduction of S mod p (~190 instructions per
output) -- it would simply follow the MFMA phase, which leaves the VALU three quarters idle.  Instead a pass ends by
moving its sums out of the AGPRs as 17 words per output (68 VGPRs), and the NEXT pass reduces, compares and stores them
between its own MFMAs: up to four K-blocks are written out, each carrying an equal share of that reduction; the K-blocks between
them are a loop of two-block bodies, so the code does not grow with the inner dimension.  After its last pass a wave
runs the reduction alone (mm8w_reduce).

Every pass exists for lanes that keep 4 sums (row tiles of 16 rows), 3 sums (12 rows: the fourth row of every group of the
MFMA tile is empty, its reduction and word assembly are not emitted) and 2 sums (8 rows); hb_mfma_wide.hip picks per matrix.

Register files (VGPRs the statement owns: v96 .. v255): v164.. the MFMA operand files (two file sets, the next group's
shifts built from the other set while the current group's MFMAs issue; element prefetch XB, digit buffers ABUF of two
K-block parities); v96 .. v163 the reduction (the row to compare with, the 32 columns of the fold, its operands, the packed
result).  SGPRs s68 .. s89: the reduction constants (scalar loads from WideParams), a saved exec.

Reduction of one output (reduce_output below has the arithmetic; tests/fold_model.py is its big-integer model):
  the high eight words of the sum go back through the matrix cores against the table t_b = 2^(256 + 8 b) mod p (16 MFMAs whose
  A operands come from LDS), the 32 columns of that, the low eight words, the top word times 2^512 mod p and the per-row constant
  are gathered per 32-bit word; a one-word Barrett quotient; R - q p in words; conditional subtraction of p; then the lane's mode
  word says store (1), compare (2) or neither (0).  ~190 instructions per output; rounds 2's fold on the VALU (19 radix-2^29
  digits, 90 MADs, a two-digit quotient) took ~340.
"""
import os

NC = 63
NWORDS = 17
RB = 96                        # first register the statement owns
# ---- MFMA operand files (the layout of gen_mm8.py: 8 terms x 8 digits per K-block) ----
XB = 164                       # 16 dwords: LDS prefetch of the next K-block's two elements, element e dword k at XB + 8 e + k
ABUF = [[180, 184, 188, 192], [196, 200, 204, 208]]   # [K-block parity][digit group]: 4 dwords each
F_SETS = [212, 234]            # 22 registers each: dword k = -2 .. 8 of element e at base + 2 (k + 2) + e
F_KMIN = -2
RHOS = (0, 1, 2, 3)
NG = 4                         # digit groups of 8
# ---- reduction file (v96 .. v163) ----
EX = 96                        # row to compare with, 8 words (requested when the output's reduction starts)
D0 = 104                       # the 32 columns of the fold: eight MFMA results of four
HB = 136                       # the high half H of the sum, biased: two B operands of four
AB = [144, 148, 152, 156]      # four buffers for the fold's A operands (LDS, three steps ahead)
T1, Q, TP = 160, 161, 162      # bit 256, the quotient, a pair
FOLD_ROW = 272                 # bytes per row of the fold table in LDS
PB = 136                       # P_w, eight pairs: over HB and two A buffers once the MFMAs have been issued
OW = 104                       # packed result, 8 words (the columns are spent by then)
PN = 112                       # 2^256 - p in registers
UB = 120                       # ow + (2^256 - p)
DIFF = 128
# ---- tail (word assembly) ----
TL_TMP = [[96, 97, 98, 99], [100, 101, 102, 103]]
TL_T = [[104, 106, 108, 110], [112, 114, 116, 118]]
# ---- SGPRs ----
S_PNEG, S_MU, S_C512 = 68, 76, 77    # WideParams: pneg[8] mu c512[8] pad[3], loaded to s68 .. s87
S_T0, S_T1 = 85, 86            # scratch: a word and a pair (the padding of WideParams lands there first)
S_SAVE = 88                    # saved exec, pair


ABLATE = os.environ.get("HB_GEN_MM8W_ABLATE", "").split(",")    # timing experiments only (wrong results): nored, notail, nomfma, nofload, nofwait, nofmfma
SPREAD = float(os.environ.get("HB_GEN_MM8W_SPREAD", "1"))       # room of a fold line in the merged stream, relative to the other reduction lines


class Ops:
    """operand numbering of the asm statement"""

    def __init__(self, check, nout=4, select=False):
        self.outs, self.ins = [], []
        self.nout = nout           # outputs per lane that are kept: 4 (16-row tiles) or 3 (12-row tiles: the fourth row of every group of
                                   # the MFMA tile is padding -- a matrix of 22 rows is two tiles either way, and a pass reduces 3 sums, not 4)
        for r in range(nout):
            for j in range(NWORDS):
                self.outs.append((f"W{r}_{j}", '"+v"', f"w[{r}][{j}]"))
        self.outs += [("XA", '"+v"', "xa"), ("VA", '"+v"', "va"), ("CNT", '"+s"', "cnt")]
        if check:
            self.outs.append(("FLAG", '"+s"', "flag"))
        self.ins += [("ABASE", '"s"', "abase"), ("K256", '"s"', "k256"), ("K64K", '"s"', "k64k"), ("K16M", '"s"', "k16m"),
                     ("B4", '"s"', "bias4"), ("B3", '"s"', "bias3"), ("WPP", '"s"', "wpa"), ("CRL", '"v"', "crl_addr"), ("ATB", '"v"', "atb_addr")]
        for r in range(nout):
            self.ins.append((f"ADDR{r}", '"v"', f"addr[{r}]"))
        for r in range(nout):
            self.ins.append((f"MODE{r}", '"v"', f"mode[{r}]"))
        if select:
            self.ins.append(("SEL", '"s"', "sel"))       # which of the bodies of a multi-body statement runs (multi_lines)
        self.idx = {name: i for i, (name, _, _) in enumerate(self.outs + self.ins)}

    def __call__(self, name):
        return f"%{self.idx[name]}"


def f(s, k, e):
    assert -2 <= k <= 8 and e in (0, 1)
    return F_SETS[s] + 2 * (k - F_KMIN) + e


def windows(rho):
    """dword offsets q of the windows s = 4 q + rho in [-7, 31]"""
    return [q for q in range(-2, 8) if -7 <= 4 * q + rho <= 31]


def acc(c):
    return f"a[{4 * c}:{4 * c + 3}]"


def loads(par, o):
    """both elements of the lane and the four digit groups of the NEXT K-block -> XB, ABUF[par]; the cursors move on by a block"""
    L = []
    for e in (0, 1):
        for h in (0, 1):
            off = (e * 2 + h) * 1024
            L.append(f"ds_read_b128 v[{XB + 8 * e + 4 * h}:{XB + 8 * e + 4 * h + 3}], {o('XA')}" + (f" offset:{off}" if off else ""))
    L.append(f"v_add_u32 {o('XA')}, 0x1000, {o('XA')}")
    for g in range(NG):
        a0 = ABUF[par][g]
        L.append(f"global_load_dwordx4 v[{a0}:{a0 + 3}], {o('VA')}, {o('ABASE')}" + (f" offset:{g * 1024}" if g else ""))
    L.append(f"v_add_u32 {o('VA')}, 0x1000, {o('VA')}")
    return L


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


def prep(gi):
    """fill file set gi & 1 for group gi = (K-block, rho): from XB for rho = 0, else one more byte of shift of the other set"""
    rho = RHOS[gi % 4]
    s = gi & 1
    ops = []
    if rho == 0:
        ops.append("s_waitcnt lgkmcnt(0)")
        for e in (0, 1):
            for k in range(8):
                ops.append(f"v_xor_b32 v{f(s, k, e)}, 0x80808080, v{XB + 8 * e + k}")
            ops.append(f"v_mov_b32 v{f(s, -1, e)}, 0")
    else:
        for e in (0, 1):
            for k in range(-1, 8):
                ops.append(f"v_alignbyte_b32 v{f(s, k, e)}, v{f(1 - s, k + 1, e)}, v{f(1 - s, k, e)}, 1")
    return ops


def mfmas(gi, par, seen):
    """MFMAs of group gi; `seen` = columns already started (None: accumulate always)"""
    rho = RHOS[gi % 4]
    s = gi & 1
    out = []
    for q in windows(rho):
        r = f(s, q, 0)
        assert r % 2 == 0
        for grp in range(NG):
            c = 4 * q + rho + 7 + 8 * grp
            assert 0 <= c < NC
            ab = ABUF[par][grp]
            cin = acc(c)
            if seen is not None and c not in seen:
                cin = "0"
                seen.add(c)
            out.append(f"v_mfma_i32_16x16x64_i8 {acc(c)}, v[{ab}:{ab + 3}], v[{r}:{r + 3}], {cin}")
    return out


def kblock(par, first, last, o):
    """one K-block of 8 terms (4 groups, 156 MFMAs): its digits in ABUF[par].  On entry the first group's file set is ready, XB has
    been consumed and this block's digits are in flight.  The LAST block must not prefetch: nothing waits for a load issued there,
    and one that lands after the asm statement would overwrite registers the compiler has taken back (it did, once per ~10^4
    launches, when L2 was cold)."""
    L = []
    seen = set() if first else None
    # the digits of this block were requested one block ago; every MFMA reading the other buffers has been issued
    L.append("s_waitcnt vmcnt(0)")
    if not last:
        L += loads(1 - par, o)
    for gi in range(4):
        L += interleave(mfmas(gi, par, seen), prep(gi + 1) if not (last and gi == 3) else [])
        L.append("s_nop 0")
    if seen is not None:
        assert seen == set(range(NC))
    return L


# ------------------------------------------------------------------------------------------------ reduction
_uid = iter(range(1 << 30))      # labels inside one statement


class Unit(list):
    """lines that stay together in the merged stream; w = its share of room there"""
    w = 1.0


def masked(o, r, mode_value, body):
    """body under exec & (mode == mode_value): one unit, never interleaved with the MFMA stream"""
    return [Unit([f"s_mov_b64 s[{S_SAVE}:{S_SAVE + 1}], exec",
                  f"v_cmp_eq_u32_e32 vcc, {mode_value}, {o(f'MODE{r}')}",
                  "s_and_b64 exec, exec, vcc"] + body + [f"s_mov_b64 exec, s[{S_SAVE}:{S_SAVE + 1}]"])]


def fold_loads(o, i):
    """A operand of fold step i = (byte half ks, column block eb) -> buffer i % 4: row i of the table at LDS offset 0 (272 bytes per
    row: sixteen lanes' 16 digits, then 16 zero bytes for the lanes outside the diagonal block -- their base points there)"""
    b = AB[i % 4]
    return f"ds_read_b128 v[{b}:{b + 3}], {o('ATB')} offset:{FOLD_ROW * i}"


def reduce_output(o, r, check):
    """units (lists of lines) reducing output r's 17 words and storing / comparing the canonical element.

    S = L + 2^256 H + 2^512 W16 (L, H eight words each).  The high half goes back through the matrix cores: with H's 32 bytes h_b,
    2^256 H = sum_b h_b 2^(256 + 8 b) = sum_b h_b t_b (mod p), t_b = 2^(256 + 8 b) mod p or that minus p, whichever has 32 balanced
    base-256 digits s_(b, e).  A lane (n, g) holds the words of ITS output, so the product runs block-diagonally: the B operand is
    16 bytes of H (biased by XOR 0x80 like every int8 input here), the A operand of lane (m, g') is s_(16 ks + pos, 4 eb + m % 4) when
    g' = m / 4 and zero otherwise, and D[m][n] = column 4 eb + m % 4 of the fold of output (4 (m / 4) + j, n) lands in lane
    (n, m / 4), register m % 4: the lane that owns that output.  16 MFMAs per output (2 byte halves x 8 column blocks) replace the
    ten-digit fold on the VALU (90 MADs and the digit extraction).  Then, in 32-bit words w < 8,
        P_w = [bias of the four columns + row constant word] + sum_k D_(4 w + k) 2^(8 k) + L_w + W16 (2^512 mod p)_w      (< 2^49)
    (the row constant carries 128 sum_b t_b and the column biases, hb_mfma_wide.hip), R = sum_w P_w 2^(32 w) < 2^272, the quotient
    qhat = floor(floor(R / 2^240) mu / 2^46), mu = floor(2^286 / p), is floor(R / p) or one less, u_w = qhat (2^256 - p)_w + P_w,
    and the words of sum_w u_w 2^(32 w) are R - qhat p with qhat on top of bit 256; one conditional subtraction of p.
    (tests/fold_model.py is this arithmetic in big integers, bounds asserted.)"""
    U = []
    wt = [1.0]
    def one(ln):
        u = Unit([ln]); u.w = wt[0]; U.append(u)
    w = lambda i: o(f"W{r}_{i}")  # noqa: E731
    if check:
        # the received row: from HBM, a microsecond away -- requested first, compared last
        U += masked(o, r, 2, ["@ELOAD", f"global_load_dwordx4 v[{EX}:{EX + 3}], {o(f'ADDR{r}')}, off",
                              f"global_load_dwordx4 v[{EX + 4}:{EX + 7}], {o(f'ADDR{r}')}, off offset:16"])
    # the fold's lines are given SPREAD times the room of the others in the merged stream: what stands between an operand's LDS
    # read and the MFMA that takes it is three steps of three lines plus the main stream's share -- a wave alone on its SIMD has
    # nothing else to cover the LDS latency with
    wt[0] = SPREAD
    for i in range(3):
        if "nofload" not in ABLATE:
            one(fold_loads(o, i))
    for i in range(8):
        one(f"v_xor_b32 v{HB + i}, 0x80808080, {w(8 + i)}")
    for i in range(16):
        ks, eb = divmod(i, 8)
        d0 = D0 + 4 * eb
        if "nofwait" not in ABLATE and "nofload" not in ABLATE:
            one(f"s_waitcnt lgkmcnt({min(2, 15 - i)})")
        cin = "0" if ks == 0 else f"v[{d0}:{d0 + 3}]"
        if "nofmfma" not in ABLATE:
            one(f"v_mfma_i32_16x16x64_i8 v[{d0}:{d0 + 3}], v[{AB[i % 4]}:{AB[i % 4] + 3}], v[{HB + 4 * ks}:{HB + 4 * ks + 3}], {cin}")
        if i + 3 < 16 and "nofload" not in ABLATE:
            one(fold_loads(o, i + 3))
    wt[0] = 1.0
    # the eight pairs [bias + row constant word] of output r's row (64 bytes; output r's row is 4 r rows further on) -> P, over the
    # operand registers of the MFMAs just issued (they have read them; the loads land a hundred cycles later)
    for i in range(4):
        one(f"ds_read_b128 v[{PB + 4 * i}:{PB + 4 * i + 3}], {o('CRL')} offset:{256 * r + 16 * i}")
    pp = lambda j: f"v[{PB + 2 * j}:{PB + 2 * j + 1}]"  # noqa: E731
    for j in range(8):
        if j % 2 == 0:
            one(f"s_waitcnt lgkmcnt({3 - j // 2})")
        one(f"v_mad_u64_u32 {pp(j)}, vcc, {w(j)}, 1, {pp(j)}")
    for j in range(8):
        one(f"v_mad_u64_u32 {pp(j)}, vcc, {w(16)}, s{S_C512 + j}, {pp(j)}")
    # (an MFMA's result may be read by the VALU 18 issue slots after it at the earliest: the 21 lines above stand between)
    K = {0: "1", 1: o("K256"), 2: o("K64K"), 3: o("K16M")}
    for k in range(4):
        for j in range(8):
            one(f"v_mad_i64_i32 {pp(j)}, vcc, v{D0 + 4 * j + k}, {K[k]}, {pp(j)}")
    for ln in [f"v_mad_u64_u32 v[{TP}:{TP + 1}], vcc, v{PB + 13}, 1, {pp(7)}",
               f"v_alignbit_b32 v{TP}, v{TP + 1}, v{TP}, 16",
               f"v_mad_u64_u32 v[{TP}:{TP + 1}], vcc, v{TP}, s{S_MU}, 0",
               f"v_lshrrev_b32 v{Q}, 14, v{TP + 1}"]:
        one(ln)
    for j in range(8):
        one(f"v_mad_u64_u32 {pp(j)}, vcc, v{Q}, s{S_PNEG + j}, {pp(j)}")
    # words of sum_w u_w 2^(32 w); what stands above bit 256 is qhat plus bit 256 of the remainder.  One unit: the carry lives in vcc
    cp = [f"v_mov_b32 v{OW}, v{PB}",
          f"v_add_co_u32_e32 v{OW + 1}, vcc, v{PB + 2}, v{PB + 1}"]
    for j in range(2, 8):
        cp.append(f"v_addc_co_u32_e32 v{OW + j}, vcc, v{PB + 2 * j}, v{PB + 2 * j - 1}, vcc")
    cp += [f"v_addc_co_u32_e32 v{T1}, vcc, 0, v{PB + 15}, vcc",
           f"v_sub_u32_e32 v{T1}, v{T1}, v{Q}"]
    U.append(Unit(cp))
    # conditional subtraction: the carry out of ow + (2^256 - p) says ow >= p.  One unit: the carry chain lives in vcc
    # (an SGPR operand beside the carry in vcc would be two constant-bus reads: 2^256 - p goes through registers).
    # r < 2p < 2^257: bit 256 of r (T1) also means r >= p, and r - p is the same sum mod 2^256.  It joins the carry chain as a
    # ninth word: T1 + 0xffffffff + carry carries out iff T1 or carry.
    # The whole of it sits behind a test that almost never fires: the quotient is one short once in ~10^4 outputs, and otherwise
    # r < p shows in the top word alone -- ow_7 + (2^256 - p)_7 + 1 < 2^32 leaves no room for a carry out, whatever the lower
    # words do.  A wave takes the 26 instructions only if one of its lanes has T1 set or ow_7 >= ~(2^256 - p)_7.
    uid = next(_uid)
    cs = [f"s_not_b32 s{S_T0}, s{S_PNEG + 7}",
          f"v_cmp_le_u32_e32 vcc, s{S_T0}, v{OW + 7}",
          f"v_cmp_ne_u32_e64 s[{S_T1}:{S_T1 + 1}], 0, v{T1}",
          f"s_or_b64 s[{S_T1}:{S_T1 + 1}], vcc, s[{S_T1}:{S_T1 + 1}]",
          f"s_cbranch_scc0 .Lcs_{uid}_%="]
    if "nocsskip" in ABLATE:
        cs = []
    for j in range(8):
        cs.append(f"v_mov_b32 v{PN + j}, s{S_PNEG + j}")
    cs.append(f"v_add_co_u32_e32 v{UB}, vcc, v{PN}, v{OW}")
    for j in range(1, 8):
        cs.append(f"v_addc_co_u32_e32 v{UB + j}, vcc, v{PN + j}, v{OW + j}, vcc")
    cs.append(f"v_addc_co_u32_e32 v{T1}, vcc, -1, v{T1}, vcc")
    for j in range(8):
        cs.append(f"v_cndmask_b32_e32 v{OW + j}, v{OW + j}, v{UB + j}, vcc")
    if "nocsskip" not in ABLATE:
        cs.append(f".Lcs_{uid}_%=:")
    u = Unit(cs)
    if "nocsskip" not in ABLATE:
        u.w = 6.0 / len(cs)           # what a wave executes of it, as a rule
    U.append(u)
    if check:
        cmp = ["@EWAIT"]
        for j in range(8):
            cmp.append(f"v_xor_b32 v{DIFF + j}, v{EX + j}, v{OW + j}")
        cmp += [f"v_or3_b32 v{DIFF}, v{DIFF}, v{DIFF + 1}, v{DIFF + 2}",
                f"v_or3_b32 v{DIFF + 3}, v{DIFF + 3}, v{DIFF + 4}, v{DIFF + 5}",
                f"v_or3_b32 v{DIFF}, v{DIFF}, v{DIFF + 6}, v{DIFF + 7}",
                f"v_or_b32 v{DIFF}, v{DIFF}, v{DIFF + 3}"]
        U.append(Unit(cmp))
        U += masked(o, r, 2, [f"v_cmp_ne_u32_e32 vcc, 0, v{DIFF}", f"s_or_b64 {o('FLAG')}, {o('FLAG')}, vcc"])
    U += masked(o, r, 1, [f"global_store_dwordx4 {o(f'ADDR{r}')}, v[{OW}:{OW + 3}], off",
                          f"global_store_dwordx4 {o(f'ADDR{r}')}, v[{OW + 4}:{OW + 7}], off offset:16"])
    return U




def merge(stream, units):
    """spread the reduction units evenly behind the MFMAs of `stream`"""
    if "nored" in ABLATE:
        units = []
    if "nomfma" in ABLATE:
        stream = [ln for ln in stream if not ln.startswith("v_mfma")] + [f"v_mfma_i32_16x16x64_i8 a[0:3], v[{ABUF[0][0]}:{ABUF[0][0] + 3}], v[{F_SETS[0]}:{F_SETS[0] + 3}], a[0:3]"]
    n_mfma = sum(1 for ln in stream if ln.startswith("v_mfma"))
    total = sum(u.w * len(u) for u in units)
    out, done, seen, placed = [], 0, 0, 0.0
    for ln in stream:
        out.append(ln)
        if ln.startswith("v_mfma"):
            seen += 1
            want = total * seen / n_mfma
            while done < len(units) and (placed < want - 1e-9 or seen == n_mfma):
                out += units[done]
                placed += units[done].w * len(units[done])
                done += 1
    assert done == len(units)
    return out


def resolve_waits(lines):
    """@ELOAD marks the two loads of the row to compare with, @EWAIT the point where they must have landed.  Vector memory
    operations return in order, so the wait is vmcnt(number issued since) -- a plain vmcnt(0) would also wait for the next term
    block's digits, requested a few instructions earlier (that cost 10 % of a CHECK pass).  Anything the walk cannot count
    (a label or a branch in between) falls back to vmcnt(0)."""
    out = []
    since = None            # VMEM operations issued after the marked loads; None = unknown
    for ln in lines:
        if ln == "@ELOAD":
            since = -2       # the two loads themselves follow
            continue
        if ln == "@EWAIT":
            out.append(f"s_waitcnt vmcnt({since})" if since is not None and 0 <= since <= 15 else "s_waitcnt vmcnt(0)")
            since = None
            continue
        out.append(ln)
        if ln.startswith(".Lcs_") or ln.startswith("s_cbranch_scc0 .Lcs_"):
            pass             # a forward skip over register-only code: no vector memory operation inside, the count stands
        elif ln.startswith(".L") or ln.startswith("s_cbranch"):
            since = None
        elif ln.startswith("global_load") or ln.startswith("global_store"):
            if since is not None:
                since += 1
        elif ln.startswith("s_waitcnt vmcnt(0)"):
            if since is not None and since >= 0:
                since = 0    # everything landed; later operations are counted from here (the wait then allows all of them)
    return out


def consts(o):
    return [f"s_load_dwordx16 s[{S_PNEG}:{S_PNEG + 15}], {o('WPP')}, 0x0",
            f"s_load_dwordx4 s[{S_PNEG + 16}:{S_PNEG + 19}], {o('WPP')}, 0x40"]


# ------------------------------------------------------------------------------------------------ tail
def tail(o):
    """The accumulators leave as 32-bit words of S = sum_c (col_c + bias) 2^(8c): per word four signed columns go into one
    64-bit sum t_j by v_mad_i64_i32 (x 1, 2^8, 2^16, 2^24; the first one adds the bias of all four: every t_j is non-negative), and
    word j = lo(t_j) + hi(t_(j-1)) + carry by one add-with-carry that writes the word where the next pass wants it.  The four outputs
    of the lane run side by side, each with its carry in an SGPR pair of its own (the reduction constants there are dead by now;
    the next statement reloads them), so consecutive links of a chain are a dozen instructions apart."""
    K = {1: o("K256"), 2: o("K64K"), 3: o("K16M")}
    L = []
    n_words = (NC + 3) // 4
    cy = lambda r: f"s[{S_PNEG + 2 * r}:{S_PNEG + 2 * r + 1}]"  # noqa: E731
    for j in range(n_words):
        ts = TL_T[j & 1]
        prev = TL_T[1 - (j & 1)]
        cols = [c for c in range(4 * j, 4 * j + 4) if c < NC]
        bias = o("B4") if len(cols) == 4 else o("B3")
        assert len(cols) in (3, 4)
        reads = [[f"v_accvgpr_read_b32 v{TL_TMP[i & 1][r]}, a{4 * c + r}" for r in range(o.nout)] for i, c in enumerate(cols)]
        L += reads[0]
        for i, c in enumerate(cols):
            if i + 1 < len(cols):
                L += reads[i + 1]
            mul = "1" if i == 0 else K[i]
            for r in range(o.nout):
                add = bias if i == 0 else f"v[{ts[r]}:{ts[r] + 1}]"
                L.append(f"v_mad_i64_i32 v[{ts[r]}:{ts[r] + 1}], vcc, v{TL_TMP[i & 1][r]}, {mul}, {add}")
        for r in range(o.nout):
            wj = o(f"W{r}_{j}")
            if j == 0:
                L.append(f"v_mov_b32 {wj}, v{ts[r]}")
            elif j == 1:
                L.append(f"v_add_co_u32_e64 {wj}, {cy(r)}, v{ts[r]}, v{prev[r] + 1}")
            else:
                L.append(f"v_addc_co_u32_e64 {wj}, {cy(r)}, v{ts[r]}, v{prev[r] + 1}, {cy(r)}")
    assert n_words == NWORDS - 1
    for r in range(o.nout):
        L.append(f"v_addc_co_u32_e64 {o(f'W{r}_{NWORDS - 1}')}, {cy(r)}, v{TL_T[(n_words - 1) & 1][r] + 1}, 0, {cy(r)}")
    return L


def split(units, parts):
    """consecutive shares of equal room"""
    total = sum(u.w * len(u) for u in units)
    out, cur, acc = [], [], 0.0
    for u in units:
        if len(out) < parts - 1 and acc >= total * (len(out) + 1) / parts - 1e-9:
            out.append(cur)
            cur = []
        cur.append(u)
        acc += u.w * len(u)
    out.append(cur)
    while len(out) < parts:
        out.append([])
    return out


def pass_lines(check, peel, nout=4, select=False):
    """`peel` K-blocks are straight-line code carrying the reduction of the pass before, in equal shares (one wave per SIMD
    issues an instruction every ~5.5 cycles at best -- profiles/r01_mad_issue_rate_vs_occupancy.txt, r02_mm8w_phase_timing.txt --
    so everything a pass executes counts); the other nkb - peel K-blocks run as a loop of two-block bodies in the middle (the digit
    buffers alternate by block parity, so the launcher picks peel = nkb for nkb <= 2, else 3 for odd and 4 for even nkb)."""
    o = Ops(check, nout, select)
    L = consts(o)
    # positions that are read but never written stay zero: k = -2 and k = 8
    for s in range(2):
        for k in (-2, 8):
            for e in (0, 1):
                L.append(f"v_mov_b32 v{f(s, k, e)}, 0")
    L += loads(0, o)                                # K-block 0
    L += prep(0) + ["s_nop 1"]                      # (its lgkmcnt(0) also covers the scalar loads)
    units = []
    for r in range(nout):
        units += reduce_output(o, r, check)
    shares = split(units, peel)
    head = (peel + 1) // 2
    for i in range(peel):
        if i == head:
            L += [f"s_cmp_eq_u32 {o('CNT')}, 0", "s_cbranch_scc1 .Lmm8w_rest_%="]
            L.append(".Lmm8w_loop_%=:")
            L += kblock(head & 1, False, False, o) + kblock(1 - (head & 1), False, False, o)
            L += [f"s_sub_u32 {o('CNT')}, {o('CNT')}, 1", f"s_cmp_lg_u32 {o('CNT')}, 0", "s_cbranch_scc1 .Lmm8w_loop_%="]
            L.append(".Lmm8w_rest_%=:")
        L += merge(kblock(i & 1, i == 0, i == peel - 1, o), shares[i])
    L += ["s_nop 7", "s_nop 7"]
    if "notail" not in ABLATE:
        L += tail(o)
    return o, resolve_waits(L)


def reduce_lines(check, nout=4):
    o = Ops(check, nout)
    L = consts(o) + ["s_waitcnt lgkmcnt(0)"]
    for r in range(nout):
        for u in reduce_output(o, r, check):
            L += u
    return o, resolve_waits(L)


def multi_lines(check, peels=(2, 3, 4), nout=4):
    """ONE statement holding the passes written out for 2, 3 and 4 K-blocks; the scalar operand SEL picks the body.  k_mm8w_flat runs
    pieces of a pass of any length >= 2 (SEL = 2: two K-blocks; 3 / 4: odd / even lengths, CNT two-block loop bodies in the middle).
    Three statements behind an if / else cost the kernel ~80 spilled registers a round: the compiler shuffles the 68 words of the sums
    between the branches; one statement it cannot look into has one register assignment."""
    o = None
    L = []
    for i, peel in enumerate(peels):
        o, lines = pass_lines(check, peel, nout, select=True)
        lines = [ln.replace(".Lmm8w_", f".Lmm8w_b{peel}_") for ln in lines]
        if i + 1 < len(peels):
            L += [f"s_cmp_lg_u32 {o('SEL')}, {peel}", f"s_cbranch_scc1 .Lmm8w_skip{peel}_%="]
        L += lines
        if i + 1 < len(peels):
            L += ["s_branch .Lmm8w_end_%=", f".Lmm8w_skip{peel}_%=:"]
    L.append(".Lmm8w_end_%=:")
    return o, L


def emit_fn(name, o, lines, check):
    out = []
    n = o.nout
    sig = (f"uint32_t (&w)[{n}][17], uint32_t &xa, uint32_t &va, uint32_t &cnt, uint64_t &flag, uint64_t abase, int32_t k256, int32_t k64k, "
           f"int32_t k16m, int64_t bias4, int64_t bias3, uint64_t wpa, uint32_t crl_addr, uint32_t atb_addr, const uint64_t (&addr)[{n}], "
           f"const uint32_t (&mode)[{n}]" + (", int32_t sel" if "SEL" in o.idx else ""))
    out.append(f"static __device__ __forceinline__ void {name}({sig}) {{")
    if not check:
        out.append("    (void)flag;")
    out.append("    asm volatile(")
    for ln in lines:
        out.append(f'        "{ln}\\n\\t"')
    out.append("        : " + ", ".join(f"{c}({e})" for _, c, e in o.outs))
    out.append("        : " + ", ".join(f"{c}({e})" for _, c, e in o.ins))
    clob = [f'"v{r}"' for r in range(RB, 256)] + [f'"a{r}"' for r in range(4 * NC)] + [f'"s{r}"' for r in range(S_PNEG, S_SAVE + 2)]
    out.append("        : " + ", ".join(clob) + ', "vcc", "scc", "memory");')
    out.append("}")
    out.append("")
    return out


PEELS = (1, 2, 3, 4)


def emit():
    out = ["// GENERATED by gen_mm8w.py -- do not edit", ""]
    for check in (False, True):
        sfx = "_check" if check else ""
        for nout in (4, 3, 2):
            for peel in PEELS:
                o, lines = pass_lines(check, peel, nout)
                out += emit_fn(f"mm8w_pass{sfx}_p{peel}_k{nout}", o, lines, check)
            o, lines = reduce_lines(check, nout)
            out += emit_fn(f"mm8w_reduce{sfx}_k{nout}", o, lines, check)
        o, lines = multi_lines(check)
        out += emit_fn(f"mm8w_pass{sfx}_multi_k4", o, lines, check)
    return "\n".join(out)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    text = emit()
    path = os.path.join(here, "hb_mm8w_body.inc")
    old = open(path).read() if os.path.exists(path) else None
    if old != text:
        open(path, "w").write(text)
    print(path)
