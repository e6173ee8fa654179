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
16 x 16 outputs live in AGPRs a0..a251), so nothing else could hide the reduction of S mod p (~310 VALU instructions per
output) -- it would simply follow the MFMA phase, which leaves the VALU three quarters idle.  Instead a pass ends by
moving its sums out of the AGPRs as 17 words per output (68 VGPRs), and the NEXT pass reduces, compares and stores them
between its own MFMAs: up to four K-blocks are written out, each carrying an equal share of that reduction; the K-blocks between
them are a loop of two-block bodies, so the code does not grow with the inner dimension.  After its last pass a wave
runs the reduction alone (mm8w_reduce).

Every pass exists for lanes that keep 4 sums (row tiles of 16 rows), 3 sums (12 rows: the fourth row of every group of the
MFMA tile is empty, its reduction and word assembly are not emitted) and 2 sums (8 rows); hb_mfma_wide.hip picks per matrix.

Register files (VGPRs the statement owns: v96 .. v255): v164.. the MFMA operand files (two file sets, the next group's
shifts built from the other set while the current group's MFMAs issue; element prefetch XB, digit buffers ABUF of two
K-block parities); v96 .. v163 the reduction (ten 64-bit columns, two buffers for the T_k rows, the packed
result, the row to compare with).  SGPRs s68 .. s89: the Barrett constants (scalar loads from WideParams), a saved exec.

Reduction of one output (the arithmetic of k_prescale_tab / the first version's C++ epilogue, same bounds):
  19 radix-2^29 digits of the 17 words; V = low nine digits + per-row constant + sum_{k >= 9} s_k T_k (T_k = 2^(29k) mod p from
  LDS, 90 MADs) < 2^290; carry; two-digit Barrett quotient against mu = floor(2^290 / p); V + q (2^261 - p) in nine digits;
  pack to eight words; conditional subtraction of p; then the lane's mode word says store (1), compare (2) or neither (0).
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
C0 = 96                        # ten 64-bit columns: C0 + 2j (low), + 1 (high)
TB = [116, 126, 136]           # three buffers of nine for the T_k rows / the row constant (even bases: 128-bit LDS reads)
T1, SK, ZERO = 125, 135, 145
OW = 146                       # packed result, 8 words
EX = 154                       # row to compare with, 8 words (requested when the output's reduction starts)
T2 = 162                       # pair
QP, Q0, Q1 = 116, 118, 119     # Barrett quotient (the T buffers are free by then)
UB = 96                        # ow + (2^256 - p): the columns are free by then
DIFF = 116
# ---- tail (word assembly) ----
TL_TMP = [[96, 97, 98, 99], [100, 101, 102, 103]]
TL_T = [[104, 106, 108, 110], [112, 114, 116, 118]]
# ---- SGPRs ----
S_PBAR, S_PNEG, S_M0, S_M1 = 68, 77, 85, 86    # WideParams: pbar[9] pneg[8] m0 m1 pad, loaded to s68 .. s87
S_SAVE = 88                    # saved exec, pair
MASK = "0x1fffffff"
LB = 29


class Ops:
    """operand numbering of the asm statement"""

    def __init__(self, check, nout=4):
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
                     ("B4", '"s"', "bias4"), ("B3", '"s"', "bias3"), ("WPP", '"s"', "wpa"), ("CRL", '"v"', "crl_addr")]
        for r in range(nout):
            self.ins.append((f"ADDR{r}", '"v"', f"addr[{r}]"))
        for r in range(nout):
            self.ins.append((f"MODE{r}", '"v"', f"mode[{r}]"))
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
def masked(o, r, mode_value, body):
    """body under exec & (mode == mode_value): one unit, never interleaved with the MFMA stream"""
    return [[f"s_mov_b64 s[{S_SAVE}:{S_SAVE + 1}], exec",
             f"v_cmp_eq_u32_e32 vcc, {mode_value}, {o(f'MODE{r}')}",
             "s_and_b64 exec, exec, vcc"] + body + [f"s_mov_b64 exec, s[{S_SAVE}:{S_SAVE + 1}]"]]


def digit(o, r, k, dst):
    """dst = digit k (29 bits) of the 17 words of output r"""
    bit = LB * k
    j, sft = bit >> 5, bit & 31
    w = lambda i: o(f"W{r}_{i}")  # noqa: E731
    if sft == 0:
        return [f"v_and_b32 v{dst}, {MASK}, {w(j)}"]
    if j + 1 >= NWORDS:
        return [f"v_lshrrev_b32 v{dst}, {sft}, {w(j)}", f"v_and_b32 v{dst}, {MASK}, v{dst}"]
    return [f"v_alignbit_b32 v{dst}, {w(j + 1)}, {w(j)}, {sft}", f"v_and_b32 v{dst}, {MASK}, v{dst}"]


def row_reads(addr, off, buf):
    """nine digits of a 48-byte LDS row -> TB[buf] (two 16-byte reads and one dword: the third buffer has nine registers)"""
    b = TB[buf]
    return [f"ds_read_b128 v[{b}:{b + 3}], {addr} offset:{off}",
            f"ds_read_b128 v[{b + 4}:{b + 7}], {addr} offset:{off + 16}",
            f"ds_read_b32 v{b + 8}, {addr} offset:{off + 32}"]


def t_reads(k):
    return row_reads(f"v{ZERO}", 48 * k, k % 3)


def cpair(j):
    return f"v[{C0 + 2 * j}:{C0 + 2 * j + 1}]"


def carry(upto):
    """columns 0 .. upto-1 keep 29 bits, the rest moves up"""
    L = []
    for j in range(upto):
        L += [f"v_lshrrev_b64 v[{T2}:{T2 + 1}], {LB}, {cpair(j)}",
              f"v_and_b32 v{C0 + 2 * j}, {MASK}, v{C0 + 2 * j}",
              f"v_lshl_add_u64 {cpair(j + 1)}, v[{T2}:{T2 + 1}], 0, {cpair(j + 1)}"]
    return L


def reduce_output(o, r, check, nfold=10):
    """units (lists of lines) reducing output r's 17 words and storing / comparing the canonical element.
    nfold = 10: all 19 digits of the biased sum; nfold = 9: the sum is known to stay below 2^(29 * 18) (the launcher checks the
    bias against that bound: any matrix of fewer than 64 terms), digit 18 is zero and its fold is not emitted."""
    assert nfold in (9, 10)
    U = []
    one = lambda ln: U.append([ln])  # noqa: E731
    if check:
        # the received row: from HBM, a microsecond away -- requested first, compared last
        U += masked(o, r, 2, ["@ELOAD", f"global_load_dwordx4 v[{EX}:{EX + 3}], {o(f'ADDR{r}')}, off",
                              f"global_load_dwordx4 v[{EX + 4}:{EX + 7}], {o(f'ADDR{r}')}, off offset:16"])
    # per-row constant -> TB[2], T_9 -> TB[0], T_10 -> TB[1]; the rows are requested THREE steps ahead of their use: one wave
    # per SIMD has only its own instructions (and the MFMAs between them) to cover the LDS latency
    for ln in row_reads(o("CRL"), 256 * r, 2) + t_reads(0) + t_reads(1):          # output r's row is 4 r rows (64 B each) further on
        one(ln)
    one("s_waitcnt lgkmcnt(6)")
    for j in range(9):
        for ln in digit(o, r, j, T1):
            one(ln)
        one(f"v_add_u32 v{C0 + 2 * j}, v{T1}, v{TB[2] + j}")
        one(f"v_mov_b32 v{C0 + 2 * j + 1}, 0")
    one(f"v_mov_b32 v{C0 + 18}, 0")
    one(f"v_mov_b32 v{C0 + 19}, 0")
    for ln in t_reads(2):
        one(ln)
    # V += s_(9+k) T_k
    for k in range(nfold):
        for ln in digit(o, r, 9 + k, SK):
            one(ln)
        one(f"s_waitcnt lgkmcnt({3 * min(2, nfold - 1 - k)})")
        b = TB[k % 3]
        for j in range(9):
            one(f"v_mad_u64_u32 {cpair(j)}, vcc, v{SK}, v{b + j}, {cpair(j)}")
        if k + 3 < nfold:
            for ln in t_reads(k + 3):
                one(ln)
    for ln in carry(9):
        one(ln)
    v8, v9 = C0 + 16, C0 + 18
    # qhat = floor(floor(V / 2^232) mu / 2^58): floor(V / p) or one less
    for ln in [f"v_mad_u64_u32 v[{QP}:{QP + 1}], vcc, v{v8}, s{S_M0}, 0",
               f"v_lshrrev_b64 v[{QP}:{QP + 1}], {LB}, v[{QP}:{QP + 1}]",
               f"v_mad_u64_u32 v[{QP}:{QP + 1}], vcc, v{v8}, s{S_M1}, v[{QP}:{QP + 1}]",
               f"v_mad_u64_u32 v[{QP}:{QP + 1}], vcc, v{v9}, s{S_M0}, v[{QP}:{QP + 1}]",
               f"v_lshrrev_b64 v[{QP}:{QP + 1}], {LB}, v[{QP}:{QP + 1}]",
               f"v_mad_u64_u32 v[{QP}:{QP + 1}], vcc, v{v9}, s{S_M1}, v[{QP}:{QP + 1}]",
               f"v_and_b32 v{Q0}, {MASK}, v{QP}",
               f"v_alignbit_b32 v{Q1}, v{QP + 1}, v{QP}, {LB}"]:
        one(ln)
    # V + q (2^261 - p), nine digits (what leaves digit 8 is the 2^261 q that was added)
    for j in range(9):
        one(f"v_mov_b32 v{C0 + 2 * j + 1}, 0")
        one(f"v_mad_u64_u32 {cpair(j)}, vcc, v{Q0}, s{S_PBAR + j}, {cpair(j)}")
        if j > 0:
            one(f"v_mad_u64_u32 {cpair(j)}, vcc, v{Q1}, s{S_PBAR + j - 1}, {cpair(j)}")
    for ln in carry(8):
        one(ln)
    one(f"v_and_b32 v{C0 + 16}, {MASK}, v{C0 + 16}")
    # nine digits -> eight words
    one(f"v_lshl_or_b32 v{OW}, v{C0 + 2}, {LB}, v{C0}")
    for j in range(1, 8):
        one(f"v_lshrrev_b32 v{T1}, {3 * j}, v{C0 + 2 * j}")
        one(f"v_lshl_or_b32 v{OW + j}, v{C0 + 2 * j + 2}, {LB - 3 * j}, v{T1}")
    # conditional subtraction: the carry out of ow + (2^256 - p) says ow >= p.  One unit: the carry chain lives in vcc
    # (an SGPR operand beside the carry in vcc would be two constant-bus reads: 2^256 - p goes through the free T buffer)
    for j in range(8):
        one(f"v_mov_b32 v{TB[1] + j}, s{S_PNEG + j}")
    # r < 2p < 2^257: bit 256 of r (bit 24 of the top digit; the eight words drop it) also means r >= p, and r - p is the
    # same sum mod 2^256.  It joins the carry chain as a ninth word: hi + 0xffffffff + carry carries out iff hi or carry.
    cs = [f"v_lshrrev_b32 v{T1}, 24, v{C0 + 16}",
          f"v_add_co_u32_e32 v{UB}, vcc, v{TB[1]}, v{OW}"]
    for j in range(1, 8):
        cs.append(f"v_addc_co_u32_e32 v{UB + j}, vcc, v{TB[1] + j}, v{OW + j}, vcc")
    cs.append(f"v_addc_co_u32_e32 v{T1}, vcc, -1, v{T1}, vcc")
    for j in range(8):
        cs.append(f"v_cndmask_b32_e32 v{OW + j}, v{OW + j}, v{UB + j}, vcc")
    U.append(cs)
    if check:
        cmp = ["@EWAIT"]
        for j in range(8):
            cmp.append(f"v_xor_b32 v{DIFF + j}, v{EX + j}, v{OW + j}")
        cmp += [f"v_or3_b32 v{DIFF}, v{DIFF}, v{DIFF + 1}, v{DIFF + 2}",
                f"v_or3_b32 v{DIFF + 3}, v{DIFF + 3}, v{DIFF + 4}, v{DIFF + 5}",
                f"v_or3_b32 v{DIFF}, v{DIFF}, v{DIFF + 6}, v{DIFF + 7}",
                f"v_or_b32 v{DIFF}, v{DIFF}, v{DIFF + 3}"]
        U.append(cmp)
        U += masked(o, r, 2, [f"v_cmp_ne_u32_e32 vcc, 0, v{DIFF}", f"s_or_b64 {o('FLAG')}, {o('FLAG')}, vcc"])
    U += masked(o, r, 1, [f"global_store_dwordx4 {o(f'ADDR{r}')}, v[{OW}:{OW + 3}], off",
                          f"global_store_dwordx4 {o(f'ADDR{r}')}, v[{OW + 4}:{OW + 7}], off offset:16"])
    return U


ABLATE = os.environ.get("HB_GEN_MM8W_ABLATE", "").split(",")    # timing experiments only (wrong results): nored, notail, nofold, nommfa


def merge(stream, units):
    """spread the reduction units evenly behind the MFMAs of `stream`"""
    if "nored" in ABLATE:
        units = []
    if "nomfma" in ABLATE:
        stream = [ln for ln in stream if not ln.startswith("v_mfma")] + [f"v_mfma_i32_16x16x64_i8 a[0:3], v[{ABUF[0][0]}:{ABUF[0][0] + 3}], v[{F_SETS[0]}:{F_SETS[0] + 3}], a[0:3]"]
    n_mfma = sum(1 for ln in stream if ln.startswith("v_mfma"))
    out, done, seen = [], 0, 0
    for ln in stream:
        out.append(ln)
        if ln.startswith("v_mfma"):
            seen += 1
            want = (len(units) * seen + n_mfma - 1) // n_mfma
            while done < want:
                out += units[done]
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
        if ln.startswith(".L") or ln.startswith("s_cbranch"):
            since = None
        elif ln.startswith("global_load") or ln.startswith("global_store"):
            if since is not None:
                since += 1
        elif ln.startswith("s_waitcnt vmcnt(0)"):
            if since is not None and since >= 0:
                since = 0    # everything landed; later operations are counted from here (the wait then allows all of them)
    return out


def consts(o):
    return [f"s_load_dwordx16 s[{S_PBAR}:{S_PBAR + 15}], {o('WPP')}, 0x0",
            f"s_load_dwordx4 s[{S_PBAR + 16}:{S_PBAR + 19}], {o('WPP')}, 0x40",
            f"v_mov_b32 v{ZERO}, 0"]


# ------------------------------------------------------------------------------------------------ tail
def tail(o):
    """The accumulators leave as 32-bit words of S = sum_c (col_c + bias) 2^(8c): per word four signed columns go into one
    64-bit sum by v_mad_i64_i32 (x 1, 2^8, 2^16, 2^24; the first one adds the bias of all four), the previous word's high half
    comes in by one more MAD (x 1) -- every term is non-negative, so there is no carry flag anywhere -- and the four outputs
    of the lane run side by side."""
    K = {1: o("K256"), 2: o("K64K"), 3: o("K16M")}
    L = []
    n_words = (NC + 3) // 4
    for j in range(n_words):
        ts = TL_T[j & 1]
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
        if j > 0:
            for r in range(o.nout):
                L.append(f"v_mad_u64_u32 v[{ts[r]}:{ts[r] + 1}], vcc, v{TL_T[1 - (j & 1)][r] + 1}, 1, v[{ts[r]}:{ts[r] + 1}]")
        for r in range(o.nout):
            L.append(f"v_mov_b32 {o(f'W{r}_{j}')}, v{ts[r]}")
    assert n_words == NWORDS - 1
    for r in range(o.nout):
        L.append(f"v_mov_b32 {o(f'W{r}_{NWORDS - 1}')}, v{TL_T[(n_words - 1) & 1][r] + 1}")
    return L


def split(units, parts):
    n = len(units)
    return [units[n * i // parts:n * (i + 1) // parts] for i in range(parts)]


def pass_lines(check, peel, nout=4, nfold=10):
    """`peel` K-blocks are straight-line code carrying the reduction of the pass before, in equal shares (one wave per SIMD
    issues an instruction every ~5.5 cycles at best -- profiles/r01_mad_issue_rate_vs_occupancy.txt, r02_mm8w_phase_timing.txt --
    so everything a pass executes counts); the other nkb - peel K-blocks run as a loop of two-block bodies in the middle (the digit
    buffers alternate by block parity, so the launcher picks peel = nkb for nkb <= 2, else 3 for odd and 4 for even nkb)."""
    o = Ops(check, nout)
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
        units += reduce_output(o, r, check, nfold)
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


def reduce_lines(check, nout=4, nfold=10):
    o = Ops(check, nout)
    L = consts(o) + ["s_waitcnt lgkmcnt(0)"]
    for r in range(nout):
        for u in reduce_output(o, r, check, nfold):
            L += u
    return o, resolve_waits(L)


def emit_fn(name, o, lines, check):
    out = []
    n = o.nout
    sig = (f"uint32_t (&w)[{n}][17], uint32_t &xa, uint32_t &va, uint32_t &cnt, uint64_t &flag, uint64_t abase, int32_t k256, int32_t k64k, "
           f"int32_t k16m, int64_t bias4, int64_t bias3, uint64_t wpa, uint32_t crl_addr, const uint64_t (&addr)[{n}], "
           f"const uint32_t (&mode)[{n}]")
    out.append(f"static __device__ __forceinline__ void {name}({sig}) {{")
    if not check:
        out.append("    (void)flag;")
    out.append("    asm volatile(")
    for ln in lines:
        out.append(f'        "{ln}\\n\\t"')
    out.append("        : " + ", ".join(f"{c}({e})" for _, c, e in o.outs))
    out.append("        : " + ", ".join(f"{c}({e})" for _, c, e in o.ins))
    clob = [f'"v{r}"' for r in range(RB, 256)] + [f'"a{r}"' for r in range(4 * NC)] + [f'"s{r}"' for r in range(S_PBAR, S_SAVE + 2)]
    out.append("        : " + ", ".join(clob) + ', "vcc", "scc", "memory");')
    out.append("}")
    out.append("")
    return out


PEELS = (1, 2, 3, 4)


def emit():
    out = ["// GENERATED by gen_mm8w.py -- do not edit", ""]
    for nfold in (10, 9):
        fs = "" if nfold == 10 else "_f9"
        for check in (False, True):
            sfx = "_check" if check else ""
            for nout in (4, 3, 2):
                for peel in PEELS:
                    o, lines = pass_lines(check, peel, nout, nfold)
                    out += emit_fn(f"mm8w_pass{sfx}{fs}_p{peel}_k{nout}", o, lines, check)
                o, lines = reduce_lines(check, nout, nfold)
                out += emit_fn(f"mm8w_reduce{sfx}{fs}_k{nout}", o, lines, check)
    return "\n".join(out)


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    text = emit()
    path = os.path.join(here, "hb_mm8w_body.inc")
    old = open(path).read() if os.path.exists(path) else None
    if old != text:
        open(path, "w").write(text)
    print(path)
