"""The generated inline-asm passes of the full-size matrix-core kernel (honeybadgermpc_amd/csrc/gen_mm8w.py): structural
invariants that the GPU parity tests can only catch as rare wrong answers (hazards inside an asm string are nobody's but the
generator's to get right).  CPU-only: the generator is plain Python."""
import importlib.util
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GEN = os.path.join(HERE, "..", "honeybadgermpc_amd", "csrc", "gen_mm8w.py")


@pytest.fixture(scope="module")
def gen():
    spec = importlib.util.spec_from_file_location("gen_mm8w", GEN)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_resolve_waits_counts_vector_memory_operations(gen):
    ld = "global_load_dwordx4 v[0:3], v[4:5], off"
    # nothing issued since the marked loads: only they (and older operations, which return first) are waited for
    assert gen.resolve_waits(["@ELOAD", ld, ld, "v_mov_b32 v0, 0", "@EWAIT"])[-1] == "s_waitcnt vmcnt(0)"
    # two younger loads may stay in flight
    assert gen.resolve_waits(["@ELOAD", ld, ld, ld, ld, "@EWAIT"])[-1] == "s_waitcnt vmcnt(2)"
    # a full wait in between: everything landed, later operations need not
    assert gen.resolve_waits(["@ELOAD", ld, ld, ld, "s_waitcnt vmcnt(0)", ld, ld, ld, "@EWAIT"])[-1] == "s_waitcnt vmcnt(3)"
    # a forward skip over register-only code (the conditional subtraction) does not disturb the count
    assert gen.resolve_waits(["@ELOAD", ld, ld, ld, "s_cbranch_scc0 .Lcs_7_%=", "v_mov_b32 v0, 0", ".Lcs_7_%=:", "@EWAIT"])[-1] == "s_waitcnt vmcnt(1)"
    # a label or a branch in between: the count is not static
    assert gen.resolve_waits(["@ELOAD", ld, ld, ".Lx_%=:", ld, "@EWAIT"])[-1] == "s_waitcnt vmcnt(0)"
    assert gen.resolve_waits(["@ELOAD", ld, ld, "s_cbranch_scc1 .Lx_%=", ld, "@EWAIT"])[-1] == "s_waitcnt vmcnt(0)"
    # stores count like loads (one counter on gfx9)
    assert gen.resolve_waits(["@ELOAD", ld, ld, "global_store_dwordx4 v[4:5], v[0:3], off", "@EWAIT"])[-1] == "s_waitcnt vmcnt(1)"


def check_skips(gen, lines, nout):
    """the conditional subtraction sits behind a forward branch: its label follows inside the same unit, nothing but register
    arithmetic in between, and resolve_waits keeps counting across it"""
    br = [i for i, ln in enumerate(lines) if ln.startswith("s_cbranch_scc0 .Lcs_")]
    assert len(br) == nout
    for i in br:
        label = lines[i].split()[1] + ":"
        j = lines.index(label)
        assert i < j <= i + 30
        body = lines[i + 1:j]
        assert all(ln.startswith(("v_mov_b32", "v_add_co_u32", "v_addc_co_u32", "v_cndmask_b32")) for ln in body), body
        assert sum(ln.startswith("v_cndmask_b32") for ln in body) == 8
        # the test in front of it: top word against ~(2^256 - p)_7, bit 256, or-ed into scc
        assert lines[i - 1].startswith("s_or_b64") and lines[i - 2].startswith("v_cmp_ne_u32_e64") and lines[i - 3].startswith("v_cmp_le_u32_e32 vcc")
    assert len({lines[i].split()[1] for i in br}) == nout


def check_fold(gen, o, lines, nout):
    """the fold of a sum's high half on the matrix cores (reduce_output): 16 MFMAs per output with VGPR results"""
    fold = [(i, ln) for i, ln in enumerate(lines) if ln.startswith("v_mfma") and ln.split()[1].startswith("v[")]
    assert len(fold) == 16 * nout
    for n, (i, ln) in enumerate(fold):
        m = re.match(r"v_mfma_i32_16x16x64_i8 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], (v\[(\d+):(\d+)\]|0)$", ln)
        assert m, ln
        d0, d1, a0, a1, b0, b1 = (int(m.group(k)) for k in range(1, 7))
        step = n % 16
        ks, eb = divmod(step, 8)
        assert (d0, d1) == (gen.D0 + 4 * eb, gen.D0 + 4 * eb + 3)
        assert (a0, a1) == (gen.AB[step % 4], gen.AB[step % 4] + 3)
        assert (b0, b1) == (gen.HB + 4 * ks, gen.HB + 4 * ks + 3) and b0 % 2 == 0
        # the first byte half starts its column block from 0, the second accumulates onto it
        assert m.group(7) == ("0" if ks == 0 else f"v[{d0}:{d1}]")
        # its A operand: read from LDS into that buffer after the buffer's previous reader was issued, and waited for
        prev_reader = max([j for j, _ in fold[:n] if f"v[{a0}:{a1}], v[{gen.HB}" in lines[j] or f"v[{a0}:{a1}], v[{gen.HB + 4}" in lines[j]] or [-1])
        loads = [j for j in range(prev_reader + 1, i) if lines[j].startswith(f"ds_read_b128 v[{a0}:{a1}], {o('ATB')}")]
        assert len(loads) == 1, (ln, loads)
        assert f"offset:{gen.FOLD_ROW * step}" in lines[loads[0]] or (step == 0 and "offset" not in lines[loads[0]]) or lines[loads[0]].endswith("offset:0")
        waits = [j for j in range(loads[0] + 1, i) if lines[j].startswith("s_waitcnt lgkmcnt")]
        assert waits, ln
        # the wait allows at most the LDS reads of this output's fold issued after that load
        younger = sum(1 for j in range(loads[0] + 1, waits[-1]) if lines[j].startswith("ds_read_b128 v[") and f"], {o('ATB')}" in lines[j])
        assert int(re.search(r"lgkmcnt\((\d+)\)", lines[waits[-1]]).group(1)) <= younger
    # an MFMA's result is not read by the VALU before 18 issue slots have passed; the biased words H are written before their readers
    for r in range(nout):
        last = fold[16 * r + 15][0]
        reads = [j for j in range(last + 1, len(lines)) if re.search(rf"v_mad_i64_i32 .*, v{gen.D0}\b|v_mad_i64_i32 .*, v{gen.D0 + 1}\b", lines[j])]
        assert reads and reads[0] - last >= 18
        first = fold[16 * r][0]
        xors = [j for j in range(fold[16 * r - 1][0] + 1 if r else 0, first) if lines[j].startswith("v_xor_b32 v") and "0x80808080" in lines[j] and int(re.match(r"v_xor_b32 v(\d+)", lines[j]).group(1)) in range(gen.HB, gen.HB + 8)]
        assert len(xors) == 8 and first - xors[-1] >= 2
        # the row constant's pairs land over the operand registers only after the last MFMA that reads those has been issued
        pl = [j for j, l in enumerate(lines) if l.startswith(f"ds_read_b128 v[{gen.PB}:{gen.PB + 3}], {o('CRL')}")]
        assert len(pl) == nout and all(any(f[0] < j for f in [fold[16 * k + 15] for k in range(nout)]) for j in pl)
    for r, j in enumerate(sorted(j for j, l in enumerate(lines) if l.startswith(f"ds_read_b128 v[{gen.PB}:{gen.PB + 3}], {o('CRL')}"))):
        assert fold[16 * r + 15][0] < j and (r + 1 == nout or j < fold[16 * (r + 1)][0])


@pytest.mark.parametrize("nout", [4, 3, 2])
@pytest.mark.parametrize("check", [False, True])
@pytest.mark.parametrize("peel", [1, 2, 3, 4])
def test_pass_structure(gen, check, peel, nout):
    o, lines = gen.pass_lines(check, peel, nout)
    # a lane keeps nout of its four sums (row tiles of 4 nout rows): that many reductions, stores, word columns
    assert sum(ln.startswith("global_store_dwordx4") for ln in lines) == 2 * nout
    assert sum(ln.startswith("v_accvgpr_read_b32") for ln in lines) == gen.NC * nout
    text = "\n".join(lines)
    assert "@E" not in text
    mf = [ln for ln in lines if ln.startswith("v_mfma") and ln.split()[1].startswith("a[")]
    # peeled K-blocks + one two-block loop body, 156 MFMAs each
    assert len(mf) == 156 * (peel + (2 if peel > 1 else 0))      # a single K-block has no loop
    # every column starts exactly once from the inline constant 0, in the first K-block
    first = [ln for ln in mf if ln.rstrip().endswith(", 0")]
    assert len(first) == gen.NC and len({ln.split()[1] for ln in first}) == gen.NC
    assert all(ln in mf[:156] for ln in first)
    # B operands are even-aligned 4-register windows of a file set, A operands whole digit buffers
    for ln in mf:
        m = re.match(r"v_mfma_i32_16x16x64_i8 a\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], (a\[\d+:\d+\]|0)$", ln)
        assert m, ln
        c0, c1, a0, a1, b0, b1 = (int(m.group(i)) for i in range(1, 7))
        assert c1 == c0 + 3 and c0 % 4 == 0 and c0 < 4 * gen.NC
        assert a1 == a0 + 3 and any(a0 in bufs for bufs in gen.ABUF)
        assert b1 == b0 + 3 and b0 % 2 == 0 and any(fs <= b0 <= fs + 18 for fs in gen.F_SETS)
    check_fold(gen, o, lines, nout)
    check_skips(gen, lines, nout)
    # nothing runs under a narrowed exec mask except the loads / stores / compare it was narrowed for
    inside = False
    for ln in lines:
        if ln.startswith("s_and_b64 exec"):
            inside = True
        elif ln.startswith("s_mov_b64 exec"):
            inside = False
        elif inside:
            assert ln.startswith(("global_load", "global_store", "v_cmp_ne", "s_or_b64", "s_waitcnt")), ln
    assert not inside
    # the statement owns what it touches: explicit VGPRs stay inside [RB, 255], accumulators inside a0..a251
    for ln in lines:
        for r in re.findall(r"\bv(\d+)\b", ln):
            assert gen.RB <= int(r) <= 255, ln
        for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", ln):
            assert gen.RB <= int(lo) <= int(hi) <= 255, ln
    # the last K-block does not prefetch: no digit load may be in flight when the statement ends
    last_block = lines[max(i for i, ln in enumerate(lines) if ln.startswith(".Lmm8w_rest") or i == 0):]
    tail_start = max(i for i, ln in enumerate(lines) if ln.startswith("v_mfma"))
    after_last_wait = lines[max(i for i, ln in enumerate(lines[:tail_start]) if ln.startswith("s_waitcnt vmcnt(0)")):]
    assert not any(ln.startswith("global_load_dwordx4 v[" + str(b)) for ln in after_last_wait for bufs in gen.ABUF for b in bufs), "digit prefetch after the last wait"
    assert last_block


@pytest.mark.parametrize("nout", [4, 3, 2])
def test_reduction_units_cover_every_output(gen, nout):
    for check in (False, True):
        o, lines = gen.reduce_lines(check, nout)
        stores = [ln for ln in lines if ln.startswith("global_store_dwordx4")]
        assert len(stores) == 2 * nout               # two 16-byte stores per output
        if check:
            assert sum(ln.startswith("global_load_dwordx4") for ln in lines) == 2 * nout
            assert sum(ln.startswith("v_cmp_ne_u32_e32 vcc, 0") for ln in lines) == nout        # the compare of an output with its received row
        check_fold(gen, o, lines, nout)
        check_skips(gen, lines, nout)
