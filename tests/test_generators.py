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
    # a label or a branch in between: the count is not static
    assert gen.resolve_waits(["@ELOAD", ld, ld, ".Lx_%=:", ld, "@EWAIT"])[-1] == "s_waitcnt vmcnt(0)"
    assert gen.resolve_waits(["@ELOAD", ld, ld, "s_cbranch_scc1 .Lx_%=", ld, "@EWAIT"])[-1] == "s_waitcnt vmcnt(0)"
    # stores count like loads (one counter on gfx9)
    assert gen.resolve_waits(["@ELOAD", ld, ld, "global_store_dwordx4 v[4:5], v[0:3], off", "@EWAIT"])[-1] == "s_waitcnt vmcnt(1)"


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
    mf = [ln for ln in lines if ln.startswith("v_mfma")]
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
            assert sum(ln.startswith("v_cmp_ne_u32") for ln in lines) == nout
