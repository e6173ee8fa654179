"""
Pins the CPU oracle (oracle/hbmpc_oracle.c) against
  (1) the known-answer vectors of the reference's own tests for this path
      (/root/reference/tests/test_ntl.py, test_reed_solomon.py, test_reed_solomon_wb.py,
       fixtures.py roots of unity), restated here with their literals, and
  (2) golden vectors produced by importing the reference's pure-Python layers
      (oracle/gen_golden.py -> tests/golden/*.json).
CPU only; no GPU needed.
"""
import random

import pytest

import oracle
from conftest import BLS

P = BLS
# first entries of the reference's hard-coded root table (tests/fixtures.py:23-29)
ROOTS = [
    1,
    52435875175126190479447740508185965837690552500527637822603658699938581184512,
    52435875175126190475982595682112313518914282969839895044333406231173219221505,
    28761180743467419819834788392525162889723178799021384024940474588120723734663,
    38476778329304481878022718993882556548812578500290864179952442003245540347252,
    39328881859443649819318207548060215749094715634259317161033277606721139812495,
]


def peval(coeffs, x, p):
    return sum(c * pow(x, j, p) for j, c in enumerate(coeffs)) % p


# ---------------------------------------------------------------- (1) reference known answers
def test_ref_interpolate():  # tests/test_ntl.py:18-28
    assert oracle.lagrange_interpolate([1, 2], [1, 2], P) == [0, 1]


def test_ref_batch_vandermonde_interpolate():  # tests/test_ntl.py:31-41
    assert oracle.vandermonde_batch_interpolate([1, 2], [[1, 2], [3, 5]], P) == [[0, 1], [1, 2]]


def test_ref_batch_vandermonde_evaluate():  # tests/test_ntl.py:44-54
    assert oracle.vandermonde_batch_evaluate([1, 2], [[0, 1], [1, 2]], P) == [[1, 2], [3, 5]]


def test_ref_fft_small():  # tests/test_ntl.py:57-68
    assert oracle.fft([0, 1], 5, 13, 4) == [1, 5, 12, 8]


def test_ref_fft_big_and_partial():  # tests/test_ntl.py:71-136
    rnd = random.Random(7)
    d, n, k, omega = 20, 32, 25, ROOTS[5]
    coeffs = [rnd.randrange(P) for _ in range(d)]
    want = [peval(coeffs, pow(omega, i, P), P) for i in range(n)]
    assert oracle.fft(coeffs, omega, P, n) == want
    assert oracle.partial_fft(coeffs, omega, P, n, k) == want[:k]
    batch = [[rnd.randrange(P) for _ in range(d)] for _ in range(64)]
    got = oracle.fft_batch_evaluate(batch, omega, P, n, k)
    assert got == [[peval(c, pow(omega, i, P), P) for i in range(k)] for c in batch]


def test_ref_fft_interpolate():  # tests/test_ntl.py:139-179
    omega, n = ROOTS[3], 8
    zs = [3, 0]
    ys = [peval([1, 2], pow(omega, z, P), P) for z in zs]
    assert oracle.fft_interpolate(zs, ys, omega, P, n) == [1, 2]
    zs = [3, 0, 5]
    polys = [[1, 2, 0], [3, 2, 1], [3, 4, 2]]
    ys = [[peval(pl, pow(omega, z, P), P) for z in zs] for pl in polys]
    assert oracle.fft_batch_interpolate(zs, ys, omega, P, n) == polys


def _corrupt(rnd, message, num_errors, num_nones, max_val=131):
    message = list(message)
    idx = rnd.sample(range(len(message)), num_errors + num_nones)
    for i in range(num_errors):
        message[idx[i]] = rnd.randint(0, max_val)
    for i in range(num_nones):
        message[idx[i + num_errors]] = None
    return message


@pytest.mark.parametrize("int_msg", [[2, 3, 2, 8, 7, 5, 9, 5], [0] * 8])
def test_ref_gao_interpolate(int_msg):  # tests/test_ntl.py:196-265
    rnd = random.Random(11)
    k, n, p = 8, 22, 53
    t = k - 1
    x = list(range(n))
    encoded = [peval(int_msg, xi, p) for xi in x]
    cmax, emax = n - 2 * t - 1, (n - 2 * t - 1) // 2
    for ne, nn in [(0, 0), (0, cmax), (emax, 0), (emax // 2, cmax // 4)]:
        for _ in range(5):
            decoded, _ = oracle.gao_interpolate(x, _corrupt(rnd, encoded, ne, nn), k, p)
            assert decoded == int_msg


def test_ref_gao_interpolate_fft():  # tests/test_ntl.py:268-314
    rnd = random.Random(12)
    int_msg = [2, 3, 2, 8, 7, 5, 9, 5]
    k, n, order, omega = 8, 22, 32, ROOTS[5]
    t = k - 1
    z = list(range(n))
    x = [pow(omega, zi, P) for zi in z]
    encoded = [peval(int_msg, xi, P) for xi in x]
    cmax, emax = n - 2 * t - 1, (n - 2 * t - 1) // 2
    for ne, nn in [(0, 0), (0, cmax), (emax, 0), (emax // 2, cmax // 4)]:
        word = _corrupt(rnd, encoded, ne, nn)
        decoded, _ = oracle.gao_interpolate(x, word, k, P, z=z, omega=omega, order=order, use_omega_powers=True)
        assert decoded == int_msg
        # the non-FFT variant must agree, including the error-locator cofactor
        assert oracle.gao_interpolate(x, word, k, P) == oracle.gao_interpolate(
            x, word, k, P, z=z, omega=omega, order=order, use_omega_powers=True)


def test_ref_sqrt_mod():  # tests/test_ntl.py:331-341
    rnd = random.Random(0)
    for _ in range(100):
        sq = pow(rnd.randrange(P), 2, P)
        assert pow(oracle.sqrt_mod(sq, P), 2, P) == sq


def test_ref_codec_vectors():  # tests/test_reed_solomon.py:19-99
    assert oracle.vandermonde_batch_evaluate([1, 2, 3, 4], [[1, 2]], P) == [[3, 5, 7, 9]]
    assert oracle.vandermonde_batch_evaluate([1, 2, 3, 4], [[1, 2], [2, 3]], P) == [[3, 5, 7, 9], [5, 8, 11, 14]]
    # decode from z = [1, 3] -> x = [2, 4]
    assert oracle.vandermonde_batch_interpolate([2, 4], [[5, 9], [8, 14]], P) == [[1, 2], [2, 3]]
    # robust: [3, 5, 0, 9] has party 2 wrong
    co, err = oracle.gao_interpolate([1, 2, 3, 4], [3, 5, 0, 9], 2, P)
    assert co == [1, 2]
    assert [i for i in range(4) if peval(err, i + 1, P) == 0] == [2]


def test_ref_roots_of_unity(golden):  # tests/fixtures.py:23-29 vs get_omega-derived constants
    c = golden("constants.json")
    # seed-0 chain: omega(2^r) from the golden file has exact order 2^r
    for r in range(1, 11):
        w = c["omega"][str(1 << r)]
        assert pow(w, 1 << r, P) == 1 and pow(w, 1 << (r - 1), P) != 1
    # EvalPoint(n=4).omega is the reference's galois_field_roots[2]
    assert c["evalpoint"][0]["omega"] == ROOTS[2]
    for r, root in enumerate(ROOTS):
        assert pow(root, 1 << r, P) == 1
        if r:
            assert pow(root, 1 << (r - 1), P) != 1


# ---------------------------------------------------------------- (2) golden vectors
def test_golden_vandermonde(golden):
    g = golden("vandermonde.json")
    for case in g["cases"]:
        p, x = case["p"], case["x"]
        assert oracle.vandermonde_batch_evaluate(x, case["polys"], p) == case["evals"]
        xz = [x[z] for z in case["z"]]
        ys = [[row[z] for z in case["z"]] for row in case["evals"]]
        assert oracle.vandermonde_batch_interpolate(xz, ys, p) == case["interp"]
        for row, want in zip(ys, case["interp"]):
            trimmed = list(want)
            while trimmed and trimmed[-1] == 0:
                trimmed.pop()
            assert oracle.lagrange_interpolate(xz, row, p) == trimmed
    ev = g["evaluate"]
    assert [oracle.evaluate(ev["coeffs"], xv, ev["p"]) for xv in ev["xs"]] == ev["ys"]


def test_golden_fft(golden):
    for case in golden("fft.json")["cases"]:
        p, om, n = case["p"], case["omega"], case["n"]
        assert oracle.fft_batch_evaluate(case["coeffs"], om, p, n, n) == case["evals"]
        k = max(1, (3 * n) // 4)
        assert oracle.fft_batch_evaluate(case["coeffs"], om, p, n, k) == [e[:k] for e in case["evals"]]
        assert oracle.fft(case["coeffs"][-1], om, p, n) == case["evals"][-1]


def test_golden_fft_interpolate(golden):
    for case in golden("fft_interpolate.json")["cases"]:
        got = oracle.fft_batch_interpolate(case["zs"], case["ys"], case["omega"], case["p"], case["n"])
        assert got == case["coeffs"]
        assert oracle.fft_interpolate(case["zs"], case["ys"][0], case["omega"], case["p"], case["n"]) == case["coeffs"][0]


def test_golden_welch_berlekamp(golden):
    cases = golden("welch_berlekamp.json")["cases"]
    assert len(cases) > 100
    n_fail = 0
    for case in cases:
        res, status = oracle.wb_decode_batch(case["x"], case["k"], [case["word"]], case["p"])[0]
        if case["error"] is None:
            assert status == 0 and res == case["coeffs"], case
        else:
            n_fail += 1
            assert res is None and oracle.WB_MESSAGES[status] == case["error"], (status, case["error"])
    assert n_fail > 0  # the beyond-radius cases exercise the failure paths


def test_golden_welch_berlekamp_cfg4_shape(golden):
    """BASELINE config 4's shape pinned by the reference itself: n = 100, k = 34, make_wb_encoder_decoder at
    the full radius (33 errors), with erasures + errors, a stripped-zero result and one word beyond the radius ("No solution")"""
    cases = golden("welch_berlekamp_cfg4.json")["cases"]
    assert len(cases) >= 8 and {c["n"] for c in cases} == {100} and {c["k"] for c in cases} == {34}
    outcomes = set()
    for case in cases:
        res, status = oracle.wb_decode_batch(case["x"], case["k"], [case["word"]], case["p"])[0]
        if case["error"] is None:
            assert status == 0 and res == case["coeffs"]
            # inside the radius the result is the message with its trailing zeros stripped (polynomial.py:14-20)
            msg = list(case["msg"])
            while msg and msg[-1] == 0:
                msg.pop()
            assert res == msg
            if not any(w is None for w in case["word"]):
                co, err = oracle.gao_interpolate(case["x"], case["word"], case["k"], case["p"])
                assert co == case["coeffs"] + [0] * (case["k"] - len(case["coeffs"]))
                assert [i for i in range(100) if peval(err, case["x"][i], case["p"]) == 0] == case["errpos"]
        else:
            assert res is None and oracle.WB_MESSAGES[status] == case["error"]
        outcomes.add(case["error"])
    assert outcomes == {None, "No solution"}


def test_golden_welch_berlekamp_low_degree_messages(golden):
    """Words with MORE than floor((n - k) / 2) errors whose message has leading zeros: Gao's decoder still accepts many of them
    (deg f + e < (n + k) / 2), the reference's Welch-Berlekamp decoder -- whose outcomes these are -- does not.  The oracle's WB
    must follow the reference; its Gao must decode a fair share (that is what makes the fixture bite on a Gao shortcut)."""
    cases = golden("welch_berlekamp_low_degree.json")["cases"]
    assert len(cases) > 100
    gao_decodes = 0
    for case in cases:
        res, status = oracle.wb_decode_batch(case["x"], case["k"], [case["word"]], case["p"])[0]
        if case["error"] is None:
            assert status == 0 and res == case["coeffs"], case
        else:
            assert res is None and oracle.WB_MESSAGES[status] == case["error"], (status, case["error"])
        gao_decodes += oracle.gao_interpolate(case["x"], case["word"], case["k"], case["p"])[0] is not None
    assert gao_decodes >= 30


def test_golden_wb_vs_gao(golden):
    """Inside the decoding radius Gao must return what the reference's WB returns, and the
    roots of its error locator must be exactly the corrupted positions."""
    for case in golden("welch_berlekamp.json")["cases"]:
        if case["error"] is not None or case.get("beyond_radius"):
            continue
        p, x, k = case["p"], case["x"], case["k"]
        co, err = oracle.gao_interpolate(x, case["word"], k, p)
        want = case["coeffs"] + [0] * (k - len(case["coeffs"]))
        assert co == want
        roots = [i for i in range(len(x)) if case["word"][i] is not None and peval(err, x[i], p) == 0] if len(err) > 1 else []
        assert roots == case["errpos"]


def test_golden_gao_cofactor_from_the_references_polynomial_class(golden):
    """the coefficients AND the un-normalised cofactor of Gao's decoder, and its (None, None) decisions, as the
    reference's own Polynomial class produces them when it runs partial_gcd's recurrence (oracle/gen_golden.py section H;
    rsdecode_impl.h:281-363 over polynomial.py:85-108, 202-234) -- nothing of this repo took part in the expected values."""
    cases = golden("gao_cofactor.json")["cases"]
    assert len(cases) >= 50 and sum(c["coeffs"] is None for c in cases) >= 8
    for c in cases:
        co, v = oracle.gao_interpolate(c["x"], c["y"], c["k"], P)
        assert co == c["coeffs"], (c["kind"], c["k"], len(c["x"]))
        assert v == c["v"], (c["kind"], c["k"], len(c["x"]))


def test_word_size_open_equals_the_256_bit_restatement():
    """orc_batch_open_u64 (the open restated for p < 2^64 with 128-bit products: CPU baseline of bench.py --workload cfg3-p64) against
    orc_batch_open on the same inputs -- which the golden batch_reconstruct transcripts pin"""
    import random

    import numpy as np

    for p in ((1 << 64) - 59, 0xFFFFFFFF00000001, (1 << 61) - 1, 13):
        n, t = (16, 5) if p > 100 else (8, 2)
        d, b = t + 1, 100
        c = (b + d - 1) // d
        rnd = random.Random(p % 1000)
        x = list(range(1, n + 1))
        secrets = [rnd.randrange(p) for _ in range(b)]
        shares = [rnd.randrange(p) for _ in range(b)]
        padded = secrets + [0] * (c * d - b)
        enc = oracle.vandermonde_batch_evaluate(x, [padded[k * d:(k + 1) * d] for k in range(c)], p)
        cols = [enc[k][j] for j in range(n) for k in range(c)]
        z = rnd.sample(range(n), d)
        zc = [i for i in range(n) if i not in z][:t]
        rc, r1, msg, res = oracle.batch_open_u64(p, n, d, x, np.array(shares, dtype=np.uint64), np.array(cols, dtype=np.uint64), np.array(cols, dtype=np.uint64), z, zc)
        rc4, a1, a2, a3 = oracle.batch_open_limbs(p, n, d, x, oracle._limbs(shares, p), oracle._limbs(cols, p), oracle._limbs(cols, p), z, zc)
        assert rc == rc4 == 0 and res.tolist() == secrets
        assert oracle._ints(a1) == r1.tolist() and oracle._ints(a2) == msg.tolist() and oracle._ints(a3) == res.tolist()
        bad = list(cols)
        bad[zc[0] * c + 3] = (bad[zc[0] * c + 3] + 1) % p
        assert oracle.batch_open_u64(p, n, d, x, np.array(shares, dtype=np.uint64), np.array(cols, dtype=np.uint64), np.array(bad, dtype=np.uint64), z, zc)[0] == 2
