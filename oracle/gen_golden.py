#!/usr/bin/env python3
"""
Generate tests/golden/*.json by importing the REFERENCE's pure-Python layers.

Runs only in the build container (needs /root/reference, which never travels to the GPU
box); the committed JSON files are data: inputs and expected outputs.

What is imported from /root/reference/honeybadgermpc (never copied):
    field.py, polynomial.py, reed_solomon_wb.py      -- independent pure-Python arithmetic
    reed_solomon.py, batch_reconstruction.py, utils/misc.py, router.py -- host logic
How:
  * the package __init__ is NOT executed (it opens log files under /var/log, outside the
    repo): a bare module object with __path__ pointing at the reference is registered
    instead, so submodules import normally;
  * `gmpy2` (is_prime/mpz) and `pypairing` (names only) are absent from this image and
    are stubbed -- neither takes part in any value written below;
  * `honeybadgermpc.ntl` is the NTL/Cython extension, which cannot be built here (no NTL).
    Sections A-E below do not touch it at all: they use the reference's own Python
    Polynomial / fft_helper / fnt_decode / Welch-Berlekamp code, an implementation of
    the same maths that is independent of everything in this repo.  Sections F-G
    (IncrementalDecoder and batch_reconstruct transcripts) exercise the reference's HOST
    LOGIC and need some arithmetic backend behind `honeybadgermpc.ntl`; the oracle is
    plugged in there.  All values crossing that boundary are canonical residues, and the
    oracle is itself checked against sections A-E, so the transcripts are pinned by the
    reference's logic, not by the oracle's.

Usage:  python oracle/gen_golden.py     (rewrites tests/golden/)
"""
import asyncio
import json
import os
import random
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)

import oracle  # noqa: E402


def install_reference():
    from sympy import isprime

    gmpy2 = types.ModuleType("gmpy2")
    gmpy2.is_prime = lambda n: bool(isprime(int(n)))
    gmpy2.mpz = int
    sys.modules["gmpy2"] = gmpy2
    pp = types.ModuleType("pypairing")
    for name in ("PyFq", "PyFq2", "PyFq12", "PyFqRepr", "PyFr", "PyG1", "PyG2"):
        setattr(pp, name, type(name, (), {}))
    sys.modules["pypairing"] = pp
    pkg = types.ModuleType("honeybadgermpc")
    pkg.__path__ = [os.path.join(REF, "honeybadgermpc")]
    sys.modules["honeybadgermpc"] = pkg
    ntl = types.ModuleType("honeybadgermpc.ntl")
    for name in (
        "lagrange_interpolate", "evaluate", "vandermonde_inverse", "InterpolationError",
        "vandermonde_batch_interpolate", "vandermonde_batch_evaluate", "fft", "partial_fft",
        "fft_batch_evaluate", "fft_interpolate", "fft_batch_interpolate", "gao_interpolate",
        "sqrt_mod", "SetNTLNumThreads", "AvailableNTLThreads", "SetNumThreads", "GetMaxThreads",
    ):
        setattr(ntl, name, getattr(oracle, name))
    sys.modules["honeybadgermpc.ntl"] = ntl
    pkg.ntl = ntl


install_reference()
from honeybadgermpc.field import GF  # noqa: E402
from honeybadgermpc.polynomial import EvalPoint, fnt_decode_step1, fnt_decode_step2, get_omega, polynomials_over  # noqa: E402
from honeybadgermpc.reed_solomon_wb import make_wb_encoder_decoder  # noqa: E402
from honeybadgermpc.utils.misc import chunk_data, flatten_lists, transpose_lists  # noqa: E402
import honeybadgermpc.reed_solomon as ref_rs  # noqa: E402
from honeybadgermpc.batch_reconstruction import batch_reconstruct  # noqa: E402
from honeybadgermpc.router import SimpleRouter  # noqa: E402

BLS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def S(v):
    """ints -> decimal strings (JSON has no bigints), recursively; None stays None"""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, int):
        return str(v)
    if isinstance(v, (list, tuple)):
        return [S(x) for x in v]
    if isinstance(v, dict):
        return {k: S(x) for k, x in v.items()}
    if hasattr(v, "value"):
        return str(v.value)
    return v


def dump(name, obj):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    with open(path, "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print(f"wrote {path} ({os.path.getsize(path)} bytes)")


# --------------------------------------------------------------------------- A
def gen_constants():
    fp = GF(BLS)
    out = {"modulus": str(BLS), "seed0_random": str(fp.random(0).value), "omega": {}, "evalpoint": []}
    for order in (2, 4, 8, 16, 32, 64, 128, 256, 512, 1024):
        out["omega"][str(order)] = str(get_omega(fp, order, seed=0).value)
    for n in (4, 7, 16, 22, 64, 100):
        pt = EvalPoint(fp, n, use_omega_powers=True)
        out["evalpoint"].append({
            "n": n, "order": pt.order, "omega": str(pt.omega.value), "omega2": str(pt.omega2.value),
            "points": [str(pt(i).value) for i in range(n)],
        })
    dump("constants.json", out)


# --------------------------------------------------------------------------- B
def gen_vandermonde():
    """Polynomial.__call__ / Polynomial.interpolate -> vandermonde_batch_evaluate /
    vandermonde_batch_interpolate / lagrange_interpolate / evaluate"""
    rnd = random.Random(1001)
    cases = []
    for p, n, d, c in [(13, 4, 2, 5), (53, 22, 8, 6), (BLS, 4, 2, 8), (BLS, 16, 6, 6), (BLS, 22, 8, 4), (BLS, 64, 22, 3), (BLS, 100, 34, 2)]:
        fp = GF(p)
        poly = polynomials_over(fp)
        x = list(range(1, n + 1))
        polys = [[rnd.randrange(p) for _ in range(d)] for _ in range(c)]
        polys[0] = [0] * d                        # zero polynomial
        polys[-1] = polys[-1][: d - 1] + [0]      # top coefficient zero
        evals = [[poly(pl)(fp(xi)).value for xi in x] for pl in polys]
        # interpolate back from a shuffled subset of d points with the reference's Lagrange code
        idx = list(range(n))
        rnd.shuffle(idx)
        zsel = idx[:d]
        interp = []
        for row in evals:
            pts = [(fp(x[z]), fp(row[z])) for z in zsel]
            co = [v.value for v in poly.interpolate(pts).coeffs]
            interp.append(co + [0] * (d - len(co)))
        cases.append({"p": p, "x": x, "polys": polys, "evals": evals, "z": zsel, "interp": interp})
    # evaluation at arbitrary (large) points
    fp = GF(BLS)
    poly = polynomials_over(fp)
    coeffs = [1, 2, 3, 4]
    xs = [rnd.randrange(BLS) for _ in range(10)]
    ev = {"p": BLS, "coeffs": coeffs, "xs": xs, "ys": [poly(coeffs)(fp(xv)).value for xv in xs]}
    dump("vandermonde.json", S({"cases": cases, "evaluate": ev}))


# --------------------------------------------------------------------------- C
def gen_fft():
    """Polynomial.evaluate_fft (fft_helper) -> fft / partial_fft / fft_batch_evaluate"""
    rnd = random.Random(1002)
    cases = []
    f13 = GF(13)
    poly13 = polynomials_over(f13)
    cases.append({"p": 13, "omega": 5, "n": 4, "coeffs": [[0, 1]], "evals": [[v.value for v in poly13([0, 1]).evaluate_fft(f13(5), 4)]]})
    fp = GF(BLS)
    poly = polynomials_over(fp)
    for n, d, c in [(2, 1, 2), (4, 2, 4), (8, 3, 3), (16, 6, 6), (16, 16, 2), (32, 20, 3), (64, 22, 3), (128, 34, 2), (256, 86, 1), (512, 5, 1)]:
        omega = get_omega(fp, n, seed=0)
        coeffs = [[rnd.randrange(BLS) for _ in range(d)] for _ in range(c)]
        coeffs[0] = [0] * d
        evals = []
        for row in coeffs:
            stripped = poly(row)
            evals.append([v.value for v in stripped.evaluate_fft(omega, n)])
        cases.append({"p": BLS, "omega": omega.value, "n": n, "coeffs": coeffs, "evals": evals})
    dump("fft.json", S({"cases": cases}))


# --------------------------------------------------------------------------- D
def gen_fft_interpolate():
    """reference Python fnt_decode_step1/2 (and Lagrange) -> fft_interpolate / fft_batch_interpolate"""
    rnd = random.Random(1003)
    fp = GF(BLS)
    poly = polynomials_over(fp)
    cases = []
    for n, k, c in [(4, 2, 3), (8, 3, 3), (16, 6, 4), (16, 16, 2), (32, 8, 2), (64, 22, 2), (128, 34, 1)]:
        omega2 = get_omega(fp, 2 * n, seed=0)
        omega = omega2 ** 2
        zs = list(range(n))
        rnd.shuffle(zs)
        zs = zs[:k]
        as_, ais_ = fnt_decode_step1(poly, zs, omega2, n)
        ys_list, out = [], []
        for _ in range(c):
            ys = [rnd.randrange(BLS) for _ in range(k)]
            prec = fnt_decode_step2(poly, zs, [fp(y) for y in ys], as_, ais_, omega2, n)
            co = [v.value for v in prec.coeffs]
            co = co + [0] * (k - len(co))
            # cross-check inside the reference: Lagrange must agree with its FNT code
            lag = poly.interpolate([(omega ** z, fp(y)) for z, y in zip(zs, ys)])
            lagc = [v.value for v in lag.coeffs]
            assert lagc + [0] * (k - len(lagc)) == co
            ys_list.append(ys)
            out.append(co)
        cases.append({"p": BLS, "omega": omega.value, "n": n, "zs": zs, "ys": ys_list, "coeffs": out})
    dump("fft_interpolate.json", S({"cases": cases}))


# --------------------------------------------------------------------------- E
def corrupt(rnd, message, num_errors, num_nones, p):
    message = list(message)
    indices = rnd.sample(range(len(message)), num_errors + num_nones)
    for i in range(num_errors):
        old = message[indices[i]]
        new = rnd.randrange(p)
        while new == old:
            new = rnd.randrange(p)
        message[indices[i]] = new
    for i in range(num_nones):
        message[indices[i + num_errors]] = None
    return message, sorted(indices[:num_errors])


def gen_wb():
    """reference pure-Python Welch-Berlekamp -> wb_decode (and an independent check for Gao)"""
    rnd = random.Random(1004)
    cases = []
    for p, n, k, reps in [(53, 22, 8, 6), (BLS, 4, 2, 4), (BLS, 7, 3, 4), (BLS, 16, 6, 4), (BLS, 22, 8, 3), (BLS, 31, 11, 2)]:
        enc, dec, _ = make_wb_encoder_decoder(n, k, p)
        t = k - 1
        x = list(range(1, n + 1))
        cmax, emax = n - 2 * t - 1, (n - 2 * t - 1) // 2
        msgs = [[rnd.randrange(p) for _ in range(k)] for _ in range(reps)]
        msgs[0] = [0] * k
        if p == 53:
            msgs[1] = [2, 3, 2, 8, 7, 5, 9, 5]   # tests/test_reed_solomon_wb.py:7
        msgs[-1] = msgs[-1][: k - 2] + [0, 0]     # stripped output shorter than k
        for msg in msgs:
            encoded = [v.value for v in enc(msg)]
            for ne, nn in [(0, 0), (0, cmax), (emax, 0), (emax // 2, cmax // 4), (1, 0), (emax, cmax - 2 * emax)]:
                if ne + nn > n or 2 * ne + nn > cmax:
                    continue
                word, errpos = corrupt(rnd, encoded, ne, nn, p)
                fpw = GF(p)
                try:
                    out = dec([None if w is None else fpw(w) for w in word], debug=False)
                    res = {"coeffs": [c.value for c in out], "error": None}
                except Exception as e:  # noqa: BLE001 - the reference raises bare Exceptions
                    res = {"coeffs": None, "error": str(e)}
                cases.append({"p": p, "n": n, "k": k, "x": x, "msg": msg, "word": word, "errpos": errpos, **res})
            # beyond the decoding radius: emax + 1 errors, no erasures
            if emax + 1 <= n:
                word, errpos = corrupt(rnd, encoded, emax + 1, 0, p)
                fpw = GF(p)
                try:
                    out = dec([fpw(w) for w in word], debug=False)
                    res = {"coeffs": [c.value for c in out], "error": None}
                except Exception as e:  # noqa: BLE001
                    res = {"coeffs": None, "error": str(e)}
                cases.append({"p": p, "n": n, "k": k, "x": x, "msg": msg, "word": word, "errpos": errpos, "beyond_radius": True, **res})
        # hopeless inputs: every symbol random (with and without erasures) -> the failure paths
        for nn in (0, cmax):
            for _ in range(3):
                word = [rnd.randrange(p) for _ in range(n)]
                for i in rnd.sample(range(n), nn):
                    word[i] = None
                fpw = GF(p)
                try:
                    out = dec([None if w is None else fpw(w) for w in word], debug=False)
                    res = {"coeffs": [c.value for c in out], "error": None}
                except Exception as e:  # noqa: BLE001
                    res = {"coeffs": None, "error": str(e)}
                cases.append({"p": p, "n": n, "k": k, "x": x, "msg": None, "word": word, "errpos": None, "beyond_radius": True, **res})
    dump("welch_berlekamp.json", S({"cases": cases}))


def gen_wb_cfg4():
    """BASELINE config 4's shape from the reference itself: n = 100, k = 34, the reference's own
    make_wb_encoder_decoder at the full radius (33 errors), beyond it (34), and with erasures + errors.  ~2.3 s per row
    reduction in the reference's pure Python, and a word beyond the radius walks the whole descending-e' loop: a few minutes."""
    rnd = random.Random(4100)
    p, n, k = BLS, 100, 34
    enc, dec, _ = make_wb_encoder_decoder(n, k, p)
    x = list(range(1, n + 1))
    cases = []
    fpw = GF(p)
    for ne, nn in [(33, 0), (33, 0), (33, 0), (32, 1), (28, 10), (28, 10), (23, 20), (34, 0)]:
        msg = [rnd.randrange(p) for _ in range(k)]
        if (ne, nn) == (32, 1):
            msg = msg[: k - 2] + [0, 0]                     # stripped output shorter than k
        encoded = [v.value for v in enc(msg)]
        word, errpos = corrupt(rnd, encoded, ne, nn, p)
        try:
            out = dec([None if w is None else fpw(w) for w in word], debug=False)
            res = {"coeffs": [c.value for c in out], "error": None}
        except Exception as e:  # noqa: BLE001 - the reference raises bare Exceptions
            res = {"coeffs": None, "error": str(e)}
        cases.append({"p": p, "n": n, "k": k, "x": x, "msg": msg, "word": word, "errpos": errpos,
                      "beyond_radius": 2 * ne + nn > n - k, **res})
    dump("welch_berlekamp_cfg4.json", S({"cases": cases}))


# --------------------------------------------------------------------------- F
def gen_wb_low_degree():
    """Words Gao's decoder accepts BEYOND floor((n - k) / 2) errors -- the message has leading zeros, so deg f + e stays below
    (n + k) / 2 -- judged by the reference's own Welch-Berlekamp decoder: out there its descending-e' loop and the particular
    solution it takes decide (n = 25, k = 4, a constant message, 11 errors: "found no divisors!"), and a batched decoder that
    shortcuts through Gao must not take Gao's word for them.  Small shapes: the reference's pure Python runs them in seconds."""
    rnd = random.Random(1414)
    cases = []
    for p, n, k in [(53, 25, 4), (53, 22, 8), (257, 16, 5), (BLS, 13, 3), (BLS, 25, 4), (BLS, 20, 7)]:
        enc, dec, _ = make_wb_encoder_decoder(n, k, p)
        x = list(range(1, n + 1))
        emax = (n - k) // 2
        fpw = GF(p)
        for keep in range(0, k):                                  # the message's degree + 1
            for extra in (1, 2, 3):
                ne = emax + extra
                if keep + ne >= (n + k + 1) // 2 + 1 or ne > n:    # (a little past what Gao still decodes, too)
                    continue
                for _ in range(2):
                    msg = [rnd.randrange(1, p) for _ in range(keep)] + [0] * (k - keep)
                    encoded = [v.value for v in enc(msg)]
                    word, errpos = corrupt(rnd, encoded, ne, 0, p)
                    try:
                        out = dec([fpw(w) for w in word], debug=False)
                        res = {"coeffs": [c.value for c in out], "error": None}
                    except Exception as e:  # noqa: BLE001 - the reference raises bare Exceptions
                        res = {"coeffs": None, "error": str(e)}
                    cases.append({"p": p, "n": n, "k": k, "x": x, "msg": msg, "word": word, "errpos": errpos, "beyond_radius": True, **res})
    # the word scratch/stress_gao.py met first
    p, n, k = 53, 25, 4
    enc, dec, _ = make_wb_encoder_decoder(n, k, p)
    word = [11, 11, 11, 11, 11, 11, 47, 11, 11, 51, 43, 13, 11, 36, 11, 21, 11, 11, 38, 13, 10, 38, 11, 11, 51]
    fpw = GF(p)
    try:
        out = dec([fpw(w) for w in word], debug=False)
        res = {"coeffs": [c.value for c in out], "error": None}
    except Exception as e:  # noqa: BLE001
        res = {"coeffs": None, "error": str(e)}
    cases.append({"p": p, "n": n, "k": k, "x": list(range(1, n + 1)), "msg": [11, 0, 0, 0], "word": word, "errpos": None, "beyond_radius": True, **res})
    # t = 0 with a single symbol present: the decoder's e = 0 branch interpolates (reed_solomon_wb.py:142-145), and the reference's
    # Polynomial returns the ZERO polynomial of that branch as [0] (strip_trailing_zeros keeps one zero of an all-zero list,
    # polynomial.py:14-20) -- oracle/diff_wb_vs_reference.py met it
    for p, n in [(53, 4), (257, 3), (BLS, 5)]:
        enc, dec, _ = make_wb_encoder_decoder(n, 1, p)
        fpw = GF(p)
        for val in (0, 7):
            for pos in (0, n - 1):
                word = [None] * n
                word[pos] = val
                try:
                    out = dec([None if w is None else fpw(w) for w in word], debug=False)
                    res = {"coeffs": [c.value for c in out], "error": None}
                except Exception as e:  # noqa: BLE001
                    res = {"coeffs": None, "error": str(e)}
                cases.append({"p": p, "n": n, "k": 1, "x": list(range(1, n + 1)), "msg": [val], "word": word, "errpos": [], "beyond_radius": False, **res})
    dump("welch_berlekamp_low_degree.json", S({"cases": cases}))


def gen_misc():
    out = {
        "chunk_data": [
            {"data": d, "size": s, "out": chunk_data(list(d), s)}
            for d, s in [([1, 2, 3, 4, 5], 2), ([], 2), ([1, 2, 3, 4], 2), ([7], 3), ([], 1)]
        ],
        "transpose_lists": [{"in": m, "out": transpose_lists(m)} for m in [[[1, 2, 3], [4, 5, 6]], [[1]], [[1, 2], [3, 4], [5, 6]]]],
        "flatten_lists": [{"in": m, "out": flatten_lists(m)} for m in [[[1, 2, 3], [4, 5, 6]], [[], [1]], []]],
    }
    dump("misc.json", out)


def gen_incremental():
    """IncrementalDecoder transcripts from the reference's state machine (reed_solomon.py:232-403)."""
    rnd = random.Random(1005)
    transcripts = []
    for p, n, t, batch, use_omega, robust, bad in [
        (BLS, 4, 1, 3, False, "gao", []), (BLS, 4, 1, 3, False, "gao", [1]), (BLS, 4, 1, 2, True, "gao", [2]),
        (BLS, 7, 2, 4, False, "gao", [0, 5]), (BLS, 7, 2, 4, False, "welch-berlekamp", [3]),
        (BLS, 16, 5, 3, False, "gao", [1, 4, 9, 12, 15]), (BLS, 16, 5, 3, True, "gao", [2, 3]),
        (BLS, 16, 5, 2, False, "welch-berlekamp", [7, 8]), (53, 10, 3, 3, False, "gao", [4]),
    ]:
        fp = GF(p)
        point = EvalPoint(fp, n, use_omega_powers=use_omega)
        algo = ref_rs.Algorithm.FFT if use_omega else ref_rs.Algorithm.VANDERMONDE
        enc = ref_rs.EncoderFactory.get(point, algo)
        dec = ref_rs.DecoderFactory.get(point, algo)
        rdec = ref_rs.RobustDecoderFactory.get(t, point, algorithm=robust)
        msgs = [[rnd.randrange(p) for _ in range(t + 1)] for _ in range(batch)]
        encoded = enc.encode(msgs)
        columns = [[encoded[b][j] for b in range(batch)] for j in range(n)]
        for j in bad:
            columns[j] = [rnd.randrange(p) for _ in range(batch)]
        order = list(range(n))
        rnd.shuffle(order)
        inc = ref_rs.IncrementalDecoder(enc, dec, rdec, degree=t, batch_size=batch, max_errors=t)
        steps = []
        for idx in order:
            inc.add(idx, columns[idx])
            res, errs = inc.get_results()
            steps.append({"idx": idx, "done": inc.done(), "result": res, "errors": None if errs is None else sorted(errs)})
            if inc.done():
                break
        transcripts.append({
            "p": p, "n": n, "t": t, "batch": batch, "use_omega_powers": use_omega, "robust": robust,
            "msgs": msgs, "columns": columns, "order": order, "steps": steps,
        })
    dump("incremental_decoder.json", S({"transcripts": transcripts}))


# --------------------------------------------------------------------------- G
class _Cfg:
    def __init__(self, algo):
        self.induce_faults = False
        self.decoding_algorithm = algo


def gen_batch_reconstruct():
    """Full batch_reconstruct runs of the reference over its SimpleRouter: inputs, every
    R1/R2 message each party sent, and outputs."""
    rnd = random.Random(1006)
    runs = []
    for p, n, t, b, use_omega, robust, bad in [
        (BLS, 4, 1, 3, False, "gao", []), (BLS, 4, 1, 3, False, "gao", [1]), (BLS, 4, 1, 2, True, "gao", []),
        (BLS, 4, 1, 2, True, "gao", [1]), (BLS, 7, 2, 10, False, "gao", [0, 6]), (BLS, 7, 2, 5, False, "welch-berlekamp", [2]),
        (BLS, 16, 5, 20, False, "gao", []), (BLS, 16, 5, 13, True, "gao", [3, 8]),
    ]:
        fp = GF(p)
        poly = polynomials_over(fp)
        point = EvalPoint(fp, n, use_omega_powers=use_omega)
        secrets = [rnd.randrange(p) for _ in range(b)]
        polys = [poly([s] + [rnd.randrange(p) for _ in range(t)]) for s in secrets]
        shares = [[polys[j](point(i)).value for j in range(b)] for i in range(n)]
        for i in bad:
            shares[i] = [rnd.randrange(p) for _ in range(b)]
        sent = [{"R1": [None] * n, "R2": None} for _ in range(n)]

        async def go():
            router = SimpleRouter(n)
            tasks = []
            for i in range(n):
                def mk(i):
                    base = router.sends[i]

                    def send(dest, msg):
                        tag, payload = msg
                        if tag == "R1":
                            sent[i]["R1"][dest] = list(payload)
                        else:
                            sent[i]["R2"] = list(payload)
                        base(dest, msg)
                    return send
                tasks.append(batch_reconstruct(
                    [fp(v) for v in shares[i]], p, t, n, i, mk(i), router.recvs[i],
                    config=_Cfg(robust), use_omega_powers=use_omega))
            return await asyncio.gather(*tasks)

        results = asyncio.run(go())
        outs = [None if r is None else [v.value for v in r] for r in results]
        for i in range(n):
            if i not in bad:
                pass
        runs.append({
            "p": p, "n": n, "t": t, "use_omega_powers": use_omega, "robust": robust, "bad": bad,
            "secrets": secrets, "shares": shares, "sent": sent, "outputs": outs,
        })
    dump("batch_reconstruct.json", S({"runs": runs}))


# --------------------------------------------------------------------------- H
def reference_polynomial_gao(p):
    """-> gao(xs, ys, k): partial_gcd's recurrence and gao_interpolate's acceptance test (rsdecode_impl.h:281-363) executed with the
    REFERENCE's own Polynomial class over GF(p) (used by gen_gao_cofactor and by oracle/diff_gao_vs_reference_polynomial.py)"""
    fp = GF(p)
    poly = polynomials_over(fp)
    one, zero = poly([fp(1)]), poly([])

    def deg(p_):
        return -1 if p_.is_zero() else p_.degree()          # NTL's deg(0) = -1; the reference's Python class keeps [0] for zero (degree 0)

    def divrem(a, b):
        """NTL's DivRem with the reference's Polynomial arithmetic; a constant divisor is handled here (the class's own __divmod__
        does not terminate on it: its zero remainder has degree 0 >= 0)"""
        if deg(b) == 0:
            return a * poly([fp(1) / b.coeffs[0]]), zero
        if deg(a) < deg(b):
            return zero, a
        return divmod(a, b)

    def gao(xs, ys, k):
        pts = [(fp(x), fp(y)) for x, y in zip(xs, ys) if y is not None]          # hbmpc_ntl_helpers.pyx:399-403 drops erasures
        n = len(pts)
        g0 = one
        for x, _ in pts:
            g0 = g0 * poly([fp(0) - x, fp(1)])
        g1 = poly.interpolate(pts)
        thr = (n + k) // 2
        r0, r1, t0, t1 = g0, g1, zero, one
        if deg(r0) < thr:
            r, v = r0, t0
        elif deg(r1) < thr:
            r, v = r1, t1
        else:
            while True:
                q, r2 = divrem(r0, r1)
                t2 = t0 - q * t1
                if deg(r2) < thr:
                    r, v = r2, t2
                    break
                r0, r1, t0, t1 = r1, r2, t1, t2
        f1, rem = divrem(r, v)
        if not rem.is_zero() or deg(f1) >= k:
            return None, None
        fc = [] if f1.is_zero() else [c.value for c in f1.coeffs]
        coeffs = fc + [0] * (k - len(fc))                                         # exactly k, zero padded (rsdecode_impl.h:351-354)
        return coeffs, [c.value for c in v.coeffs]

    return gao


def gen_gao_cofactor():
    """Gao's outputs pinned WITHOUT the oracle: the recurrence of partial_gcd and the acceptance
    test of gao_interpolate (rsdecode_impl.h:281-363) executed with the REFERENCE's own Polynomial class -- its interpolate, __mul__,
    __sub__ and __divmod__ (polynomial.py:85-108, 202-234).  What is written: for each word (points, values with None = erasure, k)
    the coefficient list and the un-normalised cofactor v = t_i exactly as the recurrence leaves it, or null / null where the
    reference returns (None, None).  Words inside, at and beyond the unique-decoding radius, with and without erasures."""
    rnd = random.Random(20240928)
    fp = GF(BLS)
    poly = polynomials_over(fp)
    gao = reference_polynomial_gao(BLS)
    cases = []
    shapes = [(7, 3), (10, 4), (16, 6), (16, 6), (22, 8), (31, 11)]
    for n, k in shapes:
        xs = list(range(1, n + 1)) if rnd.random() < 0.6 else rnd.sample(range(1, 200), n)
        for kind in ("none", "one", "half", "radius", "radius", "beyond", "beyond2", "erasures", "erasures+errors"):
            msg = [rnd.randrange(BLS) for _ in range(k)]
            if kind == "none" and rnd.random() < 0.3:
                msg[-1] = 0                                                       # a message of lower degree
            ys = [int(poly([fp(c) for c in msg])(fp(x)).value) for x in xs]
            n_er = 0
            if kind.startswith("erasures"):
                n_er = rnd.randrange(1, max(2, (n - k) // 2))
                for pos in rnd.sample(range(n), n_er):
                    ys[pos] = None
            radius = (n - n_er - k) // 2
            n_err = {"none": 0, "one": min(1, radius), "half": radius // 2, "radius": radius, "beyond": radius + 1, "beyond2": radius + 2,
                     "erasures": 0, "erasures+errors": radius}[kind]
            live = [i for i in range(n) if ys[i] is not None]
            for pos in rnd.sample(live, min(n_err, len(live))):
                ys[pos] = (ys[pos] + 1 + rnd.randrange(BLS - 1)) % BLS
            coeffs, v = gao(xs, ys, k)
            cases.append({"x": xs, "y": ys, "k": k, "kind": kind, "errors": n_err, "erasures": n_er, "coeffs": coeffs, "v": v})
    ok = sum(1 for c in cases if c["coeffs"] is not None)
    print(f"gao cofactor: {len(cases)} words, {ok} decode, {len(cases) - ok} do not")
    dump("gao_cofactor.json", S({"cases": cases}))


if __name__ == "__main__":
    gen_constants()
    gen_vandermonde()
    gen_fft()
    gen_fft_interpolate()
    gen_wb()
    gen_wb_cfg4()
    gen_wb_low_degree()
    gen_misc()
    gen_incremental()
    gen_batch_reconstruct()
    gen_gao_cofactor()
