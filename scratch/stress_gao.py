"""Randomised differential run of the batched robust decoders (hb_gao_decode, hb_wb_decode) against the oracle:
    python scratch/stress_gao.py <seconds> <seed>
Random fields (BLS12-381 r, moduli at the top of the range, tiny fields where vanishing remainders / quotient digits are common, a
narrow context), random (n, k), words with 0 .. beyond-the-radius errors, special messages (zero, constants, short).  The oracle
(oracle/hbmpc_oracle.c) is the checker; nothing here is timed."""
import random
import sys
import time

sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import oracle  # noqa: E402
from structured import coordinated_errors, structured_message  # noqa: E402  (the generators the -m gpu suites use)
from honeybadgermpc_amd import ntl  # noqa: E402
from honeybadgermpc_amd.device import wb_decode_batch  # noqa: E402

BLS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
FIELDS = [BLS, BLS, (1 << 256) - 189, (1 << 255) - 19, (1 << 254) + 79, 13, 53, 257, 65537, (1 << 61) - 1, (1 << 64) - 59]
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
t_end = time.time() + seconds
batches = words_total = refused = 0
while time.time() < t_end:
    p = rnd.choice(FIELDS)
    n = rnd.randrange(2, min(p - 1, 100) + 1)
    k = rnd.randrange(1, n + 1)
    x = rnd.sample(range(1, min(p, 4 * n + 2)), n) if rnd.random() < 0.3 else list(range(1, n + 1))
    emax = (n - k) // 2
    words = []
    for _ in range(rnd.choice([1, 7, 64, 200])):
        msg = structured_message(rnd, k, p)
        enc = oracle.vandermonde_batch_evaluate(x, [msg], p)[0]
        ne = min(n, rnd.choice([0, emax, rnd.randrange(emax + 1), emax + 1, emax + 2, rnd.randrange(n + 1)]))
        if rnd.random() < 0.25:
            enc = coordinated_errors(rnd, enc, x, k, ne, p, lambda xs, cf: oracle.vandermonde_batch_evaluate(xs, [cf], p)[0])[0]
        else:
            for i in rnd.sample(range(n), ne):
                enc[i] = (enc[i] + rnd.randrange(1, p)) % p
        words.append(enc)
    got = ntl.gao_interpolate_batch(x, words, k, p)
    want = oracle.gao_interpolate_batch(x, words, k, p)
    if got != want:
        bad = next(i for i in range(len(words)) if got[i] != want[i])
        print(f"stress_gao: MISMATCH (gao) p={p} n={n} k={k} x={x} word={words[bad]} got={got[bad]} want={want[bad]}")
        sys.exit(1)
    if n <= 40 or len(words) <= 7:               # (the oracle's Welch-Berlekamp is cubic in n)
        cmax = n - 2 * (k - 1) - 1                # erasures the reference's decoder admits (reed_solomon_wb.py:131)
        if cmax > 0 and rnd.random() < 0.5:       # ... in half of the batches some words lose symbols (the row reduction takes those throughout)
            words = [list(w) for w in words]
            for w in words:
                if rnd.random() < 0.6:
                    for i in rnd.sample(range(n), rnd.randrange(1, cmax + 1)):
                        w[i] = None
        gw = wb_decode_batch(x, k, words, p)
        ww = oracle.wb_decode_batch(x, k, words, p)
        if gw != ww:
            bad = next(i for i in range(len(words)) if gw[i] != ww[i])
            print(f"stress_gao: MISMATCH (wb) p={p} n={n} k={k} x={x} word={words[bad]} got={gw[bad]} want={ww[bad]}")
            sys.exit(1)
    batches += 1
    words_total += len(words)
    refused += sum(1 for g in got if g[0] is None)
print(f"stress_gao: {batches} batches, {words_total} words through hb_gao_decode and hb_wb_decode ({refused} beyond the radius), 0 differences from the oracle (seed {seed}, {seconds:.0f} s)")
