"""Big-integer model of the reductions of k_mm8w (gen_mm8w.py reduce_output) and k_mm8 (hb_mfma.hip, epilogue) with the high half of a
sum folded on the matrix cores: the arithmetic, the representatives t_b and every bound the kernels rely on, as assertions.
CPU only; tests/test_fold_model.py runs it, `python tests/fold_model.py` runs the long version."""
import random

def balanced_digits(t, n=32):
    """n balanced base-256 digits of t (may be negative); None if it does not fit"""
    out = []
    for _ in range(n):
        d = t & 0xff
        if d > 127:
            d -= 256
        out.append(d)
        t = (t - d) >> 8
    return out if t == 0 else None

def tables(p):
    s = []
    tsum = 0
    for b in range(32):
        T = pow(2, 256 + 8 * b, p)
        dg = balanced_digits(T)
        if dg is None:
            dg = balanced_digits(T - p)
        assert dg is not None, "no 32-digit representative"
        s.append(dg)
        tsum += T
    mu = (1 << 286) // p
    assert mu < 1 << 32
    c512 = pow(2, 512, p)
    btot = sum((1 << 20) << (8 * e) for e in range(32))
    return s, mu, c512, btot, tsum

def reduce_model(S, CRorig, p, tb):
    s, mu, c512, btot, tsum = tb
    assert S < 1 << 527
    K2 = (128 * tsum - btot) % p
    CR = (CRorig + K2) % p
    W = [(S >> (32 * j)) & 0xffffffff for j in range(17)]
    h = [((S >> (256 + 8 * b)) & 0xff) - 128 for b in range(32)]
    D = [sum(h[b] * s[b][e] for b in range(32)) for e in range(32)]
    assert all(abs(x) < 1 << 20 for x in D)
    bias4 = (1 << 20) * 0x01010101
    P = []
    for w in range(8):
        v = bias4 + ((CR >> (32 * w)) & 0xffffffff)
        for k in range(4):
            v += D[4 * w + k] << (8 * k)
        v += W[w] + W[16] * ((c512 >> (32 * w)) & 0xffffffff)
        assert 0 <= v < 1 << 49
        P.append(v)
    R = sum(P[w] << (32 * w) for w in range(8))
    assert R % p == (S + CRorig) % p
    t = (P[7] + (P[6] >> 32))
    rtop = (t >> 16)
    assert rtop < 1 << 32
    q = (rtop * mu) >> 46
    assert q in (R // p, R // p - 1), (q, R // p)
    pneg = (1 << 256) - p
    u = [q * ((pneg >> (32 * w)) & 0xffffffff) + P[w] for w in range(8)]
    assert all(x < 1 << 64 for x in u)
    rp = sum(u[w] << (32 * w) for w in range(8))
    r = rp - (q << 256)
    assert 0 <= r < 2 * p
    top = (rp >> 256) - q
    assert top in (0, 1) and top == r >> 256
    if r >= p:
        r -= p
    assert r == (S + CRorig) % p
    return r

PRIMES = [0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
          0xfffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141,
          0xfffffffffffffffffffffffffffffffffffffffffffffffffffffffefffffc2f,
          (1 << 255) - 19, (1 << 254) + 0x4f]   # the last is not prime: the arithmetic does not care


def run_wide(iterations, seed=1):
    rng = random.Random(seed)
    for p in PRIMES:
        tb = tables(p)
        for it in range(iterations):
            bits = rng.choice([527, 526, 523, 517, 300, 256, 10])
            S = rng.getrandbits(bits)
            if it % 7 == 0:
                S = (1 << 527) - 1 - rng.getrandbits(40)
            if it % 11 == 0:
                S = rng.getrandbits(15) << 512 | ((1 << 512) - 1)
            reduce_model(S, rng.randrange(p), p, tb)


def mm8_model(cols, CRorig, p, rng):
    """k_mm8's epilogue (hb_mfma.hip, round 3): 47 non-negative columns < 2 * 5.8e6; the low eight G_k go into P_w as they are, the
    high four make the exact words H (one byte half of the fold table), the word above H times 2^384 mod p."""
    assert len(cols) == 47 and all(0 <= c < 2 * 5800000 for c in cols)
    S = sum(c << (8 * i) for i, c in enumerate(cols))
    # the table with 16 bytes
    s, tsum = [], 0
    for b in range(16):
        T = pow(2, 256 + 8 * b, p)
        dg = balanced_digits(T) or balanced_digits(T - p)
        s.append(dg); tsum += T
    mu = (1 << 286) // p
    c384 = pow(2, 384, p)
    btot = sum((1 << 20) << (8 * e) for e in range(32))
    K2 = (128 * tsum - btot) % p
    CR = (CRorig + K2) % p
    # pairs: E_j = col_{4j+3} + col_{4j+4} 2^8 at bit 32 j + 24, F_k = col_{4k+1} + col_{4k+2} 2^8 at bit 32 k + 8, col_0 at bit 0
    G = []
    for k in range(12):
        f = cols[4 * k + 1] + (cols[4 * k + 2] << 8) if 4 * k + 2 < 47 else cols[4 * k + 1]
        assert f < 1 << 32
        g = f * 256 + (cols[0] if k == 0 else 0)
        if k < 11:
            e = cols[4 * k + 3] + (cols[4 * k + 4] << 8)
            assert e < 1 << 32
            g += e << 24
        assert g < 1 << 57
        G.append(g)
    assert sum(g << (32 * k) for k, g in enumerate(G)) == S
    # exact high words from G_8.. and the spill of G_7
    Hs = (G[7] >> 32) + sum(G[k] << (32 * (k - 8)) for k in range(8, 12))
    hw = [(Hs >> (32 * j)) & 0xffffffff for j in range(5)]
    assert Hs >> 160 == 0 and hw[4] < 1 << 9
    h = [((Hs >> (8 * b)) & 0xff) - 128 for b in range(16)]
    D = [sum(h[b] * s[b][e] for b in range(16)) for e in range(32)]
    bias4 = (1 << 20) * 0x01010101
    P = []
    for w in range(8):
        v = bias4 + ((CR >> (32 * w)) & 0xffffffff)
        v += G[w] if w < 7 else (G[7] & 0xffffffff)
        v += hw[4] * ((c384 >> (32 * w)) & 0xffffffff)
        for k in range(4):
            v += D[4 * w + k] << (8 * k)
        assert 0 <= v < 1 << 58
        P.append(v)
    assert P[7] < 1 << 47
    R = sum(P[w] << (32 * w) for w in range(8))
    assert R % p == (S + CRorig) % p
    rtop = (P[7] + (P[6] >> 32)) >> 16
    assert rtop < 1 << 32
    q = (rtop * mu) >> 46
    assert q in (R // p, R // p - 1)
    pneg = (1 << 256) - p
    u = [q * ((pneg >> (32 * w)) & 0xffffffff) + P[w] for w in range(8)]
    assert all(x < 1 << 64 for x in u)
    rp = sum(u[w] << (32 * w) for w in range(8))
    r = rp - (q << 256)
    assert 0 <= r < 2 * p and (rp >> 256) - q in (0, 1)
    if r >= p:
        r -= p
    assert r == (S + CRorig) % p


def run_mm8(iterations, seed=2):
    rng = random.Random(seed)
    for p in PRIMES:
        for it in range(iterations):
            hi = 2 * 5800000
            cols = [rng.randrange(hi) if it % 3 else hi - 1 - rng.randrange(3) for _ in range(47)]
            if it % 5 == 0:
                cols = [rng.randrange(4) for _ in range(47)]
            mm8_model(cols, rng.randrange(p), p, rng)


if __name__ == "__main__":
    run_wide(3000)
    print("ok")
    run_mm8(2000)
    print("mm8 ok")
