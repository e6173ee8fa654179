"""The arithmetic of the matrix-core reductions with the high half of a sum folded on the matrix cores (k_mm8w, k_mm8), as a big-integer
model with every bound the kernels rely on asserted (tests/fold_model.py).  CPU only."""
import fold_model


def test_k_mm8w_reduction_model():
    fold_model.run_wide(400)


def test_k_mm8_epilogue_model():
    fold_model.run_mm8(300)


def test_every_t_b_has_32_balanced_digits():
    # t_b = 2^(256 + 8 b) mod p or that minus p: one of the two fits 32 balanced base-256 digits for every p < 2^256
    for p in fold_model.PRIMES + [(1 << 256) - 189, (1 << 254) + 1, (1 << 255) + 12345]:
        s, mu, c512, btot, tsum = fold_model.tables(p)
        assert len(s) == 32 and all(len(d) == 32 and all(-128 <= x <= 127 for x in d) for d in s)
        assert mu < 1 << 32
