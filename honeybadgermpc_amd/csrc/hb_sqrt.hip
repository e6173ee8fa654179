// hb_sqrt.hip -- batched modular square roots (Tonelli-Shanks), one lane per element.
// Replaces sqrt_mod (reference: honeybadgermpc/ntl/hbmpc_ntl_helpers.pyx:441-444, NTL SqrRootMod).
// Which of the two roots is returned is not pinned by the reference (tests/test_ntl.py:331-341
// only checks r*r == a); this kernel returns the Tonelli-Shanks root for the smallest
// quadratic non-residue z >= 2.
#include "hb_common.hpp"

using namespace hb;

namespace {

struct SqrtExps {
    uint32_t half[9];    // (p - 1) / 2          as 29-bit digits
    uint32_t q[9];       // odd part of p - 1
    uint32_t q1h[9];     // (q + 1) / 2
    int s;               // p - 1 = q * 2^s
    int nd;              // digits in use
};

template <int NL> __device__ void pow_digits(uint32_t (&r)[NL], const uint32_t (&a)[NL], const uint32_t *e, const FpParams<NL> &P) {
    uint32_t acc[NL], base[NL];
    fp_set(acc, P.one);
    fp_set(base, a);
    for (int i = 0; i < NL; i++) {
        uint32_t ed = e[i];
        for (int b = 0; b < LB; b++) {
            if ((ed >> b) & 1u) mont_mul(acc, acc, base, P);
            mont_mul(base, base, base, P);
        }
    }
    fp_set(r, acc);
}

// c = z^q for the smallest non-residue z (one thread)
template <int NL, int NW>
__global__ void k_sqrt_setup(const FpParams<NL> P, const SqrtExps E, uint32_t *__restrict__ cz) {
    uint32_t minus1[NL];
    fp_neg(minus1, P.one, P);
    for (uint32_t z = 2;; z++) {
        uint32_t zd[NL], zm[NL], l[NL];
#pragma unroll
        for (int q = 0; q < NL; q++) zd[q] = 0;
        zd[0] = z;       // z < 2^29 and z < p are guaranteed for the tiny z that succeed (p >= 3)
        to_mont(zm, zd, P);
        if (fp_is_zero(zm)) continue;
        pow_digits<NL>(l, zm, E.half, P);
        if (fp_eq(l, minus1)) {
            uint32_t c[NL];
            pow_digits<NL>(c, zm, E.q, P);
#pragma unroll
            for (int q = 0; q < NL; q++) cz[q] = c[q];
            return;
        }
    }
}

template <int NL, int NW>
__global__ void __launch_bounds__(64) k_sqrt(const FpParams<NL> P, const SqrtExps E, const uint32_t *__restrict__ cz,
                                             const uint32_t *__restrict__ a, int64_t C, uint32_t *__restrict__ out, uint8_t *__restrict__ ok) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C) return;
    uint32_t ad[NL], am[NL];
    load_digits<NL, NW>(ad, a + i * NW);
    to_mont(am, ad, P);
    uint32_t res[NL];
    bool good = true;
    if (fp_is_zero(am)) {
#pragma unroll
        for (int q = 0; q < NL; q++) res[q] = 0;
    } else {
        uint32_t leg[NL];
        pow_digits<NL>(leg, am, E.half, P);
        if (!fp_eq(leg, P.one)) {
            good = false;
#pragma unroll
            for (int q = 0; q < NL; q++) res[q] = 0;
        } else {
            uint32_t c[NL], r[NL], t[NL];
#pragma unroll
            for (int q = 0; q < NL; q++) c[q] = cz[q];
            pow_digits<NL>(r, am, E.q1h, P);
            pow_digits<NL>(t, am, E.q, P);
            int m = E.s;
            while (!fp_eq(t, P.one)) {
                int k = 0;
                uint32_t tt[NL];
                fp_set(tt, t);
                while (!fp_eq(tt, P.one)) { mont_mul(tt, tt, tt, P); k++; }
                uint32_t b[NL];
                fp_set(b, c);
                for (int j = 0; j < m - k - 1; j++) mont_mul(b, b, b, P);
                mont_mul(r, r, b, P);
                mont_mul(c, b, b, P);
                mont_mul(t, t, c, P);
                m = k;
            }
            from_mont(res, r, P);
        }
    }
    store_digits<NL, NW>(out + i * NW, res);
    ok[i] = good ? 1 : 0;
}

// host: exponent digits from the modulus limbs
void make_exps(SqrtExps &E, const uint64_t *p_limbs, int n_limbs, int nl) {
    unsigned __int128 carry;
    uint64_t pm1[4] = {0, 0, 0, 0};
    for (int i = 0; i < n_limbs; i++) pm1[i] = p_limbs[i];
    pm1[0] -= 1;                                     // p odd => no borrow
    auto shr1 = [](uint64_t (&v)[4]) { for (int i = 0; i < 4; i++) v[i] = (v[i] >> 1) | (i < 3 ? v[i + 1] << 63 : 0); };
    auto digits = [&](uint32_t (&d)[9], const uint64_t (&v)[4]) {
        for (int i = 0; i < 9; i++) {
            int bit = 29 * i, j = bit >> 6, s = bit & 63;
            uint64_t lo = j < 4 ? v[j] >> s : 0;
            if (s > 35 && j + 1 < 4) lo |= v[j + 1] << (64 - s);
            d[i] = (uint32_t)lo & DMASK;
        }
    };
    uint64_t half[4], q[4], q1h[4];
    memcpy(half, pm1, 32); shr1(half);
    memcpy(q, pm1, 32);
    int s = 0;
    while ((q[0] & 1) == 0) { shr1(q); s++; }
    memcpy(q1h, q, 32);
    carry = 1;
    for (int i = 0; i < 4; i++) { carry += q1h[i]; q1h[i] = (uint64_t)carry; carry >>= 64; }
    shr1(q1h);
    digits(E.half, half); digits(E.q, q); digits(E.q1h, q1h);
    E.s = s; E.nd = nl;
}

}  // namespace

extern "C" int hb_sqrt_mod(hb_ctx *ctx, const uint64_t *a_dev, int64_t C, uint64_t *out_dev, uint8_t *ok_dev, void *stream) { HB_API_GUARD(ctx);
    if (!ctx || C < 0) return HB_ERR_BAD_ARG;
    if (C == 0) return HB_OK;
    if (!a_dev || !out_dev || !ok_dev) return HB_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    SqrtExps E;
    make_exps(E, ctx->p_limbs, ctx->n_limbs, ctx->nl());
    uint32_t *cz = nullptr;
    auto it = ctx->dcache.find("sqrt_cz");
    if (it != ctx->dcache.end()) cz = (uint32_t *)it->second;
    else {
        HB_HIP(ctx, hipMalloc(&cz, 9 * 4));
        HB_DISPATCH(ctx, (k_sqrt_setup<9, 8><<<1, 1, 0, s>>>(ctx->pw, E, cz)), (k_sqrt_setup<3, 2><<<1, 1, 0, s>>>(ctx->pn, E, cz)));
        HB_LAUNCH_CHECK(ctx);
        ctx->dcache["sqrt_cz"] = cz;
    }
    const int64_t blocks = (C + 63) / 64;
    if (blocks > 0x7fffffffLL) return fail(ctx, HB_ERR_UNSUPPORTED, "sqrt: batch too large");
    HB_DISPATCH(ctx,
        (k_sqrt<9, 8><<<(unsigned)blocks, 64, 0, s>>>(ctx->pw, E, cz, (const uint32_t *)a_dev, C, (uint32_t *)out_dev, ok_dev)),
        (k_sqrt<3, 2><<<(unsigned)blocks, 64, 0, s>>>(ctx->pn, E, cz, (const uint32_t *)a_dev, C, (uint32_t *)out_dev, ok_dev)));
    HB_LAUNCH_CHECK(ctx);
    return HB_OK;
}
