// fp29.hpp -- GF(p) arithmetic for gfx950 in a carry-free radix-2^29 representation.
//
// Why radix 2^29 and not 32-bit limbs: measured on MI355X (scratch/ubench.hip,
// gpurun_out/ubench2.txt) v_mad_u64_u32 issues at the same half rate (~4.4 cycles per
// wave64 instruction per SIMD) as every carry-producing add (v_add_co_u32/v_addc_co_u32),
// and CDNA4 has no multiply-add with carry-IN.  A 32-bit-limb schoolbook product therefore
// costs 64 MADs + ~128 carry adds.  With 29-bit digits a 64-bit column accumulator absorbs
// 7 nine-digit products (9 * (2^29-1)^2 * 7 < 2^64) without any carry handling: the inner
// loop of every kernel is nothing but 81 v_mad_u64_u32 per 256x256-bit product, and carries
// are propagated once per <=7 products.  Montgomery's R is 2^(29*NL) (2^261 for NL=9), so for a
// 255-bit p a whole dot product of length d <= 32 needs ONE conditional subtraction after REDC.
//
// Replaces NTL ZZ_p add/sub/mul/inv/power as used by the reference's
// honeybadgermpc/ntl/rsdecode_impl.h (every function) -- restated, not translated.
//
// Forms: "packed"  = NW 32-bit words, canonical residue (what lives in HBM, little-endian,
//                    identical to the 4 x uint64 limbs of the C ABI)
//        "digits"  = NL 29-bit digits in u32 registers
//        "columns" = 2*NL 64-bit accumulators (value = sum col[k] * 2^(29k))
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HB_HD __host__ __device__ __forceinline__

namespace hb {

constexpr int LB = 29;
constexpr uint32_t DMASK = (1u << LB) - 1u;

template <int NL> struct FpParams {
    uint32_t p[NL];     // modulus, digits
    uint32_t r2[NL];    // R^2 mod p, digits (to_mont multiplier)
    uint32_t one[NL];   // R mod p, digits (Montgomery one)
    uint32_t n0;        // -p^{-1} mod 2^29
    uint32_t pad;
};

// how many products of two NL-digit numbers fit a 64-bit column before carries must move
template <int NL> struct Lazy { static constexpr int GROUP = (NL >= 9) ? 7 : ((NL >= 5) ? 12 : 21); };

// ---------------------------------------------------------------- pack / unpack
template <int NL, int NW> HB_HD void unpack(uint32_t (&d)[NL], const uint32_t (&w)[NW]) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = LB * i, j = bit >> 5, s = bit & 31;
        uint32_t lo = (j < NW) ? w[j] : 0u;
        uint32_t hi = (j + 1 < NW) ? w[j + 1] : 0u;
        uint32_t v = (s == 0) ? lo : (uint32_t)((((uint64_t)hi << 32) | lo) >> s);
        d[i] = v & DMASK;
    }
}
// digits must be fully normalised (< 2^29 each) and the value < 2^(32*NW)
template <int NL, int NW> HB_HD void pack(uint32_t (&w)[NW], const uint32_t (&d)[NL]) {
#pragma unroll
    for (int j = 0; j < NW; j++) {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int sh = LB * i - 32 * j;       // digit i sits at bit offset sh of word j
            if (sh > -LB && sh < 32) acc |= (sh >= 0) ? (d[i] << sh) : (d[i] >> (-sh));
        }
        w[j] = acc;
    }
}

// ---------------------------------------------------------------- column arithmetic
template <int NC> HB_HD void col_zero(uint64_t (&c)[NC]) {
#pragma unroll
    for (int k = 0; k < NC; k++) c[k] = 0;
}
// col += a * b   (81 v_mad_u64_u32 for NL = 9; a may be wave-uniform => SGPR operand)
template <int NL> HB_HD void mac(uint64_t (&c)[2 * NL], const uint32_t (&a)[NL], const uint32_t (&b)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; i++)
#pragma unroll
        for (int j = 0; j < NL; j++) c[i + j] += (uint64_t)a[i] * b[j];
}
// same, but only the first na digits of a are non-zero (na wave-uniform): used for
// matrices with small entries (Vandermonde rows at x = 1..n)
template <int NL> HB_HD void mac_short(uint64_t (&c)[2 * NL], const uint32_t (&a)[NL], int na, const uint32_t (&b)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
        if (i < na) {
#pragma unroll
            for (int j = 0; j < NL; j++) c[i + j] += (uint64_t)a[i] * b[j];
        }
    }
}
// propagate carries so that c[k] < 2^29 for k < NC-1 (the top column keeps the overflow)
template <int NC> HB_HD void carry(uint64_t (&c)[NC]) {
#pragma unroll
    for (int k = 0; k < NC - 1; k++) { c[k + 1] += c[k] >> LB; c[k] &= DMASK; }
}
// Montgomery REDC of carried columns: r = T / 2^(29*NL) mod p, r < p * (1 + T/(p*R)).
// Output digits normalised except that the top digit keeps any excess.
template <int NL> HB_HD void redc(uint32_t (&r)[NL], uint64_t (&c)[2 * NL], const FpParams<NL>& P) {
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint32_t m = ((uint32_t)c[i] * P.n0) & DMASK;
#pragma unroll
        for (int j = 0; j < NL; j++) c[i + j] += (uint64_t)m * P.p[j];
        c[i + 1] += c[i] >> LB;
    }
#pragma unroll
    for (int k = NL; k < 2 * NL - 1; k++) { c[k + 1] += c[k] >> LB; r[k - NL] = (uint32_t)c[k] & DMASK; }
    r[NL - 1] = (uint32_t)c[2 * NL - 1];
}
// r >= p ? r - p : r   (digits; top digit of r may exceed 29 bits)
template <int NL> HB_HD void cond_sub_p(uint32_t (&r)[NL], const FpParams<NL>& P) {
    uint32_t t[NL];
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int32_t v = (int32_t)r[i] - (int32_t)P.p[i] + borrow;
        if (i < NL - 1) { borrow = v >> 31; t[i] = (uint32_t)v & DMASK; }
        else { borrow = v >> 31; t[i] = (uint32_t)v; }
    }
    // top digit: r[NL-1] < 2^31 by construction, so the sign of v is the final borrow
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] = borrow ? r[i] : t[i];
}
template <int NL> HB_HD void finish(uint32_t (&r)[NL], uint64_t (&c)[2 * NL], const FpParams<NL>& P, int nsub) {
    carry(c);
    redc(r, c, P);
    for (int s = 0; s < nsub; s++) cond_sub_p(r, P);
}

// ---------------------------------------------------------------- field ops on digits (values < p)
// One product needs no carry pass before REDC: a column holds at most NL products of two 29-bit digits plus, during REDC, NL more
// and the carry of its neighbour -- 2 NL 2^58 + 2^35 < 2^63 -- and REDC propagates carries itself as it goes (the pass in
// finish() is for lazy dot products of several terms).  51 of ~300 instructions saved per multiplication.
template <int NL> HB_HD void mont_mul(uint32_t (&r)[NL], const uint32_t (&a)[NL], const uint32_t (&b)[NL], const FpParams<NL>& P) {
    uint64_t c[2 * NL];
    col_zero(c);
    mac<NL>(c, a, b);
    redc(r, c, P);
    cond_sub_p(r, P);
}
template <int NL> HB_HD void fp_add(uint32_t (&r)[NL], const uint32_t (&a)[NL], const uint32_t (&b)[NL], const FpParams<NL>& P) {
    uint32_t carry_ = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint32_t v = a[i] + b[i] + carry_;
        if (i < NL - 1) { carry_ = v >> LB; r[i] = v & DMASK; } else r[i] = v;
    }
    cond_sub_p(r, P);
}
template <int NL> HB_HD void fp_sub(uint32_t (&r)[NL], const uint32_t (&a)[NL], const uint32_t (&b)[NL], const FpParams<NL>& P) {
    uint32_t t[NL], u[NL];
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int32_t v = (int32_t)a[i] - (int32_t)b[i] + borrow;
        borrow = v >> 31;
        t[i] = (i < NL - 1) ? ((uint32_t)v & DMASK) : (uint32_t)v;
    }
    uint32_t carry_ = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        uint32_t v = (t[i] & ((i < NL - 1) ? DMASK : 0xffffffffu)) + P.p[i] + carry_;
        if (i < NL - 1) { carry_ = v >> LB; u[i] = v & DMASK; } else u[i] = v & DMASK;
    }
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] = borrow ? u[i] : t[i];
}
template <int NL> HB_HD void fp_neg(uint32_t (&r)[NL], const uint32_t (&a)[NL], const FpParams<NL>& P) {
    uint32_t z[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) z[i] = 0;
    fp_sub(r, z, a, P);
}
template <int NL> HB_HD bool fp_is_zero(const uint32_t (&a)[NL]) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) o |= a[i];
    return o == 0;
}
template <int NL> HB_HD bool fp_eq(const uint32_t (&a)[NL], const uint32_t (&b)[NL]) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) o |= a[i] ^ b[i];
    return o == 0;
}
template <int NL> HB_HD void fp_set(uint32_t (&r)[NL], const uint32_t (&a)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] = a[i];
}
template <int NL> HB_HD void to_mont(uint32_t (&r)[NL], const uint32_t (&a)[NL], const FpParams<NL>& P) { mont_mul(r, a, P.r2, P); }
template <int NL> HB_HD void from_mont(uint32_t (&r)[NL], const uint32_t (&a)[NL], const FpParams<NL>& P) {
    uint64_t c[2 * NL];
    col_zero(c);
#pragma unroll
    for (int i = 0; i < NL; i++) c[i] = a[i];
    redc(r, c, P);
    cond_sub_p(r, P);
}
// a^(p-2) (Montgomery in/out); exponent digits derived from P.p.  p prime, a != 0.
template <int NL> __host__ __device__ inline void fp_inv(uint32_t (&r)[NL], const uint32_t (&a)[NL], const FpParams<NL>& P) {
    uint32_t e[NL];
    // e = p - 2 (p odd and >= 3 so no borrow beyond digit 0 unless p[0] < 2, i.e. p[0] == 1)
    int32_t borrow = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        int32_t v = (int32_t)P.p[i] - (i == 0 ? 2 : 0) + borrow;
        borrow = v >> 31;
        e[i] = (uint32_t)v & DMASK;
    }
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
template <int NL> __host__ __device__ inline void fp_pow_u32(uint32_t (&r)[NL], const uint32_t (&a)[NL], uint32_t e, const FpParams<NL>& P) {
    uint32_t acc[NL], base[NL];
    fp_set(acc, P.one);
    fp_set(base, a);
    while (e) {
        if (e & 1u) mont_mul(acc, acc, base, P);
        e >>= 1;
        if (e) mont_mul(base, base, base, P);
    }
    fp_set(r, acc);
}

// ---------------------------------------------------------------- global memory element access
// elements are NW consecutive 32-bit words; NW = 8 -> two dwordx4 accesses per lane
template <int NW> HB_HD void load_words(uint32_t (&w)[NW], const uint32_t* __restrict__ p) {
    if constexpr (NW % 4 == 0) {
#pragma unroll
        for (int q = 0; q < NW / 4; q++) {
            uint4 v = reinterpret_cast<const uint4*>(p)[q];
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
    } else {
        uint2 v = *reinterpret_cast<const uint2*>(p);
        w[0] = v.x; w[1] = v.y;
    }
}
template <int NW> HB_HD void store_words(uint32_t* __restrict__ p, const uint32_t (&w)[NW]) {
    if constexpr (NW % 4 == 0) {
#pragma unroll
        for (int q = 0; q < NW / 4; q++) reinterpret_cast<uint4*>(p)[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
    } else {
        *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
    }
}
// streaming forms for data a launch touches once (the outputs of an encode, the columns a decode reads): the non-temporal hint keeps
// them from displacing what the next launch of the open reads (NW a multiple of 4)
typedef uint32_t hb_u4v __attribute__((ext_vector_type(4)));
template <int NW> __device__ __forceinline__ void load_words_nt(uint32_t (&w)[NW], const uint32_t* __restrict__ p) {
    static_assert(NW % 4 == 0, "whole dwordx4 accesses");
#pragma unroll
    for (int q = 0; q < NW / 4; q++) {
        const hb_u4v v = __builtin_nontemporal_load(reinterpret_cast<const hb_u4v*>(p) + q);
        w[4 * q] = v[0]; w[4 * q + 1] = v[1]; w[4 * q + 2] = v[2]; w[4 * q + 3] = v[3];
    }
}
template <int NW> __device__ __forceinline__ void store_words_nt(uint32_t* __restrict__ p, const uint32_t (&w)[NW]) {
    static_assert(NW % 4 == 0, "whole dwordx4 accesses");
#pragma unroll
    for (int q = 0; q < NW / 4; q++) __builtin_nontemporal_store(hb_u4v{w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]}, reinterpret_cast<hb_u4v*>(p) + q);
}
template <int NL, int NW> HB_HD void load_digits(uint32_t (&d)[NL], const uint32_t* __restrict__ p) {
    uint32_t w[NW];
    load_words<NW>(w, p);
    unpack<NL, NW>(d, w);
}
template <int NL, int NW> HB_HD void store_digits(uint32_t* __restrict__ p, const uint32_t (&d)[NL]) {
    uint32_t w[NW];
    pack<NL, NW>(w, d);
    store_words<NW>(p, w);
}

}  // namespace hb
