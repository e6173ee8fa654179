// hb_ntt.hip -- radix-2 NTT evaluation and omega-point interpolation for gfx950.
//
// Reference functions replaced (paths under /root/reference):
//   fft / _fft (recursive radix-2 DIT with a cached 16-point Vandermonde base case)
//                                   honeybadgermpc/ntl/rsdecode_impl.h:125-192
//   fft, partial_fft, fft_batch_evaluate          hbmpc_ntl_helpers.pyx:246-316
//   fnt_decode_step1/2, fft_interpolate, fft_batch_interpolate
//                                   rsdecode_impl.h:194-265, pyx:318-381
//
// Evaluation: out[c][i] = sum_{j < min(d, n)} coeffs[c][j] * omega^(i j), i < k.
//   * k_ntt_lds: transforms of order n <= 4096 live entirely in LDS (n x 9 digits = 36 n bytes),
//     PB polynomials per workgroup, one butterfly per thread per stage, twiddles (Montgomery
//     form) from a device table, data canonical: t = REDC(w_mont * a) needs no conversions.
//     The 9-word element stride is odd, so strided butterfly accesses spread over LDS banks.
//   * larger orders (single big polynomials: the reference's benchmark sizes up to 2^20) run
//     the same butterflies stage by stage over a digit buffer in HBM.
//   * small transforms with few coefficients are cheaper as a lazily-reduced mat-vec with the
//     Vandermonde matrix at the omega powers (the reference itself bottoms out in a 16-point
//     Vandermonde product, rsdecode_impl.h:16,133-136): the launcher picks by MAD count.
// Interpolation at the points omega^zs is the unique polynomial through them, so it is
// computed with the inverse Vandermonde matrix at those points (k x k mat-vec, lazily
// reduced): for the party counts of this path (k <= a few hundred) that is fewer MADs than
// scale + n-point NTT + MulTrunc(Q, A, k) (see DESIGN.md cost table); results are
// identical because every output is a canonical residue.
#include <algorithm>
#include <type_traits>
#include <vector>

#include "hb_common.hpp"

using namespace hb;

namespace {

__host__ __device__ inline uint32_t bitrev(uint32_t v, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
}

// tw[j] = omega^j (Montgomery digits), j < n/2
template <int NL, int NW>
__global__ void k_twiddles(const FpParams<NL> P, const uint32_t *__restrict__ omega, int half, uint32_t *__restrict__ tw) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= half) return;
    uint32_t od[NL], om[NL], r[NL];
    load_digits<NL, NW>(od, omega);
    to_mont(om, od, P);
    fp_pow_u32(r, om, (uint32_t)j, P);
#pragma unroll
    for (int q = 0; q < NL; q++) tw[(size_t)j * NL + q] = r[q];
}

template <int NL>
__device__ __forceinline__ void butterfly(uint32_t *a0, uint32_t *a1, const uint32_t *w, bool trivial, const FpParams<NL> &P) {
    uint32_t u[NL], v[NL], t[NL], s0[NL], s1[NL];
#pragma unroll
    for (int q = 0; q < NL; q++) { u[q] = a0[q]; v[q] = a1[q]; }
    if (trivial) {
        fp_set(t, v);
    } else {
        uint32_t wd[NL];
#pragma unroll
        for (int q = 0; q < NL; q++) wd[q] = w[q];
        mont_mul(t, wd, v, P);
    }
    fp_add(s0, u, t, P);
    fp_sub(s1, u, t, P);
#pragma unroll
    for (int q = 0; q < NL; q++) { a0[q] = s0[q]; a1[q] = s1[q]; }
}

// ---- value-lazy butterflies -------------------------------------------------------------------------------------------
// Between the input and the output of a transform nothing has to be canonical.  Elements are kept as normalised digits of a
// value below 2^(29 NL) (the top digit takes the excess):
//     t  = REDC(w v)          < 2p for ANY v < R = 2^(29 NL)  (w < p):  no conditional subtraction
//     u' = u + t,  v' = u - t + 2p                                      one carry pass each, no comparison with p
// so a stage adds at most 2p to the bound: inputs below 2^(32 NW) and log2(n) <= 12 stages stay below 2^(32 NW) + 24 p < R.
// One canonicalisation per OUTPUT replaces three conditional subtractions per butterfly (a quarter of its instructions).
// multiples of p for the trivial butterflies of the first pass: m[u] >= bound of an element after u stages, B_0 = 2^(32 NW),
// B_{u+1} = B_u + m[u]
template <int NL> struct LazyConsts { uint32_t m[3][NL]; uint32_t p2[NL]; /* 2p */ };

template <int NL> __device__ __forceinline__ void mont_mul_lazy(uint32_t (&r)[NL], const uint32_t (&a)[NL], const uint32_t (&b)[NL], const FpParams<NL> &P) {
    uint64_t c[2 * NL];
    col_zero(c);
    mac<NL>(c, a, b);
    redc(r, c, P);                                        // no carry pass needed for a single product (fp29.hpp: mont_mul)
}
template <int NL> __device__ __forceinline__ void add_lazy(uint32_t (&r)[NL], const uint32_t (&a)[NL], const uint32_t (&b)[NL]) {
    uint32_t cy = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const uint32_t v = a[i] + b[i] + cy;
        if (i < NL - 1) { cy = v >> LB; r[i] = v & DMASK; } else r[i] = v;
    }
}
// r = a - b + p2 with p2 = 2p >= b: non-negative, one signed carry pass
template <int NL> __device__ __forceinline__ void sub_lazy(uint32_t (&r)[NL], const uint32_t (&a)[NL], const uint32_t (&b)[NL], const uint32_t (&p2)[NL]) {
    int32_t cy = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int32_t v = (int32_t)a[i] - (int32_t)b[i] + (int32_t)p2[i] + cy;      // |v| < 2^31: digits < 2^29 (top: < 2^30 by the bound above)
        if (i < NL - 1) { cy = v >> LB; r[i] = (uint32_t)v & DMASK; } else r[i] = (uint32_t)v;
    }
}
// x < R  ->  the same residue below 2p.  Wide contexts with 2^254 <= p < 2^256 (psc != nullptr): a one-digit Barrett quotient from
// the top digit, qhat = floor(x[8] mu / 2^58) with mu = floor(2^290 / p), which is floor(x / p) or one less (the digits below
// the top one and the truncation of mu cost less than 1 + 2^-21), then x - qhat p as x + qhat (2^261 - p) mod 2^261: ~30
// instructions.  Any other modulus: x = REDC(x (R mod p)).
template <int NL, bool PSC> __device__ __forceinline__ void reduce2p(uint32_t (&r)[NL], const uint32_t (&x)[NL], const FpParams<NL> &P, const PrescaleParams &psc) {
    if constexpr (NL == 9 && PSC) {
        {
            const uint64_t mid = (uint64_t)x[8] * psc.m1 + (((uint64_t)x[8] * psc.m0) >> LB);
            const uint32_t q0 = (uint32_t)(mid >> LB);          // < 2^7
            uint64_t dc[9];
#pragma unroll
            for (int k = 0; k < 9; k++) dc[k] = x[k] + (uint64_t)q0 * psc.pbar[k];
#pragma unroll
            for (int k = 0; k < 9; k++) { r[k] = (uint32_t)dc[k] & DMASK; if (k < 8) dc[k + 1] += dc[k] >> LB; }
            return;
        }
    }
    mont_mul_lazy(r, x, P.one, P);
}
// canonical residue of a lazy value x < R
template <int NL, bool PSC> __device__ __forceinline__ void canon_lazy(uint32_t (&r)[NL], const uint32_t (&x)[NL], const FpParams<NL> &P, const PrescaleParams &psc) {
    reduce2p<NL, PSC>(r, x, P, psc);
    cond_sub_p(r, P);
}

// One radix-2^R pass (R stages s .. s+R-1 of the decimation-in-time transform) of one unit: the 2^R elements at
// positions hi 2^(s+R) + q 2^s + lo, q < 2^R, are read once, transformed in registers and written back.
// Stage s+u pairs the elements whose q differ in bit u; the twiddle of the pair is omega^(j n / 2^(s+u+1)) with
// j = (q mod 2^u) 2^s + lo, read from the LDS table.
// Input pruning: after S stages the block at positions [m 2^S, (m+1) 2^S) holds the sub-transform of the coefficients
// j = r (mod n / 2^S), r = bitrev(m); with dd < n coefficients present it is identically zero when r >= dd.  A butterfly
// whose lower operand lies in such a block is a copy (u, u); the upper operand's class is smaller, so it is the only case.
template <int NL, int R, bool FIRST, bool PSC>
__device__ __forceinline__ void ntt_unit(uint32_t *__restrict__ base /* polynomial in LDS */, const uint32_t *__restrict__ twl, int n, int logn,
                                         int s, int hi, int lo, int dd, const FpParams<NL> &P,
                                         const PrescaleParams &psc, const LazyConsts<NL> &lc) {
    constexpr int M = 1 << R;
    uint32_t e[M][NL];
    const int p0 = (hi << (s + R)) + lo;
#pragma unroll
    for (int q = 0; q < M; q++) {
        const int pos = p0 + (q << s);
        // the first pass reads positions that the loader never wrote (classes >= dd) as zero instead of clearing LDS
        const bool present = !FIRST || (int)(__brev((uint32_t)pos) >> (32 - logn)) < dd;
#pragma unroll
        for (int w = 0; w < NL; w++) e[q][w] = present ? base[(size_t)pos * NL + w] : 0u;
    }
#pragma unroll
    for (int u = 0; u < R; u++) {
        const int S = s + u;                             // global stage
        const int lb = logn - S;                         // bits of a block index at this stage
#pragma unroll
        for (int q = 0; q < M; q++) {
            if (q & (1 << u)) continue;
            const int qb = q | (1 << u);
            const int posb = p0 + (qb << s);
            const int rb = lb > 0 ? (int)(__brev((uint32_t)(posb >> S)) >> (32 - lb)) : 0;
            if (rb >= dd) {                              // lower operand identically zero: (u, u)
#pragma unroll
                for (int w = 0; w < NL; w++) e[qb][w] = e[q][w];
                continue;
            }
            const int j = ((q & ((1 << u) - 1)) << s) + lo;
            uint32_t t[NL];
            // Trivial twiddles (j = 0): no product.  In the first pass (s = 0, lo = 0) j is a compile-time function of (q, u) --
            // 7/8 of all trivial butterflies live there -- and the subtraction adds a precomputed multiple of p that covers
            // the bound of its operand (values may double per stage there: 8 x 2^(32 NW) after three stages, still below R / 4);
            // later passes multiply by tw[0] = R mod p like any other twiddle (a lane-dependent branch here sends the element
            // file to scratch).
            if (FIRST && (q & ((1 << u) - 1)) == 0) {
                // (u, v) -> (u + v, u - v + M_u), M_u a multiple of p above the bound of v at stage u of the first pass
                uint32_t s0[NL], s1[NL], mu_[NL];
#pragma unroll
                for (int w = 0; w < NL; w++) mu_[w] = lc.m[u][w];
                add_lazy(s0, e[q], e[qb]);
                sub_lazy(s1, e[q], e[qb], mu_);
                fp_set(e[q], s0);
                fp_set(e[qb], s1);
                continue;
            } else {
                const uint32_t *wp = twl + (size_t)(j * (n >> (S + 1))) * NL;
                uint32_t wd[NL];
#pragma unroll
                for (int w = 0; w < NL; w++) wd[w] = wp[w];
                mont_mul_lazy(t, wd, e[qb], P);
            }
            uint32_t s0[NL], s1[NL], p2l[NL];
#pragma unroll
            for (int w = 0; w < NL; w++) p2l[w] = lc.p2[w];
            add_lazy(s0, e[q], t);
            sub_lazy(s1, e[q], t, p2l);
            fp_set(e[q], s0);
            fp_set(e[qb], s1);
        }
    }
#pragma unroll
    for (int q = 0; q < M; q++) {
        const int pos = p0 + (q << s);
#pragma unroll
        for (int w = 0; w < NL; w++) base[(size_t)pos * NL + w] = e[q][w];
    }
}

// whole transform in LDS; PB polynomials per block.  Strided views on both sides:
//   in(c, j)  at in  + (c*in_sc  + j*in_sl ) elements, zero beyond in_count (chunk_data padding)
//   out(c, i) at out + (c*out_sc + i*out_sl) elements, i < k, skipped beyond out_count
// `*_poly_fast` picks the thread->element map so that consecutive lanes touch consecutive
// addresses for coefficient-major / party-major buffers (stride_c == 1).
// CHECK: instead of storing, compare out(c, i) with the buffer for rows i in check_mask
// (the validating re-encode of IncrementalDecoder, reed_solomon.py:313-326).
// The log2(n) stages run as radix-8 passes in registers (a remainder of one or two stages first): 3 LDS round trips and
// barriers for n = 256 instead of 8, twiddles (Montgomery form) staged in LDS once per block.
template <int NL, int NW, bool CHECK, bool PSC>
__global__ void __launch_bounds__(256) k_ntt_lds(const FpParams<NL> P, const uint32_t *__restrict__ tw,
                                                 const uint32_t *__restrict__ in, int64_t in_sc, int64_t in_sl, int64_t in_count, int d,
                                                 int n, int logn, int k,
                                                 uint32_t *__restrict__ out, int64_t out_sc, int64_t out_sl, int64_t out_count,
                                                 const int32_t *__restrict__ check_mask, int32_t *__restrict__ mismatch,
                                                 int64_t C, int PB, int in_poly_fast, int out_poly_fast, PrescaleParams psc, LazyConsts<NL> lc,
                                                 uint32_t *__restrict__ copy_dst, int64_t copy_sc, int64_t copy_sl, int64_t copy_count, int copy_rows) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int64_t c0 = (int64_t)blockIdx.x * PB;
    const int npoly = (int)min((int64_t)PB, C - c0);
    const int dd = min(d, n);
    const int half = n >> 1;
    uint32_t *twl = lds;                                  // [max(n/2, 1)][NL]
    uint32_t *data = lds + (size_t)(half > 0 ? half : 1) * NL;   // [PB][n][NL], polynomials one dword further apart than n NL:
    // with party-major / coefficient-major buffers consecutive lanes hold consecutive polynomials, and n NL = 576 or 2304
    // dwords would put them all on one LDS bank (measured: order 64, party-major output 138 us against 118 chunk-major)
    const int pstride = n * NL + 1;
    for (int idx = threadIdx.x; idx < half * NL; idx += blockDim.x) twl[idx] = tw[idx];
    // load the dd coefficients of every polynomial into bit-reversed positions
    for (int idx = threadIdx.x; idx < npoly * dd; idx += blockDim.x) {
        const int pl = in_poly_fast ? idx % npoly : idx / dd;
        const int j = in_poly_fast ? idx / npoly : idx % dd;
        const int64_t e = (c0 + pl) * in_sc + (int64_t)j * in_sl;
        uint32_t dg[NL], wd_[NW];
#pragma unroll
        for (int q = 0; q < NW; q++) wd_[q] = 0;
        if (e < in_count) load_words<NW>(wd_, in + e * NW);
        unpack<NL, NW>(dg, wd_);
        if constexpr (CHECK) {
            // hand the caller its rows of the input (the decoded coefficients) in its own layout while they pass through
            // (row 0 -> the R2 message, all rows chunk-major -> the result): replaces a copy kernel
            if (copy_dst && j < copy_rows) {
                const int64_t ci = (c0 + pl) * copy_sc + (int64_t)j * copy_sl;
                if (ci < copy_count) store_words<NW>(copy_dst + ci * NW, wd_);
            }
        }
        uint32_t *dst = data + (size_t)pl * pstride + (size_t)bitrev((uint32_t)j, logn) * NL;
#pragma unroll
        for (int q = 0; q < NL; q++) dst[q] = dg[q];
    }
    __syncthreads();
    // passes: the remainder (1 or 2 stages) first, then radix-8
    int s = 0;
    bool first = true;
    const int rem = logn % 3;
    auto run_pass = [&](auto rtag) {
        constexpr int R = decltype(rtag)::value;
        const int units = n >> R;                         // per polynomial
        const int lo_bits = s, hi_bits = logn - s - R;
        for (int uidx = threadIdx.x; uidx < npoly * units; uidx += blockDim.x) {
            // first pass (units = 8 contiguous elements = 72 dwords apart: 8 banks): consecutive lanes take consecutive
            // POLYNOMIALS (odd stride: all banks); later passes: consecutive units of one polynomial (consecutive elements)
            const int pl = first ? uidx % npoly : uidx / units, u = first ? uidx / npoly : uidx - pl * units;
            // hi in bit-reversed order: units whose lower operands are pruned end up next to each other
            const int lo = u & ((1 << lo_bits) - 1);
            const int hr = u >> lo_bits;
            const int hi = hi_bits > 0 ? (int)(__brev((uint32_t)hr) >> (32 - hi_bits)) : 0;
            if (first) ntt_unit<NL, R, true, PSC>(data + (size_t)pl * pstride, twl, n, logn, 0, hi, 0, dd, P, psc, lc);
            else ntt_unit<NL, R, false, PSC>(data + (size_t)pl * pstride, twl, n, logn, s, hi, lo, dd, P, psc, lc);
        }
        __syncthreads();
        s += R;
        first = false;
    };
    if (logn == 0) { /* order 1: the coefficient itself */ }
    if (rem == 1) run_pass(std::integral_constant<int, 1>{});
    if (rem == 2) run_pass(std::integral_constant<int, 2>{});
    while (s < logn) run_pass(std::integral_constant<int, 3>{});
    for (int idx = threadIdx.x; idx < npoly * k; idx += blockDim.x) {
        const int pl = out_poly_fast ? idx % npoly : idx / k;
        const int i = out_poly_fast ? idx / npoly : idx % k;
        const int64_t e = (c0 + pl) * out_sc + (int64_t)i * out_sl;
        uint32_t dg[NL], lz[NL];
        const bool present = logn > 0 || dd > 0;         // order 1 with no coefficient: zero
#pragma unroll
        for (int q = 0; q < NL; q++) lz[q] = present ? data[(size_t)pl * pstride + (size_t)i * NL + q] : 0u;
        canon_lazy<NL, PSC>(dg, lz, P, psc);
        if constexpr (CHECK) {
            if (check_mask[i]) {
                uint32_t w[NW], ex[NW];
                pack<NL, NW>(w, dg);
                load_words<NW>(ex, out + e * NW);
                uint32_t diff = 0;
#pragma unroll
                for (int q = 0; q < NW; q++) diff |= ex[q] ^ w[q];
                if (diff) atomicOr(mismatch, 1);
            }
        } else {
            // party-major outputs (consecutive lanes = consecutive polynomials: an encode's rows) leave with the streaming hint, as k_mm8's do
            // (hb_mfma.hip): config 3 at omega points 230 -> 222 us an open, config 2 86.3 -> 84.0 (round 6)
            if (e < out_count) {
                uint32_t w_[NW];
                pack<NL, NW>(w_, dg);
                if constexpr (NW % 4 == 0) { if (out_poly_fast) store_words_nt<NW>(out + e * NW, w_); else store_words<NW>(out + e * NW, w_); }
                else store_words<NW>(out + e * NW, w_);
            }
        }
    }
}

// --- large orders: digit buffer in HBM, one launch per stage ------------------------------
template <int NL, int NW>
__global__ void k_ntt_g_load(const uint32_t *__restrict__ in, int d, int n, int logn, uint32_t *__restrict__ buf, int64_t C) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * n) return;
    const int64_t c = idx / n; const int pos = (int)(idx % n);
    const int j = (int)bitrev((uint32_t)pos, logn);
    uint32_t dg[NL];
    if (j < min(d, n)) load_digits<NL, NW>(dg, in + (c * (int64_t)d + j) * NW);
    else {
#pragma unroll
        for (int q = 0; q < NL; q++) dg[q] = 0;
    }
#pragma unroll
    for (int q = 0; q < NL; q++) buf[(size_t)idx * NL + q] = dg[q];
}
template <int NL>
__global__ void k_ntt_g_stage(const FpParams<NL> P, const uint32_t *__restrict__ tw, uint32_t *__restrict__ buf, int n, int s, int64_t C) {
    const int half = n >> 1;
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= C * half) return;
    const int64_t c = b / half; const int bb = (int)(b % half);
    const int h = 1 << s, j = bb & (h - 1);
    const int i0 = ((bb >> s) << (s + 1)) + j;
    uint32_t *base = buf + (size_t)c * n * NL;
    butterfly<NL>(base + (size_t)i0 * NL, base + (size_t)(i0 + h) * NL, tw + (size_t)j * (half >> s) * NL, j == 0, P);
}
template <int NL, int NW>
__global__ void k_ntt_g_store(const uint32_t *__restrict__ buf, int n, int k, uint32_t *__restrict__ out, int64_t C) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * k) return;
    const int64_t c = idx / k; const int i = (int)(idx % k);
    uint32_t dg[NL];
#pragma unroll
    for (int q = 0; q < NL; q++) dg[q] = buf[((size_t)c * n + i) * NL + q];
    store_digits<NL, NW>(out + (size_t)idx * NW, dg);
}

}  // namespace

namespace hb {

int get_twiddles(hb_ctx *ctx, const uint64_t *omega_host, int n, uint32_t **tw, hipStream_t s) {
    std::string key = "tw:" + std::to_string(n) + ":";
    key.append(reinterpret_cast<const char *>(omega_host), (size_t)ctx->n_limbs * 8);
    auto it = ctx->dcache.find(key);
    if (it != ctx->dcache.end()) { *tw = (uint32_t *)it->second; return HB_OK; }
    const int half = n > 1 ? n / 2 : 1;
    uint32_t *od = nullptr, *t = nullptr;
    int rc = upload_elems(ctx, omega_host, 1, &od, s); if (rc) return rc;
    HB_HIP(ctx, hipMalloc(&t, (size_t)half * ctx->nl() * 4));
    HB_DISPATCH(ctx,
        (k_twiddles<9, 8><<<(half + 63) / 64, 64, 0, s>>>(ctx->pw, od, half, t)),
        (k_twiddles<3, 2><<<(half + 63) / 64, 64, 0, s>>>(ctx->pn, od, half, t)));
    HB_LAUNCH_CHECK(ctx);
    HB_HIP(ctx, hipStreamSynchronize(s));
    HB_HIP(ctx, hipFree(od));
    ctx->dcache[key] = t;
    *tw = t;
    return HB_OK;
}

// m[u] = p * ceil(B_u / p), B_0 = 2^(32 NW), B_{u+1} = B_u + m[u]  (radix-2^29 digits; plain long arithmetic, once per launch)
template <int NL> static LazyConsts<NL> lazy_consts(hb_ctx *ctx) {
    constexpr int W = 12;                                   // 32-bit words: values stay below 2^261
    typedef std::vector<uint32_t> Big;
    Big p(W, 0);
    for (int i = 0; i < ctx->n_limbs; i++) { p[2 * i] = (uint32_t)ctx->p_limbs[i]; p[2 * i + 1] = (uint32_t)(ctx->p_limbs[i] >> 32); }
    auto ge = [](const Big &a, const Big &b) { for (int i = W - 1; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i]; return true; };
    auto sub = [](Big &a, const Big &b) { int64_t br = 0; for (int i = 0; i < W; i++) { int64_t t = (int64_t)a[i] - b[i] + br; a[i] = (uint32_t)t; br = t >> 32; } };
    auto add = [](Big &a, const Big &b) { uint64_t cy = 0; for (int i = 0; i < W; i++) { uint64_t t = (uint64_t)a[i] + b[i] + cy; a[i] = (uint32_t)t; cy = t >> 32; } };
    auto mod = [&](const Big &x) {                          // x mod p, binary long division
        Big r(W, 0);
        for (int bit = W * 32 - 1; bit >= 0; bit--) {
            for (int i = W - 1; i > 0; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 31);
            r[0] = (r[0] << 1) | ((x[bit >> 5] >> (bit & 31)) & 1u);
            if (ge(r, p)) sub(r, p);
        }
        return r;
    };
    LazyConsts<NL> lc;
    Big b(W, 0);
    b[(ctx->n_limbs * 64) >> 5] = 1u;                       // B_0 = 2^(64 limbs)
    for (int u = 0; u < 3; u++) {
        Big m(b), r = mod(b);
        bool zero = true; for (int i = 0; i < W; i++) if (r[i]) zero = false;
        if (!zero) { sub(m, r); add(m, p); }                // round up to a multiple of p
        for (int k = 0; k < NL; k++) {
            const int bit = 29 * k, j = bit >> 5, sft = bit & 31;
            const uint64_t v = m[j] | ((uint64_t)(j + 1 < W ? m[j + 1] : 0) << 32);
            lc.m[u][k] = (uint32_t)(v >> sft) & (k < NL - 1 ? DMASK : 0xffffffffu);
        }
        add(b, m);
    }
    {
        Big two(p);
        add(two, p);
        for (int k = 0; k < NL; k++) {
            const int bit = 29 * k, j = bit >> 5, sft = bit & 31;
            const uint64_t v = two[j] | ((uint64_t)(j + 1 < W ? two[j + 1] : 0) << 32);
            lc.p2[k] = (uint32_t)(v >> sft) & (k < NL - 1 ? DMASK : 0xffffffffu);
        }
    }
    return lc;
}

// LDS NTT launcher with views; returns HB_ERR_UNSUPPORTED when the order does not fit LDS
int launch_ntt_lds(hb_ctx *ctx, const uint32_t *tw, int n, const uint32_t *in, hb_view iv, int64_t in_count, int d, int k,
                   uint32_t *out, hb_view ov, int64_t out_count, const int32_t *check_mask_dev, int32_t *mismatch_dev,
                   int64_t C, hipStream_t s, uint32_t *copy_dst, hb_view cpv, int64_t copy_count, int copy_rows) {
    if (C <= 0 || k <= 0) return HB_OK;
    int logn = 0; while ((1 << logn) < n) logn++;
    const size_t elem_lds = (size_t)ctx->nl() * 4;
    // radix-8 units: n / 8 per polynomial and pass; 2048 / n polynomials give the 256 threads one unit each (72 KB of data:
    // two workgroups per CU)
    int PB = n <= 2048 ? 2048 / n : 1; if (PB > 256) PB = 256;
    if ((int64_t)PB > C) PB = (int)C;
    const size_t lds = ((size_t)PB * n + (size_t)(n > 1 ? n / 2 : 1)) * elem_lds + (size_t)PB * 4;
    if (lds > 160 * 1024) return fail(ctx, HB_ERR_UNSUPPORTED, "ntt: order does not fit LDS");
    const int64_t blocks = (C + PB - 1) / PB;
    if (blocks > 0x7fffffffLL) return fail(ctx, HB_ERR_UNSUPPORTED, "fft: batch too large");
    const int ipf = iv.stride_c == 1 ? 1 : 0, opf = ov.stride_c == 1 ? 1 : 0;
    const bool check = check_mask_dev != nullptr;
    const int psc_ok = (ctx->n_limbs == 4 && prescale_params(ctx)) ? 1 : 0;      // 2^254 <= p < 2^256: the cheap lazy -> canonical step
#define HB_NTT(NL_, NW_, CHK_, PP_)                                                                                                       \
    do { if (psc_ok) HB_NTT_(NL_, NW_, CHK_, true, PP_); else HB_NTT_(NL_, NW_, CHK_, false, PP_); } while (0)
#define HB_NTT_(NL_, NW_, CHK_, PSC_, PP_)                                                                                                \
    do {                                                                                                                                  \
        HB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_lds<NL_, NW_, CHK_, PSC_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024))); \
        k_ntt_lds<NL_, NW_, CHK_, PSC_><<<(unsigned)blocks, 256, lds, s>>>(PP_, tw, in, iv.stride_c, iv.stride_l, in_count, d, n, logn, k, out, ov.stride_c, ov.stride_l, out_count, check_mask_dev, mismatch_dev, C, PB, ipf, opf, ctx->psc, lazy_consts<NL_>(ctx), copy_dst, cpv.stride_c, cpv.stride_l, copy_count, copy_rows); \
    } while (0)
    if (ctx->n_limbs == 4) { if (check) HB_NTT(9, 8, true, ctx->pw); else HB_NTT(9, 8, false, ctx->pw); }
    else { if (check) HB_NTT(3, 2, true, ctx->pn); else HB_NTT(3, 2, false, ctx->pn); }
#undef HB_NTT
#undef HB_NTT_
    HB_LAUNCH_CHECK(ctx);
    return HB_OK;
}

}  // namespace hb

namespace {

// ---- four-step transform for orders beyond the LDS kernel (reference benchmark/test_benchmark_polynomial.py:22-48
// runs fft up to n = 2^20 through rsdecode_impl.h:125-192) ---------------------------------------------------------------------------
// n = n1 n2, j = j1 n2 + j2, i = i1 + i2 n1:   omega^(i j) = (omega^n2)^(i1 j1) * omega^(i1 j2) * (omega^n1)^(i2 j2)
//   1. for every j2: the n1-point transform over j1 (root omega^n2)            -> A[i1][j2]        k_ntt_lds, strided views
//   2. A[i1][j2] *= omega^(i1 j2)                                               k_ntt_twist (this kernel)
//   3. for every i1: the n2-point transform over j2 (root omega^n1)            -> out[i1 + i2 n1]  k_ntt_lds, strided views
// Three passes over HBM where the stage loop makes log2(n) + 2.
template <int NL, int NW>
__global__ void k_ntt_twist(const FpParams<NL> P, const uint32_t *__restrict__ tw, uint32_t *__restrict__ a, int n, int n2, int64_t count) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const int64_t i1 = e / n2, j2 = e - i1 * n2;
    const int64_t ex = (i1 * j2) & (int64_t)(n - 1);          // exponent mod n
    if (ex == 0) return;
    uint32_t x[NL], w[NL], r[NL];
    load_digits<NL, NW>(x, a + e * NW);
    const bool negw = ex >= n / 2;                              // omega^(n/2) = -1
    const uint32_t *wp = tw + (size_t)(negw ? ex - n / 2 : ex) * NL;
#pragma unroll
    for (int q = 0; q < NL; q++) w[q] = wp[q];
    mont_mul(r, x, w, P);                                       // canonical x times a Montgomery-form twiddle: canonical
    if (negw) fp_neg(r, r, P);
    store_digits<NL, NW>(a + e * NW, r);
}

// w^(2^e) mod p on the host with the kernels' own arithmetic (canonical limbs in and out)
template <int NL, int NW>
void host_pow2(const FpParams<NL> &P, const uint64_t *w_limbs, int e, uint64_t *out_limbs) {
    uint32_t ww[NW], d[NL], m[NL];
    for (int i = 0; i < NW; i++) ww[i] = (uint32_t)(w_limbs[i / 2] >> (32 * (i & 1)));
    unpack<NL, NW>(d, ww);
    to_mont(m, d, P);
    for (int i = 0; i < e; i++) mont_mul(m, m, m, P);
    from_mont(d, m, P);
    pack<NL, NW>(ww, d);
    for (int i = 0; i < NW / 2; i++) out_limbs[i] = (uint64_t)ww[2 * i] | ((uint64_t)ww[2 * i + 1] << 32);
}

}  // namespace

extern "C" {

int hb_fft_batch_evaluate(hb_ctx *ctx, const uint64_t *omega_host, int order, const uint64_t *coeffs_dev,
                          int64_t C, int d, int k, uint64_t *out_dev, void *stream) { HB_API_GUARD(ctx);
    if (!ctx || !omega_host || order <= 0 || (order & (order - 1)) || k < 0 || k > order || d < 0 || C < 0) return HB_ERR_BAD_ARG;
    if (C == 0 || k == 0) return HB_OK;
    if (!coeffs_dev || !out_dev) return HB_ERR_BAD_ARG;
    cache_trim(ctx);
    hipStream_t s = (hipStream_t)stream;
    const int n = order;
    int logn = 0; while ((1 << logn) < n) logn++;
    const int dd = d < n ? d : n;
    const int NLr = ctx->nl();
    // cost model in v_mad_u64_u32 per polynomial: lazy mat-vec vs radix-2 butterflies
    const double mads_prod = (double)NLr * NLr, mads_red = mads_prod + 4.0 * NLr;
    const double cost_mv = (double)k * dd * mads_prod + (double)k * mads_red;
    const double cost_ntt = 0.5 * n * logn * (mads_prod + mads_red);
    const bool table_ok = (double)k * dd <= 4.0e6;   // matrix kept small
    if (dd == 0 || (table_ok && cost_mv <= cost_ntt) || n == 1) {
        // Vandermonde matrix at x_i = omega^i, i < k, over the first dd coefficients
        std::string key = "W:" + std::to_string(n) + ":" + std::to_string(k) + ":" + std::to_string(dd) + ":";
        key.append(reinterpret_cast<const char *>(omega_host), (size_t)ctx->n_limbs * 8);
        hb_matrix *W = nullptr;
        auto it = ctx->mcache.find(key);
        if (it != ctx->mcache.end()) { W = it->second; cache_touch(ctx, "m|" + key); }
        else {
            uint32_t *xd = nullptr;
            int rc = pow_points_dev(ctx, omega_host, nullptr, k, &xd, s); if (rc) return rc;
            rc = vand_matrix_from_dev(ctx, key, xd, k, dd, &W, s);
            (void)hipFree(xd);
            if (rc) return rc;
        }
        hb_view iv{d, 1}, ov{k, 1};
        return launch_matvec(ctx, W, (const uint32_t *)coeffs_dev, iv, nullptr, INT64_MAX, (uint32_t *)out_dev, ov, INT64_MAX, nullptr, nullptr, C, s);
    }
    uint32_t *tw = nullptr;
    int rc = get_twiddles(ctx, omega_host, n, &tw, s); if (rc) return rc;
    const size_t elem_lds = (size_t)NLr * 4;
    if (((size_t)n + n / 2) * elem_lds <= 160 * 1024) {     // data + twiddles in LDS
        hb_view iv{d, 1}, ov{k, 1};
        return launch_ntt_lds(ctx, tw, n, (const uint32_t *)coeffs_dev, iv, INT64_MAX, d, k, (uint32_t *)out_dev, ov, INT64_MAX, nullptr, nullptr, C, s);
    }
    // large order: four steps over the LDS kernel (n = n1 n2, both factors at most 2048), one polynomial at a time
    if (logn <= 22 && !env_hook(ENV_NTT_STAGE_LOOP)) {
        const int l1 = (logn + 1) / 2, l2 = logn - l1, n1 = 1 << l1, n2 = 1 << l2;
        uint64_t w1[4] = {0, 0, 0, 0}, w2[4] = {0, 0, 0, 0};           // omega^n2 (order n1), omega^n1 (order n2)
        if (ctx->n_limbs == 4) { host_pow2<9, 8>(ctx->pw, omega_host, l2, w1); host_pow2<9, 8>(ctx->pw, omega_host, l1, w2); }
        else { host_pow2<3, 2>(ctx->pn, omega_host, l2, w1); host_pow2<3, 2>(ctx->pn, omega_host, l1, w2); }
        uint32_t *tw1 = nullptr, *tw2 = nullptr;
        rc = get_twiddles(ctx, w1, n1, &tw1, s); if (rc) return rc;
        rc = get_twiddles(ctx, w2, n2, &tw2, s); if (rc) return rc;
        const int NWr = ctx->elem_words();
        // the n-element scratch between the steps: kept per context, order and stream (a transform of this size is not called once)
        uint32_t *A = nullptr;
        {
            const std::string skey = "ntt4:" + std::to_string(n) + ":" + std::to_string((uintptr_t)stream);      // per stream: launches of one stream are ordered
            auto sit = ctx->dcache.find(skey);
            if (sit != ctx->dcache.end()) { A = (uint32_t *)sit->second; cache_touch(ctx, "d|" + skey); }
            else {
                // an entry of the context's bounded cache like every other table: the LRU trim and hb_ctx_cache_clear drop it (both synchronise the
                // device first), so a stream that was destroyed and whose address is reused cannot pin 128 MB per order for the context's life
                HB_HIP(ctx, hipMalloc(&A, (size_t)n * NWr * 4));
                ctx->dcache[skey] = A;
                cache_note(ctx, "d|" + skey, [ctx, skey]() { auto f = ctx->dcache.find(skey); if (f != ctx->dcache.end()) { (void)hipFree(f->second); ctx->dcache.erase(f); } });
            }
        }
        const int d1 = (dd + n2 - 1) / n2 < n1 ? (dd + n2 - 1) / n2 : n1;          // coefficients a column of the first step can hold
        const int rows3 = k < n1 ? k : n1;                                          // outputs i = i1 + i2 n1 < k need i1 < k
        for (int64_t c = 0; c < C && !rc; c++) {
            const uint32_t *in_c = (const uint32_t *)coeffs_dev + (size_t)c * d * NWr;
            uint32_t *out_c = (uint32_t *)out_dev + (size_t)c * k * NWr;
            rc = launch_ntt_lds(ctx, tw1, n1, in_c, hb_view{1, n2}, dd, d1, n1, A, hb_view{1, n2}, INT64_MAX, nullptr, nullptr, n2, s);
            if (rc) break;
            const int64_t cnt = (int64_t)n;
            if (ctx->n_limbs == 4) k_ntt_twist<9, 8><<<(unsigned)((cnt + 255) / 256), 256, 0, s>>>(ctx->pw, tw, A, n, n2, cnt);
            else k_ntt_twist<3, 2><<<(unsigned)((cnt + 255) / 256), 256, 0, s>>>(ctx->pn, tw, A, n, n2, cnt);
            rc = launch_ntt_lds(ctx, tw2, n2, A, hb_view{n2, 1}, INT64_MAX, n2, n2, out_c, hb_view{1, n1}, k, nullptr, nullptr, rows3, s);
        }
        if (rc) return rc;
        HB_LAUNCH_CHECK(ctx);
        return HB_OK;                                   // asynchronous like the LDS path: the scratch is only touched in stream order
    }
    // (HB_NTT_STAGE_LOOP=1, orders beyond 2^22: stage by stage over a digit buffer in HBM)
    if ((double)C * n * elem_lds > 64.0e9) return fail(ctx, HB_ERR_UNSUPPORTED, "fft: scratch too large");
    uint32_t *buf = nullptr;
    HB_HIP(ctx, hipMalloc(&buf, (size_t)C * n * elem_lds));
    const int64_t tot = C * n, hb_ = C * (n / 2), ko = C * k;
    if (ctx->n_limbs == 4) k_ntt_g_load<9, 8><<<(unsigned)((tot + 255) / 256), 256, 0, s>>>((const uint32_t *)coeffs_dev, d, n, logn, buf, C);
    else k_ntt_g_load<3, 2><<<(unsigned)((tot + 255) / 256), 256, 0, s>>>((const uint32_t *)coeffs_dev, d, n, logn, buf, C);
    for (int st = 0; st < logn; st++) {
        if (ctx->n_limbs == 4) k_ntt_g_stage<9><<<(unsigned)((hb_ + 255) / 256), 256, 0, s>>>(ctx->pw, tw, buf, n, st, C);
        else k_ntt_g_stage<3><<<(unsigned)((hb_ + 255) / 256), 256, 0, s>>>(ctx->pn, tw, buf, n, st, C);
    }
    if (ctx->n_limbs == 4) k_ntt_g_store<9, 8><<<(unsigned)((ko + 255) / 256), 256, 0, s>>>(buf, n, k, (uint32_t *)out_dev, C);
    else k_ntt_g_store<3, 2><<<(unsigned)((ko + 255) / 256), 256, 0, s>>>(buf, n, k, (uint32_t *)out_dev, C);
    HB_LAUNCH_CHECK(ctx);
    HB_HIP(ctx, hipStreamSynchronize(s));
    HB_HIP(ctx, hipFree(buf));
    return HB_OK;
}

int hb_fft_batch_interpolate(hb_ctx *ctx, const uint64_t *omega_host, int order, const int32_t *zs_host, int k,
                             const uint64_t *ys_dev, int64_t C, uint64_t *out_dev, void *stream) { HB_API_GUARD(ctx);
    if (!ctx || !omega_host || order <= 0 || (order & (order - 1)) || k < 0 || C < 0 || (k > 0 && !zs_host)) return HB_ERR_BAD_ARG;
    if (k == 0 || C == 0) return HB_OK;
    if (!ys_dev || !out_dev) return HB_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < k; i++) if (zs_host[i] < 0 || zs_host[i] >= order) return fail(ctx, HB_ERR_BAD_ARG, "zs out of range");
    cache_trim(ctx);
    // table for the sorted exponent set, columns fed through a permutation (arrival orders vary, sets less so)
    std::vector<int32_t> perm((size_t)k), zsorted((size_t)k);
    for (int i = 0; i < k; i++) perm[i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return zs_host[a] < zs_host[b]; });
    bool ident = true;
    for (int i = 0; i < k; i++) { zsorted[i] = zs_host[perm[i]]; if (perm[i] != i) ident = false; }
    std::string key = "WinvZ:" + std::to_string(order) + ":";
    key.append(reinterpret_cast<const char *>(omega_host), (size_t)ctx->n_limbs * 8);
    key.append(reinterpret_cast<const char *>(zsorted.data()), (size_t)k * 4);
    hb_matrix *Wi = nullptr;
    auto it = ctx->mcache.find(key);
    if (it != ctx->mcache.end()) { Wi = it->second; cache_touch(ctx, "m|" + key); }
    else {
        int32_t *zd = nullptr;
        int rc = get_int_array(ctx, zsorted.data(), k, &zd, s); if (rc) return rc;
        uint32_t *xd = nullptr;
        rc = pow_points_dev(ctx, omega_host, zd, k, &xd, s); if (rc) return rc;
        rc = vinv_from_dev(ctx, key, xd, k, &Wi, s);      // HB_ERR_SINGULAR <=> repeated z
        (void)hipFree(xd);
        if (rc) return rc;
    }
    int32_t *perm_dev = nullptr;
    if (!ident) { int rc = get_int_array(ctx, perm.data(), k, &perm_dev, s); if (rc) return rc; }
    hb_view v{k, 1};
    return launch_matvec(ctx, Wi, (const uint32_t *)ys_dev, v, perm_dev, INT64_MAX, (uint32_t *)out_dev, v, INT64_MAX, nullptr, nullptr, C, s);
}

}  // extern "C"
