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

// whole transform in LDS; PB polynomials per block.  Strided views on both sides:
//   in(c, j)  at in  + (c*in_sc  + j*in_sl ) elements, zero beyond in_count (chunk_data padding)
//   out(c, i) at out + (c*out_sc + i*out_sl) elements, i < k, skipped beyond out_count
// `*_poly_fast` picks the thread->element map so that consecutive lanes touch consecutive
// addresses for coefficient-major / party-major buffers (stride_c == 1).
// CHECK: instead of storing, compare out(c, i) with the buffer for rows i in check_mask
// (the validating re-encode of IncrementalDecoder, reed_solomon.py:313-326).
template <int NL, int NW, bool CHECK>
__global__ void __launch_bounds__(256) k_ntt_lds(const FpParams<NL> P, const uint32_t *__restrict__ tw,
                                                 const uint32_t *__restrict__ in, int64_t in_sc, int64_t in_sl, int64_t in_count, int d,
                                                 int n, int logn, int k,
                                                 uint32_t *__restrict__ out, int64_t out_sc, int64_t out_sl, int64_t out_count,
                                                 const int32_t *__restrict__ check_mask, int32_t *__restrict__ mismatch,
                                                 int64_t C, int PB, int in_poly_fast, int out_poly_fast) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int64_t c0 = (int64_t)blockIdx.x * PB;
    const int npoly = (int)min((int64_t)PB, C - c0);
    const int dd = min(d, n);
    for (int idx = threadIdx.x; idx < npoly * n * NL; idx += blockDim.x) lds[idx] = 0;
    __syncthreads();
    // load the dd coefficients of every polynomial into bit-reversed positions
    for (int idx = threadIdx.x; idx < npoly * dd; idx += blockDim.x) {
        const int pl = in_poly_fast ? idx % npoly : idx / dd;
        const int j = in_poly_fast ? idx / npoly : idx % dd;
        const int64_t e = (c0 + pl) * in_sc + (int64_t)j * in_sl;
        if (e < in_count) {
            uint32_t dg[NL];
            load_digits<NL, NW>(dg, in + e * NW);
            uint32_t *dst = lds + ((size_t)pl * n + bitrev((uint32_t)j, logn)) * NL;
#pragma unroll
            for (int q = 0; q < NL; q++) dst[q] = dg[q];
        }
    }
    __syncthreads();
    const int half = n >> 1;
    for (int s = 0; s < logn; s++) {
        const int h = 1 << s;
        const int tstride = half >> s;          // twiddle index step: n / (2h)
        // Input pruning.  After s stages the block at positions [m 2^s, (m+1) 2^s) holds the sub-transform of the
        // coefficients j = r (mod n / 2^s), r = bitrev(m); with only dd < n coefficients present it is identically zero
        // when r >= dd.  The blocks of a stage are visited in bit-reversed order q (butterflies of a stage are
        // independent), which makes that test monotone: the lower operand's class is r_v = nblk + q, so butterflies with
        // q >= dd - nblk have v = 0 and degenerate to a copy (u, u) -- whole waves take one side of the branch.
        const int lb = logn - s - 1, nblk = n >> (s + 1);
        const int live = dd - nblk;             // blocks q < live have a non-zero lower operand
        for (int b = threadIdx.x; b < npoly * half; b += blockDim.x) {
            const int pl = b / half, bb = b % half;
            const int j = bb & (h - 1);
            const int q = bb >> s;
            const int m = lb > 0 ? (int)bitrev((uint32_t)q, lb) : 0;
            const int i0 = (m << (s + 1)) + j;
            uint32_t *base = lds + (size_t)pl * n * NL;
            if (q < live) {
                butterfly<NL>(base + (size_t)i0 * NL, base + (size_t)(i0 + h) * NL, tw + (size_t)j * tstride * NL, j == 0, P);
            } else {
#pragma unroll
                for (int qd = 0; qd < NL; qd++) base[(size_t)(i0 + h) * NL + qd] = base[(size_t)i0 * NL + qd];
            }
        }
        __syncthreads();
    }
    for (int idx = threadIdx.x; idx < npoly * k; idx += blockDim.x) {
        const int pl = out_poly_fast ? idx % npoly : idx / k;
        const int i = out_poly_fast ? idx / npoly : idx % k;
        const int64_t e = (c0 + pl) * out_sc + (int64_t)i * out_sl;
        uint32_t dg[NL];
#pragma unroll
        for (int q = 0; q < NL; q++) dg[q] = lds[((size_t)pl * n + i) * NL + q];
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
            if (e < out_count) store_digits<NL, NW>(out + e * NW, dg);
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

// LDS NTT launcher with views; returns HB_ERR_UNSUPPORTED when the order does not fit LDS
int launch_ntt_lds(hb_ctx *ctx, const uint32_t *tw, int n, const uint32_t *in, hb_view iv, int64_t in_count, int d, int k,
                   uint32_t *out, hb_view ov, int64_t out_count, const int32_t *check_mask_dev, int32_t *mismatch_dev,
                   int64_t C, hipStream_t s) {
    if (C <= 0 || k <= 0) return HB_OK;
    int logn = 0; while ((1 << logn) < n) logn++;
    const size_t elem_lds = (size_t)ctx->nl() * 4;
    if ((size_t)n * elem_lds > 160 * 1024) return fail(ctx, HB_ERR_UNSUPPORTED, "ntt: order does not fit LDS");
    int PB = (int)((40 * 1024) / ((size_t)n * elem_lds)); if (PB < 1) PB = 1; if (PB > 64) PB = 64;
    // whole rounds of butterflies: PB * n/2 a multiple of the 256 threads (17 polynomials of order 64 would run 3 rounds, the third 12 % full)
    { const int q = n >= 2 ? 256 / (n / 2) : 1; if (q > 1 && PB > q) PB -= PB % q; }
    if ((int64_t)PB > C) PB = (int)C;
    const size_t lds = (size_t)PB * n * elem_lds;
    const int64_t blocks = (C + PB - 1) / PB;
    if (blocks > 0x7fffffffLL) return fail(ctx, HB_ERR_UNSUPPORTED, "fft: batch too large");
    const int ipf = iv.stride_c == 1 ? 1 : 0, opf = ov.stride_c == 1 ? 1 : 0;
    const bool check = check_mask_dev != nullptr;
#define HB_NTT(NL_, NW_, CHK_, PP_)                                                                                                       \
    do {                                                                                                                                  \
        HB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_ntt_lds<NL_, NW_, CHK_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024))); \
        k_ntt_lds<NL_, NW_, CHK_><<<(unsigned)blocks, 256, lds, s>>>(PP_, tw, in, iv.stride_c, iv.stride_l, in_count, d, n, logn, k, out, ov.stride_c, ov.stride_l, out_count, check_mask_dev, mismatch_dev, C, PB, ipf, opf); \
    } while (0)
    if (ctx->n_limbs == 4) { if (check) HB_NTT(9, 8, true, ctx->pw); else HB_NTT(9, 8, false, ctx->pw); }
    else { if (check) HB_NTT(3, 2, true, ctx->pn); else HB_NTT(3, 2, false, ctx->pn); }
#undef HB_NTT
    HB_LAUNCH_CHECK(ctx);
    return HB_OK;
}

}  // namespace hb

namespace {
}  // namespace

extern "C" {

int hb_fft_batch_evaluate(hb_ctx *ctx, const uint64_t *omega_host, int order, const uint64_t *coeffs_dev,
                          int64_t C, int d, int k, uint64_t *out_dev, void *stream) {
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
    if ((size_t)n * elem_lds <= 160 * 1024) {
        hb_view iv{d, 1}, ov{k, 1};
        return launch_ntt_lds(ctx, tw, n, (const uint32_t *)coeffs_dev, iv, INT64_MAX, d, k, (uint32_t *)out_dev, ov, INT64_MAX, nullptr, nullptr, C, s);
    }
    // large order: stage-by-stage over a digit buffer in HBM
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
                             const uint64_t *ys_dev, int64_t C, uint64_t *out_dev, void *stream) {
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
