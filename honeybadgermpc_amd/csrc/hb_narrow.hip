// hb_narrow.hip -- batched mat-vec over GF(p) for WORD-SIZE primes (p < 2^64, 8-byte elements): the 64-bit prime of the north star.
//
// Replaces, for one-limb contexts, the same reference calls as the wide kernels: NTL's mat_ZZ_p mul in vandermonde_batch_evaluate /
// vandermonde_batch_interpolate (hbmpc_ntl_helpers.pyx:183,237) and IncrementalDecoder's decode + re-encode + compare
// (reed_solomon.py:305-326) as ONE product with [rows of V^-1(z) ; V[zc] V^-1(z)] (the value the guess takes at a later arrival's point is
// linear in the arrival set; same canonical values, same accept / reject).
//
// Rounds 1-4 ran one-limb contexts on the <3 digits, 2 words> instantiation of the generic radix-2^29 kernels: 9 multiply-adds a product, a
// Montgomery reduction per digit column, pre-scale passes through HBM -- bench.py --workload cfg3-p64 measured 6.6 G shares/s, BELOW the
// 256-bit path's 7.2 G, at 0.11 of the HBM roofline.  An 8-byte element does not need any of that:
//
//   * lane = chunk (party-major buffers: a wave reads / writes 512 contiguous bytes per row); a thread keeps its chunk's d inputs in
//     registers as THREE digits (22 + 22 + 20 bits) and walks the output rows of its row group;
//   * a matrix entry is wave-uniform: two 32-bit halves in SGPRs (scalar loads, three 64-byte loads a row), so a 64 x 64-bit product is SIX
//     v_mad_u64_u32 with a scalar operand into six 64-bit accumulators, one per digit weight (2^0, 2^22, 2^44 | 2^32, 2^54, 2^76): 40
//     products of 2^54 fit, no carry inside the dot product.  (Measured alternatives, each parity-green: the inputs kept as two halves and the
//     ENTRY cut into three digits on the scalar unit -- 112 registers instead of 160, 40 % SLOWER: one scalar unit serves a CU's four SIMDs
//     and 5 scalar operations an entry saturate it; the entry's three digits laid out by the host -- 72 scalar words a row do not fit the
//     scalar file, the compiler falls back to single-word loads and 212 vector registers; the row group's entries as three digits in LDS,
//     read by broadcast against the inputs' two halves -- 124 registers, four waves a SIMD, 102 us an open against 88: the staged inputs
//     and the entries share the LDS.  A term-major nest -- a thread accumulating all rows of its group, inputs and entries fetched a term
//     ahead -- 27.7 us against 25.6 for the R2 launch.  PMC: 8.4 M vector instructions in that launch are 15 us of issue; a row is 144
//     multiply-adds and ~100 instructions of assembling and reducing its 134-bit sum.)
//   * entries are kept as M 2^128 mod p, so the sum (< 40 p 2^64) comes back through four 32-bit Montgomery steps (R = 2^128) and one
//     conditional subtraction: canonical output, no pre-scale, nothing but the inputs and the outputs touches HBM;
//   * per row a mode as in hb_mfma_fused.hip: store (a coefficient row / an encoded row) or compare with the received row of a later arrival.
//
// Shapes: d <= 40 terms (the inputs live in registers), any number of rows, any odd p < 2^64 (the sum is below 40 p 2^64 for inputs that are ANY
// 64-bit words, so R = 2^128 brings it below p + 40 / 2^64 p < 2 p).  Wider products stay on the generic kernels.
#include <algorithm>

#include "hb_common.hpp"

namespace hb {

constexpr int MV64_DMAX = 40;
constexpr int MV64_RESIDENT = 768;       // 3 waves a SIMD (160 registers) x 1024 SIMDs / 4 waves a workgroup (1024: 97 us an open instead of 88)

struct Mv64Params { uint32_t p0, p1, pinv32; };          // p = p1 2^32 + p0; pinv32 = -p^-1 mod 2^32

// the six accumulators of a row -> the canonical residue -> stored or compared (shared by the two kernels)
struct Mv64Out { const uint64_t *in; int64_t in_sc, in_sl; uint64_t *out; int64_t out_sc, out_sl, out_count; int32_t *mismatch, *first_bad; };
__device__ __forceinline__ void mv64_finish(uint64_t a0, uint64_t a1, uint64_t a2, uint64_t b0, uint64_t b1, uint64_t b2, const Mv64Params &prm, int md, int64_t chunk,
                                            int64_t cc, bool live, const Mv64Out &o) {
    const uint64_t *in = o.in; uint64_t *out = o.out;
    const int64_t in_sc = o.in_sc, in_sl = o.in_sl, out_sc = o.out_sc, out_sl = o.out_sl, out_count = o.out_count;
    int32_t *mismatch = o.mismatch, *first_bad = o.first_bad;
    {
        // S = a0 + a1 2^22 + a2 2^44 + b0 2^32 + b1 2^54 + b2 2^76 as six 32-bit words (S < 40 p 2^64 < 2^134)
    uint32_t w[6];
    {
        // 64-bit pieces by word offset: offset 0: a0 + (a1 << 22) low ...; done with 128-bit-free carries
        unsigned __int128 s = (unsigned __int128)a0 + ((unsigned __int128)a1 << 22) + ((unsigned __int128)a2 << 44) + ((unsigned __int128)b0 << 32) +
                              ((unsigned __int128)b1 << 54);
        // b2 << 76 does not fit 128 bits with the rest: split off the top
        const unsigned __int128 hi = (unsigned __int128)b2 << 12;        // weight 2^64
        const uint64_t lo64 = (uint64_t)s;
        unsigned __int128 up = (s >> 64) + hi;                           // < 2^72
        w[0] = (uint32_t)lo64; w[1] = (uint32_t)(lo64 >> 32);
        w[2] = (uint32_t)up; w[3] = (uint32_t)(up >> 32); w[4] = (uint32_t)(up >> 64); w[5] = 0;
    }
    // four Montgomery steps of 32 bits: S <- (S + u p) / 2^32, u = w0 pinv32 mod 2^32
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t u = w[0] * prm.pinv32;
        uint64_t t = (uint64_t)u * prm.p0 + w[0];                        // low word becomes 0
        t = (t >> 32) + (uint64_t)u * prm.p1 + w[1];
        w[0] = (uint32_t)t;
        t = (t >> 32) + w[2];
        w[1] = (uint32_t)t;
        t = (t >> 32) + w[3];
        w[2] = (uint32_t)t;
        t = (t >> 32) + w[4];
        w[3] = (uint32_t)t;
        w[4] = (uint32_t)(t >> 32);
    }
    // result < 40 p / 2^64 + p < 2 p in (w0, w1, w2 <= 1)
    uint64_t r = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    const uint64_t pp = (uint64_t)prm.p0 | ((uint64_t)prm.p1 << 32);
    if (w[2] || r >= pp) r -= pp;
    if (md < 0) {
        const int64_t oidx = chunk * out_sc + (int64_t)(-md - 1) * out_sl;
        if (live && oidx < out_count) out[oidx] = r;
    } else {
        const uint64_t got = in[live ? cc * in_sc + (int64_t)(md - 1) * in_sl : 0];
        const bool bad = live && got != r;
        const unsigned long long bl = __builtin_amdgcn_ballot_w64(bad);
        if (bl && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(bl)) {
            if (*reinterpret_cast<volatile int32_t *>(mismatch) == 0) atomicOr(mismatch, 1);
            if (first_bad && *reinterpret_cast<volatile int32_t *>(first_bad) > (int32_t)chunk) atomicMin(first_bad, (int32_t)chunk);
        }
    }
    }
}

// rowmode[i]: 0 = nothing, v > 0: compare the result with row v - 1 of the input buffer, v < 0: store it as output row -v - 1
template <int DT, bool STAGE>
__global__ __launch_bounds__(256) void k_mv64(const uint2 *__restrict__ M, const int32_t *__restrict__ rowmode, int n_out, int d, int rows_per_group,
                                              const uint64_t *__restrict__ in, int64_t in_sc, int64_t in_sl, const int32_t *__restrict__ in_rows, int64_t in_count,
                                              uint64_t *__restrict__ out, int64_t out_sc, int64_t out_sl, int64_t out_count,
                                              int32_t *__restrict__ mismatch, int32_t *__restrict__ first_bad, int64_t C, const Mv64Params prm) {
    const int64_t chunk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = chunk < C;
    const int64_t cc = live ? chunk : C - 1;
    uint32_t x0[DT], x1[DT], x2[DT];
    extern __shared__ uint64_t mv_lds[];
    if (STAGE) {
        // chunk-major input (a share vector: chunk c = elements [c d, (c + 1) d)): the block's 256 d elements are contiguous -- read coalesced into LDS
        // (row stride d | 1 words of 8 bytes: a thread's own d elements are then read without bank conflicts beyond the two-cycle b64 access)
        const int ls = d | 1;
        const int64_t base = (int64_t)blockIdx.x * 256 * d;
        for (int e = threadIdx.x; e < 256 * d; e += 256) {
            const int64_t idx = base + e;
            mv_lds[(e / d) * ls + (e % d)] = idx < in_count ? in[idx] : 0ull;
        }
        __syncthreads();
#pragma unroll
        for (int l = 0; l < DT; l++) {
            const uint64_t v = l < d ? mv_lds[threadIdx.x * ls + l] : 0ull;
            x0[l] = (uint32_t)v & 0x3fffffu;
            x1[l] = (uint32_t)(v >> 22) & 0x3fffffu;
            x2[l] = (uint32_t)(v >> 44);
        }
    } else {
#pragma unroll
        for (int l = 0; l < DT; l++) {
            uint64_t v = 0;
            if (l < d) {
                const int64_t idx = cc * in_sc + (int64_t)(in_rows ? in_rows[l] : l) * in_sl;
                if (idx < in_count) v = in[idx];
            }
            x0[l] = (uint32_t)v & 0x3fffffu;
            x1[l] = (uint32_t)(v >> 22) & 0x3fffffu;
            x2[l] = (uint32_t)(v >> 44);
        }
    }
    const int row_lo = blockIdx.y * rows_per_group, row_hi = min(row_lo + rows_per_group, n_out);
    for (int i = row_lo; i < row_hi; i++) {
        const int md = rowmode[i];
        if (md == 0) continue;
        // the row's DT entries (rows are padded to DT with zeros: no branch inside the dot product, and the scalar loads of a row are wide)
        // the row's DT entries (rows are padded to DT with zeros: no branch inside the dot product, and the scalar loads of a row are wide)
        const uint2 *__restrict__ mrow = M + (size_t)i * DT;
        uint2 mr[DT];
#pragma unroll
        for (int l = 0; l < DT; l++) mr[l] = mrow[l];       // wave-uniform: scalar loads
        uint64_t a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
#pragma unroll
        for (int l = 0; l < DT; l++) {
            a0 += (uint64_t)mr[l].x * x0[l];
            a1 += (uint64_t)mr[l].x * x1[l];
            a2 += (uint64_t)mr[l].x * x2[l];
            b0 += (uint64_t)mr[l].y * x0[l];
            b1 += (uint64_t)mr[l].y * x1[l];
            b2 += (uint64_t)mr[l].y * x2[l];
        }
        mv64_finish(a0, a1, a2, b0, b1, b2, prm, md, chunk, cc, live, Mv64Out{in, in_sc, in_sl, out, out_sc, out_sl, out_count, mismatch, first_bad});
    }
}

// rows are padded to DT terms; 22 is config 3's degree + 1 (a row of 24 is 9 % of multiply-adds on zeros)
static int mv64_dt(int d) { return d <= 8 ? 8 : (d <= 16 ? 16 : (d <= 22 ? 22 : (d <= 24 ? 24 : (d <= 32 ? 32 : MV64_DMAX)))); }

struct Mv64Matrix {
    int n_out, d;
    uint2 *M;            // [n_out][DT], DT = mv64_dt(d): (M[i][l] 2^128 mod p) as (low, high) 32-bit halves, rows zero padded
    int32_t *mode;       // [n_out]
};

static inline uint64_t mulmod_u64(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)(((unsigned __int128)a * b) % p); }

bool mv64_applies(const hb_ctx *ctx, int d) {
    return ctx->n_limbs == 1 && ctx->p_limbs[0] >= 3 && d >= 1 && d <= MV64_DMAX && !env_hook(ENV_NO_NARROW_FAST);
}

void mv64_free(Mv64Matrix *m) {
    if (!m) return;
    if (m->M) (void)hipFree(m->M);
    if (m->mode) (void)hipFree(m->mode);
    delete m;
}

// m_host: n_out x d canonical residues, row-major; mode_host: the per-row modes
int mv64_from_host(hb_ctx *ctx, const uint64_t *m_host, int n_out, int d, const int32_t *mode_host, Mv64Matrix **out, hipStream_t s) {
    *out = nullptr;
    if (!mv64_applies(ctx, d) || n_out < 1) return HB_ERR_UNSUPPORTED;
    const uint64_t p = ctx->p_limbs[0];
    const uint64_t r64 = (uint64_t)(((unsigned __int128)1 << 64) % p), r128 = mulmod_u64(r64, r64, p);
    const int dt = mv64_dt(d);
    std::vector<uint2> img((size_t)n_out * dt, make_uint2(0, 0));
    for (int i = 0; i < n_out; i++)
        for (int l = 0; l < d; l++) {
            const uint64_t v = mulmod_u64(m_host[(size_t)i * d + l] % p, r128, p);
            img[(size_t)i * dt + l] = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
        }
    Mv64Matrix *m = new Mv64Matrix();
    m->n_out = n_out; m->d = d; m->M = nullptr; m->mode = nullptr;
    hipError_t e = hipMalloc(&m->M, img.size() * sizeof(uint2));
    if (e == hipSuccess) e = hipMalloc(&m->mode, (size_t)n_out * 4);
    if (e != hipSuccess) { mv64_free(m); ctx->err = std::string("narrow matrix: ") + hipGetErrorString(e); return HB_ERR_HIP; }
    int rc = upload_table(ctx, m->M, img.data(), img.size() * sizeof(uint2), s);
    if (!rc) rc = upload_table(ctx, m->mode, mode_host, (size_t)n_out * 4, s);
    if (rc) { mv64_free(m); return rc; }
    *out = m;
    return HB_OK;
}

int launch_mv64(hb_ctx *ctx, const Mv64Matrix *m, const uint64_t *in, hb_view iv, const int32_t *in_rows_dev, int64_t in_count, uint64_t *out, hb_view ov,
                int64_t out_count, int32_t *mismatch_dev, int32_t *first_bad_dev, int64_t C, hipStream_t s) {
    if (C <= 0) return HB_OK;
    const uint64_t p = ctx->p_limbs[0];
    Mv64Params prm;
    prm.p0 = (uint32_t)p; prm.p1 = (uint32_t)(p >> 32);
    uint32_t inv = 1;                                   // Newton: p^-1 mod 2^32
    for (int k = 0; k < 5; k++) inv *= 2u - prm.p0 * inv;
    prm.pinv32 = 0u - inv;
    // The rows are cut into groups so that ONE round of resident workgroups covers the launch (a second, partly filled round costs as much as the
    // first): MV64_RESIDENT workgroups of 256 fit the chip at this kernel's register count
    const int64_t cblocks = (C + 255) / 256;
    int groups = (int)std::max<int64_t>(1, std::min<int64_t>(m->n_out, MV64_RESIDENT / std::max<int64_t>(cblocks, 1)));
    int rpg = (m->n_out + groups - 1) / groups;
    groups = (m->n_out + rpg - 1) / rpg;
    const dim3 grid((unsigned)cblocks, (unsigned)groups);
    // chunk-major input read whole (a share vector): staged through LDS, coalesced
    const bool stage = !in_rows_dev && iv.stride_l == 1 && iv.stride_c == m->d && m->d > 1;
    const size_t lds = stage ? (size_t)256 * (m->d | 1) * 8 : 0;
#define MV64_LAUNCH(DT)                                                                                                                                  \
    do {                                                                                                                                                 \
        if (stage) k_mv64<DT, true><<<grid, 256, lds, s>>>(m->M, m->mode, m->n_out, m->d, rpg, in, iv.stride_c, iv.stride_l, in_rows_dev, in_count, out, \
                                                           ov.stride_c, ov.stride_l, out_count, mismatch_dev, first_bad_dev, C, prm);                    \
        else k_mv64<DT, false><<<grid, 256, 0, s>>>(m->M, m->mode, m->n_out, m->d, rpg, in, iv.stride_c, iv.stride_l, in_rows_dev, in_count, out,        \
                                                    ov.stride_c, ov.stride_l, out_count, mismatch_dev, first_bad_dev, C, prm);                           \
    } while (0)
    if (m->d <= 8) MV64_LAUNCH(8);
    else if (m->d <= 16) MV64_LAUNCH(16);
    else if (m->d <= 22) MV64_LAUNCH(22);
    else if (m->d <= 24) MV64_LAUNCH(24);
    else if (m->d <= 32) MV64_LAUNCH(32);
    else MV64_LAUNCH(40);
#undef MV64_LAUNCH
    HB_LAUNCH_CHECK(ctx);
    return HB_OK;
}

// host-side tables of a plan over a word-size prime: V (n x d), V(z)^-1 by Gauss-Jordan, V[zc] V(z)^-1.  HB_ERR_SINGULAR for repeated points.
int mv64_plan_tables(hb_ctx *ctx, const uint64_t *x, int n, int d, const int32_t *z, const int32_t *zc, int nc, std::vector<uint64_t> &V, std::vector<uint64_t> &Vinv,
                     std::vector<uint64_t> &P) {
    const uint64_t p = ctx->p_limbs[0];
    V.assign((size_t)n * d, 0);
    for (int i = 0; i < n; i++) {
        uint64_t pw = 1 % p;
        const uint64_t xi = x[i] % p;
        for (int l = 0; l < d; l++) { V[(size_t)i * d + l] = pw; pw = mulmod_u64(pw, xi, p); }
    }
    std::vector<uint64_t> A((size_t)d * 2 * d, 0);
    for (int i = 0; i < d; i++)
        for (int l = 0; l < d; l++) { A[(size_t)i * 2 * d + l] = V[(size_t)z[i] * d + l]; A[(size_t)i * 2 * d + d + l] = (i == l) ? 1 % p : 0; }
    auto powmod = [&](uint64_t a, uint64_t e) { uint64_t r = 1 % p; while (e) { if (e & 1) r = mulmod_u64(r, a, p); a = mulmod_u64(a, a, p); e >>= 1; } return r; };
    for (int col = 0; col < d; col++) {
        int piv = -1;
        for (int r = col; r < d; r++) if (A[(size_t)r * 2 * d + col]) { piv = r; break; }
        if (piv < 0) return fail(ctx, HB_ERR_SINGULAR, "Interpolation failed: repeated points");
        if (piv != col) for (int l = 0; l < 2 * d; l++) std::swap(A[(size_t)col * 2 * d + l], A[(size_t)piv * 2 * d + l]);
        const uint64_t inv = powmod(A[(size_t)col * 2 * d + col], p - 2);
        for (int l = 0; l < 2 * d; l++) A[(size_t)col * 2 * d + l] = mulmod_u64(A[(size_t)col * 2 * d + l], inv, p);
        for (int r = 0; r < d; r++) {
            if (r == col) continue;
            const uint64_t f = A[(size_t)r * 2 * d + col];
            if (!f) continue;
            for (int l = 0; l < 2 * d; l++) {
                const uint64_t a = A[(size_t)r * 2 * d + l], sub = mulmod_u64(f, A[(size_t)col * 2 * d + l], p);
                A[(size_t)r * 2 * d + l] = a >= sub ? a - sub : a + (p - sub);
            }
        }
    }
    Vinv.assign((size_t)d * d, 0);
    for (int i = 0; i < d; i++) for (int l = 0; l < d; l++) Vinv[(size_t)i * d + l] = A[(size_t)i * 2 * d + d + l];
    P.assign((size_t)nc * d, 0);
    for (int j = 0; j < nc; j++)
        for (int l = 0; l < d; l++) {
            unsigned __int128 acc = 0;
            for (int m = 0; m < d; m++) acc += (unsigned __int128)mulmod_u64(V[(size_t)zc[j] * d + m], Vinv[(size_t)m * d + l], p);
            P[(size_t)j * d + l] = (uint64_t)(acc % p);
        }
    return HB_OK;
}

}  // namespace hb
