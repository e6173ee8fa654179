// hb_mfma.hip -- third-generation mat-vec: batch (small-entry matrix) x (field elements) as an
// exact integer GEMM on the int8 matrix cores, followed by a Barrett reduction on the VALU.
//
//   out(c, i) = sum_l M[i][l] * in(c, l)  (mod p),   |M[i][l]| < 2^126, in(c, l) any 256-bit value
//
// This is NTL's mat_mul in vandermonde_batch_evaluate / vandermonde_batch_interpolate
// (reference honeybadgermpc/ntl/_ntl.pyx: vandermonde_batch_evaluate, "mul(result, vm, m)"),
// which *is* a matrix product; nothing is reshaped.  The integers are split in base 2^8:
//   in  = sum_a X_a 2^(8a)   a < 32   (the bytes of the packed element as they lie in HBM)
//   M   = sum_b M_b 2^(8b)   b < 16   (balanced signed digits, precomputed)
//   S   = sum_c 2^(8c) col_c,   col_c = sum_l sum_b M_b[i][l] * X_{c-b}[l]        c < 47
// so that for a fixed column c the sum over (l, b) is ONE int8 dot product of length 16*d between a
// constant row (the digits of M, register resident) and a 16-byte sliding window of every input
// element.  v_mfma_i32_16x16x64_i8 takes 16 rows i x 16 chunks x (4 elements l x 16 digits b) per
// instruction.  The window for column c is bytes [c-15, c] of the element; it is dword aligned for
// c = 3 mod 4 and otherwise built with v_alignbyte from the element's registers, once per residue
// class of c (27 ops per element per wave pass instead of 4 per MFMA).
//
// int8 operands are signed: M uses balanced digits; the input bytes are biased by XOR 0x80
// (u = s + 128) and the constant 128 * sum_a 2^(8a) * sum_l M[i][l] is added back per row, mod p,
// together with a multiple of p that keeps the total non-negative (crow[]).
//
// Epilogue per output: 47 int32 columns -> 13 carried 32-bit words -> 14 radix-2^29 digits ->
// Barrett (quotient from the top 6 digits x mu, 36 + 35 v_mad_u64_u32) -> two conditional
// subtractions -> packed 4 x u64.  No Montgomery form anywhere on this path.
#include "hb_common.hpp"

namespace hb {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int MM8_NC = 47;       // int32 columns per output
constexpr int MM8_CW = 13;       // 32-bit words of the per-row constant / the carried sum
constexpr int MM8_SD = 14;       // radix-2^29 digits of the carried sum (S < 2^388)

struct BarrettParams {
    uint32_t p[9];    // modulus digits
    uint32_t mu[6];   // floor(2^406 / p) digits
};

struct Mm8Matrix {
    int n_out, d, nkb, n_rt;
    int4 *a8;          // [n_rt][nkb][64 lanes] 16 signed digits each
    uint32_t *crow;    // [n_rt * 16][MM8_CW]
    BarrettParams bp;
};

typedef int v16i __attribute__((ext_vector_type(16)));
#include "hb_mm8_body.inc"

constexpr int MM8_BIAS = 5800000;   // > 352 * 128 * 128 >= |column|, and 2 * BIAS * 257 < 2^32

// Persistent workgroups of 4 waves, 2 workgroups per CU (2 waves per SIMD, <= 256 registers each,
// so that one wave's MFMA stream overlaps its neighbour's VALU epilogue).
//
// A unit is TPW tiles of 16 chunks.  Its input elements are DMA'd (global_load_lds_dwordx4) into one
// of two LDS buffers in MFMA-operand order -- slot (tile, kb, half)[lane] x 16 B, lane (n, g) owning
// element l = 4 kb + g of chunk n -- one unit ahead of the arithmetic.  Wave w of the workgroup takes
// row tile rt (16 output rows) of tile tl: (tl, rt) = (w / n_rt, w % n_rt) when n_rt divides 4,
// otherwise tl = 0 and rt = w, w + 4, ...
//
// Per (tile, rt) the 47 columns are produced in two halves (24 + 23 accumulators).  Accumulators
// start from BIAS so that every column is non-negative and the carry chain is unsigned 32-bit:
//   E_j = col_2j + (col_2j+1 << 8) < 2^32,  t = E_j + cy,  halfword_j = t & 0xffff,  cy = t >> 16
// The low half's words are parked (7 registers per output).  The bias, the XOR-0x80 correction and a
// multiple of p are one per-row constant (crowd), added digit-wise (radix 2^29) after the chain.
template <int NKB>
__global__ __launch_bounds__(256, 2) void k_mm8(const int4 *__restrict__ a8, const uint32_t *__restrict__ crowd,
                                                const uint32_t *__restrict__ in_pk, int64_t in_sc, int64_t in_sl,
                                                const int32_t *__restrict__ in_rows, int64_t in_count, int d,
                                                uint32_t *__restrict__ out_pk, int64_t out_sc, int64_t out_sl, int64_t out_count,
                                                int n_out, int n_rt, int tpw, int64_t n_chunks, int64_t n_units, BarrettParams bp) {
    extern __shared__ uint4 mm8_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    uint32_t *crl = reinterpret_cast<uint32_t *>(mm8_lds);     // [n_rt * 16][16] digits
    int4 *abuf = reinterpret_cast<int4 *>(mm8_lds + n_rt * 64);   // [n_rt][NKB][64] matrix digits
    uint4 *xbuf = mm8_lds + n_rt * 64 + n_rt * NKB * 64;        // 2 x [tpw][NKB][2][64] uint4
    const int bufsz = tpw * NKB * 2 * 64;
    for (int i = threadIdx.x; i < n_rt * 64; i += 256) mm8_lds[i] = reinterpret_cast<const uint4 *>(crowd)[i];
    for (int i = threadIdx.x; i < n_rt * NKB * 64; i += 256) abuf[i] = a8[i];

    const bool single = (tpw * n_rt == 4);
    const int tl = single ? wave / n_rt : 0;
    const int rt0 = single ? wave % n_rt : wave;
    const int rstep = single ? 1024 : 4;

    // l -> input row table (arrival order for decodes), clamped to d - 1
    int32_t *rowl = reinterpret_cast<int32_t *>(xbuf + 2 * bufsz);
    if (threadIdx.x < 32) { const int lc = (int)threadIdx.x < d ? (int)threadIdx.x : d - 1; rowl[threadIdx.x] = in_rows ? in_rows[lc] : lc; }
    __syncthreads();
    const int n_slots = tpw * NKB * 2;
    // slot s = (t * NKB + kb) * 2 + h, dealt round-robin to the 4 waves; all of it wave-uniform
    auto issue_loads = [&](int64_t unit, int buf) {
        for (int s = wave; s < n_slots; s += 4) {
            const int h = s & 1, q = s >> 1, t = q / NKB, kb = q - t * NKB;
            int64_t chunk = (unit * tpw + t) * 16 + n;
            if (chunk >= n_chunks) chunk = n_chunks - 1;
            int64_t idx = chunk * in_sc + (int64_t)rowl[4 * kb + g] * in_sl;
            if (idx >= in_count) idx = 0;
            const uint4 *src = reinterpret_cast<const uint4 *>(in_pk) + idx * 2 + h;
            // issued from asm so that hipcc does not drain it at the next LDS read (its vmcnt bookkeeping only
            // over-waits for ops it cannot see); the wait is the explicit vmcnt(0) before the epilogue
            const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(xbuf + (size_t)buf * bufsz + s * 64));
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
        }
    };
    const v4i biasv = v4i{MM8_BIAS, MM8_BIAS, MM8_BIAS, MM8_BIAS};

    int buf = 0;
    int64_t unit = blockIdx.x;
    if (unit < n_units) issue_loads(unit, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; unit < n_units; unit += gridDim.x, buf ^= 1) {
        // every wave has waited for its share of this unit's DMA (below, before its epilogue) and is
        // done reading the other buffer; no vmcnt wait here, so the output stores stay in flight
        __builtin_amdgcn_s_barrier();
        if (unit + gridDim.x < n_units) issue_loads(unit + gridDim.x, buf ^ 1);
        const int64_t chunk = (unit * tpw + tl) * 16 + n;
        const uint4 *xs = xbuf + (size_t)buf * bufsz + (size_t)tl * NKB * 2 * 64 + lane;
        for (int rt = rt0; rt < n_rt; rt += rstep) {
            const int4 *as = abuf + (size_t)rt * NKB * 64 + lane;
            uint32_t wlo[4][6];   // parked low words
            uint32_t cyp[4];      // parked carry
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int c0 = half ? 24 : 0, c1 = half ? MM8_NC : 24;
                v4i acc[24];
                asm volatile("" ::: "memory");
                if (half == 0) { MM8_MFMA_HALF0(NKB, xs, as, acc, biasv) } else { MM8_MFMA_HALF1(NKB, xs, as, acc, biasv) }
                if (half == 1 && rt + rstep >= n_rt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next unit's DMA (issued a pass ago)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    uint32_t cy = half ? cyp[reg] : 0u;
                    uint32_t t[12];
#pragma unroll
                    for (int j = 0; j < 12; j++) {
                        const int ca = c0 + 2 * j, cb = ca + 1;
                        uint32_t e = (uint32_t)acc[ca - c0][reg];
                        if (cb < c1) e += (uint32_t)acc[cb - c0][reg] << 8;
                        t[j] = e + cy;
                        cy = t[j] >> 16;
                    }
                    if (half == 0) {
#pragma unroll
                        for (int k = 0; k < 6; k++) wlo[reg][k] = __builtin_amdgcn_perm(t[2 * k + 1], t[2 * k], 0x05040100u);
                        cyp[reg] = cy;
                    } else {
                        uint32_t w[MM8_CW];
#pragma unroll
                        for (int k = 0; k < 6; k++) w[k] = wlo[reg][k];
#pragma unroll
                        for (int k = 0; k < 6; k++) w[6 + k] = __builtin_amdgcn_perm(t[2 * k + 1], t[2 * k], 0x05040100u);
                        w[12] = cy;
                        const int i = 16 * rt + 4 * g + reg;
                        uint32_t sd[MM8_SD];
#pragma unroll
                        for (int k = 0; k < MM8_SD; k++) {   // digit k = bits [29k, 29k + 29) of the 13 words
                            const int bit = LB * k, j = bit >> 5, sft = bit & 31;
                            const uint32_t lo = w[j], hi = (j + 1 < MM8_CW) ? w[j + 1] : 0u;
                            sd[k] = (sft == 0 ? lo : __builtin_amdgcn_alignbit(hi, lo, (uint32_t)sft)) & DMASK;
                        }
                        {
                            const uint4 *cr = reinterpret_cast<const uint4 *>(crl + (size_t)i * 16);
                            const uint4 c0v = cr[0], c1v = cr[1], c2v = cr[2], c3v = cr[3];
                            sd[0] += c0v.x; sd[1] += c0v.y; sd[2] += c0v.z; sd[3] += c0v.w;
                            sd[4] += c1v.x; sd[5] += c1v.y; sd[6] += c1v.z; sd[7] += c1v.w;
                            sd[8] += c2v.x; sd[9] += c2v.y; sd[10] += c2v.z; sd[11] += c2v.w;
                            sd[12] += c3v.x; sd[13] += c3v.y;
                        }
                        // Barrett: qhat = floor(floor(S / 2^232) * mu / 2^174) >= floor(S / p) - 2
                        uint64_t pc[12];
                        col_zero(pc);
#pragma unroll
                        for (int aa = 0; aa < 6; aa++)
#pragma unroll
                            for (int bb = 0; bb < 6; bb++) pc[aa + bb] += (uint64_t)sd[8 + aa] * bp.mu[bb];
                        carry(pc);
                        uint32_t qd[5];
#pragma unroll
                        for (int k = 0; k < 5; k++) qd[k] = (uint32_t)pc[6 + k];
                        uint64_t qp[9];
                        col_zero(qp);
#pragma unroll
                        for (int aa = 0; aa < 5; aa++)
#pragma unroll
                            for (int bb = 0; bb < 9; bb++)
                                if (aa + bb < 9) qp[aa + bb] += (uint64_t)qd[aa] * bp.p[bb];
                        uint32_t r[9];
                        int64_t br = 0;
#pragma unroll
                        for (int k = 0; k < 9; k++) {
                            int64_t tt = (int64_t)(uint64_t)sd[k] - (int64_t)qp[k] + br;
                            r[k] = (uint32_t)tt & DMASK;
                            br = tt >> LB;
                        }
#pragma unroll
                        for (int rep = 0; rep < 2; rep++) {
                            uint32_t tsub[9];
                            int32_t bw = 0;
#pragma unroll
                            for (int k = 0; k < 9; k++) {
                                const int32_t v = (int32_t)r[k] - (int32_t)bp.p[k] + bw;
                                bw = v >> 31;
                                tsub[k] = (k < 8) ? ((uint32_t)v & DMASK) : (uint32_t)v;
                            }
#pragma unroll
                            for (int k = 0; k < 9; k++) r[k] = bw ? r[k] : tsub[k];
                        }
                        uint32_t ow[8];
                        pack<9, 8>(ow, r);
                        const int64_t oidx = chunk * out_sc + (int64_t)i * out_sl;
                        if (chunk < n_chunks && i < n_out && oidx < out_count) store_words<8>(out_pk + oidx * 8, ow);
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one output at a time: interleaving them only costs registers
                }
            }
        }
    }
}

}  // namespace hb

using namespace hb;

// ---- prototype C entry points (digits and row constants supplied by the caller) ----------------
static int mm8_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

extern "C" int hb_mm8_create(hb_ctx *ctx, int n_out, int d, const int8_t *limbs /* [n_out][d][16] */,
                             const uint32_t *crowd /* [n_out][16] */, const uint32_t *mu6, void **out) {
    if (!ctx || ctx->n_limbs != 4 || d < 1 || d > 32 || n_out < 1) return HB_ERR_BAD_ARG;
    Mm8Matrix *m = new Mm8Matrix();
    m->n_out = n_out; m->d = d; m->nkb = (d + 3) / 4; m->n_rt = (n_out + 15) / 16;
    std::vector<int8_t> a((size_t)m->n_rt * m->nkb * 64 * 16, 0);
    for (int rt = 0; rt < m->n_rt; rt++)
        for (int kb = 0; kb < m->nkb; kb++)
            for (int lane = 0; lane < 64; lane++) {
                int i = 16 * rt + (lane & 15), l = 4 * kb + (lane >> 4);
                if (i >= n_out || l >= d) continue;
                for (int j = 0; j < 16; j++)
                    a[(((size_t)rt * m->nkb + kb) * 64 + lane) * 16 + j] = limbs[((size_t)i * d + l) * 16 + (15 - j)];
            }
    std::vector<uint32_t> cr((size_t)m->n_rt * 16 * 16, 0);
    memcpy(cr.data(), crowd, (size_t)n_out * 16 * 4);
    HB_HIP(ctx, hipMalloc(&m->a8, a.size()));
    HB_HIP(ctx, hipMalloc(&m->crow, cr.size() * 4));
    HB_HIP(ctx, hipMemcpy(m->a8, a.data(), a.size(), hipMemcpyHostToDevice));
    HB_HIP(ctx, hipMemcpy(m->crow, cr.data(), cr.size() * 4, hipMemcpyHostToDevice));
    for (int k = 0; k < 9; k++) m->bp.p[k] = ctx->pw.p[k];
    for (int k = 0; k < 6; k++) m->bp.mu[k] = mu6[k];
    *out = m;
    return HB_OK;
}

extern "C" int hb_mm8_apply(hb_ctx *ctx, void *mat, const void *in_dev, int64_t in_sc, int64_t in_sl, int64_t in_count,
                            void *out_dev, int64_t out_sc, int64_t out_sl, int64_t out_count, int64_t n_chunks) {
    Mm8Matrix *m = (Mm8Matrix *)mat;
    const int tpw = (m->n_rt == 1) ? 4 : (m->n_rt == 2) ? 2 : 1;
    const int64_t n_tiles = (n_chunks + 15) / 16;
    const int64_t n_units = (n_tiles + tpw - 1) / tpw;
    int64_t blocks = 2 * (int64_t)mm8_num_cus();
    if (blocks > n_units) blocks = n_units;
    const size_t lds = ((size_t)m->n_rt * 64 + (size_t)m->n_rt * m->nkb * 64 + (size_t)2 * tpw * m->nkb * 2 * 64) * 16 + 128;
#define MM8_LAUNCH(NKB)                                                                                          \
    do {                                                                                                          \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_mm8<NKB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_mm8<NKB>), dim3((unsigned)blocks), dim3(256), lds, 0, m->a8, m->crow, (const uint32_t *)in_dev, in_sc,  \
                           in_sl, (const int32_t *)nullptr, in_count, m->d, (uint32_t *)out_dev, out_sc, out_sl, out_count, \
                           m->n_out, m->n_rt, tpw, n_chunks, n_units, m->bp);                                    \
    } while (0)
    switch (m->nkb) {
        case 1: MM8_LAUNCH(1); break;
        case 2: MM8_LAUNCH(2); break;
        case 3: MM8_LAUNCH(3); break;
        case 4: MM8_LAUNCH(4); break;
        case 5: MM8_LAUNCH(5); break;
        case 6: MM8_LAUNCH(6); break;
        case 7: MM8_LAUNCH(7); break;
        case 8: MM8_LAUNCH(8); break;
        default: return HB_ERR_BAD_ARG;
    }
#undef MM8_LAUNCH
    HB_LAUNCH_CHECK(ctx);
    return HB_OK;
}
