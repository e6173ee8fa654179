// hb_mfma_wide.hip -- the matrix-core mat-vec for FULL-SIZE matrix entries.
//
//   out(c, i) = sum_l M[i][l] * in(c, rows[l])  (mod p),   M[i][l] any residue, in(c, l) any 256-bit value
//
// Replaces NTL's mat_ZZ_p mul wherever the matrix entries are not small integers
// (reference honeybadgermpc/ntl/hbmpc_ntl_helpers.pyx:183,237 through rsdecode_impl.h:23-36,97-122):
//   * the inverse Vandermonde matrix at omega-power points (fft_interpolate / fft_batch_interpolate,
//     rsdecode_impl.h:194-265 -- any interpolation algorithm yields the same canonical coefficients);
//   * Vandermonde matrices at the points 1..n whose powers outgrow 2^127 (n = 100, t = 33: 100^33 > 2^219);
//   * arbitrary hb_matrix operands (hb_matvec), the interpolant of gao_interpolate (rsdecode_impl.h:281-405).
// hb_mfma.hip covers the small-entry case (16 digits, 47 columns, VALU-bound by its reduction); here the
// entries have 32 base-256 digits, the sum has 63 int32 columns, and the kernel is matrix-pipe bound:
// 94 v_mfma_i32_16x16x64_i8 per block of 4 terms and 16 x 16 outputs (gen_mm8w.py emits that phase).
//
// One wave per SIMD (512-register budget): all 63 accumulators of a 16-chunk x 16-row pass live in AGPRs.
// A workgroup owns a unit of `tpw` tiles of 16 chunks, DMA'd into LDS in MFMA-operand order; its 4 waves take the
// (tile, row tile) pairs.  The digits stream from L2 inside the MFMA phase.  Epilogue per output (VALU):
// 63 columns + bias -> 16 base-2^32 groups (3 v_mad_u64_u32 each) -> 17 words -> 19 radix-2^29 digits ->
// fold the ten high digits through T_k = 2^(29k) mod p (90 MADs) + per-row constant -> two-digit Barrett quotient
// (as k_prescale_tab, hb_fast.hip) -> conditional subtraction -> packed canonical element.
// Inputs are biased by XOR 0x80 (int8 operands are signed); the per-row constant takes that and the accumulator
// bias back out, mod p.  No Montgomery form anywhere.
#include <algorithm>

#include "hb_common.hpp"

namespace hb {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int MM8W_NC = 63;     // int32 columns per output
constexpr int MM8W_WORDS = 17;  // 32-bit words of the biased sum
constexpr int MM8W_SD = 19;     // its radix-2^29 digits

struct WideParams {
    uint32_t T[10][9];   // 2^(29 (9 + k)) mod p, digits
    uint32_t pbar[9];    // 2^261 - p, digits
    uint32_t pneg[8];    // 2^256 - p, words
    uint32_t m0, m1;     // floor(2^290 / p), digits
};

struct Mm8wMatrix {
    int n_out, d, nkb, n_rt;
    int4 *a8;          // [n_rt][nkb][2 digit groups][64 lanes] 16 digits each (+ one block of padding): lane (r, g) = row
                       // 16 rt + 4 (r % 4) + r / 4, term 4 kb + g, digit 16 G + 15 - j
    uint32_t *crow;    // [n_rt * 16][16]: 9 radix-2^29 digits of the per-row constant
    uint32_t *zero;    // 32 zero bytes: DMA source for inputs beyond in_count
    uint32_t bias;     // >= every |column| of every row
    WideParams *wp;    // device copy: the reduction constants are fetched by scalar loads where they are used
};

#include "hb_mm8w_body.inc"

#ifdef HB_MM8_TIMING
// debug build only (scratch/mm8w_phase_timing.py): per-wave tick sums of the phases of a pass
__device__ unsigned long long g_mm8w_t[1024 * 8];
#define MM8W_T(k) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tacc[k] += tn_ - tlast; tlast = tn_; } while (0)
#else
#define MM8W_T(k) do { } while (0)
#endif

template <bool CHECK>
__global__ __launch_bounds__(256, 1) void k_mm8w(const int4 *__restrict__ a8, const uint32_t *__restrict__ crowd,
                                                 const uint32_t *__restrict__ zero_src,
                                                 const uint32_t *__restrict__ in_pk, int64_t in_sc, int64_t in_sl,
                                                 const int32_t *__restrict__ in_rows, int64_t in_count, int d,
                                                 uint32_t *__restrict__ out_pk, int64_t out_sc, int64_t out_sl, int64_t out_count,
                                                 const int32_t *__restrict__ check_mask, int32_t *__restrict__ mismatch,
                                                 const uint32_t *__restrict__ cmp_pk, int64_t cmp_sc, int64_t cmp_sl, int mask_is_map, int n_store,
                                                 int n_out, int n_rt, int nkb, int tpw, int nbuf, int64_t n_chunks, int64_t n_units,
                                                 uint32_t bias, const WideParams *__restrict__ wpp) {
    extern __shared__ uint4 mm8w_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    uint32_t *crl = reinterpret_cast<uint32_t *>(mm8w_lds);                 // [n_rt * 16][16]
    uint4 *tlds = mm8w_lds + n_rt * 64;                                     // [10][3] uint4: T_k, 9 digits + 3 pad (32 uint4 reserved)
    uint4 *xbuf = tlds + 32;                                                // nbuf x [tpw][nkb][2][64] uint4, then 2 KB of slack
    const int bufsz = tpw * nkb * 2 * 64;
    int32_t *rowl = reinterpret_cast<int32_t *>(xbuf + (size_t)nbuf * bufsz + 128);   // [4 nkb] term -> input row
    int32_t *maskl = rowl + 4 * nkb;                                        // [16 n_rt] CHECK: 1 + row to compare with, or 0
    if (threadIdx.x < 120) {
        const int k = threadIdx.x / 12, j = threadIdx.x % 12;
        reinterpret_cast<uint32_t *>(tlds)[threadIdx.x] = j < 9 ? wpp->T[k][j] : 0u;
    }
    for (int l = threadIdx.x; l < 4 * nkb; l += 256) {
        const int lc = l < d ? l : d - 1;
        rowl[l] = in_rows ? in_rows[lc] : lc;
    }
    for (int i = threadIdx.x; i < n_rt * 16 * 16; i += 256) crl[i] = crowd[i];
    if constexpr (CHECK) {
        // a flag per row (compare with the same row), or a map: 1 + the row of the compare view, 0 = a row to store
        for (int i = threadIdx.x; i < n_rt * 16; i += 256) maskl[i] = i < n_out ? (mask_is_map ? check_mask[i] : (check_mask[i] ? i + 1 : 0)) : 0;
    }
    __syncthreads();
    const int n_slots = tpw * nkb * 2;
    // slot s = (t * nkb + kb) * 2 + h holds half h of element (chunk n, term 4 kb + g) of tile t for lane (n, g)
    auto issue_loads = [&](int64_t unit, int buf) {
        for (int s = wave; s < n_slots; s += 4) {
            const int h = s & 1, q = s >> 1, t = q / nkb, kb = q - t * nkb;
            int64_t chunk = (unit * tpw + t) * 16 + n;
            if (chunk >= n_chunks) chunk = n_chunks - 1;
            const int64_t idx = chunk * in_sc + (int64_t)rowl[4 * kb + g] * in_sl;
            const uint4 *src = (idx < in_count) ? reinterpret_cast<const uint4 *>(in_pk) + idx * 2 + h
                                                : reinterpret_cast<const uint4 *>(zero_src) + h;
            const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(xbuf + (size_t)buf * bufsz + s * 64));
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
        }
    };
    const int n_pairs = tpw * n_rt;
    int32_t k1 = 1, k256 = 256, k64k = 1 << 16, k16m = 1 << 24;  // opaque, so that the word assembly stays one v_mad_i64_i32 per column
    asm volatile("" : "+s"(k1), "+s"(k256), "+s"(k64k), "+s"(k16m));
    const int64_t bias4 = (int64_t)bias * 0x01010101ll, bias3 = (int64_t)bias * 0x00010101ll;   // the accumulator bias of 4 (3) columns
    int buf = 0;
    int64_t unit = blockIdx.x;
#ifdef HB_MM8_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    if (unit < n_units) issue_loads(unit, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    MM8W_T(0);   // prologue
    for (; unit < n_units; unit += gridDim.x) {
        const int64_t next = unit + gridDim.x;
        bool dma_issued = false;
        for (int pidx = wave; pidx < n_pairs; pidx += 4) {
            const int tl = pidx / n_rt, rt = pidx - tl * n_rt;
            const int64_t chunk = (unit * tpw + tl) * 16 + n;
            v4i acc[MM8W_NC];
            {
                uint32_t xa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)(xbuf + (size_t)buf * bufsz + (size_t)tl * nkb * 2 * 64 + lane);
                uint32_t va = (uint32_t)lane * 16u;
                uint32_t cnt = (uint32_t)(nkb / 2 - 1);
                const uint64_t abase = (uint64_t)(uintptr_t)(a8 + (size_t)rt * nkb * 2 * 64);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                mm8w_phase(acc, xa, va, cnt, abase);
                __builtin_amdgcn_sched_barrier(0);
            }
            MM8W_T(1);   // MFMA phase
            // the next unit's tiles: requested here, behind the MFMA phase (whose waits on its digit loads are vmcnt(0))
            // and ahead of the epilogue, which hides their latency
            if (nbuf == 2 && !dma_issued) { if (next < n_units) issue_loads(next, buf ^ 1); dma_issued = true; }
            MM8W_T(2);   // DMA issue
            // the reduction constants come by scalar loads once per pass: kept across the MFMA phase they would spill
            uint64_t wqa = (uint64_t)(uintptr_t)wpp;
            asm volatile("" : "+s"(wqa));
            const __attribute__((address_space(4))) WideParams *wq = (const __attribute__((address_space(4))) WideParams *)wqa;
            // Two outputs at a time, STAGE by stage: one wave per SIMD has nothing but its own independent work to cover the
            // 8-cycle dependent-issue latency of the carry / MAD chains, and an inline-asm statement per output (as in the
            // first version) is a scheduling boundary that serialised the four reductions.
#pragma unroll
            for (int rp = 0; rp < 4; rp += 2) {
                if (16 * rt + 4 * rp >= n_out) break;                // whole outputs of padding rows (wave-uniform)
                uint32_t ew[2][8];
                bool cmp[2] = {false, false};
                // S = sum_c (col_c + bias) 2^(8c): four SIGNED columns per 32-bit step go into one 64-bit accumulator that starts
                // from bias (1 + 2^8 + 2^16 + 2^24) -- four v_mad_i64_i32, no per-column bias add -- then one add-with-carry per word
                uint32_t w[2][MM8W_WORDS + 1];
                {
                    uint32_t hi_prev[2] = {0, 0};
                    unsigned cy[2] = {0, 0};
#pragma unroll
                    for (int j = 0; j < 16; j++) {
#pragma unroll
                        for (int o = 0; o < 2; o++) {
                            const int reg = rp + o;
                            int64_t a64 = (int64_t)acc[4 * j][reg] * k1 + (4 * j + 3 < MM8W_NC ? bias4 : bias3);
                            a64 += (int64_t)acc[4 * j + 1][reg] * k256;
                            a64 += (int64_t)acc[4 * j + 2][reg] * k64k;
                            if (4 * j + 3 < MM8W_NC) a64 += (int64_t)acc[4 * j + 3][reg] * k16m;
                            if (j == 0) w[o][0] = (uint32_t)a64;
                            else w[o][j] = __builtin_addc((uint32_t)a64, hi_prev[o], cy[o], &cy[o]);
                            hi_prev[o] = (uint32_t)((uint64_t)a64 >> 32);
                        }
                    }
#pragma unroll
                    for (int o = 0; o < 2; o++) { w[o][16] = hi_prev[o] + cy[o]; w[o][17] = 0; }
                }
                uint32_t sd[2][MM8W_SD];
#pragma unroll
                for (int k = 0; k < MM8W_SD; k++) {
                    const int bit = LB * k, j = bit >> 5, sft = bit & 31;
#pragma unroll
                    for (int o = 0; o < 2; o++)
                        sd[o][k] = (sft == 0 ? w[o][j] : __builtin_amdgcn_alignbit(w[o][j + 1], w[o][j], (uint32_t)sft)) & DMASK;
                }
                // V = S_lo + sum_{k >= 9} s_k T_k + row constant  <  2^261 + (9 2^29 + 2^7) p + p  <  2^290
                uint64_t col[2][10];
#pragma unroll
                for (int o = 0; o < 2; o++) {
                    const int i = 16 * rt + 4 * (rp + o) + g;
                    const uint4 *cr = reinterpret_cast<const uint4 *>(crl + (size_t)i * 16);
                    const uint4 c0v = cr[0], c1v = cr[1], c2v = cr[2];
                    col[o][0] = (uint64_t)sd[o][0] + c0v.x; col[o][1] = (uint64_t)sd[o][1] + c0v.y; col[o][2] = (uint64_t)sd[o][2] + c0v.z; col[o][3] = (uint64_t)sd[o][3] + c0v.w;
                    col[o][4] = (uint64_t)sd[o][4] + c1v.x; col[o][5] = (uint64_t)sd[o][5] + c1v.y; col[o][6] = (uint64_t)sd[o][6] + c1v.z; col[o][7] = (uint64_t)sd[o][7] + c1v.w;
                    col[o][8] = (uint64_t)sd[o][8] + c2v.x; col[o][9] = 0;
                }
#pragma unroll
                for (int k = 0; k < 10; k++) {
                    const uint4 t0 = tlds[3 * k], t1 = tlds[3 * k + 1], t2 = tlds[3 * k + 2];
#pragma unroll
                    for (int o = 0; o < 2; o++) {
                        const uint32_t sk = sd[o][9 + k];
                        col[o][0] += (uint64_t)sk * t0.x; col[o][1] += (uint64_t)sk * t0.y; col[o][2] += (uint64_t)sk * t0.z; col[o][3] += (uint64_t)sk * t0.w;
                        col[o][4] += (uint64_t)sk * t1.x; col[o][5] += (uint64_t)sk * t1.y; col[o][6] += (uint64_t)sk * t1.z; col[o][7] += (uint64_t)sk * t1.w;
                        col[o][8] += (uint64_t)sk * t2.x;
                    }
                }
                uint32_t v[2][10];
#pragma unroll
                for (int k = 0; k < 9; k++)
#pragma unroll
                    for (int o = 0; o < 2; o++) { v[o][k] = (uint32_t)col[o][k] & DMASK; col[o][k + 1] += col[o][k] >> LB; }
                // the rows to compare with: requested once the fold has released its registers, used after the Barrett step
                if constexpr (CHECK) {
#pragma unroll
                    for (int o = 0; o < 2; o++) {
                        const int erow = maskl[16 * rt + 4 * (rp + o) + g];
                        cmp[o] = (chunk < n_chunks) && erow;
                        if (cmp[o]) load_words<8>(ew[o], cmp_pk + (chunk * cmp_sc + (int64_t)(erow - 1) * cmp_sl) * 8);
                    }
                }
                (void)ew; (void)cmp;
                uint64_t dc[2][9];
#pragma unroll
                for (int o = 0; o < 2; o++) {
                    v[o][9] = (uint32_t)col[o][9];                        // < 2^29
                    // qhat = floor(floor(V / 2^232) mu / 2^58) is floor(V / p) or one less (V / 2^290 + 2^232 / p < 1)
                    const uint64_t mid = (uint64_t)v[o][9] * wq->m0 + (uint64_t)v[o][8] * wq->m1 + (((uint64_t)v[o][8] * wq->m0) >> LB);
                    const uint64_t qh = (uint64_t)v[o][9] * wq->m1 + (mid >> LB);
                    const uint32_t q0 = (uint32_t)qh & DMASK, q1 = (uint32_t)(qh >> LB);
#pragma unroll
                    for (int k = 0; k < 9; k++) {
                        dc[o][k] = v[o][k] + (uint64_t)q0 * wq->pbar[k];
                        if (k > 0) dc[o][k] += (uint64_t)q1 * wq->pbar[k - 1];
                    }
                }
                uint32_t r[2][9];
#pragma unroll
                for (int k = 0; k < 9; k++)
#pragma unroll
                    for (int o = 0; o < 2; o++) {
                        r[o][k] = (uint32_t)dc[o][k] & DMASK;
                        if (k < 8) dc[o][k + 1] += dc[o][k] >> LB;
                    }
                uint32_t ow[2][8];
#pragma unroll
                for (int o = 0; o < 2; o++) {
                    pack<9, 8>(ow[o], r[o]);
                    uint32_t u[8];
                    unsigned cy2 = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) u[k] = __builtin_addc(ow[o][k], wq->pneg[k], cy2, &cy2);
#pragma unroll
                    for (int k = 0; k < 8; k++) ow[o][k] = cy2 ? u[k] : ow[o][k];
                }
                if constexpr (CHECK) {
#pragma unroll
                    for (int o = 0; o < 2; o++)
                        if (cmp[o]) {
                            uint32_t diff = 0;
#pragma unroll
                            for (int k = 0; k < 8; k++) diff |= ew[o][k] ^ ow[o][k];
                            if (diff) atomicOr(mismatch, 1);
                        }
                    // the rows of a fused decode + validate that are results, not predictions
                    if (n_store > 0) {
#pragma unroll
                        for (int o = 0; o < 2; o++) {
                            const int i = 16 * rt + 4 * (rp + o) + g;
                            const int64_t oidx = chunk * out_sc + (int64_t)i * out_sl;
                            if (chunk < n_chunks && i < n_store && !maskl[i] && oidx < out_count) store_words<8>(out_pk + oidx * 8, ow[o]);
                        }
                    }
                } else {
                    asm volatile("" ::"v"(ow[0][0]), "v"(ow[0][1]), "v"(ow[0][2]), "v"(ow[0][3]), "v"(ow[0][4]), "v"(ow[0][5]), "v"(ow[0][6]), "v"(ow[0][7]),
                                 "v"(ow[1][0]), "v"(ow[1][1]), "v"(ow[1][2]), "v"(ow[1][3]), "v"(ow[1][4]), "v"(ow[1][5]), "v"(ow[1][6]), "v"(ow[1][7]));
#pragma unroll
                    for (int o = 0; o < 2; o++) {
                        const int i = 16 * rt + 4 * (rp + o) + g;
                        const int64_t oidx = chunk * out_sc + (int64_t)i * out_sl;
                        if (chunk < n_chunks && i < n_out && oidx < out_count) store_words<8>(out_pk + oidx * 8, ow[o]);
                    }
                }
            }
            MM8W_T(3);   // epilogue
        }
        if (nbuf == 2 && !dma_issued && next < n_units) issue_loads(next, buf ^ 1);   // a wave without a pair still owns DMA slots
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MM8W_T(4);   // vmcnt wait
        __syncthreads();       // every wave is done with this unit's buffer and has seen its share of the next one land
        MM8W_T(5);   // barrier
        if (nbuf == 1) {
            if (next < n_units) issue_loads(next, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        } else {
            buf ^= 1;
        }
        MM8W_T(6);   // single-buffer reload
    }
#ifdef HB_MM8_TIMING
    if (lane == 0 && blockIdx.x < 256) for (int k = 0; k < 8; k++) g_mm8w_t[(blockIdx.x * 4 + wave) * 8 + k] = tacc[k];
#endif
}

}  // namespace hb

#ifdef HB_MM8_TIMING
extern "C" int hb_debug_mm8w_timing(unsigned long long *out, int count) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(hb::g_mm8w_t), sizeof(unsigned long long) * (size_t)count) == hipSuccess ? 0 : 1;
}
#endif

using namespace hb;

namespace {

typedef std::vector<uint32_t> Big;      // little-endian 32-bit words
Big big_from_limbs(const uint64_t *l, int n) { Big r((size_t)2 * n); for (int i = 0; i < n; i++) { r[2 * i] = (uint32_t)l[i]; r[2 * i + 1] = (uint32_t)(l[i] >> 32); } return r; }
bool big_ge(const Big &a, const Big &b) {   // same length
    for (size_t i = a.size(); i-- > 0;) if (a[i] != b[i]) return a[i] > b[i];
    return true;
}
void big_sub(Big &a, const Big &b) {
    int64_t br = 0;
    for (size_t i = 0; i < a.size(); i++) { int64_t t = (int64_t)a[i] - (i < b.size() ? b[i] : 0) + br; a[i] = (uint32_t)t; br = t >> 32; }
}
void big_add(Big &a, const Big &b) {
    uint64_t cy = 0;
    for (size_t i = 0; i < a.size(); i++) { uint64_t t = (uint64_t)a[i] + (i < b.size() ? b[i] : 0) + cy; a[i] = (uint32_t)t; cy = t >> 32; }
}
Big big_mul(const Big &a, const Big &b) {
    Big r(a.size() + b.size(), 0);
    for (size_t i = 0; i < a.size(); i++) {
        uint64_t cy = 0;
        for (size_t j = 0; j < b.size(); j++) { uint64_t t = (uint64_t)a[i] * b[j] + r[i + j] + cy; r[i + j] = (uint32_t)t; cy = t >> 32; }
        r[i + b.size()] = (uint32_t)cy;
    }
    return r;
}
// x mod p by binary long division (one-time table work); result has p.size() words
Big big_mod(const Big &x, const Big &p) {
    Big r(p.size() + 1, 0), pp(p); pp.push_back(0);
    for (int bit = (int)x.size() * 32 - 1; bit >= 0; bit--) {
        const uint32_t in = (x[bit >> 5] >> (bit & 31)) & 1u;
        for (size_t i = r.size(); i-- > 0;) r[i] = (r[i] << 1) | (i ? r[i - 1] >> 31 : in);
        if (big_ge(r, pp)) big_sub(r, pp);
    }
    r.pop_back();
    return r;
}
Big big_pow2(int bits, size_t words) { Big r(words, 0); r[bits >> 5] = 1u << (bits & 31); return r; }
void to_digits(const Big &v, uint32_t *dg, int nd) {
    for (int k = 0; k < nd; k++) {
        const int bit = 29 * k, j = bit >> 5, sft = bit & 31;
        const uint64_t lo = j < (int)v.size() ? v[j] : 0, hi = j + 1 < (int)v.size() ? v[j + 1] : 0;
        dg[k] = (uint32_t)((lo | (hi << 32)) >> sft) & DMASK;
    }
}

size_t mm8w_lds_bytes(int n_rt, int nkb, int tpw, int nbuf) {
    return ((size_t)n_rt * 64 + (size_t)nbuf * tpw * nkb * 2 * 64 + 128 + 32) * 16 + (size_t)(4 * nkb + 16 * n_rt) * 4;
}
constexpr size_t MM8W_LDS_LIMIT = 156 * 1024;

// tiles per unit and buffers: the (tile, row tile) pairs of a unit should fill the 4 waves' rounds, units should outnumber
// the CUs when the batch allows, and the element buffer is doubled when LDS allows
bool mm8w_shape(int n_rt, int nkb, int64_t n_tiles, int n_cus, int *tpw, int *nbuf) {
    int best = 0;
    double best_eff = 0.0;
    for (int t = 1; t <= 4; t++) {
        if (mm8w_lds_bytes(n_rt, nkb, t, 1) > MM8W_LDS_LIMIT) break;
        if (t > 1 && n_tiles / t < (int64_t)n_cus) break;              // keep every CU busy first
        const int pairs = t * n_rt;
        const double eff = (double)pairs / (4.0 * ((pairs + 3) / 4));
        if (eff > best_eff + 1e-9) { best_eff = eff; best = t; }
    }
    if (!best) return false;
    *tpw = best;
    *nbuf = mm8w_lds_bytes(n_rt, nkb, best, 2) <= MM8W_LDS_LIMIT ? 2 : 1;
    return true;
}

int mm8w_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

}  // namespace

namespace hb {

void mm8w_free(Mm8wMatrix *m) {
    if (!m) return;
    if (m->a8) (void)hipFree(m->a8);
    if (m->crow) (void)hipFree(m->crow);
    if (m->zero) (void)hipFree(m->zero);
    if (m->wp) (void)hipFree(m->wp);
    delete m;
}

// Image of an n_out x n_in matrix given as canonical residues, row-major, 4 x u64 limbs each.
// HB_ERR_UNSUPPORTED when the path does not apply (narrow context, modulus outside [2^254, 2^256), inner
// dimension beyond the LDS budget, HB_NO_MFMA / HB_NO_MFMA_WIDE set).
int mm8w_from_host(hb_ctx *ctx, const uint64_t *m_host, int n_out, int n_in, Mm8wMatrix **out, hipStream_t s) {
    *out = nullptr;
    if (getenv("HB_NO_MFMA") || getenv("HB_NO_MFMA_WIDE")) return HB_ERR_UNSUPPORTED;
    if (ctx->n_limbs != 4 || n_out < 1 || n_in < 1) return HB_ERR_UNSUPPORTED;
    if (!prescale_params(ctx)) return HB_ERR_UNSUPPORTED;              // 2^254 <= p < 2^256
    const int d = n_in, nkb = 2 * ((d + 7) / 8), n_rt = (n_out + 15) / 16;
    int tpw = 0, nbuf = 0;
    if (!mm8w_shape(n_rt, nkb, 1, 1, &tpw, &nbuf)) return HB_ERR_UNSUPPORTED;
    const Big p = big_from_limbs(ctx->p_limbs, 4);
    // balanced base-256 digits (an int8 operand is signed), row sums and the column bound
    std::vector<uint8_t> a(((size_t)n_rt * nkb + 1) * 2 * 64 * 16, 0);      // one block of padding: the phase prefetches past the end
    std::vector<Big> rowsum((size_t)n_out, Big(10, 0));
    uint64_t maxdig = 0;
    for (int i = 0; i < n_out; i++) {
        uint64_t dsum = 0;
        for (int l = 0; l < d; l++) {
            const uint64_t *e = m_host + ((size_t)i * n_in + l) * 4;
            const Big ev = big_from_limbs(e, 4);
            if (big_ge(ev, p)) return fail(ctx, HB_ERR_BAD_ARG, "mm8w: matrix entry is not a canonical residue");
            big_add(rowsum[i], ev);
            const int rt = i / 16, j16 = i % 16, r = 4 * (j16 % 4) + j16 / 4, kb = l / 4, g = l % 4;
            int carry = 0;
            for (int b = 0; b < 32; b++) {
                int t = (int)((e[b >> 3] >> (8 * (b & 7))) & 0xffu) + carry;
                if (t > 127) { t -= 256; carry = 1; } else carry = 0;
                const int grp = b >> 4, j = 15 - (b & 15);
                a[((((size_t)rt * nkb + kb) * 2 + grp) * 64 + (size_t)(r + 16 * g)) * 16 + j] = (uint8_t)(int8_t)t;
                dsum += (uint64_t)(t < 0 ? -t : t);
            }
            if (carry) return fail(ctx, HB_ERR_UNSUPPORTED, "mm8w: entry needs a 33rd digit");   // not for p < 2^255 + 2^254
        }
        maxdig = std::max(maxdig, dsum);
    }
    const uint64_t bias64 = 128 * maxdig + 1;
    if (bias64 >= (1ull << 30)) return fail(ctx, HB_ERR_UNSUPPORTED, "mm8w: column bound too large");
    const uint32_t bias = (uint32_t)bias64;
    // per-row constant: (0x80..80 * sum_l M[i][l] - bias * sum_c 2^(8c)) mod p
    Big c80(8, 0x80808080u);
    Big biasall(17, 0);
    for (int c = 0; c < MM8W_NC; c++) {
        const int bit = 8 * c, j = bit >> 5, sft = bit & 31;
        Big t(17, 0);
        const uint64_t v = (uint64_t)bias << sft;
        t[j] = (uint32_t)v; t[j + 1] = (uint32_t)(v >> 32);
        big_add(biasall, t);
    }
    const Big biasmod = big_mod(biasall, p);
    std::vector<uint32_t> cr((size_t)n_rt * 16 * 16, 0);
    for (int i = 0; i < n_out; i++) {
        Big corr = big_mod(big_mul(c80, rowsum[i]), p);
        if (!big_ge(corr, biasmod)) big_add(corr, p);
        big_sub(corr, biasmod);                                 // in [0, p)
        to_digits(corr, &cr[(size_t)i * 16], 9);
    }
    Mm8wMatrix *m = new Mm8wMatrix();
    m->n_out = n_out; m->d = d; m->nkb = nkb; m->n_rt = n_rt; m->a8 = nullptr; m->crow = nullptr; m->zero = nullptr; m->bias = bias; m->wp = nullptr;
    WideParams wph;
    for (int k = 0; k < 10; k++) to_digits(big_mod(big_pow2(29 * (9 + k), 18), p), wph.T[k], 9);
    memcpy(wph.pbar, ctx->psc.pbar, sizeof wph.pbar);
    memcpy(wph.pneg, ctx->psc.pneg, sizeof wph.pneg);
    wph.m0 = ctx->psc.m0; wph.m1 = ctx->psc.m1;
    hipError_t e = hipMalloc(&m->a8, a.size());
    if (e == hipSuccess) e = hipMalloc(&m->wp, sizeof(WideParams));
    if (e == hipSuccess) e = hipMalloc(&m->crow, cr.size() * 4);
    if (e == hipSuccess) e = hipMalloc(&m->zero, 64);
    if (e != hipSuccess) { mm8w_free(m); ctx->err = std::string("mm8w tables: ") + hipGetErrorString(e); return HB_ERR_HIP; }
    {
        static_assert(sizeof(WideParams) % 4 == 0, "WideParams is a dword array");
        std::vector<uint8_t> zero64(64, 0), wpb((sizeof(WideParams) + 15) / 16 * 16, 0);
        memcpy(wpb.data(), &wph, sizeof wph);
        int rc = upload_table(ctx, m->a8, a.data(), a.size(), s);
        if (!rc) rc = upload_table(ctx, m->crow, cr.data(), cr.size() * 4, s);
        if (!rc) rc = upload_table(ctx, m->zero, zero64.data(), 64, s);
        if (!rc) { void *w16 = nullptr; if (hipMalloc(&w16, wpb.size()) != hipSuccess) rc = HB_ERR_HIP; else { (void)hipFree(m->wp); m->wp = (WideParams *)w16; rc = upload_table(ctx, m->wp, wpb.data(), wpb.size(), s); } }
        if (rc) { mm8w_free(m); return rc; }
    }
    *out = m;
    return HB_OK;
}

// out(c, i) = sum_l M[i][l] in(c, rows[l]) mod p, canonical; CHECK mode when check_mask_dev != nullptr: a flag per row and
// out = the expected values, or -- with a compare view cmp -- a map (1 + row of cmp to compare row i with; 0: row i is a
// result, stored to out when i < n_store): the fused decode + validate of hb_open.hip
int launch_mm8w(hb_ctx *ctx, const Mm8wMatrix *m, const uint32_t *in, hb_view iv, const int32_t *in_rows_dev, int64_t in_count,
                uint32_t *out, hb_view ov, int64_t out_count, const int32_t *check_mask_dev, int32_t *mismatch_dev,
                int64_t C, hipStream_t s, const uint32_t *cmp, hb_view cv, int n_store) {
    if (C <= 0) return HB_OK;
    int tpw = 1, nbuf = 1;
    const int64_t n_tiles = (C + 15) / 16;
    if (!mm8w_shape(m->n_rt, m->nkb, n_tiles, mm8w_num_cus(), &tpw, &nbuf)) return fail(ctx, HB_ERR_UNSUPPORTED, "mm8w: shape");
    const int64_t n_units = (n_tiles + tpw - 1) / tpw;
    int64_t blocks = mm8w_num_cus();
    if (blocks > n_units) blocks = n_units;
    const size_t lds = mm8w_lds_bytes(m->n_rt, m->nkb, tpw, nbuf);
    const bool check = check_mask_dev != nullptr;
#define MM8W_LAUNCH(CHK)                                                                                                              \
    do {                                                                                                                              \
        static bool attr_done = false;                                                                                                \
        if (!attr_done) {                                                                                                             \
            HB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mm8w<CHK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr_done = true;                                                                                                         \
        }                                                                                                                             \
        hipLaunchKernelGGL((k_mm8w<CHK>), dim3((unsigned)blocks), dim3(256), lds, s, m->a8, m->crow, m->zero, in, iv.stride_c, iv.stride_l, \
                           in_rows_dev, in_count, m->d, out, ov.stride_c, ov.stride_l, out_count, check_mask_dev, mismatch_dev,      \
                           cmp ? cmp : out, cmp ? cv.stride_c : ov.stride_c, cmp ? cv.stride_l : ov.stride_l, cmp ? 1 : 0, cmp ? n_store : 0, \
                           m->n_out, m->n_rt, m->nkb, tpw, nbuf, C, n_units, m->bias, m->wp);                                        \
    } while (0)
    if (check) MM8W_LAUNCH(true); else MM8W_LAUNCH(false);
#undef MM8W_LAUNCH
    HB_LAUNCH_CHECK(ctx);
    return HB_OK;
}

}  // namespace hb
