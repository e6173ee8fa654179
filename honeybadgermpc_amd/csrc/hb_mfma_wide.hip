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
// entries have 32 base-256 digits and the sum has 63 int32 columns: 156 v_mfma_i32_16x16x64_i8 per K-block of 8 terms
// and 16 x 16 outputs (8 terms x 8 digits per MFMA, as hb_mfma.hip).
//
// One wave per SIMD (512-register budget): all 63 accumulators of a 16-chunk x 16-row pass live in AGPRs a0..a251.
// A workgroup owns a unit of `tpw` tiles of 16 chunks, DMA'd into LDS in MFMA-operand order; its 4 waves take the
// (tile, row tile) pairs.  The digits stream from L2 inside the MFMA phase.
// A pass is ONE asm statement (gen_mm8w.py), software-pipelined over passes: it ends by moving its sums out of the
// AGPRs as 17 words per output, and the NEXT pass reduces them mod p between its own MFMAs -- the high eight words go
// back through the matrix cores against t_b = 2^(256 + 8b) mod p (a block-diagonal product: 16 MFMAs per output,
// round 3; round 2 folded ten radix-2^29 digits on the VALU, 90 MADs), the columns of that, the low words, the top
// word and the per-row constant are gathered per 32-bit word, a one-word Barrett quotient, conditional subtraction --
// and stores the canonical element or compares it with a received one (the fused decode + validate of hb_open.hip).
// A wave alone on its SIMD issues one instruction every ~5.5 cycles whatever its kind, so every instruction taken out
// of the serial part of a pass counts.
// Inputs are biased by XOR 0x80 (int8 operands are signed); the per-row constant takes that and the accumulator
// bias back out, mod p.  No Montgomery form anywhere.
#include <algorithm>
#include <atomic>
#include <cstddef>

#include "hb_common.hpp"

namespace hb {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int MM8W_NC = 63;     // int32 columns per output
constexpr int MM8W_WORDS = 17;  // 32-bit words of the biased sum
constexpr int MM8W_FOLD_ROW = 272;   // bytes per row of the fold table: sixteen lanes' 16 digits, then 16 zero bytes (gen_mm8w.py: FOLD_ROW)
constexpr int MM8W_FOLD_Q = 16 * MM8W_FOLD_ROW / 16;   // the table in uint4

struct WideParams {     // the asm passes fetch the first 20 dwords by scalar loads (gen_mm8w.py: s68 .. s87)
    uint32_t pneg[8];    // 2^256 - p, words
    uint32_t mu;         // floor(2^286 / p)
    uint32_t c512[8];    // 2^512 mod p, words
    uint32_t pad[3];
    // the A operands of the fold: row i = (byte half ks = i / 8, column block eb = i % 8), lane m < 16 of its diagonal block holds
    // the balanced digits s_(16 ks + pos, 4 eb + m % 4), pos < 16, of t_b = 2^(256 + 8 b) mod p (or that minus p)
    uint8_t fold[16][MM8W_FOLD_ROW];
};
static_assert(offsetof(WideParams, fold) == 80, "gen_mm8w.py loads pneg, mu, c512 from offsets 0 .. 67");
static_assert(sizeof(WideParams) == 80 + 16 * MM8W_FOLD_ROW, "the kernel copies the fold table as uint4");

struct Mm8wMatrix {
    int n_out, d, nkb, n_rt;
    int tile_rows;     // 16, or 12 / 8 when that makes fewer or cheaper passes (three / two sums per lane instead of four)
    mutable int64_t shape_tiles;                 // the launch geometry chosen for the last batch size (mm8w_shape simulates: not per launch)
    mutable int shape_tpw, shape_nbuf, shape_rq, shape_flat;   // shape_flat: ring slots of the balanced launch (k_mm8w_flat), 0 = the unit launch
    int4 *a8;          // [n_rt][nkb][4 digit groups][64 lanes] 16 bytes each (+ one block of padding): lane (r, g) = row
                       // 16 rt + 4 (r % 4) + r / 4, terms 8 kb + 2 g and + 1, eight digits of group G each
    uint32_t *crow;    // [n_rt * 16][16]: 9 radix-2^29 digits of the per-row constant (in [0, p); the kernel turns them into words + column bias)
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

// K = outputs kept per lane: 4 (row tiles of 16 rows), 3 (row tiles of 12: the fourth row of every group of the MFMA tile is padding) or 2 (8)
template <bool CHECK, int PEEL, int K>
__global__ __launch_bounds__(256, 1) void k_mm8w(const int4 *__restrict__ a8, const uint32_t *__restrict__ crowd,
                                                 const uint32_t *__restrict__ zero_src,
                                                 const uint32_t *__restrict__ in_pk, int64_t in_sc, int64_t in_sl,
                                                 const int32_t *__restrict__ in_rows, int64_t in_count, int d,
                                                 uint32_t *__restrict__ out_pk, int64_t out_sc, int64_t out_sl, int64_t out_count,
                                                 const int32_t *__restrict__ check_mask, int32_t *__restrict__ mismatch,
                                                 const uint32_t *__restrict__ cmp_pk, int64_t cmp_sc, int64_t cmp_sl, int mask_is_map, int n_store,
                                                 int n_out, int n_rt, int nkb, int tpw, int nbuf, int rq, int64_t n_chunks, int64_t n_units,
                                                 uint32_t bias, const WideParams *__restrict__ wpp, int32_t *__restrict__ first_bad, uint32_t *__restrict__ bad_map,
                                                 const FsDone done) {
    extern __shared__ uint4 mm8w_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    uint4 *tlds = mm8w_lds;                                                 // LDS offset 0: the fold table, 16 rows of 272 bytes (WideParams::fold)
    uint32_t *crl = reinterpret_cast<uint32_t *>(mm8w_lds + MM8W_FOLD_Q);   // [n_rt * 16][16]: per row eight pairs [column bias + constant word]
    uint4 *xbuf = mm8w_lds + MM8W_FOLD_Q + n_rt * 64;                       // nbuf x [tpw][nkb][2 elements][2 halves][64] uint4, then 2 KB of slack
    const int bufsz = tpw * nkb * 4 * 64;
    int64_t *rowoff = reinterpret_cast<int64_t *>(xbuf + (size_t)nbuf * bufsz + 128);   // [8 nkb] term -> element offset of its input row
    uint64_t *rowdst = reinterpret_cast<uint64_t *>(rowoff + 8 * nkb);      // [16 n_rt] where row i's elements go: address of its chunk 0 | mode (1 store, 2 compare), 0 = nowhere
    for (int l = threadIdx.x; l < 8 * nkb; l += 256) {
        const int lc = l < d ? l : d - 1;
        rowoff[l] = (int64_t)(in_rows ? in_rows[lc] : lc) * in_sl;
    }
    __syncthreads();
    // slot s = ((t * nkb + kb) * 2 + e) * 2 + h holds half h of element (chunk n, term 8 kb + 2 g + e) of tile t for lane (n, g): a
    // wave always moves the same (e, h) of every K-block -- per slot a 64-bit LDS read (the row's offset), an add, a bound and the
    // DMA; the general address arithmetic per slot was ~10 % of a short pass (one wave per SIMD pays ~5.5 cycles per instruction)
    // A unit = `tpw` tiles of 16 chunks x `rq` row tiles (unit u: tiles of u / n_quads, row tiles from rq (u % n_quads)): matrices
    // of many row tiles and few chunk tiles (86 x 86 over 381 tiles) are cut by rows too, or a workgroup's two units of 6 passes
    // each would take 4 pass-times where 2.2 are the average.
    const int n_quads = (n_rt + rq - 1) / rq;
    // unit -> (tile group, row group).  The row groups of one tile group read the same tile; block b runs on XCD b % 8 (observed,
    // a speed assumption only) and units go to blocks round-robin with a stride that is a multiple of 8, so the row groups of a
    // tile group are given unit numbers that agree mod 8: they run on ONE XCD at about the same time and the tile comes from HBM
    // once, the other row groups find it in that XCD's L2 (cfg5's shard, 11 row tiles in groups of 4: 103.6 -> 62.7 MB measured
    // without row groups; profiles/r03_pmc_cfg5-shard_row_groups.txt).  Whole blocks of 8 n_quads units are remapped, the tail keeps
    // the plain order: a bijection either way.
    const int64_t n_tg = n_units / n_quads, xg = 8 * (int64_t)n_quads, x_full = (n_tg / 8) * xg;
    auto unit_tg = [&](int64_t u) -> int64_t {
        if (n_quads == 1) return u;
        if (u < x_full) return (u / xg) * 8 + (u % 8);
        return (n_tg / 8) * 8 + (u - x_full) / n_quads;
    };
    auto unit_q = [&](int64_t u) -> int {
        if (n_quads == 1) return 0;
        if (u < x_full) return (int)((u % xg) / 8);
        return (int)((u - x_full) % n_quads);
    };
    auto issue_loads = [&](int64_t unit, int buf) {
        const int e = (wave >> 1) & 1, h = wave & 1;
        const uint4 *base = reinterpret_cast<const uint4 *>(in_pk) + h;
        const uint4 *zsrc = reinterpret_cast<const uint4 *>(zero_src) + h;
        for (int t = 0; t < tpw; t++) {
            int64_t chunk = (unit_tg(unit) * tpw + t) * 16 + n;
            if (chunk >= n_chunks) chunk = n_chunks - 1;
            const int64_t cbase = chunk * in_sc;
            int64_t ro = rowoff[2 * g + e];
            for (int kb = 0; kb < nkb; kb++) {
                const int64_t idx = cbase + ro;
                ro = rowoff[8 * (kb + 1 < nkb ? kb + 1 : kb) + 2 * g + e];      // the next slot's row offset: its LDS latency under this slot's work
                const uint4 *src = (idx < in_count) ? base + idx * 2 : zsrc;
                const int s = ((t * nkb + kb) * 2 + e) * 2 + h;
                const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(
                    (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(xbuf + (size_t)buf * bufsz + s * 64));
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
            }
        }
    };
    const int32_t k256 = 256, k64k = 1 << 16, k16m = 1 << 24;
    const int64_t bias4 = (int64_t)bias * 0x01010101ll, bias3 = (int64_t)bias * 0x00010101ll;   // the accumulator bias of 4 (3) columns
    const uint64_t wpa = (uint64_t)(uintptr_t)wpp;
    const uint64_t out_lim = out_count >= (int64_t)1 << 56 ? ~(uint64_t)0 : (uint64_t)(uintptr_t)(out_pk + out_count * 8);
    // The sums of the pass before (17 words per output) and where they go: reduced, compared and stored INSIDE the next
    // pass's MFMA phase (gen_mm8w.py).  mode: 0 nothing, 1 store to addr, 2 compare with the row at addr.
    uint32_t w[K][17];
    uint32_t crl_addr, mode[K];
    // the fold's A operand: lane (m, g') of the diagonal block g' = m / 4 reads its 16 digits, every other lane the row's 16 zero bytes
    const uint32_t atb_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)tlds + (g == (n >> 2) ? 16u * (uint32_t)n : 256u);
    uint64_t addr[K];
    crl_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)crl;
#pragma unroll
    for (int r = 0; r < K; r++) {
#pragma unroll
        for (int j = 0; j < 17; j++) w[r][j] = 0;
        mode[r] = 0; addr[r] = 0;
    }
    int buf = 0;
    int64_t unit = blockIdx.x;
    // CHECK mode with first_bad: the first chunk whose compare failed.  A pass's compares run inside the NEXT pass, and a wave's
    // passes walk the chunk tiles in increasing order: the first time its flag becomes non-zero names its smallest such chunk.
    int64_t cmp_chunk0 = 0, bad_chunk = -1;
    uint64_t any_bad = 0;
#ifdef HB_MM8_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    if (unit < n_units) issue_loads(unit, 0);
    // the other tables are filled while the first unit's tiles are in flight (the DMA only needs the row offsets)
    {
        const uint4 *fsrc = reinterpret_cast<const uint4 *>(wpp->fold);
        for (int i = threadIdx.x; i < MM8W_FOLD_Q; i += 256) tlds[i] = fsrc[i];
    }
    // per row: the constant's nine digits -> eight words, each with the bias of the fold's four columns (2^20 each) as one 64-bit addend
    for (int i = threadIdx.x; i < n_rt * 16 * 8; i += 256) {
        const int row = i >> 3, j = i & 7, bit = 32 * j, k = bit / 29, sft = bit - 29 * k;
        const uint32_t *dg = crowd + row * 16;
        uint64_t v = (uint64_t)dg[k] >> sft;
        v |= (uint64_t)dg[k + 1] << (29 - sft);                 // k + 1 <= 8
        if (k + 2 < 9 && 58 - sft < 32) v |= (uint64_t)dg[k + 2] << (58 - sft);
        const uint64_t pair = (v & 0xffffffffull) + ((0x1010ull << 32) | 0x10100000ull);
        crl[row * 16 + 2 * j] = (uint32_t)pair;
        crl[row * 16 + 2 * j + 1] = (uint32_t)(pair >> 32);
    }
    for (int i = threadIdx.x; i < n_rt * 16; i += 256) {
        uint64_t e = 0;
        const int row = (i >> 4) * (4 * K) + (i & 15);          // slot i = 16 rt + 4 r + g holds row 4 K rt + 4 r + g when r < K
        if ((i & 15) < 4 * K && row < n_out) {
            int erow = 0;
            // CHECK: a flag per row (compare with the same row), or a map: 1 + the row of the compare view, 0 = a row to store
            if constexpr (CHECK) erow = mask_is_map ? check_mask[row] : (check_mask[row] ? row + 1 : 0);
            if (erow) e = (uint64_t)(uintptr_t)(cmp_pk + (int64_t)(erow - 1) * cmp_sl * 8) | 2u;
            else if (!CHECK || row < n_store) e = (uint64_t)(uintptr_t)(out_pk + (int64_t)row * out_sl * 8) | 1u;
        }
        rowdst[i] = e;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    MM8W_T(0);   // prologue
    for (; unit < n_units; unit += gridDim.x) {
        const int64_t next = unit + gridDim.x;
        bool dma_issued = false;
        const int rt_lo = unit_q(unit) * rq, rt_cnt = n_rt - rt_lo < rq ? n_rt - rt_lo : rq;
        const int64_t tg = unit_tg(unit);
        const int n_pairs = tpw * rt_cnt;
        for (int pidx = wave; pidx < n_pairs; pidx += 4) {
            const int tl = pidx / rt_cnt, rt = rt_lo + pidx - tl * rt_cnt;
            const int64_t chunk = (tg * tpw + tl) * 16 + n;
            uint64_t flag = 0;               // this pass's compares (of the sums the pass before left): a bit per lane
            {
                uint32_t xa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)(xbuf + (size_t)buf * bufsz + (size_t)tl * nkb * 4 * 64 + lane);
                uint32_t va = (uint32_t)lane * 16u;
                uint32_t cnt = (uint32_t)((nkb - PEEL) / 2);     // two-K-block loop bodies between the peeled K-blocks
                const uint64_t abase = (uint64_t)(uintptr_t)(a8 + (size_t)rt * nkb * 4 * 64);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#define MM8W_ARGS w, xa, va, cnt, flag, abase, k256, k64k, k16m, bias4, bias3, wpa, crl_addr, atb_addr, addr, mode
#define MM8W_PASS(SFX)                                                                     \
    do {                                                                                   \
        if constexpr (K == 4) {                                                            \
            if constexpr (PEEL == 1) mm8w_pass##SFX##_p1_k4(MM8W_ARGS);                    \
            else if constexpr (PEEL == 2) mm8w_pass##SFX##_p2_k4(MM8W_ARGS);               \
            else if constexpr (PEEL == 3) mm8w_pass##SFX##_p3_k4(MM8W_ARGS);               \
            else mm8w_pass##SFX##_p4_k4(MM8W_ARGS);                                        \
        } else if constexpr (K == 3) {                                                     \
            if constexpr (PEEL == 1) mm8w_pass##SFX##_p1_k3(MM8W_ARGS);                    \
            else if constexpr (PEEL == 2) mm8w_pass##SFX##_p2_k3(MM8W_ARGS);               \
            else if constexpr (PEEL == 3) mm8w_pass##SFX##_p3_k3(MM8W_ARGS);               \
            else mm8w_pass##SFX##_p4_k3(MM8W_ARGS);                                        \
        } else {                                                                           \
            if constexpr (PEEL == 1) mm8w_pass##SFX##_p1_k2(MM8W_ARGS);                    \
            else if constexpr (PEEL == 2) mm8w_pass##SFX##_p2_k2(MM8W_ARGS);               \
            else if constexpr (PEEL == 3) mm8w_pass##SFX##_p3_k2(MM8W_ARGS);               \
            else mm8w_pass##SFX##_p4_k2(MM8W_ARGS);                                        \
        }                                                                                  \
    } while (0)
                if constexpr (CHECK) MM8W_PASS(_check); else MM8W_PASS();
#undef MM8W_PASS
#undef MM8W_ARGS
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (CHECK) {
                // (the statement above was handed a fresh flag: what it holds now belongs to the tile compared inside this pass)
                if (flag != 0) {
                    const uint32_t m16 = (uint32_t)((flag | (flag >> 16) | (flag >> 32) | (flag >> 48)) & 0xffffu);    // lane = chunk + 16 g
                    if (bad_chunk < 0) bad_chunk = cmp_chunk0 + (__builtin_ctz(m16));
                    // every disagreeing chunk, not only the first: a bit per chunk (hb_quick_interp_check_map)
                    if (bad_map && lane == 0) atomicOr(bad_map + (cmp_chunk0 >> 5), m16 << (cmp_chunk0 & 16));
                    any_bad = 1;
                }
                cmp_chunk0 = chunk - n;          // the tile whose sums this pass leaves for the next one to compare
            }
            MM8W_T(1);   // MFMA phase + the reduction of the pass before + word assembly
            // where this pass's outputs go (used by the next pass, or by the drain below); output r's row constant is 4 r rows on
            crl_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)(crl + (size_t)(16 * rt + g) * 16);
            {
                const bool in_batch = chunk < n_chunks;
                const uint64_t c_out = (uint64_t)(chunk * out_sc) * 32u, c_cmp = (uint64_t)(chunk * cmp_sc) * 32u;
#pragma unroll
                for (int r = 0; r < K; r++) {
                    const uint64_t e = rowdst[16 * rt + 4 * r + g];
                    const uint32_t m = (uint32_t)e & 3u;
                    const uint64_t a = (e & ~(uint64_t)3) + (m == 2 ? c_cmp : c_out);
                    const bool ok = in_batch && m != 0 && (m == 2 || a < out_lim);       // out_lim: the view's end (truncated results)
                    mode[r] = ok ? m : 0;
                    addr[r] = ok ? a : 0;
                }
            }
            MM8W_T(3);   // bookkeeping
            // the next unit's tiles: requested here, behind the MFMA phase (whose waits on its digit loads are vmcnt(0))
            if (nbuf == 2 && !dma_issued) { if (next < n_units) issue_loads(next, buf ^ 1); dma_issued = true; }
            MM8W_T(2);   // DMA issue
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
    // drain: the last pass's sums
    uint64_t flag = 0;
    {
        uint32_t xa = 0, va = 0, cnt = 0;
        __builtin_amdgcn_sched_barrier(0);
#define MM8W_ARGS w, xa, va, cnt, flag, 0, k256, k64k, k16m, bias4, bias3, wpa, crl_addr, atb_addr, addr, mode
        if constexpr (CHECK) { if constexpr (K == 4) mm8w_reduce_check_k4(MM8W_ARGS); else if constexpr (K == 3) mm8w_reduce_check_k3(MM8W_ARGS); else mm8w_reduce_check_k2(MM8W_ARGS); }
        else { if constexpr (K == 4) mm8w_reduce_k4(MM8W_ARGS); else if constexpr (K == 3) mm8w_reduce_k3(MM8W_ARGS); else mm8w_reduce_k2(MM8W_ARGS); }
#undef MM8W_ARGS
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (CHECK) {
        if (flag != 0) {
            const uint32_t m16 = (uint32_t)((flag | (flag >> 16) | (flag >> 32) | (flag >> 48)) & 0xffffu);
            if (bad_chunk < 0) bad_chunk = cmp_chunk0 + (__builtin_ctz(m16));
            if (bad_map && lane == 0) atomicOr(bad_map + (cmp_chunk0 >> 5), m16 << (cmp_chunk0 & 16));
        }
        if ((flag | any_bad) != 0 && lane == 0) {
            atomicOr(mismatch, 1);
            if (first_bad) atomicMin(first_bad, (int32_t)(bad_chunk > 0x7fffffff ? 0x7fffffff : bad_chunk));
        }
    }
#ifdef HB_MM8_TIMING
    if (lane == 0 && blockIdx.x < 256) for (int k = 0; k < 8; k++) g_mm8w_t[(blockIdx.x * 4 + wave) * 8 + k] = tacc[k];
#endif
    // a caller that waits for the verdict (hb_quick_dec_decide at points that are not small integers): the last workgroup to finish hands the
    // status words to pinned host memory and resets them, the sequence number last -- as k_mm8f does (a one-thread kernel behind this launch
    // did it until round 5: a dispatch and its gap on the path of every first-sight decode)
    if constexpr (CHECK) {
        fs_workgroup_done(done, mismatch, first_bad);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The balanced launch (round 6).  k_mm8w above deals UNITS (a chunk tile x a group of row tiles) to the workgroups: at config 5's
// shard the fused R2 launch is 382 tiles x 11 row tiles = 4202 passes for 1024 waves -- 4.1 a wave, so a fifth of the launch is the
// waves that take a fifth pass while nine tenths of the chip are done (R1: 2292 passes, 2.24 a wave, three rounds).  Here the passes
// are ONE list, tile-major, cut into contiguous ranges of floor / ceil (N / G) passes, one range a workgroup; a workgroup walks its
// range four passes a round (one a wave; a round may straddle two chunk tiles: the tiles live in a ring of LDS slots, the next
// round's new tile is requested behind this round's passes).  What is left after the full rounds is ONE short round:
//   3 passes  three waves take one each;
//   2 passes  two waves a pass, each half of the K-blocks;
//   1 pass    the four waves take a quarter of its K-blocks each
// -- the pieces of a pass run the ordinary pass statements on their share of the K-blocks (every statement starts its columns
// from zero), leave their sums as 17 words an output like any pass, and the followers hand theirs to the pass's leader through LDS
// (over ring slots no tile of the round lives in); the leader adds them word by word, takes the accumulator bias the extra pieces
// added back out (FlatCorr: 2^544 - (pieces - 1) bias sum_c 2^(8c)) and reduces / stores / compares the total in the drain.
// 4 + ~0.4 rounds instead of 5 (R2), 2 + ~0.4 instead of 3 (R1).  Row tiles of 16 rows, at least 4 row tiles and 8 K-blocks.
constexpr int MM8W_PART_Q = 4 * 17 * 64 / 4;        // a wave's partial sums in uint4: 4 outputs x 17 words x 64 lanes
struct FlatCorr { uint32_t w[2][17]; };              // [0]: two pieces, [1]: four

template <bool CHECK>
__global__ __launch_bounds__(256, 1) void k_mm8w_flat(const int4 *__restrict__ a8, const uint32_t *__restrict__ crowd,
                                                      const uint32_t *__restrict__ zero_src,
                                                      const uint32_t *__restrict__ in_pk, int64_t in_sc, int64_t in_sl,
                                                      const int32_t *__restrict__ in_rows, int64_t in_count, int d,
                                                      uint32_t *__restrict__ out_pk, int64_t out_sc, int64_t out_sl, int64_t out_count,
                                                      const int32_t *__restrict__ check_mask, int32_t *__restrict__ mismatch,
                                                      const uint32_t *__restrict__ cmp_pk, int64_t cmp_sc, int64_t cmp_sl, int mask_is_map, int n_store,
                                                      int n_out, int n_rt, int nkb, int nb, int64_t n_chunks, int64_t n_pass,
                                                      uint32_t bias, const WideParams *__restrict__ wpp, int32_t *__restrict__ first_bad, uint32_t *__restrict__ bad_map,
                                                      const FsDone done, const FlatCorr corr) {
    constexpr int K = 4;
    extern __shared__ uint4 mm8w_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    uint4 *tlds = mm8w_lds;                                                 // the fold table
    uint32_t *crl = reinterpret_cast<uint32_t *>(mm8w_lds + MM8W_FOLD_Q);   // [n_rt * 16][16]
    uint4 *xbuf = mm8w_lds + MM8W_FOLD_Q + n_rt * 64;                       // the ring: nb x [nkb][2 elements][2 halves][64] uint4, then 2 KB of slack
    const int bufsz = nkb * 4 * 64;
    int64_t *rowoff = reinterpret_cast<int64_t *>(xbuf + (size_t)nb * bufsz + 128);
    uint64_t *rowdst = reinterpret_cast<uint64_t *>(rowoff + 8 * nkb);
    for (int l = threadIdx.x; l < 8 * nkb; l += 256) {
        const int lc = l < d ? l : d - 1;
        rowoff[l] = (int64_t)(in_rows ? in_rows[lc] : lc) * in_sl;
    }
    __syncthreads();
    // this workgroup's range of the pass list.  Block b runs on XCD b % 8 (observed; a speed assumption only): neighbouring ranges --
    // they share the chunk tile their border cuts -- go to blocks that agree mod 8, so that tile comes from HBM once
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int rho = (G & 7) == 0 ? (b & 7) * (G >> 3) + (b >> 3) : b;
    const int64_t q_lo = n_pass / G, q_rem = n_pass - q_lo * G;
    const int64_t p_first = rho * q_lo + (rho < q_rem ? rho : q_rem);
    const int q = (int)(q_lo + (rho < q_rem ? 1 : 0));
    const int64_t t0 = p_first / n_rt;
    auto issue_tile = [&](int64_t tile, int slot) {
        const int e = (wave >> 1) & 1, h = wave & 1;
        const uint4 *base = reinterpret_cast<const uint4 *>(in_pk) + h;
        const uint4 *zsrc = reinterpret_cast<const uint4 *>(zero_src) + h;
        int64_t chunk = tile * 16 + n;
        if (chunk >= n_chunks) chunk = n_chunks - 1;
        const int64_t cbase = chunk * in_sc;
        int64_t ro = rowoff[2 * g + e];
        for (int kb = 0; kb < nkb; kb++) {
            const int64_t idx = cbase + ro;
            ro = rowoff[8 * (kb + 1 < nkb ? kb + 1 : kb) + 2 * g + e];
            const uint4 *src = (idx < in_count) ? base + idx * 2 : zsrc;
            const int s = (kb * 2 + e) * 2 + h;
            const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(
                (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(xbuf + (size_t)slot * bufsz + s * 64));
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
        }
    };
    int64_t issued_hi = t0 - 1;
    auto tiles_upto = [&](int64_t t_hi) {
        while (issued_hi < t_hi) { issued_hi++; issue_tile(issued_hi, (int)((issued_hi - t0) % nb)); }
    };
    // rounds: `full` of four passes, then at most one short one
    const int full = q >> 2, left = q & 3, n_rounds = full + (left ? 1 : 0);
    auto round_last_tile = [&](int r) -> int64_t { return (p_first + 4 * (int64_t)r + (r < full ? 4 : left) - 1) / n_rt; };
    const int32_t k256 = 256, k64k = 1 << 16, k16m = 1 << 24;
    const int64_t bias4 = (int64_t)bias * 0x01010101ll, bias3 = (int64_t)bias * 0x00010101ll;
    const uint64_t wpa = (uint64_t)(uintptr_t)wpp;
    const uint64_t out_lim = out_count >= (int64_t)1 << 56 ? ~(uint64_t)0 : (uint64_t)(uintptr_t)(out_pk + out_count * 8);
    uint32_t w[K][17];
    uint32_t crl_addr, mode[K];
    const uint32_t atb_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)tlds + (g == (n >> 2) ? 16u * (uint32_t)n : 256u);
    uint64_t addr[K];
    crl_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)crl;
#pragma unroll
    for (int r = 0; r < K; r++) {
#pragma unroll
        for (int j = 0; j < 17; j++) w[r][j] = 0;
        mode[r] = 0; addr[r] = 0;
    }
    int64_t cmp_chunk0 = 0, bad_chunk = -1;
    uint64_t any_bad = 0;
    if (n_rounds > 0) tiles_upto(round_last_tile(0));
    {
        const uint4 *fsrc = reinterpret_cast<const uint4 *>(wpp->fold);
        for (int i = threadIdx.x; i < MM8W_FOLD_Q; i += 256) tlds[i] = fsrc[i];
    }
    for (int i = threadIdx.x; i < n_rt * 16 * 8; i += 256) {
        const int row = i >> 3, j = i & 7, bit = 32 * j, k = bit / 29, sft = bit - 29 * k;
        const uint32_t *dg = crowd + row * 16;
        uint64_t v = (uint64_t)dg[k] >> sft;
        v |= (uint64_t)dg[k + 1] << (29 - sft);
        if (k + 2 < 9 && 58 - sft < 32) v |= (uint64_t)dg[k + 2] << (58 - sft);
        const uint64_t pair = (v & 0xffffffffull) + ((0x1010ull << 32) | 0x10100000ull);
        crl[row * 16 + 2 * j] = (uint32_t)pair;
        crl[row * 16 + 2 * j + 1] = (uint32_t)(pair >> 32);
    }
    for (int i = threadIdx.x; i < n_rt * 16; i += 256) {
        uint64_t e = 0;
        const int row = (i >> 4) * 16 + (i & 15);
        if (row < n_out) {
            int erow = 0;
            if constexpr (CHECK) erow = mask_is_map ? check_mask[row] : (check_mask[row] ? row + 1 : 0);
            if (erow) e = (uint64_t)(uintptr_t)(cmp_pk + (int64_t)(erow - 1) * cmp_sl * 8) | 2u;
            else if (!CHECK || row < n_store) e = (uint64_t)(uintptr_t)(out_pk + (int64_t)row * out_sl * 8) | 1u;
        }
        rowdst[i] = e;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // where follower k (0, 1, 2) of a short round leaves its sums: the k-th MM8W_PART_Q uint4 of the ring outside the slots busy_a, busy_b
    auto part_off = [&](int k, int busy_a, int busy_b) -> int {
        int off = 0;
        for (int i = 0;; i++) {
            for (int rep = 0; rep < 2; rep++) {
                if (busy_a >= 0 && off < (busy_a + 1) * bufsz && off + MM8W_PART_Q > busy_a * bufsz) off = (busy_a + 1) * bufsz;
                if (busy_b >= 0 && off < (busy_b + 1) * bufsz && off + MM8W_PART_Q > busy_b * bufsz) off = (busy_b + 1) * bufsz;
            }
            if (i == k) return off;
            off += MM8W_PART_Q;
        }
    };
    for (int r = 0; r < n_rounds; r++) {
        const int64_t pf = p_first + 4 * (int64_t)r;
        const int cnt_r = r < full ? 4 : left;
        const int pieces = (r < full || cnt_r == 3) ? 1 : (cnt_r == 1 ? 4 : 2);     // waves a pass of this round
        // this wave's share: pass pf + po, K-blocks [kb0, kb0 + len)
        int po = wave, kb0 = 0, len = nkb, part = 0;
        if (pieces == 4) {
            const int bl = nkb >> 2, ex = nkb & 3;
            po = 0; part = wave; len = bl + (wave < ex ? 1 : 0); kb0 = wave * bl + (wave < ex ? wave : ex);
        } else if (pieces == 2) {
            const int l0 = (nkb + 1) >> 1;
            po = wave >> 1; part = wave & 1; len = part ? nkb - l0 : l0; kb0 = part ? l0 : 0;
        }
        const bool have = pieces > 1 || wave < cnt_r;
        if (have) {
            const int64_t pass = pf + po, tile = pass / n_rt;
            const int rt = (int)(pass - tile * n_rt), slot = (int)((tile - t0) % nb);
            const int64_t chunk = tile * 16 + n;
            uint64_t flag = 0;
            {
                uint32_t xa = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)(xbuf + (size_t)slot * bufsz + (size_t)kb0 * 256 + lane);
                uint32_t va = (uint32_t)lane * 16u;
                const uint64_t abase = (uint64_t)(uintptr_t)(a8 + ((size_t)rt * nkb + kb0) * 4 * 64);
                // the statement for `len` K-blocks: 2 written out; or 3 / 4 written out and a loop of two-block bodies between them
                const int var = len == 2 ? 2 : ((len & 1) ? 3 : 4);
                uint32_t cnt = (uint32_t)((len - var) >> 1);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#define MM8W_ARGS w, xa, va, cnt, flag, abase, k256, k64k, k16m, bias4, bias3, wpa, crl_addr, atb_addr, addr, mode, var
                if constexpr (CHECK) mm8w_pass_check_multi_k4(MM8W_ARGS); else mm8w_pass_multi_k4(MM8W_ARGS);
#undef MM8W_ARGS
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (CHECK) {
                if (flag != 0) {
                    const uint32_t m16 = (uint32_t)((flag | (flag >> 16) | (flag >> 32) | (flag >> 48)) & 0xffffu);
                    if (bad_chunk < 0) bad_chunk = cmp_chunk0 + (__builtin_ctz(m16));
                    if (bad_map && lane == 0) atomicOr(bad_map + (cmp_chunk0 >> 5), m16 << (cmp_chunk0 & 16));
                    any_bad = 1;
                }
                cmp_chunk0 = chunk - n;
            }
            if (part == 0) {
                // where this pass's outputs go (a leader's: once the followers' sums have joined them)
                crl_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)(crl + (size_t)(16 * rt + g) * 16);
                const bool in_batch = chunk < n_chunks;
                const uint64_t c_out = (uint64_t)(chunk * out_sc) * 32u, c_cmp = (uint64_t)(chunk * cmp_sc) * 32u;
#pragma unroll
                for (int o = 0; o < K; o++) {
                    const uint64_t e = rowdst[16 * rt + 4 * o + g];
                    const uint32_t m = (uint32_t)e & 3u;
                    const uint64_t a = (e & ~(uint64_t)3) + (m == 2 ? c_cmp : c_out);
                    const bool ok = in_batch && m != 0 && (m == 2 || a < out_lim);
                    mode[o] = ok ? m : 0;
                    addr[o] = ok ? a : 0;
                }
            }
        }
        if (r + 1 < n_rounds) tiles_upto(round_last_tile(r + 1));
        if (pieces > 1) {
            // (the last round: no tile is in flight, and only the slots of this round's one or two tiles are read by anybody)
            const int sa = (int)((pf / n_rt - t0) % nb), sb = (int)(((pf + cnt_r - 1) / n_rt - t0) % nb);
            const int fol = pieces == 4 ? wave - 1 : (wave >> 1);        // follower number of this wave (when part != 0)
            if (part != 0) {
                uint32_t *dst = reinterpret_cast<uint32_t *>(xbuf + part_off(fol, sa, sb)) + lane;
#pragma unroll
                for (int o = 0; o < K; o++) {
#pragma unroll
                    for (int j = 0; j < 17; j++) dst[(o * 17 + j) * 64] = w[o][j];
                    mode[o] = 0; addr[o] = 0;
                }
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (part == 0) {
                const int f_lo = pieces == 4 ? 0 : (wave >> 1), f_n = pieces == 4 ? 3 : 1;
                for (int f = f_lo; f < f_lo + f_n; f++) {
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(xbuf + part_off(f, sa, sb)) + lane;
#pragma unroll
                    for (int o = 0; o < K; o++) {
                        unsigned cy = 0;
#pragma unroll
                        for (int j = 0; j < 17; j++) w[o][j] = __builtin_addc(w[o][j], src[(o * 17 + j) * 64], cy, &cy);
                    }
                }
                const uint32_t *cw = corr.w[pieces == 4 ? 1 : 0];
#pragma unroll
                for (int o = 0; o < K; o++) {
                    unsigned cy = 0;
#pragma unroll
                    for (int j = 0; j < 17; j++) w[o][j] = __builtin_addc(w[o][j], cw[j], cy, &cy);
                }
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    // drain: the last pass's sums
    uint64_t flag = 0;
    {
        uint32_t xa = 0, va = 0, cnt = 0;
        __builtin_amdgcn_sched_barrier(0);
#define MM8W_ARGS w, xa, va, cnt, flag, 0, k256, k64k, k16m, bias4, bias3, wpa, crl_addr, atb_addr, addr, mode
        if constexpr (CHECK) mm8w_reduce_check_k4(MM8W_ARGS); else mm8w_reduce_k4(MM8W_ARGS);
#undef MM8W_ARGS
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (CHECK) {
        if (flag != 0) {
            const uint32_t m16 = (uint32_t)((flag | (flag >> 16) | (flag >> 32) | (flag >> 48)) & 0xffffu);
            if (bad_chunk < 0) bad_chunk = cmp_chunk0 + (__builtin_ctz(m16));
            if (bad_map && lane == 0) atomicOr(bad_map + (cmp_chunk0 >> 5), m16 << (cmp_chunk0 & 16));
        }
        if ((flag | any_bad) != 0 && lane == 0) {
            atomicOr(mismatch, 1);
            if (first_bad) atomicMin(first_bad, (int32_t)(bad_chunk > 0x7fffffff ? 0x7fffffff : bad_chunk));
        }
        fs_workgroup_done(done, mismatch, first_bad);
    }
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

// x <- 2 x mod p (x < p, both n words)
void big_dbl_mod(Big &x, const Big &p) {
    uint32_t top = 0;
    for (size_t i = 0; i < x.size(); i++) { const uint32_t nt = x[i] >> 31; x[i] = (x[i] << 1) | top; top = nt; }
    if (top || big_ge(x, p)) big_sub(x, p);       // 2x < 2p < 2^(32 n + 1): one subtraction, mod 2^(32 n) when the top bit left
}

// The constants of the reduction (gen_mm8w.py reduce_output; hb_mfma.hip's epilogue with n_bytes = 16): the fold table of
// t_b = 2^(256 + 8 b) mod p for the n_bytes bytes of H, 2^(256 + 8 n_bytes) mod p for the word above H, mu = floor(2^286 / p) --
// and what the fold adds to every sum, which the per-row constants take back out:
// shift = (128 sum_b t_b - sum_e 2^20 2^(8 e)) mod p (the operand bias of XOR 0x80 and the bias of the fold's 32 columns).
bool fold_tables_impl(hb_ctx *ctx, int n_bytes, uint8_t *fold, uint32_t *top8, uint32_t *mu_out, Big *shift) {
    const Big p = big_from_limbs(ctx->p_limbs, 4);
    memset(fold, 0, (size_t)(n_bytes / 16) * 8 * MM8W_FOLD_ROW);
    Big T = big_mod(big_pow2(256, 9), p);          // t_0; then t_(b+1) = 256 t_b mod p
    Big tsum(10, 0);
    for (int b = 0; b < n_bytes; b++) {
        big_add(tsum, T);
        // 32 balanced digits of T, or of T - p (two's complement over ten words) when T needs a 33rd
        int8_t dg[32];
        bool ok = false;
        for (int rep_i = 0; rep_i < 2 && !ok; rep_i++) {
            Big v(T); v.resize(10, 0);
            if (rep_i) big_sub(v, p);
            int carry = 0;
            for (int k = 0; k < 32; k++) {
                int t = (int)((v[k >> 2] >> (8 * (k & 3))) & 0xffu) + carry;
                if (t > 127) { t -= 256; carry = 1; } else carry = 0;
                dg[k] = (int8_t)t;
            }
            const uint64_t rest = (((uint64_t)v[9] << 32) | v[8]) + (uint64_t)carry;
            ok = rest == 0;
        }
        if (!ok) return false;                      // cannot happen for p < 2^256: one of the two lies in [-0.502, 0.498] 2^256
        const int ks = b / 16, pos = b % 16;
        for (int e = 0; e < 32; e++) {
            const int eb = e / 4, r = e % 4;
            for (int G = 0; G < 4; G++) fold[(size_t)(ks * 8 + eb) * MM8W_FOLD_ROW + 16 * (4 * G + r) + pos] = (uint8_t)dg[e];   // lane m = 4 G + r of block G
        }
        for (int k = 0; k < 8; k++) big_dbl_mod(T, p);
    }
    for (int j = 0; j < 8; j++) top8[j] = T[j];    // t_(n_bytes) = 2^(256 + 8 n_bytes) mod p
    // mu = floor(2^286 / p) by binary long division
    {
        Big r(9, 0), pp(p); pp.push_back(0);
        uint64_t q = 0;
        for (int bit = 286; bit >= 0; bit--) {
            for (size_t i = r.size(); i-- > 0;) r[i] = (r[i] << 1) | (i ? r[i - 1] >> 31 : (bit == 286 ? 1u : 0u));
            q <<= 1;
            if (big_ge(r, pp)) { big_sub(r, pp); q |= 1; }
        }
        if (q >> 32) return false;                  // p > 2^254
        *mu_out = (uint32_t)q;
    }
    // shift = (128 tsum - btot) mod p
    Big a = big_mod(big_mul(tsum, Big(1, 128u)), p);
    Big btot(9, 0);
    for (int e = 0; e < 32; e++) {
        const int bit = 8 * e + 20, j = bit >> 5, sft = bit & 31;
        Big t(9, 0);
        const uint64_t v = 1ull << sft;
        t[j] = (uint32_t)v; if (j + 1 < 9) t[j + 1] = (uint32_t)(v >> 32);
        big_add(btot, t);
    }
    const Big bm = big_mod(btot, p);
    if (!big_ge(a, bm)) big_add(a, p);
    big_sub(a, bm);
    *shift = a;                                      // in [0, p), 8 words
    return true;
}
struct FoldConsts { WideParams wp; Big shift; };
bool fold_consts(hb_ctx *ctx, FoldConsts *fc) {
    memset(&fc->wp, 0, sizeof fc->wp);
    memcpy(fc->wp.pneg, ctx->psc.pneg, sizeof fc->wp.pneg);
    return fold_tables_impl(ctx, 32, &fc->wp.fold[0][0], fc->wp.c512, &fc->wp.mu, &fc->shift);
}
// (the bias of the main columns, summed over the columns) - fold shift, mod p: what the per-row constants subtract
Big fold_biasmod(const Big &biasall, const Big &p, const Big &shift) {
    Big b = big_mod(biasall, p);
    if (!big_ge(b, shift)) big_add(b, p);
    big_sub(b, shift);
    return b;
}

size_t mm8w_lds_bytes(int n_rt, int nkb, int tpw, int nbuf) {
    return ((size_t)n_rt * 64 + (size_t)nbuf * tpw * nkb * 4 * 64 + 128 + MM8W_FOLD_Q) * 16 + (size_t)(16 * nkb + 32 * n_rt) * 4;
}
constexpr size_t MM8W_LDS_LIMIT = 156 * 1024;

// Tiles per unit, row tiles per unit and buffers.  A workgroup's four waves take the (tile, row tile) passes of a unit in rounds,
// units go round-robin over the workgroups: the candidates are simulated (pass-times of the slowest workgroup) and the fastest
// wins; ties go to the larger row group (a tile is loaded once per row group), then to double buffering.
bool mm8w_shape(int n_rt, int nkb, int64_t n_tiles, int n_cus, int *tpw, int *nbuf, int *rq, double *cost_out = nullptr) {
    double best_cost = 0.0;
    int best_t = 0, best_rq = 0, best_nbuf = 0;
    int rqs[3] = {n_rt, 8, 4};
    // HB_MM8W_RQ=all: no row groups (every unit takes all row tiles of its chunk tiles) -- the traffic experiment of
    // profiles/r03_pmc_cfg5-shard_row_groups.txt; never set in production
    static const bool rq_all = [] { const char *e = env_hook(ENV_MM8W_RQ); return e && !strcmp(e, "all"); }();
    if (rq_all) rqs[1] = rqs[2] = n_rt;
    for (int t = 1; t <= 4; t++) {
        if (mm8w_lds_bytes(n_rt, nkb, t, 1) > MM8W_LDS_LIMIT) break;
        const int nb = mm8w_lds_bytes(n_rt, nkb, t, 2) <= MM8W_LDS_LIMIT ? 2 : 1;
        for (int ri = 0; ri < 3; ri++) {
            const int r = rqs[ri];
            if (r > n_rt || (ri > 0 && r >= n_rt)) continue;
            const int nq = (n_rt + r - 1) / r;
            const int64_t units = ((n_tiles + t - 1) / t) * nq;
            const int64_t G = units < n_cus ? units : n_cus;
            // rounds of unit u = ceil(t * rows(u % nq) / 4); workgroup b takes u = b, b + G, ...
            int64_t worst = 0;
            if (units > 64 * G) {
                // many units per workgroup: every workgroup sees every kind of unit equally often
                int64_t per_period = 0;
                for (int q = 0; q < nq; q++) { const int cnt = n_rt - q * r < r ? n_rt - q * r : r; per_period += (t * cnt + 3) / 4; }
                worst = (per_period * ((units + G - 1) / G) + nq - 1) / nq;
            } else
            for (int64_t b = 0; b < G; b++) {
                int64_t sum = 0;
                for (int64_t u = b; u < units; u += G) {
                    const int q = (int)(u % nq), cnt = n_rt - q * r < r ? n_rt - q * r : r;
                    sum += (t * cnt + 3) / 4;
                }
                if (sum > worst) worst = sum;
            }
            // a pass-time each, plus what a unit costs beyond its passes (barrier, DMA wait: ~6 % of a pass, twice that unbuffered)
            const double cost = (double)worst + 0.06 * (nb == 2 ? 1 : 2) * (double)((units + G - 1) / G);
            if (!best_t || cost < best_cost - 1e-9) { best_cost = cost; best_t = t; best_rq = r; best_nbuf = nb; }
        }
    }
    if (!best_t) return false;
    *tpw = best_t; *rq = best_rq; *nbuf = best_nbuf;
    if (cost_out) *cost_out = best_cost;
    return true;
}

// uint4 offset of follower k's sums in the ring of a short round whose tiles live in slots busy_a, busy_b (k_mm8w_flat: part_off)
int mm8w_flat_part_off(int k, int busy_a, int busy_b, int bufsz) {
    int off = 0;
    for (int i = 0;; i++) {
        for (int rep = 0; rep < 2; rep++) {
            if (busy_a >= 0 && off < (busy_a + 1) * bufsz && off + 1088 > busy_a * bufsz) off = (busy_a + 1) * bufsz;
            if (busy_b >= 0 && off < (busy_b + 1) * bufsz && off + 1088 > busy_b * bufsz) off = (busy_b + 1) * bufsz;
        }
        if (i == k) return off;
        off += 1088;
    }
}

// The balanced launch (k_mm8w_flat): ring slots, or 0 when the shape does not qualify or promises nothing over `unit_cost`, the
// pass-times of the slowest workgroup of the unit launch (mm8w_shape).  A range of q passes is q / 4 full rounds and a short one:
// three passes -> a pass-time; two -> half the K-blocks a wave, their sums joined through LDS; one -> a quarter.
// HB_MM8W_FLAT=0 / 1: never / whenever the shape qualifies (A-B runs: profiles/r06_mm8w_balanced_launch.txt).
int mm8w_flat_slots(int n_rt, int nkb, int tile_rows, int64_t n_tiles, int n_cus, double unit_cost) {
    static const int force = [] { const char *e = env_hook(ENV_MM8W_FLAT); return e ? atoi(e) : -1; }();
    if (force == 0 || tile_rows != 16 || n_rt < 4 || nkb < 8) return 0;
    const int64_t n_pass = n_tiles * n_rt;
    if (n_pass < 4 * (int64_t)n_cus) return 0;
    const int bufsz = nkb * 4 * 64;
    int nb = 0;
    for (int cand = 3; cand <= 8 && !nb; cand++) {
        if (mm8w_lds_bytes(n_rt, nkb, 1, cand) > MM8W_LDS_LIMIT) break;
        bool ok = true;
        for (int a = 0; a < cand && ok; a++) {
            // one tile (three followers) or two neighbouring tiles (two followers) busy
            if (mm8w_flat_part_off(2, a, -1, bufsz) + 1088 > cand * bufsz + 128) ok = false;
            if (mm8w_flat_part_off(1, a, (a + 1) % cand, bufsz) + 1088 > cand * bufsz + 128) ok = false;
        }
        if (ok) nb = cand;
    }
    if (!nb) return 0;
    if (force == 1) return nb;
    auto rounds = [&](int64_t q) {
        const int left = (int)(q & 3);
        const double tail = left == 0 ? 0.0 : left == 3 ? 1.0 : left == 2 ? (double)((nkb + 1) / 2) / nkb + 0.12 : (double)((nkb + 3) / 4) / nkb + 0.15;
        return (double)(q >> 2) + tail + 0.06 * (double)((q >> 2) + (left ? 1 : 0));
    };
    const int64_t q_lo = n_pass / n_cus;
    const double cost = (n_pass % n_cus) ? std::max(rounds(q_lo), rounds(q_lo + 1)) : rounds(q_lo);
    return cost < 0.97 * unit_cost ? nb : 0;
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

// for hb_mfma.hip (its sums have 16 bytes of H): the table's first byte half, 2^384 mod p, mu and the shift of the row constants
bool fold_tables(hb_ctx *ctx, int n_bytes, uint8_t *fold, uint32_t *top8, uint32_t *mu, uint32_t *shift8) {
    Big sh;
    if (!fold_tables_impl(ctx, n_bytes, fold, top8, mu, &sh)) return false;
    for (int j = 0; j < 8; j++) shift8[j] = sh[j];
    return true;
}

void mm8w_free(Mm8wMatrix *m) {
    if (!m) return;
    if (m->a8) (void)hipFree(m->a8);
    if (m->crow) (void)hipFree(m->crow);
    if (m->zero) (void)hipFree(m->zero);
    if (m->wp) (void)hipFree(m->wp);
    delete m;
}

int mm8w_tile_rows(int n_out, int nkb);

// Image of an n_out x n_in matrix given as canonical residues, row-major, 4 x u64 limbs each.
// HB_ERR_UNSUPPORTED when the path does not apply (narrow context, modulus outside [2^254, 2^256), inner
// dimension beyond the LDS budget, HB_NO_MFMA / HB_NO_MFMA_WIDE set).
int mm8w_from_host(hb_ctx *ctx, const uint64_t *m_host, int n_out, int n_in, Mm8wMatrix **out, hipStream_t s) {
    *out = nullptr;
    if (env_hook(ENV_NO_MFMA) || env_hook(ENV_NO_MFMA_WIDE)) return HB_ERR_UNSUPPORTED;
    if (ctx->n_limbs != 4 || n_out < 1 || n_in < 1) return HB_ERR_UNSUPPORTED;
    if (!prescale_params(ctx)) return HB_ERR_UNSUPPORTED;              // 2^254 <= p < 2^256
    const int d = n_in, nkb = (d + 7) / 8;                                 // K-blocks of 8 terms
    // Row tiles of 16 rows, or of 12 / 8 (the last rows of every group of the MFMA tile left empty): a pass costs its MFMA phase plus
    // ~490 instructions per output a lane keeps (reduction + word assembly), all at the same ~5.5 cycles each -- 22 rows are two
    // passes either way, and two passes of three outputs beat two of four.
    const int tile_rows = mm8w_tile_rows(n_out, nkb);
    const int n_rt = (n_out + tile_rows - 1) / tile_rows;
    int tpw = 0, nbuf = 0, rq = 0;
    if (!mm8w_shape(n_rt, nkb, 1, 1, &tpw, &nbuf, &rq)) return HB_ERR_UNSUPPORTED;
    const Big p = big_from_limbs(ctx->p_limbs, 4);
    // balanced base-256 digits (an int8 operand is signed), row sums and the column bound
    std::vector<uint8_t> a(((size_t)n_rt * nkb + 1) * 4 * 64 * 16, 0);      // one block of padding
    std::vector<Big> rowsum((size_t)n_out, Big(10, 0));
    uint64_t maxdig = 0;
    for (int i = 0; i < n_out; i++) {
        uint64_t dsum = 0;
        for (int l = 0; l < d; l++) {
            const uint64_t *e = m_host + ((size_t)i * n_in + l) * 4;
            const Big ev = big_from_limbs(e, 4);
            if (big_ge(ev, p)) return fail(ctx, HB_ERR_BAD_ARG, "mm8w: matrix entry is not a canonical residue");
            big_add(rowsum[i], ev);
            // lane (r, g) of K-block kb = l / 8 holds terms 8 kb + 2 g and 8 kb + 2 g + 1: g = (l % 8) / 2, element el = l % 2
            const int rt = i / tile_rows, j16 = i % tile_rows, r = 4 * (j16 % 4) + j16 / 4, kb = l / 8, g = (l % 8) / 2, el = l & 1;
            int carry = 0;
            for (int b = 0; b < 32; b++) {
                int t = (int)((e[b >> 3] >> (8 * (b & 7))) & 0xffu) + carry;
                if (t > 127) { t -= 256; carry = 1; } else carry = 0;
                // digit b = 8 G + 7 - 4 hi - bi sits at byte 4 (2 hi + el) + bi of the lane's 16 bytes of group G (gen_mm8w.py: the
                // operand of window q is [q of e0, q of e1, q + 1 of e0, q + 1 of e1])
                const int grp = b >> 3, r7 = 7 - (b & 7), hi = r7 >> 2, bi = r7 & 3;
                a[((((size_t)rt * nkb + kb) * 4 + grp) * 64 + (size_t)(r + 16 * g)) * 16 + 4 * (2 * hi + el) + bi] = (uint8_t)(int8_t)t;
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
    FoldConsts fc;
    if (!fold_consts(ctx, &fc)) return fail(ctx, HB_ERR_UNSUPPORTED, "mm8w: fold table");
    const Big biasmod = fold_biasmod(biasall, p, fc.shift);
    std::vector<uint32_t> cr((size_t)n_rt * 16 * 16, 0);
    for (int i = 0; i < n_out; i++) {
        Big corr = big_mod(big_mul(c80, rowsum[i]), p);
        if (!big_ge(corr, biasmod)) big_add(corr, p);
        big_sub(corr, biasmod);                                 // in [0, p)
        to_digits(corr, &cr[((size_t)(i / tile_rows) * 16 + i % tile_rows) * 16], 9);      // slot 16 rt + 4 r + g
    }
    Mm8wMatrix *m = new Mm8wMatrix();
    m->n_out = n_out; m->d = d; m->nkb = nkb; m->n_rt = n_rt; m->tile_rows = tile_rows; m->shape_tiles = -1; m->a8 = nullptr; m->crow = nullptr; m->zero = nullptr; m->bias = bias; m->wp = nullptr;
    const WideParams &wph = fc.wp;
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
static int launch_mm8w_impl(hb_ctx *ctx, const Mm8wMatrix *m, const uint32_t *in, hb_view iv, const int32_t *in_rows_dev, int64_t in_count,
                            uint32_t *out, hb_view ov, int64_t out_count, const int32_t *check_mask_dev, int32_t *mismatch_dev,
                            int64_t C, hipStream_t s, const uint32_t *cmp, hb_view cv, int n_store, int32_t *first_bad_dev, uint32_t *bad_map_dev,
                            const FsDone *done_p = nullptr);

int launch_mm8w(hb_ctx *ctx, const Mm8wMatrix *m, const uint32_t *in, hb_view iv, const int32_t *in_rows_dev, int64_t in_count,
                uint32_t *out, hb_view ov, int64_t out_count, const int32_t *check_mask_dev, int32_t *mismatch_dev,
                int64_t C, hipStream_t s, const uint32_t *cmp, hb_view cv, int n_store) {
    return launch_mm8w_impl(ctx, m, in, iv, in_rows_dev, in_count, out, ov, out_count, check_mask_dev, mismatch_dev, C, s, cmp, cv, n_store, nullptr, nullptr);
}

static int launch_mm8w_impl(hb_ctx *ctx, const Mm8wMatrix *m, const uint32_t *in, hb_view iv, const int32_t *in_rows_dev, int64_t in_count,
                            uint32_t *out, hb_view ov, int64_t out_count, const int32_t *check_mask_dev, int32_t *mismatch_dev,
                            int64_t C, hipStream_t s, const uint32_t *cmp, hb_view cv, int n_store, int32_t *first_bad_dev, uint32_t *bad_map_dev,
                            const FsDone *done_p) {
    if (C <= 0) return HB_OK;
    FsDone done;
    memset(&done, 0, sizeof done);
    if (done_p) done = *done_p;
    int tpw = 1, nbuf = 1, rq = m->n_rt, flat_nb = 0;
    const int64_t n_tiles = (C + 15) / 16;
    if (m->shape_tiles == n_tiles) { tpw = m->shape_tpw; nbuf = m->shape_nbuf; rq = m->shape_rq; flat_nb = m->shape_flat; }
    else {
        // the simulation costs tens of microseconds: remembered per matrix, and per context for images that live for one launch
        const std::string sk = std::to_string(m->n_rt) + ":" + std::to_string(m->nkb) + ":" + std::to_string(m->tile_rows) + ":" + std::to_string((long long)n_tiles);
        auto hit = ctx->wide_shapes.find(sk);
        if (hit != ctx->wide_shapes.end()) { tpw = hit->second[0]; nbuf = hit->second[1]; rq = hit->second[2]; flat_nb = hit->second[3]; }
        else {
            double unit_cost = 0.0;
            if (!mm8w_shape(m->n_rt, m->nkb, n_tiles, mm8w_num_cus(), &tpw, &nbuf, &rq, &unit_cost)) return fail(ctx, HB_ERR_UNSUPPORTED, "mm8w: shape");
            flat_nb = mm8w_flat_slots(m->n_rt, m->nkb, m->tile_rows, n_tiles, mm8w_num_cus(), unit_cost);
            if (ctx->wide_shapes.size() > 4096) ctx->wide_shapes.clear();
            ctx->wide_shapes[sk] = std::vector<int>{tpw, nbuf, rq, flat_nb};
        }
        m->shape_tiles = n_tiles; m->shape_tpw = tpw; m->shape_nbuf = nbuf; m->shape_rq = rq; m->shape_flat = flat_nb;
    }
    if (flat_nb) {
        // the balanced launch: one range of the pass list a workgroup (k_mm8w_flat)
        const int64_t n_pass = n_tiles * m->n_rt;
        FlatCorr corr;
        {   // 2^544 - (pieces - 1) bias sum_c 2^(8c), pieces = 2 and 4: what a pass's extra pieces added to its sum
            Big biasall(17, 0);
            for (int c = 0; c < MM8W_NC; c++) {
                const int bit = 8 * c, j = bit >> 5, sft = bit & 31;
                Big t(17, 0);
                const uint64_t v = (uint64_t)m->bias << sft;
                t[j] = (uint32_t)v; t[j + 1] = (uint32_t)(v >> 32);
                big_add(biasall, t);
            }
            for (int k = 0; k < 2; k++) {
                Big x = big_mul(biasall, Big(1, k == 0 ? 1u : 3u));
                x.resize(17);
                unsigned cy = 1;
                for (int j = 0; j < 17; j++) { const uint64_t t = (uint64_t)(uint32_t)~x[j] + cy; corr.w[k][j] = (uint32_t)t; cy = (unsigned)(t >> 32); }
            }
        }
        const size_t lds_f = mm8w_lds_bytes(m->n_rt, m->nkb, 1, flat_nb);
        const bool chk = check_mask_dev != nullptr;
#define MM8W_FLAT_LAUNCH(CHK)                                                                                                              \
    do {                                                                                                                                   \
        static std::atomic<unsigned long long> attr_done{0};                                                                               \
        if (!((attr_done.load() >> (ctx->device & 63)) & 1ull)) {                                                                          \
            HB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mm8w_flat<CHK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr_done.fetch_or(1ull << (ctx->device & 63));                                                                                \
        }                                                                                                                                  \
        hipLaunchKernelGGL((k_mm8w_flat<CHK>), dim3((unsigned)mm8w_num_cus()), dim3(256), lds_f, s, m->a8, m->crow, m->zero, in, iv.stride_c, iv.stride_l, \
                           in_rows_dev, in_count, m->d, out, ov.stride_c, ov.stride_l, out_count, check_mask_dev, mismatch_dev,           \
                           cmp ? cmp : out, cmp ? cv.stride_c : ov.stride_c, cmp ? cv.stride_l : ov.stride_l, cmp ? 1 : 0, cmp ? n_store : 0, \
                           m->n_out, m->n_rt, m->nkb, flat_nb, C, n_pass, m->bias, m->wp, first_bad_dev, bad_map_dev, done, corr);         \
    } while (0)
        if (chk) MM8W_FLAT_LAUNCH(true); else MM8W_FLAT_LAUNCH(false);
#undef MM8W_FLAT_LAUNCH
        HB_LAUNCH_CHECK(ctx);
        return HB_OK;
    }
    const int64_t n_units = ((n_tiles + tpw - 1) / tpw) * ((m->n_rt + rq - 1) / rq);
    int64_t blocks = mm8w_num_cus();
    if (blocks > n_units) blocks = n_units;
    const size_t lds = mm8w_lds_bytes(m->n_rt, m->nkb, tpw, nbuf);
    const bool check = check_mask_dev != nullptr;
#define MM8W_LAUNCH_K(CHK, PL, KK)                                                                                                    \
    do {                                                                                                                              \
        static std::atomic<unsigned long long> attr_done{0};   /* one bit per device: the attribute is per device */                                                                                                \
        if (!((attr_done.load() >> (ctx->device & 63)) & 1ull)) {                                                                                                             \
            HB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mm8w<CHK, PL, KK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            attr_done.fetch_or(1ull << (ctx->device & 63));                                                                                                         \
        }                                                                                                                             \
        hipLaunchKernelGGL((k_mm8w<CHK, PL, KK>), dim3((unsigned)blocks), dim3(256), lds, s, m->a8, m->crow, m->zero, in, iv.stride_c, iv.stride_l, \
                           in_rows_dev, in_count, m->d, out, ov.stride_c, ov.stride_l, out_count, check_mask_dev, mismatch_dev,      \
                           cmp ? cmp : out, cmp ? cv.stride_c : ov.stride_c, cmp ? cv.stride_l : ov.stride_l, cmp ? 1 : 0, cmp ? n_store : 0, \
                           m->n_out, m->n_rt, m->nkb, tpw, nbuf, rq, C, n_units, m->bias, m->wp, first_bad_dev, bad_map_dev, done);      \
    } while (0)
    // K-blocks written out with a share of the reduction each (gen_mm8w.py); the rest is a loop of two-block bodies
    const int peel = m->nkb <= 2 ? m->nkb : ((m->nkb & 1) ? 3 : 4);
#define MM8W_LAUNCH(CHK, PL) do { if (m->tile_rows == 12) MM8W_LAUNCH_K(CHK, PL, 3); else if (m->tile_rows == 8) MM8W_LAUNCH_K(CHK, PL, 2); else MM8W_LAUNCH_K(CHK, PL, 4); } while (0)
    if (check) { if (peel == 1) MM8W_LAUNCH(true, 1); else if (peel == 2) MM8W_LAUNCH(true, 2); else if (peel == 3) MM8W_LAUNCH(true, 3); else MM8W_LAUNCH(true, 4); }
    else { if (peel == 1) MM8W_LAUNCH(false, 1); else if (peel == 2) MM8W_LAUNCH(false, 2); else if (peel == 3) MM8W_LAUNCH(false, 3); else MM8W_LAUNCH(false, 4); }
#undef MM8W_LAUNCH_K
#undef MM8W_LAUNCH
    HB_LAUNCH_CHECK(ctx);
    return HB_OK;
}

int mm8w_tile_rows(int n_out, int nkb) {
    int tile_rows = 16;
    if (!env_hook(ENV_MM8W_TILE16)) {
        double best = 0.0;
        for (int tr = 16; tr >= 8; tr -= 4) {
            const double cost = (double)((n_out + tr - 1) / tr) * (228.0 * nkb + 160 + (tr / 4) * 490);
            if (tr == 16 || cost < best - 1e-9) { best = cost; tile_rows = tr; }
        }
    }
    return tile_rows;
}

// geometry of the int8 image of an n_out x d matrix (the same choices as mm8w_from_host)
int mm8w_geometry(int n_out, int d, int *tile_rows, int *n_rt, int *nkb, size_t *a8_bytes, size_t *crow_words) {
    if (n_out < 1 || d < 1) return HB_ERR_UNSUPPORTED;
    *nkb = (d + 7) / 8;
    *tile_rows = mm8w_tile_rows(n_out, *nkb);
    *n_rt = (n_out + *tile_rows - 1) / *tile_rows;
    int tpw = 0, nbuf = 0, rq = 0;
    if (!mm8w_shape(*n_rt, *nkb, 1, 1, &tpw, &nbuf, &rq)) return HB_ERR_UNSUPPORTED;
    *a8_bytes = ((size_t)*n_rt * *nkb + 1) * 4 * 64 * 16;          // one block of padding
    *crow_words = (size_t)*n_rt * 16 * 16;
    return HB_OK;
}

// What every device-built image of inner dimension d shares: the reduction constants (one device copy per context), the zero
// block, and a bias that bounds the columns of ANY matrix of that width (|digit| <= 128, 32 digits, d terms) together with the
// two constants of the per-row correction, (0x80..80 R) mod p and (bias sum_c 2^(8c)) mod p.
int mm8w_shared(hb_ctx *ctx, int d, const Mm8wShared **out, hipStream_t s) {
    auto it = ctx->wide_shared.find(d);
    if (it != ctx->wide_shared.end()) { *out = static_cast<const Mm8wShared *>(it->second); return HB_OK; }
    if (ctx->n_limbs != 4 || !prescale_params(ctx)) return fail(ctx, HB_ERR_UNSUPPORTED, "mm8w: modulus");
    const uint64_t bias64 = 128ull * ((uint64_t)d * 32 * 128) + 1;
    if (bias64 >= (1ull << 30)) return fail(ctx, HB_ERR_UNSUPPORTED, "mm8w: column bound too large");
    Mm8wShared *sh = new Mm8wShared();
    sh->bias = (uint32_t)bias64; sh->wp = nullptr; sh->zero = nullptr;
    const Big p = big_from_limbs(ctx->p_limbs, 4);
    Big biasall(17, 0);
    for (int c = 0; c < MM8W_NC; c++) {
        const int bit = 8 * c, j = bit >> 5, sft = bit & 31;
        Big t(17, 0);
        const uint64_t v = (uint64_t)sh->bias << sft;
        t[j] = (uint32_t)v; t[j + 1] = (uint32_t)(v >> 32);
        big_add(biasall, t);
    }
    FoldConsts fc;
    if (!fold_consts(ctx, &fc)) { delete sh; return fail(ctx, HB_ERR_UNSUPPORTED, "mm8w: fold table"); }
    to_digits(fold_biasmod(biasall, p, fc.shift), sh->biasmod, 9);
    Big c80(8, 0x80808080u);
    to_digits(big_mod(big_mul(c80, big_pow2(261, 10)), p), sh->c80r, 9);
    // the context-wide device copies (shared by every d)
    void *wp = nullptr;
    uint32_t *zero = nullptr;
    for (auto &kv : ctx->wide_shared) { wp = static_cast<Mm8wShared *>(kv.second)->wp; zero = static_cast<Mm8wShared *>(kv.second)->zero; break; }
    if (!wp) {
        const WideParams &wph = fc.wp;
        std::vector<uint8_t> zero64(64, 0), wpb((sizeof(WideParams) + 15) / 16 * 16, 0);
        memcpy(wpb.data(), &wph, sizeof wph);
        hipError_t e = hipMalloc(&wp, wpb.size());
        if (e == hipSuccess) e = hipMalloc(&zero, 64);
        int rc = e == hipSuccess ? upload_table(ctx, wp, wpb.data(), wpb.size(), s) : HB_ERR_HIP;
        if (!rc) rc = upload_table(ctx, zero, zero64.data(), 64, s);
        if (rc) { if (wp) (void)hipFree(wp); if (zero) (void)hipFree(zero); delete sh; return rc == HB_ERR_HIP ? fail(ctx, rc, "mm8w: shared tables") : rc; }
    }
    sh->wp = wp; sh->zero = zero;
    ctx->wide_shared[d] = sh;
    *out = sh;
    return HB_OK;
}

void mm8w_shared_free(hb_ctx *ctx) {
    bool first = true;
    for (auto &kv : ctx->wide_shared) {
        Mm8wShared *sh = static_cast<Mm8wShared *>(kv.second);
        if (first) { (void)hipFree(sh->wp); (void)hipFree(sh->zero); first = false; }
        delete sh;
    }
    ctx->wide_shared.clear();
}

int launch_mm8w_raw(hb_ctx *ctx, int n_out, int d, int tile_rows, const void *a8, const uint32_t *crow, const Mm8wShared *sh,
                    const uint32_t *in, hb_view iv, const int32_t *in_rows_dev, int64_t in_count, uint32_t *out, hb_view ov, int64_t out_count,
                    const int32_t *check_mask_dev, int32_t *mismatch_dev, int64_t C, hipStream_t s, const uint32_t *cmp, hb_view cv, int n_store,
                    int32_t *first_bad_dev, uint32_t *bad_map_dev, const FsDone *done) {
    Mm8wMatrix m;
    m.n_out = n_out; m.d = d; m.nkb = (d + 7) / 8; m.tile_rows = tile_rows; m.n_rt = (n_out + tile_rows - 1) / tile_rows;
    m.shape_tiles = -1; m.shape_tpw = m.shape_nbuf = m.shape_rq = m.shape_flat = 0;
    m.a8 = (int4 *)const_cast<void *>(a8); m.crow = const_cast<uint32_t *>(crow); m.zero = sh->zero; m.bias = sh->bias; m.wp = (WideParams *)sh->wp;
    return launch_mm8w_impl(ctx, &m, in, iv, in_rows_dev, in_count, out, ov, out_count, check_mask_dev, mismatch_dev, C, s, cmp, cv, n_store, first_bad_dev, bad_map_dev, done);
}

}  // namespace hb
