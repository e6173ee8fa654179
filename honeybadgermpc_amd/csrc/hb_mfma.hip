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
// so that for a fixed column c the sum over (l, b) is an int8 dot product between a constant row (the digits of M)
// and a sliding window of bytes of every input element.
// v_mfma_i32_16x16x64_i8 takes 16 rows i x 16 chunks x 64 products per instruction, cut here as 8 elements l x 8 digits b:
// the 16 digits are two groups G, and for column c and group G the window is bytes [s, s + 7], s = c - 7 - 8 G, of the
// element, i.e. dwords q, q + 1 of the element shifted right by rho bytes (s = 4q + rho).  How the shifted copies are kept
// in an aligned register file so that no operand is ever assembled per MFMA is gen_mm8.py's business (it emits the
// MFMA phases as inline asm, hb_mm8_body.inc, and says why 8 x 8 beats 4 elements x 16 digits).
//
// int8 operands are signed: M uses balanced digits; the input bytes are biased by XOR 0x80
// (u = s + 128) and the constant 128 * sum_a 2^(8a) * sum_l M[i][l] is added back per row, mod p,
// together with a multiple of p that keeps the total non-negative and minus the accumulator bias
// (crow[], radix-2^29 digits).
//
// Epilogue per output: 47 int32 columns -> pairs E = col + col' 2^8 (< 2^32) -> 13 words by one
// add-with-carry each.  S = L + 2^256 H + 2^384 w12 (L eight words, H four): H goes back through the matrix
// cores against t_b = 2^(256 + 8b) mod p, b < 16 -- a block-diagonal product, 8 MFMAs per output, the table and the
// layout of hb_mfma_wide.hip / gen_mm8w.py (round 3; rounds 1-2 reduced 14 radix-2^29 digits by a six-digit Barrett
// quotient: 26 + 35 v_mad_u64_u32 and the digit conversions both ways) -- its 32 columns, L, the top word times
// 2^384 mod p and the per-row constant are gathered per 32-bit word (six MADs a word), a one-word Barrett quotient,
// R - q p in words, one conditional subtraction.  No Montgomery form anywhere on this path.
#include <atomic>

#include "hb_mm8.hpp"

namespace hb {

#include "hb_mm8_body.inc"

// Persistent workgroups of 4 waves, 2 workgroups per CU (2 waves per SIMD, <= 256 registers each,
// so that one wave's MFMA stream overlaps its neighbour's VALU epilogue).
//
// A unit is TPW tiles of 16 chunks.  Its input elements are DMA'd (global_load_lds_dwordx4) into one
// of two LDS buffers in MFMA-operand order -- slot (tile, kb, e, half)[lane] x 16 B, lane (n, g) owning
// elements l = 8 kb + 2 g + e, e = 0, 1, of chunk n -- one unit ahead of the arithmetic.  Wave w of the workgroup takes
// (tile tl, row tile rt) pairs w, w + 4, ... of the unit's TPW x n_rt, pair index = tl n_rt + rt (rt: 16 output rows).
//
// Per (tile, rt) the 47 columns are produced in two halves by byte shift (rho in {0,1}: 23 columns, rho in
// {2,3}: 24).  Accumulators start from BIAS so that every column is non-negative; a pair of adjacent
// columns then fits 32 bits, and the pairs of the two halves interleave into the 13 words of the sum with
// one add-with-carry per word.  Half 0's twelve values per output are parked across the second MFMA block.
// The bias, the XOR-0x80 correction and a multiple of p are one per-row constant (crowd), added digit-wise
// (radix 2^29) before the Barrett reduction.
// CHECK: out_pk holds the expected values; rows with check_mask[i] != 0 are compared, any difference
// sets *mismatch (the validating re-encode of reed_solomon.py:316-326).
#ifdef HB_MM8_TIMING
// debug build only (scratch/mm8_phase_timing.py): per-wave cycle sums of the phases of a pass
__device__ unsigned long long g_mm8_t[2048 * 8];
#define MM8_T(k) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tacc[k] += tn_ - tlast; tlast = tn_; } while (0)
#else
#define MM8_T(k) do { } while (0)
#endif
// SKIP: no entry of the matrix has a digit above the eighth in its first eight terms -- K-block 0 runs without its second digit group
// (gen_mm8.py; checked when the image is built: Vandermonde matrices at the points 1 .. n, whose l-th powers stay below 2^56 for l < 8)
template <int NKB, bool CHECK, bool RAGGED, bool SKIP>
__global__ __launch_bounds__(256, 2) void k_mm8(const int4 *__restrict__ a8, const uint32_t *__restrict__ crowd,
                                                const uint32_t *__restrict__ zero_src,
                                                const uint32_t *__restrict__ in_pk, int64_t in_sc, int64_t in_sl,
                                                const int32_t *__restrict__ in_rows, int64_t in_count, int d,
                                                uint32_t *__restrict__ out_pk, int64_t out_sc, int64_t out_sl, int64_t out_count,
                                                const int32_t *__restrict__ check_mask, const int32_t *__restrict__ check_rows, int32_t *__restrict__ mismatch,
                                                uint32_t *__restrict__ copy_dst, int64_t copy_sc, int64_t copy_sl, int64_t copy_count, int copy_rows,
                                                int n_out, int n_rt, int tpw, int64_t n_chunks, int64_t n_units, BarrettParams bp) {
    extern __shared__ uint4 mm8_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    uint32_t *crl = reinterpret_cast<uint32_t *>(mm8_lds);     // [n_rt * 16][16]: row constants as eight 64-bit addends
    int4 *abuf = reinterpret_cast<int4 *>(mm8_lds + n_rt * 64);   // [n_rt][NKB][2][64] matrix digits
    uint4 *xbuf = mm8_lds + n_rt * 64 + n_rt * NKB * 2 * 64;    // 2 x [tpw][NKB][2 elements][2 halves][64] uint4
    const int bufsz = tpw * NKB * 4 * 64;

    const int n_pairs = tpw * n_rt;      // (tile, row tile) pairs of a unit, dealt to the 4 waves: wave w takes w, w + 4, ...

    // l -> input row table (arrival order for decodes), clamped to d - 1
    int32_t *rowl = reinterpret_cast<int32_t *>(xbuf + 2 * bufsz);
    uint32_t *rowoff = reinterpret_cast<uint32_t *>(rowl + 32);   // byte offset of row l from the chunk's first element (fast DMA path)
    v4i *foldl = reinterpret_cast<v4i *>(rowl + 96 + 16 * n_rt);   // the fold table (16-byte aligned: everything before it is a multiple of 64 bytes)
    if (threadIdx.x < 32) {
        const int lc = (int)threadIdx.x < d ? (int)threadIdx.x : d - 1;
        const int row = in_rows ? in_rows[lc] : lc;
        rowl[threadIdx.x] = row;
        rowoff[threadIdx.x] = (uint32_t)((int64_t)row * in_sl * 32);
        int m = row;
        for (int o = 16; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 32));
        if (threadIdx.x == 0) rowl[64] = m;
    }
    int32_t *maskl = rowl + 96;   // CHECK: rows to compare
    if constexpr (CHECK) {
        // maskl[i] != 0: compare output row i; with a row map (compact check matrices) it holds 1 + the row of the
        // expected buffer that output i is compared with
        for (int i = threadIdx.x; i < n_rt * 16; i += 256) maskl[i] = (i < n_out && check_mask[i]) ? (check_rows ? check_rows[i] + 1 : i + 1) : 0;
    }
    __syncthreads();
    const int rowmax = __builtin_amdgcn_readfirstlane(rowl[64]);
    // interior units (every element present, offsets within 32 bits) take their addresses from one scalar base per slot
    // plus a per-lane 32-bit offset: the general form below costs ~40 VALU instructions per slot
    const bool fast_dma = in_sc >= 0 && in_sl >= 0 && in_count < (1ll << 27);
    const uint32_t lane_off = (uint32_t)((int64_t)n * in_sc * 32);
    const int n_slots = tpw * NKB * 4;
    // slot s = ((t * NKB + kb) * 2 + e) * 2 + h, dealt round-robin to the 4 waves; all of it wave-uniform
    auto issue_loads = [&](int64_t unit, int buf) {
        const int64_t c_last = (unit * tpw + tpw) * 16 - 1;
        if (fast_dma && c_last < n_chunks && c_last * in_sc + (int64_t)rowmax * in_sl < in_count) {
            for (int s = wave; s < n_slots; s += 4) {
                const int h = s & 1, e = (s >> 1) & 1, q = s >> 2, t = q / NKB, kb = q - t * NKB;
                const uint64_t sbase = (uint64_t)(uintptr_t)in_pk + (uint64_t)((unit * tpw + t) * 16 * in_sc) * 32 + h * 16;
                const uint32_t voff = lane_off + rowoff[8 * kb + 2 * g + e];
                const uint32_t lds_dst = __builtin_amdgcn_readfirstlane(
                    (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(xbuf + (size_t)buf * bufsz + s * 64));
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
            }
            return;
        }
        for (int s = wave; s < n_slots; s += 4) {
            const int h = s & 1, e = (s >> 1) & 1, q = s >> 2, t = q / NKB, kb = q - t * NKB;
            int64_t chunk = (unit * tpw + t) * 16 + n;
            if (chunk >= n_chunks) chunk = n_chunks - 1;
            const int64_t idx = chunk * in_sc + (int64_t)rowl[8 * kb + 2 * g + e] * in_sl;
            const uint4 *src = (idx < in_count) ? reinterpret_cast<const uint4 *>(in_pk) + idx * 2 + h
                                                : reinterpret_cast<const uint4 *>(zero_src) + h;
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
    uint32_t k256 = 256u, k16m = 1u << 24;      // opaque, so that the word assembly stays two v_mad_u64_u32 per word
    int32_t s1 = 1, s256 = 256, s64k = 1 << 16, s16m = 1 << 24;   // and the gathering of the fold's columns one v_mad_i64_i32 each
    uint32_t u1 = 1u;
    asm volatile("" : "+s"(k256), "+s"(k16m), "+s"(s1), "+s"(s256), "+s"(s64k), "+s"(s16m), "+s"(u1));

    int buf = 0;
    int64_t unit = blockIdx.x;
#ifdef HB_MM8_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    if (unit < n_units) issue_loads(unit, 0);
    // the constant tables ride behind the first unit's DMA instead of in front of it
    // (crowd and the digits are adjacent in LDS; LDS-DMA like the elements, 1 KB per instruction, all in flight at once --
    // a load / wait / ds_write loop cost one round trip per 4 KB, 3 us per launch)
    {
        const int n_cr = n_rt, n_blk = n_rt * (1 + 2 * NKB);  // in 64-lane blocks of uint4
        for (int blk = wave; blk < n_blk; blk += 4) {
            const uint4 *src = (blk < n_cr ? reinterpret_cast<const uint4 *>(crowd) + blk * 64
                                           : reinterpret_cast<const uint4 *>(a8) + (blk - n_cr) * 64) + lane;
            const uint32_t lds_dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(mm8_lds + blk * 64));
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
        }
    }
    if (threadIdx.x < MM8_FOLD_Q) foldl[threadIdx.x] = reinterpret_cast<const v4i *>(crowd + (size_t)n_rt * 256)[threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // the fold's A operand: lane (m, g') of the diagonal block g' = m / 4 reads its 16 digits, every other lane the row's zero bytes
    const v4i *fold_lane = reinterpret_cast<const v4i *>(reinterpret_cast<const char *>(foldl) + (g == (n >> 2) ? 16 * n : 256));
    MM8_T(0);   // prologue
    for (; unit < n_units; unit += gridDim.x, buf ^= 1) {
        // every wave has waited for its share of this unit's DMA (below, before its epilogue) and is
        // done reading the other buffer; no vmcnt wait here, so the output stores stay in flight
        __builtin_amdgcn_s_barrier();
        MM8_T(1);   // barrier wait
        if (unit + gridDim.x < n_units) issue_loads(unit + gridDim.x, buf ^ 1);
        MM8_T(2);   // DMA issue
        if constexpr (CHECK) {
            // hand the caller its rows of the input (the decoded coefficients) in its own layout while they are in LDS
            if (copy_dst) {
                for (int e = wave; e < tpw * NKB * 2; e += 4) {
                    const int t = e / (NKB * 2), kb = (e >> 1) - t * NKB, l = 8 * kb + 2 * g + (e & 1);
                    const int64_t ch = (unit * tpw + t) * 16 + n;
                    const int64_t idx = ch * copy_sc + (int64_t)l * copy_sl;
                    if (l < copy_rows && l < d && ch < n_chunks && idx < copy_count) {
                        const uint4 *src = xbuf + (size_t)buf * bufsz + (size_t)e * 2 * 64 + lane;
                        uint4 *dst = reinterpret_cast<uint4 *>(copy_dst) + idx * 2;
                        dst[0] = src[0];
                        dst[1] = src[64];
                    }
                }
            }
        }
        if (wave >= n_pairs) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // a wave without a row tile still owns part of the DMA
        for (int pidx = wave; pidx < n_pairs; pidx += 4) {
            const int tl = pidx / n_rt, rt = pidx - tl * n_rt;
            const int64_t chunk = (unit * tpw + tl) * 16 + n;
            const uint4 *xs = xbuf + (size_t)buf * bufsz + (size_t)tl * NKB * 4 * 64 + lane;
            const int4 *as = abuf + (size_t)rt * NKB * 2 * 64 + lane;
            uint32_t eap[4][11], c0p[4];     // half 0's column pairs, parked across the second MFMA block
            // Output `reg` of lane (n, g) is row 16 rt + 4 reg + g (the host places matrix row 16 rt + j at tile row
            // 4 (j % 4) + j / 4), so the rows beyond n_out of a ragged last tile fill whole outputs from the top: when
            // outputs 2 and 3 are all padding (22 rows: the decode's second tile) their reduction is skipped
            const int64_t obase = chunk * out_sc + (int64_t)(16 * rt + g) * out_sl, ostep = 4 * out_sl;   // element index of output reg
            const bool pair1 = !RAGGED || 16 * rt + 8 < n_out;   // RAGGED is instantiated only where it pays (launch_mm8)
            const uint32_t xs_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)xs;
            const uint32_t as_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)as;
            {   // ---- half 0: c = 0 and the pairs (4j+3, 4j+4); E_j = col_4j+3 + col_4j+4 2^8 sits at bit 32j + 24
                v4i acc[24];
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // the MFMA phase yields to its neighbour's reduction: whatever runs outside the two phases (the reductions, the DMA issue) has the
                // higher priority -- a wave in a phase loses little by waiting a slot (the matrix pipe is 16 cycles an MFMA whatever rides beside
                // it), a wave in its reduction holds the unit's barrier (round 6: the open 136.5 -> 133.3 us; the phases ABOVE the rest: 135.5)
                __builtin_amdgcn_s_setprio(0);
                Mm8Phase<NKB, 0, SKIP>::run(acc, xs_addr, as_addr, biasv);
                __builtin_amdgcn_s_setprio(2);
                __builtin_amdgcn_sched_barrier(0);
                MM8_T(3);   // MFMA half 0
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {   // the accumulators die here
                    if (reg == 2 && !pair1) break;
                    c0p[reg] = (uint32_t)acc[0][reg];
#pragma unroll
                    for (int j = 0; j < 11; j++) eap[reg][j] = (uint32_t)acc[1 + 2 * j][reg] + ((uint32_t)acc[2 + 2 * j][reg] << 8);
                    // pin the values here: without a use ordered against the asm phases hipcc sinks this below the
                    // second MFMA block and keeps both sets of accumulators alive
                    asm volatile("" ::"v"(c0p[reg]), "v"(eap[reg][0]), "v"(eap[reg][1]), "v"(eap[reg][2]), "v"(eap[reg][3]), "v"(eap[reg][4]),
                                 "v"(eap[reg][5]), "v"(eap[reg][6]), "v"(eap[reg][7]), "v"(eap[reg][8]), "v"(eap[reg][9]), "v"(eap[reg][10]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            {   // ---- half 1: the pairs (4k+1, 4k+2) at bit 32k + 8.  S = col_0 + sum_k (F_k 2^8 + E_k 2^24) 2^(32k)
                v4i acc[24];
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                MM8_T(4);   // parking of half 0
                __builtin_amdgcn_s_setprio(0);
                Mm8Phase<NKB, 1, SKIP>::run(acc, xs_addr, as_addr, biasv);
                __builtin_amdgcn_s_setprio(2);
                __builtin_amdgcn_sched_barrier(0);
                MM8_T(5);   // MFMA half 1
                // next unit's DMA (issued a pass ago) must have landed before this wave reaches the barrier; waiting
                // here, ahead of the epilogue, keeps this pass's output stores out of the wait
                if (pidx + 4 >= n_pairs) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                MM8_T(6);   // wait for the next unit's DMA (and the previous pass's stores)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    if (reg == 2 && !pair1) break;
                    const int i = 16 * rt + 4 * reg + g;
                    uint32_t ew[8];
                    bool cmp = false;
                    if constexpr (CHECK) {   // expected value: in flight while this output is reduced
                        const int erow = maskl[i];
                        cmp = (chunk < n_chunks) && erow;
                        // unconditional (from the buffer's first element when there is nothing to compare): a load under `cmp` makes hipcc wrap
                        // the whole reduction in that divergent branch, and the register copies at its join spill
                        load_words<8>(ew, out_pk + (cmp ? (chunk * out_sc + (int64_t)(erow - 1) * out_sl) * 8 : 0));
                    }
                    (void)ew; (void)cmp;
                    // G_k = F_k 2^8 + E_k 2^24 < 2^56 (+ col_0 for k = 0), S = sum_k G_k 2^(32 k).  The low eight go into the gathered words
                    // P_w as they are -- P_w is a 64-bit addend of weight 2^(32 w), so no carry chain and no separate addition: the two
                    // MADs that build G_w start from the row constant's pair [bias of the fold's four columns + constant word] -- except
                    // that what G_7 has above bit 256 joins the high part.  H = that + G_8 .. G_11, exact words by add-with-carry.
                    uint64_t pw[8];
                    {
                        const uint4 *cr = reinterpret_cast<const uint4 *>(crl + (size_t)i * 16);
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const uint4 c = cr[k];
                            pw[2 * k] = (uint64_t)c.x | ((uint64_t)c.y << 32);
                            pw[2 * k + 1] = (uint64_t)c.z | ((uint64_t)c.w << 32);
                        }
                    }
                    uint32_t glo[5], ghi[5];       // G_7 .. G_11
#pragma unroll
                    for (int k = 0; k < 12; k++) {
                        const uint32_t f = (uint32_t)acc[2 * k][reg] + ((uint32_t)acc[2 * k + 1][reg] << 8);
                        uint64_t gk = (uint64_t)f * k256 + (k < 7 ? pw[k] : 0ull);
                        if (k < 11) gk += (uint64_t)eap[reg][k] * k16m;
                        if (k < 7) pw[k] = gk;
                        else { glo[k - 7] = (uint32_t)gk; ghi[k - 7] = (uint32_t)(gk >> 32); }
                    }
                    pw[0] += (uint64_t)c0p[reg] * u1;
                    pw[7] += (uint64_t)glo[0] * u1;
                    uint32_t hw[5];
                    {
                        unsigned cyw = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) hw[k] = __builtin_addc(glo[k + 1], ghi[k], cyw, &cyw);
                        hw[4] = ghi[4] + cyw;
                    }
                    // the fold of H (biased like every int8 input here; the row constant takes 128 sum_b t_b back out): eight MFMAs
                    v4i hb;
#pragma unroll
                    for (int k = 0; k < 4; k++) hb[k] = (int)(hw[k] ^ 0x80808080u);
                    v4i dcol[8];
#pragma unroll
                    for (int eb = 0; eb < 8; eb++)
                        dcol[eb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fold_lane[eb * (MM8_FOLD_ROW / 16)], hb, v4i{0, 0, 0, 0}, 0, 0, 0);
                    // P_w += hw4 (2^384 mod p)_w + sum_k D_(4 w + k) 2^(8 k)      (P_w < 2^58, P_7 < 2^47)
#pragma unroll
                    for (int k = 0; k < 8; k++) pw[k] += (uint64_t)hw[4] * bp.c384[k];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        int64_t t = (int64_t)pw[k] + (int64_t)dcol[k][0] * s1;
                        t += (int64_t)dcol[k][1] * s256;
                        t += (int64_t)dcol[k][2] * s64k;
                        t += (int64_t)dcol[k][3] * s16m;
                        pw[k] = (uint64_t)t;
                    }
                    // R = sum_w P_w 2^(32 w) < 2^272;  qhat = floor(floor(R / 2^240) mu / 2^46), mu = floor(2^286 / p): floor(R / p) or one
                    // less (tests/fold_model.py has the bounds);  u_w = qhat (2^256 - p)_w + P_w: the words of sum_w u_w 2^(32 w)
                    // are R - qhat p, with qhat on top of bit 256
                    const uint64_t tq = pw[7] + (pw[6] >> 32);
                    const uint32_t qh = (uint32_t)(((uint64_t)(uint32_t)(tq >> 16) * bp.mu) >> 46);
#pragma unroll
                    for (int k = 0; k < 8; k++) pw[k] += (uint64_t)qh * bp.pneg[k];
                    uint32_t ow[8];
                    uint32_t top;
                    {
                        unsigned cy = 0;
                        ow[0] = (uint32_t)pw[0];
#pragma unroll
                        for (int k = 1; k < 8; k++) ow[k] = __builtin_addc((uint32_t)pw[k], (uint32_t)(pw[k - 1] >> 32), cy, &cy);
                        top = (uint32_t)(pw[7] >> 32) + cy - qh;          // bit 256 of the remainder (r < 2p < 2^257)
                    }
                    // r >= p  <=>  bit 256 of r is set (possible once p > 2^255) or r mod 2^256 + (2^256 - p) carries out of word 7;
                    // either way r - p is that sum mod 2^256
                    // -- behind a test that almost never fires: the quotient is one short once in ~10^4 outputs, and otherwise r < p shows
                    // in the top word alone (ow_7 + (2^256 - p)_7 + 1 < 2^32 leaves no room for a carry out, whatever the lower words do)
                    if (__builtin_amdgcn_ballot_w64(top != 0 || ow[7] >= ~bp.pneg[7]) != 0) {
                        uint32_t u[8];
                        unsigned cy2 = 0;
#pragma unroll
                        for (int k = 0; k < 8; k++) u[k] = __builtin_addc(ow[k], bp.pneg[k], cy2, &cy2);
                        const bool take = cy2 || top;
#pragma unroll
                        for (int k = 0; k < 8; k++) ow[k] = take ? u[k] : ow[k];
                    }
                    const int64_t oidx = obase + reg * ostep;
                    if constexpr (CHECK) {
                        uint32_t diff = 0;
#pragma unroll
                        for (int k = 0; k < 8; k++) diff |= ew[k] ^ ow[k];
                        asm volatile("" : "+v"(diff));
                        // once per wave, and only while the flag is still clear: a sender that corrupts everything must not queue a million atomics
                        if (__builtin_amdgcn_ballot_w64(cmp && diff) != 0 && lane == 0 && *reinterpret_cast<volatile int32_t *>(mismatch) == 0) atomicOr(mismatch, 1);
                    } else {
                        // keep the reduction outside the store's exec mask: hipcc otherwise wraps the whole output in a
                        // divergent branch, and the register copies at its join spill
                        asm volatile("" ::"v"(ow[0]), "v"(ow[1]), "v"(ow[2]), "v"(ow[3]), "v"(ow[4]), "v"(ow[5]), "v"(ow[6]), "v"(ow[7]));
                        // party-major outputs (an encode: the wave's sixteen chunks of a row are 512 contiguous bytes) leave with the streaming
                        // hint: nobody on this GPU reads them next, and 97 MB of them parked in the caches held up the loads of the launch
                        // behind (config 3's open 134.5 -> 128.0 us, round 6).  Chunk-major outputs are 128-byte pieces: the hint costs there
                        // (k_mm8f's R2 launch 49.6 -> 57 us with it), so they stay ordinary stores
                        if (chunk < n_chunks && i < n_out && oidx < out_count) {
                            bool stream_out = false;
                            if constexpr (SKIP && !RAGGED) stream_out = out_sc == 1;      // (SKIP: Vandermonde matrices at small points -- the encodes)
                            if (stream_out) store_words_nt<8>(out_pk + oidx * 8, ow);
                            else store_words<8>(out_pk + oidx * 8, ow);
                        }
                    }
                    if (reg & 1) __builtin_amdgcn_sched_barrier(0);   // two reductions at a time: ILP for the carry chains
                }
            }
            MM8_T(7);   // word assembly + reduction + stores
        }
    }
#ifdef HB_MM8_TIMING
    if (lane == 0 && blockIdx.x < 512) for (int k = 0; k < 8; k++) g_mm8_t[(blockIdx.x * 4 + wave) * 8 + k] = tacc[k];
#endif
}

int64_t mm8_trimmed_grid(int64_t n_units, int64_t blocks) {
    if (blocks < 1 || n_units < 1) return blocks;
    const int64_t rounds = (n_units + blocks - 1) / blocks;
    return (n_units + rounds - 1) / rounds;
}

}  // namespace hb

using namespace hb;

// ---- host side -----------------------------------------------------------------------------------
namespace {

// dynamic LDS of k_mm8: row constants, matrix digits, two element buffers, then the int tables of the prologue
// (rowl[32], rowoff[32], rowmax + padding [32], maskl[16 n_rt]) and the fold table
constexpr size_t MM8_LDS_LIMIT = 78 * 1024;   // two workgroups per CU share 160 KB
size_t mm8_lds_bytes(int n_rt, int nkb, int tpw);
// tiles of 16 chunks per unit: enough (tile, row tile) pairs for the 4 waves.  (Two tiles per unit with four row tiles --
// two passes per barrier -- measured the same 65 us as one: the barrier is not what the kernel waits for.)
int mm8_tpw(int n_rt, int nkb) {
    int tpw = (n_rt == 1) ? 4 : (n_rt == 2) ? 2 : 1;
    return tpw;
}
size_t mm8_lds_bytes(int n_rt, int nkb, int tpw) {
    return ((size_t)n_rt * 64 + (size_t)n_rt * nkb * 2 * 64 + (size_t)2 * tpw * nkb * 4 * 64 + MM8_FOLD_Q) * 16 + 96 * 4 + (size_t)n_rt * 16 * 4;
}

int mm8_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

// little multi-word helpers for the one-time table build (32-bit words, little endian)
typedef std::vector<uint32_t> Big;
Big big_mul(const Big &a, const Big &b) {
    Big r(a.size() + b.size(), 0);
    for (size_t i = 0; i < a.size(); i++) {
        uint64_t cy = 0;
        for (size_t j = 0; j < b.size(); j++) {
            uint64_t t = (uint64_t)a[i] * b[j] + r[i + j] + cy;
            r[i + j] = (uint32_t)t; cy = t >> 32;
        }
        r[i + b.size()] = (uint32_t)cy;
    }
    return r;
}
bool big_ge(const Big &a, const Big &b) {   // same length
    for (size_t i = a.size(); i-- > 0;) if (a[i] != b[i]) return a[i] > b[i];
    return true;
}
void big_sub(Big &a, const Big &b) {        // a -= b, same length, a >= b
    int64_t br = 0;
    for (size_t i = 0; i < a.size(); i++) {
        int64_t t = (int64_t)a[i] - (i < b.size() ? b[i] : 0) + br;
        a[i] = (uint32_t)t; br = t >> 32;
    }
}
void big_add(Big &a, const Big &b) {        // a += b (a at least as long; final carry dropped)
    uint64_t cy = 0;
    for (size_t i = 0; i < a.size(); i++) {
        uint64_t t = (uint64_t)a[i] + (i < b.size() ? b[i] : 0) + cy;
        a[i] = (uint32_t)t; cy = t >> 32;
    }
}
// binary long division of x by p (p padded to np words): quotient into q (same length as x), remainder returned
Big big_divmod(const Big &x, const Big &p, Big *q) {
    Big r(p.size() + 1, 0), pp(p); pp.push_back(0);
    if (q) q->assign(x.size(), 0);
    for (int bit = (int)x.size() * 32 - 1; bit >= 0; bit--) {
        uint32_t in = (x[bit >> 5] >> (bit & 31)) & 1u;
        for (size_t i = r.size(); i-- > 0;) r[i] = (r[i] << 1) | (i ? r[i - 1] >> 31 : in);
        if (big_ge(r, pp)) { big_sub(r, pp); if (q) (*q)[bit >> 5] |= 1u << (bit & 31); }
    }
    r.pop_back();
    return r;
}
Big big_shl(const Big &a, int bits, size_t words) {
    Big r(words, 0);
    for (size_t i = 0; i < a.size(); i++) {
        const size_t w = i + bits / 32; const int s = bits % 32;
        if (w < words) r[w] |= a[i] << s;
        if (s && w + 1 < words) r[w + 1] |= a[i] >> (32 - s);
    }
    return r;
}

}  // namespace

namespace hb {

bool prescale_params(hb_ctx *ctx) {
    if (ctx->psc_state) return ctx->psc_state > 0;
    ctx->psc_state = -1;
    if (ctx->n_limbs != 4 || (ctx->p_limbs[3] >> 62) == 0) return false;       // 2^254 <= p is what the quotient bound assumes
    Big p(8);
    for (int i = 0; i < 4; i++) { p[2 * i] = (uint32_t)ctx->p_limbs[i]; p[2 * i + 1] = (uint32_t)(ctx->p_limbs[i] >> 32); }
    Big pw9(p); pw9.push_back(0);
    Big pb(9, 0); pb[8] = 1u << 5;            // 2^261
    big_sub(pb, pw9);
    for (int k = 0; k < 9; k++) {
        const int bit = 29 * k, j = bit >> 5, sft = bit & 31;
        const uint64_t v = pb[j] | ((uint64_t)(j + 1 < 9 ? pb[j + 1] : 0) << 32);
        ctx->psc.pbar[k] = (uint32_t)(v >> sft) & DMASK;
    }
    Big pn(9, 0); pn[8] = 1;                   // 2^256
    big_sub(pn, pw9);
    for (int k = 0; k < 8; k++) ctx->psc.pneg[k] = pn[k];
    Big two290(10, 0); two290[9] = 1u << 2;    // 2^290
    Big mu; big_divmod(two290, p, &mu);        // < 2^37
    const uint64_t muv = mu[0] | ((uint64_t)mu[1] << 32);
    ctx->psc.m0 = (uint32_t)muv & DMASK;
    ctx->psc.m1 = (uint32_t)(muv >> LB);
    ctx->psc_state = 1;
    return true;
}

void mm8_free(Mm8Matrix *m) {
    if (!m) return;
    (void)hipFree(m->a8); (void)hipFree(m->crow); (void)hipFree(m->zero);
    delete m;
}

// The per-context constants of this kernel family (hb_mm8.hpp), for images that are built on the device (hb_mfma_fused.hip).
int mm8_shared(hb_ctx *ctx, const Mm8Shared **out, hipStream_t s) {
    if (ctx->mm8_shared) { *out = static_cast<const Mm8Shared *>(ctx->mm8_shared); return HB_OK; }
    if (ctx->n_limbs != 4 || (ctx->p_limbs[3] >> 62) == 0 || !prescale_params(ctx)) return fail(ctx, HB_ERR_UNSUPPORTED, "mm8: modulus");
    Mm8Shared *sh = new Mm8Shared();
    memset(sh, 0, sizeof *sh);
    Big p(8);
    for (int k = 0; k < 4; k++) { p[2 * k] = (uint32_t)ctx->p_limbs[k]; p[2 * k + 1] = (uint32_t)(ctx->p_limbs[k] >> 32); }
    uint32_t shift8[8];
    if (!fold_tables(ctx, 16, sh->fold_host, sh->bp.c384, &sh->bp.mu, shift8)) { delete sh; return fail(ctx, HB_ERR_UNSUPPORTED, "mm8: fold table"); }
    {   // 2^256 - p, 32-bit words
        Big pn(9, 0); pn[8] = 1;
        Big pw9(p); pw9.push_back(0);
        big_sub(pn, pw9);
        for (int k = 0; k < 8; k++) sh->bp.pneg[k] = pn[k];
    }
    auto digits9 = [](const Big &v, uint32_t *dg) {      // v < 2^256 as nine radix-2^29 digits
        for (int k = 0; k < 9; k++) {
            const int bit = 29 * k, j = bit >> 5, sft = bit & 31;
            const uint64_t w = (uint64_t)(j < (int)v.size() ? v[j] : 0) | ((uint64_t)(j + 1 < (int)v.size() ? v[j + 1] : 0) << 32);
            dg[k] = (uint32_t)(w >> sft) & DMASK;
        }
    };
    Big biasmod;
    { Big bt(14, 0); for (int c = 0; c < MM8_NC; c++) { Big t(1, (uint32_t)MM8_BIAS); big_add(bt, big_shl(t, 8 * c, 14)); } biasmod = big_divmod(bt, p, nullptr); }
    { Big sft(shift8, shift8 + 8); if (!big_ge(biasmod, sft)) big_add(biasmod, p); big_sub(biasmod, sft); }
    digits9(biasmod, sh->biasmod);
    {   // (0x80..80 * 2^261) mod p: mont_mul(row sum, .) = 0x80..80 * row sum
        Big c80(8, 0x80808080u);
        Big r261(9, 0); r261[8] = 1u << 5;
        digits9(big_divmod(big_mul(c80, r261), p, nullptr), sh->c80r);
    }
    if (hipMalloc(&sh->fold_dev, sizeof sh->fold_host) != hipSuccess) { delete sh; return fail(ctx, HB_ERR_HIP, "mm8: hipMalloc(fold table)"); }
    if (int rc = upload_table(ctx, sh->fold_dev, sh->fold_host, sizeof sh->fold_host, s)) { (void)hipFree(sh->fold_dev); delete sh; return rc; }
    ctx->mm8_shared = sh;
    *out = sh;
    return HB_OK;
}

void mm8_shared_free(hb_ctx *ctx) {
    Mm8Shared *sh = static_cast<Mm8Shared *>(ctx->mm8_shared);
    if (!sh) return;
    (void)hipFree(sh->fold_dev);
    delete sh;
    ctx->mm8_shared = nullptr;
}

// Build the int8 operand image of a raw small-entry matrix (hb_fast.hip tables: canonical digits of
// |M[i][l]| plus a per-row sign).  HB_ERR_UNSUPPORTED when the path does not apply: an entry of
// 2^126 or more, more than 32 terms, a modulus outside [2^254, 2^256), or tables that exceed the LDS budget.
int mm8_from_fast(hb_ctx *ctx, const FastMatrix *f, Mm8Matrix **out, hipStream_t s, const int32_t *rows, int n_rows) {
    *out = nullptr;
    if (env_hook(ENV_NO_MFMA)) return HB_ERR_UNSUPPORTED;
    if (ctx->n_limbs != 4 || f->n_in < 1 || f->n_in > 32 || f->n_out < 1) return HB_ERR_UNSUPPORTED;
    if ((ctx->p_limbs[3] >> 62) == 0) return HB_ERR_UNSUPPORTED;      // Barrett constants assume 2^254 <= p
    // rows != nullptr: the matrix made of rows[0 .. n_rows) of f (a compact check matrix)
    const int n_out = rows ? n_rows : f->n_out, d = f->n_in, nkb = (d + 7) / 8, n_rt = (n_out + 15) / 16;
    if (n_out < 1) return HB_ERR_UNSUPPORTED;
    if (mm8_lds_bytes(n_rt, nkb, (n_rt == 1) ? 4 : (n_rt == 2) ? 2 : 1) > MM8_LDS_LIMIT) return HB_ERR_UNSUPPORTED;
    const int tiles = (f->n_out + f->ot - 1) / f->ot;
    std::vector<uint32_t> Mh((size_t)tiles * d * f->ot * 9);
    std::vector<int32_t> neg((size_t)f->n_out, 0);
    HB_HIP(ctx, hipMemcpyAsync(Mh.data(), f->M, Mh.size() * 4, hipMemcpyDeviceToHost, s));
    if (f->negrow) HB_HIP(ctx, hipMemcpyAsync(neg.data(), f->negrow, neg.size() * 4, hipMemcpyDeviceToHost, s));
    HB_HIP(ctx, hipStreamSynchronize(s));

    Big p(8);
    for (int k = 0; k < 4; k++) { p[2 * k] = (uint32_t)ctx->p_limbs[k]; p[2 * k + 1] = (uint32_t)(ctx->p_limbs[k] >> 32); }
    Big c80(8, 0x80808080u);
    c80 = big_divmod(c80, p, nullptr);                                   // 0x80..80 mod p
    // what every row's constant takes out: the accumulator bias of the 47 columns, less what the fold of H adds (hb_mfma_wide.hip)
    Big biasmod;
    { Big bt(14, 0); for (int c = 0; c < MM8_NC; c++) { Big t(1, (uint32_t)MM8_BIAS); big_add(bt, big_shl(t, 8 * c, 14)); } biasmod = big_divmod(bt, p, nullptr); }
    std::vector<uint8_t> foldtab((size_t)MM8_FOLD_Q * 16, 0);
    BarrettParams bpar;
    {
        uint32_t shift8[8];
        if (!fold_tables(ctx, 16, foldtab.data(), bpar.c384, &bpar.mu, shift8)) return HB_ERR_UNSUPPORTED;
        Big sh(shift8, shift8 + 8);
        if (!big_ge(biasmod, sh)) big_add(biasmod, p);
        big_sub(biasmod, sh);                                            // (bias sum - shift) mod p
    }

    std::vector<int8_t> a((size_t)n_rt * nkb * 2 * 64 * 16, 0);
    bool skip01 = nkb >= 2 && !env_hook(ENV_MM8_NO_SKIP);
    std::vector<uint32_t> cr((size_t)n_rt * 16 * 16 + (size_t)MM8_FOLD_Q * 4, 0);
    memcpy(&cr[(size_t)n_rt * 16 * 16], foldtab.data(), foldtab.size());
    // rows beyond n_out: the bias pairs alone (their outputs are never stored)
    for (size_t i = 0; i < (size_t)n_rt * 16; i++) for (int k = 0; k < 8; k++) { cr[i * 16 + 2 * k] = 0x10100000u; cr[i * 16 + 2 * k + 1] = 0x1010u; }
    for (int i = 0; i < n_out; i++) {
        Big pos(5, 0), ngs(5, 0);   // sums of the positive / negated entries of the row (each < 32 * 2^127)
        int64_t colsum = 0;         // 128 * sum |digit| bounds every int32 column of this row
        const int src = rows ? rows[i] : i;
        if (src < 0 || src >= f->n_out) return HB_ERR_BAD_ARG;
        for (int l = 0; l < d; l++) {
            uint32_t dg[9], w8[8];
            for (int q = 0; q < 9; q++) dg[q] = Mh[mf_index(src, l, d, 9, q, f->ot)];
            pack<9, 8>(w8, dg);
            if (w8[4] | w8[5] | w8[6] | w8[7]) return HB_ERR_UNSUPPORTED;   // does not fit 16 balanced digits
            const int sgn = neg[src] ? -1 : 1;
            int carry = 0;
            // lane (r, g) of K-block kb = l / 8 holds terms 8 kb + 2 g and 8 kb + 2 g + 1: g = (l % 8) / 2, element e = l % 2
            const int lane_ = (4 * ((i % 16) % 4) + (i % 16) / 4) + 16 * ((l % 8) / 2), el = l & 1;
            for (int b = 0; b < 16; b++) {
                int t = sgn * (int)((w8[b >> 2] >> (8 * (b & 3))) & 0xffu) + carry;
                if (t > 127) { t -= 256; carry = 1; } else if (t < -128) { t += 256; carry = -1; } else carry = 0;
                // digit b = 7 + 8 G - 4 hi - bi  sits at byte j = 4 (2 hi + el) + bi of the lane's 16 bytes of group G
                const int grp = b >> 3, r7 = 7 - (b & 7), hi = r7 >> 2, bi = r7 & 3;
                a[((((size_t)(i / 16) * nkb + l / 8) * 2 + grp) * 64 + (size_t)lane_) * 16 + 4 * (2 * hi + el) + bi] = (int8_t)t;
                if (t != 0 && grp == 1 && l < 8) skip01 = false;
                colsum += (t < 0) ? -t : t;
            }
            if (carry) return HB_ERR_UNSUPPORTED;                           // |entry| >= 127 * 256^15 or so
            Big e(w8, w8 + 4);
            big_add(neg[src] ? ngs : pos, e);
        }
        if (colsum * 128 > MM8_BIAS) return HB_ERR_UNSUPPORTED;
        // corr = 0x80..80 * (pos - ngs) mod p  (the XOR-0x80 bias of the input bytes)
        Big cp = big_divmod(big_mul(c80, pos), p, nullptr), cn = big_divmod(big_mul(c80, ngs), p, nullptr);
        if (!big_ge(cp, cn)) big_add(cp, p);
        big_sub(cp, cn);
        if (!big_ge(cp, biasmod)) big_add(cp, p);
        big_sub(cp, biasmod);                                               // the row's constant, in [0, p)
        for (int k = 0; k < 8; k++) {                                       // per word one 64-bit addend: the bias of the fold's four columns (2^20 each) + the word
            const uint64_t pair = (uint64_t)cp[k] + ((0x1010ull << 32) | 0x10100000ull);
            cr[(size_t)i * 16 + 2 * k] = (uint32_t)pair;
            cr[(size_t)i * 16 + 2 * k + 1] = (uint32_t)(pair >> 32);
        }
    }
    Mm8Matrix *m = new Mm8Matrix();
    m->n_out = n_out; m->d = d; m->nkb = nkb; m->n_rt = n_rt; m->skip01 = skip01; m->a8 = nullptr; m->crow = nullptr; m->zero = nullptr;
    m->bp = bpar;
    {   // 2^256 - p, 32-bit words
        Big pn(9, 0); pn[8] = 1;
        Big pw9(p); pw9.push_back(0);
        big_sub(pn, pw9);
        for (int k = 0; k < 8; k++) m->bp.pneg[k] = pn[k];
    }
    HB_HIP(ctx, hipMalloc(&m->a8, a.size()));
    HB_HIP(ctx, hipMalloc(&m->crow, cr.size() * 4));
    HB_HIP(ctx, hipMalloc(&m->zero, 64));
    {
        std::vector<uint8_t> zero64(64, 0);
        int rc = upload_table(ctx, m->a8, a.data(), a.size(), s);
        if (!rc) rc = upload_table(ctx, m->crow, cr.data(), cr.size() * 4, s);
        if (!rc) rc = upload_table(ctx, m->zero, zero64.data(), 64, s);
        if (rc) { mm8_free(m); return rc; }
    }
    *out = m;
    return HB_OK;
}

// out(c, i) = sum_l M[i][l] * in(c, rows[l]) mod p, canonical; CHECK mode when check_mask_dev != nullptr.
// CHECK mode can also hand rows < copy_rows of the input to copy_dst (view cpv, clipped at copy_count).
int launch_mm8(hb_ctx *ctx, const Mm8Matrix *m, const uint32_t *in, hb_view iv, const int32_t *in_rows_dev, int64_t in_count,
               uint32_t *out, hb_view ov, int64_t out_count, const int32_t *check_mask_dev, int32_t *mismatch_dev,
               int64_t C, hipStream_t s, uint32_t *copy_dst, hb_view cpv, int64_t copy_count, int copy_rows,
               const int32_t *check_rows_dev) {
    if (C <= 0) return HB_OK;
    const int tpw = mm8_tpw(m->n_rt, m->nkb);
    const int64_t n_tiles = (C + 15) / 16;
    const int64_t n_units = (n_tiles + tpw - 1) / tpw;
    int64_t blocks = 2 * (int64_t)mm8_num_cus();
    if (blocks > n_units) blocks = n_units;
    // no more workgroups than the launch's length needs: with the slowest workgroup at `rounds` units, ceil(n_units / rounds) of them do --
    // config 3's encode is 497 of 512 -- and the slots left free are where a decoder's small launches on other streams (the builder of the
    // next decode, a symbol fetch, the probe) run beside this one instead of behind it
    blocks = mm8_trimmed_grid(n_units, blocks);
    const size_t lds = mm8_lds_bytes(m->n_rt, m->nkb, tpw);
    const bool check = check_mask_dev != nullptr;
    // the last row tile holds at most 8 rows: its outputs 2 and 3 are padding and their reduction can be skipped
    const bool ragged = !check && (m->n_out % 16) >= 1 && (m->n_out % 16) <= 8;
#define MM8_LAUNCH_(NKB, CHK, RG, SK)                                                                                 \
    do {                                                                                                          \
        static std::atomic<unsigned long long> attr_done{0};   /* one bit per device: the attribute is per device */                                                                            \
        if (!((attr_done.load() >> (ctx->device & 63)) & 1ull)) {                                                                                         \
            HB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mm8<NKB, CHK, RG, SK>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)); \
            attr_done.fetch_or(1ull << (ctx->device & 63));                                                                                     \
        }                                                                                                         \
        hipLaunchKernelGGL((k_mm8<NKB, CHK, RG, SK>), dim3((unsigned)blocks), dim3(256), lds, s, m->a8, m->crow, m->zero, in, iv.stride_c, \
                           iv.stride_l, in_rows_dev, in_count, m->d, out, ov.stride_c, ov.stride_l, out_count,   \
                           check_mask_dev, check_rows_dev, mismatch_dev, copy_dst, cpv.stride_c, cpv.stride_l, copy_count, copy_rows,         \
                           m->n_out, m->n_rt, tpw, C, n_units, m->bp);             \
    } while (0)
#define MM8_LAUNCH(NKB) do { if (check) MM8_LAUNCH_(NKB, true, false, false); else if (ragged) MM8_LAUNCH_(NKB, false, true, false); else MM8_LAUNCH_(NKB, false, false, false); } while (0)
#define MM8_LAUNCH_SK(NKB) do { if (check) MM8_LAUNCH_(NKB, true, false, true); else if (ragged) MM8_LAUNCH_(NKB, false, true, false); else MM8_LAUNCH_(NKB, false, false, true); } while (0)
    switch (m->nkb) {
        case 1: MM8_LAUNCH(1); break;
        case 2: if (m->skip01) MM8_LAUNCH_SK(2); else MM8_LAUNCH(2); break;
        case 3: if (m->skip01) MM8_LAUNCH_SK(3); else MM8_LAUNCH(3); break;
        case 4: if (m->skip01) MM8_LAUNCH_SK(4); else MM8_LAUNCH(4); break;
        default: return fail(ctx, HB_ERR_UNSUPPORTED, "mm8: more than 32 terms");
    }
#undef MM8_LAUNCH
#undef MM8_LAUNCH_SK
#undef MM8_LAUNCH_
    HB_LAUNCH_CHECK(ctx);
    return HB_OK;
}

}  // namespace hb

// ---- diagnostic entry points (scratch/test_mm8.py): the matrix-core mat-vec on its own ----------
extern "C" int hb_debug_mm8_create(hb_ctx *ctx, const uint64_t *x_host, int n, int d, void **out) { HB_API_GUARD(ctx);
    uint32_t *xd = nullptr;
    int rc = upload_elems(ctx, x_host, (size_t)n, &xd, 0);
    if (rc) return rc;
    FastMatrix *V = nullptr;
    rc = fast_vand_create(ctx, xd, n, d, &V, 0);
    (void)hipFree(xd);
    if (rc) return rc;
    Mm8Matrix *m = nullptr;
    rc = mm8_from_fast(ctx, V, &m, 0);
    fast_matrix_free(V);
    *out = m;
    return rc;
}
extern "C" int hb_debug_mm8_apply(hb_ctx *ctx, void *mat, const void *in_dev, int64_t in_sc, int64_t in_sl, int64_t in_count,
                                  void *out_dev, int64_t out_sc, int64_t out_sl, int64_t out_count, int64_t n_chunks,
                                  const int32_t *check_mask_dev, int32_t *mismatch_dev) { HB_API_GUARD(ctx);
    hb_view iv{in_sc, in_sl}, ov{out_sc, out_sl};
    return launch_mm8(ctx, (const Mm8Matrix *)mat, (const uint32_t *)in_dev, iv, nullptr, in_count, (uint32_t *)out_dev, ov, out_count,
                      check_mask_dev, mismatch_dev, n_chunks, 0, nullptr, hb_view{0, 0}, 0, 0, nullptr);
}

#ifdef HB_MM8_TIMING
extern "C" int hb_debug_mm8_timing(unsigned long long *out, int count) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(hb::g_mm8_t), sizeof(unsigned long long) * (size_t)count) == hipSuccess ? 0 : 1;
}
#endif
