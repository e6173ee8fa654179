// hb_mfma_fused.hip -- decode + validate at small-integer points as ONE launch of the small-entry matrix-core kernel.
//
// Reference: IncrementalDecoder's optimistic step (honeybadgermpc/reed_solomon.py:305-326): decoder.decode_batch over the first
// d arrivals z (vandermonde_batch_interpolate, hbmpc_ntl_helpers.pyx:139-197: V(z)^-1 rebuilt per call, then a mat-mul),
// encoder.encode_batch of the guess, and the compare loop over the later arrivals zc.
//
// Round 2 fused the three into one launch of the FULL-SIZE kernel (hb_mfma_wide.hip) over [V^-1 ; V[zc] V^-1], whose entries are
// arbitrary residues: 32 digits an entry, 63 columns an output.  At the production points x_i = i + 1 that throws a structure
// away.  With A_j(X) = prod_{q != j} (X - x_zq) and den_j = A_j(x_zj),
//
//     V^-1[m][j]        = coeff_m(A_j) / den_j           coeff_m(A_j): an INTEGER, |.| <= prod (1 + x_zq) < 2^122 at n = 64, t = 21
//     (V[zc] V^-1)[i][j] = A_j(x_zci) / den_j            A_j(x_i) = prod_{q != j} (x_i - x_zq): an INTEGER, |.| < 2^124
//
// so  [V^-1 ; V[zc] V^-1] y  =  [N ; P] (y ./ den)  with a matrix of SMALL integers (16 balanced base-256 digits: half the MFMAs
// of the full-size kernel, 47 columns instead of 63 per output) applied to the received columns divided by den_j -- one modular
// multiplication per INPUT element instead of per matrix entry.  Round 1's pipeline already knew this factorisation for the
// decode alone (k_prescale_tab + k_mm8 over N, hb_fast.hip / hb_mfma.hip), as three launches with the scaled columns making a
// round trip through HBM.  Here:
//
//   * k_mm8f: the k_mm8 pass (gen_mm8.py's MFMA phases, the same epilogue) in workgroups of EIGHT waves over units of 64 chunks.
//     The input elements do not arrive by LDS-DMA: the waves that have a pass less than the others (43 rows = 3 row tiles: 12
//     passes for 8 waves) read the next unit's elements, multiply them by 1 / den_l (x / den = sum_q x_q T_q over the 29-bit
//     digits of x with T_q = 2^(29 q) / den_l tabulated -- wave-uniform, scalar loads -- and a two-digit Barrett quotient: the
//     arithmetic of k_prescale_tab) and write them to LDS in MFMA operand order.  The GEMM takes ANY 256-bit representative, so
//     the scaled element is not even made canonical.  Per row a mode: store the canonical sum (a coefficient row) or compare it
//     with the received row of a later arrival (flag, first disagreeing chunk, bitmap of disagreeing chunks).
//   * k_fs_build: [N ; P], the per-row constants, the T tables and the row modes ON THE DEVICE, in plain 128-bit integer
//     arithmetic (the entries are small integers, so A(X), the synthetic divisions and the products are a microsecond of shifts
//     and multiplies); the only field arithmetic is 1 / den_j = prod_q 1 / (x_zj - x_zq) from the point set's table of inverse
//     differences (hb_quick.hip) and one multiplication per row constant.  Two halves that can be launched apart: everything
//     that depends on z alone (N, T: known when the d-th column arrives) and the rows of the compared senders (P: known when the
//     last column arrives).
//
// Applies to wide contexts with 2^254 <= p < 2^256, points that are distinct integers below 2^16, 4 <= d <= 22, and bounds
// (checked on the host from the point set) that keep every entry below 2^125; everything else stays on the full-size kernel.
#include <math.h>

#include <algorithm>
#include <atomic>

#include "hb_mm8.hpp"

namespace hb {

#include "hb_mm8_body.inc"

constexpr int FS_WAVES = 8, FS_TPW = 4;
#ifdef HB_MM8_TIMING
__device__ unsigned long long g_fs_t[256 * 8 * 8];
#define FS_T(k) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); tacc[k] += tn_ - tlast; tlast = tn_; } while (0)
#else
#define FS_T(k) do { } while (0)
#endif
constexpr int FS_MAXD = 24, FS_MAXC = 104;      // terms (three K-blocks), compared rows
constexpr size_t FS_LDS_LIMIT = 156 * 1024;

struct FsIdx { uint16_t z[FS_MAXD], xz[FS_MAXD], zc[FS_MAXC], xzc[FS_MAXC]; };
// the compared senders of a launch whose rows come from the per-party candidate store (k_fs_cand): rows n_coef .. n_coef + nc - 1 of the
// matrix are the candidates of parties zc[0 .. nc)
struct FsPick { const uint4 *cand; const uint32_t *cand_crow; int n_coef, nc; uint16_t zc[FS_MAXC]; };
// which waves scale which terms of the NEXT unit while a unit's passes run (bit l of terms[w]: wave w takes term l): dealt on the host so
// that the two waves of a SIMD (w and w + 4) reach the unit's barrier together -- a wave that waits leaves its SIMD to ONE wave, and one
// wave issues at 0.7 of the rate two reach (fs_split)
struct FsSplit { uint32_t terms[FS_WAVES]; };
constexpr int FS_CAND_MAXN = 128;               // parties a candidate store is built for (n d 128-bit entries in the builder's LDS)

// ---------------------------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------------------------
// LDS: row constants [n_rt * 16][16] | matrix digits [n_rt][NKB][2][64] x 16 B | two element buffers [4 tiles][NKB][2][2][64] x 16 B |
// rowl[32] | maskl[16 n_rt] | fold table
__host__ __device__ inline size_t fs_lds_bytes(int n_rt, int nkb) {
    return ((size_t)n_rt * 64 + (size_t)n_rt * nkb * 2 * 64 + (size_t)2 * FS_TPW * nkb * 4 * 64 + MM8_FOLD_Q) * 16 + (size_t)(32 + 16 * n_rt) * 4;
}

// rowmode[i]: 0 = nothing, v > 0: compare the sum with row v - 1 of the input buffer, v < 0: store it as output row -v - 1
template <int NKB>
__global__ __launch_bounds__(64 * FS_WAVES, 2) void k_mm8f(const int4 *__restrict__ a8, const uint32_t *__restrict__ crowd, const v4i *__restrict__ foldg,
                                                          const uint32_t *__restrict__ KT, const PrescaleParams PP,
                                                          const uint32_t *__restrict__ in_pk, int64_t in_sc, int64_t in_sl, const int32_t *__restrict__ in_rows,
                                                          int64_t in_count, int d, const int32_t *__restrict__ rowmode,
                                                          uint32_t *__restrict__ out_pk, int64_t out_sc, int64_t out_sl, int64_t out_count,
                                                          int32_t *__restrict__ mismatch, int32_t *__restrict__ first_bad, uint32_t *__restrict__ bad_map,
                                                          int n_out, int n_rt, int64_t n_chunks, int64_t n_units, BarrettParams bp, const FsDone done, const FsPick pick,
                                                          const FsSplit split) {
    constexpr int NT = 64 * FS_WAVES, NL = 9, NW = 8;
    extern __shared__ uint4 fs_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    uint32_t *crl = reinterpret_cast<uint32_t *>(fs_lds);
    int4 *abuf = reinterpret_cast<int4 *>(fs_lds + n_rt * 64);
    uint4 *xbuf = fs_lds + n_rt * 64 + n_rt * NKB * 2 * 64;
    constexpr int bufsz = FS_TPW * NKB * 4 * 64;
    int32_t *rowl = reinterpret_cast<int32_t *>(xbuf + 2 * bufsz);
    int32_t *maskl = rowl + 32;
    v4i *foldl = reinterpret_cast<v4i *>(maskl + 16 * n_rt);

    // ---- prologue: tables, zeroed element buffers (terms d .. 8 NKB - 1 are never written again) ------------------------
    if (threadIdx.x < 32) {
        const int lc = (int)threadIdx.x < d ? (int)threadIdx.x : d - 1;
        rowl[threadIdx.x] = in_rows ? in_rows[lc] : lc;
    }
    // the rows of the compared senders were built per PARTY when the first d arrivals were known (k_fs_cand); which of them this launch
    // compares, and in which row, is only known now: their digit pieces and row constants are FETCHED here, ahead of the tables' copy (two
    // dependent loads, the sender's number and then its row: they ride under the copy), and written over the image's empty rows behind it
    uint4 pk_piece = make_uint4(0, 0, 0, 0), pk_crow = make_uint4(0, 0, 0, 0);
    const int pk_e = threadIdx.x;
    const bool pk_has_piece = pick.cand && pk_e < pick.nc * NKB * 8, pk_has_crow = pick.cand && pk_e < pick.nc * 4;
    // (one piece a thread ahead of the copy: config 3's 21 compared rows are 504 pieces for the 512 threads; what a larger launch has beyond
    // that is fetched behind the copy, below -- five pieces a thread in registers went to scratch memory)
    if (pk_has_piece) {
        const int j = pk_e / (NKB * 8), pc = pk_e - j * (NKB * 8);
        pk_piece = pick.cand[(size_t)pick.zc[j] * (NKB * 8) + pc];
    }
    if (pk_has_crow) pk_crow = reinterpret_cast<const uint4 *>(pick.cand_crow)[(size_t)pick.zc[pk_e >> 2] * 4 + (pk_e & 3)];
    for (int i = threadIdx.x; i < n_rt * 16; i += NT) maskl[i] = i < n_out ? rowmode[i] : 0;
    for (int i = threadIdx.x; i < n_rt * 64; i += NT) fs_lds[i] = reinterpret_cast<const uint4 *>(crowd)[i];
    for (int i = threadIdx.x; i < n_rt * NKB * 2 * 64; i += NT) abuf[i] = a8[i];
    for (int i = threadIdx.x; i < 2 * bufsz; i += NT) xbuf[i] = make_uint4(0, 0, 0, 0);
    if (threadIdx.x < MM8_FOLD_Q) foldl[threadIdx.x] = foldg[threadIdx.x];
    if (pick.cand) {
        __syncthreads();
        auto put = [&](int e, const uint4 &v) {
            const int j = e / (NKB * 8), pc = e - j * (NKB * 8);
            const int ri = pick.n_coef + j, rt = ri >> 4, r16 = ri & 15, r = 4 * (r16 & 3) + (r16 >> 2);
            const int kb = pc >> 3, grp = (pc >> 2) & 1, gg = pc & 3;
            reinterpret_cast<uint4 *>(abuf)[((rt * NKB + kb) * 2 + grp) * 64 + r + 16 * gg] = v;
        };
        if (pk_has_piece) put(pk_e, pk_piece);
        for (int e = pk_e + NT; e < pick.nc * NKB * 8; e += NT) {
            const int j = e / (NKB * 8), pc = e - j * (NKB * 8);
            put(e, pick.cand[(size_t)pick.zc[j] * (NKB * 8) + pc]);
        }
        if (pk_has_crow) fs_lds[(pick.n_coef + (pk_e >> 2)) * 4 + (pk_e & 3)] = pk_crow;
        for (int j = threadIdx.x; j < pick.nc; j += NT) maskl[pick.n_coef + j] = (int32_t)pick.zc[j] + 1;
    }
    __syncthreads();

    // (tile, row tile) pairs of a unit: wave w takes w, w + 8, ...; which terms of the next unit a wave scales beside its passes: split.terms
    const int n_pairs = FS_TPW * n_rt;
    const uint32_t my_terms = (uint32_t)__builtin_amdgcn_readfirstlane((int)split.terms[wave]);

    // one task = term l of the unit's 64 chunks: lane = chunk (tile lane >> 4, column lane & 15); T_q of this term wave-uniform
    auto scale_unit = [&](int64_t unit, uint4 *dst, uint32_t terms) {
        int64_t chunk = unit * 64 + lane;
        if (chunk >= n_chunks) chunk = n_chunks - 1;              // (results of padding chunks are never stored or compared)
        const int t = lane >> 4;
        uint32_t xw[NW];
        bool ok = false;
        auto fetch = [&](int l, uint32_t (&w)[NW], bool &okk) {
            const int64_t idx = chunk * in_sc + (int64_t)rowl[l] * in_sl;
            okk = idx < in_count;
            load_words<NW>(w, in_pk + (okk ? idx : 0) * NW);
        };
        terms &= d < 32 ? (1u << d) - 1u : ~0u;
        int l = terms ? __builtin_ctz(terms) : d;
        if (l < d) fetch(l, xw, ok);
        while (l < d) {
            terms &= terms - 1u;
            const int ln = terms ? __builtin_ctz(terms) : d;
            uint32_t xn[NW];
            bool okn = false;
            if (ln < d) fetch(ln, xn, okn);
            else {
#pragma unroll
                for (int k = 0; k < NW; k++) xn[k] = 0;
            }
            const uint32_t *__restrict__ T = KT + (size_t)l * (NL * NL);
            uint32_t xd[NL];
            unpack<NL, NW>(xd, xw);
            uint64_t col[NL + 1];
#pragma unroll
            for (int j = 0; j <= NL; j++) col[j] = 0;
#pragma unroll
            for (int q = 0; q < NL; q++)
#pragma unroll
                for (int j = 0; j < NL; j++) col[j] += (uint64_t)xd[q] * T[q * NL + j];
            uint32_t v[NL + 1];
#pragma unroll
            for (int k = 0; k < NL; k++) { v[k] = (uint32_t)col[k] & DMASK; col[k + 1] += col[k] >> LB; }
            v[NL] = (uint32_t)col[NL];
            const uint64_t mid = (uint64_t)v[9] * PP.m0 + (uint64_t)v[8] * PP.m1 + (((uint64_t)v[8] * PP.m0) >> LB);
            const uint64_t qh = (uint64_t)v[9] * PP.m1 + (mid >> LB);
            const uint32_t q0 = (uint32_t)qh & DMASK, q1 = (uint32_t)(qh >> LB);
            uint64_t dc[NL];
#pragma unroll
            for (int k = 0; k < NL; k++) {
                dc[k] = v[k] + (uint64_t)q0 * PP.pbar[k];
                if (k > 0) dc[k] += (uint64_t)q1 * PP.pbar[k - 1];
            }
            uint32_t r[NL];
#pragma unroll
            for (int k = 0; k < NL; k++) {
                r[k] = (uint32_t)dc[k] & DMASK;
                if (k < NL - 1) dc[k + 1] += dc[k] >> LB;
            }
            uint32_t w[NW];
            pack<NL, NW>(w, r);
            if (PP.pneg[NW - 1] >> 31) {
                // p < 2^255: r < 2p < 2^256 is a 256-bit representative as it is, and the GEMM takes any
#pragma unroll
                for (int k = 0; k < NW; k++) w[k] = ok ? w[k] : 0u;
            } else {
                // r < 2p may pass 2^256, where the packed words would drop bit 256: made canonical
                uint32_t u[NW];
                unsigned cy = 0;
#pragma unroll
                for (int k = 0; k < NW; k++) u[k] = __builtin_addc(w[k], PP.pneg[k], cy, &cy);
                const bool take = cy || (r[NL - 1] >> 24);
#pragma unroll
                for (int k = 0; k < NW; k++) w[k] = ok ? (take ? u[k] : w[k]) : 0u;
            }
            const int kb = l >> 3, gg = (l & 7) >> 1, e = l & 1;
            uint4 *slot = dst + (size_t)(((t * NKB + kb) * 2 + e) * 2) * 64 + (n + 16 * gg);
            slot[0] = make_uint4(w[0], w[1], w[2], w[3]);
            slot[64] = make_uint4(w[4], w[5], w[6], w[7]);
#pragma unroll
            for (int k = 0; k < NW; k++) xw[k] = xn[k];
            ok = okn;
            l = ln;
        }
    };

    const v4i biasv = v4i{MM8_BIAS, MM8_BIAS, MM8_BIAS, MM8_BIAS};
    uint32_t k256 = 256u, k16m = 1u << 24;      // opaque, so that the word assembly stays two v_mad_u64_u32 per word
    int32_t s1 = 1, s256 = 256, s64k = 1 << 16, s16m = 1 << 24;   // and the gathering of the fold's columns one v_mad_i64_i32 each
    uint32_t u1 = 1u;
    asm volatile("" : "+s"(k256), "+s"(k16m), "+s"(s1), "+s"(s256), "+s"(s64k), "+s"(s16m), "+s"(u1));
    // the fold's A operand: lane (m, g') of the diagonal block g' = m / 4 reads its 16 digits, every other lane the row's zero bytes
    const v4i *fold_lane = reinterpret_cast<const v4i *>(reinterpret_cast<const char *>(foldl) + (g == (n >> 2) ? 16 * n : 256));

    int buf = 0;
    int64_t unit = blockIdx.x;
#ifdef HB_MM8_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#endif
    if (unit < n_units) scale_unit(unit, xbuf, 0x01010101u << wave);  // the first unit: every wave takes its share
    FS_T(0);
    for (; unit < n_units; unit += gridDim.x, buf ^= 1) {
        // every wave's share of this unit's elements is in LDS, and nobody reads the other buffer any more
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        FS_T(7);
        __builtin_amdgcn_s_barrier();
        FS_T(1);
        if (my_terms && unit + gridDim.x < n_units) scale_unit(unit + gridDim.x, xbuf + (size_t)(buf ^ 1) * bufsz, my_terms);
        FS_T(2);
        for (int pidx = wave; pidx < n_pairs; pidx += FS_WAVES) {
            // two row tiles: the waves of one SIMD (w and w + 4) get one of each, so the ragged second tile's shorter passes spread evenly
            const int tl = n_rt == 2 ? (pidx >> 1) : pidx / n_rt;
            const int rt = n_rt == 2 ? ((pidx ^ (pidx >> 2)) & 1) : pidx - tl * n_rt;
            const int64_t chunk = (unit * FS_TPW + tl) * 16 + n;
            const uint4 *xs = xbuf + (size_t)buf * bufsz + (size_t)tl * NKB * 4 * 64 + lane;
            const int4 *as = abuf + (size_t)rt * NKB * 2 * 64 + lane;
            uint32_t eap[4][11], c0p[4];     // half 0's column pairs, parked across the second MFMA block
            // output `reg` of lane (n, g) is row 16 rt + 4 reg + g: a ragged last tile fills its outputs from the top
            const int rows_here = n_out - 16 * rt < 16 ? n_out - 16 * rt : 16;
            const int nreg = (rows_here + 3) >> 2;
            const uint32_t xs_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)xs;
            const uint32_t as_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)as;
            {   // ---- half 0: c = 0 and the pairs (4j+3, 4j+4)
                v4i acc[24];
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(0);        // (the phases yield to reductions and scaling: hb_mfma.hip, k_mm8)
                Mm8Phase<NKB, 0, false>::run(acc, xs_addr, as_addr, biasv);
                __builtin_amdgcn_s_setprio(2);
                __builtin_amdgcn_sched_barrier(0);
                FS_T(3);
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    if (reg >= 2 && reg >= nreg) break;
                    c0p[reg] = (uint32_t)acc[0][reg];
#pragma unroll
                    for (int j = 0; j < 11; j++) eap[reg][j] = (uint32_t)acc[1 + 2 * j][reg] + ((uint32_t)acc[2 + 2 * j][reg] << 8);
                    asm volatile("" ::"v"(c0p[reg]), "v"(eap[reg][0]), "v"(eap[reg][1]), "v"(eap[reg][2]), "v"(eap[reg][3]), "v"(eap[reg][4]),
                                 "v"(eap[reg][5]), "v"(eap[reg][6]), "v"(eap[reg][7]), "v"(eap[reg][8]), "v"(eap[reg][9]), "v"(eap[reg][10]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            {   // ---- half 1: the pairs (4k+1, 4k+2)
                v4i acc[24];
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                FS_T(4);
                __builtin_amdgcn_s_setprio(0);
                Mm8Phase<NKB, 1, false>::run(acc, xs_addr, as_addr, biasv);
                __builtin_amdgcn_s_setprio(2);
                __builtin_amdgcn_sched_barrier(0);
                FS_T(5);
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    if (reg >= 2 && reg >= nreg) break;
                    const int i = 16 * rt + 4 * reg + g;
                    const int md = maskl[i];
                    const bool live = chunk < n_chunks;
                    const bool cmp = live && md > 0, st = live && md < 0;
                    uint32_t ew[8];
                    // unconditional (from the buffer's first element when there is nothing to compare): a load under `cmp` makes hipcc wrap
                    // the whole reduction in that divergent branch, and the register copies at its join spill
                    load_words<8>(ew, in_pk + (cmp ? (chunk * in_sc + (int64_t)(md - 1) * in_sl) * 8 : 0));
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
                    v4i hb;
#pragma unroll
                    for (int k = 0; k < 4; k++) hb[k] = (int)(hw[k] ^ 0x80808080u);
                    v4i dcol[8];
#pragma unroll
                    for (int eb = 0; eb < 8; eb++)
                        dcol[eb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fold_lane[eb * (MM8_FOLD_ROW / 16)], hb, v4i{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < 8; k++) pw[k] += (uint64_t)hw[4] * bp.c384[k];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        int64_t tt = (int64_t)pw[k] + (int64_t)dcol[k][0] * s1;
                        tt += (int64_t)dcol[k][1] * s256;
                        tt += (int64_t)dcol[k][2] * s64k;
                        tt += (int64_t)dcol[k][3] * s16m;
                        pw[k] = (uint64_t)tt;
                    }
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
                    if (__builtin_amdgcn_ballot_w64(top != 0 || ow[7] >= ~bp.pneg[7]) != 0) {
                        uint32_t u[8];
                        unsigned cy2 = 0;
#pragma unroll
                        for (int k = 0; k < 8; k++) u[k] = __builtin_addc(ow[k], bp.pneg[k], cy2, &cy2);
                        const bool take = cy2 || top;
#pragma unroll
                        for (int k = 0; k < 8; k++) ow[k] = take ? u[k] : ow[k];
                    }
                    uint32_t diff = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) diff |= ew[k] ^ ow[k];
                    // keep the reduction outside the exec masks of the compare and the store (hipcc otherwise wraps the whole output in a
                    // divergent branch, and the register copies at its join spill)
                    asm volatile("" : "+v"(diff), "+v"(ow[0]), "+v"(ow[1]), "+v"(ow[2]), "+v"(ow[3]), "+v"(ow[4]), "+v"(ow[5]), "+v"(ow[6]), "+v"(ow[7]));
                    // disagreements are reported once per wave and output (an adversary that corrupts everything would otherwise queue a
                    // million atomics on one word: 430 us instead of 55 for config 3's launch): lane (n, g) holds chunk tile_base + n, so the
                    // sixteen chunk bits of the wave are the OR of its four lane groups
                    const unsigned long long bad_lanes = __builtin_amdgcn_ballot_w64(cmp && diff);
                    if (bad_lanes) {
                        const uint32_t bits16 = (uint32_t)((bad_lanes | (bad_lanes >> 16) | (bad_lanes >> 32) | (bad_lanes >> 48)) & 0xffffull);
                        if (lane == 0) {
                            const int64_t tile_base = chunk - n;
                            const int32_t cmin = (int32_t)tile_base + __builtin_ctz(bits16);
                            if (*reinterpret_cast<volatile int32_t *>(mismatch) == 0) atomicOr(mismatch, 1);
                            if (first_bad && *reinterpret_cast<volatile int32_t *>(first_bad) > cmin) atomicMin(first_bad, cmin);
                            if (bad_map) atomicOr(bad_map + (tile_base >> 5), bits16 << (tile_base & 31));
                        }
                    }
                    const int64_t oidx = chunk * out_sc + (int64_t)(-md - 1) * out_sl;
                    if (st && oidx < out_count) store_words<8>(out_pk + oidx * 8, ow);
                    if (reg & 1) __builtin_amdgcn_sched_barrier(0);   // two reductions at a time: ILP for the carry chains
                }
            }
            FS_T(6);
        }
    }
#ifdef HB_MM8_TIMING
    if (lane == 0 && blockIdx.x < 256) for (int k = 0; k < 8; k++) g_fs_t[(blockIdx.x * 8 + wave) * 8 + k] = tacc[k];
#endif
    // a caller that waits for the verdict: the last workgroup to finish hands the status words to pinned host memory (and resets
    // them for the next launch), the sequence number last -- the host polls that word instead of synchronising the stream
    fs_workgroup_done(done, mismatch, first_bad);
}

// ---------------------------------------------------------------------------------------------------------------------
// the builder: [N ; P] as int8 digits, row constants, T tables, row modes
// ---------------------------------------------------------------------------------------------------------------------
typedef __int128 i128;

struct FsConsts { uint32_t c80r[9], biasmod[9]; };
constexpr int32_t FS_OVERFLOW = 0x40000000;       // OR-ed into the status word if an entry does not fit (the host's bound rules it out)

// byte address of digit b (< 16) of entry (row i, term l) in the image (the layout of mm8_from_fast, hb_mfma.hip)
__device__ __forceinline__ size_t fs_digit_addr(int i, int l, int b, int nkb) {
    const int lane_ = (4 * ((i % 16) % 4) + (i % 16) / 4) + 16 * ((l % 8) / 2), el = l & 1;
    const int grp = b >> 3, r7 = 7 - (b & 7), hi = r7 >> 2, bi = r7 & 3;
    return ((((size_t)(i / 16) * nkb + l / 8) * 2 + grp) * 64 + (size_t)lane_) * 16 + 4 * (2 * hi + el) + bi;
}

// the constant of one matrix row from its d entries: (0x80..80 * sum - bias sum) mod p as eight pairs [bias of the fold's columns + word]
__device__ __forceinline__ void fs_row_constant(const FpParams<9> &P, const FsConsts &cs, const i128 *row, int d, uint32_t *dst16) {
    constexpr int NL = 9, NW = 8;
    uint32_t s5[5] = {0, 0, 0, 0, 0};                   // the row sum in 160 bits two's complement
    for (int l = 0; l < d; l++) {
        const i128 v = row[l];
        const uint32_t vw[5] = {(uint32_t)v, (uint32_t)(v >> 32), (uint32_t)(v >> 64), (uint32_t)(v >> 96), (uint32_t)(v >> 127 >> 1)};
        unsigned cy = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) s5[k] = __builtin_addc(s5[k], vw[k], cy, &cy);
    }
    const bool negs = (s5[4] >> 31) != 0;
    if (negs) {
        unsigned cy = 1;
#pragma unroll
        for (int k = 0; k < 5; k++) s5[k] = __builtin_addc(~s5[k], 0u, cy, &cy);
    }
    uint32_t w8[NW] = {s5[0], s5[1], s5[2], s5[3], s5[4], 0, 0, 0}, fe[NL], prod[NL], corr[NL], k80[NL], bm[NL];
    unpack<NL, NW>(fe, w8);
    if (negs) fp_neg(fe, fe, P);
#pragma unroll
    for (int k = 0; k < NL; k++) { k80[k] = cs.c80r[k]; bm[k] = cs.biasmod[k]; }
    mont_mul(prod, fe, k80, P);
    fp_sub(corr, prod, bm, P);
    uint32_t cw[NW];
    pack<NL, NW>(cw, corr);
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint64_t pair = (uint64_t)cw[k] + ((0x1010ull << 32) | 0x10100000ull);
        dst16[2 * k] = (uint32_t)pair;
        dst16[2 * k + 1] = (uint32_t)(pair >> 32);
    }
}

// One workgroup of 1024.  Roles of the first phase:
//   wave 0        A(X) = prod_q (X - x_zq) over the integers (coefficient t on lane t), then per arrival j the coefficients of
//                 A_j = A / (X - x_zj) by synthetic division                                                    [Z]
//   waves 1, 2    1 / den_j = prod_{q != j} 1 / (x_zj - x_zq): four lanes per arrival, a quarter of the factors each      [Z]
//   waves 3 ..    P[i][j] = prod_{q != j} (x_zci - x_zq), one thread per entry                                         [ZC]
// Second phase: a thread per entry writes its 16 balanced digits; a thread per row its constant; 9 d threads the T tables.
__device__ __forceinline__ void fs_build_body(uint8_t *fb_lds, const FpParams<9> &P, const uint32_t *__restrict__ inv, int n, const FsIdx &ix, int d, int nc, int n_coef,
                                              int flags, const FsConsts &cs, uint8_t *__restrict__ a8, uint32_t *__restrict__ crow, uint32_t *__restrict__ KT,
                                              int32_t *__restrict__ rowmode, int32_t *__restrict__ z_dev, int32_t *__restrict__ status) {
    constexpr int NL = 9, NW = 8;
    const int n_out = n_coef + nc, nkb = (d + 7) / 8, n_rt = (n_out + 15) / 16;
    i128 *ent = reinterpret_cast<i128 *>(fb_lds);                       // [n_out][d]
    i128 *Ac = ent + (size_t)n_out * d;                                 // [d + 1]
    uint32_t *wj = reinterpret_cast<uint32_t *>(Ac + (d + 1));          // [d][NL] Montgomery
    uint32_t *part = wj + (size_t)d * NL;                               // [d][4][NL]
    const int tid = threadIdx.x;
    const bool do_z = flags & FS_BUILD_Z, do_zc = flags & FS_BUILD_ZC;
    if (do_z) {
        // the image and the row constants of ALL row tiles start from "no row": zero digits, the bias pairs alone
        const size_t img = (size_t)n_rt * nkb * 2 * 64 * 16;
        for (size_t i = tid; i < img / 16; i += 1024) reinterpret_cast<uint4 *>(a8)[i] = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < n_rt * 16 * 8; i += 1024) { crow[2 * i] = 0x10100000u; crow[2 * i + 1] = 0x1010u; }
        for (int i = tid; i < n_rt * 16; i += 1024) rowmode[i] = i < n_coef ? -(i + 1) : 0;
        if (tid < d) z_dev[tid] = ix.z[tid];
    }
    if (do_z && tid < 64) {
        if (tid <= d) Ac[tid] = tid == 0 ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int q = 0; q < d; q++) {
            i128 nv = 0;
            const bool act = tid <= q + 1 && tid <= d;
            if (act) nv = (tid > 0 ? Ac[tid - 1] : (i128)0) - (i128)ix.xz[q] * Ac[tid];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (act) Ac[tid] = nv;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (tid < d) {
            // A_j[d-1] = 1, A_j[m-1] = A[m] + x_j A_j[m]
            const i128 xj = ix.xz[tid];
            i128 cur = 1;
            for (int m = d - 1; m >= 0; m--) {
                if (m < n_coef) ent[(size_t)m * d + tid] = cur;
                if (m > 0) cur = Ac[m] + xj * cur;
            }
        }
    } else if (do_z && tid < 64 + 4 * FS_MAXD + 32) {
        const int j = (tid - 64) >> 2, pq = (tid - 64) & 3;
        if (j < d) {
            uint32_t w[NL];
            fp_set(w, P.one);
            const uint32_t *row = inv + (size_t)ix.z[j] * n * NL;
            for (int q = pq; q < d; q += 4) {
                if (q == j) continue;
                uint32_t f[NL];
#pragma unroll
                for (int k = 0; k < NL; k++) f[k] = row[(size_t)ix.z[q] * NL + k];
                mont_mul(w, w, f, P);
            }
#pragma unroll
            for (int k = 0; k < NL; k++) part[((size_t)j * 4 + pq) * NL + k] = w[k];
        }
    } else if (do_zc && tid >= 256) {
        for (int e = tid - 256; e < nc * d; e += 1024 - 256) {
            const int i = e / d, j = e - i * d;
            const int64_t xi = ix.xzc[i];
            i128 v = 1;
            for (int q = 0; q < d; q++)
                if (q != j) v *= (i128)(xi - (int64_t)ix.xz[q]);
            ent[(size_t)(n_coef + i) * d + j] = v;
        }
    }
    __syncthreads();
    if (do_z && tid < d) {
        uint32_t a[NL], b[NL], c[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) { a[k] = part[((size_t)tid * 4) * NL + k]; b[k] = part[((size_t)tid * 4 + 1) * NL + k]; }
        mont_mul(c, a, b, P);
#pragma unroll
        for (int k = 0; k < NL; k++) { a[k] = part[((size_t)tid * 4 + 2) * NL + k]; b[k] = part[((size_t)tid * 4 + 3) * NL + k]; }
        mont_mul(a, a, b, P);
        mont_mul(c, c, a, P);
#pragma unroll
        for (int k = 0; k < NL; k++) wj[(size_t)tid * NL + k] = c[k];
    }
    __syncthreads();
    // ---- digits: one thread per entry of the rows this launch owns -----------------------------------------------------
    const int row_lo = do_z ? 0 : n_coef, row_hi = do_zc ? n_out : n_coef;
    for (int e = tid + row_lo * d; e < row_hi * d; e += 1024) {
        const int i = e / d, l = e - i * d;
        i128 v = ent[e];
        const bool negv = v < 0;
        int carry = 0;
#pragma unroll
        for (int b = 0; b < 16; b++) {
            int t = (int)((uint32_t)(v >> (8 * b)) & 0xffu) + carry;
            if (t > 127) { t -= 256; carry = 1; } else carry = 0;
            a8[fs_digit_addr(i, l, b, nkb)] = (uint8_t)(int8_t)t;
        }
        if (carry != (negv ? 1 : 0) && status) atomicOr(status, FS_OVERFLOW);
    }
    // ---- row constants: (0x80..80 * sum_l M[i][l] - bias sum) mod p as eight pairs [bias of the fold's columns + word] --------
    for (int i = tid + row_lo; i < row_hi; i += 1024) fs_row_constant(P, cs, ent + (size_t)i * d, d, crow + (size_t)i * 16);
    if (do_zc)
        for (int i = tid; i < nc; i += 1024) rowmode[n_coef + i] = (int32_t)ix.zc[i] + 1;
    if (do_z) {
        // T[l][q] = 2^(29 q) / den_l, canonical digits: mont_mul(w_l R, 2^(29 q)) = w_l 2^(29 q)
        for (int e = tid; e < d * NL; e += 1024) {
            const int l = e / NL, q = e - l * NL;
            uint32_t w[NL], two[NL], tq[NL];
#pragma unroll
            for (int k = 0; k < NL; k++) { w[k] = wj[(size_t)l * NL + k]; two[k] = (k == q) ? 1u : 0u; }
            mont_mul(tq, w, two, P);
#pragma unroll
            for (int k = 0; k < NL; k++) KT[((size_t)l * NL + q) * NL + k] = tq[k];
        }
    }
}

// The rows of the compared senders before anybody knows who they will be: P[i][j] = prod_{q != j} (x_i - x_zq) for EVERY party i, its 16
// digits per entry laid out as the 8 NKB sixteen-byte pieces a row occupies in the image, and its row constant -- a candidate store
// indexed by party (k_mm8f gathers the rows it compares in its prologue: FsPick).  Depends on the first d arrivals alone.
__device__ __forceinline__ void fs_cand_body(uint8_t *fc_lds, const FpParams<9> &P, int n, const uint16_t *__restrict__ xs, const FsIdx &ix, int d, const FsConsts &cs,
                                             uint8_t *__restrict__ cand, uint32_t *__restrict__ cand_crow, int32_t *__restrict__ status) {
    i128 *ent = reinterpret_cast<i128 *>(fc_lds);                      // [n][d]
    const int tid = threadIdx.x, nkb = (d + 7) / 8;
    const size_t row_bytes = (size_t)nkb * 8 * 16;
    for (size_t i = tid; i < (size_t)n * row_bytes / 16; i += 1024) reinterpret_cast<uint4 *>(cand)[i] = make_uint4(0, 0, 0, 0);
    for (int e = tid; e < n * d; e += 1024) {
        const int i = e / d, j = e - i * d;
        const int64_t xi = xs[i];
        i128 v = 1;
        for (int q = 0; q < d; q++)
            if (q != j) v *= (i128)(xi - (int64_t)ix.xz[q]);
        ent[e] = v;
    }
    __syncthreads();
    for (int e = tid; e < n * d; e += 1024) {
        const int i = e / d, l = e - i * d;
        const i128 v = ent[e];
        const bool negv = v < 0;
        const int kb = l >> 3, gg = (l & 7) >> 1, el = l & 1;
        int carry = 0;
#pragma unroll
        for (int b = 0; b < 16; b++) {
            int t = (int)((uint32_t)(v >> (8 * b)) & 0xffu) + carry;
            if (t > 127) { t -= 256; carry = 1; } else carry = 0;
            const int grp = b >> 3, r7 = 7 - (b & 7), hi = r7 >> 2, bi = r7 & 3;
            cand[(size_t)i * row_bytes + (size_t)(((kb * 2 + grp) * 4 + gg) * 16) + 4 * (2 * hi + el) + bi] = (uint8_t)(int8_t)t;
        }
        if (carry != (negv ? 1 : 0) && status) atomicOr(status, FS_OVERFLOW);
    }
    for (int i = tid; i < n; i += 1024) fs_row_constant(P, cs, ent + (size_t)i * d, d, cand_crow + (size_t)i * 16);
}

__global__ void __launch_bounds__(1024) k_fs_build(const FpParams<9> P, const uint32_t *__restrict__ inv, int n, const FsIdx ix, int d, int nc, int n_coef,
                                                   int flags, const FsConsts cs, uint8_t *__restrict__ a8, uint32_t *__restrict__ crow, uint32_t *__restrict__ KT,
                                                   int32_t *__restrict__ rowmode, int32_t *__restrict__ z_dev, int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fs_b_lds[];
    fs_build_body(fs_b_lds, P, inv, n, ix, d, nc, n_coef, flags, cs, a8, crow, KT, rowmode, z_dev, status);
}

__global__ void __launch_bounds__(1024) k_fs_cand(const FpParams<9> P, int n, const uint16_t *__restrict__ xs, const FsIdx ix, int d, const FsConsts cs,
                                                  uint8_t *__restrict__ cand, uint32_t *__restrict__ cand_crow, int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fs_c_lds[];
    fs_cand_body(fs_c_lds, P, n, xs, ix, d, cs, cand, cand_crow, status);
}

// What a decoder enqueues when its (degree + 1)-th column lands, as ONE launch of two workgroups: block 0 builds everything that depends on the
// arrivals alone (k_fs_build with FS_BUILD_Z), block 1 the candidate store (k_fs_cand) -- the two are independent, and as two launches
// the second waited for the first (8 us + a dispatch gap on the path of the column that completes the quorum).
__global__ void __launch_bounds__(1024) k_fs_build_z_cand(const FpParams<9> P, const uint32_t *__restrict__ inv, int n, const uint16_t *__restrict__ xs, const FsIdx ix, int d, int nc,
                                                          int n_coef, const FsConsts cs, uint8_t *__restrict__ a8, uint32_t *__restrict__ crow, uint32_t *__restrict__ KT,
                                                          int32_t *__restrict__ rowmode, int32_t *__restrict__ z_dev, uint8_t *__restrict__ cand, uint32_t *__restrict__ cand_crow,
                                                          int32_t *__restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fs_zc_lds[];
    if (blockIdx.x == 0) fs_build_body(fs_zc_lds, P, inv, n, ix, d, nc, n_coef, FS_BUILD_Z, cs, a8, crow, KT, rowmode, z_dev, status);
    else fs_cand_body(fs_zc_lds, P, n, xs, ix, d, cs, cand, cand_crow, status);
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static int fs_num_cus() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    }
    return n;
}

// Does [n_coef rows of N ; nc rows of P] over d arrivals of this point set fit the kernel?  The bounds hold for EVERY choice of
// arrivals: |coeff_m(A_j)| <= prod over the d - 1 largest points of (1 + x), |A_j(x_i)| <= the product of the d - 1 largest
// distances from a point, each distance taken by at most two points.
int fs_layout(hb_ctx *ctx, const PointTable *pt, int d, int nc, int n_coef, FsLayout *L) {
    if (ctx->n_limbs != 4 || !pt || !pt->usable || !pt->small) return HB_ERR_UNSUPPORTED;
    if (d < 4 || d > 22 || nc < 0 || nc > FS_MAXC || n_coef < 1 || n_coef > d) return HB_ERR_UNSUPPORTED;
    if (env_hook(ENV_NO_MFMA) || env_hook(ENV_NO_FUSED_SMALL) || env_hook(ENV_NO_QUICK) || !prescale_params(ctx)) return HB_ERR_UNSUPPORTED;
    // 128 * (16 digits of at most 128 in each of d terms) must stay below the accumulator bias
    if ((int64_t)d * 16 * 128 * 128 > MM8_BIAS) return HB_ERR_UNSUPPORTED;
    std::vector<uint16_t> xs(pt->xs);
    std::sort(xs.begin(), xs.end());
    double bits_n = 0.0, bits_p = 0.0;
    for (int k = 0; k < d - 1 && k < (int)xs.size(); k++) bits_n += log2(1.0 + xs[xs.size() - 1 - k]);
    const int R = xs.empty() ? 0 : xs.back() - xs.front();
    for (int k = 0; k < d - 1; k++) { const int dist = R - k / 2; if (dist < 1) break; bits_p += log2((double)dist); }
    if (bits_n > 124.5 || bits_p > 124.5) return HB_ERR_UNSUPPORTED;
    L->n = pt->n; L->d = d; L->nc = nc; L->n_coef = n_coef; L->n_out = n_coef + nc;
    L->nkb = (d + 7) / 8; L->n_rt = (L->n_out + 15) / 16;
    if (fs_lds_bytes(L->n_rt, L->nkb) > FS_LDS_LIMIT) return HB_ERR_UNSUPPORTED;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    L->o_a8 = 0;
    L->o_crow = al((size_t)L->n_rt * L->nkb * 2 * 64 * 16);
    L->o_kt = L->o_crow + al((size_t)L->n_rt * 16 * 16 * 4);
    L->o_mode = L->o_kt + al((size_t)d * 81 * 4);
    L->o_z = L->o_mode + al((size_t)L->n_rt * 16 * 4);
    L->need = L->o_z + al((size_t)FS_MAXD * 4);
    L->o_cand = L->o_cand_crow = 0;
    if (pt->n <= FS_CAND_MAXN && nc > 0) {
        L->o_cand = L->need;
        L->o_cand_crow = L->o_cand + al((size_t)pt->n * L->nkb * 8 * 16);
        L->need = L->o_cand_crow + al((size_t)pt->n * 16 * 4);
    }
    return HB_OK;
}

// enqueue the build into `base` (L.need bytes): FS_BUILD_Z needs z, FS_BUILD_ZC both (the rows of P depend on z too)
int fs_build(hb_ctx *ctx, const PointTable *pt, const int32_t *z, const int32_t *zc, const FsLayout &L, uint8_t *base, int flags, int32_t *status_dev, hipStream_t s) {
    const int n = L.n, d = L.d, nc = L.nc;
    FsIdx ix;
    memset(&ix, 0, sizeof ix);
    uint64_t seen[4] = {0, 0, 0, 0};            // (n <= 65535 in general: the bitmap covers the first 256, larger sets use the vector)
    std::vector<uint8_t> seen_big;
    if (n > 256) seen_big.assign((size_t)n, 0);
    auto mark = [&](int v) -> bool {
        if (v < 0 || v >= n) return false;
        if (n > 256) { if (seen_big[v]) return false; seen_big[v] = 1; return true; }
        if (seen[v >> 6] >> (v & 63) & 1) return false;
        seen[v >> 6] |= 1ull << (v & 63);
        return true;
    };
    for (int i = 0; i < d; i++) { if (!mark(z[i])) return fail(ctx, HB_ERR_BAD_ARG, "fused decode: arrival indices"); ix.z[i] = (uint16_t)z[i]; ix.xz[i] = pt->xs[z[i]]; }
    if (flags & FS_BUILD_ZC)
        for (int j = 0; j < nc; j++) { if (!mark(zc[j])) return fail(ctx, HB_ERR_BAD_ARG, "fused decode: compared indices"); ix.zc[j] = (uint16_t)zc[j]; ix.xzc[j] = pt->xs[zc[j]]; }
    const Mm8Shared *sh = nullptr;
    int rc = mm8_shared(ctx, &sh, s); if (rc) return rc;
    FsConsts cs;
    memcpy(cs.c80r, sh->c80r, sizeof cs.c80r);
    memcpy(cs.biasmod, sh->biasmod, sizeof cs.biasmod);
    const size_t lds = ((size_t)L.n_out * d + d + 1) * 16 + ((size_t)d * 9 + (size_t)d * 4 * 9) * 4;
    k_fs_build<<<1, 1024, lds, s>>>(ctx->pw, pt->inv, n, ix, d, nc, L.n_coef, flags, cs, base + L.o_a8, (uint32_t *)(base + L.o_crow), (uint32_t *)(base + L.o_kt),
                                    (int32_t *)(base + L.o_mode), (int32_t *)(base + L.o_z), status_dev);
    HB_LAUNCH_CHECK(ctx);
    return HB_OK;
}

int fs_build_cand(hb_ctx *ctx, PointTable *pt, const int32_t *z, const FsLayout &L, uint8_t *base, int32_t *status_dev, hipStream_t s, bool with_z) {
    if (!L.o_cand) return fail(ctx, HB_ERR_BAD_ARG, "fused decode: no candidate store in this layout");
    if (!pt->xs_dev) {
        HB_HIP(ctx, hipMalloc(&pt->xs_dev, (size_t)pt->n * 2));
        int rc = upload_table(ctx, pt->xs_dev, pt->xs.data(), (size_t)pt->n * 2, s);      // (synchronises; once per point set)
        if (rc) { (void)hipFree(pt->xs_dev); pt->xs_dev = nullptr; return rc; }
    }
    FsIdx ix;
    memset(&ix, 0, sizeof ix);
    uint64_t seen[2] = {0, 0};                  // (a candidate store exists for at most 128 parties)
    for (int i = 0; i < L.d; i++) {
        const int v = z[i];
        if (v < 0 || v >= pt->n || (seen[v >> 6] >> (v & 63) & 1)) return fail(ctx, HB_ERR_BAD_ARG, "fused decode: arrival indices");
        seen[v >> 6] |= 1ull << (v & 63);
        ix.z[i] = (uint16_t)v; ix.xz[i] = pt->xs[v];
    }
    const Mm8Shared *sh = nullptr;
    int rc = mm8_shared(ctx, &sh, s); if (rc) return rc;
    FsConsts cs;
    memcpy(cs.c80r, sh->c80r, sizeof cs.c80r);
    memcpy(cs.biasmod, sh->biasmod, sizeof cs.biasmod);
    if (with_z) {
        const size_t lds_b = ((size_t)L.n_out * L.d + L.d + 1) * 16 + ((size_t)L.d * 9 + (size_t)L.d * 4 * 9) * 4, lds_c = (size_t)pt->n * L.d * 16;
        k_fs_build_z_cand<<<2, 1024, std::max(lds_b, lds_c), s>>>(ctx->pw, pt->inv, pt->n, pt->xs_dev, ix, L.d, L.nc, L.n_coef, cs, base + L.o_a8, (uint32_t *)(base + L.o_crow),
                                                                  (uint32_t *)(base + L.o_kt), (int32_t *)(base + L.o_mode), (int32_t *)(base + L.o_z), base + L.o_cand,
                                                                  (uint32_t *)(base + L.o_cand_crow), status_dev);
        HB_LAUNCH_CHECK(ctx);
        return HB_OK;
    }
    k_fs_cand<<<1, 1024, (size_t)pt->n * L.d * 16, s>>>(ctx->pw, pt->n, pt->xs_dev, ix, L.d, cs, base + L.o_cand, (uint32_t *)(base + L.o_cand_crow), status_dev);
    HB_LAUNCH_CHECK(ctx);
    return HB_OK;
}

// The terms of the next unit dealt to the waves.  A wave's load in a unit: its passes (wave w takes pairs w, w + 8, ... of the 4 n_rt; a pass
// over a ragged last row tile with at most 8 rows reduces two of the four outputs a lane holds) plus its terms; measured on config 3's R2
// launch (scratch/fs_phase_timing.py) a pass is 9.5 k cycles of a wave's time and a term 3.6 k when two waves share the SIMD.  Each term goes
// to the wave with the least load so far (ties: the higher wave -- the older waves 0..3 win the issue arbitration anyway).
static FsSplit fs_split(int d, int n_rt, int n_out) {
    constexpr int PASS = 95, RAGGED = 70, TERM = 36;
    FsSplit sp;
    int load[FS_WAVES];
    const int n_pairs = FS_TPW * n_rt;
    const bool ragged = n_out - 16 * (n_rt - 1) <= 8;
    for (int w = 0; w < FS_WAVES; w++) {
        sp.terms[w] = 0;
        load[w] = 0;
        for (int pidx = w; pidx < n_pairs; pidx += FS_WAVES) {
            const int tl = n_rt == 2 ? (pidx >> 1) : pidx / n_rt;
            const int rt = n_rt == 2 ? ((pidx ^ (pidx >> 2)) & 1) : pidx - tl * n_rt;
            load[w] += (ragged && rt == n_rt - 1) ? RAGGED : PASS;
        }
    }
    for (int l = 0; l < d; l++) {
        int best = FS_WAVES - 1;
        for (int w = FS_WAVES - 2; w >= 0; w--)
            if (load[w] < load[best]) best = w;
        sp.terms[best] |= 1u << l;
        load[best] += TERM;
    }
    return sp;
}

// the launch over a built image: rows with a store mode go to `out` (view ov, clipped at out_count), rows with a compare mode are
// checked against the rows of `cols` they name
int fs_launch(hb_ctx *ctx, const FsLayout &L, const uint8_t *base, const uint32_t *cols, hb_view cv, uint32_t *out, hb_view ov, int64_t out_count,
              int32_t *mismatch_dev, int32_t *first_bad_dev, uint32_t *bad_map_dev, int64_t C, hipStream_t s, const FsDone *done_p, const int32_t *pick_zc) {
    FsDone done;
    memset(&done, 0, sizeof done);
    if (done_p) done = *done_p;
    FsPick pick;
    memset(&pick, 0, sizeof pick);
    if (pick_zc) {
        if (!L.o_cand) return fail(ctx, HB_ERR_BAD_ARG, "fused decode: no candidate store to pick from");
        pick.cand = (const uint4 *)(base + L.o_cand); pick.cand_crow = (const uint32_t *)(base + L.o_cand_crow);
        pick.n_coef = L.n_coef; pick.nc = L.nc;
        for (int j = 0; j < L.nc; j++) pick.zc[j] = (uint16_t)pick_zc[j];
    }
    if (C <= 0) return HB_OK;
    const Mm8Shared *sh = nullptr;
    int rc = mm8_shared(ctx, &sh, s); if (rc) return rc;
    const int64_t n_units = (C + 63) / 64;
    int64_t blocks = fs_num_cus();
    if (blocks > n_units) blocks = n_units;
    blocks = mm8_trimmed_grid(n_units, blocks);          // (config 3: 249 of 256 -- seven CUs stay free for what other streams launch meanwhile)
    const size_t lds = fs_lds_bytes(L.n_rt, L.nkb);
    const FsSplit split = fs_split(L.d, L.n_rt, L.n_out);
#define FS_LAUNCH(NKB)                                                                                                                     \
    do {                                                                                                                                   \
        static std::atomic<unsigned long long> attr_done{0};   /* one bit per device: the attribute is per device */                                                                                                     \
        if (!((attr_done.load() >> (ctx->device & 63)) & 1ull)) { HB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_mm8f<NKB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_done.fetch_or(1ull << (ctx->device & 63)); } \
        k_mm8f<NKB><<<dim3((unsigned)blocks), dim3(64 * FS_WAVES), lds, s>>>((const int4 *)(base + L.o_a8), (const uint32_t *)(base + L.o_crow), sh->fold_dev,        \
            (const uint32_t *)(base + L.o_kt), ctx->psc, cols, cv.stride_c, cv.stride_l, (const int32_t *)(base + L.o_z), INT64_MAX, L.d, (const int32_t *)(base + L.o_mode), \
            out, ov.stride_c, ov.stride_l, out_count, mismatch_dev, first_bad_dev, bad_map_dev, L.n_out, L.n_rt, C, n_units, sh->bp, done, pick, split); \
    } while (0)
    switch (L.nkb) {
        case 1: FS_LAUNCH(1); break;
        case 2: FS_LAUNCH(2); break;
        case 3: FS_LAUNCH(3); break;
        default: return fail(ctx, HB_ERR_UNSUPPORTED, "fused decode: more than 24 terms");
    }
#undef FS_LAUNCH
    HB_LAUNCH_CHECK(ctx);
    return HB_OK;
}

}  // namespace hb

#ifdef HB_MM8_TIMING
extern "C" int hb_debug_fs_timing(unsigned long long *out, int count) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(hb::g_fs_t), sizeof(unsigned long long) * (size_t)count) == hipSuccess ? 0 : 1;
}
#endif
