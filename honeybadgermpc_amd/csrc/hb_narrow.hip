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

// the six accumulators of a row -> the canonical residue -> stored or compared (shared by the kernels)
struct Mv64Out { const uint64_t *in; int64_t in_sc, in_sl; uint64_t *out; int64_t out_sc, out_sl, out_count; int32_t *mismatch, *first_bad; };
// S = sum_k w[k] 2^(32 k) (five words; S < p 2^128) -> S 2^-128 mod p, canonical -> stored as output row -md - 1 or compared with row md - 1
// (got: the compared row's value when the caller has fetched it ahead -- have_got; otherwise it is loaded here)
// S (five words) -> S 2^(-32 STEPS) mod p, canonical, where S / 2^(32 STEPS) < p: four steps for k_mv64's sums (< 40 p 2^64), three for k_mv64m's
// (< 2^136 and p >= 2^41)
template <int STEPS = 4>
__device__ __forceinline__ uint64_t mv64_montgomery(uint32_t (&w)[6], const Mv64Params &prm) {
    // Montgomery steps of 32 bits: S <- (S + u p) / 2^32, u = w0 pinv32 mod 2^32
#pragma unroll
    for (int k = 0; k < STEPS; k++) {
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
    // result < S / 2^(32 STEPS) + p < 2 p in (w0, w1, w2 <= 1)
    uint64_t r = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    const uint64_t pp = (uint64_t)prm.p0 | ((uint64_t)prm.p1 << 32);
    if (w[2] || r >= pp) r -= pp;
    return r;
}
__device__ __forceinline__ void mv64_reduce_emit(uint32_t (&w)[6], const Mv64Params &prm, int md, int64_t chunk, int64_t cc, bool live, const Mv64Out &o) {
    const uint64_t r = mv64_montgomery(w, prm);
    if (md < 0) {
        const int64_t oidx = chunk * o.out_sc + (int64_t)(-md - 1) * o.out_sl;
        if (live && oidx < o.out_count) o.out[oidx] = r;
    } else {
        const uint64_t got = o.in[live ? cc * o.in_sc + (int64_t)(md - 1) * o.in_sl : 0];
        const bool bad = live && got != r;
        const unsigned long long bl = __builtin_amdgcn_ballot_w64(bad);
        if (bl && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(bl)) {
            if (*reinterpret_cast<volatile int32_t *>(o.mismatch) == 0) atomicOr(o.mismatch, 1);
            if (o.first_bad && *reinterpret_cast<volatile int32_t *>(o.first_bad) > (int32_t)chunk) atomicMin(o.first_bad, (int32_t)chunk);
        }
    }
}
__device__ __forceinline__ void mv64_finish(uint64_t a0, uint64_t a1, uint64_t a2, uint64_t b0, uint64_t b1, uint64_t b2, const Mv64Params &prm, int md, int64_t chunk,
                                            int64_t cc, bool live, const Mv64Out &o) {
    // S = a0 + a1 2^22 + a2 2^44 + b0 2^32 + b1 2^54 + b2 2^76 as six 32-bit words (S < 40 p 2^64 < 2^134)
    uint32_t w[6];
    unsigned __int128 s = (unsigned __int128)a0 + ((unsigned __int128)a1 << 22) + ((unsigned __int128)a2 << 44) + ((unsigned __int128)b0 << 32) +
                          ((unsigned __int128)b1 << 54);
    // b2 << 76 does not fit 128 bits with the rest: split off the top
    const unsigned __int128 hi = (unsigned __int128)b2 << 12;        // weight 2^64
    const uint64_t lo64 = (uint64_t)s;
    unsigned __int128 up = (s >> 64) + hi;                           // < 2^72
    w[0] = (uint32_t)lo64; w[1] = (uint32_t)(lo64 >> 32);
    w[2] = (uint32_t)up; w[3] = (uint32_t)(up >> 32); w[4] = (uint32_t)(up >> 64); w[5] = 0;
    mv64_reduce_emit(w, prm, md, chunk, cc, live, o);
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

// ---------------------------------------------------------------------------------------------------------------------------------
// k_mv64m (round 6): the same product on the int8 MATRIX CORES, as k_mm8 does it for 32-byte elements (hb_mfma.hip).
//
//   out(c, i) = sum_l M[i][l] in(c, l) mod p:   in = sum_a X_a 2^(8a), a < 8 (the element's bytes as they lie in HBM, biased by XOR 0x80);
//   M 2^96 mod p -- the factor three Montgomery steps take out again (a sum here is < 2^136: three suffice for p >= 2^41) -- as EIGHT balanced base-256 digits: the digits' range,
//   [-128 S, 127 S] with S = (2^64 - 1) / 255, is 2^64 - 1 wide, so every residue of a p < 2^64 has a representative in it;
//   S = sum_c col_c 2^(8c), col_c = sum_l sum_b M_b[l] X_(c-b)[l], c < 15: for a column the sum over (l, b) is an int8 dot product of a constant row and
//   an 8-byte WINDOW of every input element (bytes c - 7 .. c, zeros outside the element).  v_mfma_i32_16x16x64_i8 contracts 64 products: a K-block
//   is 8 terms x 8 digits, lane (n, g) of the B operand holds the windows of terms 8 kb + 2 g, + 1 of chunk n, lane (r, g) of the A operand the
//   digits of row r at those terms (one dwordx4 of the image, the same for all 15 columns: the window slides on the B side).  15 MFMAs a K-block
//   and 16 x 16 outputs where k_mv64 issues 144 multiply-adds a row and thread; no fold -- a sum is 136 bits, five words, straight into the three
//   Montgomery steps.  Accumulators start from a bias (columns non-negative: pairs of them fit 32 bits); bias and XOR correction are one residue a row.
// A work item is a tile of 16 chunks and a pair of row tiles (120 accumulator registers); persistent waves take the items round-robin.  d <= 24
// (three K-blocks); any number of rows whose image fits a launch's 64 KB of LDS (mv64_from_host).
constexpr int MV64M_BIAS = 3200000;          // >= 24 terms x 8 digit pairs x 128 x 128 = |column|, and 2 x BIAS x 257 < 2^32
typedef int mv_v4i __attribute__((ext_vector_type(4)));
typedef uint32_t mv_v16u __attribute__((ext_vector_type(16)));

template <int NKB>
__global__ __launch_bounds__(256, 2) void k_mv64m(const uint4 *__restrict__ a8, const uint64_t *__restrict__ crow, const int32_t *__restrict__ rowmode, int n_out, int n_rt, int d,
                                               const uint64_t *__restrict__ in, int64_t in_sc, int64_t in_sl, const int32_t *__restrict__ in_rows, int64_t in_count,
                                               uint64_t *__restrict__ out, int64_t out_sc, int64_t out_sl, int64_t out_count,
                                               int32_t *__restrict__ mismatch, int32_t *__restrict__ first_bad, int64_t C, int64_t n_items, const Mv64Params prm) {
    extern __shared__ uint4 mvm_lds[];
    uint4 *al = mvm_lds;                                                        // [n_rt][NKB][64] digits
    // per output row: its constant, its offset in the buffer it goes to (stored rows) or is compared with, its mode -- read four rows at a time
    uint64_t *crl = reinterpret_cast<uint64_t *>(mvm_lds + (size_t)n_rt * NKB * 64);   // [16 n_rt]
    int64_t *rol = reinterpret_cast<int64_t *>(crl + 16 * n_rt);                // [16 n_rt]
    int32_t *mdl = reinterpret_cast<int32_t *>(rol + 16 * n_rt);               // [16 n_rt]
    for (int i = threadIdx.x; i < n_rt * NKB * 64; i += 256) al[i] = a8[i];
    for (int i = threadIdx.x; i < 16 * n_rt; i += 256) {
        const int md = i < n_out ? rowmode[i] : 0;
        crl[i] = i < n_out ? crow[i] : 0ull;
        rol[i] = md > 0 ? (int64_t)(md - 1) * in_sl : md < 0 ? (int64_t)(-md - 1) * out_sl : 0;
        mdl[i] = md;
    }
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (uniform, and the compiler is told: items, row tiles
    const int n = lane & 15, g = lane >> 4;                                                            //  and every branch on them are scalar)
    const int n_pairs = (n_rt + 1) >> 1;
    const int64_t n_tiles = (C + 15) >> 4;
    const bool pair_major = (n_rt & 1) && n_pairs > 1;      // (see the item loop)
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    // Persistent waves over items = (tile of 16 chunks, pair of row tiles): a launch is a few items a wave, and an item's serial chain -- elements from
    // HBM, the products, eight reductions, stores -- is hidden under the wave's NEXT item's loads (issued ahead) and the SIMD's other wave.  (A wave
    // per chunk tile and all its rows, one round of workgroups: 2.9 waves a SIMD at two resident = two rounds of that chain, slower than k_mv64.)
    // this lane's two elements of every K-block: terms 8 kb + 2 g and + 1 of its chunk, biased (low, high dword); their rows' offsets are the lane's own
    // for the whole launch (a gathered decode looks its rows up once, not a fetch)
    int64_t toff[NKB][2];
#pragma unroll
    for (int kb = 0; kb < NKB; kb++)
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int t = 8 * kb + 2 * g + e;
            toff[kb][e] = t < d ? (int64_t)(in_rows ? in_rows[t] : t) * in_sl : INT64_MIN;
        }
    auto fetch = [&](int64_t item, uint32_t (&X)[NKB][2][2]) {
        const int64_t ch = (pair_major ? item % n_tiles : item / n_pairs) * 16 + n;
        const int64_t base = (ch < C ? ch : C - 1) * in_sc;
#pragma unroll
        for (int kb = 0; kb < NKB; kb++)
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const int64_t idx = base + toff[kb][e];
                const uint64_t v = (toff[kb][e] != INT64_MIN && idx < in_count) ? in[idx] : 0ull;
                X[kb][e][0] = (uint32_t)v ^ 0x80808080u;
                X[kb][e][1] = (uint32_t)(v >> 32) ^ 0x80808080u;
            }
    };
    uint32_t X[NKB][2][2], Xn[NKB][2][2];
    int64_t item = (int64_t)blockIdx.x * 4 + wave;
    if (item < n_items) fetch(item, X);
    __syncthreads();                                  // (the image's copy above and the first elements' loads are in flight together)
    for (; item < n_items; item += n_waves) {
        // An odd tile count's last pair is half the work of the others: dealt tile-major with an even number of waves it went to the same waves --
        // the same SIMDs -- every time (R2's 43 rows: SIMDs 0 and 2 of every CU did twice the work of 1 and 3).  Those launches are dealt
        // pair-major, the light items last; the others tile-major (the two pairs of a tile read the same elements: neighbours in time)
        const int64_t pair = pair_major ? item / n_tiles : item % n_pairs, tile = pair_major ? item - pair * n_tiles : item / n_pairs;
        const int rp = 2 * (int)pair;
        const int64_t chunk = tile * 16 + n;
        const bool live = chunk < C;
        const int64_t cc = live ? chunk : C - 1;
        const bool more = item + n_waves < n_items;
        if (more) fetch(item + n_waves, Xn);
        mv_v4i acc[2][15];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int c = 0; c < 15; c++) acc[r][c] = mv_v4i{MV64M_BIAS, MV64M_BIAS, MV64M_BIAS, MV64M_BIAS};
        const bool two = rp + 1 < n_rt;
#ifndef MVM_NO_MFMA
#pragma unroll
        for (int kb = 0; kb < NKB; kb++) {
            const uint4 a0 = al[((size_t)rp * NKB + kb) * 64 + lane];
            const uint4 a1 = al[((size_t)(two ? rp + 1 : rp) * NKB + kb) * 64 + lane];
            const mv_v4i A0 = mv_v4i{(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w}, A1 = mv_v4i{(int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
            // The padded element is the dwords (0, 0, lo, hi, 0, 0); the window of column c starts at its byte c + 1 = 4 q + rho and is the dwords
            // q, q + 1 of the element shifted right by rho bytes.  Per shift ONE file F[2 q + e], q < 5, of both elements interleaved: a window is
            // four consecutive registers of it (q = 0 and q = 4 are zero; three real entries an element: lo << , the straddling dword, hi >>)
#pragma unroll
            for (int rho = 0; rho < 4; rho++) {
                mv_v16u F;
#pragma unroll
                for (int e = 0; e < 2; e++) {
                    const uint32_t lo = X[kb][e][0], hi = X[kb][e][1];
                    F[0 + e] = 0u;
                    F[2 + e] = rho ? __builtin_amdgcn_alignbyte(lo, 0u, rho) : 0u;
                    F[4 + e] = rho ? __builtin_amdgcn_alignbyte(hi, lo, rho) : lo;
                    F[6 + e] = rho ? __builtin_amdgcn_alignbyte(0u, hi, rho) : hi;
                    F[8 + e] = 0u;
                }
                F[10] = F[11] = F[12] = F[13] = F[14] = F[15] = 0u;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int c = 4 * q + rho - 1;
                    if (c < 0) continue;
                    mv_v4i B;
                    if (q == 0) B = (mv_v4i)__builtin_shufflevector(F, F, 0, 1, 2, 3);
                    else if (q == 1) B = (mv_v4i)__builtin_shufflevector(F, F, 2, 3, 4, 5);
                    else if (q == 2) B = (mv_v4i)__builtin_shufflevector(F, F, 4, 5, 6, 7);
                    else B = (mv_v4i)__builtin_shufflevector(F, F, 6, 7, 8, 9);
                    acc[0][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0, B, acc[0][c], 0, 0, 0);
                    acc[1][c] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1, B, acc[1][c], 0, 0, 0);      // (an odd tile count's last pair: the first tile again, not used;
                                                                                                       //  behind a scalar branch the MFMAs no longer issue back to back: encode 24.0 -> 25.8 us)
                }
            }
        }
#endif
        // output j of lane (n, g) of row tile rt is row 16 rt + 4 g + j of chunk n.  A tile's compared rows (a wave-uniform question first: an encode
        // has none) are fetched for all its outputs ahead of the reductions: one round trip, not four
        const int64_t in_base = cc * in_sc, out_base = chunk * out_sc;
        const uint64_t pp = (uint64_t)prm.p0 | ((uint64_t)prm.p1 << 32);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            if (r == 1 && !two) break;
            const int i4 = 16 * (rp + r) + 4 * g;
            const int4 md4 = *reinterpret_cast<const int4 *>(mdl + i4);
            const int mdv[4] = {md4.x, md4.y, md4.z, md4.w};
            int64_t ro[4];
            uint64_t cr[4], got[4] = {0, 0, 0, 0};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint4 a = *reinterpret_cast<const uint4 *>(rol + i4 + 2 * h), b = *reinterpret_cast<const uint4 *>(crl + i4 + 2 * h);
                ro[2 * h] = (int64_t)((uint64_t)a.x | ((uint64_t)a.y << 32)); ro[2 * h + 1] = (int64_t)((uint64_t)a.z | ((uint64_t)a.w << 32));
                cr[2 * h] = (uint64_t)b.x | ((uint64_t)b.y << 32); cr[2 * h + 1] = (uint64_t)b.z | ((uint64_t)b.w << 32);
            }
            const bool compares = __builtin_amdgcn_ballot_w64((md4.x > 0) | (md4.y > 0) | (md4.z > 0) | (md4.w > 0)) != 0;
            if (compares) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (mdv[j] > 0) got[j] = in[in_base + ro[j]];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int md = mdv[j];
                // S = sum_c col_c 2^(8 c): pairs f_k = col_2k + col_(2k+1) 2^8 (< 2^32) at bit 16 k.  The even pairs are the words of one number, the
                // odd pairs of a second, 16 bits up: S is one five-word add
                uint32_t f[8];
#pragma unroll
                for (int k = 0; k < 7; k++) f[k] = (uint32_t)acc[r][2 * k][j] + ((uint32_t)acc[r][2 * k + 1][j] << 8);
                f[7] = (uint32_t)acc[r][14][j];
                uint32_t w[6], cy;
                w[0] = __builtin_addc(f[0], f[1] << 16, 0u, &cy);
                w[1] = __builtin_addc(f[2], __builtin_amdgcn_alignbit(f[3], f[1], 16), cy, &cy);
                w[2] = __builtin_addc(f[4], __builtin_amdgcn_alignbit(f[5], f[3], 16), cy, &cy);
                w[3] = __builtin_addc(f[6], __builtin_amdgcn_alignbit(f[7], f[5], 16), cy, &cy);
                w[4] = (f[7] >> 16) + cy;
                w[5] = 0;
#ifdef MVM_NO_EPI
                uint64_t res = (uint64_t)w[0] | ((uint64_t)(w[1] ^ w[2] ^ w[3] ^ w[4]) << 32);
#else
                uint64_t res = mv64_montgomery<3>(w, prm);
                const uint64_t r2 = res + cr[j];                           // (the row's constant: a canonical residue)
                res = (r2 < res || r2 >= pp) ? r2 - pp : r2;
#endif
                if (md < 0) {
                    const int64_t oidx = out_base + ro[j];
                    if (live && oidx < out_count) out[oidx] = res;
                } else if (compares) {
                    const bool bad = md > 0 && live && got[j] != res;
                    const unsigned long long bl = __builtin_amdgcn_ballot_w64(bad);
                    if (bl && lane == __builtin_ctzll(bl)) {
                        if (*reinterpret_cast<volatile int32_t *>(mismatch) == 0) atomicOr(mismatch, 1);
                        if (first_bad && *reinterpret_cast<volatile int32_t *>(first_bad) > (int32_t)chunk) atomicMin(first_bad, (int32_t)chunk);
                    }
                }
            }
        }
        if (more) {
#pragma unroll
            for (int kb = 0; kb < NKB; kb++)
#pragma unroll
                for (int e = 0; e < 2; e++) { X[kb][e][0] = Xn[kb][e][0]; X[kb][e][1] = Xn[kb][e][1]; }
        }
    }
}

// rows are padded to DT terms; 22 is config 3's degree + 1 (a row of 24 is 9 % of multiply-adds on zeros)
static int mv64_dt(int d) { return d <= 8 ? 8 : (d <= 16 ? 16 : (d <= 22 ? 22 : (d <= 24 ? 24 : (d <= 32 ? 32 : MV64_DMAX)))); }

struct Mv64Matrix {
    int n_out, d;
    uint2 *M;            // [n_out][DT], DT = mv64_dt(d): (M[i][l] 2^128 mod p) as (low, high) 32-bit halves, rows zero padded
    int32_t *mode;       // [n_out]
    // the matrix-core image (d <= 24; null: k_mv64 alone): [n_rt][nkb][64 lanes] x 16 digits, lane (r, g) = row 16 rt + r at terms 8 kb + 2 g, + 1:
    // byte 4 dd + bi = digit 7 - 4 (dd >> 1) - bi of term 8 kb + 2 g + (dd & 1); and per row the residue that takes bias and XOR correction out
    uint4 *a8;
    uint64_t *crow;
    int nkb, n_rt;
};

static inline uint64_t mulmod_u64(uint64_t a, uint64_t b, uint64_t p) { return (uint64_t)(((unsigned __int128)a * b) % p); }

bool mv64_applies(const hb_ctx *ctx, int d) {
    return ctx->n_limbs == 1 && ctx->p_limbs[0] >= 3 && d >= 1 && d <= MV64_DMAX && !env_hook(ENV_NO_NARROW_FAST);
}

// the matrix-core image is built (k_mv64m): rows of up to 24 terms, p >= 2^41 (three Montgomery steps bring its sums below 2 p)
bool mv64_matrix_cores(const hb_ctx *ctx, int d) {
    return mv64_applies(ctx, d) && d <= 24 && (ctx->p_limbs[0] >> 41) != 0 && !env_hook(ENV_NO_MFMA);
}

void mv64_free(Mv64Matrix *m) {
    if (!m) return;
    if (m->M) (void)hipFree(m->M);
    if (m->mode) (void)hipFree(m->mode);
    if (m->a8) (void)hipFree(m->a8);
    if (m->crow) (void)hipFree(m->crow);
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
    m->n_out = n_out; m->d = d; m->M = nullptr; m->mode = nullptr; m->a8 = nullptr; m->crow = nullptr; m->nkb = m->n_rt = 0;
    hipError_t e = hipMalloc(&m->M, img.size() * sizeof(uint2));
    if (e == hipSuccess) e = hipMalloc(&m->mode, (size_t)n_out * 4);
    if (e != hipSuccess) { mv64_free(m); ctx->err = std::string("narrow matrix: ") + hipGetErrorString(e); return HB_ERR_HIP; }
    int rc = upload_table(ctx, m->M, img.data(), img.size() * sizeof(uint2), s);
    if (!rc) rc = upload_table(ctx, m->mode, mode_host, (size_t)n_out * 4, s);
    if (rc) { mv64_free(m); return rc; }
    // (the image and the row tables of a launch live in LDS: matrices whose image passes the 64 KB a launch gets without asking stay on k_mv64)
    if (mv64_matrix_cores(ctx, d) && (size_t)((n_out + 15) / 16) * (((size_t)(d + 7) / 8) * 64 * 16 + 16 * (8 + 8 + 4)) <= 64 * 1024) {
        // the matrix-core image: every entry's representative in the eight balanced digits' range, the row constants.  A sum there is bounded by
        // the digits, < 2^136, not by p: THREE Montgomery steps leave S / 2^96 + p < 2 p once p >= 2^41 (entries are kept as M 2^96 mod p)
        const int nkb = (d + 7) / 8, n_rt = (n_out + 15) / 16;
        std::vector<uint8_t> dig((size_t)n_rt * nkb * 64 * 16, 0);
        std::vector<uint64_t> cr((size_t)n_rt * 16, 0);
        const unsigned __int128 P = p;
        const uint64_t S255 = 0x0101010101010101ull;                       // (2^64 - 1) / 255
        const unsigned __int128 hi_max = (unsigned __int128)127 * S255;     // the largest value eight balanced digits hold
        // 2^-128 mod p = (2^-64)^2: by p (p^-1 mod 2^64) = 1 mod 2^64
        uint64_t pinv = 1;
        for (int k = 0; k < 6; k++) pinv *= 2 - p * pinv;
        auto inv128 = [&](uint64_t v) -> uint64_t {                         // v 2^-128 mod p by two exact steps v <- (v + u p) / 2^64
            unsigned __int128 t = v;
            for (int k = 0; k < 2; k++) {
                const uint64_t u = (uint64_t)t * (0 - pinv);
                const unsigned __int128 up = (unsigned __int128)u * P;
                // (t + up) / 2^64 without overflow: the low words cancel to zero with a carry of (low(t) != 0)
                t = (t >> 64) + (up >> 64) + (((uint64_t)t) ? 1 : 0);
            }
            return (uint64_t)(t % P);
        };
        const uint64_t r96 = (uint64_t)((((unsigned __int128)1 << 96)) % P);
        const uint64_t k8 = S255;                                           // sum_{a < 8} 256^a
        unsigned __int128 k15 = 0;
        for (int c = 0; c < 15; c++) k15 += (unsigned __int128)1 << (8 * c);
        const uint64_t bias_mod = (uint64_t)((((unsigned __int128)MV64M_BIAS % P) * (k15 % P)) % P);
        for (int i = 0; i < n_out; i++) {
            const int rt = i >> 4, r = i & 15;
            uint64_t sum_mod = 0;                                           // sum_l (the representative) mod p
            for (int l = 0; l < d; l++) {
                const uint64_t v = mulmod_u64(m_host[(size_t)i * d + l] % p, r96, p);
                sum_mod = (uint64_t)(((unsigned __int128)sum_mod + v) % P);
                // the representative: v where eight balanced digits hold it, else v - p (>= -128 S: the digits' range is 2^64 - 1 >= p - 1 wide)
                __int128 val = (unsigned __int128)v > hi_max ? (__int128)v - (__int128)P : (__int128)v;
                int dg[8];
                for (int b = 0; b < 8; b++) {
                    int tdig = (int)(uint64_t)(val & 0xff);                 // (two's complement: the low byte of a negative value too)
                    val >>= 8;                                              // arithmetic
                    if (tdig > 127) { tdig -= 256; val += 1; }
                    dg[b] = tdig;
                }
                if (val != 0) { mv64_free(m); return fail(ctx, HB_ERR_HIP, "narrow matrix: an entry left the eight digits' range"); }
                const int kb = l >> 3, gg = (l & 7) >> 1, e = l & 1;
                uint8_t *lane16 = &dig[((((size_t)rt * nkb + kb) * 64) + (size_t)(r + 16 * gg)) * 16];
                for (int b = 0; b < 8; b++) {
                    const int k = 7 - b;                                    // window byte that digit b meets
                    lane16[4 * (2 * (k >> 2) + e) + (k & 3)] = (uint8_t)(int8_t)dg[b];
                }
            }
            // out = REDC(S_mfma) + (128 K8 sum_l rep - BIAS K15) 2^-96:  sum_l rep = sum_mod (mod p);  2^-96 = 2^-128 2^32
            const uint64_t corr = (uint64_t)((((unsigned __int128)128 * (k8 % p)) % P * sum_mod) % P);
            const uint64_t tot = corr >= bias_mod ? corr - bias_mod : corr + (p - bias_mod);
            cr[(size_t)i] = inv128(mulmod_u64(tot, (uint64_t)1 << 32, p));
        }
        hipError_t e2 = hipMalloc(&m->a8, dig.size());
        if (e2 == hipSuccess) e2 = hipMalloc(&m->crow, cr.size() * 8);
        if (e2 != hipSuccess) { mv64_free(m); ctx->err = std::string("narrow matrix: ") + hipGetErrorString(e2); return HB_ERR_HIP; }
        rc = upload_table(ctx, m->a8, dig.data(), dig.size(), s);
        if (!rc) rc = upload_table(ctx, m->crow, cr.data(), cr.size() * 8, s);
        if (rc) { mv64_free(m); return rc; }
        m->nkb = nkb; m->n_rt = n_rt;
    }
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
    if (m->a8 && !env_hook(ENV_NO_MFMA)) {
        const int64_t n_items = ((C + 15) / 16) * ((m->n_rt + 1) / 2);
        static int cus = 0;
        if (!cus) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256; }
        const unsigned blocks = (unsigned)std::min<int64_t>((n_items + 3) / 4, (int64_t)2 * cus);      // two workgroups of four waves a CU: two waves a SIMD
        const size_t lds = (size_t)m->n_rt * m->nkb * 64 * 16 + (size_t)m->n_rt * 16 * (8 + 8 + 4);
#define MV64M_LAUNCH(NKB) k_mv64m<NKB><<<blocks, 256, lds, s>>>(m->a8, m->crow, m->mode, m->n_out, m->n_rt, m->d, in, iv.stride_c, iv.stride_l, in_rows_dev, in_count, out, \
                                                              ov.stride_c, ov.stride_l, out_count, mismatch_dev, first_bad_dev, C, n_items, prm)
        if (m->nkb == 1) MV64M_LAUNCH(1); else if (m->nkb == 2) MV64M_LAUNCH(2); else MV64M_LAUNCH(3);
#undef MV64M_LAUNCH
        HB_LAUNCH_CHECK(ctx);
        return HB_OK;
    }
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
