// hb_quick.hip -- the robust path of the device decoder without plans, tables built on the host, or synchronisation.
//
// Reference: IncrementalDecoder (honeybadgermpc/reed_solomon.py:232-403).  Every arrival set is new to a decoder that is
// probing its way past liars, so whatever it needs per arrival set must cost microseconds, not the ~2 ms of an open plan
// (hb_open.hip) or the ~0.7 ms of an n' x n' inverse:
//
//   * hb_quick_interp_check: decoder.decode_batch over the first d arrivals + encoder.encode_batch + the compare loop
//     (reed_solomon.py:305-326) for ANY (z, zc) as ONE launch of the full-size matrix-core kernel (hb_mfma_wide.hip) over
//     [V^-1(z) ; V[zc] V^-1(z)].  The matrix is built ON THE DEVICE from a per-point-set table of 1 / (x_a - x_b)
//     (Lagrange: row m of V^-1 is the m-th coefficient of L_j(X) = prod_{q != j} (X - x_q) / (x_j - x_q); the prediction at
//     a later arrival is L_j(x_i)), turned into the kernel's int8 digit image by a second small kernel, and launched --
//     three small launches and the big one, all enqueued, none waited for.
//
//   * hb_probe_*: gao_interpolate for ONE codeword (rsdecode_impl.h:325-363, what robust_decode runs per polynomial,
//     reed_solomon.py:334-365), incremental in the points: Koetter / Welch-Berlekamp rational interpolation keeps two
//     polynomial pairs Q_j = (A_j, B_j) with A_j(x_i) + y_i B_j(x_i) = 0 on every point fed so far, a reduced basis of that
//     module for the (1, k-1)-weighted degree; a new point costs O(n') multiplications done by one workgroup in parallel.
//     The decision is the reference's exactly (scratch/koetter_check.py, tests/test_probe_rule.py: 12 073 prefix decisions
//     against the oracle, 118 of them beyond the unique-decoding radius): with Q_0 the pair whose leading term is in A and
//     T = (n' + k) / 2 (integer division, as partial_gcd's stopping rule, rsdecode_impl.h:281-323)
//         Gao decodes  <=>  deg A_0 >= T  and  B_1 has deg B_1 distinct roots among the points fed
//     (Q_1 first reduced against Q_0 to deg A_1 < T; B_1 then divides A_1 because A_1 + y B_1 vanishes on every point, and
//     conversely the locator of a decodable word is exactly the product over its errors), and those roots are the erroneous
//     senders (reed_solomon.py:174-184).  No division anywhere, of field elements or of polynomials: the updates are fraction-free.
#include <algorithm>
#include <atomic>
#include <chrono>

#include "hb_common.hpp"

using namespace hb;

namespace hb {

constexpr int QUICK_MAX = 128;       // interpolation points (the kernel's inner dimension)
constexpr int QUICK_MAXC = 256;      // compared rows
constexpr int PROBE_MAXN = 256;      // party points of a probe
constexpr int PROBE_NT = 1024;       // threads of the probe's workgroup: one item of a point's update each (2 (n + 3) coefficients + 2 n values)

struct QuickIdx { uint16_t z[QUICK_MAX], zc[QUICK_MAXC]; };
struct ProbeIdx { uint16_t idx[PROBE_MAXN]; };

template <int NL> __device__ __forceinline__ void ldg(uint32_t (&r)[NL], const uint32_t *p) {
#pragma unroll
    for (int i = 0; i < NL; i++) r[i] = p[i];
}
template <int NL> __device__ __forceinline__ void stg(uint32_t *p, const uint32_t (&r)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; i++) p[i] = r[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// per point set: x (Montgomery), the powers x_a^i (the probe evaluates its polynomials with them) and 1 / (x_a - x_b)
// ---------------------------------------------------------------------------------------------------------------------
template <int NL, int NW>
__global__ void k_pt_mont(const FpParams<NL> P, const uint32_t *__restrict__ x_pk, int n, uint32_t *__restrict__ xm, uint32_t *__restrict__ pw, int S) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    uint32_t xd[NL], x[NL], acc[NL];
    load_digits<NL, NW>(xd, x_pk + (size_t)a * NW);
    to_mont(x, xd, P);
    stg<NL>(xm + (size_t)a * NL, x);
    fp_set(acc, P.one);
    for (int i = 0; i < S; i++) {
        stg<NL>(pw + ((size_t)a * S + i) * NL, acc);
        mont_mul(acc, acc, x, P);
    }
}
// row a of the table by one inversion: prefix products forward, the inverse of the whole product walked back
template <int NL>
__global__ void k_pt_inv(const FpParams<NL> P, const uint32_t *__restrict__ xm, int n, uint32_t *__restrict__ inv, int32_t *__restrict__ bad) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n) return;
    uint32_t xa[NL], pref[NL], df[NL], xb[NL];
    ldg<NL>(xa, xm + (size_t)a * NL);
    fp_set(pref, P.one);
    uint32_t *row = inv + (size_t)a * n * NL;
    for (int b = 0; b < n; b++) {
        if (b == a) continue;
        ldg<NL>(xb, xm + (size_t)b * NL);
        fp_sub(df, xa, xb, P);
        if (fp_is_zero(df)) { atomicOr(bad, 1); fp_set(df, P.one); }       // a repeated point: the table is unusable, say so
        stg<NL>(row + (size_t)b * NL, pref);
        mont_mul(pref, pref, df, P);
    }
    uint32_t ip[NL];
    fp_inv(ip, pref, P);
    for (int b = n - 1; b >= 0; b--) {
        if (b == a) continue;
        ldg<NL>(xb, xm + (size_t)b * NL);
        fp_sub(df, xa, xb, P);
        if (fp_is_zero(df)) fp_set(df, P.one);
        uint32_t t[NL], v[NL];
        ldg<NL>(t, row + (size_t)b * NL);
        mont_mul(v, ip, t, P);
        stg<NL>(row + (size_t)b * NL, v);
        mont_mul(ip, ip, df, P);
    }
    uint32_t z[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) z[i] = 0;
    stg<NL>(row + (size_t)a * NL, z);
}

// ---------------------------------------------------------------------------------------------------------------------
// [V^-1(z) rows ; V[zc] V^-1(z)] on the device
// ---------------------------------------------------------------------------------------------------------------------
// One workgroup of 512, three roles in disjoint thread ranges, each packed densely into as few waves as it needs (a wave
// instruction costs the same for one active lane as for 64, and v_mad_u64_u32 is pipe-bound: spreading a few dozen chains over
// many waves of ONE CU only makes them queue behind each other -- measured 450-620 us that way):
//   [0, 128)    A(X) = prod_q (X - x_zq) in LDS, then per arrival j the coefficients of N_j = A / (X - x_zj) (synthetic division)
//   [128, 256)  per arrival j: w_j = prod_{q != j} 1 / (x_zj - x_zq)
//   [256, 512)  per compared row i: full_i = prod_q (x_zci - x_zq); and the row index arrays of the big launch
// Up to 63 interpolation points the first role is ONE wave working through LDS with wave-level synchronisation only, and the three
// roles run side by side on three SIMDs: the critical path is the 2 d dependent multiplications of the first role.  Beyond that
// A is built with workgroup barriers first and the roles follow.
constexpr int QM_NT = 1024, QM_SEG = 8, QM_FSEG = 16, QM_FPB = 32, QM_TS = 4;      // (QM_FPB rows x QM_FSEG partial products = 512 threads of a row-product workgroup: two waves a SIMD)     // k_quick_matrix: threads a workgroup; Horner / w_j segments (d QM_SEG <= QM_NT); partial products of a full_i (nc QM_FSEG <= QM_NT); lanes per coefficient of a tree level (d QM_TS <= QM_NT)
template <int NL>
__global__ void __launch_bounds__(QM_NT) k_quick_matrix(const FpParams<NL> P, const uint32_t *__restrict__ xm, const uint32_t *__restrict__ inv, int n,
                                                      const QuickIdx ix, int d, int nc, int n_coef, uint32_t *__restrict__ wj, uint32_t *__restrict__ full,
                                                      uint32_t *__restrict__ nraw, int32_t *__restrict__ z_dev, int32_t *__restrict__ fmap, int flags,
                                                      uint4 *__restrict__ zero_base, int zero_q, uint4 *__restrict__ zero2_base, int zero2_q,
                                                      int nh, int nw, int nf, const uint32_t *__restrict__ pwt, int S) {
    // Three workgroups, each a set of SHORT chains of dependent multiplications (rounds 3 and 4 ran one workgroup of d-step chains: A(X) by
    // d sequential multiplications by (X - x_q) with two barriers each, then d-step Horner / product chains per thread -- 177 us at d = 86):
    //   block 0: A(X) = prod (X - x_q) by a PRODUCT TREE in LDS (log2 d levels; a level multiplies adjacent monic polynomials, one output
    //            coefficient a thread: a lazily accumulated dot product and ONE reduction), then the N_j by Horner cut into QM_SEG segments
    //            (segment sums in parallel, their combination through x_j^Ls, the segments' own steps in parallel: 2 d / QM_SEG + QM_SEG steps);
    //   block 1: w_j = prod_{q != j} 1 / (x_j - x_q), QM_SEG partial products a row, QM_SEG - 1 multiplications to join them; the arrival list;
    //   block 2: full_i = prod_q (x_i - x_q) for the compared senders, QM_FSEG partial products each; the row map.
    // flags: QUICK_Z -- what depends on the arrivals z alone (A, the N_j, the w_j, the row map of the coefficient rows); QUICK_ZC -- the
    // compared senders' full_i and their rows of the map.  A decoder builds the first half when its (degree+1)-th column lands.
    //   block 3 (first half only): the image and the row constants start from zero (padding rows and terms) -- a hipMemsetAsync before the
    //            launch cost a dispatch of its own (5 us on the path of a first-sight decode).
    // flags & 4 (with QUICK_Z): block 2 computes full_i for EVERY party (a decoder's candidate store: k_quick_fill makes a row of each), not
    // for the compared senders, who are not known yet.
    // Round 5: the chains of ONE workgroup are issue-bound on their CU once there are a thousand of them (config 5's shard: 256 parties x 86
    // factors of full_i = 49 us of multiply-adds on one CU; 86 x 8 Horner segments = 29 us).  So: workgroups 0 .. nh - 1 share the N_j (each
    // builds A(X) for itself: round 6, below -- no workgroup of the launch waits for another); workgroup nh
    // is the w_j; workgroups nh + 1 .. nh + nf take QM_FPB parties' full_i each (QM_FSEG partial products joined by a tree); the last one zeroes.
    extern __shared__ uint32_t q_lds[];
    const int tid = threadIdx.x;
    const bool do_z = flags & 1, do_zc = flags & 2, do_cand = flags & 4;
    uint32_t *xz = q_lds;                               // [d][NL]: the arrivals' points (every block's chains read them)
    const int b_w = nh, b_f0 = nh + nw, b_zero = nh + nw + nf;
    if ((int)blockIdx.x == b_zero) {
        for (int i = tid; i < zero_q; i += QM_NT) zero_base[i] = make_uint4(0, 0, 0, 0);
        for (int i = tid; i < zero2_q; i += QM_NT) zero2_base[i] = make_uint4(0, 0, 0, 0);      // (the candidate rows' padding terms: d .. 8 nkb - 1)
        return;
    }
    if ((int)blockIdx.x < b_f0 && !do_z) return;
    for (int e = tid; e < d * NL; e += QM_NT) xz[e] = xm[(size_t)ix.z[e / NL] * NL + e % NL];
    __syncthreads();
    // A row product: out[r] = prod_{q < d, q != skip(r)} factor(r, q), QM_FSEG partial products a row joined by a tree through LDS, QM_FPB rows a
    // workgroup (512 threads: a step of dependent multiplications costs what its waves issue, so the rows are spread thin).  Three users:
    // the constant terms N_j[0] = prod_{q != j} (-x_q) (R1 decoders: no A(X), no Horner), the w_j, the full_i.
    auto row_products = [&](int rb, int rows, int kind, uint32_t *__restrict__ out) {
        uint32_t *Fp = xz + (size_t)d * NL;                                   // [QM_FPB][QM_FSEG][NL]
        const int il = tid / QM_FSEG, g = tid % QM_FSEG, part = (d + QM_FSEG - 1) / QM_FSEG;
        const int r = rb * QM_FPB + il;
        const bool act = il < QM_FPB && r < rows;
        if (act) {
            const int lo = g * part, hi = min((g + 1) * part, d);
            uint32_t xi[NL], f[NL];
            fp_set(f, P.one);
            const uint32_t *irow = nullptr;
            if (kind == 2) ldg<NL>(xi, xm + (size_t)(do_cand ? r : (int)ix.zc[r]) * NL);
            if (kind == 1) irow = inv + (size_t)ix.z[r] * n * NL;
            for (int q = lo; q < hi; q++) {
                uint32_t fac[NL];
                if (kind == 0) { uint32_t xq[NL]; ldg<NL>(xq, xz + (size_t)q * NL); fp_neg(fac, xq, P); }
                else if (kind == 1) ldg<NL>(fac, irow + (size_t)ix.z[q] * NL);
                else { uint32_t xq[NL]; ldg<NL>(xq, xz + (size_t)q * NL); fp_sub(fac, xi, xq, P); }
                if (kind == 2 || q != r) mont_mul(f, f, fac, P);
            }
            stg<NL>(Fp + ((size_t)il * QM_FSEG + g) * NL, f);
        }
        __syncthreads();
        for (int half = QM_FSEG / 2; half >= 1; half >>= 1) {
            if (act && g < half) {
                uint32_t f[NL], f1[NL];
                ldg<NL>(f, Fp + ((size_t)il * QM_FSEG + g) * NL);
                ldg<NL>(f1, Fp + ((size_t)il * QM_FSEG + g + half) * NL);
                mont_mul(f, f, f1, P);
                if (half == 1) stg<NL>(out + (size_t)r * NL, f);
                else stg<NL>(Fp + ((size_t)il * QM_FSEG + g) * NL, f);
            }
            __syncthreads();
        }
    };
    if ((int)blockIdx.x < nh && n_coef == 1) {
        // only the constant terms are wanted (what R1 forwards): N_j[0] = prod_{q != j} (-x_q) -- workgroups 0 .. nh - 1 take QM_FPB arrivals each
        row_products((int)blockIdx.x, d, 0, nraw);
        return;
    }
    if ((int)blockIdx.x < nh) {
        uint32_t *B0 = xz + (size_t)d * NL, *B1 = B0 + (size_t)d * NL;      // [d][NL] each: the level's monic polynomials without their leading 1
        uint32_t *Ac = B1 + (size_t)d * NL;                                   // [d + 1][NL]
        uint32_t *Tb = Ac + (size_t)(d + 1) * NL;                             // [d][QM_SEG][NL]: segment sums
      // EVERY workgroup of the Horner phase builds A(X) itself, by the same product tree in its own LDS.  Rounds 5-6 had block 0 build it and
      // publish it through global memory, a token telling the helpers when: they waited as long as the tree takes anyway (measured with every
      // helper building its own: config 5's first-sight open 379.8 -> 374.5 us), and no workgroup of this launch depends on another any more.
      {
        if (tid < d) {
            uint32_t x[NL], nx[NL];
            ldg<NL>(x, xz + (size_t)tid * NL);
            fp_neg(nx, x, P);
            stg<NL>(B0 + (size_t)tid * NL, nx);                              // X - x_q
        }
        __syncthreads();
        uint32_t *src = B0, *dst = B1;
        for (int len = 1; len < d; len <<= 1) {
            // polynomial i of this level: stored coefficients [i len, min((i + 1) len, d)); pair (2 i', 2 i' + 1) -> [2 i' len, ...) of the next.
            // QM_TS adjacent lanes share an output coefficient: each takes every QM_TS-th term of its dot product (up to 64 terms at the
            // last level), the carried columns are summed across the lanes, the first of them reduces and stores
            const int o = tid / QM_TS, part = tid % QM_TS;
            if (o < d) {                                                     // (whole groups of QM_TS lanes: the shuffles below stay inside a group)
                const int base = (o / (2 * len)) * 2 * len, c = o - base;
                const int m = min(len, d - base), k = min(len, d - base - m);
                uint32_t r[NL];
                if (k == 0) ldg<NL>(r, src + (size_t)o * NL);               // no partner: carried over
                else {
                    // (X^m + p)(X^k + q) = X^(m+k) + [p q + X^m q + X^k p]: coefficient c of the bracket
                    const uint32_t *pp = src + (size_t)base * NL, *qq = pp + (size_t)m * NL;
                    uint64_t col[2 * NL];
                    col_zero(col);
                    int cnt = 0;
                    for (int a_ = max(0, c - k + 1) + part; a_ <= min(m - 1, c); a_ += QM_TS) {
                        uint32_t u[NL], v[NL];
                        ldg<NL>(u, pp + (size_t)a_ * NL);
                        ldg<NL>(v, qq + (size_t)(c - a_) * NL);
                        mac<NL>(col, u, v);
                        if ((++cnt & 3) == 0) carry(col);                    // four products of canonical elements fit a column
                    }
                    carry(col);                                              // columns below 2^29 (the top one below 2^36): QM_TS of them add up
#pragma unroll
                    for (int q = 0; q < 2 * NL; q++) {
#pragma unroll
                        for (int sh = 1; sh < QM_TS; sh <<= 1) {
                            const uint32_t lo_ = (uint32_t)__shfl_xor((int)(uint32_t)col[q], sh), hi_ = (uint32_t)__shfl_xor((int)(uint32_t)(col[q] >> 32), sh);
                            col[q] += ((uint64_t)hi_ << 32) | lo_;
                        }
                    }
                    finish<NL>(r, col, P, 1);                                // at most 64 products below p^2: < 2 p before the subtraction
                    if (c >= m) { uint32_t v[NL]; ldg<NL>(v, qq + (size_t)(c - m) * NL); fp_add(r, r, v, P); }
                    if (c >= k) { uint32_t u[NL]; ldg<NL>(u, pp + (size_t)(c - k) * NL); fp_add(r, r, u, P); }
                }
                if (part == 0) stg<NL>(dst + (size_t)o * NL, r);
            }
            __syncthreads();
            uint32_t *t_ = src; src = dst; dst = t_;
        }
        if (tid < d) { uint32_t v[NL]; ldg<NL>(v, src + (size_t)tid * NL); stg<NL>(Ac + (size_t)tid * NL, v); }
        if (tid == 0) stg<NL>(Ac + (size_t)d * NL, P.one);
        __syncthreads();
      }
        // N(m) := N_j[m - 1] = sum_{k >= m} A[k] x_j^(k - m), m = 1 .. d; segment g holds k in [lo, hi); this workgroup's share of the j
        const int Ls = (d + QM_SEG - 1) / QM_SEG;
        const int jb = (d + nh - 1) / nh;
        const int jl = tid / QM_SEG, g = tid % QM_SEG;
        const int j = jl < jb ? (int)blockIdx.x * jb + jl : d;
        const int lo = 1 + g * Ls, hi = min(1 + (g + 1) * Ls, d + 1);
        const bool live = j < d && lo < hi;
        uint32_t xj[NL], y[NL], v[NL];
        fp_set(y, P.one);
#pragma unroll
        for (int q = 0; q < NL; q++) v[q] = 0;
        if (j < d) {
            ldg<NL>(xj, xz + (size_t)j * NL);
            for (int k = hi - 1; k >= lo; k--) {                             // T_g = sum_{k in segment} A[k] x^(k - lo), and y = x^Ls beside it
                uint32_t a_[NL], pr[NL];
                ldg<NL>(a_, Ac + (size_t)k * NL);
                mont_mul(pr, xj, v, P);
                fp_add(v, a_, pr, P);
            }
            ldg<NL>(y, pwt + ((size_t)ix.z[j] * S + Ls) * NL);             // x_j^Ls from the point set's table of powers (Ls <= d <= n < S)
            stg<NL>(Tb + ((size_t)j * QM_SEG + g) * NL, v);
        }
        __syncthreads();
        if (live) {
            // S = N(hi) = T_{g+1} + y (T_{g+2} + y (...)): the segments above, top first (a segment below the top one is Ls long)
            uint32_t S[NL];
#pragma unroll
            for (int q = 0; q < NL; q++) S[q] = 0;
            for (int g2 = QM_SEG - 1; g2 > g; g2--) {
                uint32_t t2[NL], pr[NL];
                ldg<NL>(t2, Tb + ((size_t)j * QM_SEG + g2) * NL);
                mont_mul(pr, y, S, P);
                fp_add(S, t2, pr, P);
            }
            for (int k = hi - 1; k >= lo; k--) {                             // N(k) = A[k] + x N(k + 1)
                uint32_t a_[NL], pr[NL];
                ldg<NL>(a_, Ac + (size_t)k * NL);
                mont_mul(pr, xj, S, P);
                fp_add(S, a_, pr, P);
                if (k - 1 < n_coef) stg<NL>(nraw + ((size_t)(k - 1) * d + j) * NL, S);
            }
        }
        return;
    }
    if ((int)blockIdx.x >= b_w && (int)blockIdx.x < b_f0) {
        // w_j = prod_{q != j} 1 / (x_j - x_q) from the point set's table of inverse differences; the first of these workgroups also the arrival list
        if ((int)blockIdx.x == b_w && tid < d) z_dev[tid] = ix.z[tid];
        row_products((int)blockIdx.x - b_w, d, 1, wj);
        return;
    }
    // workgroups b_f0 ..: QM_FPB rows' full_i each (QM_FSEG partial products a row, joined by a tree through LDS); the first one also the row map
    {
        const int fb = (int)blockIdx.x - b_f0;
        if (fb == 0) {
            if (do_z) { for (int r = tid; r <= n_coef + nc; r += QM_NT) fmap[r] = (do_zc && r >= n_coef && r < n_coef + nc) ? (int32_t)ix.zc[r - n_coef] + 1 : 0; }
            else { for (int r = tid; r < nc; r += QM_NT) fmap[n_coef + r] = (int32_t)ix.zc[r] + 1; }
        }
        if (!do_zc && do_z && !do_cand) return;
        row_products(fb, do_cand ? n : nc, 2, full);
    }
}

// one thread per matrix entry: its canonical value and its 32 balanced base-256 digits at their place in the int8 image
// (the layout of mm8w_from_host, hb_mfma_wide.hip)
__global__ void k_quick_image(const FpParams<9> P, const uint32_t *__restrict__ inv, int n, const QuickIdx ix, int d, int nc, int n_coef,
                              const uint32_t *__restrict__ wj, const uint32_t *__restrict__ full, const uint32_t *__restrict__ nraw,
                              uint32_t *__restrict__ mcan, uint8_t *__restrict__ a8, int tile_rows, int nkb, int row_lo, int row_hi) {
    constexpr int NL = 9, NW = 8;
    const int e = blockIdx.x * blockDim.x + threadIdx.x + row_lo * d;
    if (e >= row_hi * d) return;
    const int i = e / d, l = e - i * d;
    uint32_t w[NL], v[NL];
    ldg<NL>(w, wj + (size_t)l * NL);
    if (i < n_coef) {
        uint32_t c[NL];
        ldg<NL>(c, nraw + ((size_t)i * d + l) * NL);
        mont_mul(v, c, w, P);
    } else {
        const int r = i - n_coef;
        uint32_t f[NL], g[NL], t[NL];
        ldg<NL>(f, full + (size_t)r * NL);
        ldg<NL>(g, inv + ((size_t)ix.zc[r] * n + ix.z[l]) * NL);
        mont_mul(t, f, g, P);
        mont_mul(v, t, w, P);
    }
    uint32_t cd[NL], cw[NW];
    from_mont(cd, v, P);
    pack<NL, NW>(cw, cd);
    store_words<NW>(mcan + (size_t)e * NW, cw);
    const int rt = i / tile_rows, j16 = i % tile_rows, r = 4 * (j16 % 4) + j16 / 4, kb = l / 8, g = (l % 8) / 2, el = l & 1;
    int carry = 0;
#pragma unroll
    for (int b = 0; b < 32; b++) {
        int tt = (int)((cw[b >> 2] >> (8 * (b & 3))) & 0xffu) + carry;
        if (tt > 127) { tt -= 256; carry = 1; } else carry = 0;
        const int grp = b >> 3, r7 = 7 - (b & 7), hi = r7 >> 2, bi = r7 & 3;
        a8[((((size_t)rt * nkb + kb) * 4 + grp) * 64 + (size_t)(r + 16 * g)) * 16 + 4 * (2 * hi + el) + bi] = (uint8_t)(int8_t)tt;
    }
}

// per row: the constant that takes the XOR-0x80 input bias and the accumulator bias back out,
// (0x80..80 * sum_l M[i][l] - bias * sum_c 2^(8c)) mod p, as 9 digits in the kernel's row-constant slot
struct QuickRowConst { uint32_t c80r[9], biasmod[9]; };
__global__ void k_quick_rows(const FpParams<9> P, const uint32_t *__restrict__ mcan, int row_lo, int row_hi, int d, const QuickRowConst rc,
                             uint32_t *__restrict__ crow, int tile_rows) {
    constexpr int NL = 9, NW = 8;
    const int i = blockIdx.x * blockDim.x + threadIdx.x + row_lo;
    if (i >= row_hi) return;
    uint32_t sum[NL];
#pragma unroll
    for (int q = 0; q < NL; q++) sum[q] = 0;
    for (int l = 0; l < d; l++) {
        uint32_t e[NL];
        load_digits<NL, NW>(e, mcan + ((size_t)i * d + l) * NW);
        fp_add(sum, sum, e, P);
    }
    uint32_t k[NL], b[NL], prod[NL], corr[NL];
#pragma unroll
    for (int q = 0; q < NL; q++) { k[q] = rc.c80r[q]; b[q] = rc.biasmod[q]; }
    mont_mul(prod, sum, k, P);                  // sum * (0x80..80 R) / R
    fp_sub(corr, prod, b, P);
    uint32_t *dst = crow + ((size_t)(i / tile_rows) * 16 + i % tile_rows) * 16;
#pragma unroll
    for (int q = 0; q < NL; q++) dst[q] = corr[q];
}

// A decoder's first half in ONE launch behind k_quick_matrix: block b < n_coef makes coefficient row b of the image (entry, canonical value, 32
// balanced digits at their place, and the row's constant by a tree of additions over the block); block n_coef + i makes the row party i WOULD
// contribute as a compared sender -- (V[i] V^-1)[l] = full_i w_l / (x_i - x_zl) -- into the candidate store: its digit pieces in image order and
// its row constant.  Parties among the arrivals have no such row (they are never compared): their blocks leave at once.
// (k_quick_image + k_quick_rows: two launches, the row constants one thread a row over d sequential loads: 33 us at d = 86.)
constexpr int QF_NT = 128;
__global__ void __launch_bounds__(QF_NT) k_quick_fill(const FpParams<9> P, const uint32_t *__restrict__ inv, int n, const QuickIdx ix, int d, int n_coef,
                                                     const uint32_t *__restrict__ wj, const uint32_t *__restrict__ full, const uint32_t *__restrict__ nraw,
                                                     uint8_t *__restrict__ a8, uint32_t *__restrict__ crow, int tile_rows, int nkb, const QuickRowConst rc,
                                                     uint8_t *__restrict__ cand, uint32_t *__restrict__ cand_crow, int direct) {
    constexpr int NL = 9, NW = 8;
    __shared__ uint32_t red[QF_NT][NL];
    __shared__ int among;
    // direct: the compared senders are known (a plan, a robust-phase launch): block n_coef + j makes row n_coef + j of the image from sender
    // zc[j] -- full[] is indexed by j then -- instead of a candidate row per party
    const int b = blockIdx.x, l = threadIdx.x;
    const bool coef = b < n_coef, into_image = coef || direct;
    const int party = coef ? 0 : (direct ? (int)ix.zc[b - n_coef] : b - n_coef);
    if (!coef && !direct) {
        if (l == 0) among = 0;
        __syncthreads();
        if (l < d && ix.z[l] == party) among = 1;
        __syncthreads();
        if (among) return;
    }
    uint32_t cd[NL];
#pragma unroll
    for (int q = 0; q < NL; q++) cd[q] = 0;
    if (l < d) {
        uint32_t w[NL], v[NL];
        ldg<NL>(w, wj + (size_t)l * NL);
        if (coef) {
            uint32_t c[NL];
            ldg<NL>(c, nraw + ((size_t)b * d + l) * NL);
            mont_mul(v, c, w, P);
        } else {
            uint32_t f[NL], g[NL], t[NL];
            ldg<NL>(f, full + (size_t)(direct ? b - n_coef : party) * NL);
            ldg<NL>(g, inv + ((size_t)party * n + ix.z[l]) * NL);
            mont_mul(t, f, g, P);
            mont_mul(v, t, w, P);
        }
        uint32_t cw[NW];
        from_mont(cd, v, P);
        pack<NL, NW>(cw, cd);
        const int kb = l / 8, g = (l % 8) / 2, el = l & 1;
        // where the row's 16-byte piece (kb, grp, g) lies: in the image at lane position r + 16 g of its row tile, in the store at piece index
        size_t base4[4];
        if (into_image) {
            const int rt = b / tile_rows, j16 = b % tile_rows, r = 4 * (j16 % 4) + j16 / 4;
#pragma unroll
            for (int grp = 0; grp < 4; grp++) base4[grp] = ((((size_t)rt * nkb + kb) * 4 + grp) * 64 + (size_t)(r + 16 * g)) * 16;
        } else {
#pragma unroll
            for (int grp = 0; grp < 4; grp++) base4[grp] = ((size_t)party * nkb * 16 + (size_t)((kb * 4 + grp) * 4 + g)) * 16;
        }
        uint8_t *dst = into_image ? a8 : cand;
        int carry = 0;
#pragma unroll
        for (int bb = 0; bb < 32; bb++) {
            int tt = (int)((cw[bb >> 2] >> (8 * (bb & 3))) & 0xffu) + carry;
            if (tt > 127) { tt -= 256; carry = 1; } else carry = 0;
            const int grp = bb >> 3, r7 = 7 - (bb & 7), hi = r7 >> 2, bi = r7 & 3;
            dst[base4[grp] + 4 * (2 * hi + el) + bi] = (uint8_t)(int8_t)tt;
        }
    }
    // the row constant: (0x80..80 * sum_l M[l] - bias * sum_c 2^(8c)) mod p -- the sum over the block by halving
#pragma unroll
    for (int q = 0; q < NL; q++) red[l][q] = cd[q];
    __syncthreads();
    for (int half = QF_NT / 2; half >= 1; half >>= 1) {
        if (l < half) {
            uint32_t a[NL], c[NL];
#pragma unroll
            for (int q = 0; q < NL; q++) { a[q] = red[l][q]; c[q] = red[l + half][q]; }
            fp_add(a, a, c, P);
#pragma unroll
            for (int q = 0; q < NL; q++) red[l][q] = a[q];
        }
        __syncthreads();
    }
    if (l == 0) {
        uint32_t sum[NL], k[NL], bm[NL], prod[NL], corr[NL];
#pragma unroll
        for (int q = 0; q < NL; q++) { sum[q] = red[0][q]; k[q] = rc.c80r[q]; bm[q] = rc.biasmod[q]; }
        mont_mul(prod, sum, k, P);
        fp_sub(corr, prod, bm, P);
        uint32_t *dstc = into_image ? crow + ((size_t)(b / tile_rows) * 16 + b % tile_rows) * 16 : cand_crow + (size_t)party * 16;
#pragma unroll
        for (int q = 0; q < NL; q++) dstc[q] = corr[q];
    }
}

// ... and the second half: the rows of the senders that DID become compared ones, copied from the candidate store to rows n_coef .. of the
// image, with their constants and their entries of the row map (one small launch behind the column that completes the quorum)
__global__ void k_quick_pick(const QuickIdx ix, int nc, int n_coef, int tile_rows, int nkb, const uint4 *__restrict__ cand, const uint32_t *__restrict__ cand_crow,
                             uint4 *__restrict__ a8, uint32_t *__restrict__ crow, int32_t *__restrict__ fmap) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = nkb * 16;
    if (e < nc * per) {
        const int j = e / per, pc = e - j * per;
        const int i = n_coef + j, rt = i / tile_rows, j16 = i % tile_rows, r = 4 * (j16 % 4) + j16 / 4;
        const int g = pc & 3, kg = pc >> 2;                 // kg = kb * 4 + grp
        a8[((size_t)rt * nkb * 4 + kg) * 64 + (size_t)(r + 16 * g)] = cand[(size_t)ix.zc[j] * per + pc];
    }
    if (e < nc * 9) {
        const int j = e / 9, q = e - j * 9, i = n_coef + j;
        crow[((size_t)(i / tile_rows) * 16 + i % tile_rows) * 16 + q] = cand_crow[(size_t)ix.zc[j] * 16 + q];
    }
    if (e < nc) fmap[n_coef + e] = (int32_t)ix.zc[e] + 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// the probe: one workgroup, the four polynomials A_0, B_0, A_1, B_1 in LDS
// ---------------------------------------------------------------------------------------------------------------------
struct ProbeResult { int32_t ok, n_err, npts, seq; uint8_t err[PROBE_MAXN]; };    // seq: written last, the host polls it

// A probe over several workgroups (point sets above PROBE_SPLIT_N points): workgroup 0 holds the coefficients, the others a slice of the
// value table each.  What they tell each other, per point fed, through global memory (a slot per point of the launch, so nobody overwrites
// what a slower workgroup has not read).  Every word of a message travels as 64 bits, the point's sequence number above its 32 bits of
// payload: a reader polls the word it needs and knows from the word itself that it is this point's -- no flag to wait for before the
// payload may be read, no fence between them, one trip through the L2 per message:
//   the discrepancies of the point (from the workgroup that holds the values at that party): 2 NL digits + 2 flags    [PROBE_DW words]
//   the degrees after the point (from workgroup 0: a leading coefficient may cancel, only the coefficients show it)    [PROBE_GW words]
// and, at the end of a launch, "my slice is back in the state" from every value workgroup (PROBE_MAXG words, flag + fence).
constexpr int PROBE_SPLIT_N = 128;
constexpr int PROBE_MAXG = 8;
constexpr int PROBE_DW = 40, PROBE_GW = 8;
constexpr size_t PROBE_MSG_WORDS = (size_t)PROBE_MAXN * (PROBE_DW + PROBE_GW) + PROBE_MAXG + 8;      // (+ the abort word)
constexpr size_t PROBE_STATE_WORDS = (size_t)4 * (2 * PROBE_MAXN + 2) * 9 + 8 + PROBE_MAXN;      // the largest state: coefficients + values of 256 points, degrees, the fed list
// (every wait is bounded: a workgroup that never shows up -- a launch that lost a workgroup, a bug -- must not hang the device.  ~2^22 polls
// of >= 64 cycles are tenths of a second where a message takes microseconds; the waiter then raises the launch's abort word, which the
// deciding workgroup hands to the host as a failed launch)
constexpr int PROBE_SPIN_MAX = 1 << 22;
__device__ __forceinline__ void probe_wait(const uint32_t *flag, uint32_t want, uint32_t *abort_word, uint32_t abort_value) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > PROBE_SPIN_MAX) { __hip_atomic_store(abort_word, abort_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    __threadfence();
}
__device__ __forceinline__ void probe_post(uint32_t *flag, uint32_t value) {
    __threadfence();
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void probe_send(uint32_t *slot, int word, uint32_t q, uint32_t payload) {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(slot) + word, ((unsigned long long)q << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t probe_recv(const uint32_t *slot, int word, uint32_t q, uint32_t *abort_word, uint32_t abort_value) {
    const unsigned long long *w = reinterpret_cast<const unsigned long long *>(slot) + word;
    unsigned long long v;
    int spins = 0;
    while ((uint32_t)((v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != q) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > PROBE_SPIN_MAX) { __hip_atomic_store(abort_word, abort_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    return (uint32_t)v;
}

template <int NL> __device__ __forceinline__ bool lds_nonzero(const uint32_t *p) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < NL; i++) o |= p[i];
    return o != 0;
}

// The state of a probe: the four polynomials by their COEFFICIENTS (the decision reads degrees and the locator's roots off them) and by
// their VALUES at all n party points.  With the values at hand the discrepancy of a new point is a look-up, A_j(x_a) + y_a B_j(x_a) (round
// 3 evaluated the polynomials at it: four multiplications and a 64-lane reduction, 9 us a point at n = 64, 20 at n = 256); the update
//     Q_jo <- d_js Q_jo - d_jo Q_js,     Q_js <- (X - x_a) Q_js
// acts on values pointwise (val_js[i] <- (x_i - x_a) val_js[i]) and on coefficients as before.  A point's critical path, as round 5 left it:
//   * the discrepancies of ALL the points of a launch are formed up front and carried along by every update like any other value: the
//     pivot of a point is picked from a look-up, no multiplication after the previous point's update;
//   * an item (a coefficient, a value, a pending discrepancy) is two UNITS for two threads -- Q_jo's two products and one reduction, Q_js's one
//     product -- every unit reads the state as the last point left it, results wait in registers until all have read;
//   * the degrees after a point follow from the degrees before it; the thread that computes a leading coefficient that may cancel (equal
//     degrees of the two pairs) looks, and only then are the degrees found by a scan.
//
// One workgroup does all of it up to PROBE_SPLIT_N points (4.0 us a point at n = 64).  Above (config 5's 256 parties: 2 (n + 3) coefficient
// items, 2 n value items and the pending discrepancies are 3+ waves a SIMD of ONE CU, 8.4 us a point) the launch is G workgroups -- the grid is
// 8 (G - 1) + 1 and only the workgroups whose index is a multiple of 8 stay, so that they sit on ONE XCD and talk through its L2 -- and a
// point is: the workgroup holding the values at the new party posts the two discrepancies; everybody picks the pivot from them and the
// degrees and updates its own items; the workgroups holding coefficients post the degrees that result (3.8 us a point with six).
template <int NL, int NW>
__global__ void __launch_bounds__(PROBE_NT) k_probe_feed(const FpParams<NL> P, const uint32_t *__restrict__ xm, const uint32_t *__restrict__ pw, int n, int S,
                                                    uint32_t *__restrict__ state, const ProbeIdx ix, int count, int reset,
                                                    const uint32_t *__restrict__ cols, int64_t C, int64_t poly, int k, int decide,
                                                    ProbeResult *__restrict__ result, int seq, uint32_t *__restrict__ msgs, uint32_t uq) {
    extern __shared__ __attribute__((aligned(16))) uint32_t p_lds[];
    if (gridDim.x > 1 && (blockIdx.x & 7)) return;
    const int G = gridDim.x > 1 ? (int)(gridDim.x >> 3) + 1 : 1, g = (int)(blockIdx.x >> 3);
    // who holds what: workgroup 0 the coefficients (and the decision) -- from three workgroups on, workgroup 0 the A parts and workgroup 1 the
    // B parts (an item of one part never reads the other) --, the rest a slice of the parties' values each
    const bool leader = g == 0;
    const int NC = G >= 3 ? 2 : 1;                           // workgroups that hold coefficients
    const int cp_lo = G >= 3 ? (g < 2 ? g : 0) : 0, cp_hi = G >= 3 ? (g < 2 ? g + 1 : 0) : (leader ? 2 : 0), cp_n = cp_hi - cp_lo;
    const bool c_own = cp_n > 0;
    const int v_per = G > 1 ? (n + G - NC - 1) / (G - NC) : n;     // ... and the values at the parties [v_lo, v_hi)
    const int v_lo = G > 1 ? (c_own ? 0 : min(n, (g - NC) * v_per)) : 0, v_hi = G > 1 ? (c_own ? 0 : min(n, (g - NC + 1) * v_per)) : n, v_n = v_hi - v_lo;
    uint32_t *dmsg = msgs, *gmsg = msgs + (size_t)PROBE_MAXN * PROBE_DW, *gdone = gmsg + (size_t)PROBE_MAXN * PROBE_GW, *gabort = gdone + PROBE_MAXG;
    const uint32_t abort_v = 0x80000000u | uq;               // what a waiter that gave up leaves in the abort word (cleared by a reset)
    if (G > 1 && leader && reset && threadIdx.x == 0) __hip_atomic_store(gabort, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t q0 = uq << 9;                             // sequence numbers of this launch's points: q0 + 1 .. (uq: unique among the launches that use this buffer)
    // coef[q][i][NL], q: 0 = A_0, 1 = B_0, 2 = A_1, 3 = B_1; val[q][party][NL]; a scratch polynomial for the decision; the reduction buffer
    uint32_t *coef = p_lds;
    uint32_t *val = coef + (size_t)4 * S * NL;               // [4][n][NL]
    uint32_t *scr = val + (size_t)4 * n * NL;                // [S][NL]: the locator the decision looks at
    uint32_t *xl = scr + (size_t)S * NL;                     // [n][NL]: the party points (Montgomery), read by every value update
    __shared__ int deg[4], ctl[8], sdeg[2];
    __shared__ uint16_t fedl[PROBE_MAXN];                     // the parties fed so far, in order (persisted with the state)
    __shared__ uint32_t serr[PROBE_MAXN / 4];                 // the verdict's error bytes, gathered before they cross to the host
    __shared__ uint32_t dl[2][NL];
    const int tid = threadIdx.x;
    const size_t cwords = (size_t)4 * S * NL, vwords = (size_t)4 * n * NL, words = cwords + vwords;
    for (int i = tid; i < n * NL; i += PROBE_NT) xl[i] = xm[i];
    int32_t *st_i = reinterpret_cast<int32_t *>(state + words);   // [0..3] degrees, [4] points fed, [8 + i] the i-th party fed
    if (reset) {
        for (int i = tid; i < 2 * cp_n * S * NL; i += PROBE_NT) {
            const int b_ = i / (S * NL), r = i - b_ * (S * NL), q = 2 * (b_ / cp_n) + cp_lo + b_ % cp_n;
            coef[(size_t)q * S * NL + r] = 0;
        }
        // Q_0 = (1, 0), Q_1 = (0, 1): A_0 = 1 and B_1 = 1 everywhere, A_1 = B_0 = 0
        for (int i = tid; i < 4 * v_n * NL; i += PROBE_NT) {
            const int q = i / (v_n * NL), r = i - q * (v_n * NL);
            val[((size_t)q * n + v_lo) * NL + r] = (q == 0 || q == 3) ? P.one[r % NL] : 0u;
        }
        __syncthreads();
        if (tid < NL) {                                     // Q_0 = 1, Q_1 = Y
            if (cp_lo <= 0 && 0 < cp_hi) coef[(size_t)0 * S * NL + tid] = P.one[tid];
            if (cp_lo <= 1 && 1 < cp_hi) coef[(size_t)3 * S * NL + tid] = P.one[tid];
        }
        if (tid == 0) { deg[0] = 0; deg[1] = -1; deg[2] = -1; deg[3] = 0; ctl[6] = 0; }
    } else if (G > 1) {
        // this workgroup's part of the state: the coefficients, or its parties' rows of the four value tables
        for (int i = tid; i < 2 * cp_n * S * NL; i += PROBE_NT) {
            const int b_ = i / (S * NL), r = i - b_ * (S * NL), q = 2 * (b_ / cp_n) + cp_lo + b_ % cp_n;
            coef[(size_t)q * S * NL + r] = state[(size_t)q * S * NL + r];
        }
        for (int i = tid; i < 4 * v_n * NL; i += PROBE_NT) {
            const int q = i / (v_n * NL), r = i - q * (v_n * NL);
            const size_t o = ((size_t)q * n + v_lo) * NL + r;
            val[o] = state[cwords + o];
        }
        if (tid < 4) deg[tid] = st_i[tid];
        if (tid == 0) ctl[6] = st_i[4];
        if (leader && tid < PROBE_MAXN) fedl[tid] = (uint16_t)st_i[8 + tid];
    } else {
        // (val follows coef in LDS as in the state; both are multiples of four words.)  Several loads in flight per thread: one word at a
        // time this prologue was a chain of dependent round trips
        {
            const uint4 *src = reinterpret_cast<const uint4 *>(state);
            uint4 *dst = reinterpret_cast<uint4 *>(coef);
            const int nq = (int)(words / 4);
            for (int base = 0; base < nq; base += PROBE_NT * 4) {
                uint4 tmp[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int i = base + u * PROBE_NT + tid; if (i < nq) tmp[u] = src[i]; }
#pragma unroll
                for (int u = 0; u < 4; u++) { const int i = base + u * PROBE_NT + tid; if (i < nq) dst[i] = tmp[u]; }
            }
        }
        if (tid < 4) deg[tid] = st_i[tid];
        if (tid == 0) ctl[6] = st_i[4];
        if (tid < PROBE_MAXN) fedl[tid] = (uint16_t)st_i[8 + tid];
    }
    __syncthreads();
    // the received symbols of all the points of this launch, in Montgomery form, up front and in parallel (whatever words a sender
    // packed mean their residue: the reference reduces at its boundary); a point's x comes from the LDS table
    uint32_t *yml = xl + (size_t)n * NL;                     // [count][NL]
    for (int pt = tid; pt < count; pt += PROBE_NT) {
        uint32_t yw[NW], yd[NL], ym[NL];
        load_words<NW>(yw, cols + ((size_t)ix.idx[pt] * (size_t)C + (size_t)poly) * NW);
        unpack<NL, NW>(yd, yw);
        to_mont(ym, yd, P);
        stg<NL>(yml + (size_t)pt * NL, ym);
    }
    __syncthreads();
    // The discrepancies of ALL the points of this launch, once: D_j[pt] = A_j(x) + y B_j(x) at the point's party, from the value table.  They are
    // linear in the pair like everything else, so a point's update carries the not-yet-fed ones along (one more item each) and the
    // discrepancy of the next point is a look-up again -- no multiplication between one point's update and the next one's pivot.
    uint32_t *Dl = yml + (size_t)n * NL;                     // [2][count][NL]; a point's two are its party's workgroup's
    __shared__ uint16_t own_pts[PROBE_MAXN];                  // this workgroup's points of the launch, in order
    __shared__ int own_cnt;
    if (tid == 0) {
        int c_ = 0;
        for (int pt = 0; pt < count; pt++) { const int a_ = ix.idx[pt]; if (a_ >= v_lo && a_ < v_hi) own_pts[c_++] = (uint16_t)pt; }
        own_cnt = c_;
    }
    for (int e = tid; e < 2 * count; e += PROBE_NT) {
        const int j = e & 1, pt = e >> 1, a_ = ix.idx[pt];
        if (a_ >= v_lo && a_ < v_hi) {
            uint32_t ym[NL], va[NL], vb[NL], m[NL], dd[NL];
            ldg<NL>(ym, yml + (size_t)pt * NL);
            ldg<NL>(va, val + ((size_t)(2 * j) * n + a_) * NL);
            ldg<NL>(vb, val + ((size_t)(2 * j + 1) * n + a_) * NL);
            mont_mul(m, ym, vb, P);
            fp_add(dd, va, m, P);
            stg<NL>(Dl + ((size_t)j * count + pt) * NL, dd);
        }
    }
    __syncthreads();
    int own_done = 0;                                         // own points fed so far (the same in every thread)
    for (int pt = 0; pt < count; pt++) {
        const int a = ix.idx[pt];
        const bool d_own = a >= v_lo && a < v_hi;             // the values at the new party are this workgroup's
        if (tid == 0) ctl[3] = 0;                             // (set by whoever sees a leading coefficient cancel in this point's update)
        if (d_own && tid < 2) {
            uint32_t dd[NL];
            ldg<NL>(dd, Dl + ((size_t)tid * count + pt) * NL);
            stg<NL>(dl[tid], dd);
            ctl[tid] = fp_is_zero(dd) ? 0 : 1;
        }
        if (d_own) own_done++;
        __syncthreads();
        if (G > 1) {
            uint32_t *dm = dmsg + (size_t)pt * PROBE_DW;
            const uint32_t q = q0 + (uint32_t)pt + 1;
            if (d_own) {
                if (tid < 2 * NL) probe_send(dm, tid, q, dl[tid / NL][tid % NL]);
                else if (tid < 2 * NL + 2) probe_send(dm, tid, q, (uint32_t)ctl[tid - 2 * NL]);
            } else {
                if (tid < 2 * NL) dl[tid / NL][tid % NL] = probe_recv(dm, tid, q, gabort, abort_v);
                else if (tid < 2 * NL + 2) ctl[tid - 2 * NL] = (int)probe_recv(dm, tid, q, gabort, abort_v);
            }
            // the degrees the previous point left (only the coefficients can tell when a leading one cancelled)
            if (pt > 0 && tid >= 64 && tid < 68 && !(((tid - 64) & 1) >= cp_lo && ((tid - 64) & 1) < cp_hi))
                deg[tid - 64] = (int)probe_recv(gmsg + (size_t)(pt - 1) * PROBE_GW, tid - 64, q - 1, gabort, abort_v);
            __syncthreads();
        }
        // the pair of smaller leading monomial among those with a discrepancy; (1, k-1)-weighted degree, ties: Y terms larger
        // (every thread works it out for itself from the shared flags and degrees)
        int js = -1;
        {
            int best_w = 0, best_y = 0;
            for (int j = 0; j < 2; j++) {
                if (!ctl[j]) continue;
                const int wa = deg[2 * j], wb = deg[2 * j + 1] >= 0 ? deg[2 * j + 1] + k - 1 : -1;
                const int w = wa > wb ? wa : wb, yy = wb >= wa ? 1 : 0;
                if (js < 0 || w < best_w || (w == best_w && yy < best_y)) { js = j; best_w = w; best_y = yy; }
            }
        }
        if (tid == 0) { if (leader) fedl[ctl[6]] = (uint16_t)a; ctl[6] += 1; }
        if (js >= 0) {
            const int jo = 1 - js;
            const int top = max(max(deg[0], deg[1]), max(deg[2], deg[3])) + 1;      // highest index any polynomial reaches after this step
            const bool upd_o = ctl[jo] != 0;
            // the degrees after the step, as they follow from the degrees before it: the pivot's parts grow by one, the other pair's parts
            // become the larger of the two.  Only where the two are EQUAL can the leading coefficients cancel (d_js u - d_jo v with both
            // non-zero); the thread that computes that coefficient looks (an event of probability ~1/p for random data, but exactness does
            // not gamble) and a scan finds the degrees then
            // (scalars, not an array: an array indexed by the pivot's number lives in scratch memory -- a round trip to L2 per access)
            const int djs0 = deg[2 * js], djs1 = deg[2 * js + 1], djo0 = deg[2 * jo], djo1 = deg[2 * jo + 1];
            const int nds0 = djs0 >= 0 ? djs0 + 1 : -1, nds1 = djs1 >= 0 ? djs1 + 1 : -1;
            const int ndo0 = upd_o ? max(djs0, djo0) : djo0, ndo1 = upd_o ? max(djs1, djo1) : djo1;
            // Units of work: an ITEM is coefficient i of part `part` (A or B) for i <= top, the value at party i of part `part`, or the pair of
            // discrepancies of a point of this launch still to come; each item is two units, for two threads -- Q_jo <- d_js Q_jo - d_jo Q_js
            // (two products, one reduction) and Q_js <- (X - x_a) Q_js (one product; a coefficient takes its lower neighbour) -- so a point's
            // update is as long as the longer of them, not their sum.  Every unit reads the state as the last point left it; results wait in
            // registers until all have read (a thread has at most two: 2 (2 (n + 3) + 2 n + n) <= 2 PROBE_NT)
            const int nce = cp_n * (top + 1), nD = own_cnt - own_done, U = nce + 2 * v_n + nD;
            uint32_t xa[NL], ds[NL], ndj[NL];
            ldg<NL>(xa, xl + (size_t)a * NL);
            ldg<NL>(ds, dl[js]);
            {
                uint32_t dj[NL];
                ldg<NL>(dj, dl[jo]);
                fp_neg(ndj, dj, P);
            }
            // (two named results, not an array over the unit's number: that array lived in scratch memory)
            auto unit = [&](int e, uint32_t (&res)[NL], uint32_t *&dst) {
                dst = nullptr;
                if (e >= 2 * U) return;
                const int sub = e >= U ? 1 : 0, it = e - sub * U;
                if (sub == 0 && !upd_o) return;
                // where the item's two elements live (po: of Q_jo, ps: of Q_js), and what the pivot's element is multiplied by
                uint32_t *po, *ps;
                int ci = -1, cpart = 0;                               // a coefficient item's index and part
                const uint32_t *xi = nullptr;                         // a value's / a coming point's party point
                if (it < nce) {
                    cpart = it >= top + 1 ? 1 : 0; ci = it - cpart * (top + 1); cpart += cp_lo;      // (at most two parts: no division)
                    po = coef + ((size_t)(2 * jo + cpart) * S + ci) * NL;
                    ps = coef + ((size_t)(2 * js + cpart) * S + ci) * NL;
                } else if (it < nce + 2 * v_n) {
                    const int ee = it - nce, part = ee >= v_n ? 1 : 0, i = v_lo + ee - part * v_n;
                    po = val + ((size_t)(2 * jo + part) * n + i) * NL;
                    ps = val + ((size_t)(2 * js + part) * n + i) * NL;
                    xi = xl + (size_t)i * NL;
                } else {
                    const int p2 = own_pts[own_done + (it - nce - 2 * v_n)];
                    po = Dl + ((size_t)jo * count + p2) * NL;
                    ps = Dl + ((size_t)js * count + p2) * NL;
                    xi = xl + (size_t)ix.idx[p2] * NL;
                }
                uint32_t v[NL];
                ldg<NL>(v, ps);
                if (sub == 0) {
                    // Q_jo <- d_js Q_jo - d_jo Q_js: one reduction for the two products
                    uint32_t u[NL];
                    uint64_t col[2 * NL];
                    ldg<NL>(u, po);
                    col_zero(col);
                    mac<NL>(col, ds, u);
                    mac<NL>(col, ndj, v);
                    redc(res, col, P);
                    cond_sub_p(res, P);
                    dst = po;
                    if (ci >= 0 && (cpart ? (djs1 == djo1 && ci == djo1) : (djs0 == djo0 && ci == djo0)) && fp_is_zero(res)) ctl[3] = 1;
                } else if (ci >= 0) {
                    // Q_js <- (X - x_a) Q_js, a coefficient
                    uint32_t prev[NL], m[NL];
                    if (ci > 0) ldg<NL>(prev, ps - NL);
                    else {
#pragma unroll
                        for (int w = 0; w < NL; w++) prev[w] = 0;
                    }
                    mont_mul(m, xa, v, P);
                    fp_sub(res, prev, m, P);
                    dst = ps;
                } else {
                    uint32_t xv[NL], df[NL];
                    ldg<NL>(xv, xi);
                    fp_sub(df, xv, xa, P);
                    mont_mul(res, df, v, P);
                    dst = ps;
                }
            };
            uint32_t res0[NL], res1[NL];
            uint32_t *dst0, *dst1;
            unit(tid, res0, dst0);
            unit(tid + PROBE_NT, res1, dst1);
            __syncthreads();
            const bool rescan = ctl[3] != 0;                  // (read before the stores' barrier: thread 0 clears the flag for the next point after it)
            if (dst0) stg<NL>(dst0, res0);
            if (dst1) stg<NL>(dst1, res1);
            if (!rescan && tid < 4) deg[tid] = (tid >> 1) == js ? ((tid & 1) ? nds1 : nds0) : ((tid & 1) ? ndo1 : ndo0);
            __syncthreads();
            if (rescan) {
                // (of the parts this workgroup holds: the others' degrees arrive with the next point)
                if (tid < 4 && (tid & 1) >= cp_lo && (tid & 1) < cp_hi) deg[tid] = -1;
                __syncthreads();
                for (int e = tid; e < 2 * cp_n * (top + 1); e += PROBE_NT) {
                    const int b_ = e / (top + 1), i = e - b_ * (top + 1), q = 2 * (b_ / cp_n) + cp_lo + b_ % cp_n;
                    if (lds_nonzero<NL>(coef + ((size_t)q * S + i) * NL)) atomicMax(&deg[q], i);
                }
                __syncthreads();
            }
        } else {
            __syncthreads();          // (the list of fed parties and the flags are read again by the next point)
        }
        if (G > 1 && c_own) {
            if (tid < 4 && (tid & 1) >= cp_lo && (tid & 1) < cp_hi) probe_send(gmsg + (size_t)pt * PROBE_GW, tid, q0 + (uint32_t)pt + 1, (uint32_t)deg[tid]);
        }
    }
    // (the degrees of the parts another workgroup holds, as the last point left them)
    if (G > 1 && leader && count > 0 && tid >= 64 && tid < 68 && !(((tid - 64) & 1) >= cp_lo && ((tid - 64) & 1) < cp_hi))
        deg[tid - 64] = (int)probe_recv(gmsg + (size_t)(count - 1) * PROBE_GW, tid - 64, q0 + (uint32_t)count, gabort, abort_v);
    // persistent state back (the decision below works on copies)
    __syncthreads();
    if (G == 1) {
        uint4 *dst = reinterpret_cast<uint4 *>(state);
        const uint4 *src = reinterpret_cast<const uint4 *>(coef);
        const int nq = (int)(words / 4);
        for (int i = tid; i < nq; i += PROBE_NT) dst[i] = src[i];
    } else {
        for (int i = tid; i < 2 * cp_n * S * NL; i += PROBE_NT) {
            const int b_ = i / (S * NL), r = i - b_ * (S * NL), q = 2 * (b_ / cp_n) + cp_lo + b_ % cp_n;
            state[(size_t)q * S * NL + r] = coef[(size_t)q * S * NL + r];
        }
        for (int i = tid; i < 4 * v_n * NL; i += PROBE_NT) {
            const int q = i / (v_n * NL), r = i - q * (v_n * NL);
            const size_t o = ((size_t)q * n + v_lo) * NL + r;
            state[cwords + o] = val[o];
        }
    }
    if (leader) {
        if (tid < 4) st_i[tid] = deg[tid];
        if (tid == 0) st_i[4] = ctl[6];
        if (tid < PROBE_MAXN) st_i[8 + tid] = fedl[tid];
    }
    if (G > 1) {
        // a value workgroup is done once its slice is back in the state; the deciding one reads the whole table from there
        __syncthreads();
        if (!leader) { if (tid == 0) probe_post(gdone + g, uq); return; }
        if (!decide) return;
        if (tid > 0 && tid < G) probe_wait(gdone + tid, uq, gabort, abort_v);
        __syncthreads();
        for (size_t i = tid; i < vwords; i += PROBE_NT) val[i] = __builtin_nontemporal_load(state + cwords + i);
        for (int i = tid; G >= 3 && i < 2 * S * NL; i += PROBE_NT) {       // the B parts' coefficients
            const size_t o = (size_t)(2 * (i / (S * NL)) + 1) * S * NL + i % (S * NL);
            coef[o] = __builtin_nontemporal_load(state + o);
        }
    }
    if (!decide) return;
    __syncthreads();
    // ---- the reference's outcome for the points fed so far -----------------------------------------------------------
    const int npts = ctl[6];
    if (tid == 0) {
        // Q_0: the pair whose leading monomial is in A
        int cls[2];
        for (int j = 0; j < 2; j++) {
            const int wa = deg[2 * j], wb = deg[2 * j + 1] >= 0 ? deg[2 * j + 1] + k - 1 : -1;
            cls[j] = wb >= wa ? 1 : 0;
        }
        const int j0 = cls[0] == 0 ? 0 : 1, j1 = 1 - j0;
        const int T = (npts + k) / 2;
        ctl[3] = j0; ctl[4] = j1;
        ctl[5] = (cls[j0] == 0 && cls[j1] == 1 && deg[2 * j0] >= T) ? 1 : 0;       // 0: Gao's row has deg f >= k -> not decodable
        ctl[7] = (deg[2 * j0] == T && deg[2 * j1] >= T) ? 1 : 0;                    // the tie: reduce Q_1 against Q_0
    }
    __syncthreads();
    bool fine = ctl[5] != 0;
    const int j0 = ctl[3], j1 = ctl[4];
    uint32_t *Bd = scr;                                       // the locator: B_1, reduced against Q_0 in the tie case
    if (fine) {
        const int T = (npts + k) / 2;
        if (ctl[7]) {
            uint32_t c0[NL], c1[NL];
            ldg<NL>(c0, coef + ((size_t)(2 * j0) * S + T) * NL);
            ldg<NL>(c1, coef + ((size_t)(2 * j1) * S + T) * NL);
            for (int i = tid; i < S; i += PROBE_NT) {
                uint32_t u[NL], v[NL], m1[NL], m2[NL], r[NL];
                ldg<NL>(u, coef + ((size_t)(2 * j1 + 1) * S + i) * NL);
                ldg<NL>(v, coef + ((size_t)(2 * j0 + 1) * S + i) * NL);
                mont_mul(m1, c0, u, P);
                mont_mul(m2, c1, v, P);
                fp_sub(r, m1, m2, P);
                stg<NL>(Bd + (size_t)i * NL, r);
            }
        } else {
            for (int i = tid; i < S; i += PROBE_NT) {
                uint32_t u[NL];
                ldg<NL>(u, coef + ((size_t)(2 * j1 + 1) * S + i) * NL);
                stg<NL>(Bd + (size_t)i * NL, u);
            }
        }
        if (tid < 2) sdeg[tid] = -1;
        if (tid == 0) ctl[0] = 0;
        __syncthreads();
        for (int e = tid; e < S; e += PROBE_NT)
            if (lds_nonzero<NL>(Bd + (size_t)e * NL)) atomicMax(&sdeg[1], e);
        __syncthreads();
    }
    // Gao decodes  <=>  the locator B_1 has deg B_1 DISTINCT roots among the points fed (then it divides A_1 -- A_1 + y B_1
    // vanishes on every point -- and conversely a decodable word's locator is exactly the product over its errors; checked
    // against the oracle on 15 000 prefixes, 299 beyond the unique-decoding radius: tests/test_probe_rule.py).  Four threads
    // per point, each a quarter of the coefficients; the senders in error are those roots (reed_solomon.py:174-184).
    if (tid < PROBE_MAXN / 4) serr[tid] = 0;
    __syncthreads();
    const int db = fine ? sdeg[1] : -1;
    if (fine) {
        // the locator's values at the points fed are in the value table (B_1 itself, or the same combination of the two B's that
        // reduced it in the tie case): a root is a zero there -- no evaluation
        uint32_t c0[NL], nc1[NL];
        const bool tie = ctl[7] != 0;
        if (tie) {
            const int T = (npts + k) / 2;
            uint32_t c1[NL];
            ldg<NL>(c0, coef + ((size_t)(2 * j0) * S + T) * NL);
            ldg<NL>(c1, coef + ((size_t)(2 * j1) * S + T) * NL);
            fp_neg(nc1, c1, P);
        }
        for (int pi = tid; pi < npts; pi += PROBE_NT) {
            const int a = fedl[pi];
            uint32_t v[NL];
            ldg<NL>(v, val + ((size_t)(2 * j1 + 1) * n + a) * NL);
            if (tie) {
                uint32_t v0[NL], r[NL];
                uint64_t col[2 * NL];
                ldg<NL>(v0, val + ((size_t)(2 * j0 + 1) * n + a) * NL);
                col_zero(col);
                mac<NL>(col, c0, v);
                mac<NL>(col, nc1, v0);
                redc(r, col, P);
                cond_sub_p(r, P);
                fp_set(v, r);
            }
            if (fp_is_zero(v) && db >= 1) {            // a constant locator names nobody
                atomicOr(&serr[a >> 2], 1u << (8 * (a & 3)));
                atomicAdd(&ctl[0], 1);
            }
        }
        __syncthreads();
        if (db < 0 || ctl[0] != (db >= 1 ? db : 0)) fine = false;
    }
    __syncthreads();
    // one wave hands the verdict over: its stores, a system-scope fence, then the sequence number the host is polling
    if (tid < 64) {
        reinterpret_cast<uint32_t *>(result->err)[tid] = fine ? serr[tid] : 0u;
        if (tid == 0) {
            const bool gave_up = G > 1 && (__hip_atomic_load(gabort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 31) != 0;      // (in this launch or in one since the reset: the state is void either way)
            result->ok = gave_up ? -1 : (fine ? 1 : 0); result->npts = npts; result->n_err = fine ? ctl[0] : 0;
        }
        __threadfence_system();
        if (tid == 0) *reinterpret_cast<volatile int32_t *>(&result->seq) = seq;
    }
}

static void point_table_unref(PointTable *pt) {
    if (!pt || --pt->refs > 0) return;
    (void)hipFree(pt->xm); (void)hipFree(pt->inv); (void)hipFree(pt->pw);
    if (pt->xs_dev) (void)hipFree(pt->xs_dev);
    delete pt;
}

// per (context, point set): an entry of the context's LRU like every other table (n^2 elements: 2.4 MB at n = 256), so a caller that
// keeps inventing point sets cannot grow it without bound; a probe holds its own reference (cache_trim synchronises the device
// before it drops anything, and only the outermost entry point trims)
int point_table(hb_ctx *ctx, const uint64_t *x_host, int n, PointTable **out, hipStream_t s) {
    std::string key = table_key("PT", ctx, x_host, n, 0);
    auto it = ctx->ptcache.find(key);
    if (it != ctx->ptcache.end()) { cache_touch(ctx, "pt|" + key); *out = static_cast<PointTable *>(it->second); return HB_OK; }
    PointTable *pt = new PointTable();
    pt->n = n; pt->S = n + 2; pt->xm = pt->inv = pt->pw = nullptr; pt->usable = false; pt->refs = 1; pt->xs_dev = nullptr;
    pt->small = ctx->n_limbs == 4;
    for (int i = 0; i < n && pt->small; i++) {
        const uint64_t *e = x_host + (size_t)i * 4;
        if (e[0] >= 65536 || e[1] || e[2] || e[3]) pt->small = false; else pt->xs.push_back((uint16_t)e[0]);
    }
    if (!pt->small) pt->xs.clear();
    const int NLr = ctx->nl();
    uint32_t *xd = nullptr;
    int32_t *bad = nullptr;
    int rc = upload_elems(ctx, x_host, (size_t)n, &xd, s);
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMalloc(&pt->xm, (size_t)n * NLr * 4);
        if (e == hipSuccess) e = hipMalloc(&pt->inv, (size_t)n * n * NLr * 4);
        if (e == hipSuccess) e = hipMalloc(&pt->pw, (size_t)n * pt->S * NLr * 4);
        if (e == hipSuccess) e = hipMalloc(&bad, sizeof(int32_t));
        if (e == hipSuccess) e = hipMemsetAsync(bad, 0, sizeof(int32_t), s);
    }
    int32_t badh = 0;
    if (!rc && e == hipSuccess) {
        const unsigned blocks = (unsigned)((n + 63) / 64);
        if (ctx->n_limbs == 4) {
            k_pt_mont<9, 8><<<blocks, 64, 0, s>>>(ctx->pw, xd, n, pt->xm, pt->pw, pt->S);
            k_pt_inv<9><<<blocks, 64, 0, s>>>(ctx->pw, pt->xm, n, pt->inv, bad);
        } else {
            k_pt_mont<3, 2><<<blocks, 64, 0, s>>>(ctx->pn, xd, n, pt->xm, pt->pw, pt->S);
            k_pt_inv<3><<<blocks, 64, 0, s>>>(ctx->pn, pt->xm, n, pt->inv, bad);
        }
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(&badh, bad, sizeof badh, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    if (xd) (void)hipFree(xd);
    if (bad) (void)hipFree(bad);
    if (rc || e != hipSuccess) {
        if (pt->xm) (void)hipFree(pt->xm);
        if (pt->inv) (void)hipFree(pt->inv);
        if (pt->pw) (void)hipFree(pt->pw);
        delete pt;
        if (!rc) { ctx->err = std::string("point table: ") + hipGetErrorString(e); rc = HB_ERR_HIP; }
        return rc;
    }
    pt->usable = badh == 0;                       // repeated points: interpolation over them is singular (callers fall back and report it)
    ctx->ptcache[key] = pt;
    cache_note(ctx, "pt|" + key, [ctx, key]() {
        auto f = ctx->ptcache.find(key);
        if (f != ctx->ptcache.end()) { PointTable *q = static_cast<PointTable *>(f->second); ctx->ptcache.erase(f); point_table_unref(q); }
    });
    *out = pt;
    return HB_OK;
}

void point_tables_free(hb_ctx *ctx) {
    for (auto &kv : ctx->ptcache) point_table_unref(static_cast<PointTable *>(kv.second));
    ctx->ptcache.clear();
    for (auto &sl : ctx->qslots) {
        if (sl.buf) (void)hipFree(sl.buf);
        if (sl.ev) (void)hipEventDestroy((hipEvent_t)sl.ev);
    }
    ctx->qslots.clear();
    for (void *p : ctx->probe_pool) (void)hipFree(p);
    ctx->probe_pool.clear();
    for (void *p : ctx->probe_host_pool) (void)hipHostFree(p);
    ctx->probe_host_pool.clear();
    if (ctx->fetch_host) (void)hipHostFree(ctx->fetch_host);
    ctx->fetch_host = ctx->fetch_dev = nullptr;
    if (ctx->cand_host) (void)hipHostFree(ctx->cand_host);
    if (ctx->cand_ticket) (void)hipFree(ctx->cand_ticket);
    ctx->cand_host = ctx->cand_dev = nullptr; ctx->cand_ticket = nullptr;
    if (ctx->after_event) (void)hipEventDestroy((hipEvent_t)ctx->after_event);
    ctx->after_event = nullptr;
}

// hb_symbols_fetch: a few symbols of one polynomial cross to the host through pinned memory, the sequence number last
constexpr int FETCH_MAX = 64;
struct SymFetch { uint64_t w[FETCH_MAX * 4]; int32_t seq; };
struct FetchIdx { int32_t idx[FETCH_MAX]; };
__global__ void __launch_bounds__(256) k_symbols_fetch(const uint64_t *__restrict__ cols, int64_t C, int64_t chunk, int L, const FetchIdx ix, int count,
                                                       SymFetch *__restrict__ out, int seq) {
    const int t = threadIdx.x;
    if (t < count * L) out->w[t] = cols[((size_t)ix.idx[t / L] * (size_t)C + (size_t)chunk) * L + t % L];
    __threadfence_system();
    __syncthreads();
    if (t == 0) *reinterpret_cast<volatile int32_t *>(&out->seq) = seq;
}

// A candidate against the arrived symbols of its chunk, in one launch: workgroup i evaluates the polynomial at party i's point
// (eval_at_point_128, hb_common.hpp: k_eval_few's arithmetic), writes the value to pinned memory and says
// whether the party's symbol of the chunk differs; the last workgroup to finish (a ticket) writes the sequence number the host polls.
constexpr int CAND_MAXN = 1024;
struct CandCheck { uint64_t vals[CAND_MAXN * 4]; uint8_t differs[CAND_MAXN]; int32_t seq; };
template <int NL, int NW>
__global__ void __launch_bounds__(128) k_candidate_check(const FpParams<NL> P, const uint32_t *__restrict__ x, int n, const uint32_t *__restrict__ poly, int d,
                                                         const uint32_t *__restrict__ cols, int64_t C, int64_t chunk, CandCheck *__restrict__ out,
                                                         int32_t *__restrict__ ticket, int seq) {
    __shared__ uint32_t red[128][NL];
    const int i = blockIdx.x, tid = threadIdx.x;
    uint32_t r[NL];
    eval_at_point_128<NL, NW>(r, x + (size_t)i * NW, poly, d, P, red);
    if (tid == 0) {
        uint32_t w[NW], got[NW];
        pack<NL, NW>(w, r);
        load_words<NW>(got, cols + ((size_t)i * (size_t)C + (size_t)chunk) * NW);
        uint32_t df = 0;
#pragma unroll
        for (int q = 0; q < NW; q++) df |= w[q] ^ got[q];
        uint32_t *o = reinterpret_cast<uint32_t *>(out->vals) + (size_t)i * NW;      // (a narrow context's element is one 64-bit word: NW = 2)
#pragma unroll
        for (int q = 0; q < NW; q++) o[q] = w[q];
        out->differs[i] = df ? 1 : 0;
        __threadfence_system();
        if (atomicAdd(ticket, 1) == n - 1) {
            *ticket = 0;
            __threadfence_system();
            *reinterpret_cast<volatile int32_t *>(&out->seq) = seq;
        }
    }
}

}  // namespace hb

struct hb_probe {
    hb_ctx *ctx;
    int n, k;
    PointTable *pt;
    uint32_t *state;              // 4 polynomials of n + 2 coefficients + 8 ints
    size_t state_bytes;
    ProbeResult *res_host;        // pinned, device-visible: the kernel writes the verdict where the host reads it
    ProbeResult *res_dev;
    std::vector<int32_t> fed;
    int64_t poly;
    int seq;                      // launches so far: the kernel echoes it into res_host->seq when its verdict is complete
    int wgs;                      // workgroups of a launch: 1, or 6 above PROBE_SPLIT_N points (HB_PROBE_WGS overrides: tests)
};

extern "C" {

}  // extern "C" (the builder below is shared with hb_open.hip)

namespace hb {

// sizes and offsets of everything a device-built image of [n_coef rows of V^-1(z) ; V[zc] V^-1(z)] needs, in one buffer:
// image | row constants | (build scratch: w_j, prod_q (x_i - x_zq), N_j, canonical entries) | z as int32 | the row map
int quick_layout(hb_ctx *ctx, int n, int d, int nc, int n_coef, QuickLayout *L) {
    if (ctx->n_limbs != 4 || d < 4 || d > QUICK_MAX || nc < 0 || nc > QUICK_MAXC || n > 65535 || n_coef < 1 || n_coef > d) return fail(ctx, HB_ERR_UNSUPPORTED, "quick: shape");
    if (env_hook(ENV_NO_MFMA) || env_hook(ENV_NO_MFMA_WIDE) || env_hook(ENV_NO_QUICK)) return fail(ctx, HB_ERR_UNSUPPORTED, "quick: disabled");
    // entries below p must fit 32 balanced base-256 digits: top byte of p at most 0x7e
    if (!prescale_params(ctx) || (ctx->p_limbs[3] >> 56) > 0x7e) return fail(ctx, HB_ERR_UNSUPPORTED, "quick: modulus");
    L->n = n; L->d = d; L->nc = nc; L->n_coef = n_coef; L->n_out = n_coef + nc;
    int n_rt = 0;
    size_t a8_bytes = 0, crow_words = 0;
    const int rc = mm8w_geometry(L->n_out, d, &L->tile_rows, &n_rt, &L->nkb, &a8_bytes, &crow_words);
    if (rc) return fail(ctx, rc, "quick: geometry");
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    L->o_a8 = 0; L->o_crow = L->o_a8 + al(a8_bytes); L->o_wj = L->o_crow + al(crow_words * 4); L->o_full = L->o_wj + al((size_t)d * 36);
    L->o_nraw = L->o_full + al((size_t)(nc ? nc : 1) * 36); L->o_mcan = L->o_nraw + al((size_t)n_coef * d * 36);
    L->o_z = L->o_mcan + al((size_t)L->n_out * d * 32); L->o_map = L->o_z + al((size_t)d * 4); L->o_sync = L->o_map + al((size_t)(L->n_out + 2) * 4);
    L->need = L->o_sync + al((size_t)(d + 1) * 36 + 64);                    // (until round 6: A(X) for the builder's helper workgroups, and their token)
    L->o_cand = L->o_cand_crow = 0;
    return HB_OK;
}

int quick_layout_cand(hb_ctx *ctx, QuickLayout *L) {
    (void)ctx;
    if (L->nc < 1 || L->n > 1024 || L->d > QF_NT || env_hook(ENV_QUICK_NO_CAND)) return HB_ERR_UNSUPPORTED;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    // full_i for every party takes the place of the compared senders' (o_full is sized for nc): moved to the end with the store
    L->o_full = L->need;
    L->o_cand = L->o_full + al((size_t)L->n * 36);
    L->o_cand_crow = L->o_cand + al((size_t)L->n * L->nkb * 16 * 16);
    L->need = L->o_cand_crow + al((size_t)L->n * 16 * 4);
    return HB_OK;
}

// enqueue the build of that image into `base` (L.need bytes, the caller's): two to three small kernels, nothing waited for
int quick_build(hb_ctx *ctx, const uint64_t *x_host, const int32_t *z, const int32_t *zc, const QuickLayout &L, uint8_t *base, const Mm8wShared **shared, hipStream_t s, int flags) {
    const int n = L.n, d = L.d, nc = L.nc;
    const bool do_z = flags & 1, do_zc = flags & 2, cand = (flags & 4) && L.o_cand;
    std::vector<uint8_t> seen((size_t)n, 0);
    for (int i = 0; i < d; i++) { if (z[i] < 0 || z[i] >= n || seen[z[i]]) return fail(ctx, HB_ERR_BAD_ARG, "quick: arrival indices"); seen[z[i]] = 1; }
    if (do_zc)
        for (int j = 0; j < nc; j++) { if (zc[j] < 0 || zc[j] >= n || seen[zc[j]]) return fail(ctx, HB_ERR_BAD_ARG, "quick: compared indices"); seen[zc[j]] = 1; }
    PointTable *pt = nullptr;
    int rc = point_table(ctx, x_host, n, &pt, s); if (rc) return rc;
    if (!pt->usable) return fail(ctx, HB_ERR_UNSUPPORTED, "quick: repeated points");
    const Mm8wShared *sh = nullptr;
    rc = mm8w_shared(ctx, d, &sh, s); if (rc) return rc;
    *shared = sh;
    QuickIdx ix;
    memset(&ix, 0, sizeof ix);
    for (int i = 0; i < d; i++) ix.z[i] = (uint16_t)z[i];
    if (do_zc) for (int j = 0; j < nc; j++) ix.zc[j] = (uint16_t)zc[j];
    uint32_t *wj = (uint32_t *)(base + L.o_wj), *full = (uint32_t *)(base + L.o_full), *nraw = (uint32_t *)(base + L.o_nraw), *mcan = (uint32_t *)(base + L.o_mcan);
    QuickRowConst rcs;
    memcpy(rcs.c80r, sh->c80r, sizeof rcs.c80r);
    memcpy(rcs.biasmod, sh->biasmod, sizeof rcs.biasmod);
    if (cand && do_zc && !do_z) {
        // second half of a decoder with a candidate store: the compared senders' rows are there already
        if (nc > 0) {
            k_quick_pick<<<(unsigned)((nc * L.nkb * 16 + 255) / 256), 256, 0, s>>>(ix, nc, L.n_coef, L.tile_rows, L.nkb, (const uint4 *)(base + L.o_cand),
                                                                                 (const uint32_t *)(base + L.o_cand_crow), (uint4 *)(base + L.o_a8), (uint32_t *)(base + L.o_crow),
                                                                                 (int32_t *)(base + L.o_map));
            HB_LAUNCH_CHECK(ctx);
        }
        return HB_OK;
    }
    // the rows this call owns: the coefficient rows with the first half, the compared senders' rows with the second
    const int row_lo = do_z ? 0 : L.n_coef, row_hi = do_zc ? L.n_out : L.n_coef;
    const int frows = (cand && do_z) ? n : nc;                               // rows a full_i is computed for
    if (do_z || nc > 0) {
        // (the image and the row constants are zeroed by the launch's last workgroup when it builds the first half: padding rows / terms)
        const int nw = (d + QM_FPB - 1) / QM_FPB, nf = std::max(1, (frows + QM_FPB - 1) / QM_FPB);
        const int nh = L.n_coef == 1 ? nw : ((do_z && d > 40) ? 4 : 1);
        const size_t lds = (size_t)std::max((4 + QM_SEG) * d + 1, d + QM_FSEG * QM_FPB) * 36;
        k_quick_matrix<9><<<nh + nw + nf + (do_z ? 1 : 0), QM_NT, lds, s>>>(ctx->pw, pt->xm, pt->inv, n, ix, d, nc, L.n_coef, wj, full, nraw, (int32_t *)(base + L.o_z),
                                                          (int32_t *)(base + L.o_map), (flags & 3) | (cand && do_z ? 4 : 0), (uint4 *)base, (int)(L.o_wj / 16),
                                                          (uint4 *)(base + L.o_cand), (cand && do_z) ? (int)((L.o_cand_crow - L.o_cand) / 16) : 0,
                                                          nh, nw, nf, pt->pw, pt->S);
    }
    HB_LAUNCH_CHECK(ctx);
    if (cand && do_z) {
        k_quick_fill<<<(unsigned)(L.n_coef + n), QF_NT, 0, s>>>(ctx->pw, pt->inv, n, ix, d, L.n_coef, wj, full, nraw, base + L.o_a8, (uint32_t *)(base + L.o_crow),
                                                               L.tile_rows, L.nkb, rcs, base + L.o_cand, (uint32_t *)(base + L.o_cand_crow), 0);
        HB_LAUNCH_CHECK(ctx);
        return HB_OK;
    }
    if (do_z && do_zc && d <= QF_NT) {
        // the whole image at once (a plan, a robust-phase launch): coefficient rows and the compared senders' rows by the same launch
        k_quick_fill<<<(unsigned)L.n_out, QF_NT, 0, s>>>(ctx->pw, pt->inv, n, ix, d, L.n_coef, wj, full, nraw, base + L.o_a8, (uint32_t *)(base + L.o_crow),
                                                        L.tile_rows, L.nkb, rcs, nullptr, nullptr, 1);
        HB_LAUNCH_CHECK(ctx);
        return HB_OK;
    }
    if (row_hi > row_lo) {
        k_quick_image<<<(unsigned)(((row_hi - row_lo) * d + 255) / 256), 256, 0, s>>>(ctx->pw, pt->inv, n, ix, d, nc, L.n_coef, wj, full, nraw, mcan, base + L.o_a8, L.tile_rows, L.nkb, row_lo, row_hi);
        HB_LAUNCH_CHECK(ctx);
        k_quick_rows<<<(unsigned)((row_hi - row_lo + 63) / 64), 64, 0, s>>>(ctx->pw, mcan, row_lo, row_hi, d, rcs, (uint32_t *)(base + L.o_crow), L.tile_rows);
        HB_LAUNCH_CHECK(ctx);
    }
    return HB_OK;
}

// the launch over a built image: n_store coefficient rows go to out (view ov, out_count elements), the compared rows (if any) are
// checked against the rows of `cols` they belong to
int quick_launch(hb_ctx *ctx, const QuickLayout &L, const uint8_t *base, const Mm8wShared *sh, const uint32_t *cols, hb_view cv, uint32_t *out, hb_view ov,
                 int64_t out_count, int n_store, int32_t *mismatch_dev, int32_t *first_bad_dev, int64_t C, hipStream_t s, uint32_t *bad_map_dev, const FsDone *done) {
    const int32_t *fmap = (const int32_t *)(base + L.o_map);
    return launch_mm8w_raw(ctx, L.n_out, L.d, L.tile_rows, (const void *)(base + L.o_a8), (const uint32_t *)(base + L.o_crow), sh,
                           cols, cv, (const int32_t *)(base + L.o_z), INT64_MAX, out, ov, out_count,
                           L.nc > 0 ? fmap : nullptr, mismatch_dev, C, s, cols, cv, n_store, L.nc > 0 ? first_bad_dev : nullptr, L.nc > 0 ? bad_map_dev : nullptr,
                           L.nc > 0 ? done : nullptr);
}

}  // namespace hb

extern "C" {

int hb_quick_interp_check(hb_ctx *ctx, const uint64_t *x_host, int n, const int32_t *z, int d, const int32_t *zc, int nc,
                          const uint64_t *cols_dev, int64_t C, int64_t chunk_lo, int64_t chunk_hi, uint64_t *coeffs_dev, int32_t *status_dev, void *stream) {
    return hb_quick_interp_check_map(ctx, x_host, n, z, d, zc, nc, cols_dev, C, chunk_lo, chunk_hi, coeffs_dev, status_dev, nullptr, stream);
}

int hb_quick_interp_check_map(hb_ctx *ctx, const uint64_t *x_host, int n, const int32_t *z, int d, const int32_t *zc, int nc,
                              const uint64_t *cols_dev, int64_t C, int64_t chunk_lo, int64_t chunk_hi, uint64_t *coeffs_dev, int32_t *status_dev,
                              uint32_t *bad_map_dev, void *stream) { HB_API_GUARD(ctx);
    if (!ctx || !x_host || !z || n < 1 || d < 1 || d > n || nc < 0 || (nc > 0 && !zc) || C < 0 || chunk_lo < 0 || chunk_hi > C || chunk_lo > chunk_hi) return HB_ERR_BAD_ARG;
    if (chunk_hi == chunk_lo) return HB_OK;
    if (!cols_dev || (nc > 0 && !status_dev)) return HB_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    cache_trim(ctx);
    // one slot of the context's ring: image, row constants, scratch, index arrays
    auto take_slot = [&](size_t need, hb_ctx::QuickSlot **out) -> int {
        if (ctx->qslots.empty()) ctx->qslots.resize(4);
        hb_ctx::QuickSlot &sl = ctx->qslots[ctx->qnext++ % ctx->qslots.size()];
        if (sl.ev) HB_HIP(ctx, hipEventSynchronize((hipEvent_t)sl.ev));      // the launch that last used this slot is over
        else { hipEvent_t ev; HB_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming)); sl.ev = ev; }
        if (sl.cap < need) {
            if (sl.buf) HB_HIP(ctx, hipFree(sl.buf));
            sl.buf = nullptr; sl.cap = 0;
            HB_HIP(ctx, hipMalloc(&sl.buf, need));
            sl.cap = need;
        }
        *out = &sl;
        return HB_OK;
    };
    const int64_t cnt = chunk_hi - chunk_lo;
    const uint32_t *in = (const uint32_t *)cols_dev + (size_t)chunk_lo * 8;
    hb_view pm{1, C}, dv{d, 1};
    // points that are small integers: [N ; P] on the small-entry kernel, the inputs divided by den_j inside it (hb_mfma_fused.hip)
    if (ctx->n_limbs == 4 && !env_hook(ENV_NO_QUICK)) {
        PointTable *pt = nullptr;
        FsLayout F;
        int rc = point_table(ctx, x_host, n, &pt, s); if (rc) return rc;
        if (fs_layout(ctx, pt, d, nc, coeffs_dev ? d : 1, &F) == HB_OK) {
            hb_ctx::QuickSlot *sl = nullptr;
            rc = take_slot(F.need, &sl); if (rc) return rc;
            uint8_t *base = static_cast<uint8_t *>(sl->buf);
            rc = fs_build(ctx, pt, z, zc, F, base, FS_BUILD_Z | FS_BUILD_ZC, status_dev, s); if (rc) return rc;
            uint32_t *out = coeffs_dev ? (uint32_t *)coeffs_dev + (size_t)chunk_lo * d * 8 : nullptr;
            rc = fs_launch(ctx, F, base, in, pm, out, dv, coeffs_dev ? cnt * (int64_t)d : 0, status_dev, (nc > 0 && status_dev) ? status_dev + 1 : nullptr,
                           nc > 0 ? bad_map_dev : nullptr, cnt, s);
            if (rc) return rc;
            HB_HIP(ctx, hipEventRecord((hipEvent_t)sl->ev, s));
            return HB_OK;
        }
    }
    QuickLayout L;
    int rc = quick_layout(ctx, n, d, nc, coeffs_dev ? d : 1, &L); if (rc) return rc;     // nothing to store: one coefficient row keeps the kernel's shapes simple
    hb_ctx::QuickSlot *slp = nullptr;
    rc = take_slot(L.need, &slp); if (rc) return rc;
    hb_ctx::QuickSlot &sl = *slp;
    uint8_t *base = static_cast<uint8_t *>(sl.buf);
    const Mm8wShared *sh = nullptr;
    rc = quick_build(ctx, x_host, z, zc, L, base, &sh, s); if (rc) return rc;
    // chunks [chunk_lo, chunk_hi): the views keep the buffer's row stride C, the bases move to chunk_lo
    uint32_t *out = coeffs_dev ? (uint32_t *)coeffs_dev + (size_t)chunk_lo * d * 8 : (uint32_t *)(base + L.o_mcan);       // n_store = 0 when there is nothing to store
    rc = quick_launch(ctx, L, base, sh, in, pm, out, dv, coeffs_dev ? cnt * (int64_t)d : 0, coeffs_dev ? d : 0, status_dev, status_dev ? status_dev + 1 : nullptr, cnt, s, bad_map_dev);
    if (rc) return rc;
    HB_HIP(ctx, hipEventRecord((hipEvent_t)sl.ev, s));
    return HB_OK;
}

}  // extern "C"

// the verdict of a launch that cannot hand it over itself (the full-size kernel): status words -> pinned record, sequence number last
__global__ void k_publish_verdict(int32_t *status, FsVerdict *host, int seq) {
    fs_publish_verdict(status, status + 1, nullptr, host, seq);
}

// the context's ONE side stream (highest priority, non-blocking): builds beside a busy caller's stream, the probes' feeds.  One, because a
// process has four hardware queues: a stream per decoder / probe object shares them with the caller's, and what was meant to run beside it
// queues behind it (round 6: bench.py's first-sight leg 5.4 -> 4.5 G once its two-streams leg had made its streams)
static hipError_t ctx_side_stream(hb_ctx *ctx, hipStream_t *out) {
    if (!ctx->side_stream) {
        int lo_prio = 0, hi_prio = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio);
        hipStream_t ss = nullptr;
        const hipError_t e = hipStreamCreateWithPriority(&ss, hipStreamNonBlocking, hi_prio);
        if (e != hipSuccess) return e;
        ctx->side_stream = ss;
    }
    *out = (hipStream_t)ctx->side_stream;
    return hipSuccess;
}
extern "C" int hb_side_stream(hb_ctx *ctx, void **stream) { HB_API_GUARD(ctx);
    if (!ctx || !stream) return HB_ERR_BAD_ARG;
    hipStream_t ss = nullptr;
    HB_HIP(ctx, ctx_side_stream(ctx, &ss));
    *stream = ss;
    return HB_OK;
}

struct hb_quick_dec {
    hb_ctx *ctx;
    int n;
    PointTable *pt;
    bool wide;                    // this set of arrivals goes through the full-size kernel (points that are not small integers, shapes hb_mfma_fused.hip does not take)
    QuickLayout Q;
    const Mm8wShared *qsh;
    std::vector<uint64_t> x;      // the points (the full-size builder looks its tables up by them)
    FsLayout L;
    uint8_t *buf;
    size_t cap;
    int32_t *status;              // device: disagreement flag, first disagreeing chunk, finished workgroups
    FsVerdict *res_host, *res_dev;
    int seq;
    bool prepared;
    std::vector<int32_t> z;
    // What depends on the first d arrivals alone reads no column: it is enqueued on a stream of the decoder's own (highest priority), so it runs
    // BESIDE whatever the caller's stream is busy with (the open's encode while R1's columns come in: the persistent launches leave a few
    // workgroup slots free for exactly this, mm8_trimmed_grid) instead of behind it; the decode launch waits for `built`.
    hipStream_t bstream;
    hipEvent_t built, launched;
    bool launch_open;             // a decode launch whose verdict nobody has read yet may still read `buf`: the next build waits for `launched`
    hipStream_t launch_stream;
    bool beside;                  // hb_quick_dec_beside: builds go to bstream
    bool built_beside;            // the last build went there and no launch has waited for it yet
};

namespace hb {
int quick_dec_supported(hb_quick_dec *qd, int d, int nc, int n_coef) { HB_API_GUARD((qd ? qd->ctx : nullptr));
    if (!qd || d < 1 || d > qd->n || nc < 0 || n_coef < 1 || n_coef > d) return HB_ERR_BAD_ARG;
    FsLayout L;
    if (fs_layout(qd->ctx, qd->pt, d, nc, n_coef, &L) == HB_OK) return HB_OK;
    QuickLayout Q;
    return quick_layout(qd->ctx, qd->n, d, nc, n_coef, &Q);
}
}  // namespace hb

extern "C" {

int hb_quick_dec_create(hb_ctx *ctx, const uint64_t *x_host, int n, hb_quick_dec **out, void *stream) { HB_API_GUARD(ctx);
    if (!ctx || !x_host || !out || n < 1) return HB_ERR_BAD_ARG;
    *out = nullptr;
    if (ctx->n_limbs != 4 || env_hook(ENV_NO_QUICK)) return fail(ctx, HB_ERR_UNSUPPORTED, "quick decoder: narrow context or disabled");
    hipStream_t s = (hipStream_t)stream;
    cache_trim(ctx);
    PointTable *pt = nullptr;
    int rc = point_table(ctx, x_host, n, &pt, s); if (rc) return rc;
    if (!pt->usable) return fail(ctx, HB_ERR_UNSUPPORTED, "quick decoder: repeated points");
    if (n > 65535) return fail(ctx, HB_ERR_UNSUPPORTED, "quick decoder: more than 65535 points");
    hb_quick_dec *qd = new hb_quick_dec();
    qd->ctx = ctx; qd->n = n; qd->pt = pt; qd->wide = false; qd->qsh = nullptr; qd->x.assign(x_host, x_host + (size_t)n * 4); qd->buf = nullptr; qd->cap = 0; qd->status = nullptr; qd->res_host = qd->res_dev = nullptr; qd->seq = 0; qd->prepared = false;
    const int32_t init[4] = {0, INT32_MAX, 0, 0};
    hipError_t e = hipMalloc(&qd->status, sizeof init);
    if (e == hipSuccess) e = hipMemcpyAsync(qd->status, init, sizeof init, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);          // (init is on this stack)
    if (e == hipSuccess) e = hipHostMalloc((void **)&qd->res_host, sizeof(FsVerdict), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&qd->res_dev, qd->res_host, 0);
    qd->bstream = nullptr; qd->built = qd->launched = nullptr; qd->launch_open = false; qd->launch_stream = nullptr; qd->beside = qd->built_beside = false;
    if (e == hipSuccess) e = ctx_side_stream(ctx, &qd->bstream);             // the context's, shared by its decoders and probes
    if (e == hipSuccess) e = hipEventCreateWithFlags(&qd->built, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&qd->launched, hipEventDisableTiming);
    if (e != hipSuccess) {
        if (qd->status) (void)hipFree(qd->status);
        if (qd->res_host) (void)hipHostFree(qd->res_host);
        if (qd->built) (void)hipEventDestroy(qd->built);
        if (qd->launched) (void)hipEventDestroy(qd->launched);
        delete qd;
        ctx->err = std::string("quick decoder: ") + hipGetErrorString(e);
        return HB_ERR_HIP;
    }
    memset(qd->res_host, 0, sizeof(FsVerdict));
    pt->refs++;                                // the table outlives its cache entry while this decoder lives
    *out = qd;
    return HB_OK;
}

int hb_quick_dec_arrivals(hb_quick_dec *qd, const int32_t *z, int d, int nc, int n_coef, void *stream) { HB_API_GUARD((qd ? qd->ctx : nullptr));
    if (!qd || !z || d < 1 || d > qd->n || nc < 0 || n_coef < 1 || n_coef > d) return HB_ERR_BAD_ARG;
    hb_ctx *ctx = qd->ctx;
    hipStream_t s = (hipStream_t)stream;
    qd->prepared = false;
    // Where the build goes: in order on the caller's stream, or -- hb_quick_dec_beside(qd, 1): the caller knows its stream is busy (the open's
    // encode while R1's columns come in; R1's decode launch while R2's first columns come in) -- on the decoder's OWN stream, beside that work
    // (the persistent launches leave workgroup slots free, mm8_trimmed_grid).  It reads no column and nothing the caller's stream produces.
    // The decode launch then waits for an event: ~12 us before the waiting queue moves again (round 6, kernel trace) -- a gain where the
    // build hides behind more than that, a loss on an idle stream, hence the caller's choice.  (Built and dropped, round 6: the builder counting
    // its workgroups in and out and the decode launch polling the count instead of the event -- the launch started as late with two queues
    // active, and an acquire per poll, an invalidation of the XCD's L2, made it 15 us longer.)
    hipStream_t bs = qd->beside ? qd->bstream : s;
    qd->built_beside = false;
    // (a decode launch of this object that nobody has waited for -- an abandoned round -- may still read the buffers a build writes)
    if (qd->launch_open && qd->launch_stream != bs) { HB_HIP(ctx, hipStreamWaitEvent(bs, qd->launched, 0)); }
    FsLayout L;
    int rc = fs_layout(ctx, qd->pt, d, nc, n_coef, &L);
    if (rc) {
        // not a small-integer shape: the full-size kernel's image, built in the same two halves (hb_quick_interp_check's builder)
        QuickLayout Q;
        rc = quick_layout(ctx, qd->n, d, nc, n_coef, &Q);
        if (rc) return rc;
        (void)quick_layout_cand(ctx, &Q);          // a row per party with the first half, where the point set allows it: nothing is built behind the last column
        if (qd->cap < Q.need) {
            if (qd->buf) { HB_HIP(ctx, hipStreamSynchronize(s)); HB_HIP(ctx, hipStreamSynchronize(qd->bstream)); HB_HIP(ctx, hipFree(qd->buf)); }
            qd->buf = nullptr; qd->cap = 0;
            HB_HIP(ctx, hipMalloc(&qd->buf, Q.need));
            qd->cap = Q.need;
        }
        rc = quick_build(ctx, qd->x.data(), z, nullptr, Q, qd->buf, &qd->qsh, bs, 1 | 4); if (rc) return rc;
        if (bs != s) { HB_HIP(ctx, hipEventRecord(qd->built, bs)); qd->built_beside = true; }
        qd->Q = Q;
        qd->wide = true;
        qd->z.assign(z, z + d);
        qd->prepared = true;
        return HB_OK;
    }
    qd->wide = false;
    if (qd->cap < L.need) {
        // (a launch of this decoder may still read the old buffer: decide() waits for its verdict, so only an abandoned one can)
        if (qd->buf) { HB_HIP(ctx, hipStreamSynchronize(s)); HB_HIP(ctx, hipStreamSynchronize(qd->bstream)); HB_HIP(ctx, hipFree(qd->buf)); }
        qd->buf = nullptr; qd->cap = 0;
        HB_HIP(ctx, hipMalloc(&qd->buf, L.need));
        qd->cap = L.need;
    }
    if (L.o_cand && nc > 0) { rc = fs_build_cand(ctx, qd->pt, z, L, qd->buf, qd->status, bs, true); if (rc) return rc; }      // one launch, two workgroups
    else { rc = fs_build(ctx, qd->pt, z, nullptr, L, qd->buf, FS_BUILD_Z, qd->status, bs); if (rc) return rc; }
    if (bs != s) { HB_HIP(ctx, hipEventRecord(qd->built, bs)); qd->built_beside = true; }
    qd->L = L;
    qd->z.assign(z, z + d);
    qd->prepared = true;
    return HB_OK;
}

// the enqueue half of decide, under the context's mutex
static int quick_dec_launch(hb_quick_dec *qd, const int32_t *zc, int nc, const uint64_t *cols_dev, int64_t C, int64_t chunk_lo, int64_t chunk_hi,
                            uint64_t *coeffs_dev, void *stream, int *seq_out) { HB_API_GUARD((qd ? qd->ctx : nullptr));
    if (!qd || !qd->prepared || nc != (qd->wide ? qd->Q.nc : qd->L.nc) || (nc > 0 && !zc) || !cols_dev || C < 1 || chunk_lo < 0 || chunk_hi > C || chunk_lo >= chunk_hi) return HB_ERR_BAD_ARG;
    if (chunk_hi - chunk_lo >= FS_VERDICT_NONE) return fail(qd->ctx, HB_ERR_UNSUPPORTED, "quick decoder: more than 2^30 - 2 chunks a launch");
    hb_ctx *ctx = qd->ctx;
    hipStream_t s = (hipStream_t)stream;
    if (qd->built_beside) { HB_HIP(ctx, hipStreamWaitEvent(s, qd->built, 0)); qd->built_beside = false; }      // the first half, on the decoder's own stream
    if (qd->wide) {
        const QuickLayout &Q = qd->Q;
        int rc = HB_OK;
        if (nc > 0) { rc = quick_build(ctx, qd->x.data(), qd->z.data(), zc, Q, qd->buf, &qd->qsh, s, 2 | 4); if (rc) return rc; }
        const int64_t cnt = chunk_hi - chunk_lo;
        hb_view pm{1, C}, ov = Q.n_coef == 1 ? hb_view{1, C} : hb_view{Q.d, 1};
        const uint32_t *in = (const uint32_t *)cols_dev + (size_t)chunk_lo * 8;
        uint32_t *out = coeffs_dev ? (uint32_t *)coeffs_dev + (size_t)chunk_lo * Q.n_coef * 8 : (uint32_t *)(qd->buf + Q.o_mcan);
        // (the launch's last workgroup publishes the verdict itself when it compares anything; a launch with nothing to compare has none to give)
        FsDone done{qd->status + 2, qd->res_dev, qd->seq + 1};
        rc = quick_launch(ctx, Q, qd->buf, qd->qsh, in, pm, out, ov, coeffs_dev ? (Q.n_coef == 1 ? cnt : cnt * (int64_t)Q.d) : 0, coeffs_dev ? Q.n_coef : 0,
                          qd->status, qd->status + 1, cnt, s, nullptr, &done);
        if (rc) return rc;
        if (Q.nc == 0) { k_publish_verdict<<<1, 1, 0, s>>>(qd->status, qd->res_dev, qd->seq + 1); HB_LAUNCH_CHECK(ctx); }
        HB_HIP(ctx, hipEventRecord(qd->launched, s));
        qd->launch_open = true; qd->launch_stream = s;
        qd->seq += 1;
        qd->prepared = false;
        *seq_out = qd->seq;
        return HB_OK;
    }
    const FsLayout &L = qd->L;
    int rc = HB_OK;
    const bool picked = L.o_cand != 0 && nc > 0;          // the compared senders' rows are waiting in the candidate store
    if (nc > 0 && !picked) { rc = fs_build(ctx, qd->pt, qd->z.data(), zc, L, qd->buf, FS_BUILD_ZC, qd->status, s); if (rc) return rc; }
    if (picked) {
        uint64_t seen[4] = {0, 0, 0, 0};                   // (a candidate store exists for at most 128 parties)
        for (int v : qd->z) seen[v >> 6] |= 1ull << (v & 63);
        for (int j = 0; j < nc; j++) {
            const int v = zc[j];
            if (v < 0 || v >= qd->n || (seen[v >> 6] >> (v & 63) & 1)) return fail(ctx, HB_ERR_BAD_ARG, "quick decoder: compared indices");
            seen[v >> 6] |= 1ull << (v & 63);
        }
    }
    const int64_t cnt = chunk_hi - chunk_lo;
    hb_view pm{1, C}, ov = L.n_coef == 1 ? hb_view{1, C} : hb_view{L.d, 1};
    const uint32_t *in = (const uint32_t *)cols_dev + (size_t)chunk_lo * 8;
    uint32_t *out = coeffs_dev ? (uint32_t *)coeffs_dev + (size_t)chunk_lo * L.n_coef * 8 : nullptr;
    FsDone done{qd->status + 2, qd->res_dev, qd->seq + 1};
    rc = fs_launch(ctx, L, qd->buf, in, pm, out, ov, coeffs_dev ? (L.n_coef == 1 ? cnt : cnt * (int64_t)L.d) : 0, qd->status, qd->status + 1, nullptr, cnt, s, &done,
                   picked ? zc : nullptr);
    if (rc) return rc;
    HB_HIP(ctx, hipEventRecord(qd->launched, s));
    qd->launch_open = true; qd->launch_stream = s;
    qd->seq += 1;
    qd->prepared = false;                      // one verdict per set of arrivals
    *seq_out = qd->seq;
    return HB_OK;
}

int hb_quick_dec_beside(hb_quick_dec *qd, int on) { HB_API_GUARD((qd ? qd->ctx : nullptr));
    if (!qd) return HB_ERR_BAD_ARG;
    qd->beside = on != 0;
    return HB_OK;
}

int hb_quick_dec_launch(hb_quick_dec *qd, const int32_t *zc, int nc, const uint64_t *cols_dev, int64_t C, int64_t chunk_lo, int64_t chunk_hi,
                        uint64_t *coeffs_dev, void *stream) {
    int seq = 0;
    return quick_dec_launch(qd, zc, nc, cols_dev, C, chunk_lo, chunk_hi, coeffs_dev, stream, &seq);
}

int hb_quick_dec_verdict(hb_quick_dec *qd, int32_t *flag, int32_t *first) {
    if (!qd || !flag || !first) return HB_ERR_BAD_ARG;
    hb_ctx *ctx = qd->ctx;
    if (!qd->launch_open) return fail(ctx, HB_ERR_BAD_ARG, "quick decoder: no launch is waiting for its verdict");
    const int seq = qd->seq;
    // outside the context's mutex: poll the sequence number the last workgroup writes after the verdict; past 2 ms, synchronise
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (!fs_verdict_is(qd->res_host, seq)) {
        if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
            HB_HIP(ctx, hipStreamSynchronize(qd->launch_stream));
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!fs_verdict_is(qd->res_host, seq)) return fail(ctx, HB_ERR_HIP, "quick decoder: the kernel finished without a verdict");
    qd->launch_open = false;                   // the launch has ended: nothing reads the image any more
    fs_verdict_read(qd->res_host, flag, first);
    return HB_OK;
}

int hb_quick_dec_decide(hb_quick_dec *qd, const int32_t *zc, int nc, const uint64_t *cols_dev, int64_t C, int64_t chunk_lo, int64_t chunk_hi,
                        uint64_t *coeffs_dev, int32_t *flag, int32_t *first, void *stream) {
    if (!flag || !first) return HB_ERR_BAD_ARG;
    const int rc = hb_quick_dec_launch(qd, zc, nc, cols_dev, C, chunk_lo, chunk_hi, coeffs_dev, stream);
    if (rc) return rc;
    return hb_quick_dec_verdict(qd, flag, first);
}

void hb_quick_dec_destroy(hb_quick_dec *qd) { HB_API_GUARD((qd ? qd->ctx : nullptr));
    if (!qd) return;
    (void)hipDeviceSynchronize();              // an abandoned launch may still be running
    if (qd->buf) (void)hipFree(qd->buf);
    if (qd->status) (void)hipFree(qd->status);
    if (qd->res_host) (void)hipHostFree(qd->res_host);
    if (qd->built) (void)hipEventDestroy(qd->built);
    if (qd->launched) (void)hipEventDestroy(qd->launched);
    point_table_unref(qd->pt);
    delete qd;
}

int hb_probe_create(hb_ctx *ctx, const uint64_t *x_host, int n, int k, hb_probe **out, void *stream) { HB_API_GUARD(ctx);
    if (!ctx || !x_host || !out || n < 1 || k < 1 || k > n) return HB_ERR_BAD_ARG;
    *out = nullptr;
    if (n > PROBE_MAXN) return fail(ctx, HB_ERR_UNSUPPORTED, "probe: more than 256 points");
    if (env_hook(ENV_NO_QUICK)) return fail(ctx, HB_ERR_UNSUPPORTED, "probe: disabled");
    hipStream_t s = (hipStream_t)stream;
    cache_trim(ctx);
    PointTable *pt = nullptr;
    int rc = point_table(ctx, x_host, n, &pt, s); if (rc) return rc;
    if (!pt->usable) return fail(ctx, HB_ERR_UNSUPPORTED, "probe: repeated points");
    pt->refs++;                                // this probe's reference: the table outlives its cache entry while the probe lives
    hb_probe *pr = new hb_probe();
    pr->ctx = ctx; pr->n = n; pr->k = k; pr->pt = pt; pr->poly = -1; pr->seq = 0;
    pr->wgs = n > PROBE_SPLIT_N ? 6 : 1;
    if (const char *e = env_hook(ENV_PROBE_WGS)) { const int v = atoi(e); if (v == 1 || (v >= 2 && v <= PROBE_MAXG)) pr->wgs = v; }
    if (n > PROBE_SPLIT_N && pr->wgs < 2) pr->wgs = 2;       // (one workgroup's 1024 threads hold two units of a point's update each: 5 n + 6 items are 2 x 1286 units at n = 256)
    pr->state_bytes = ((size_t)4 * (pt->S + n) * ctx->nl() + 8 + PROBE_MAXN) * 4;
    // pooled states are all of the largest size: coefficients + values, the fed list, the workgroups' messages (probe_msgs)
    const size_t pool_bytes = (PROBE_STATE_WORDS + PROBE_MSG_WORDS) * 4;
    {
        // (5 S + 8 n) NL words of LDS: 120 KB at the 256-point limit -- above the 64 KB a launch may ask for without saying so.  Per
        // device, not per process: set whenever a probe is created
        const int lim = 128 * 1024;
        hipError_t ae = ctx->n_limbs == 4 ? hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe_feed<9, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, lim)
                                          : hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe_feed<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lim);
        if (ae != hipSuccess) { point_table_unref(pt); delete pr; return fail(ctx, HB_ERR_HIP, "probe: hipFuncSetAttribute"); }
    }
    pr->state = nullptr; pr->res_host = nullptr; pr->res_dev = nullptr;
    if (!ctx->probe_pool.empty()) { pr->state = (uint32_t *)ctx->probe_pool.back(); ctx->probe_pool.pop_back(); }
    else if (hipMalloc(&pr->state, pool_bytes) != hipSuccess || hipMemsetAsync(pr->state + PROBE_STATE_WORDS, 0, PROBE_MSG_WORDS * 4, s) != hipSuccess) {
        if (pr->state) (void)hipFree(pr->state);
        point_table_unref(pt); delete pr; return fail(ctx, HB_ERR_HIP, "probe: hipMalloc");
    }
    if (!ctx->probe_host_pool.empty()) { pr->res_host = (ProbeResult *)ctx->probe_host_pool.back(); ctx->probe_host_pool.pop_back(); }
    else if (hipHostMalloc((void **)&pr->res_host, sizeof(ProbeResult), hipHostMallocMapped) != hipSuccess) {
        ctx->probe_pool.push_back(pr->state); point_table_unref(pt); delete pr; return fail(ctx, HB_ERR_HIP, "probe: hipHostMalloc");
    }
    if (hipHostGetDevicePointer((void **)&pr->res_dev, pr->res_host, 0) != hipSuccess) {
        ctx->probe_pool.push_back(pr->state); ctx->probe_host_pool.push_back(pr->res_host); point_table_unref(pt); delete pr; return fail(ctx, HB_ERR_HIP, "probe: device pointer");
    }
    pr->res_host->seq = 0;                 // a pooled buffer keeps its last owner's number
    *out = pr;
    return HB_OK;
}

// Feed the values polynomial `poly` takes at the parties idx[0..count) -- element `poly` of their columns in the party-major
// buffer cols [n][C] -- and, with decide != 0, wait for the reference's outcome over everything fed: *ok and err_mask[0..n)
// (1 = that party is a root of the error locator).  The points of one polynomial are fed in arrival order, each once;
// hb_probe_reset starts another polynomial (or another arrival list).
// the launch: under the context's mutex like every entry point; *seq_out = the number the kernel will echo when its verdict is complete
static int probe_launch(hb_probe *pr, const int32_t *idx, int count, const uint64_t *cols_dev, int64_t C, int64_t poly, int decide,
                        int32_t *ok, uint8_t *err_mask, void *stream, int *seq_out) { HB_API_GUARD((pr ? pr->ctx : nullptr));
    if (!pr || count < 0 || (count > 0 && (!idx || !cols_dev)) || poly < 0 || poly >= C) return HB_ERR_BAD_ARG;
    if (decide && (!ok || !err_mask)) return HB_ERR_BAD_ARG;
    hb_ctx *ctx = pr->ctx;
    hipStream_t s = (hipStream_t)stream;
    const bool reset = pr->fed.empty();
    if (!reset && poly != pr->poly) return fail(ctx, HB_ERR_BAD_ARG, "probe: another polynomial without a reset");
    if ((int)pr->fed.size() + count > pr->n) return fail(ctx, HB_ERR_BAD_ARG, "probe: more points than parties");
    ProbeIdx ix;
    memset(&ix, 0, sizeof ix);
    // validate first, remember only what a launch was enqueued for: a refused call leaves the probe as it was
    for (int i = 0; i < count; i++) {
        if (idx[i] < 0 || idx[i] >= pr->n || std::find(pr->fed.begin(), pr->fed.end(), idx[i]) != pr->fed.end() || std::find(idx, idx + i, idx[i]) != idx + i)
            return fail(ctx, HB_ERR_BAD_ARG, "probe: point fed twice or out of range");
        ix.idx[i] = (uint16_t)idx[i];
    }
    if (count == 0 && reset) return fail(ctx, HB_ERR_BAD_ARG, "probe: nothing fed yet");
    const int S = pr->pt->S, NLr = ctx->nl();
    const int seq = pr->seq + 1;
    // (a number no other launch into a pooled state buffer carries: the workgroups' messages are recognised by it)
    static std::atomic<uint32_t> launches{0};
    const uint32_t uq = (launches.fetch_add(1) + 1) & 0x7fffffu;
    uint32_t *msgs = pr->state + PROBE_STATE_WORDS;
    const unsigned grid = pr->wgs > 1 ? 8u * (unsigned)(pr->wgs - 1) + 1u : 1u;
    const size_t lds = ((size_t)(5 * S + 8 * pr->n) * NLr) * 4;      // coefficients, values, the decision's scratch polynomial, the points, this launch's symbols and discrepancies
    if (ctx->n_limbs == 4)
        k_probe_feed<9, 8><<<grid, PROBE_NT, lds, s>>>(ctx->pw, pr->pt->xm, pr->pt->pw, pr->n, S, pr->state, ix, count, reset ? 1 : 0,
                                               (const uint32_t *)cols_dev, C, poly, pr->k, decide ? 1 : 0, pr->res_dev, seq, msgs, uq);
    else
        k_probe_feed<3, 2><<<grid, PROBE_NT, lds, s>>>(ctx->pn, pr->pt->xm, pr->pt->pw, pr->n, S, pr->state, ix, count, reset ? 1 : 0,
                                               (const uint32_t *)cols_dev, C, poly, pr->k, decide ? 1 : 0, pr->res_dev, seq, msgs, uq);
    {
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            // nothing ran: whatever the device state holds is no longer described by `fed` -- the next call starts from a reset
            pr->fed.clear();
            pr->poly = -1;
            ctx->err = std::string("probe launch: ") + hipGetErrorString(le);
            return HB_ERR_HIP;
        }
    }
    pr->seq = seq;
    pr->poly = poly;
    for (int i = 0; i < count; i++) pr->fed.push_back(idx[i]);
    *seq_out = seq;
    return HB_OK;
}

int hb_probe_feed(hb_probe *pr, const int32_t *idx, int count, const uint64_t *cols_dev, int64_t C, int64_t poly, int decide,
                  int32_t *ok, uint8_t *err_mask, void *stream) {
    int seq = 0;
    const int rc = probe_launch(pr, idx, count, cols_dev, C, poly, decide, ok, err_mask, stream, &seq);
    if (rc || !decide) return rc;
    hb_ctx *ctx = pr->ctx;
    hipStream_t s = (hipStream_t)stream;
    {
        // Outside the context's mutex (other threads of the context go on while this one waits).  The kernel writes its verdict
        // into pinned host memory and the sequence number last: polling that word sees it a few microseconds after the last
        // store instead of a stream synchronisation's wake-up later; past 2 ms, synchronise
        volatile int32_t *flag = &pr->res_host->seq;
        const auto t0 = std::chrono::steady_clock::now();
        int spins = 0;
        while (*flag != seq) {
            if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                HB_HIP(ctx, hipStreamSynchronize(s));
                break;
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (*flag != seq) return fail(ctx, HB_ERR_HIP, "probe: the kernel finished without a verdict");
    }
    if (pr->res_host->ok < 0) { pr->fed.clear(); pr->poly = -1; return fail(ctx, HB_ERR_RETRY, "probe: a workgroup of the launch waited in vain for another (the launch is void; the probe starts from a reset)"); }
    *ok = pr->res_host->ok;
    memcpy(err_mask, pr->res_host->err, (size_t)pr->n);
    if (!*ok) memset(err_mask, 0, (size_t)pr->n);
    return HB_OK;
}

int hb_symbols_fetch(hb_ctx *ctx, const uint64_t *cols_dev, int n, int64_t C, int64_t chunk, const int32_t *idx, int count, uint64_t *out_host, void *stream) {
    HB_API_GUARD(ctx);
    if (!ctx || !cols_dev || !idx || !out_host || n < 1 || count < 1 || count > FETCH_MAX || C < 1 || chunk < 0 || chunk >= C) return fail(ctx, HB_ERR_BAD_ARG, "symbols_fetch: arguments");
    for (int i = 0; i < count; i++)
        if (idx[i] < 0 || idx[i] >= n) return fail(ctx, HB_ERR_BAD_ARG, "symbols_fetch: party index out of range");
    if (!ctx->fetch_host) {
        void *h = nullptr, *dv = nullptr;
        HB_HIP(ctx, hipHostMalloc(&h, sizeof(SymFetch), hipHostMallocMapped));
        if (hipHostGetDevicePointer(&dv, h, 0) != hipSuccess) { (void)hipHostFree(h); return fail(ctx, HB_ERR_HIP, "symbols_fetch: device pointer"); }
        memset(h, 0, sizeof(SymFetch));
        ctx->fetch_host = h; ctx->fetch_dev = dv;
    }
    SymFetch *host = static_cast<SymFetch *>(ctx->fetch_host);
    FetchIdx ix;
    for (int i = 0; i < count; i++) ix.idx[i] = idx[i];
    const int L = ctx->n_limbs, seq = ++ctx->fetch_seq;
    hipStream_t s = (hipStream_t)stream;
    k_symbols_fetch<<<1, 256, 0, s>>>(cols_dev, C, chunk, L, ix, count, static_cast<SymFetch *>(ctx->fetch_dev), seq);
    HB_LAUNCH_CHECK(ctx);
    // (the buffer belongs to the context: the wait stays inside its mutex -- it is a few microseconds; past 2 ms, synchronise)
    volatile int32_t *flag = &host->seq;
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (*flag != seq) {
        if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
            HB_HIP(ctx, hipStreamSynchronize(s));
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (*flag != seq) return fail(ctx, HB_ERR_HIP, "symbols_fetch: the kernel finished without handing over");
    memcpy(out_host, host->w, (size_t)count * L * sizeof(uint64_t));
    return HB_OK;
}

int hb_candidate_check(hb_ctx *ctx, const uint64_t *x_host, int n, const uint64_t *coeffs_dev, int d, const uint64_t *cols_dev, int64_t C, int64_t chunk,
                       uint64_t *out_values_host, uint8_t *out_differs_host, void *stream) { HB_API_GUARD(ctx);
    if (!ctx || !x_host || !coeffs_dev || !cols_dev || !out_values_host || !out_differs_host || n < 1 || d < 1 || C < 1 || chunk < 0 || chunk >= C)
        return fail(ctx, HB_ERR_BAD_ARG, "candidate_check: arguments");
    if (n > CAND_MAXN) return fail(ctx, HB_ERR_UNSUPPORTED, "candidate_check: more than 1024 points");
    hipStream_t s = (hipStream_t)stream;
    if (!ctx->cand_host) {
        void *h = nullptr, *dv = nullptr;
        HB_HIP(ctx, hipHostMalloc(&h, sizeof(CandCheck), hipHostMallocMapped));
        if (hipHostGetDevicePointer(&dv, h, 0) != hipSuccess || hipMalloc(&ctx->cand_ticket, sizeof(int32_t)) != hipSuccess ||
            hipMemsetAsync(ctx->cand_ticket, 0, sizeof(int32_t), s) != hipSuccess) {
            (void)hipHostFree(h);
            if (ctx->cand_ticket) { (void)hipFree(ctx->cand_ticket); ctx->cand_ticket = nullptr; }
            return fail(ctx, HB_ERR_HIP, "candidate_check: buffers");
        }
        memset(h, 0, sizeof(CandCheck));
        ctx->cand_host = h; ctx->cand_dev = dv;
    }
    uint32_t *xd = nullptr;
    { const int rcx = points_on_device(ctx, x_host, n, &xd, s); if (rcx) return rcx; }
    CandCheck *host = static_cast<CandCheck *>(ctx->cand_host);
    const int seq = ++ctx->cand_seq;
    if (ctx->n_limbs == 4)
        k_candidate_check<9, 8><<<(unsigned)n, 128, 0, s>>>(ctx->pw, xd, n, (const uint32_t *)coeffs_dev, d, (const uint32_t *)cols_dev, C, chunk, static_cast<CandCheck *>(ctx->cand_dev), ctx->cand_ticket, seq);
    else
        k_candidate_check<3, 2><<<(unsigned)n, 128, 0, s>>>(ctx->pn, xd, n, (const uint32_t *)coeffs_dev, d, (const uint32_t *)cols_dev, C, chunk, static_cast<CandCheck *>(ctx->cand_dev), ctx->cand_ticket, seq);
    HB_LAUNCH_CHECK(ctx);
    // (the buffer belongs to the context: the wait stays inside its mutex -- it is some tens of microseconds; past 5 ms, synchronise)
    volatile int32_t *flag = &host->seq;
    const auto t0 = std::chrono::steady_clock::now();
    int spins = 0;
    while (*flag != seq) {
        if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) {
            HB_HIP(ctx, hipStreamSynchronize(s));
            break;
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (*flag != seq) return fail(ctx, HB_ERR_HIP, "candidate_check: the kernel finished without handing over");
    memcpy(out_values_host, host->vals, (size_t)n * ctx->n_limbs * sizeof(uint64_t));
    memcpy(out_differs_host, host->differs, (size_t)n);
    return HB_OK;
}

// `stream` goes on only after everything enqueued on `after` so far (one event of the context, recorded and waited for here: what
// torch's Stream.wait_stream does in 30 us of Python)
int hb_stream_after(hb_ctx *ctx, void *stream, void *after) { HB_API_GUARD(ctx);
    if (!ctx) return HB_ERR_BAD_ARG;
    if (stream == after) return HB_OK;
    if (!ctx->after_event) {
        hipEvent_t ev = nullptr;
        HB_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->after_event = ev;
    }
    HB_HIP(ctx, hipEventRecord((hipEvent_t)ctx->after_event, (hipStream_t)after));
    HB_HIP(ctx, hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ctx->after_event, 0));
    return HB_OK;
}

int hb_probe_workgroups(hb_probe *pr, int wgs) { HB_API_GUARD((pr ? pr->ctx : nullptr));
    if (!pr) return HB_ERR_BAD_ARG;
    const int least = pr->n > PROBE_SPLIT_N ? 2 : 1;
    if (wgs == 0) wgs = least;
    if (wgs < least || wgs > PROBE_MAXG || !pr->fed.empty()) return fail(pr->ctx, HB_ERR_BAD_ARG, "probe: workgroup count (or points already fed)");
    pr->wgs = wgs;
    return HB_OK;
}

int hb_probe_reset(hb_probe *pr) { HB_API_GUARD((pr ? pr->ctx : nullptr));
    if (!pr) return HB_ERR_BAD_ARG;
    pr->fed.clear();
    pr->poly = -1;
    return HB_OK;
}

int hb_probe_points_fed(hb_probe *pr) { return pr ? (int)pr->fed.size() : 0; }

void hb_probe_destroy(hb_probe *pr) { HB_API_GUARD((pr ? pr->ctx : nullptr));
    if (!pr) return;
    // a kernel of this probe may still be running: the buffers go back to the pool only once the device is done with them
    (void)hipDeviceSynchronize();
    pr->ctx->probe_pool.push_back(pr->state);
    pr->ctx->probe_host_pool.push_back(pr->res_host);
    point_table_unref(pr->pt);
    delete pr;
}

}  // extern "C"
