// hb_wb.hip -- Welch-Berlekamp Reed-Solomon decoding, batched: one workgroup per codeword.
//
// Reference functions replaced (pure Python in the reference, paths under /root/reference):
//   make_wb_encoder_decoder.decode / solve_system   honeybadgermpc/reed_solomon_wb.py:79-151
//   rref / no_solution / is_pivot_column / some_solution   reed_solomon_wb.py:157-273
//   Polynomial.__divmod__ (exact division Q / E)            polynomial.py:219-234
//
// For e' = e .. 1 (e = (n - c - t) // 2): build the (n'+1) x (2e'+k+2) system
//     b_i * E(a_i) - Q(a_i) = 0   for every non-erased point,   coefficient of X^e' in E = 1,
// reduce it, set free variables to 1, and accept the first e' for which E | Q.
//
// The reference normalises every pivot row (one field inversion per pivot, ~n of them).
// Here the elimination is FRACTION-FREE -- row_r <- piv * row_r - row_r[j] * row_piv --
// which needs no inversion and spans the same row space, so the reduced row echelon form
// (which is unique) is recovered at the end by scaling each pivot row with the inverse of
// its pivot; all pivots are inverted together with Montgomery's trick (ONE inversion per
// attempt).  The classification of columns into pivot / free variables replicates
// is_pivot_column on that RREF, including its quirk of treating a free column that happens
// to look like a unit vector as a pivot column, so the solution vector -- and therefore the
// result outside the unique-decoding radius and the "No solution" / "found no divisors!"
// outcomes -- is identical to the reference's.
//
// Matrix: Montgomery digits in HBM scratch (one slab per resident workgroup; at n = 100 a
// slab is 371 KB, which does not fit the 160 KB LDS), multipliers and the pivot row staged in
// LDS for every elimination step.
#include "hb_common.hpp"

using namespace hb;

namespace {

constexpr int WB_THREADS = 256;
constexpr int WB_MAXROWS = 264;   // n' + 1 <= 264: config 5's 256 parties with every symbol present are 257 rows
constexpr int WB_MAXCOLS = 272;

template <int NL> __device__ __forceinline__ void mget(uint32_t (&d)[NL], const uint32_t *p) {
#pragma unroll
    for (int q = 0; q < NL; q++) d[q] = p[q];
}
template <int NL> __device__ __forceinline__ void mput(uint32_t *p, const uint32_t (&d)[NL]) {
#pragma unroll
    for (int q = 0; q < NL; q++) p[q] = d[q];
}
template <int NL> __device__ __forceinline__ bool mzero(const uint32_t *p) {
    uint32_t o = 0;
#pragma unroll
    for (int q = 0; q < NL; q++) o |= p[q];
    return o == 0;
}
template <int NL> __device__ __forceinline__ bool meq(const uint32_t *a, const uint32_t *b) {
    uint32_t o = 0;
#pragma unroll
    for (int q = 0; q < NL; q++) o |= a[q] ^ b[q];
    return o == 0;
}

template <int NL, int NW>
__global__ void __launch_bounds__(WB_THREADS) k_wb(const FpParams<NL> P, const uint32_t *__restrict__ xm /* [n][NL] mont */,
                                                   const uint32_t *__restrict__ ys, const uint8_t *__restrict__ present,
                                                   int n, int k, int64_t C, uint32_t *__restrict__ scratch, size_t slab_words,
                                                   uint32_t *__restrict__ coeffs, int32_t *__restrict__ coeff_len, int32_t *__restrict__ status,
                                                   const int32_t *__restrict__ todo /* nullptr: all C codewords; else the C indices Gao left undecided */) {
    __shared__ uint32_t s_mult[WB_MAXROWS * NL];     // column-j multipliers of every row
    __shared__ uint32_t s_prow[WB_MAXCOLS * NL];     // pivot row / later: solution vector
    __shared__ uint32_t s_dinv[WB_MAXROWS * NL];     // inverse pivots
    __shared__ uint32_t s_wrk[WB_MAXCOLS * NL];      // Q work copy / quotient
    __shared__ int16_t s_perm[WB_MAXROWS];           // row order -> physical row
    __shared__ int16_t s_pivcol[WB_MAXROWS];         // pivot column of the row at each position
    __shared__ int16_t s_idx[WB_MAXROWS];            // present point indices, ascending
    __shared__ int16_t s_cls[WB_MAXCOLS];            // per variable: -1 free, else the row position of its pivot
    __shared__ int16_t s_free[WB_MAXCOLS];           // list of true free columns (for elimination range)
    __shared__ int s_i[8];
    const int tid = threadIdx.x;
    uint32_t *M = scratch + (size_t)blockIdx.x * slab_words;

    for (int64_t it = blockIdx.x; it < C; it += gridDim.x) {
        const int64_t cw = todo ? (int64_t)todo[it] : it;    // block-uniform
        const uint32_t *y = ys + (size_t)cw * n * NW;
        const uint8_t *pr = present + (size_t)cw * n;
        __syncthreads();
        if (tid == 0) {
            int np_ = 0;
            for (int i = 0; i < n; i++) if (pr[i]) s_idx[np_++] = (int16_t)i;
            s_i[0] = np_;
        }
        __syncthreads();
        const int np = s_i[0];
        const int t = k - 1, cer = n - np;
        uint32_t *out = coeffs + (size_t)cw * k * NW;
        // zero the output row
        for (int i = tid; i < k * NW; i += WB_THREADS) out[i] = 0;
        if (2 * t + 1 + cer > n) {                   // assert 2t+1+c <= n  (reed_solomon_wb.py:132)
            if (tid == 0) { status[cw] = 3; coeff_len[cw] = 0; }
            continue;
        }
        const int e = (np - t) / 2;
        if (e == 0) {                                // only k == 1, n' == 1: plain interpolation (:142-145)
            if (tid == 0) {
                uint32_t w[NW];
                load_words<NW>(w, y + (size_t)s_idx[0] * NW);
                store_words<NW>(out, w);
                // length 1 also for a zero symbol: the reference's Polynomial.interpolate returns the zero polynomial as [0]
                // (strip_trailing_zeros keeps one zero of an all-zero list, polynomial.py:14-20; oracle/diff_wb_vs_reference.py)
                status[cw] = 0; coeff_len[cw] = 1;
            }
            continue;
        }
        int result_status = 1;                       // "found no divisors!" unless an e' succeeds
        for (int ee = e; ee >= 1; ee--) {
            const int env = ee + 1, qnv = ee + k, cols = env + qnv + 1, rows = np + 1;
            // ---- build the system ---------------------------------------------------
            for (int r_ = tid; r_ < np; r_ += WB_THREADS) {
                uint32_t a[NL], bd[NL], b[NL], pw[NL];
                mget<NL>(a, xm + (size_t)s_idx[r_] * NL);
                load_digits<NL, NW>(bd, y + (size_t)s_idx[r_] * NW);
                to_mont(b, bd, P);
                fp_set(pw, P.one);
                uint32_t *row = M + (size_t)r_ * cols * NL;
                for (int j = 0; j < qnv; j++) {
                    if (j < env) { uint32_t v[NL]; mont_mul(v, b, pw, P); mput<NL>(row + (size_t)j * NL, v); }
                    uint32_t ng[NL]; fp_neg(ng, pw, P); mput<NL>(row + (size_t)(env + j) * NL, ng);
                    mont_mul(pw, pw, a, P);
                }
                uint32_t z[NL];
#pragma unroll
                for (int q = 0; q < NL; q++) z[q] = 0;
                mput<NL>(row + (size_t)(cols - 1) * NL, z);
            }
            {
                uint32_t *row = M + (size_t)np * cols * NL;
                for (int j = tid; j < cols; j += WB_THREADS) {
                    uint32_t v[NL];
#pragma unroll
                    for (int q = 0; q < NL; q++) v[q] = (j == env - 1 || j == cols - 1) ? P.one[q] : 0u;
                    mput<NL>(row + (size_t)j * NL, v);
                }
            }
            for (int r = tid; r < rows; r += WB_THREADS) s_perm[r] = (int16_t)r;
            __syncthreads();
            // ---- fraction-free Gauss-Jordan ----------------------------------------------
            int ipos = 0, nfree = 0;
            bool inconsistent = false;
            for (int j = 0; j < cols && ipos < rows; j++) {
                if (tid == 0) s_i[1] = rows;
                __syncthreads();
                for (int r = ipos + tid; r < rows; r += WB_THREADS)
                    if (!mzero<NL>(M + ((size_t)s_perm[r] * cols + j) * NL)) atomicMin(&s_i[1], r);
                __syncthreads();
                const int found = s_i[1];
                if (found == rows) {                 // no pivot in this column: free variable
                    if (tid == 0 && j < cols - 1) s_free[nfree] = (int16_t)j;
                    if (j < cols - 1) nfree++;
                    __syncthreads();
                    continue;
                }
                if (j == cols - 1) { inconsistent = true; break; }   // pivot in the constants column
                if (tid == 0) {
                    int16_t tmp = s_perm[ipos]; s_perm[ipos] = s_perm[found]; s_perm[found] = tmp;
                    s_pivcol[ipos] = (int16_t)j;
                }
                __syncthreads();
                const int prow = s_perm[ipos];
                const int cmin = nfree > 0 ? min((int)s_free[0], j) : j;
                const int width = cols - cmin;
                // stage multipliers (column j of every row) and the pivot row
                for (int r = tid; r < rows; r += WB_THREADS) {
                    uint32_t v[NL]; mget<NL>(v, M + ((size_t)s_perm[r] * cols + j) * NL); mput<NL>(s_mult + (size_t)r * NL, v);
                }
                for (int cc = tid; cc < width; cc += WB_THREADS) {
                    uint32_t v[NL]; mget<NL>(v, M + ((size_t)prow * cols + cmin + cc) * NL); mput<NL>(s_prow + (size_t)cc * NL, v);
                }
                __syncthreads();
                uint32_t piv[NL];
                mget<NL>(piv, s_mult + (size_t)ipos * NL);
                const int tot = rows * width;
                for (int idx = tid; idx < tot; idx += WB_THREADS) {
                    const int r = idx / width, cc = idx % width;
                    if (r == ipos) continue;
                    uint32_t m[NL];
                    mget<NL>(m, s_mult + (size_t)r * NL);
                    if (fp_is_zero(m)) continue;     // rsdecode: rows with a zero in column j are left alone (:187)
                    uint32_t *cell = M + ((size_t)s_perm[r] * cols + cmin + cc) * NL;
                    uint32_t v[NL], pv[NL], t1[NL], t2[NL], res[NL];
                    mget<NL>(v, cell);
                    mget<NL>(pv, s_prow + (size_t)cc * NL);
                    mont_mul(t1, piv, v, P);
                    mont_mul(t2, m, pv, P);
                    fp_sub(res, t1, t2, P);
                    mput<NL>(cell, res);
                }
                // a scaled row also scales its own (earlier) pivot entry, which lies left of cmin
                for (int r = tid; r < ipos; r += WB_THREADS) {
                    const int pc = s_pivcol[r];
                    if (pc >= cmin) continue;        // already covered by the range update
                    uint32_t m[NL];
                    mget<NL>(m, s_mult + (size_t)r * NL);
                    if (fp_is_zero(m)) continue;
                    uint32_t *cell = M + ((size_t)s_perm[r] * cols + pc) * NL;
                    uint32_t v[NL], res[NL];
                    mget<NL>(v, cell);
                    mont_mul(res, piv, v, P);
                    mput<NL>(cell, res);
                }
                __syncthreads();
                ipos++;
            }
            __syncthreads();
            if (inconsistent) { result_status = 2; break; }          // raise Exception("No solution") (:245)
            const int npiv = ipos;
            const int nvars = cols - 1;
            // ---- invert all pivots at once (Montgomery's trick) --------------------------
            if (tid == 0) {
                uint32_t acc[NL];
                fp_set(acc, P.one);
                for (int r = 0; r < npiv; r++) {      // prefix products into s_dinv
                    uint32_t pv[NL];
                    mget<NL>(pv, M + ((size_t)s_perm[r] * cols + s_pivcol[r]) * NL);
                    mput<NL>(s_dinv + (size_t)r * NL, acc);
                    mont_mul(acc, acc, pv, P);
                }
                uint32_t inv[NL];
                fp_inv(inv, acc, P);
                for (int r = npiv - 1; r >= 0; r--) {
                    uint32_t pv[NL], pre[NL], d[NL];
                    mget<NL>(pv, M + ((size_t)s_perm[r] * cols + s_pivcol[r]) * NL);
                    mget<NL>(pre, s_dinv + (size_t)r * NL);
                    mont_mul(d, inv, pre, P);
                    mput<NL>(s_dinv + (size_t)r * NL, d);
                    mont_mul(inv, inv, pv, P);
                }
            }
            // ---- classify the variables like is_pivot_column on the RREF (:217-237) ----------
            for (int j = tid; j < nvars; j += WB_THREADS) {
                int first = -1; bool others_zero = true;
                for (int r = 0; r < npiv; r++) {
                    if (!mzero<NL>(M + ((size_t)s_perm[r] * cols + j) * NL)) {
                        if (first < 0) first = r; else { others_zero = false; break; }
                    }
                }
                int cls = -1;
                if (first >= 0 && others_zero) {
                    // normalised entry == 1  <=>  entry == the row's pivot entry
                    const uint32_t *row = M + (size_t)s_perm[first] * cols * NL;
                    if (meq<NL>(row + (size_t)j * NL, row + (size_t)s_pivcol[first] * NL)) cls = first;
                }
                s_cls[j] = (int16_t)cls;
            }
            __syncthreads();
            // ---- solution: free variables = 1, pivot variables from their row (:254-271) -----
            for (int j = tid; j < nvars; j += WB_THREADS) {
                uint32_t val[NL];
                const int r = s_cls[j];
                if (r < 0) fp_set(val, P.one);
                else {
                    const uint32_t *row = M + (size_t)s_perm[r] * cols * NL;
                    uint32_t acc[NL];
                    mget<NL>(acc, row + (size_t)(cols - 1) * NL);
                    for (int f = 0; f < nvars; f++) {
                        if (s_cls[f] >= 0) continue;
                        uint32_t v[NL];
                        mget<NL>(v, row + (size_t)f * NL);
                        fp_sub(acc, acc, v, P);
                    }
                    uint32_t dv[NL];
                    mget<NL>(dv, s_dinv + (size_t)r * NL);
                    mont_mul(val, acc, dv, P);
                }
                mput<NL>(s_prow + (size_t)j * NL, val);
            }
            __syncthreads();
            // ---- E | Q ?  (E = sol[0..env), Q = sol[env..), both with trailing zeros stripped) --
            const uint32_t *E = s_prow;
            const uint32_t *Q = s_prow + (size_t)env * NL;
            if (tid == 0) {
                int dE = env - 1; while (dE >= 0 && mzero<NL>(E + (size_t)dE * NL)) dE--;
                int dQ = qnv - 1; while (dQ >= 0 && mzero<NL>(Q + (size_t)dQ * NL)) dQ--;
                s_i[2] = dE; s_i[3] = dQ;
            }
            for (int j = tid; j < qnv; j += WB_THREADS) { uint32_t v[NL]; mget<NL>(v, Q + (size_t)j * NL); mput<NL>(s_wrk + (size_t)j * NL, v); }
            __syncthreads();
            const int dE = s_i[2], dQ = s_i[3];
            bool exact;
            int dP = -1;
            if (dE < 0) exact = false;               // cannot happen: coefficient ee of E is pinned to 1
            else if (dQ < dE) exact = (dQ < 0);      // quotient 0, remainder Q
            else {
                // long division of the work copy by E (leading coefficient lcE, inverted once)
                uint32_t lcinv[NL];
                {
                    uint32_t lc[NL];
                    mget<NL>(lc, E + (size_t)dE * NL);
                    if (fp_eq(lc, P.one)) fp_set(lcinv, P.one); else fp_inv(lcinv, lc, P);
                }
                uint32_t *quo = s_mult;               // reuse: quotient coefficients
                for (int i = dQ - dE; i >= 0; i--) {
                    uint32_t top[NL], coef[NL];
                    mget<NL>(top, s_wrk + (size_t)(i + dE) * NL);
                    mont_mul(coef, top, lcinv, P);
                    __syncthreads();
                    if (tid == 0) mput<NL>(quo + (size_t)i * NL, coef);
                    for (int idx = tid; idx <= dE; idx += WB_THREADS) {
                        uint32_t u[NL], w[NL], t2[NL], r[NL];
                        mget<NL>(u, s_wrk + (size_t)(i + idx) * NL);
                        mget<NL>(w, E + (size_t)idx * NL);
                        mont_mul(t2, coef, w, P);
                        fp_sub(r, u, t2, P);
                        mput<NL>(s_wrk + (size_t)(i + idx) * NL, r);
                    }
                    __syncthreads();
                }
                if (tid == 0) {
                    int rem = dE - 1; while (rem >= 0 && mzero<NL>(s_wrk + (size_t)rem * NL)) rem--;
                    int dp = dQ - dE; while (dp >= 0 && mzero<NL>(quo + (size_t)dp * NL)) dp--;
                    s_i[4] = rem; s_i[5] = dp;
                }
                __syncthreads();
                exact = s_i[4] < 0;
                dP = s_i[5];
            }
            if (exact) {
                // P = Q / E, coefficients stripped of trailing zeros (polynomial.py:36)
                for (int i = tid; i <= dP && i < k; i += WB_THREADS) {
                    uint32_t v[NL], cnn[NL];
                    mget<NL>(v, s_mult + (size_t)i * NL);
                    from_mont(cnn, v, P);
                    store_digits<NL, NW>(out + (size_t)i * NW, cnn);
                }
                if (tid == 0) coeff_len[cw] = dP + 1;
                result_status = 0;
                __syncthreads();
                break;
            }
            __syncthreads();
        }
        if (tid == 0) { status[cw] = result_status; if (result_status != 0) coeff_len[cw] = 0; }
    }
}

template <int NL, int NW>
__global__ void k_points_to_mont(const FpParams<NL> P, const uint32_t *__restrict__ x, int n, uint32_t *__restrict__ xm) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t d[NL], m[NL];
    load_digits<NL, NW>(d, x + (size_t)i * NW);
    to_mont(m, d, P);
#pragma unroll
    for (int q = 0; q < NL; q++) xm[(size_t)i * NL + q] = m[q];
}

}  // namespace

// any erasure in the batch (bit 0)?  does some codeword's pattern differ from the first one's (bit 1)?
__global__ void k_wb_erasure_pattern(const uint8_t *__restrict__ present, int64_t total, int n, int32_t *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const bool here = present[i] != 0, first = present[i % n] != 0;
    int bits = (here ? 0 : 1) | (here != first ? 2 : 0);
    if (bits && (*reinterpret_cast<volatile int32_t *>(flag) & bits) != bits) atomicOr(flag, bits);
}
// Gao's coefficient rows -> the reference's outcome for the codewords it decoded WITHIN THE RADIUS: status 0 and the length after
// stripping trailing zeros (polynomial.py:14-20); the others are left to k_wb.  Gao's acceptance alone is not enough: a message of
// low degree (leading coefficients zero) is decoded by Gao with MORE than floor((n - k) / 2) errors (deg r = deg f + e stays below
// (n + k) / 2), and out there the reference's solver is on its own -- at e' = 11 of n = 25, k = 4 it meets an underdetermined system,
// takes a particular solution that does not divide and ends in "found no divisors!" where Gao returns the constant polynomial
// (scratch/stress_gao.py found the word).  So: only locators of degree <= emax count.
template <int NW>
__global__ void k_wb_take_gao(const uint8_t *__restrict__ ok, const int32_t *__restrict__ errlen, int emax, const uint32_t *__restrict__ coeffs, int k, int64_t C,
                              int32_t *__restrict__ coeff_len, int32_t *__restrict__ status, int32_t *__restrict__ rejected,
                              int32_t *__restrict__ todo) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (!ok[c] || errlen[c] - 1 > emax) { todo[atomicAdd(rejected, 1)] = (int32_t)c; return; }   // the row reduction's work list (any order: codewords are independent)
    int len = k;
    while (len > 0) {
        const uint32_t *e = coeffs + ((size_t)c * k + (len - 1)) * NW;
        uint32_t o = 0;
        for (int q = 0; q < NW; q++) o |= e[q];
        if (o) break;
        len--;
    }
    coeff_len[c] = len;
    status[c] = 0;
}

// ---- per-codeword erasure patterns: a batch with a FEW distinct patterns is a few batches with a shared one ---------------------------
// presence bytes -> a bitmask per codeword (W = ceil(n / 64) words)
__global__ void k_wb_masks(const uint8_t *__restrict__ present, int64_t C, int n, int W, unsigned long long *__restrict__ masks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * W) return;
    const int64_t c = i / W;
    const int w = (int)(i - c * W);
    unsigned long long m = 0;
    for (int b = 0; b < 64 && 64 * w + b < n; b++)
        if (present[c * n + 64 * w + b]) m |= 1ull << b;
    masks[i] = m;
}
// the group's codewords (all their n symbols: the interpolant's launch picks the surviving ones by its row map) next to each other
__global__ void k_wb_gather(const unsigned long long *__restrict__ ys, const int32_t *__restrict__ idx, int64_t Cg, int row_q, unsigned long long *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Cg * row_q) return;
    const int64_t g = i / row_q;
    out[i] = ys[(size_t)idx[g] * row_q + (i - g * row_q)];
}
// k_wb_take_gao for a gathered group: codeword g of the group is codeword idx[g] of the batch; an accepted row is copied to its place
template <int NW>
__global__ void k_wb_take_gao_group(const uint8_t *__restrict__ ok, const int32_t *__restrict__ errlen, int emax, const uint32_t *__restrict__ gcoeffs, int k, int64_t Cg,
                                    const int32_t *__restrict__ idx, uint32_t *__restrict__ coeffs, int32_t *__restrict__ coeff_len, int32_t *__restrict__ status,
                                    int32_t *__restrict__ rejected, int32_t *__restrict__ todo) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= Cg) return;
    const int32_t c = idx[g];
    if (!ok[g] || errlen[g] - 1 > emax) { todo[atomicAdd(rejected, 1)] = c; return; }
    int len = 0;
    for (int j = 0; j < k; j++) {
        uint32_t o = 0;
        for (int q = 0; q < NW; q++) { const uint32_t v = gcoeffs[((size_t)g * k + j) * NW + q]; coeffs[((size_t)c * k + j) * NW + q] = v; o |= v; }
        if (o) len = j + 1;
    }
    coeff_len[c] = len;
    status[c] = 0;
}
constexpr int WB_MAX_PATTERNS = 64;        // distinct erasure patterns a batch is cut into; more: the row reduction, as before
constexpr int WB_MIN_GROUP = 64;           // a pattern's codewords go through Gao's kernels from this many on (its tables cost ~a millisecond)

extern "C" int hb_wb_decode(hb_ctx *ctx, const uint64_t *x_host, int n, int k, const uint64_t *ys_dev,
                            const uint8_t *present_dev, int64_t C, uint64_t *coeffs_dev, int32_t *coeff_len_dev,
                            int32_t *status_dev, void *stream) { HB_API_GUARD(ctx);
    if (!ctx || !x_host || n < 1 || k < 1 || k > n || C < 0) return HB_ERR_BAD_ARG;
    if (C == 0) return HB_OK;
    if (!ys_dev || !present_dev || !coeffs_dev || !coeff_len_dev || !status_dev) return HB_ERR_BAD_ARG;
    if (n + 1 > WB_MAXROWS || n + 4 > WB_MAXCOLS) return fail(ctx, HB_ERR_UNSUPPORTED, "welch-berlekamp: n > 263");
    hipStream_t s = (hipStream_t)stream;
    const int NLr = ctx->nl();
    // every temporary of this call is released on every way out (the HB_HIP / launch checks return early);
    // hipFree waits for the device, so nothing in flight loses its buffers
    struct Temps {
        std::vector<void *> bufs;
        ~Temps() { for (void *q : bufs) if (q) (void)hipFree(q); }
        int alloc(hb_ctx *c, void **out, size_t bytes) {
            HB_HIP(c, hipMalloc(out, bytes ? bytes : 4));
            bufs.push_back(*out);
            return HB_OK;
        }
    } tmp;
    int rc = HB_OK;
    // worst-case slab: (n+1) rows x (2e+k+2) columns with e <= (n - k + 1) / 2
    const int emax = (n - (k - 1)) / 2;
    const size_t slab_words = (size_t)(n + 1) * (size_t)(2 * emax + k + 2) * NLr;
    // Inside the unique-decoding radius the reference's answer is the closest codeword's polynomial whatever the solver
    // (SURVEY appendix C): a codeword with at most floor((n' - k) / 2) errors is decoded by the Gao kernel (hb_gao.hip),
    // three orders of magnitude cheaper than an (n+1) x (2e+k+2) elimination, and only what it rejects -- where the
    // reference's particular solution, its descending-e' loop and its two failure messages matter -- goes through the
    // row reduction below.  A batch whose codewords all lost the SAME symbols -- the protocol's case: the erasures are the parties that have
    // not arrived, one set for the whole batch (reed_solomon.py:201-204) -- is a batch over the points that are left: Gao's kernels on the
    // reduced point set (the interpolant's launch reads the surviving columns in place), radius floor((n' - k) / 2).  Only per-codeword
    // erasure patterns take the row reduction throughout (round 4: any erasure did -- 72 k codewords/s where the kernels do millions).
    uint8_t *gao_ok = nullptr;
    int32_t *todo = nullptr;
    int64_t rejected = C;                            // codewords the row reduction still has to look at
    if (!env_hook(ENV_WB_NO_GAO) && n - k >= 1 && 2 * (k - 1) + 1 <= n) {
        int32_t erased = 0;
        HB_HIP(ctx, hipMemsetAsync(ctx->flag_dev, 0, 2 * sizeof(int32_t), s));
        const int64_t tot = C * n;
        k_wb_erasure_pattern<<<(unsigned)((tot + 255) / 256), 256, 0, s>>>(present_dev, tot, n, ctx->flag_dev);
        HB_LAUNCH_CHECK(ctx);
        // the verdict and -- in case there is a shared pattern -- the first codeword's presence bytes, behind ONE synchronisation
        std::vector<uint8_t> pat((size_t)n);
        HB_HIP(ctx, hipMemcpyAsync(&erased, ctx->flag_dev, sizeof erased, hipMemcpyDeviceToHost, s));
        HB_HIP(ctx, hipMemcpyAsync(pat.data(), present_dev, (size_t)n, hipMemcpyDeviceToHost, s));
        HB_HIP(ctx, hipStreamSynchronize(s));
        // the shared pattern, if there is one: the points that are left, in party order (as the reference enumerates them)
        std::vector<int32_t> sel;
        std::vector<uint64_t> xsel;
        int ns = n;
        if (erased == 1 && !env_hook(ENV_WB_NO_UNIFORM)) {
            for (int i = 0; i < n; i++)
                if (pat[i]) { sel.push_back(i); for (int q = 0; q < ctx->n_limbs; q++) xsel.push_back(x_host[(size_t)i * ctx->n_limbs + q]); }
            ns = (int)sel.size();
            if (ns - k < 1 || 2 * (k - 1) + 1 > ns) { sel.clear(); ns = n; erased = 3; }          // too few points left: the row reduction's refusals
            else erased = 0;
        }
        if ((erased & 2) && !env_hook(ENV_WB_NO_UNIFORM) && C >= WB_MIN_GROUP && C <= 0x7fffffffLL) {
            // Per-codeword patterns (reed_solomon_wb.py:129-151 takes any): the batch is cut by pattern.  A pattern shared by at least WB_MIN_GROUP
            // codewords is a batch over the points that are left -- gathered, decoded by Gao's kernels on the reduced point set, accepted within
            // the radius floor((n' - k) / 2) exactly as a shared pattern is; what is left (small groups, patterns with too few points, whatever
            // Gao's kernels reject) is the row reduction's work list.  Round 5: any second pattern sent the whole batch through the row reduction.
            const int W = (n + 63) / 64;
            unsigned long long *masks_dev = nullptr;
            rc = ctx_scratch(ctx, "wb.masks", (size_t)C * W * 8, (void **)&masks_dev); if (rc) return rc;
            k_wb_masks<<<(unsigned)((C * W + 255) / 256), 256, 0, s>>>(present_dev, C, n, W, masks_dev);
            HB_LAUNCH_CHECK(ctx);
            std::vector<unsigned long long> masks((size_t)C * W);
            HB_HIP(ctx, hipMemcpyAsync(masks.data(), masks_dev, masks.size() * 8, hipMemcpyDeviceToHost, s));
            HB_HIP(ctx, hipStreamSynchronize(s));
            // group ids by exact mask: an open-addressing table of (first codeword of the pattern, group)
            std::vector<std::vector<int32_t>> groups;
            {
                const size_t cap = 256;                  // > 2 x WB_MAX_PATTERNS slots
                std::vector<int32_t> slot_first(cap, -1), slot_group(cap, -1);
                bool too_many = false;
                for (int64_t c = 0; c < C && !too_many; c++) {
                    const unsigned long long *m = &masks[(size_t)c * W];
                    unsigned long long h = 0x9e3779b97f4a7c15ull;
                    for (int w = 0; w < W; w++) { h ^= m[w]; h *= 0xff51afd7ed558ccdull; h ^= h >> 29; }
                    size_t at = (size_t)h & (cap - 1);
                    for (;;) {
                        if (slot_first[at] < 0) {
                            if ((int)groups.size() >= WB_MAX_PATTERNS) { too_many = true; break; }
                            slot_first[at] = (int32_t)c; slot_group[at] = (int32_t)groups.size();
                            groups.emplace_back();
                            groups.back().push_back((int32_t)c);
                            break;
                        }
                        if (!memcmp(&masks[(size_t)slot_first[at] * W], m, (size_t)W * 8)) { groups[slot_group[at]].push_back((int32_t)c); break; }
                        at = (at + 1) & (cap - 1);
                    }
                }
                if (too_many) groups.clear();
            }
            if (!groups.empty()) {
                int32_t *errlen = nullptr, *gidx = nullptr;
                unsigned long long *gys = nullptr;
                uint32_t *gco = nullptr;
                size_t maxg = 0;
                for (auto &g : groups) maxg = std::max(maxg, g.size());
                const int row_q = n * ctx->n_limbs;                           // 64-bit words per codeword
                rc = ctx_scratch(ctx, "wb.gao_ok", (size_t)C, (void **)&gao_ok); if (rc) return rc;
                rc = ctx_scratch(ctx, "wb.errlen", (size_t)C * sizeof(int32_t), (void **)&errlen); if (rc) return rc;
                rc = ctx_scratch(ctx, "wb.todo", (size_t)C * sizeof(int32_t), (void **)&todo); if (rc) return rc;
                rc = ctx_scratch(ctx, "wb.gidx", maxg * sizeof(int32_t), (void **)&gidx); if (rc) return rc;
                rc = ctx_scratch(ctx, "wb.gys", maxg * (size_t)row_q * 8, (void **)&gys); if (rc) return rc;
                rc = ctx_scratch(ctx, "wb.gco", maxg * (size_t)k * ctx->n_limbs * 8, (void **)&gco); if (rc) return rc;
                // the row reduction's list starts with what no group decodes: small groups, patterns that leave too few points
                std::vector<int32_t> direct;
                std::vector<char> by_gao(groups.size(), 0);
                for (size_t gi = 0; gi < groups.size(); gi++) {
                    int ns_g = 0;
                    const unsigned long long *m = &masks[(size_t)groups[gi][0] * W];
                    for (int i = 0; i < n; i++) ns_g += (int)((m[i >> 6] >> (i & 63)) & 1);
                    if ((int64_t)groups[gi].size() >= WB_MIN_GROUP && ns_g - k >= 1 && 2 * (k - 1) + 1 <= ns_g) by_gao[gi] = 1;
                    else direct.insert(direct.end(), groups[gi].begin(), groups[gi].end());
                }
                const int32_t n_direct = (int32_t)direct.size();
                if (n_direct) HB_HIP(ctx, hipMemcpyAsync(todo, direct.data(), (size_t)n_direct * 4, hipMemcpyHostToDevice, s));
                HB_HIP(ctx, hipMemcpyAsync(ctx->flag_dev + 1, &n_direct, sizeof n_direct, hipMemcpyHostToDevice, s));
                HB_HIP(ctx, hipStreamSynchronize(s));                           // (the two sources are this stack's)
                for (size_t gi = 0; gi < groups.size(); gi++) {
                    if (!by_gao[gi]) continue;
                    const std::vector<int32_t> &g = groups[gi];
                    const int64_t Cg = (int64_t)g.size();
                    const unsigned long long *m = &masks[(size_t)g[0] * W];
                    std::vector<int32_t> gsel;
                    std::vector<uint64_t> gx;
                    for (int i = 0; i < n; i++)
                        if ((m[i >> 6] >> (i & 63)) & 1) { gsel.push_back(i); for (int q = 0; q < ctx->n_limbs; q++) gx.push_back(x_host[(size_t)i * ctx->n_limbs + q]); }
                    const int ns_g = (int)gsel.size();
                    HB_HIP(ctx, hipMemcpyAsync(gidx, g.data(), (size_t)Cg * 4, hipMemcpyHostToDevice, s));
                    HB_HIP(ctx, hipStreamSynchronize(s));
                    k_wb_gather<<<(unsigned)((Cg * row_q + 255) / 256), 256, 0, s>>>((const unsigned long long *)ys_dev, gidx, Cg, row_q, gys);
                    HB_LAUNCH_CHECK(ctx);
                    if (ns_g == n) rc = hb::gao_decode(ctx, x_host, n, k, (const uint64_t *)gys, Cg, (uint64_t *)gco, nullptr, errlen, gao_ok, stream);
                    else rc = hb::gao_decode(ctx, gx.data(), ns_g, k, (const uint64_t *)gys, Cg, (uint64_t *)gco, nullptr, errlen, gao_ok, stream, n, gsel.data());
                    if (rc) return rc;
                    if (ctx->n_limbs == 4) k_wb_take_gao_group<8><<<(unsigned)((Cg + 255) / 256), 256, 0, s>>>(gao_ok, errlen, (ns_g - k) / 2, gco, k, Cg, gidx, (uint32_t *)coeffs_dev, coeff_len_dev, status_dev, ctx->flag_dev + 1, todo);
                    else k_wb_take_gao_group<2><<<(unsigned)((Cg + 255) / 256), 256, 0, s>>>(gao_ok, errlen, (ns_g - k) / 2, gco, k, Cg, gidx, (uint32_t *)coeffs_dev, coeff_len_dev, status_dev, ctx->flag_dev + 1, todo);
                    HB_LAUNCH_CHECK(ctx);
                    HB_HIP(ctx, hipStreamSynchronize(s));                       // (gidx, gys, gco are the next group's too)
                }
                int32_t rej = 0;
                HB_HIP(ctx, hipMemcpyAsync(&rej, ctx->flag_dev + 1, sizeof rej, hipMemcpyDeviceToHost, s));
                HB_HIP(ctx, hipStreamSynchronize(s));
                rejected = rej;
                erased = -1;                               // settled here: not the shared-pattern branch below
            }
        }
        if (!erased) {
            int32_t *errlen = nullptr;
            // (context scratch, reused by the next call -- ctx_scratch, hb_common.hpp)
            rc = ctx_scratch(ctx, "wb.gao_ok", (size_t)C, (void **)&gao_ok); if (rc) return rc;
            rc = ctx_scratch(ctx, "wb.errlen", (size_t)C * sizeof(int32_t), (void **)&errlen); if (rc) return rc;
            rc = ctx_scratch(ctx, "wb.todo", (size_t)C * sizeof(int32_t), (void **)&todo); if (rc) return rc;
            if (sel.empty()) rc = hb::gao_decode(ctx, x_host, n, k, ys_dev, C, coeffs_dev, nullptr, errlen, gao_ok, stream);      // (the locators' lengths, not the locators)
            else rc = hb::gao_decode(ctx, xsel.data(), ns, k, ys_dev, C, coeffs_dev, nullptr, errlen, gao_ok, stream, n, sel.data());
            if (rc) return rc;
            if (ctx->n_limbs == 4) k_wb_take_gao<8><<<(unsigned)((C + 255) / 256), 256, 0, s>>>(gao_ok, errlen, (ns - k) / 2, (const uint32_t *)coeffs_dev, k, C, coeff_len_dev, status_dev, ctx->flag_dev + 1, todo);
            else k_wb_take_gao<2><<<(unsigned)((C + 255) / 256), 256, 0, s>>>(gao_ok, errlen, (ns - k) / 2, (const uint32_t *)coeffs_dev, k, C, coeff_len_dev, status_dev, ctx->flag_dev + 1, todo);
            HB_LAUNCH_CHECK(ctx);
            int32_t rej = 0;
            HB_HIP(ctx, hipMemcpyAsync(&rej, ctx->flag_dev + 1, sizeof rej, hipMemcpyDeviceToHost, s));
            HB_HIP(ctx, hipStreamSynchronize(s));
            rejected = rej;
        }
    }
    if (rejected > 0) {
        // the slab-resident row reduction: one block per codeword in turn; no more blocks (and slabs) than codewords left for it
        int64_t blocks = rejected < 1024 ? rejected : 1024;
        // the points in Montgomery form: only the row reduction reads them (a batch Gao's kernels settle entirely pays for no
        // upload, no allocation and no synchronous hipFree)
        uint32_t *xd = nullptr, *xm = nullptr;
        rc = upload_elems(ctx, x_host, (size_t)n, &xd, s); if (rc) return rc;
        tmp.bufs.push_back(xd);
        rc = tmp.alloc(ctx, (void **)&xm, (size_t)n * NLr * 4); if (rc) return rc;
        HB_DISPATCH(ctx,
            (k_points_to_mont<9, 8><<<(n + 63) / 64, 64, 0, s>>>(ctx->pw, xd, n, xm)),
            (k_points_to_mont<3, 2><<<(n + 63) / 64, 64, 0, s>>>(ctx->pn, xd, n, xm)));
        HB_LAUNCH_CHECK(ctx);
        uint32_t *scratch = nullptr;
        rc = tmp.alloc(ctx, (void **)&scratch, slab_words * 4 * (size_t)blocks); if (rc) return rc;
        HB_DISPATCH(ctx,
            (k_wb<9, 8><<<(unsigned)blocks, WB_THREADS, 0, s>>>(ctx->pw, xm, (const uint32_t *)ys_dev, present_dev, n, k, rejected, scratch, slab_words,
                                                              (uint32_t *)coeffs_dev, coeff_len_dev, status_dev, todo)),
            (k_wb<3, 2><<<(unsigned)blocks, WB_THREADS, 0, s>>>(ctx->pn, xm, (const uint32_t *)ys_dev, present_dev, n, k, rejected, scratch, slab_words,
                                                              (uint32_t *)coeffs_dev, coeff_len_dev, status_dev, todo)));
        HB_LAUNCH_CHECK(ctx);
    }
    // (the row reduction's temporaries go with this scope: the stream must have let go of them.  A batch Gao's kernels settled entirely has
    // none, and the call returns with its last launch enqueued, like every other entry point)
    if (!tmp.bufs.empty()) HB_HIP(ctx, hipStreamSynchronize(s));
    return HB_OK;
}
