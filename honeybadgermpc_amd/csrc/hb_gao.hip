// hb_gao.hip -- Gao's Reed-Solomon decoder, batched: one wavefront per codeword.
//
// Reference functions replaced (paths under /root/reference):
//   gao_interpolate / gao_interpolate_fft     honeybadgermpc/ntl/rsdecode_impl.h:325-405
//   partial_gcd                               honeybadgermpc/ntl/rsdecode_impl.h:281-323
//   gao_interpolate (python boundary)         honeybadgermpc/ntl/hbmpc_ntl_helpers.pyx:389-439
// The reference decodes one codeword per call (reed_solomon.py:335-338); here C codewords
// that share the evaluation points are decoded by one launch.
//
//   g0 = prod (X - x_i)                  computed once per point set (k_poly_from_roots)
//   g1 = interpolant of the codeword     = V(x)^-1 * y, one lazily-reduced mat-vec for the
//                                          whole batch (the FFT variant of the reference,
//                                          rsdecode_impl.h:376, yields the same g1)
//   (r, v) = partial_gcd(g0, g1, (n+k)/2), f = r / v must be exact with deg f < k.
//
// The extended Euclid loop is run FRACTION-FREE: a pseudo-division step
//     r0 <- lc(r1) * r0 - r0[top] * X^j * r1        (and the same combination on t0)
// needs no field inversion; after the deg(r0)-deg(r1)+1 steps of one division the pair
// (r0, t0) equals c * (r2, t2) of the reference's true division for the scalar
// c = lc(r1)^(delta+1) * c0, which is tracked.  At the end ONE inversion (of c * lc(v),
// Montgomery's trick) recovers both the reference's un-normalised cofactor v = V / c and the
// monic divisor for the exact division.  Per codeword: O(n * e / 64) mulmods per lane plus
// one Fermat inversion, instead of one inversion per Euclid step.
#include <algorithm>

#include "hb_common.hpp"

using namespace hb;

namespace {

// A(X) = prod_{j<k} (X - x_j), Montgomery digits [k+1][NL].  One block.
template <int NL, int NW>
__global__ void __launch_bounds__(1024) k_poly_from_roots(const FpParams<NL> P, const uint32_t *__restrict__ x, int k, uint32_t *__restrict__ A) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t *xs = smem;
    uint32_t *A0 = xs + (size_t)k * NL;
    uint32_t *A1 = A0 + (size_t)(k + 1) * NL;
    const int t = threadIdx.x;
    if (t < k) {
        uint32_t xd[NL], xm[NL];
        load_digits<NL, NW>(xd, x + (size_t)t * NW);
        to_mont(xm, xd, P);
#pragma unroll
        for (int q = 0; q < NL; q++) xs[t * NL + q] = xm[q];
    }
    if (t <= k) {
#pragma unroll
        for (int q = 0; q < NL; q++) A0[t * NL + q] = (t == 0) ? P.one[q] : 0u;
    }
    __syncthreads();
    uint32_t *cur = A0, *nxt = A1;
    for (int j = 0; j < k; j++) {
        if (t <= k) {
            uint32_t a[NL], am1[NL], xv[NL], prod[NL], r[NL];
#pragma unroll
            for (int q = 0; q < NL; q++) { a[q] = cur[t * NL + q]; am1[q] = (t > 0) ? cur[(t - 1) * NL + q] : 0u; xv[q] = xs[j * NL + q]; }
            mont_mul(prod, xv, a, P);
            fp_sub(r, am1, prod, P);
#pragma unroll
            for (int q = 0; q < NL; q++) nxt[t * NL + q] = r[q];
        }
        __syncthreads();
        uint32_t *tmp = cur; cur = nxt; nxt = tmp;
    }
    if (t <= k) {
#pragma unroll
        for (int q = 0; q < NL; q++) A[(size_t)t * NL + q] = cur[t * NL + q];
    }
}

template <int NL> __device__ __forceinline__ void lds_get(uint32_t (&d)[NL], const uint32_t *p) {
#pragma unroll
    for (int q = 0; q < NL; q++) d[q] = p[q];
}
template <int NL> __device__ __forceinline__ void lds_put(uint32_t *p, const uint32_t (&d)[NL]) {
#pragma unroll
    for (int q = 0; q < NL; q++) p[q] = d[q];
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return v;
}
// exact degree of the polynomial in LDS (coefficients 0..hi), -1 for zero
template <int NL> __device__ int poly_degree(const uint32_t *p, int hi, int lane) {
    int best = -1;
    for (int idx = lane; idx <= hi; idx += 64) {
        uint32_t o = 0;
#pragma unroll
        for (int q = 0; q < NL; q++) o |= p[(size_t)idx * NL + q];
        if (o) best = idx;
    }
    return wave_max(best);
}

// K = p 2^j, the multiple of p with 29 (NL - 1) + 27 bits, in REDUNDANT digits: every digit but the top one at least 2^29 - 1 (a unit
// borrowed from the next digit), the top one at least 2^26 - 1.  K - x, digit by digit, is then a non-negative representation of
// -x mod p for any lazy residue x (digits below 2^29, top digit below 2^25): what a round subtracts it multiplies by such a
// negation, so every sum of products is a sum of non-negative terms -- unsigned columns starting from zero, no offset, no signed
// carries.  (Round 4's first pointer-driven version added 2 p^2 to signed columns: 36 SGPRs of zero-extended constants, and a middle
// column's worst case, 35 products of 2^58, did not fit 63 bits.)
template <int NL> struct GaoConsts { uint32_t kd[NL]; };

// ONE round of the Euclid loop / the division for one lane: r = (m0 u + m1 w1 + m2 w0) / R mod p, every operand read from LDS through
// a pointer of the lane's own (a lane without a term points at the zero element).  No conditional subtraction: the kernel's
// residues are LAZY -- with operands below 1.5 p (and K < R / 4) the sum stays below 6.8 p^2 + 1.5 p K < p R / 2, so REDC returns a
// value below 1.45 p again (R = 2^(29 NL) >= 32 p), in normalised digits (a multiple of p comes out as 0 or as p itself:
// is_zero_lazy).  Columns: a negation's digits are below 2^30, so a column holds at most 9 (2^59 + 2^58 + 2^58) + REDC's 9 2^58 < 2^64.
template <int NL> __device__ __forceinline__ void gao_round(uint32_t (&r)[NL], const uint32_t *pm0, const uint32_t *pu, const uint32_t *pm1, const uint32_t *pw1,
                                                            const uint32_t *pm2, const uint32_t *pw0, const FpParams<NL> &P) {
    uint64_t col[2 * NL];
    col_zero(col);
    { uint32_t m[NL], v[NL]; lds_get<NL>(m, pm0); lds_get<NL>(v, pu); mac<NL>(col, m, v); }
    { uint32_t m[NL], v[NL]; lds_get<NL>(m, pm1); lds_get<NL>(v, pw1); mac<NL>(col, m, v); }
    { uint32_t m[NL], v[NL]; lds_get<NL>(m, pm2); lds_get<NL>(v, pw0); mac<NL>(col, m, v); }
    redc(r, col, P);
}
template <int NL> __device__ __forceinline__ void gao_round2(uint32_t (&r)[NL], const uint32_t *pm0, const uint32_t *pu, const uint32_t *pm1, const uint32_t *pw1,
                                                             const FpParams<NL> &P) {
    uint64_t col[2 * NL];
    col_zero(col);
    { uint32_t m[NL], v[NL]; lds_get<NL>(m, pm0); lds_get<NL>(v, pu); mac<NL>(col, m, v); }
    { uint32_t m[NL], v[NL]; lds_get<NL>(m, pm1); lds_get<NL>(v, pw1); mac<NL>(col, m, v); }
    redc(r, col, P);
}
// a lazy residue is zero when its digits are all zero or are p's
template <int NL> __device__ __forceinline__ bool is_zero_lazy(const uint32_t (&a)[NL], const FpParams<NL> &P) {
    uint32_t o = 0, e = 0;
#pragma unroll
    for (int q = 0; q < NL; q++) { o |= a[q]; e |= a[q] ^ P.p[q]; }
    return o == 0 || e == 0;
}
// exact degree of the polynomial of lazy residues in LDS (coefficients 0..hi), -1 for zero
template <int NL> __device__ int poly_degree_lazy(const uint32_t *p, int hi, int lane, const FpParams<NL> &P) {
    int best = -1;
    for (int idx = lane; idx <= hi; idx += 64) {
        uint32_t a[NL];
        lds_get<NL>(a, p + (size_t)idx * NL);
        if (!is_zero_lazy<NL>(a, P)) best = idx;
    }
    return wave_max(best);
}

// One wave per codeword (three to a SIMD).  The kernel stops short of the ONE field inversion a codeword needs (Fermat: 255 squarings + ~128
// multiplications, every lane computing the same thing -- it cost as much as everything else here together, 47 of 102 ms at config 4):
// the division f = r / v runs as a PSEUDO-division by the un-normalised cofactor V (r <- l r - c_i x^i V, l = lc(V): the true quotient
// digit is q_i = c_i / l^(dq - i + 1)), the raw c_i and V leave in Montgomery form, packed, in the output buffers, with cs and l in
// a side record -- and k_gao_finish, one LANE per group of codewords, inverts w = cs l and scales the outputs in place.
//
// Everything a round multiplies lives in LDS and is reached through per-lane pointers: the step's three multipliers (S), the scale
// factors c0 / c1 (CA / CB), the jobs' operands.  A lane's role in a round is nothing but its seven pointers -- no selects between
// register files, no negations in the round (round 4's first version of the fused step spent ~150 of a round's ~600 vector
// instructions on them) -- and residues stay lazy (gao_round), so a round is 243 multiply-adds + REDC and little else.
#ifndef GAO_WAVES_PER_EU
#define GAO_WAVES_PER_EU 3
#endif
// one codeword's state: its arrays in LDS and the (wave-uniform) degrees
template <int NL> struct GaoCw {
    uint32_t *R0, *R1, *T0, *T1;                         // this codeword's arrays in LDS (the pairs swap every step)
    uint32_t *M;                                         // ... and its scalars: three multipliers, -X, the two scale factors
    int csw;                                             // which of the two scale factors is c0 at the moment
    __device__ __forceinline__ uint32_t *S() const { return M; }
    __device__ __forceinline__ uint32_t *NEGX() const { return M + 3 * NL; }
    __device__ __forceinline__ uint32_t *CA() const { return M + (4 + csw) * NL; }
    __device__ __forceinline__ uint32_t *CB() const { return M + (5 - csw) * NL; }
    int dR0, dR1, dT0, dT1;                              // wave-uniform degrees
    bool have_sc, done;
    uint32_t *rp, *vp, *csp;                             // what the loop ends with: remainder, cofactor, the scale factor that goes with them
    int dr, dvb;
    uint32_t *F;                                         // the division: raw quotient digits, degrees, the round counter
    int dv, dq, df, i;
    bool ok, divided;
};

// words of one codeword's arrays: R0, R1, T0, T1, three multipliers, -X, c0, c1
template <int NL> __device__ __forceinline__ int gao_cw_words(int len, int lenT) { return (2 * len + 2 * lenT + 6) * NL; }

template <int NL> __device__ __forceinline__ void gao_loop_ends(GaoCw<NL> &w, bool first) {
    if (first) { w.rp = w.R0; w.vp = w.T0; w.dr = w.dR0; w.dvb = w.dT0; w.csp = w.CA(); }
    else { w.rp = w.R1; w.vp = w.T1; w.dr = w.dR1; w.dvb = w.dT1; w.csp = w.CB(); }
    w.done = true;
}

// arrays of one codeword: g0 -> R0, the interpolant (packed, chunk-major) -> R1 in Montgomery form, T0 = 0, T1 = 1, c0 = c1 = 1
template <int NL, int NW>
__device__ __forceinline__ void gao_cw_init(GaoCw<NL> &w, uint32_t *base, int len, int lenT, const uint32_t *__restrict__ g0, const uint32_t *__restrict__ g1row,
                                            int npts, int k, int lane, const FpParams<NL> &P) {
    w.R0 = base; w.R1 = w.R0 + (size_t)len * NL; w.T0 = w.R1 + (size_t)len * NL; w.T1 = w.T0 + (size_t)lenT * NL;
    w.M = w.T1 + (size_t)lenT * NL; w.csw = 0;
    if (lane < NL) { w.CA()[lane] = P.one[lane]; w.CB()[lane] = P.one[lane]; }
    for (int idx = lane; idx < len; idx += 64) {
        uint32_t a[NL], z[NL], m[NL];
#pragma unroll
        for (int q = 0; q < NL; q++) { a[q] = g0[(size_t)idx * NL + q]; z[q] = 0; }
        lds_put<NL>(w.R0 + (size_t)idx * NL, a);
        if (idx < lenT) {
            lds_put<NL>(w.T0 + (size_t)idx * NL, z);
            if (idx == 0) lds_put<NL>(w.T1, P.one); else lds_put<NL>(w.T1 + (size_t)idx * NL, z);
        }
        if (idx < npts) {
            uint32_t yd[NL];
            load_digits<NL, NW>(yd, g1row + (size_t)idx * NW);
            to_mont(m, yd, P);
            lds_put<NL>(w.R1 + (size_t)idx * NL, m);
        } else lds_put<NL>(w.R1 + (size_t)idx * NL, z);
    }
    __syncthreads();
    w.dR0 = npts; w.dR1 = __builtin_amdgcn_readfirstlane(poly_degree<NL>(w.R1, npts - 1, lane)); w.dT0 = -1; w.dT1 = 0;
    w.have_sc = false; w.done = false;
    const int D = (npts + k) / 2;
    if (w.dR0 < D) gao_loop_ends<NL>(w, true);              // rsdecode_impl.h:289-294 (cannot fire: deg g0 = n >= D)
    else if (w.dR1 < D) gao_loop_ends<NL>(w, false);        // rsdecode_impl.h:296-301
}

// one Euclid step of one codeword: the fused step in its two rounds, or the generic pseudo-division
template <int NL> __device__ __forceinline__ void gao_step_single(GaoCw<NL> &w, int lane, const uint32_t *ZERO, const uint32_t *KD, const FpParams<NL> &P, const GaoConsts<NL> &GK) {
    const int delta = w.dR0 - w.dR1;
    const int dR1 = w.dR1;
    // The generic division step's common case (the degrees drop one at a time), fused: its two pseudo-division sub-steps
    //     r' = L r0 - a1 X r1,   a0 = r'[deg r1],   r'' = L r' - a0 r1
    // are ONE update r''[i] = L^2 r0[i] - (L a1) r1[i-1] - a0 r1[i] (the same on the cofactor; c0 <- L^2 c0): three products and one
    // reduction per coefficient where the sub-steps take four and two.  The three multipliers of a step cost four modular
    // multiplications -- executed by the whole wave they would eat the gain (round 4 measured it: 45 ms against 38) -- so those
    // of the NEXT step are computed by three otherwise idle lanes of this step's second round: the first round takes the TOP 64
    // coefficients of the remainder, so the two leading coefficients of r'' the next multipliers need are there when the second
    // round (the rest of the remainder, the cofactor, c0, the three jobs) starts.  With (X, Y, Z, W) = (lc r'', L, r1[deg r1 - 1],
    // r''[deg r'' - 1]): job0 = X X, job1 = -X Y, job2 = Y W - X Z -- the residues of the NEGATED multipliers, so that the
    // update is a plain sum S0 u + S1 w1 + S2 w0; the jobs subtract through NEGX, the digit-wise negation of X that the lane which
    // computed X left in LDS (K - X in K's redundant digits: GaoConsts).  Values equal to the sub-steps' mod p, term by term.
    const int ttop_f = max(w.dT0, w.dT1 + 1), n2_f = max(0, dR1 - 64);
    if (delta == 1 && dR1 >= 2 && n2_f + ttop_f + 1 <= 60) {
        const int top = dR1, ttop = ttop_f, n2 = n2_f;
        const int jb = lane - 60;
        const bool isJ = lane >= 60 && lane < 63;
        if (!w.have_sc) {
            // the multipliers of THIS step alone (first step, or after a degree anomaly): (X, Y, Z, W) = (L, lc r0, r0[deg r1], r1[deg r1 - 1])
            __syncthreads();
            if (lane < NL) w.NEGX()[lane] = KD[lane] - w.R1[(size_t)dR1 * NL + lane];
            __syncthreads();
            const uint32_t *Xa = w.R1 + (size_t)dR1 * NL, *Ya = w.R0 + (size_t)(dR1 + 1) * NL, *Za = w.R0 + (size_t)dR1 * NL, *Wa = w.R1 + (size_t)(dR1 - 1) * NL;
            const uint32_t *pm0 = ZERO, *pu = ZERO, *pm1 = ZERO, *pw1 = ZERO;
            if (isJ) { pm0 = jb == 0 ? Xa : w.NEGX(); pu = jb == 0 ? Xa : (jb == 1 ? Ya : Za); if (jb == 2) { pm1 = Ya; pw1 = Wa; } }
            uint32_t r[NL];
            gao_round2<NL>(r, pm0, pu, pm1, pw1, P);
            if (isJ) lds_put<NL>(w.S() + (size_t)jb * NL, r);
            __syncthreads();
        }
        {
            const bool act = lane < min(64, top);
            const int idx = top - 1 - lane;
            const uint32_t *pu = ZERO, *pw1 = ZERO, *pw0 = ZERO;
            if (act) { pu = w.R0 + (size_t)idx * NL; pw1 = idx >= 1 ? w.R1 + (size_t)(idx - 1) * NL : ZERO; pw0 = w.R1 + (size_t)idx * NL; }
            uint32_t r[NL];
            gao_round<NL>(r, w.S(), pu, w.S() + NL, pw1, w.S() + 2 * NL, pw0, P);
            __syncthreads();
            if (act) lds_put<NL>(w.R0 + (size_t)idx * NL, r);
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < NL; q++) w.NEGX()[q] = GK.kd[q] - r[q];
            }
            __syncthreads();
        }
        {
            const uint32_t *Xa = w.R0 + (size_t)(top - 1) * NL, *Ya = w.R1 + (size_t)dR1 * NL, *Za = w.R1 + (size_t)(dR1 - 1) * NL, *Wa = w.R0 + (size_t)(top - 2) * NL;
            const uint32_t *pm0 = w.S(), *pu = ZERO, *pm1 = ZERO, *pw1 = ZERO, *pm2 = ZERO, *pw0 = ZERO;
            uint32_t *dst = nullptr;
            if (lane < n2) {
                const int idx = n2 - 1 - lane;
                pu = w.R0 + (size_t)idx * NL; pw1 = idx >= 1 ? w.R1 + (size_t)(idx - 1) * NL : ZERO; pw0 = w.R1 + (size_t)idx * NL;
                pm1 = w.S() + NL; pm2 = w.S() + 2 * NL; dst = w.R0 + (size_t)idx * NL;
            } else if (lane <= n2 + ttop) {
                const int idx = lane - n2;
                pu = w.T0 + (size_t)idx * NL;
                pw1 = (idx >= 1 && idx - 1 <= w.dT1) ? w.T1 + (size_t)(idx - 1) * NL : ZERO;
                pw0 = idx <= w.dT1 ? w.T1 + (size_t)idx * NL : ZERO;
                pm1 = w.S() + NL; pm2 = w.S() + 2 * NL; dst = w.T0 + (size_t)idx * NL;
            } else if (isJ) {
                pm0 = jb == 0 ? Xa : w.NEGX(); pu = jb == 0 ? Xa : (jb == 1 ? Ya : Za);
                if (jb == 2) { pm1 = Ya; pw1 = Wa; }
                dst = w.S() + (size_t)jb * NL;
            } else if (lane == 63) { pu = w.CA(); dst = w.CA(); }
            uint32_t r[NL];
            gao_round<NL>(r, pm0, pu, pm1, pw1, pm2, pw0, P);
            __syncthreads();
            if (dst) lds_put<NL>(dst, r);
            if (lane < 2) {
#pragma unroll
                for (int q = 0; q < NL; q++) w.R0[(size_t)(top + lane) * NL + q] = 0;
            }
            __syncthreads();
        }
        w.dT0 = ttop;
        w.have_sc = true;
        return;
    }
    // The generic division step (degrees drop one at a time): delta + 1 pseudo-division sub-steps r0 <- L r0 - r0[top] X^j r1, the same on
    // the cofactor, c0 <- L c0 each
    w.have_sc = false;
    uint32_t L[NL], c0[NL];
    lds_get<NL>(L, w.R1 + (size_t)dR1 * NL);
    lds_get<NL>(c0, w.CA());
    for (int j = delta; j >= 0; j--) {
        uint32_t a[NL], an[NL];
        lds_get<NL>(a, w.R0 + (size_t)(dR1 + j) * NL);
        cond_sub_p(a, P);                 // (lazy residues: canonical before the negation)
        fp_neg(an, a, P);                 // L u - a w = L u + (p - a) w: ONE reduction for the two products (columns stay below
                                          // 3 NL 2^58 < 2^63; the sum is < 3 p^2 < p R / 8, so REDC leaves < 2p: one conditional subtraction)
        __syncthreads();
        const int top = dR1 + j;
        const int ttop = max(w.dT0, w.dT1 + j);
        // One update, r <- L u + (p - a) w, for every coefficient of the remainder R0 below `top` (w = R1[idx - j]) and of the cofactor
        // T0 up to `ttop` (w = T1[idx - j]), and c0 <- L c0.  deg R0 + deg T0 stays about npts, so what the remainder leaves of its
        // last round of 64 lanes holds the whole cofactor AND the scale factor: two rounds a step at n = 100 where three and a
        // wave-wide multiplication were.
        int base = 0;
        for (; base + 64 <= top; base += 64) {
            const int idx = base + lane;
            uint32_t u[NL], r[NL];
            uint64_t col[2 * NL];
            lds_get<NL>(u, w.R0 + (size_t)idx * NL);
            col_zero(col);
            mac<NL>(col, L, u);
            if (idx >= j) {
                uint32_t ww[NL];
                lds_get<NL>(ww, w.R1 + (size_t)(idx - j) * NL);
                mac<NL>(col, an, ww);
            }
            redc(r, col, P);
            cond_sub_p(r, P);
            lds_put<NL>(w.R0 + (size_t)idx * NL, r);
        }
        const int n2 = top - base;                       // lanes the remainder still needs: 0 .. 63
        const bool merged = n2 + ttop + 1 <= 63;         // ... and the cofactor beside them, lane 63 for c0
        {
            const bool isR = lane < n2, isT = merged && lane >= n2 && lane <= n2 + ttop, cz = merged && lane == 63;
            uint32_t hand[NL];
#pragma unroll
            for (int q = 0; q < NL; q++) hand[q] = 0;
            if (isR || isT || cz) {
                uint32_t *A = isR ? w.R0 : w.T0;
                const uint32_t *B = isR ? w.R1 : w.T1;
                const int idx = isR ? base + lane : lane - n2;
                const bool second = isR ? idx >= j : (isT && idx >= j && idx - j <= w.dT1);
                uint32_t u[NL], r[NL];
                uint64_t col[2 * NL];
                if (cz) fp_set(u, c0); else lds_get<NL>(u, A + (size_t)idx * NL);
                col_zero(col);
                mac<NL>(col, L, u);
                if (second) {
                    uint32_t ww[NL];
                    lds_get<NL>(ww, B + (size_t)(idx - j) * NL);
                    mac<NL>(col, an, ww);
                }
                redc(r, col, P);
                cond_sub_p(r, P);
                if (!cz) lds_put<NL>(A + (size_t)idx * NL, r);
#pragma unroll
                for (int q = 0; q < NL; q++) hand[q] = r[q];
            }
            if (merged) {
#pragma unroll
                for (int q = 0; q < NL; q++) c0[q] = (uint32_t)__builtin_amdgcn_readlane((int)hand[q], 63);
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < NL; q++) w.R0[(size_t)top * NL + q] = 0;
        }
        if (!merged) {
            for (int idx = lane; idx <= ttop; idx += 64) {
                uint32_t u[NL], r[NL];
                uint64_t col[2 * NL];
                lds_get<NL>(u, w.T0 + (size_t)idx * NL);
                col_zero(col);
                mac<NL>(col, L, u);
                if (idx >= j && idx - j <= w.dT1) {
                    uint32_t ww[NL];
                    lds_get<NL>(ww, w.T1 + (size_t)(idx - j) * NL);
                    mac<NL>(col, an, ww);
                }
                redc(r, col, P);
                cond_sub_p(r, P);
                lds_put<NL>(w.T0 + (size_t)idx * NL, r);
            }
            mont_mul(c0, c0, L, P);           // c0 <- lc(r1)^(delta+1) * c0, one factor per step
        }
        w.dT0 = ttop;
        __syncthreads();
    }
    if (lane == 0) lds_put<NL>(w.CA(), c0);
    __syncthreads();
}

// after a step: the new remainder's degree (it drops by exactly one as a rule), the loop's end, (r0, r1) <- (r1, r2)
template <int NL> __device__ __forceinline__ void gao_after_step(GaoCw<NL> &w, int D, int lane, const FpParams<NL> &P) {
    uint32_t topc[NL];
    bool nz = false;
    if (w.dR1 >= 1) { lds_get<NL>(topc, w.R0 + (size_t)(w.dR1 - 1) * NL); nz = !is_zero_lazy<NL>(topc, P); }
    w.dR0 = __builtin_amdgcn_readfirstlane(nz ? w.dR1 - 1 : poly_degree_lazy<NL>(w.R0, w.dR1 - 1, lane, P));
    if (w.dR0 != w.dR1 - 1) w.have_sc = false;
    if (w.dR0 < D) { gao_loop_ends<NL>(w, true); return; }
    uint32_t *tp = w.R0; w.R0 = w.R1; w.R1 = tp;
    tp = w.T0; w.T0 = w.T1; w.T1 = tp;
    w.csw ^= 1;
    int ti = w.dR0; w.dR0 = w.dR1; w.dR1 = ti;
    ti = w.dT0; w.dT0 = w.dT1; w.dT1 = ti;
}

// the division f = r / v up to its rounds: the cofactor leaves as it is, lc(V) takes the place of the loop's multipliers
template <int NL, int NW>
__device__ __forceinline__ void gao_division_begins(GaoCw<NL> &w, int64_t c, int npts, uint32_t *__restrict__ errloc, int lane, const FpParams<NL> &P) {
    w.dv = __builtin_amdgcn_readfirstlane(poly_degree_lazy<NL>(w.vp, w.dvb, lane, P));
    w.ok = w.dv >= 0;
    w.F = (w.rp == w.R0) ? w.R1 : w.R0;
    w.df = -1; w.dq = -1; w.i = -1; w.divided = false;
    if (!w.ok) return;
    for (int idx = lane; errloc && idx <= w.dv; idx += 64) {
        uint32_t u[NL];
        lds_get<NL>(u, w.vp + (size_t)idx * NL);
        cond_sub_p(u, P);
        store_digits<NL, NW>(errloc + ((size_t)c * (npts + 1) + idx) * NW, u);
    }
    if (w.dr < 0) return;
    if (w.dr < w.dv) { w.ok = false; return; }       // non-zero remainder
    w.dq = w.dr - w.dv; w.i = w.dq; w.divided = true;
    __syncthreads();
    if (lane < NL) w.S()[lane] = w.vp[(size_t)w.dv * NL + lane];
}

// one round of one codeword's pseudo-division: r <- l r - c_i x^i V below the leading term
template <int NL> __device__ __forceinline__ void gao_division_round(GaoCw<NL> &w, int lane, const uint32_t *ZERO, const uint32_t *KD, const FpParams<NL> &P) {
    uint32_t *LV = w.S(), *CN = w.S() + NL;
    const int i = w.i, dv = w.dv;
    if (lane < NL) { const uint32_t cq = w.rp[(size_t)(i + dv) * NL + lane]; w.F[(size_t)i * NL + lane] = cq; CN[lane] = KD[lane] - cq; }
    __syncthreads();
    for (int base = 0; base < i + dv; base += 64) {
        const int idx = base + lane;
        const bool act = idx < i + dv;
        const uint32_t *pu = act ? w.rp + (size_t)idx * NL : ZERO;
        const uint32_t *pw = (act && idx >= i) ? w.vp + (size_t)(idx - i) * NL : ZERO;
        uint32_t r[NL];
        gao_round2<NL>(r, LV, pu, CN, pw, P);
        if (act) lds_put<NL>(w.rp + (size_t)idx * NL, r);
    }
    __syncthreads();
    w.i = i - 1;
}

template <int NL, int NW>
__device__ __forceinline__ void gao_cw_leaves(GaoCw<NL> &w, int64_t c, int k, uint32_t *__restrict__ coeffs, int32_t *__restrict__ errlen, uint8_t *__restrict__ okflag,
                                              uint32_t *__restrict__ side, int lane, const FpParams<NL> &P) {
    if (w.divided) {
        if (poly_degree_lazy<NL>(w.rp, w.dv - 1, lane, P) >= 0) w.ok = false;      // remainder must vanish (l != 0: scaled or not)
        w.df = poly_degree_lazy<NL>(w.F, w.dq, lane, P);                            // c_i = l^(dq - i + 1) q_i: zero exactly where q_i is
        if (w.df >= k) w.ok = false;
    }
    if (w.ok) {
        for (int i = lane; i < k; i += 64) {
            uint32_t o[NL];
            if (i <= w.df) { lds_get<NL>(o, w.F + (size_t)i * NL); cond_sub_p(o, P); }
            else {
#pragma unroll
                for (int q = 0; q < NL; q++) o[q] = 0;
            }
            store_digits<NL, NW>(coeffs + ((size_t)c * k + i) * NW, o);
        }
        if (lane == 0) {
            uint32_t cs[NL], lcv[NL];
            lds_get<NL>(cs, w.csp);
            cond_sub_p(cs, P);
            if (w.dv >= 0) lds_get<NL>(lcv, w.vp + (size_t)w.dv * NL); else fp_set(lcv, P.one);
            cond_sub_p(lcv, P);
            store_digits<NL, NW>(side + (size_t)c * (2 * NW + 4), cs);
            store_digits<NL, NW>(side + (size_t)c * (2 * NW + 4) + NW, lcv);
            side[(size_t)c * (2 * NW + 4) + 2 * NW] = (uint32_t)w.dq;
            side[(size_t)c * (2 * NW + 4) + 2 * NW + 1] = (uint32_t)w.df;
        }
    }
    if (lane == 0) { okflag[c] = w.ok ? 1 : 0; errlen[c] = w.ok ? w.dv + 1 : 0; }
}

template <int NL, int NW>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GAO_WAVES_PER_EU))) k_gao(const FpParams<NL> P, const uint32_t *__restrict__ g0, const uint32_t *__restrict__ g1buf,
                                            int npts, int k, int64_t C, uint32_t *__restrict__ coeffs, uint32_t *__restrict__ errloc,
                                            int32_t *__restrict__ errlen, uint8_t *__restrict__ okflag, uint32_t *__restrict__ side, const GaoConsts<NL> GK) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x;
    const int64_t c = blockIdx.x;
    // deg t_{i+1} + deg r_i = npts throughout the Euclid loop and it runs while deg r >= D: the cofactors never exceed degree
    // npts - D, so their arrays are short -- 10 KB of LDS per codeword instead of 14.5 at n = 100: 16 resident waves per CU, not 11
    const int len = npts + 1, lenT = npts - (npts + k) / 2 + 3;
    uint32_t *ZERO = smem + (size_t)gao_cw_words<NL>(len, lenT), *KD = ZERO + NL;      // a zero element; K's redundant digits, for the lanes that negate element-wise
    if (lane < NL) {
        uint32_t kq = 0;
#pragma unroll
        for (int q = 0; q < NL; q++) kq = lane == q ? GK.kd[q] : kq;
        ZERO[lane] = 0; KD[lane] = kq;
    }
    const int D = (npts + k) / 2;
    GaoCw<NL> w;
    gao_cw_init<NL, NW>(w, smem, len, lenT, g0, g1buf + (size_t)c * npts * NW, npts, k, lane, P);
    // (degrees are the same in every lane: said to the compiler, so that the loop's control and pointer arithmetic stay scalar)
    while (!w.done) {
        gao_step_single<NL>(w, lane, ZERO, KD, P, GK);
        gao_after_step<NL>(w, D, lane, P);
    }
    // ---- f = r / v (exact, deg f < k), v = V / cs: everything but the inversion ------------
    gao_division_begins<NL, NW>(w, c, npts, errloc, lane, P);
    while (w.i >= 0) gao_division_round<NL>(w, lane, ZERO, KD, P);
    gao_cw_leaves<NL, NW>(w, c, k, coeffs, errlen, okflag, side, lane, P);
}

// ---------------------------------------------------------------------------------------------------------------------
// Two codewords a wave (point sets of at most 64 points): k_gao_pair
// ---------------------------------------------------------------------------------------------------------------------
// A fused Euclid step of k_gao is two rounds: the remainder (deg r1 lanes: 63 down to (n + k) / 2) and then the cofactor, the scale factor and
// the three jobs that prepare the next step's multipliers (deg t + 5 lanes: 6 up to n - (n + k) / 2 + 5) -- the jobs need the first round's
// two leading coefficients, so the two cannot be one round, and they are 69 lanes anyway.  The second round keeps a third of the wave
// busy and costs what the first one does.  Here a wave works on TWO codewords: their remainder rounds one after the other, their
// second rounds as ONE (codeword A's roles in lanes 0-31, codeword B's in lanes 32-63: a lane's role is nothing but its pointers) --
// three rounds for two steps where k_gao spends four.  The pseudo-division shares its rounds the same way once a codeword's
// remainder is down to 32 coefficients (the last dv + 11 of its 22 rounds at config 4).  A codeword that leaves the regular pattern
// (degree anomalies: structured messages, few errors) or finishes early steps alone (gao_step_single: what k_gao runs),
// and the pairing resumes when both are regular again.  Values are k_gao's, term by term.
// the multipliers of the coming fused step when the previous step did not leave them (first step, after a degree anomaly); the codewords of
// `who` (bit 0: A, bit 1: B), A's three jobs in lanes 28-30, B's in lanes 60-62
template <int NL> __device__ __forceinline__ void gao_pair_multipliers(GaoCw<NL> &A, GaoCw<NL> &B, int who, int lane, const uint32_t *ZERO, const uint32_t *KD, const FpParams<NL> &P) {
    const bool hb = lane >= 32;
    const int l32 = lane & 31;
    const bool mine = hb ? (who & 2) != 0 : (who & 1) != 0;
    uint32_t *R0 = hb ? B.R0 : A.R0, *R1 = hb ? B.R1 : A.R1, *NEGX = hb ? B.NEGX() : A.NEGX(), *S = hb ? B.S() : A.S();
    const int dR1 = hb ? B.dR1 : A.dR1;
    __syncthreads();
    if (mine && l32 < NL) NEGX[l32] = KD[l32] - R1[(size_t)dR1 * NL + l32];
    __syncthreads();
    // (X, Y, Z, W) = (L, lc r0, r0[deg r1], r1[deg r1 - 1])
    const uint32_t *Xa = R1 + (size_t)dR1 * NL, *Ya = R0 + (size_t)(dR1 + 1) * NL, *Za = R0 + (size_t)dR1 * NL, *Wa = R1 + (size_t)(dR1 - 1) * NL;
    const int jb = l32 - 28;
    const bool isJ = mine && jb >= 0 && jb < 3;
    const uint32_t *pm0 = ZERO, *pu = ZERO, *pm1 = ZERO, *pw1 = ZERO;
    if (isJ) { pm0 = jb == 0 ? Xa : NEGX; pu = jb == 0 ? Xa : (jb == 1 ? Ya : Za); if (jb == 2) { pm1 = Ya; pw1 = Wa; } }
    uint32_t r[NL];
    gao_round2<NL>(r, pm0, pu, pm1, pw1, P);
    if (isJ) lds_put<NL>(S + (size_t)jb * NL, r);
    __syncthreads();
}

constexpr int GAO_PAIR_TTOP = 27;        // a codeword's half of the shared second round: cofactor coefficients 0 .. 27, three jobs, the scale factor

template <int NL> __device__ __forceinline__ bool gao_pair_regular(const GaoCw<NL> &w) {
    return !w.done && w.dR0 - w.dR1 == 1 && w.dR1 >= 2 && w.dR1 <= 64 && max(w.dT0, w.dT1 + 1) <= GAO_PAIR_TTOP;
}

// one fused Euclid step of BOTH codewords: three rounds
template <int NL> __device__ __forceinline__ void gao_step_pair(GaoCw<NL> &A, GaoCw<NL> &B, int lane, const uint32_t *ZERO, const uint32_t *KD, const FpParams<NL> &P, const GaoConsts<NL> &GK) {
    const int who = (A.have_sc ? 0 : 1) | (B.have_sc ? 0 : 2);
    if (who) gao_pair_multipliers<NL>(A, B, who, lane, ZERO, KD, P);
    {   // the two remainder rounds as ONE instruction stream: two independent chains for the scheduler (operand loads of one under the
        // multiply-adds of the other, two reductions' carry chains interleaved)
        const int topA = A.dR1, topB = B.dR1;
        const bool actA = lane < topA, actB = lane < topB;
        const int ia = topA - 1 - lane, ib = topB - 1 - lane;
        const uint32_t *puA = ZERO, *pw1A = ZERO, *pw0A = ZERO, *puB = ZERO, *pw1B = ZERO, *pw0B = ZERO;
        if (actA) { puA = A.R0 + (size_t)ia * NL; pw1A = ia >= 1 ? A.R1 + (size_t)(ia - 1) * NL : ZERO; pw0A = A.R1 + (size_t)ia * NL; }
        if (actB) { puB = B.R0 + (size_t)ib * NL; pw1B = ib >= 1 ? B.R1 + (size_t)(ib - 1) * NL : ZERO; pw0B = B.R1 + (size_t)ib * NL; }
        uint32_t ra[NL], rb[NL];
        uint64_t ca[2 * NL], cb[2 * NL];
        col_zero(ca); col_zero(cb);
        { uint32_t m[NL], v[NL], m2[NL], v2[NL]; lds_get<NL>(m, A.S()); lds_get<NL>(v, puA); lds_get<NL>(m2, B.S()); lds_get<NL>(v2, puB); mac<NL>(ca, m, v); mac<NL>(cb, m2, v2); }
        { uint32_t m[NL], v[NL], m2[NL], v2[NL]; lds_get<NL>(m, A.S() + NL); lds_get<NL>(v, pw1A); lds_get<NL>(m2, B.S() + NL); lds_get<NL>(v2, pw1B); mac<NL>(ca, m, v); mac<NL>(cb, m2, v2); }
        { uint32_t m[NL], v[NL], m2[NL], v2[NL]; lds_get<NL>(m, A.S() + 2 * NL); lds_get<NL>(v, pw0A); lds_get<NL>(m2, B.S() + 2 * NL); lds_get<NL>(v2, pw0B); mac<NL>(ca, m, v); mac<NL>(cb, m2, v2); }
        redc(ra, ca, P);
        redc(rb, cb, P);
        __syncthreads();
        if (actA) lds_put<NL>(A.R0 + (size_t)ia * NL, ra);
        if (actB) lds_put<NL>(B.R0 + (size_t)ib * NL, rb);
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < NL; q++) { A.NEGX()[q] = GK.kd[q] - ra[q]; B.NEGX()[q] = GK.kd[q] - rb[q]; }
        }
    }
    __syncthreads();
    {   // the shared second round: cofactor, c0 <- L^2 c0 and the next step's multipliers, of A in lanes 0-31 and of B in lanes 32-63
        const bool hb = lane >= 32;
        const int l32 = lane & 31;
        uint32_t *R0 = hb ? B.R0 : A.R0, *R1 = hb ? B.R1 : A.R1, *T0 = hb ? B.T0 : A.T0, *T1 = hb ? B.T1 : A.T1;
        uint32_t *S = hb ? B.S() : A.S(), *NEGX = hb ? B.NEGX() : A.NEGX(), *CA = hb ? B.CA() : A.CA();
        const int top = hb ? B.dR1 : A.dR1, dT1 = hb ? B.dT1 : A.dT1, dT0 = hb ? B.dT0 : A.dT0;
        const int ttop = max(dT0, dT1 + 1);
        const uint32_t *Xa = R0 + (size_t)(top - 1) * NL, *Ya = R1 + (size_t)top * NL, *Za = R1 + (size_t)(top - 1) * NL, *Wa = R0 + (size_t)(top - 2) * NL;
        const uint32_t *pm0 = S, *pu = ZERO, *pm1 = ZERO, *pw1 = ZERO, *pm2 = ZERO, *pw0 = ZERO;
        uint32_t *dst = nullptr;
        const int jb = l32 - 28;
        if (l32 <= ttop) {
            const int idx = l32;
            pu = T0 + (size_t)idx * NL;
            pw1 = (idx >= 1 && idx - 1 <= dT1) ? T1 + (size_t)(idx - 1) * NL : ZERO;
            pw0 = idx <= dT1 ? T1 + (size_t)idx * NL : ZERO;
            pm1 = S + NL; pm2 = S + 2 * NL; dst = T0 + (size_t)idx * NL;
        } else if (jb >= 0 && jb < 3) {
            pm0 = jb == 0 ? Xa : NEGX; pu = jb == 0 ? Xa : (jb == 1 ? Ya : Za);
            if (jb == 2) { pm1 = Ya; pw1 = Wa; }
            dst = S + (size_t)jb * NL;
        } else if (l32 == 31) { pu = CA; dst = CA; }
        uint32_t r[NL];
        gao_round<NL>(r, pm0, pu, pm1, pw1, pm2, pw0, P);
        __syncthreads();                 // (the jobs overwrite the multipliers every other lane has just read)
        if (dst) lds_put<NL>(dst, r);
        if (l32 < 2) {
#pragma unroll
            for (int q = 0; q < NL; q++) R0[(size_t)(top + l32) * NL + q] = 0;
        }
        __syncthreads();
    }
    A.dT0 = max(A.dT0, A.dT1 + 1); B.dT0 = max(B.dT0, B.dT1 + 1);
    A.have_sc = true; B.have_sc = true;
}

// ... and of two codewords whose remainders are down to 32 coefficients each
template <int NL> __device__ __forceinline__ void gao_division_round_pair(GaoCw<NL> &A, GaoCw<NL> &B, int lane, const uint32_t *ZERO, const uint32_t *KD, const FpParams<NL> &P) {
    const bool hb = lane >= 32;
    const int l32 = lane & 31;
    uint32_t *rp = hb ? B.rp : A.rp, *vp = hb ? B.vp : A.vp, *F = hb ? B.F : A.F, *S = hb ? B.S() : A.S();
    const int i = hb ? B.i : A.i, dv = hb ? B.dv : A.dv;
    uint32_t *LV = S, *CN = S + NL;
    if (l32 < NL) { const uint32_t cq = rp[(size_t)(i + dv) * NL + l32]; F[(size_t)i * NL + l32] = cq; CN[l32] = KD[l32] - cq; }
    __syncthreads();
    const bool act = l32 < i + dv;
    const uint32_t *pu = act ? rp + (size_t)l32 * NL : ZERO;
    const uint32_t *pw = (act && l32 >= i) ? vp + (size_t)(l32 - i) * NL : ZERO;
    uint32_t r[NL];
    gao_round2<NL>(r, LV, pu, CN, pw, P);
    if (act) lds_put<NL>(rp + (size_t)l32 * NL, r);
    __syncthreads();
    A.i -= 1; B.i -= 1;
}

template <int NL, int NW>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GAO_WAVES_PER_EU))) k_gao_pair(const FpParams<NL> P, const uint32_t *__restrict__ g0, const uint32_t *__restrict__ g1buf,
                                            int npts, int k, int64_t C, uint32_t *__restrict__ coeffs, uint32_t *__restrict__ errloc,
                                            int32_t *__restrict__ errlen, uint8_t *__restrict__ okflag, uint32_t *__restrict__ side, const GaoConsts<NL> GK) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int lane = threadIdx.x;
    const int64_t ca = 2 * (int64_t)blockIdx.x, cb = ca + 1;
    const bool two = cb < C;                                 // (an odd batch's last wave has one codeword)
    const int len = npts + 1, lenT = npts - (npts + k) / 2 + 3;
    const int per = gao_cw_words<NL>(len, lenT);
    uint32_t *ZERO = smem + (size_t)2 * per, *KD = ZERO + NL;
    if (lane < NL) {
        uint32_t kq = 0;
#pragma unroll
        for (int q = 0; q < NL; q++) kq = lane == q ? GK.kd[q] : kq;
        ZERO[lane] = 0; KD[lane] = kq;
    }
    const int D = (npts + k) / 2;
    GaoCw<NL> A, B;
    gao_cw_init<NL, NW>(A, smem, len, lenT, g0, g1buf + (size_t)ca * npts * NW, npts, k, lane, P);
    gao_cw_init<NL, NW>(B, smem + per, len, lenT, g0, g1buf + (size_t)(two ? cb : ca) * npts * NW, npts, k, lane, P);
    if (!two) gao_loop_ends<NL>(B, true);
    while (!A.done || !B.done) {
        if (gao_pair_regular<NL>(A) && gao_pair_regular<NL>(B)) {
            gao_step_pair<NL>(A, B, lane, ZERO, KD, P, GK);
            gao_after_step<NL>(A, D, lane, P);
            gao_after_step<NL>(B, D, lane, P);
        } else {
            if (!A.done) { gao_step_single<NL>(A, lane, ZERO, KD, P, GK); gao_after_step<NL>(A, D, lane, P); }
            if (!B.done) { gao_step_single<NL>(B, lane, ZERO, KD, P, GK); gao_after_step<NL>(B, D, lane, P); }
        }
    }
    // ---- f = r / v (exact, deg f < k), v = V / cs: everything but the inversion ------------
    gao_division_begins<NL, NW>(A, ca, npts, errloc, lane, P);
    if (two) gao_division_begins<NL, NW>(B, cb, npts, errloc, lane, P); else { B.i = -1; B.divided = false; B.ok = false; }
    __syncthreads();
    while (A.i >= 0 || B.i >= 0) {
        if (A.i >= 0 && B.i >= 0 && A.i + A.dv <= 32 && B.i + B.dv <= 32) gao_division_round_pair<NL>(A, B, lane, ZERO, KD, P);
        else {
            if (A.i >= 0) gao_division_round<NL>(A, lane, ZERO, KD, P);
            if (B.i >= 0) gao_division_round<NL>(B, lane, ZERO, KD, P);
        }
    }
    gao_cw_leaves<NL, NW>(A, ca, k, coeffs, errlen, okflag, side, lane, P);
    if (two) gao_cw_leaves<NL, NW>(B, cb, k, coeffs, errlen, okflag, side, lane, P);
}

// A workgroup of one wave finishes 64 GAO_FIN_G consecutive codewords.  Lane t owns codewords base + t, base + 64 + t, ... (GAO_FIN_G of them):
// w_j = cs_j l_j and ONE inversion for the lane's group (Montgomery's trick: prefix products, invert the last, peel backwards -- an inversion is
// ~380 multiplications, the three per codeword that replace it are not), then 1 / cs = l / w scales the cofactor into the reference's
// un-normalised error locator, and powers of 1 / l = cs / w turn the raw quotient digits into coefficients: f_i = c_i / l^(dq - i + 1).
// The scale factors are taken OUT of Montgomery form once, so that one multiplication both scales an element and converts it.
// Rounds 3-4 let every lane walk its own codewords' elements in place -- 32-byte accesses 1 KB apart across the wave: 0.86 ms at config 4, bound
// by the 36 M uncoalesced requests, not by the arithmetic.  Now a lane only produces its codeword's FACTORS (the chain of powers, into LDS);
// the wave then scales the 64 codewords of slice j together: their coefficients are one contiguous run of 64 k elements, element e on lane
// e mod 64 -- coalesced in, coalesced out.
#ifndef GAO_FIN_GROUP
#define GAO_FIN_GROUP 4
#endif
constexpr int GAO_FIN_G = GAO_FIN_GROUP;
template <int NL, int NW>
__global__ void __launch_bounds__(64) k_gao_finish(const FpParams<NL> P, int npts, int k, int64_t C, uint32_t *__restrict__ coeffs,
                                                   uint32_t *__restrict__ errloc, const int32_t *__restrict__ errlen,
                                                   const uint8_t *__restrict__ okflag, const uint32_t *__restrict__ side, int kp) {
    extern __shared__ uint32_t fin_lds[];
    uint32_t *T = fin_lds;                                   // [64][kp][NL]: codeword t's factor of coefficient lo + i (plain, not Montgomery)
    uint32_t *IC = T + (size_t)64 * kp * NL;                 // [64][NL]: 1 / cs
    int32_t *DF = reinterpret_cast<int32_t *>(IC + 64 * NL); // [64]: df, or -2 for a codeword that did not decode
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * 64 * GAO_FIN_G;
    if (base >= C) return;
    uint32_t pref[GAO_FIN_G][NL];            // pref[j] = w_0 ... w_j over the lane's decoded codewords (the others contribute 1)
    bool any = false;
#pragma unroll
    for (int j = 0; j < GAO_FIN_G; j++) {
        const int64_t c = base + (int64_t)j * 64 + lane;
        uint32_t w[NL];
        fp_set(w, P.one);
        if (c < C && okflag[c]) {
            uint32_t cs[NL], l[NL];
            load_digits<NL, NW>(cs, side + (size_t)c * (2 * NW + 4));
            load_digits<NL, NW>(l, side + (size_t)c * (2 * NW + 4) + NW);
            mont_mul(w, cs, l, P);
            any = true;
        }
        if (j == 0) fp_set(pref[0], w); else mont_mul(pref[j], pref[j - 1], w, P);
    }
    uint32_t run[NL];                        // 1 / (w_0 ... w_j) while codeword j is being finished
    if (any) fp_inv(run, pref[GAO_FIN_G - 1], P); else fp_set(run, P.one);
#pragma unroll
    for (int j = GAO_FIN_G - 1; j >= 0; j--) {
        const int64_t c = base + (int64_t)j * 64 + lane;
        const bool live = c < C && okflag[c];
        int df = -2, dq = -1;
        uint32_t pw[NL], linv[NL];
        if (live) {
            uint32_t cs[NL], l[NL], w[NL], winv[NL], inv_c[NL];
            load_digits<NL, NW>(cs, side + (size_t)c * (2 * NW + 4));
            load_digits<NL, NW>(l, side + (size_t)c * (2 * NW + 4) + NW);
            dq = (int)side[(size_t)c * (2 * NW + 4) + 2 * NW];
            df = (int)side[(size_t)c * (2 * NW + 4) + 2 * NW + 1];
            if (j > 0) mont_mul(winv, run, pref[j - 1], P); else fp_set(winv, run);
            mont_mul(w, cs, l, P);
            mont_mul(run, run, w, P);
            mont_mul(inv_c, winv, l, P);               // 1 / cs
            mont_mul(linv, winv, cs, P);               // 1 / l
            uint32_t inv_c_plain[NL];
            from_mont(inv_c_plain, inv_c, P);          // out of Montgomery form: (u R) b / R = u b, canonical
            from_mont(pw, linv, P);
#pragma unroll
            for (int q = 0; q < NL; q++) IC[lane * NL + q] = inv_c_plain[q];
            // the chain runs i = dq .. 0, the power of 1 / l growing by one per step: the steps above the table's reach first
            for (int i = dq; i >= k; i--) mont_mul(pw, pw, linv, P);
        }
        DF[lane] = df;
        const int64_t c0 = base + (int64_t)j * 64;
        const int cnt = (int)min((int64_t)64, C - c0);
        // the coefficients in parts of kp (the factor table of a whole codeword slice does not leave room for four workgroups a CU, and the
        // inversion's latency wants them all resident): coefficients [lo, hi) from the top, the chain continuing from part to part
        for (int hi = k; hi > 0; hi -= kp) {
            const int lo = max(hi - kp, 0);
            if (live) {
                for (int i = min(hi - 1, dq); i >= lo; i--) {
                    if (i <= df) {
#pragma unroll
                        for (int q = 0; q < NL; q++) T[((size_t)lane * kp + (i - lo)) * NL + q] = pw[q];
                    }
                    if (i > 0) mont_mul(pw, pw, linv, P);
                }
            }
            __syncthreads();
            const int wdt = hi - lo;
            for (int e = lane; e < cnt * wdt; e += 64) {
                const int t = e / wdt, i = lo + (e - t * wdt);
                if (i <= DF[t]) {
                    uint32_t u[NL], f[NL], fac[NL];
                    load_digits<NL, NW>(u, coeffs + (((size_t)c0 + t) * k + i) * NW);
#pragma unroll
                    for (int q = 0; q < NL; q++) fac[q] = T[((size_t)t * kp + (i - lo)) * NL + q];
                    mont_mul(f, u, fac, P);
                    store_digits<NL, NW>(coeffs + (((size_t)c0 + t) * k + i) * NW, f);
                }
            }
            __syncthreads();
        }
        if (errloc) {
            const int W = npts + 1;
            for (int e = lane; e < cnt * W; e += 64) {
                const int t = e / W, i = e - t * W;
                if (DF[t] != -2 && i < errlen[c0 + t]) {
                    uint32_t u[NL], f[NL], fac[NL];
                    load_digits<NL, NW>(u, errloc + ((size_t)c0 * W + e) * NW);
#pragma unroll
                    for (int q = 0; q < NL; q++) fac[q] = IC[t * NL + q];
                    mont_mul(f, u, fac, P);
                    store_digits<NL, NW>(errloc + ((size_t)c0 * W + e) * NW, f);
                }
            }
        }
        __syncthreads();
    }
}

// The walk of rounds 3-4 (one lane per GAO_FIN_G consecutive codewords, every element visited in place by its codeword's lane): kept for message
// lengths whose factor table does not fit the LDS (k > 66 at 32-byte elements).
template <int NL, int NW>
__global__ void __launch_bounds__(64) k_gao_finish_walk(const FpParams<NL> P, int npts, int k, int64_t C, uint32_t *__restrict__ coeffs,
                                                   uint32_t *__restrict__ errloc, const int32_t *__restrict__ errlen,
                                                   const uint8_t *__restrict__ okflag, const uint32_t *__restrict__ side) {
    const int64_t base = ((int64_t)blockIdx.x * 64 + threadIdx.x) * GAO_FIN_G;
    if (base >= C) return;
    const int cnt = (int)min((int64_t)GAO_FIN_G, C - base);
    uint32_t pref[GAO_FIN_G][NL];            // pref[j] = w_0 ... w_j over the decoded codewords (the others contribute 1)
    bool any = false;
#pragma unroll
    for (int j = 0; j < GAO_FIN_G; j++) {
        uint32_t w[NL];
        fp_set(w, P.one);
        if (j < cnt && okflag[base + j]) {
            uint32_t cs[NL], l[NL];
            load_digits<NL, NW>(cs, side + (size_t)(base + j) * (2 * NW + 4));
            load_digits<NL, NW>(l, side + (size_t)(base + j) * (2 * NW + 4) + NW);
            mont_mul(w, cs, l, P);
            any = true;
        }
        if (j == 0) fp_set(pref[0], w); else mont_mul(pref[j], pref[j - 1], w, P);
    }
    if (!any) return;
    uint32_t run[NL];                        // 1 / (w_0 ... w_j) while codeword j is being finished
    fp_inv(run, pref[GAO_FIN_G - 1], P);
#pragma unroll
    for (int j = GAO_FIN_G - 1; j >= 0; j--) {
        if (j >= cnt || !okflag[base + j]) continue;          // (its factor was 1)
        const int64_t c = base + j;
        uint32_t cs[NL], l[NL], w[NL], winv[NL], inv_c[NL], linv[NL];
        load_digits<NL, NW>(cs, side + (size_t)c * (2 * NW + 4));
        load_digits<NL, NW>(l, side + (size_t)c * (2 * NW + 4) + NW);
        const int dq = (int)side[(size_t)c * (2 * NW + 4) + 2 * NW], df = (int)side[(size_t)c * (2 * NW + 4) + 2 * NW + 1];
        if (j > 0) mont_mul(winv, run, pref[j - 1], P); else fp_set(winv, run);
        mont_mul(w, cs, l, P);
        mont_mul(run, run, w, P);
        mont_mul(inv_c, winv, l, P);               // 1 / cs
        mont_mul(linv, winv, cs, P);               // 1 / l
        uint32_t inv_c_plain[NL], pw[NL];
        from_mont(inv_c_plain, inv_c, P);          // out of Montgomery form: (u R) b / R = u b, canonical
        from_mont(pw, linv, P);
        const int nloc = errloc ? errlen[c] : 0;
        // (four elements' loads in flight at a time: in place, so the compiler will not move a load above the previous store by itself)
        for (int e0 = 0; e0 < nloc; e0 += 4) {
            uint32_t u[4][NL];
#pragma unroll
            for (int v = 0; v < 4; v++)
                if (e0 + v < nloc) load_digits<NL, NW>(u[v], errloc + ((size_t)c * (npts + 1) + e0 + v) * NW);
#pragma unroll
            for (int v = 0; v < 4; v++)
                if (e0 + v < nloc) {
                    uint32_t e[NL];
                    mont_mul(e, u[v], inv_c_plain, P);
                    store_digits<NL, NW>(errloc + ((size_t)c * (npts + 1) + e0 + v) * NW, e);
                }
        }
        // i = dq .. 0: the power of 1 / l grows by one per step; only i <= df < k carry a non-zero digit
        for (int i0 = dq; i0 >= 0; i0 -= 4) {
            uint32_t u[4][NL];
#pragma unroll
            for (int v = 0; v < 4; v++)
                if (i0 - v >= 0 && i0 - v <= df) load_digits<NL, NW>(u[v], coeffs + ((size_t)c * k + i0 - v) * NW);
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int i = i0 - v;
                if (i < 0) break;
                if (i <= df) {
                    uint32_t f[NL];
                    mont_mul(f, u[v], pw, P);
                    store_digits<NL, NW>(coeffs + ((size_t)c * k + i) * NW, f);
                }
                if (i > 0) mont_mul(pw, pw, linv, P);
            }
        }
    }
}

}  // namespace

extern "C" int hb_gao_decode(hb_ctx *ctx, const uint64_t *x_host, int npts, int k, const uint64_t *ys_dev, int64_t C,
                             uint64_t *coeffs_dev, uint64_t *errloc_dev, int32_t *errloc_len_dev, uint8_t *ok_dev, void *stream) {
    if (C > 0 && !errloc_dev) return HB_ERR_BAD_ARG;
    return hb::gao_decode(ctx, x_host, npts, k, ys_dev, C, coeffs_dev, errloc_dev, errloc_len_dev, ok_dev, stream);
}

// errloc_dev == nullptr: the error locators are not written (nor scaled by the finishing kernel), their lengths are -- what
// hb_wb_decode's pre-pass needs of them (3.3 KB of HBM traffic per codeword at config 4 and a third of the finisher's work)
// sel_host != nullptr: the codewords lie in rows of ys_stride symbols and symbol i of a word is its column sel_host[i] (a batch whose codewords
// all lost the SAME symbols: the reference drops the erasures and decodes over the points that are left, hbmpc_ntl_helpers.pyx:399-403,
// reed_solomon.py:201-204 -- here the interpolant's launch simply reads the surviving columns in place)
int hb::gao_decode(hb_ctx *ctx, const uint64_t *x_host, int npts, int k, const uint64_t *ys_dev, int64_t C,
                   uint64_t *coeffs_dev, uint64_t *errloc_dev, int32_t *errloc_len_dev, uint8_t *ok_dev, void *stream, int ys_stride, const int32_t *sel_host) { HB_API_GUARD(ctx);
    if (!ctx || !x_host || npts < 1 || k < 0 || C < 0) return HB_ERR_BAD_ARG;
    if (C == 0) return HB_OK;
    if (!ys_dev || !coeffs_dev || !errloc_len_dev || !ok_dev) return HB_ERR_BAD_ARG;
    if (npts > 1023) return fail(ctx, HB_ERR_UNSUPPORTED, "gao: more than 1023 points");
    if (C > 0x7fffffffLL) return fail(ctx, HB_ERR_UNSUPPORTED, "gao: batch too large");
    hipStream_t s = (hipStream_t)stream;
    const int NLr = ctx->nl();
    cache_trim(ctx);
    // tables for this point set
    hb_matrix *Vi = nullptr;
    int rc = hb_vand_inverse_create(ctx, x_host, npts, &Vi, stream); if (rc) return rc;
    struct Unref { hb_matrix *m; ~Unref() { matrix_unref(m); } } unref_vi{Vi};   // this call's handle; the cache keeps its own
    // g0 = prod (X - x_i) does not depend on the order of the points: keyed by the sorted set
    std::vector<std::string> pts((size_t)npts);
    for (int i = 0; i < npts; i++) pts[i].assign(reinterpret_cast<const char *>(x_host + (size_t)i * ctx->n_limbs), (size_t)ctx->n_limbs * 8);
    std::sort(pts.begin(), pts.end());
    std::string key = "g0:" + std::to_string(npts) + ":";
    for (auto &pt : pts) key += pt;
    uint32_t *g0 = nullptr;
    auto it = ctx->dcache.find(key);
    if (it != ctx->dcache.end()) { g0 = (uint32_t *)it->second; cache_touch(ctx, "d|" + key); }
    else {
        uint32_t *xd = nullptr;
        rc = upload_elems(ctx, x_host, (size_t)npts, &xd, s); if (rc) return rc;
        HB_HIP(ctx, hipMalloc(&g0, (size_t)(npts + 1) * NLr * 4));
        int threads = ((npts + 1 + 63) / 64) * 64;
        size_t lds = (size_t)(npts + 2 * (npts + 1)) * NLr * 4;
        if (ctx->n_limbs == 4) {
            HB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_poly_from_roots<9, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            k_poly_from_roots<9, 8><<<1, threads, lds, s>>>(ctx->pw, xd, npts, g0);
        } else {
            HB_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(k_poly_from_roots<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            k_poly_from_roots<3, 2><<<1, threads, lds, s>>>(ctx->pn, xd, npts, g0);
        }
        HB_LAUNCH_CHECK(ctx);
        HB_HIP(ctx, hipStreamSynchronize(s));
        HB_HIP(ctx, hipFree(xd));
        ctx->dcache[key] = g0;
        cache_note(ctx, "d|" + key, [ctx, key]() { auto f = ctx->dcache.find(key); if (f != ctx->dcache.end()) { (void)hipFree(f->second); ctx->dcache.erase(f); } });
    }
    // g1 for every codeword: chunk-major [C][npts] (coefficient-major made every 32-byte read of k_gao fetch a whole 128-byte line: 12.5 KB per codeword for 3.2)
    // (g1 and the side records are context scratch -- ctx_scratch, hb_common.hpp: this call holds the context's mutex and synchronises before it returns)
    uint32_t *g1 = nullptr;
    rc = ctx_scratch(ctx, "gao.g1", (size_t)npts * C * ctx->elem_words() * 4, (void **)&g1); if (rc) return rc;
    hb_view iv{sel_host ? ys_stride : npts, 1}, ov{npts, 1};
    int32_t *sel_dev = nullptr;
    if (sel_host) { rc = get_int_array(ctx, sel_host, npts, &sel_dev, s); if (rc) return rc; }
    rc = launch_matvec(ctx, Vi, (const uint32_t *)ys_dev, iv, sel_dev, INT64_MAX, g1, ov, INT64_MAX, nullptr, nullptr, C, s);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }          // (nothing of this call may still be writing the scratch when the next one starts)
    size_t lds = (size_t)(2 * (npts + 1) + 2 * (npts - (npts + k) / 2 + 3) + 8) * NLr * 4;      // R0, R1, T0, T1, a zero element, three multipliers, -X, c0, c1, K
    // point sets of at most 64 points, batches that fill the chip several times over: two codewords a wave (k_gao_pair: 19 % fewer vector
    // instructions a codeword, 11 % less time at n = 64; a wave's own run is 1.4 times longer, so a batch that leaves SIMDs idle anyway keeps
    // one codeword a wave).  HB_GAO_PAIR=1 / 0 forces the choice (same values either way: tests/test_gpu_parity.py)
    const char *pair_env = env_hook(ENV_GAO_PAIR);
    const bool pair = npts <= 64 && C >= 2 && (pair_env ? pair_env[0] == '1' : C >= 12288);
    const size_t pair_lds = (size_t)(2 * (2 * (npts + 1) + 2 * (npts - (npts + k) / 2 + 3) + 6) + 2) * NLr * 4;
    // side record per codeword (cs, lc(V), dq, df) between the Euclid kernel and the finishing one
    uint32_t *side = nullptr;
    const size_t side_words = (size_t)(2 * ctx->elem_words() + 4);      // (16-byte rows)
    rc = ctx_scratch(ctx, "gao.side", (size_t)C * side_words * 4, (void **)&side);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    const unsigned fin_blocks = (unsigned)((C + 64 * GAO_FIN_G - 1) / (64 * GAO_FIN_G));
    // the finisher's factor table: kp coefficients of 64 codewords at a time, sized so that four workgroups share a CU's LDS
    const int kp = std::max(1, std::min(k, (int)((28 * 1024 - 64 * NLr * 4 - 256) / (64 * NLr * 4))));
    const size_t fin_lds = ((size_t)64 * kp * NLr + 64 * NLr + 64) * 4;
    const bool fin_walk = false;
    const unsigned walk_blocks = (unsigned)(((C + GAO_FIN_G - 1) / GAO_FIN_G + 63) / 64);
    // K = p 2^j with 29 (NL - 1) + 27 bits, in redundant radix-2^29 digits (GaoConsts): a unit of every digit above the lowest lent to the digit below
    uint32_t kd[9];
    {
        const int W = ctx->n_limbs * 2;                       // p as 32-bit words
        uint32_t pw32[8], kw[10];
        for (int i = 0; i < W; i++) pw32[i] = (uint32_t)(ctx->p_limbs[i / 2] >> (32 * (i & 1)));
        int bits = 0;
        for (int i = W - 1; i >= 0 && !bits; i--)
            if (pw32[i]) bits = 32 * i + (32 - __builtin_clz(pw32[i]));
        const int j = 29 * (NLr - 1) + 27 - bits, jw = j >> 5, jb = j & 31;      // (bits <= 32 W <= 29 (NL - 1) + 27: j >= 0)
        memset(kw, 0, sizeof kw);
        for (int i = 0; i < W; i++) {
            const uint64_t v = (uint64_t)pw32[i] << jb;
            if (i + jw < 10) kw[i + jw] |= (uint32_t)v;
            if (i + jw + 1 < 10) kw[i + jw + 1] |= (uint32_t)(v >> 32);
        }
        for (int q = 0; q < NLr; q++) {
            const int bit = 29 * q, w = bit >> 5, sft = bit & 31;
            const uint64_t v = (uint64_t)kw[w] | ((uint64_t)(w + 1 < 10 ? kw[w + 1] : 0) << 32);
            const uint32_t d = q < NLr - 1 ? (uint32_t)(v >> sft) & DMASK : (uint32_t)(v >> sft);
            kd[q] = q == 0 ? d + (1u << LB) : (q < NLr - 1 ? d + (1u << LB) - 1 : d - 1);
        }
    }
    if (ctx->n_limbs == 4) {
        GaoConsts<9> gk;
        memcpy(gk.kd, kd, sizeof gk.kd);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gao<9, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (!fin_walk) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gao_finish<9, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_lds);
        if (pair) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gao_pair<9, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_lds);
            k_gao_pair<9, 8><<<(unsigned)((C + 1) / 2), 64, pair_lds, s>>>(ctx->pw, g0, g1, npts, k, C, (uint32_t *)coeffs_dev, (uint32_t *)errloc_dev, errloc_len_dev, ok_dev, side, gk);
        } else
            k_gao<9, 8><<<(unsigned)C, 64, lds, s>>>(ctx->pw, g0, g1, npts, k, C, (uint32_t *)coeffs_dev, (uint32_t *)errloc_dev, errloc_len_dev, ok_dev, side, gk);
        if (fin_walk) k_gao_finish_walk<9, 8><<<walk_blocks, 64, 0, s>>>(ctx->pw, npts, k, C, (uint32_t *)coeffs_dev, (uint32_t *)errloc_dev, errloc_len_dev, ok_dev, side);
        else k_gao_finish<9, 8><<<fin_blocks, 64, fin_lds, s>>>(ctx->pw, npts, k, C, (uint32_t *)coeffs_dev, (uint32_t *)errloc_dev, errloc_len_dev, ok_dev, side, kp);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gao<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (!fin_walk) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gao_finish<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fin_lds);
        GaoConsts<3> gk;
        memcpy(gk.kd, kd, sizeof gk.kd);
        if (pair) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_gao_pair<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_lds);
            k_gao_pair<3, 2><<<(unsigned)((C + 1) / 2), 64, pair_lds, s>>>(ctx->pn, g0, g1, npts, k, C, (uint32_t *)coeffs_dev, (uint32_t *)errloc_dev, errloc_len_dev, ok_dev, side, gk);
        } else
            k_gao<3, 2><<<(unsigned)C, 64, lds, s>>>(ctx->pn, g0, g1, npts, k, C, (uint32_t *)coeffs_dev, (uint32_t *)errloc_dev, errloc_len_dev, ok_dev, side, gk);
        if (fin_walk) k_gao_finish_walk<3, 2><<<walk_blocks, 64, 0, s>>>(ctx->pn, npts, k, C, (uint32_t *)coeffs_dev, (uint32_t *)errloc_dev, errloc_len_dev, ok_dev, side);
        else k_gao_finish<3, 2><<<fin_blocks, 64, fin_lds, s>>>(ctx->pn, npts, k, C, (uint32_t *)coeffs_dev, (uint32_t *)errloc_dev, errloc_len_dev, ok_dev, side, kp);
    }
    const hipError_t le = hipGetLastError();
    const hipError_t se = hipStreamSynchronize(s);
    if (le != hipSuccess || se != hipSuccess) { ctx->err = std::string("gao: ") + hipGetErrorString(le != hipSuccess ? le : se); return HB_ERR_HIP; }
    return HB_OK;
}
