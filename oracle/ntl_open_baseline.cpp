// ntl_open_baseline.cpp -- OPTIONAL CPU baseline driver, test/bench infrastructure only.
//
// "probe for NTL at run time; if present, time the same NTL calls the reference
// makes".  This is our own driver (not reference code): it performs, with NTL's own types, the call
// sequence of one fault-free per-party open as the reference issues it through
// hbmpc_ntl_helpers.pyx -- set_vm_matrix (rsdecode_impl.h:23-36), mat_ZZ_p mul for
// vandermonde_batch_evaluate (pyx:237), NTL inv + mul for vandermonde_batch_interpolate
// (rsdecode_impl.h:97-122, pyx:160-183; the inverse is recomputed on every call like the reference)
// -- 3 encodes + 2 decodes + the column comparisons, on buffers that are already NTL objects
// ("kernel-only" variant).  bench.py compiles it only when <NTL/ZZ_p.h> and libntl are found on the
// host; NTL is NOT installed in the build container, so this file could not be exercised there and
// bench.py falls back to oracle/hbmpc_oracle.c (labelled) whenever compiling or running it fails.
//
//   usage: ntl_open_baseline n t B threads   -> prints seconds for one open
#include <NTL/BasicThreadPool.h>
#include <NTL/ZZ_p.h>
#include <NTL/mat_ZZ_p.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

using namespace NTL;

static void vm_matrix(mat_ZZ_p &V, long n, long d) {
    V.SetDims(n, d);
    for (long i = 0; i < n; i++) {
        ZZ_p x = to_ZZ_p(i + 1), pw = to_ZZ_p(1);
        for (long l = 0; l < d; l++) { V[i][l] = pw; pw *= x; }
    }
}

static long decode_validate(const mat_ZZ_p &V, const mat_ZZ_p &cols, long d, long t, mat_ZZ_p &coef) {
    const long C = cols.NumCols();
    mat_ZZ_p Vd, Vi, Y, re;
    Vd.SetDims(d, d); Y.SetDims(d, C);
    for (long j = 0; j < d; j++) { Vd[j] = V[j]; Y[j] = cols[j]; }
    inv(Vi, Vd);                       // vandermonde_inverse, recomputed per call (pyx:160-170)
    mul(coef, Vi, Y);                  // decode
    mul(re, V, coef);                  // validating re-encode of all n rows (reed_solomon.py:313)
    long bad = 0;
    for (long j = d; j < d + t; j++)   // compare the next t arrivals (reed_solomon.py:316-326)
        for (long c = 0; c < C; c++) bad += (re[j][c] != cols[j][c]);
    return bad;
}

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    const long n = atol(argv[1]), t = atol(argv[2]), B = atol(argv[3]), threads = atol(argv[4]);
    const long d = t + 1, C = (B + d - 1) / d;
    ZZ p = conv<ZZ>("52435875175126190479447740508185965837690552500527637822603658699938581184513");
    ZZ_p::init(p);
    SetNumThreads(threads);
    mat_ZZ_p V, M1, M2, R1c, R2c;
    vm_matrix(V, n, d);
    M1.SetDims(d, C); M2.SetDims(d, C);
    for (long l = 0; l < d; l++)
        for (long c = 0; c < C; c++) { M1[l][c] = random_ZZ_p(); M2[l][c] = random_ZZ_p(); }
    mul(R1c, V, M1);                   // consistent received columns (setup, untimed)
    mul(R2c, V, M2);
    auto t0 = std::chrono::steady_clock::now();
    mat_ZZ_p out, c1, c2;
    mul(out, V, M1);                                  // R1 encode
    long bad = decode_validate(V, R1c, d, t, c1);     // R1 decode + validate
    bad += decode_validate(V, R2c, d, t, c2);         // R2 decode + validate
    auto t1 = std::chrono::steady_clock::now();
    if (bad || c2 != M2) { fprintf(stderr, "mismatch\n"); return 1; }
    printf("%.6f\n", std::chrono::duration<double>(t1 - t0).count());
    return 0;
}
