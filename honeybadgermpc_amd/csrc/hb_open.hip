// hb_open.hip -- one party's fault-free batch open, entirely on device.
//
// Restates the compute of batch_reconstruct (reference: honeybadgermpc/batch_reconstruction.py:158-227)
// with IncrementalDecoder's optimistic path (honeybadgermpc/reed_solomon.py:305-330):
//   R1: chunk_data + enc.encode + transpose_lists            (batch_reconstruction.py:158-167)
//   R1 receive: decode_batch from the first d arrivals, encode_batch of the guess, compare
//       every later arrival against it                       (reed_solomon.py:308-326)
//   R2 message = constant terms                              (batch_reconstruction.py:194)
//   R2 receive: same decode / re-encode / compare, then flatten_lists + truncate (:223-227)
// = 3 batch encodes + 2 batch decodes per open (SURVEY.md 3.1 census).
//
// Layout: everything that crosses the party boundary is party-major [n][C] (row j is the
// message to / from party j), so the reference's transpose_lists is a stride choice in the
// encode kernel, not a pass over memory.  Decoded coefficients are kept coefficient-major
// [d][C] so lane = chunk loads are fully coalesced and the R2 message is row 0.
#include "hb_common.hpp"

using namespace hb;

// Kernel pipeline of one open (hb_fast.hip): raw small-entry matrices, inputs pre-scaled into
// Montgomery digit planes, decode emitting Montgomery-form coefficients for the validating re-encode:
//   R1 encode : k_prescale(shares, K=R^2)        -> in_dg ; k_matvec2(V)  -> r1_out (canonical)
//   decode    : k_prescale(cols[z], K=R^3/den_j) -> in_dg ; k_matvec2(N, negrow) -> coef_dg (Montgomery) [+ canonical rows]
//   validate  : k_matvec2<CHECK>(V, coef_dg) compared with cols[zc] in the epilogue
struct hb_open_plan {
    hb_ctx *ctx;
    int n, d, n_check;
    int64_t max_B, max_C;
    FastMatrix *V;       // n x d  raw Vandermonde at the n party points
    FastMatrix *Vinv;    // d x d  factored inverse for the arrival set z
    Mm8Matrix *V8;       // int8 matrix-core image of V (hb_mfma.hip); nullptr when that path does not apply
    Mm8Matrix *Vzc8;     // rows zc of V only: the validating re-encode under HB_OPEN_OPT_VALIDATE_ARRIVED_ONLY (built on demand)
    int32_t *zc_dev, *ones_dev;   // its expected-row map and compare mask
    std::vector<int32_t> zc;
    std::vector<int32_t> z;            // the arrival set and the points: kept for tables built on request (set_option)
    std::vector<uint64_t> x;
    Mm8Matrix *Vinv8;    // same for the numerators N of the factored inverse (decode on the matrix cores); may be nullptr
    uint32_t *scaled_pk; // [d][max_C] received columns / den_j, the matrix-core decode's input
    int use_v8;          // option HB_OPEN_OPT_MATRIX_CORES (default 1)
    int use_fused;       // option HB_OPEN_OPT_FUSED_VALIDATE (default 1)
    int fused_pending;   // the fused matrices apply to this plan but are not built yet: a plan pays for them (1-2.5 ms) once it
    int decode_calls;    // has decoded twice -- arrival sets that are seen once (a decoder probing its way past liars) never do
    int32_t *z_dev;      // d row indices
    int32_t *mask_dev;   // n+1 ints: rows to validate
    uint32_t *in_dg;     // [d][NL][max_C] pre-scaled inputs (digit planes)
    uint32_t *coef_dg;   // [d][NL][max_C] decoded coefficients, Montgomery digit planes
    int32_t *mismatch_dev;
    int validate_arrived_only;   // option: re-encode only the tiles that contain compared rows
    // omega-power points with a transform large enough to pay: encodes run as radix-2 NTTs
    // (hb_ntt.hip) on canonical coefficients instead of n x d products
    int ntt_order;               // 0 = mat-vec encodes
    uint32_t *tw;                // twiddles (ctx-owned cache)
    uint32_t *coef_pk;           // [d][max_C] canonical decoded coefficients (NTT input), NTT mode only
    // entries that are not small integers (omega-power points; powers beyond 2^127 such as n = 100, t = 33): the
    // full-size matrix-core kernel (hb_mfma_wide.hip) over the plain tables; the plan holds references to ctx tables
    hb_matrix *Winv;             // full d x d inverse at the arrival set
    const Mm8wMatrix *Winv8;     // its int8 image, owned by Winv; nullptr: integer-VALU decode
    hb_matrix *Vw;               // full n x d Vandermonde table (only when the encodes are not NTTs)
    const Mm8wMatrix *Vw8;       // its int8 image; nullptr: integer-VALU encodes
    // Fused decode + validate on that kernel: the prediction of a later arrival is a LINEAR function of the arrival set,
    // V[zc] (Winv y) = (V[zc] Winv) y, so one launch over the rows [Winv rows the caller wants ; V[zc] Winv] decodes and
    // compares (same canonical values as interpolating first and evaluating after: exact arithmetic mod p).
    Mm8wMatrix *F1, *F2;         // [Winv row 0 ; V[zc] Winv] for the R1 message, [Winv ; V[zc] Winv] for the R2 result (owned)
    int32_t *fmap1, *fmap2;      // per row: 0 = a result row, else 1 + the received row to compare the prediction with
    // The same two matrices built ON THE DEVICE at plan creation (hb_quick.hip: Lagrange from the point set's table of inverse
    // differences, image and row constants by two small kernels; nothing waited for): where that applies the plan decodes +
    // validates in one launch from its first decode on, and F1 / F2 (host-built: 1-2.5 ms with a stream synchronise, which is why
    // they used to wait for a plan's third decode) are never needed.
    uint8_t *q1, *q2;            // owned buffers (QuickLayout::need bytes each)
    QuickLayout l1, l2;
    const Mm8wShared *qsh;
    // At points that are small integers the same two products factor as [N ; P] (y ./ den) with small INTEGER matrices: the small-entry
    // kernel with the division inside (hb_mfma_fused.hip), half the matrix-core work of the full-size one.  Built on the device at plan
    // creation; where it applies q1 / q2 are only built on request (HB_OPEN_OPT_FUSED_VALIDATE = 2).
    uint8_t *fs1, *fs2;          // owned buffers (FsLayout::need bytes each)
    FsLayout fl1, fl2;
    // One-limb contexts (p < 2^64, the north star's 64-bit prime): the 8-byte mat-vec of hb_narrow.hip -- V for the encode, and decode + validate
    // as one launch over [V^-1 row 0 ; V[zc] V^-1] (R1) / [V^-1 ; V[zc] V^-1] (R2), every entry a word-size residue
    Mv64Matrix *nv_enc, *nv_f1, *nv_f2;
};

// the device-built images of the full-size kernel (q1, q2); failures leave the plan as it is
static void build_quick_wide(hb_open_plan *pl, hipStream_t s) {
    hb_ctx *ctx = pl->ctx;
    if (pl->q1 || env_hook(ENV_NO_QUICK_PLAN)) return;
    QuickLayout a, b;
    const int n = pl->n, d = pl->d, n_check = pl->n_check;
    if (quick_layout(ctx, n, d, n_check, 1, &a) == HB_OK && quick_layout(ctx, n, d, n_check, d, &b) == HB_OK) {
        uint8_t *b1 = nullptr, *b2 = nullptr;
        const Mm8wShared *sh = nullptr;
        int qrc = HB_OK;
        if (hipMalloc(&b1, a.need) != hipSuccess || hipMalloc(&b2, b.need) != hipSuccess) qrc = HB_ERR_HIP;
        const int32_t *zcp = n_check > 0 ? pl->zc.data() : pl->z.data();      // (unused when n_check == 0)
        if (!qrc) qrc = quick_build(ctx, pl->x.data(), pl->z.data(), zcp, a, b1, &sh, s);
        if (!qrc) qrc = quick_build(ctx, pl->x.data(), pl->z.data(), zcp, b, b2, &sh, s);
        if (!qrc) { pl->q1 = b1; pl->q2 = b2; pl->l1 = a; pl->l2 = b; pl->qsh = sh; pl->fused_pending = 0; }
        else { if (b1) (void)hipFree(b1); if (b2) (void)hipFree(b2); }   // repeated points, overlapping index sets, ...: the host-built path stays
    }
    ctx->err.clear();                    // "unsupported" from the device builder is not this call's error
}

// the device-built images of the small-entry kernel (fs1, fs2); failures leave the plan as it is
static void build_fused_small(hb_open_plan *pl, hipStream_t s) {
    hb_ctx *ctx = pl->ctx;
    if (pl->fs1 || env_hook(ENV_NO_QUICK_PLAN) || env_hook(ENV_NO_QUICK)) return;
    PointTable *pt = nullptr;
    FsLayout a, b;
    if (point_table(ctx, pl->x.data(), pl->n, &pt, s) == HB_OK && fs_layout(ctx, pt, pl->d, pl->n_check, 1, &a) == HB_OK &&
        fs_layout(ctx, pt, pl->d, pl->n_check, pl->d, &b) == HB_OK) {
        uint8_t *b1 = nullptr, *b2 = nullptr;
        int qrc = HB_OK;
        if (hipMalloc(&b1, a.need) != hipSuccess || hipMalloc(&b2, b.need) != hipSuccess) qrc = HB_ERR_HIP;
        const int32_t *zcp = pl->n_check > 0 ? pl->zc.data() : pl->z.data();
        if (!qrc) qrc = fs_build(ctx, pt, pl->z.data(), zcp, a, b1, FS_BUILD_Z | FS_BUILD_ZC, pl->mismatch_dev, s);
        if (!qrc) qrc = fs_build(ctx, pt, pl->z.data(), zcp, b, b2, FS_BUILD_Z | FS_BUILD_ZC, pl->mismatch_dev, s);
        if (!qrc) { pl->fs1 = b1; pl->fs2 = b2; pl->fl1 = a; pl->fl2 = b; pl->fused_pending = 0; }
        else { if (b1) (void)hipFree(b1); if (b2) (void)hipFree(b2); }
    }
    ctx->err.clear();
}

// the digit planes of the integer-VALU kernels (2 x 36 B per share): plans on the matrix cores never touch them
static int valu_planes(hb_open_plan *pl) {
    if (pl->in_dg && pl->coef_dg) return HB_OK;
    hb_ctx *ctx = pl->ctx;
    const size_t bytes = (size_t)pl->max_C * pl->d * ctx->nl() * 4;
    if (!pl->in_dg && hipMalloc(&pl->in_dg, bytes) != hipSuccess) return fail(ctx, HB_ERR_HIP, "open plan: hipMalloc(in_dg)");
    if (!pl->coef_dg && hipMalloc(&pl->coef_dg, bytes) != hipSuccess) return fail(ctx, HB_ERR_HIP, "open plan: hipMalloc(coef_dg)");
    return HB_OK;
}

// [rows of Winv ; V[zc] Winv] as int8 images.  V[zc] Winv is computed on the device (the plain mat-vec over the columns of
// Winv) and read back once; UNSUPPORTED shapes leave F1 / F2 null and the plan on its two-launch path.
static int build_fused(hb_open_plan *pl, const uint64_t *x_host, hipStream_t s) {
    hb_ctx *ctx = pl->ctx;
    const int d = pl->d, nc = pl->n_check, L = ctx->n_limbs;
    std::vector<uint64_t> W((size_t)d * d * L), P((size_t)nc * d * L);
    int rc = hb_matrix_to_host(ctx, pl->Winv, W.data(), s);
    if (rc) return rc;
    if (nc > 0) {
        std::vector<uint64_t> xzc((size_t)nc * L);
        for (int j = 0; j < nc; j++) memcpy(&xzc[(size_t)j * L], x_host + (size_t)pl->zc[j] * L, (size_t)L * 8);
        hb_matrix *Vzc = nullptr;
        uint32_t *Wd = nullptr, *Pd = nullptr;
        rc = hb_vand_matrix_create(ctx, xzc.data(), nc, d, &Vzc, s);
        if (!rc) rc = upload_elems(ctx, W.data(), (size_t)d * d, &Wd, s);
        if (!rc && hipMalloc(&Pd, P.size() * 8) != hipSuccess) rc = fail(ctx, HB_ERR_HIP, "fused validate: hipMalloc");
        // in(c, l) = Winv[l][c], out(c, i) = (V[zc] Winv)[i][c]: both row-major d-wide
        if (!rc) rc = hb_matvec(ctx, Vzc, (const uint64_t *)Wd, hb_view{1, d}, nullptr, (uint64_t *)Pd, hb_view{1, d}, d, s);
        if (!rc && hipMemcpyAsync(P.data(), Pd, P.size() * 8, hipMemcpyDeviceToHost, s) != hipSuccess) rc = fail(ctx, HB_ERR_HIP, "fused validate: copy");
        if (!rc && hipStreamSynchronize(s) != hipSuccess) rc = fail(ctx, HB_ERR_HIP, "fused validate: sync");
        if (Vzc) matrix_unref(Vzc);
        if (Wd) (void)hipFree(Wd);
        if (Pd) (void)hipFree(Pd);
        if (rc) return rc;
    }
    std::vector<uint64_t> m2((size_t)(d + nc) * d * L), m1((size_t)(1 + nc) * d * L);
    memcpy(m2.data(), W.data(), W.size() * 8);
    memcpy(m1.data(), W.data(), (size_t)d * L * 8);
    if (nc > 0) {
        memcpy(&m2[(size_t)d * d * L], P.data(), P.size() * 8);
        memcpy(&m1[(size_t)d * L], P.data(), P.size() * 8);
    }
    std::vector<int32_t> map2((size_t)d + nc + 1, 0), map1((size_t)1 + nc + 1, 0);
    for (int j = 0; j < nc; j++) { map2[(size_t)d + j] = pl->zc[j] + 1; map1[(size_t)1 + j] = pl->zc[j] + 1; }
    rc = mm8w_from_host(ctx, m2.data(), d + nc, d, &pl->F2, s);
    if (!rc) rc = mm8w_from_host(ctx, m1.data(), 1 + nc, d, &pl->F1, s);
    if (!rc) rc = own_int_array(ctx, map2.data(), (int)map2.size(), &pl->fmap2, s);
    if (!rc) rc = own_int_array(ctx, map1.data(), (int)map1.size(), &pl->fmap1, s);
    if (rc) {
        mm8w_free(pl->F1); mm8w_free(pl->F2); pl->F1 = pl->F2 = nullptr;
        if (pl->fmap1) { (void)hipFree(pl->fmap1); pl->fmap1 = nullptr; }
        if (pl->fmap2) { (void)hipFree(pl->fmap2); pl->fmap2 = nullptr; }
        return rc == HB_ERR_UNSUPPORTED ? HB_OK : rc;
    }
    return HB_OK;
}

// builds the full inverse (small-entry plans have only the factored one) and the fused matrices; shapes the full-size kernel
// does not take leave the plan as it is
static int ensure_fused(hb_open_plan *pl, hipStream_t s) {
    pl->fused_pending = 0;
    if (pl->q1 || pl->F1 || pl->d < 4 || pl->n < 4 || env_hook(ENV_NO_MFMA_DECODE)) return HB_OK;
    hb_ctx *ctx = pl->ctx;
    const int L = ctx->n_limbs;
    if (!pl->Winv) {
        std::vector<uint64_t> xz((size_t)pl->d * L);
        for (int i = 0; i < pl->d; i++) memcpy(&xz[(size_t)i * L], &pl->x[(size_t)pl->z[i] * L], (size_t)L * 8);
        int rc = hb_vand_inverse_create(ctx, xz.data(), pl->d, &pl->Winv, (void *)s);
        if (rc) return rc;
        pl->Winv8 = matrix_wide(ctx, pl->Winv, s);
    }
    return pl->Winv8 ? build_fused(pl, pl->x.data(), s) : HB_OK;
}

extern "C" {

int hb_open_plan_create(hb_ctx *ctx, int n, int d, int use_omega_powers, const uint64_t *x_host,
                        const uint64_t *omega_host, int order, const int32_t *z_host, const int32_t *zc_host,
                        int n_check, int64_t max_B, hb_open_plan **out, void *stream) { HB_API_GUARD(ctx);
    if (!ctx || !out || n <= 0 || d <= 0 || d > n || !x_host || !z_host || max_B < 0) return HB_ERR_BAD_ARG;
    if (n_check < 0 || n_check > n || (n_check > 0 && !zc_host)) return HB_ERR_BAD_ARG;
    for (int i = 0; i < d; i++) if (z_host[i] < 0 || z_host[i] >= n) return HB_ERR_BAD_ARG;
    for (int j = 0; j < n_check; j++) if (zc_host[j] < 0 || zc_host[j] >= n) return HB_ERR_BAD_ARG;
    *out = nullptr;
    hipStream_t s = (hipStream_t)stream;
    cache_trim(ctx);                            // plan creation is the busiest user of the table caches: bound them here too
    hb_open_plan *pl = new hb_open_plan();      // value-initialised: every pointer null, so destroy is safe at any point
    pl->ctx = ctx; pl->n = n; pl->d = d; pl->n_check = n_check; pl->max_B = max_B;
    pl->max_C = (max_B + d - 1) / d; if (pl->max_C < 1) pl->max_C = 1;
    pl->use_v8 = 1;
    pl->use_fused = 1;
    const int L = ctx->n_limbs;
    uint32_t *xd = nullptr, *xzd = nullptr;
    int rc = HB_OK;
    // every failure leaves through `done`, which releases whatever the plan owns by then
#define PLAN_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { ctx->err = std::string(#call) + ": " + hipGetErrorString(e__); rc = HB_ERR_HIP; goto done; } } while (0)
    {
        std::vector<uint64_t> xz((size_t)d * L);
        for (int i = 0; i < d; i++) memcpy(&xz[(size_t)i * L], x_host + (size_t)z_host[i] * L, (size_t)L * 8);
        rc = upload_elems(ctx, x_host, (size_t)n, &xd, s); if (rc) goto done;
        rc = upload_elems(ctx, xz.data(), (size_t)d, &xzd, s); if (rc) goto done;
        rc = fast_vand_create(ctx, xd, n, d, &pl->V, s);
        if (!rc) rc = fast_vinv_create(ctx, xzd, d, &pl->Vinv, s);
        (void)hipStreamSynchronize(s);
        if (rc) goto done;
    }
    rc = own_int_array(ctx, z_host, d, &pl->z_dev, s); if (rc) goto done;
    pl->z.assign(z_host, z_host + d);
    pl->x.assign(x_host, x_host + (size_t)n * L);
    {
        std::vector<int32_t> mask((size_t)n + 1, 0);
        mask[n] = -1;
        for (int j = 0; j < n_check; j++) { mask[zc_host[j]] = 1; pl->zc.push_back(zc_host[j]); }
        rc = own_int_array(ctx, mask.data(), n + 1, &pl->mask_dev, s); if (rc) goto done;
    }
    // (in_dg / coef_dg, the digit planes of the integer-VALU kernels, are allocated by the first launch that needs them: valu_planes)
    if (use_omega_powers && omega_host && order >= n && order > 0 && (order & (order - 1)) == 0) {
        // MADs per chunk: full-digit mat-vec n*d*NL^2 vs butterflies (order/2)*log2(order)*(2 NL^2 + 4 NL)
        int logn = 0; while ((1 << logn) < order) logn++;
        const double nl2 = (double)ctx->nl() * ctx->nl();
        const double mv = (double)n * d * nl2, ntt = 0.5 * order * logn * (2.0 * nl2 + 4.0 * ctx->nl());
        if (ntt < mv && (size_t)order * ctx->nl() * 4 <= 40 * 1024) {
            rc = get_twiddles(ctx, omega_host, order, &pl->tw, s); if (rc) goto done;     // pinned in the ctx cache
            pl->ntt_order = order;
            PLAN_HIP(hipMalloc(&pl->coef_pk, (size_t)pl->max_C * d * ctx->elem_words() * 4));
        }
    }
    if (!pl->ntt_order) {
        // third generation: the encode and the validating re-encode on the int8 matrix cores when the
        // Vandermonde entries are small enough (hb_mfma.hip); otherwise V8 stays null
        rc = mm8_from_fast(ctx, pl->V, &pl->V8, s);
        if (rc && rc != HB_ERR_UNSUPPORTED) goto done;
        rc = HB_OK;
        if (pl->V8) {
            PLAN_HIP(hipMalloc(&pl->coef_pk, (size_t)pl->max_C * d * ctx->elem_words() * 4));
            rc = env_hook(ENV_NO_MFMA_DECODE) ? HB_ERR_UNSUPPORTED : mm8_from_fast(ctx, pl->Vinv, &pl->Vinv8, s);
            if (rc && rc != HB_ERR_UNSUPPORTED) goto done;
            rc = HB_OK;
            if (pl->Vinv8) PLAN_HIP(hipMalloc(&pl->scaled_pk, (size_t)pl->max_C * d * ctx->elem_words() * 4));
        }
    }
    // Full tables on the full-size matrix-core kernel for plans whose entries are not small integers: the decode, and the encodes
    // unless they are NTTs.  (Every plan from 4 coefficients up also decodes + validates there in ONE launch over
    // [V^-1 rows ; V[zc] V^-1] once it has decoded twice -- ensure_fused; it beats pre-scale + decode + validating re-encode on
    // k_mm8 at every size: n = 64, t = 21: 0.127 against 0.200 ms for the two decodes, n = 8, t = 3: 0.170 against 0.180;
    // scratch/fused_vs_default.py.)
    if (!pl->V8 && d >= 4 && n >= 4 && !env_hook(ENV_NO_MFMA_DECODE)) {
        std::vector<uint64_t> xz((size_t)d * L);
        for (int i = 0; i < d; i++) memcpy(&xz[(size_t)i * L], x_host + (size_t)z_host[i] * L, (size_t)L * 8);
        rc = hb_vand_inverse_create(ctx, xz.data(), d, &pl->Winv, stream); if (rc) goto done;
        pl->Winv8 = matrix_wide(ctx, pl->Winv, s);
        if (!pl->ntt_order) {
            rc = hb_vand_matrix_create(ctx, x_host, n, d, &pl->Vw, stream); if (rc) goto done;
            pl->Vw8 = matrix_wide(ctx, pl->Vw, s);
        }
        if ((pl->Winv8 || pl->Vw8) && !pl->coef_pk) PLAN_HIP(hipMalloc(&pl->coef_pk, (size_t)pl->max_C * d * ctx->elem_words() * 4));
    }
    // (plans at omega powers keep their NTT encode on the integer kernel; on the matrix cores an entry is full-size whatever the points are --
    //  M 2^96 mod p -- and the mat-vec kernel is the faster of the two)
    if (mv64_applies(ctx, d) && (!use_omega_powers || mv64_matrix_cores(ctx, d))) {
        std::vector<uint64_t> Vh, Vi, Ph;
        rc = mv64_plan_tables(ctx, x_host, n, d, z_host, zc_host, n_check, Vh, Vi, Ph);
        if (rc) goto done;
        std::vector<int32_t> me((size_t)n), m1((size_t)1 + n_check), m2((size_t)d + n_check);
        for (int i = 0; i < n; i++) me[i] = -(i + 1);
        std::vector<uint64_t> f1((size_t)(1 + n_check) * d), f2((size_t)(d + n_check) * d);
        memcpy(f1.data(), Vi.data(), (size_t)d * 8);
        memcpy(f2.data(), Vi.data(), (size_t)d * d * 8);
        if (n_check > 0) { memcpy(&f1[(size_t)d], Ph.data(), Ph.size() * 8); memcpy(&f2[(size_t)d * d], Ph.data(), Ph.size() * 8); }
        m1[0] = -1;
        for (int i = 0; i < d; i++) m2[i] = -(i + 1);
        for (int j = 0; j < n_check; j++) { m1[(size_t)1 + j] = zc_host[j] + 1; m2[(size_t)d + j] = zc_host[j] + 1; }
        rc = mv64_from_host(ctx, Vh.data(), n, d, me.data(), &pl->nv_enc, s);
        if (!rc) rc = mv64_from_host(ctx, f1.data(), 1 + n_check, d, m1.data(), &pl->nv_f1, s);
        if (!rc) rc = mv64_from_host(ctx, f2.data(), d + n_check, d, m2.data(), &pl->nv_f2, s);
        if (rc == HB_ERR_UNSUPPORTED) rc = HB_OK;
        if (rc) goto done;
    }
    // the fused matrices: built on the device right now where hb_quick.hip takes the shape; otherwise on the host when the plan decodes
    // for the third time (ensure_fused), or at once on request (set_option)
    pl->fused_pending = ((pl->V8 || pl->Winv8) && d >= 4 && n >= 4 && !env_hook(ENV_NO_MFMA_DECODE) && !env_hook(ENV_NO_FUSED_VALIDATE) &&
                         !env_hook(ENV_NO_MFMA_WIDE) && ctx->n_limbs == 4 && prescale_params(ctx)) ? 1 : 0;
    PLAN_HIP(hipMalloc(&pl->mismatch_dev, sizeof(int32_t)));
    PLAN_HIP(hipMemsetAsync(pl->mismatch_dev, 0, sizeof(int32_t), s));
    if (pl->fused_pending) {
        build_fused_small(pl, s);
        if (!pl->fs1) build_quick_wide(pl, s);
    }
#undef PLAN_HIP
done:
    if (xd) (void)hipFree(xd);
    if (xzd) (void)hipFree(xzd);
    if (rc) { hb_open_plan_destroy(pl); return rc; }
    *out = pl;
    return HB_OK;
}

// R1: shares [B] (chunk c = shares[c*d .. c*d+d), zero padded) -> r1_out [n][C]
int hb_open_r1_encode(hb_open_plan *pl, const uint64_t *shares_dev, int64_t B, uint64_t *r1_out_dev, void *stream) { HB_API_GUARD((pl ? pl->ctx : nullptr));
    if (!pl || B < 0 || B > pl->max_B) return HB_ERR_BAD_ARG;
    const int64_t C = (B + pl->d - 1) / pl->d;
    hb_view iv{pl->d, 1}, ov{1, C};
    hipStream_t s = (hipStream_t)stream;
    if (pl->nv_enc && pl->use_v8)
        return launch_mv64(pl->ctx, pl->nv_enc, shares_dev, iv, nullptr, B, r1_out_dev, ov, INT64_MAX, nullptr, nullptr, C, s);
    if (pl->ntt_order)
        return launch_ntt_lds(pl->ctx, pl->tw, pl->ntt_order, (const uint32_t *)shares_dev, iv, B, pl->d, pl->n,
                              (uint32_t *)r1_out_dev, ov, INT64_MAX, nullptr, nullptr, C, s);
    if (pl->V8 && pl->use_v8)
        return launch_mm8(pl->ctx, pl->V8, (const uint32_t *)shares_dev, iv, nullptr, B, (uint32_t *)r1_out_dev, ov, INT64_MAX,
                          nullptr, nullptr, C, s);
    if (pl->Vw8 && pl->use_v8)
        return launch_mm8w(pl->ctx, pl->Vw8, (const uint32_t *)shares_dev, iv, nullptr, B, (uint32_t *)r1_out_dev, ov, INT64_MAX,
                           nullptr, nullptr, C, s);
    if (int rc = valu_planes(pl)) return rc;
    return launch_matvec2(pl->ctx, pl->V, nullptr, (const uint32_t *)shares_dev, iv, nullptr, B, pl->in_dg,
                          (uint32_t *)r1_out_dev, ov, INT64_MAX, pl->n, 0, nullptr, nullptr, nullptr, C, s);
}

// canonical coefficients of rows < pk_rows go to pk_dst (view pv); Montgomery planes always to coef_dg
static int decode_and_validate(hb_open_plan *pl, const uint64_t *cols_dev, int64_t C, uint32_t *pk_dst, hb_view pv, int64_t pk_count,
                               int pk_rows, hipStream_t s) {
    hb_view pm{1, C};
    if (pl->nv_f1 && pl->nv_f2 && pl->use_v8 && pl->use_fused && (pk_rows == 1 || pk_rows == pl->d)) {
        // word-size prime: [V^-1 rows ; V[zc] V^-1] over the received columns, one launch of the 8-byte mat-vec
        const bool r1 = pk_rows == 1 && pl->d > 1;
        return launch_mv64(pl->ctx, r1 ? pl->nv_f1 : pl->nv_f2, cols_dev, pm, pl->z_dev, INT64_MAX, (uint64_t *)pk_dst, pv, pk_count, pl->mismatch_dev, nullptr, C, s);
    }
    if (pl->fused_pending && pl->use_v8 && pl->use_fused && ++pl->decode_calls > 2) {
        // Building the fused matrices is an optimisation of a plan that already decodes correctly on the two-launch path: if
        // it fails for any reason (out of memory while building F1 / F2, ...) the decode goes on unfused and the error is dropped.
        // ensure_fused clears fused_pending first, so a failure is not retried on every call.
        if (ensure_fused(pl, s) != HB_OK) pl->ctx->err.clear();
    }
    if (pl->fs1 && pl->fs2 && pl->use_v8 && pl->use_fused == 1 && (pk_rows == 1 || pk_rows == pl->d)) {
        // small-integer points: [N ; P] over the received columns, divided by den_j inside the kernel
        const bool r1 = pk_rows == 1 && pl->d > 1;
        return fs_launch(pl->ctx, r1 ? pl->fl1 : pl->fl2, r1 ? pl->fs1 : pl->fs2, (const uint32_t *)cols_dev, pm, pk_dst, pv, pk_count, pl->mismatch_dev,
                         nullptr, nullptr, C, s);
    }
    if (pl->q1 && pl->q2 && pl->use_v8 && pl->use_fused && (pk_rows == 1 || pk_rows == pl->d)) {
        // device-built images: ONE launch decodes the rows the caller wants and compares the predictions of the later arrivals
        const bool r1 = pk_rows == 1 && pl->d > 1;
        return quick_launch(pl->ctx, r1 ? pl->l1 : pl->l2, r1 ? pl->q1 : pl->q2, pl->qsh, (const uint32_t *)cols_dev, pm, pk_dst, pv, pk_count, pk_rows,
                            pl->mismatch_dev, nullptr, C, s);
    }
    if (pl->F1 && pl->F2 && pl->use_v8 && pl->use_fused && (pk_rows == 1 || pk_rows == pl->d)) {
        // full-size entries: ONE launch decodes the rows the caller wants and compares the predictions of the later arrivals
        const bool r1 = pk_rows == 1 && pl->d > 1;
        return launch_mm8w(pl->ctx, r1 ? pl->F1 : pl->F2, (const uint32_t *)cols_dev, pm, pl->z_dev, INT64_MAX, pk_dst, pv, pk_count,
                           r1 ? pl->fmap1 : pl->fmap2, pl->mismatch_dev, C, s, (const uint32_t *)cols_dev, pm, pk_rows);
    }
    if (pl->ntt_order) {
        // decode to canonical coefficient-major coefficients, validate with an NTT in CHECK mode,
        // then hand the caller the rows it asked for
        int rc = (pl->Winv8 && pl->use_v8) ? HB_OK : valu_planes(pl);
        if (rc) return rc;
        rc = pl->Winv8 && pl->use_v8
                     ? launch_mm8w(pl->ctx, pl->Winv8, (const uint32_t *)cols_dev, pm, pl->z_dev, INT64_MAX, pl->coef_pk, pm, INT64_MAX, nullptr, nullptr, C, s)
                     : launch_matvec2(pl->ctx, pl->Vinv, nullptr, (const uint32_t *)cols_dev, pm, pl->z_dev, INT64_MAX, pl->in_dg,
                                      pl->coef_pk, pm, INT64_MAX, pl->d, 1, nullptr, nullptr, nullptr, C, s);
        if (rc) return rc;
        // the NTT in CHECK mode hands the caller its rows of the coefficients as they pass through
        return launch_ntt_lds(pl->ctx, pl->tw, pl->ntt_order, pl->coef_pk, pm, INT64_MAX, pl->d, pl->n,
                              (uint32_t *)const_cast<uint64_t *>(cols_dev), pm, INT64_MAX, pl->mask_dev, pl->mismatch_dev, C, s,
                              pk_dst, pv, pk_count, pk_rows);
    }
    if (pl->V8 && pl->use_v8) {
        // decode to canonical coefficients (VALU path, d outputs), validate on the matrix cores: the
        // re-encode of all n points compared with the received columns in the kernel's epilogue; the
        // same kernel hands the caller its rows of the coefficients while they sit in LDS
        int rc;
        if (pl->Vinv8) {
            // c = N (x / den): the division as an elementwise pass, the small-integer mat-vec on the matrix cores
            rc = launch_prescale_pk(pl->ctx, pl->Vinv, (const uint32_t *)cols_dev, pm, pl->z_dev, INT64_MAX, pl->scaled_pk, C, s);
            if (rc) return rc;
            rc = launch_mm8(pl->ctx, pl->Vinv8, pl->scaled_pk, pm, nullptr, INT64_MAX, pl->coef_pk, pm, INT64_MAX, nullptr, nullptr, C, s);
        } else {
            rc = valu_planes(pl);
            if (rc) return rc;
            rc = launch_matvec2(pl->ctx, pl->Vinv, nullptr, (const uint32_t *)cols_dev, pm, pl->z_dev, INT64_MAX, pl->in_dg,
                                pl->coef_pk, pm, INT64_MAX, pl->d, 0, nullptr, nullptr, nullptr, C, s, 0, pl->Vinv->K2);
        }
        if (rc) return rc;
        if (pl->validate_arrived_only && pl->Vzc8)
            // option: evaluate the guess at the compared points only (a compact matrix of the rows zc of V)
            return launch_mm8(pl->ctx, pl->Vzc8, pl->coef_pk, pm, nullptr, INT64_MAX, (uint32_t *)const_cast<uint64_t *>(cols_dev), pm,
                              INT64_MAX, pl->ones_dev, pl->mismatch_dev, C, s, pk_dst, pv, pk_count, pk_rows, pl->zc_dev);
        return launch_mm8(pl->ctx, pl->V8, pl->coef_pk, pm, nullptr, INT64_MAX, (uint32_t *)const_cast<uint64_t *>(cols_dev), pm,
                          INT64_MAX, pl->mask_dev, pl->mismatch_dev, C, s, pk_dst, pv, pk_count, pk_rows);
    }
    if (pl->Vw8 && pl->Winv8 && pl->use_v8) {
        // full-size entries, mat-vec encodes: decode, validating re-encode of all n points compared in the epilogue, hand-off
        int rc = launch_mm8w(pl->ctx, pl->Winv8, (const uint32_t *)cols_dev, pm, pl->z_dev, INT64_MAX, pl->coef_pk, pm, INT64_MAX, nullptr, nullptr, C, s);
        if (rc) return rc;
        rc = launch_mm8w(pl->ctx, pl->Vw8, pl->coef_pk, pm, nullptr, INT64_MAX, (uint32_t *)const_cast<uint64_t *>(cols_dev), pm, INT64_MAX,
                         pl->mask_dev, pl->mismatch_dev, C, s);
        if (rc) return rc;
        return launch_copy_view(pl->ctx, pl->coef_pk, pm, pk_dst, pv, C, pk_rows, pk_count, s);
    }
    // one launch when the shapes allow it (decode + validating re-encode of the same 64-chunk group)
    int rc = valu_planes(pl);
    if (rc) return rc;
    rc = launch_decode_check(pl->ctx, pl->Vinv, pl->V, (const uint32_t *)cols_dev, pm, pl->z_dev, pk_dst, pv, pk_count, pk_rows,
                                 pl->coef_dg, pl->mask_dev, pl->mismatch_dev, C, s, pl->validate_arrived_only);
    if (rc != HB_ERR_UNSUPPORTED) return rc;
    rc = launch_matvec2(pl->ctx, pl->Vinv, nullptr, (const uint32_t *)cols_dev, pm, pl->z_dev, INT64_MAX, pl->in_dg,
                        pk_dst, pv, pk_count, pk_rows, 1, pl->coef_dg, nullptr, nullptr, C, s);
    if (rc) return rc;
    // validating re-encode of the guess, compared in the epilogue against the later arrivals
    hb_view none{0, 0};
    return launch_matvec2(pl->ctx, pl->V, pl->coef_dg, nullptr, none, nullptr, 0, nullptr,
                          (uint32_t *)const_cast<uint64_t *>(cols_dev), pm, INT64_MAX, 0, 0, nullptr,
                          pl->mask_dev, pl->mismatch_dev, C, s, pl->validate_arrived_only);
}

int hb_open_r1_decode(hb_open_plan *pl, const uint64_t *r1_cols_dev, int64_t B, uint64_t *r2_msg_dev, void *stream) { HB_API_GUARD((pl ? pl->ctx : nullptr));
    if (!pl || B < 0 || B > pl->max_B) return HB_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t C = (B + pl->d - 1) / pl->d;
    // message = [chunk[0] for chunk in recons_r2]: only row 0 (the constant terms) is needed in
    // canonical form, and the decode kernel writes it straight into the caller's buffer
    hb_view pv{1, C};
    return decode_and_validate(pl, r1_cols_dev, C, (uint32_t *)r2_msg_dev, pv, INT64_MAX, 1, s);
}

int hb_open_r2_decode(hb_open_plan *pl, const uint64_t *r2_cols_dev, int64_t B, uint64_t *result_dev, void *stream) { HB_API_GUARD((pl ? pl->ctx : nullptr));
    if (!pl || B < 0 || B > pl->max_B) return HB_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t C = (B + pl->d - 1) / pl->d;
    // flatten_lists(recons_p)[:B]: the decode kernel stores chunk-major, truncated at B
    hb_view dv{pl->d, 1};
    return decode_and_validate(pl, r2_cols_dev, C, (uint32_t *)result_dev, dv, B, pl->d, s);
}

int hb_open_status(hb_open_plan *pl, void *stream) { HB_API_GUARD((pl ? pl->ctx : nullptr));
    if (!pl) return HB_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    int32_t flag = 0;
    HB_HIP(pl->ctx, hipMemcpyAsync(&flag, pl->mismatch_dev, sizeof flag, hipMemcpyDeviceToHost, s));
    HB_HIP(pl->ctx, hipStreamSynchronize(s));
    if (flag) {
        HB_HIP(pl->ctx, hipMemsetAsync(pl->mismatch_dev, 0, sizeof(int32_t), s));
        return fail(pl->ctx, HB_ERR_MISMATCH, "Optimistic decoding failed");   // reed_solomon.py:323
    }
    return HB_OK;
}

int hb_open_plan_set_option(hb_open_plan *pl, int option, int value) { HB_API_GUARD((pl ? pl->ctx : nullptr));
    if (!pl) return HB_ERR_BAD_ARG;
    if (option == HB_OPEN_OPT_VALIDATE_ARRIVED_ONLY) {
        pl->validate_arrived_only = value ? 1 : 0;
        if (value && pl->V8 && !pl->Vzc8 && !pl->zc.empty()) {
            // matrix-core path: a compact check matrix (the rows zc of V) compared with the received rows zc
            int rc = mm8_from_fast(pl->ctx, pl->V, &pl->Vzc8, 0, pl->zc.data(), (int)pl->zc.size());
            if (rc && rc != HB_ERR_UNSUPPORTED) return rc;
            if (pl->Vzc8) {
                std::vector<int32_t> ones(pl->zc.size() + 2, 1);
                rc = own_int_array(pl->ctx, pl->zc.data(), (int)pl->zc.size(), &pl->zc_dev, 0);
                if (!rc) rc = own_int_array(pl->ctx, ones.data(), (int)ones.size(), &pl->ones_dev, 0);
                if (rc) return rc;
            }
        }
        return HB_OK;
    }
    if (option == HB_OPEN_OPT_MATRIX_CORES) { pl->use_v8 = value ? 1 : 0; return HB_OK; }
    if (option == HB_OPEN_OPT_FUSED_VALIDATE) {
        pl->use_fused = value == 2 ? 2 : (value ? 1 : 0);
        if (value == 2 && pl->fs1 && !pl->q1) build_quick_wide(pl, 0);        // the full-size kernel where the small-entry one is the default
        if (value && !pl->F1 && !pl->q1 && !(pl->fs1 && value == 1)) {
            // asked for explicitly: built now instead of at the third decode
            int rc = ensure_fused(pl, 0);
            if (rc) return rc;
        }
        return HB_OK;
    }
    return HB_ERR_BAD_ARG;
}

int hb_open_plan_get_option(hb_open_plan *pl, int option, int *value) { HB_API_GUARD((pl ? pl->ctx : nullptr));
    if (!pl || !value) return HB_ERR_BAD_ARG;
    if (option == HB_OPEN_OPT_VALIDATE_ARRIVED_ONLY) { *value = pl->validate_arrived_only; return HB_OK; }
    if (option == HB_OPEN_OPT_MATRIX_CORES) { *value = ((pl->V8 || pl->Winv8 || pl->Vw8) && pl->use_v8) ? 1 : 0; return HB_OK; }
    if (option == HB_OPEN_OPT_FUSED_VALIDATE) { *value = (((pl->F1 && pl->F2) || (pl->q1 && pl->q2) || (pl->nv_f1 && pl->nv_f2) || pl->fused_pending) && pl->use_v8 && pl->use_fused) ? 1 : 0;
        if (pl->fs1 && pl->fs2 && pl->use_v8 && pl->use_fused == 1) *value = 3;          // ... on the small-entry kernel
        return HB_OK; }
    return HB_ERR_BAD_ARG;
}

void hb_open_plan_destroy(hb_open_plan *pl) { HB_API_GUARD((pl ? pl->ctx : nullptr));
    if (!pl) return;
    if (pl->in_dg) (void)hipFree(pl->in_dg);
    if (pl->coef_dg) (void)hipFree(pl->coef_dg);
    if (pl->coef_pk) (void)hipFree(pl->coef_pk);
    if (pl->mismatch_dev) (void)hipFree(pl->mismatch_dev);
    if (pl->z_dev) (void)hipFree(pl->z_dev);
    if (pl->mask_dev) (void)hipFree(pl->mask_dev);
    if (pl->zc_dev) (void)hipFree(pl->zc_dev);
    if (pl->ones_dev) (void)hipFree(pl->ones_dev);
    fast_matrix_free(pl->V); fast_matrix_free(pl->Vinv); mm8_free(pl->V8); mm8_free(pl->Vinv8); mm8_free(pl->Vzc8);
    if (pl->scaled_pk) (void)hipFree(pl->scaled_pk);
    mm8w_free(pl->F1); mm8w_free(pl->F2);
    if (pl->q1) (void)hipFree(pl->q1);
    if (pl->q2) (void)hipFree(pl->q2);
    if (pl->fs1) (void)hipFree(pl->fs1);
    if (pl->fs2) (void)hipFree(pl->fs2);
    if (pl->fmap1) (void)hipFree(pl->fmap1);
    if (pl->fmap2) (void)hipFree(pl->fmap2);
    if (pl->Winv) matrix_unref(pl->Winv);
    if (pl->Vw) matrix_unref(pl->Vw);
    mv64_free(pl->nv_enc); mv64_free(pl->nv_f1); mv64_free(pl->nv_f2);
    delete pl;
}

}  // extern "C"
