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

struct hb_open_plan {
    hb_ctx *ctx;
    int n, d, n_check;
    int64_t max_B, max_C;
    hb_matrix *V;        // n x d  encode matrix at the n party points
    hb_matrix *Vinv;     // d x d  decode matrix for the arrival set z
    int32_t *z_dev;      // d row indices
    int32_t *mask_dev;   // n+1 ints: rows to validate
    uint32_t *coef;      // [d][max_C] decoded coefficients
    int32_t *mismatch_dev;
};

extern "C" {

int hb_open_plan_create(hb_ctx *ctx, int n, int d, int use_omega_powers, const uint64_t *x_host,
                        const uint64_t *omega_host, int order, const int32_t *z_host, const int32_t *zc_host,
                        int n_check, int64_t max_B, hb_open_plan **out, void *stream) {
    (void)use_omega_powers; (void)omega_host; (void)order;   // the points x already carry the policy
    if (!ctx || !out || n <= 0 || d <= 0 || d > n || !x_host || !z_host || max_B < 0) return HB_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    hb_open_plan *pl = new hb_open_plan();
    pl->ctx = ctx; pl->n = n; pl->d = d; pl->n_check = n_check; pl->max_B = max_B;
    pl->max_C = (max_B + d - 1) / d; if (pl->max_C < 1) pl->max_C = 1;
    pl->coef = nullptr; pl->mismatch_dev = nullptr;
    int rc = hb_vand_matrix_create(ctx, x_host, n, d, &pl->V, stream);
    if (rc) { delete pl; return rc; }
    const int L = ctx->n_limbs;
    std::vector<uint64_t> xz((size_t)d * L);
    for (int i = 0; i < d; i++) {
        if (z_host[i] < 0 || z_host[i] >= n) { delete pl; return HB_ERR_BAD_ARG; }
        memcpy(&xz[(size_t)i * L], x_host + (size_t)z_host[i] * L, (size_t)L * 8);
    }
    rc = hb_vand_inverse_create(ctx, xz.data(), d, &pl->Vinv, stream);
    if (rc) { delete pl; return rc; }
    rc = get_int_array(ctx, z_host, d, &pl->z_dev, s);
    if (rc) { delete pl; return rc; }
    std::vector<int32_t> mask((size_t)n + 1, 0);
    mask[n] = -1;
    for (int j = 0; j < n_check; j++) {
        if (zc_host[j] < 0 || zc_host[j] >= n) { delete pl; return HB_ERR_BAD_ARG; }
        mask[zc_host[j]] = 1;
    }
    rc = get_int_array(ctx, mask.data(), n + 1, &pl->mask_dev, s);
    if (rc) { delete pl; return rc; }
    HB_HIP(ctx, hipMalloc(&pl->coef, (size_t)pl->max_C * d * ctx->elem_words() * 4));
    HB_HIP(ctx, hipMalloc(&pl->mismatch_dev, sizeof(int32_t)));
    HB_HIP(ctx, hipMemsetAsync(pl->mismatch_dev, 0, sizeof(int32_t), s));
    *out = pl;
    return HB_OK;
}

// R1: shares [B] (chunk c = shares[c*d .. c*d+d), zero padded) -> r1_out [n][C]
int hb_open_r1_encode(hb_open_plan *pl, const uint64_t *shares_dev, int64_t B, uint64_t *r1_out_dev, void *stream) {
    if (!pl || B < 0 || B > pl->max_B) return HB_ERR_BAD_ARG;
    const int64_t C = (B + pl->d - 1) / pl->d;
    hb_view iv{pl->d, 1}, ov{1, C};
    return launch_matvec(pl->ctx, pl->V, (const uint32_t *)shares_dev, iv, nullptr, B, (uint32_t *)r1_out_dev, ov, INT64_MAX,
                         nullptr, nullptr, C, (hipStream_t)stream);
}

static int decode_and_validate(hb_open_plan *pl, const uint64_t *cols_dev, int64_t C, hipStream_t s) {
    hb_view pm{1, C};
    int rc = launch_matvec(pl->ctx, pl->Vinv, (const uint32_t *)cols_dev, pm, pl->z_dev, INT64_MAX, pl->coef, pm, INT64_MAX,
                           nullptr, nullptr, C, s);
    if (rc) return rc;
    // validating re-encode of the guess, compared in the epilogue against the later arrivals
    return launch_matvec(pl->ctx, pl->V, pl->coef, pm, nullptr, INT64_MAX, (uint32_t *)const_cast<uint64_t *>(cols_dev), pm, INT64_MAX,
                         pl->mask_dev, pl->mismatch_dev, C, s);
}

int hb_open_r1_decode(hb_open_plan *pl, const uint64_t *r1_cols_dev, int64_t B, uint64_t *r2_msg_dev, void *stream) {
    if (!pl || B < 0 || B > pl->max_B) return HB_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t C = (B + pl->d - 1) / pl->d;
    int rc = decode_and_validate(pl, r1_cols_dev, C, s);
    if (rc) return rc;
    // message = [chunk[0] for chunk in recons_r2]  == row 0 of the coefficient-major buffer
    HB_HIP(pl->ctx, hipMemcpyAsync(r2_msg_dev, pl->coef, (size_t)C * pl->ctx->elem_words() * 4, hipMemcpyDeviceToDevice, s));
    return HB_OK;
}

int hb_open_r2_decode(hb_open_plan *pl, const uint64_t *r2_cols_dev, int64_t B, uint64_t *result_dev, void *stream) {
    if (!pl || B < 0 || B > pl->max_B) return HB_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int64_t C = (B + pl->d - 1) / pl->d;
    int rc = decode_and_validate(pl, r2_cols_dev, C, s);
    if (rc) return rc;
    // flatten_lists(recons_p)[:B]: coefficient-major -> chunk-major, truncated
    hb_view sv{1, C}, dv{pl->d, 1};
    return launch_copy_view(pl->ctx, pl->coef, sv, (uint32_t *)result_dev, dv, C, pl->d, B, s);
}

int hb_open_status(hb_open_plan *pl, void *stream) {
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

void hb_open_plan_destroy(hb_open_plan *pl) {
    if (!pl) return;
    (void)hipFree(pl->coef);
    (void)hipFree(pl->mismatch_dev);
    delete pl;
}

}  // extern "C"
