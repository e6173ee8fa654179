// hb_dec.hip -- the optimistic phase of ONE IncrementalDecoder as an object behind the C ABI.
//
// Reference: honeybadgermpc/reed_solomon.py:232-403.  IncrementalDecoder.add(idx, data) (:367-403) ignores a sender it has already
// counted or that is a confirmed error (:369-372), collects columns until degree + 1 are in, decodes the guess from those and
// re-encodes it (:305-313), compares every later column with its row of the guess (:316-326) and is done once
// degree + 1 + max_errors - |confirmed errors| columns agree (:302-303, :328-330); the first disagreement switches to the robust
// decoder for good (:334-365).  batch_reconstruct drives one such object per round, one add() per message (batch_reconstruction.py:43-61).
//
// Rounds 1-4 kept that state machine in Python (device.py) over hb_quick_dec_*: per open ~86 add() calls of set / list bookkeeping,
// pooled helper objects borrowed and returned, two C calls with freshly built index arrays and a tensor allocated behind the last column.
// Here the host keeps nothing: the transport announces "column idx has landed in row idx of the party-major buffer" (hb_dec_arrived1,
// or bursts through hb_dec_arrived); the object counts arrivals, enqueues what depends on the first degree + 1 of them when the last
// of those lands (hb_quick_dec_arrivals: [N ; T] and a candidate row per party), launches decode + validate behind the column that
// completes the quorum and waits for the verdict in pinned memory (hb_quick_dec_decide).  Until that column nothing of the reference's
// state is observable (a guess nobody has compared yet), so nothing is computed.  What the host gets back is a state:
//
//   HB_DEC_COLLECTING   more columns needed
//   HB_DEC_DONE         every compared column agreed: the coefficients (or the constant terms) are in the caller's buffer
//   HB_DEC_DISAGREE     some compared column differs from the guess: the robust phase takes over from the arrival list
//                       (hb_dec_arrivals_list); with all coefficients asked for, the refuted guess is in the caller's buffer and
//                       hb_dec_verdict names the first disagreeing chunk (everything before it is settled)
//   HB_DEC_UNSUPPORTED  this shape / point set / modulus is outside the plan-free kernels (found when the first degree + 1 arrivals
//                       were known): the caller decodes the arrival list some other way
//   HB_DEC_PENDING      (only with HB_DEC_OPT_DEFER) the quorum is complete and decode + validate is enqueued, the verdict not read
//                       yet: hb_dec_settle waits for it.  The caller's thread is free meanwhile -- batch_reconstruct makes the next
//                       round's decoder and takes its first messages while this round's launch runs (batch_reconstruction.py:158-227
//                       subscribes to both rounds up front for the same reason)
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include <vector>

#include "hb_common.hpp"

using namespace hb;

struct hb_dec {
    hb_ctx *ctx;
    hb_quick_dec *qd;
    int n, degree, max_errors;
    // the round in progress
    const uint64_t *cols;
    int64_t C;
    uint64_t *coeffs;
    int n_coef, need, nc;
    void *stream;
    std::vector<uint8_t> seen;          // 1: counted, 2: excluded when the round began (a confirmed error)
    std::vector<int32_t> z;             // arrival order
    int state;
    int32_t first;
    bool begun;
    bool deferred;                      // the quorum's arrival returns HB_DEC_PENDING instead of waiting for the verdict
};

extern "C" {

int hb_dec_create(hb_ctx *ctx, const uint64_t *x_host, int n, int degree, int max_errors, hb_dec **out, void *stream) {
    if (!ctx || !x_host || !out || n < 1 || degree < 0 || degree >= n || max_errors < 0) return HB_ERR_BAD_ARG;
    *out = nullptr;
    hb_quick_dec *qd = nullptr;
    const int rc = hb_quick_dec_create(ctx, x_host, n, &qd, stream);
    if (rc) return rc;
    hb_dec *dec = new hb_dec();
    dec->ctx = ctx; dec->qd = qd; dec->n = n; dec->degree = degree; dec->max_errors = max_errors;
    dec->cols = nullptr; dec->C = 0; dec->coeffs = nullptr; dec->n_coef = 0; dec->need = 0; dec->nc = 0; dec->stream = nullptr;
    dec->seen.assign((size_t)n, 0);
    dec->z.reserve((size_t)n);
    dec->state = HB_DEC_COLLECTING; dec->first = INT32_MAX; dec->begun = false; dec->deferred = false;
    *out = dec;
    return HB_OK;
}

int hb_dec_begin(hb_dec *dec, const uint64_t *cols_dev, int64_t C, int n_coef, uint64_t *coeffs_dev, const int32_t *excluded, int n_excluded, void *stream) {
    if (!dec) return HB_ERR_BAD_ARG;
    const int d = dec->degree + 1;
    // (a round abandoned after its (degree + 1)-th arrival has a build enqueued: every build of this object runs on the quick decoder's own
    // stream, in order, and a decode launch nobody waited for is waited for by the next build -- hb_quick.hip)
    dec->begun = false;                       // (a begin that fails leaves no round in progress)
    if (!cols_dev || !coeffs_dev || C < 1 || n_excluded < 0 || (n_excluded > 0 && !excluded)) return HB_ERR_BAD_ARG;
    if (n_coef != 1 && n_coef != d) return HB_ERR_BAD_ARG;
    std::fill(dec->seen.begin(), dec->seen.end(), (uint8_t)0);
    int ex = 0;
    for (int i = 0; i < n_excluded; i++) {
        if (excluded[i] < 0 || excluded[i] >= dec->n) return HB_ERR_BAD_ARG;
        if (!dec->seen[excluded[i]]) { dec->seen[excluded[i]] = 2; ex++; }
    }
    // reed_solomon.py:302-303: degree + 1 + max_errors - |confirmed errors| agreeing columns finish
    const int need = d + dec->max_errors - ex;
    const int nc = need - d;
    if (nc < 1) return fail(dec->ctx, HB_ERR_UNSUPPORTED, "decoder: nothing to compare the guess with (the caller's own path decides at the (degree+1)-th column)");
    const int rc = quick_dec_supported(dec->qd, d, nc, n_coef);
    if (rc) return rc;
    dec->cols = cols_dev; dec->C = C; dec->coeffs = coeffs_dev; dec->n_coef = n_coef; dec->need = need; dec->nc = nc; dec->stream = stream;
    dec->z.clear();
    dec->state = HB_DEC_COLLECTING;
    dec->first = INT32_MAX;
    dec->begun = true;
    return HB_OK;
}

int hb_dec_arrived1(hb_dec *dec, int32_t idx) {
    if (!dec || !dec->begun) return -HB_ERR_BAD_ARG;
    if (dec->state != HB_DEC_COLLECTING) return dec->state;          // (done, or handed over: later arrivals are the caller's)
    if (idx < 0 || idx >= dec->n) return -HB_ERR_BAD_ARG;
    if (dec->seen[idx]) return HB_DEC_COLLECTING;                    // reed_solomon.py:369-372
    dec->seen[idx] = 1;
    dec->z.push_back(idx);
    const int k = (int)dec->z.size(), d = dec->degree + 1;
    if (k == d) {
        // what depends on the first degree + 1 arrivals alone: enqueued, nothing waited for
        const int rc = hb_quick_dec_arrivals(dec->qd, dec->z.data(), d, dec->nc, dec->n_coef, dec->stream);
        if (rc == HB_ERR_UNSUPPORTED) return dec->state = HB_DEC_UNSUPPORTED;
        if (rc) return -rc;
        return HB_DEC_COLLECTING;
    }
    if (k < dec->need) return HB_DEC_COLLECTING;
    if (dec->deferred) {
        const int rc = hb_quick_dec_launch(dec->qd, dec->z.data() + d, dec->nc, dec->cols, dec->C, 0, dec->C, dec->coeffs, dec->stream);
        if (rc) return -rc;
        return dec->state = HB_DEC_PENDING;
    }
    int32_t flag = 0, first = INT32_MAX;
    const int rc = hb_quick_dec_decide(dec->qd, dec->z.data() + d, dec->nc, dec->cols, dec->C, 0, dec->C, dec->coeffs, &flag, &first, dec->stream);
    if (rc) return -rc;
    if (flag & 0x40000000) { (void)fail(dec->ctx, HB_ERR_HIP, "fused decode: a matrix entry left the range its host-side bound promised"); return -HB_ERR_HIP; }
    dec->first = flag ? first : INT32_MAX;
    return dec->state = flag ? HB_DEC_DISAGREE : HB_DEC_DONE;
}

int hb_dec_options(hb_dec *dec, int flags) {
    if (!dec || (flags & ~(HB_DEC_OPT_DEFER | HB_DEC_OPT_BESIDE))) return HB_ERR_BAD_ARG;
    if (dec->begun && dec->state == HB_DEC_PENDING) return fail(dec->ctx, HB_ERR_BAD_ARG, "decoder: a verdict is pending (hb_dec_settle first)");
    dec->deferred = (flags & HB_DEC_OPT_DEFER) != 0;
    return hb_quick_dec_beside(dec->qd, (flags & HB_DEC_OPT_BESIDE) ? 1 : 0);
}

int hb_dec_settle(hb_dec *dec) {
    if (!dec || !dec->begun) return -HB_ERR_BAD_ARG;
    if (dec->state != HB_DEC_PENDING) return dec->state;
    int32_t flag = 0, first = INT32_MAX;
    const int rc = hb_quick_dec_verdict(dec->qd, &flag, &first);
    if (rc) return -rc;
    if (flag & 0x40000000) { (void)fail(dec->ctx, HB_ERR_HIP, "fused decode: a matrix entry left the range its host-side bound promised"); return -HB_ERR_HIP; }
    dec->first = flag ? first : INT32_MAX;
    return dec->state = flag ? HB_DEC_DISAGREE : HB_DEC_DONE;
}

int hb_dec_arrived(hb_dec *dec, const int32_t *idx, int count, int32_t *consumed, int32_t *state) {
    if (!dec || !state || count < 0 || (count > 0 && !idx)) return HB_ERR_BAD_ARG;
    int used = 0, st = dec->begun ? dec->state : -HB_ERR_BAD_ARG;
    while (used < count && st == HB_DEC_COLLECTING) st = hb_dec_arrived1(dec, idx[used++]);
    if (consumed) *consumed = used;
    if (st < 0) return -st;
    *state = st;
    return HB_OK;
}

int hb_dec_verdict(const hb_dec *dec, int32_t *state, int32_t *first_bad) {
    if (!dec || !state) return HB_ERR_BAD_ARG;
    *state = dec->state;
    if (first_bad) *first_bad = dec->first;
    return HB_OK;
}

int hb_dec_arrivals_list(const hb_dec *dec, int32_t *out, int cap, int32_t *count) {
    if (!dec || !count || cap < 0 || (cap > 0 && !out)) return HB_ERR_BAD_ARG;
    const int k = (int)dec->z.size();
    *count = k;
    for (int i = 0; i < k && i < cap; i++) out[i] = dec->z[i];
    return HB_OK;
}

// ---- a candidate's waiting phase (reference reed_solomon.py:334-346) ---------------------------------------------------------------
// After the first disagreement the decoder holds candidates for the disagreeing polynomial: polynomials that few enough arrived senders
// contradict (device.py _candidate_cap has the proof that, while a candidate's contradictions stay within max_errors - confirmed, "accept it
// once |z| - E >= need, else wait" IS the reference's behaviour).  Until then every arrival is ONE question -- does the new sender's symbol of
// that chunk equal the candidate's value at its point? -- and rounds 4-5 asked it from Python: ~4 of an arrival's 13 us, 85 times in a row at
// n = 256 with the liars first.  This object asks it in C: the candidates' values at the n points stay here, an arrival is a symbol fetch
// (hb_symbols_fetch) and a compare, and the host hears about it only when something is to be DONE (a candidate can be accepted, or none is
// left) -- HB_WAIT_EVENT; the caller then reads the contradictions counted here (hb_wait_result) and goes on as before.
struct hb_wait {
    hb_ctx *ctx;
    int n, L;
    const uint64_t *cols;
    int64_t C, chunk;
    void *stream;
    int degree, max_errors, zlen, n_cands;
    std::vector<uint64_t> ev;                       // [n_cands][n][L]: the candidates at the parties' points
    std::vector<int32_t> base_err;                  // contradictions each had when the wait began
    std::vector<std::vector<int32_t>> new_err;      // ... and the senders that contradicted it since
    std::vector<uint8_t> dead;                      // left the cap at some arrival: gone for good (as the host's list drops it)
    bool armed;
};

int hb_wait_create(hb_ctx *ctx, int n, hb_wait **out) {
    if (!ctx || !out || n < 1) return HB_ERR_BAD_ARG;
    hb_wait *w = new hb_wait();
    w->ctx = ctx; w->n = n; w->L = ctx->n_limbs; w->cols = nullptr; w->C = 0; w->chunk = 0; w->stream = nullptr;
    w->degree = w->max_errors = w->zlen = w->n_cands = 0; w->armed = false;
    *out = w;
    return HB_OK;
}

int hb_wait_begin(hb_wait *w, const uint64_t *cols_dev, int64_t C, int64_t chunk, int degree, int max_errors, int zlen, int n_cands,
                  const uint64_t *values_host, const int32_t *contradictions, void *stream) {
    if (!w) return HB_ERR_BAD_ARG;
    w->armed = false;
    if (!cols_dev || !values_host || !contradictions || C < 1 || chunk < 0 || chunk >= C || degree < 0 || max_errors < 0 || zlen < 0 || n_cands < 1 || n_cands > 8)
        return fail(w->ctx, HB_ERR_BAD_ARG, "candidate wait: arguments");
    w->cols = cols_dev; w->C = C; w->chunk = chunk; w->degree = degree; w->max_errors = max_errors; w->zlen = zlen; w->n_cands = n_cands; w->stream = stream;
    w->ev.assign(values_host, values_host + (size_t)n_cands * w->n * w->L);
    w->base_err.assign(contradictions, contradictions + n_cands);
    w->new_err.assign((size_t)n_cands, std::vector<int32_t>());
    w->dead.assign((size_t)n_cands, 0);
    w->armed = true;
    return HB_OK;
}

// one arrival (the caller has filtered duplicates and confirmed errors, reed_solomon.py:369-372): HB_WAIT_ON -- nothing to do but wait for the
// next; HB_WAIT_EVENT -- a candidate can be accepted or none is left; -(hb_status) on an error.  n_confirmed: |confirmed errors| now.
int hb_wait_arrived1(hb_wait *w, int32_t idx, int n_confirmed) {
    if (!w || !w->armed || idx < 0 || idx >= w->n || n_confirmed < 0) return -HB_ERR_BAD_ARG;
    uint64_t sym[4];
    const int rc = hb_symbols_fetch(w->ctx, w->cols, w->n, w->C, w->chunk, &idx, 1, sym, w->stream);
    if (rc) return -rc;
    w->zlen += 1;
    const int need = w->degree + 1 + w->max_errors - n_confirmed;
    const int cap = std::max((w->zlen - w->degree - 1) / 2, w->max_errors - n_confirmed);      // (Gao: device.py _candidate_cap)
    bool alive = false, accept = false;
    for (int c = 0; c < w->n_cands; c++) {
        if (w->dead[c]) continue;
        if (memcmp(sym, &w->ev[((size_t)c * w->n + idx) * w->L], (size_t)w->L * 8) != 0) w->new_err[c].push_back(idx);
        const int e = w->base_err[c] + (int)w->new_err[c].size();
        if (e > cap) { w->dead[c] = 1; continue; }
        alive = true;
        if (w->zlen - e >= need) accept = true;
    }
    if (alive && !accept) return HB_WAIT_ON;
    w->armed = false;
    return HB_WAIT_EVENT;
}

// candidate `cand`: is it still standing, and the senders that contradicted it since hb_wait_begin (the first min(cap, *count) of them)
int hb_wait_result(const hb_wait *w, int cand, int32_t *standing, int32_t *senders, int cap, int32_t *count) {
    if (!w || cand < 0 || cand >= w->n_cands || !standing || !count || cap < 0 || (cap > 0 && !senders)) return HB_ERR_BAD_ARG;
    *standing = w->dead[cand] ? 0 : 1;
    const int k = (int)w->new_err[cand].size();
    *count = k;
    for (int i = 0; i < k && i < cap; i++) senders[i] = w->new_err[cand][i];
    return HB_OK;
}

void hb_wait_destroy(hb_wait *w) { delete w; }

void hb_dec_destroy(hb_dec *dec) {
    if (!dec) return;
    hb_quick_dec_destroy(dec->qd);
    delete dec;
}

}  // extern "C"
