/*
 * hbmpc_hip.h -- C ABI of libhbmpc_hip.so: the MI355X (gfx950) implementation of
 * HoneyBadgerMPC's batch share-reconstruction arithmetic.
 *
 * This is the drop-in boundary for the reference's NTL/Cython extension
 * `honeybadgermpc.ntl` (reference: honeybadgermpc/ntl/hbmpc_ntl_helpers.pyx,
 * re-exported by honeybadgermpc/ntl/__init__.py:1).  Each entry point names the
 * reference interface it replaces.  The reference marshals Python ints to NTL ZZ_p
 * through little-endian bytes (pyx:20-29); here a field element is the same
 * little-endian integer in a fixed width:
 *
 *     4 x uint64_t (32 bytes) for a context created with n_limbs = 4 (p < 2^256)
 *     1 x uint64_t ( 8 bytes) for a context created with n_limbs = 1 (p < 2^64)
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in signatures.  `stream` is a
 *     hipStream_t passed as void* (NULL = the default stream).
 *   - `*_dev` pointers are DEVICE pointers owned by the caller (e.g. a torch tensor's
 *     data_ptr()); everything else (points x, exponents zs, omega, the modulus) is a
 *     small HOST array copied at call time.
 *   - all element values are canonical residues in [0, p) on output; inputs may be any
 *     value < 2^(64*n_limbs) ("reduced on entry", pyx:31-32).
 *   - every function returns an hb_status; nothing throws or aborts.  Work is enqueued
 *     on `stream`; functions that must report a data-dependent result (singular matrix,
 *     validation mismatch) synchronise that stream before returning and say so below.
 *   - a context is bound to one device; one process per GPU is the intended use.
 */
#ifndef HBMPC_HIP_H
#define HBMPC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    HB_OK = 0,
    HB_ERR_SINGULAR = 1,     /* Vandermonde matrix not invertible / repeated point -> InterpolationError (pyx:167-169) */
    HB_ERR_BAD_ARG = 2,      /* ValueError-class problems (pyx:44,62) */
    HB_ERR_UNSUPPORTED = 3,  /* even modulus, size beyond a kernel's limits */
    HB_ERR_NO_DEVICE = 4,    /* no usable gfx950 device: the product path fails loudly, there is no CPU fallback */
    HB_ERR_HIP = 5,          /* a HIP runtime call failed; see hb_last_error() */
    HB_ERR_MISMATCH = 6,     /* hb_batch_open: re-encoded guess disagrees with a received column (reed_solomon.py:316-326) */
    HB_ERR_RETRY = 7         /* hb_probe_feed: the launch was void (a workgroup of a probe over several workgroups waited in vain for another: they
                                talk through memory and need each other resident); nothing was fed, the probe has been reset -- feed the whole list
                                again, after hb_probe_workgroups(pr, 0) if the chip is crowded */
} hb_status;

typedef struct hb_ctx hb_ctx;        /* modulus + Montgomery constants + device + table cache */
typedef struct hb_matrix hb_matrix;  /* a device-resident n_out x n_in matrix over GF(p) in kernel layout */

/* element layout selector for batched buffers: element (c, l) of a C x L batch lives at
 * base + (c*stride_c + l*stride_l) elements.  Chunk-major [C][L] is {L, 1}; party-major /
 * coefficient-major [L][C] is {1, C} (the layout R1/R2 messages travel in,
 * batch_reconstruction.py:165-167). */
typedef struct {
    int64_t stride_c;
    int64_t stride_l;
} hb_view;

/* ---- library / context ------------------------------------------------------------ */
int hb_version(void);
int hb_device_count(void);
/* Replaces ZZ_p::init(modulus) at the top of every pyx entry point (pyx:107,220,250,...). */
int hb_ctx_create(hb_ctx **out, const uint64_t *p_limbs, int n_limbs, int device);
void hb_ctx_destroy(hb_ctx *ctx);
const char *hb_last_error(const hb_ctx *ctx);
/* Tables derived from point sets (V, V^-1, error-locator bases, index maps) are cached per context, keyed by the
 * sorted point set, and bounded: once more than a cap of entries (192; HB_CACHE_CAP in the environment) are
 * resident, the least recently used ones are dropped on entry to the next call.  hb_ctx_cache_clear drops them all
 * (synchronises the device); hb_ctx_cache_entries reports how many are resident.  The reference keeps one
 * process-global FFT base-case cache flushed on modulus change (rsdecode_impl.h:18-20,52-65). */
int hb_ctx_cache_clear(hb_ctx *ctx);
int hb_ctx_cache_entries(const hb_ctx *ctx);
int hb_elem_bytes(const hb_ctx *ctx);

/* device-memory helpers for callers that have no allocator of their own */
int hb_malloc(hb_ctx *ctx, void **dptr, size_t bytes);
int hb_free(hb_ctx *ctx, void *dptr);
int hb_memcpy_h2d(hb_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes, void *stream);
int hb_memcpy_d2h(hb_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes, void *stream);
int hb_stream_sync(hb_ctx *ctx, void *stream);

/* ---- tables ------------------------------------------------------------------------ */
/* V[i][l] = x_i^l, n x d.  Replaces set_vm_matrix (rsdecode_impl.h:23-36).  The two point-set constructors hand out
 * shared, reference-counted tables: release every handle with hb_matrix_destroy. */
int hb_vand_matrix_create(hb_ctx *ctx, const uint64_t *x_host, int n, int d, hb_matrix **out, void *stream);
/* V(x)^-1, k x k.  Replaces vandermonde_inverse (rsdecode_impl.h:97-122).  Synchronises;
 * returns HB_ERR_SINGULAR when two points coincide mod p. */
int hb_vand_inverse_create(hb_ctx *ctx, const uint64_t *x_host, int k, hb_matrix **out, void *stream);
/* arbitrary matrix from host canonical elements, row-major n_out x n_in */
int hb_matrix_from_host(hb_ctx *ctx, const uint64_t *m_host, int n_out, int n_in, hb_matrix **out, void *stream);
/* copy a matrix back to host canonical row-major (testing / vandermonde_inverse's legacy dump, pyx:115-132) */
int hb_matrix_to_host(hb_ctx *ctx, const hb_matrix *m, uint64_t *m_host, void *stream);
void hb_matrix_destroy(hb_matrix *m);

/* ---- the hot kernel: batched mat-vec over GF(p) ---------------------------------------
 * out(c, i) = sum_l M[i][l] * in(c, in_rows ? in_rows[l] : l)     c < C, i < n_out
 * Replaces NTL mat_ZZ_p mul at pyx:183 and pyx:237.  in_rows (host, n_in ints or NULL)
 * selects which rows of a party-major buffer feed the product (the arrival set z of
 * IncrementalDecoder, reed_solomon.py:305-313). */
int hb_matvec(hb_ctx *ctx, const hb_matrix *m, const uint64_t *in_dev, hb_view in, const int32_t *in_rows,
              uint64_t *out_dev, hb_view out, int64_t C, void *stream);
/* Same product, but instead of storing, compare row i (for every i in check_rows[0..n_check))
 * with expect(c, i) and OR 1 into *mismatch_dev on any difference.  Replaces the Python
 * loop reed_solomon.py:316-319.  Asynchronous. */
int hb_matvec_check(hb_ctx *ctx, const hb_matrix *m, const uint64_t *in_dev, hb_view in, const int32_t *in_rows,
                    const uint64_t *expect_dev, hb_view expect, const int32_t *check_rows, int n_check,
                    int32_t *mismatch_dev, int64_t C, void *stream);

/* to_ZZ_p for packed batches (hbmpc_ntl_helpers.pyx:31-32: every value entering the reference's boundary is reduced
 * mod p): out[i] = in[i] mod p for `count` packed elements of any word content (in == out allowed).  *changed_dev, when
 * given, is incremented once per element that was not already canonical.  The kernels behind every other entry point
 * expect canonical residues; list-of-int inputs are reduced by the Python glue, packed numpy / torch batches pass through
 * this call.  Asynchronous. */
int hb_reduce(hb_ctx *ctx, const uint64_t *in_dev, uint64_t *out_dev, int64_t count, int32_t *changed_dev, void *stream);

/* ---- the robust path of IncrementalDecoder without plans (hb_quick.hip) -------------------------------------------
 * A decoder that is working its way past faulty senders sees every arrival set once: these entry points build what they
 * need on the device and enqueue it; none of them creates tables on the host. */
/* decoder.decode_batch over the arrivals z[0..d) + encoder.encode_batch + the compare loop with the later arrivals zc[0..nc)
 * (reed_solomon.py:305-326) as one launch: cols_dev is the party-major buffer [n][C] (row j = what party j sent), x_host the n
 * party points; coefficients go chunk-major to coeffs_dev ((C, d) elements; NULL: validate only).  On a disagreement
 * status_dev[0] |= 1 and status_dev[1] = min(status_dev[1], first disagreeing chunk - chunk_lo); the caller initialises both (0, INT32_MAX)
 * and reads them after synchronising.  Only the chunks [chunk_lo, chunk_hi) are read, written and compared (a decoder that has accepted
 * its first polynomials goes on from there).  z and zc are disjoint party indices.  Asynchronous.  HB_ERR_UNSUPPORTED outside the
 * full-size matrix-core kernel's range (narrow contexts, p outside [2^254, 0x7f 2^248), d < 4 or > 128, repeated points):
 * callers use an open plan then. */
int hb_quick_interp_check(hb_ctx *ctx, const uint64_t *x_host, int n, const int32_t *z, int d, const int32_t *zc, int nc,
                          const uint64_t *cols_dev, int64_t C, int64_t chunk_lo, int64_t chunk_hi, uint64_t *coeffs_dev, int32_t *status_dev,
                          void *stream);
/* The same, and every disagreeing chunk rather than only the first: bit (c - chunk_lo) of bad_map_dev (uint32 words, at least
 * ceil((chunk_hi - chunk_lo + 31) / 32) + 1 of them, zeroed by the caller) is set for every chunk c some compared column disagrees on.
 * A decoder whose liars corrupt one late chunk each reads the whole list off one launch (reed_solomon.py:334-365 walks the
 * polynomials in order; the candidates of one interpolation set do not change when a COMPARED sender is expelled).
 * bad_map_dev = NULL: hb_quick_interp_check. */
int hb_quick_interp_check_map(hb_ctx *ctx, const uint64_t *x_host, int n, const int32_t *z, int d, const int32_t *zc, int nc,
                              const uint64_t *cols_dev, int64_t C, int64_t chunk_lo, int64_t chunk_hi, uint64_t *coeffs_dev, int32_t *status_dev,
                              uint32_t *bad_map_dev, void *stream);

/* The optimistic step of ONE IncrementalDecoder (reed_solomon.py:305-330) in two halves, on wide contexts (points that are small
 * integers -- the production points 1 .. n -- on the small-entry kernel, hb_mfma_fused.hip; any other point set on the full-size one
 * with hb_quick_interp_check's device-built image): what depends on the first d arrivals alone is built when the d-th column lands
 * (hb_quick_dec_arrivals: enqueued, nothing waited for), and when the last column lands hb_quick_dec_decide builds the rows of the
 * compared senders, launches decode + validate (hb_mfma_fused.hip) over chunks [chunk_lo, chunk_hi) of the party-major buffer and
 * WAITS for the verdict, which the kernel hands over in pinned memory: *flag != 0 -- some compared column disagrees, *first = the
 * first disagreeing chunk - chunk_lo (INT32_MAX when none).  n_coef = d: all coefficients go chunk-major to coeffs_dev ((C, d)
 * elements); n_coef = 1: only the constant terms, to coeffs_dev[0 .. C) (what R1 forwards, batch_reconstruction.py:194).
 * HB_ERR_UNSUPPORTED from _create / _arrivals (narrow contexts, repeated points, fewer than 4 coefficients, more than 128, moduli
 * outside the matrix-core kernels' range): use an open plan. */
typedef struct hb_quick_dec hb_quick_dec;
int hb_quick_dec_create(hb_ctx *ctx, const uint64_t *x_host, int n, hb_quick_dec **out, void *stream);
int hb_quick_dec_arrivals(hb_quick_dec *qd, const int32_t *z, int d, int nc, int n_coef, void *stream);
int hb_quick_dec_decide(hb_quick_dec *qd, const int32_t *zc, int nc, const uint64_t *cols_dev, int64_t C, int64_t chunk_lo, int64_t chunk_hi,
                        uint64_t *coeffs_dev, int32_t *flag, int32_t *first, void *stream);
/* hb_quick_dec_decide in its two halves: _launch enqueues (the compared senders' rows, decode + validate) and returns, _verdict waits for
 * the last launch's verdict.  Between the two the calling thread is free (the next round's decoder, the next messages). */
int hb_quick_dec_launch(hb_quick_dec *qd, const int32_t *zc, int nc, const uint64_t *cols_dev, int64_t C, int64_t chunk_lo, int64_t chunk_hi,
                        uint64_t *coeffs_dev, void *stream);
int hb_quick_dec_verdict(hb_quick_dec *qd, int32_t *flag, int32_t *first);
/* on: hb_quick_dec_arrivals enqueues its build on a stream of the decoder's own (the caller's stream is busy with something the build does
 * not depend on -- it reads no column) and the next launch waits for it; off (default): in order on the caller's stream */
int hb_quick_dec_beside(hb_quick_dec *qd, int on);
void hb_quick_dec_destroy(hb_quick_dec *qd);

/* ---- IncrementalDecoder's optimistic phase as an object (hb_dec.hip) ------------------------------------------------------------
 * Replaces, for one round of one open, the state machine of IncrementalDecoder.add (reed_solomon.py:367-403) up to its first verdict:
 * _validate / duplicate and confirmed-error filtering (:288-300, :369-372), the optimistic decode + re-encode from the first degree + 1
 * arrivals (:305-313), the comparison of the later arrivals (:316-326) and the quorum test degree + 1 + max_errors - |confirmed|
 * (:302-303, :328-330) -- as batch_reconstruct drives it, one add() per received message (batch_reconstruction.py:43-61).
 * The host announces arrivals BY INDEX: party idx's column has landed in row idx of the party-major buffer cols_dev [n][C].  The
 * object enqueues what depends on the first degree + 1 arrivals when the last of them is announced, launches decode + validate when
 * the quorum is complete and waits for the verdict (pinned memory; the stream is not synchronised).  A decoder object is reusable:
 * hb_dec_begin starts the next round (another buffer, another set of confirmed errors).
 *   n_coef = degree + 1: all coefficients, chunk-major (C, degree + 1) to coeffs_dev;  n_coef = 1: the constant terms only, (C) elements
 *   (what R1 forwards, batch_reconstruction.py:194).  excluded[0..n_excluded): senders confirmed in error before this round
 *   (their arrivals are ignored; each lowers the quorum by one).
 * hb_dec_create / hb_dec_begin return HB_ERR_UNSUPPORTED for contexts / point sets / shapes outside the plan-free kernels
 * (hb_quick_dec_*'s conditions; also max_errors - n_excluded < 1: nothing to compare) -- callers keep their own path for those. */
#define HB_DEC_COLLECTING 0   /* more columns needed */
#define HB_DEC_DONE 1         /* every compared column agreed: the results are in coeffs_dev */
#define HB_DEC_DISAGREE 2     /* a compared column differs from the guess (reed_solomon.py:321-326): the robust phase takes over from
                                 hb_dec_arrivals_list; with n_coef = degree + 1 the refuted guess is in coeffs_dev and hb_dec_verdict
                                 gives the first disagreeing chunk */
#define HB_DEC_UNSUPPORTED 3  /* found at the (degree + 1)-th arrival: decode the arrival list another way */
#define HB_DEC_PENDING 4      /* only with HB_DEC_OPT_DEFER: decode + validate is enqueued, its verdict not read yet (hb_dec_settle) */
typedef struct hb_dec hb_dec;
int hb_dec_create(hb_ctx *ctx, const uint64_t *x_host, int n, int degree, int max_errors, hb_dec **out, void *stream);
int hb_dec_begin(hb_dec *dec, const uint64_t *cols_dev, int64_t C, int n_coef, uint64_t *coeffs_dev, const int32_t *excluded, int n_excluded,
                 void *stream);
/* one arrival: returns the state (HB_DEC_*) after it, or -(hb_status) on an error.  A sender already counted or excluded is ignored;
 * so is every arrival once the state has left HB_DEC_COLLECTING.  The call that completes the quorum returns when the verdict is in. */
int hb_dec_arrived1(hb_dec *dec, int32_t idx);
/* Options of the rounds to come (they hold across hb_dec_begin):
 *   HB_DEC_OPT_DEFER   the arrival that completes the quorum enqueues decode + validate and returns HB_DEC_PENDING instead of waiting;
 *                      hb_dec_settle waits for that verdict and returns the state it leads to (HB_DEC_DONE / HB_DEC_DISAGREE; any other
 *                      state as it is; -(hb_status) on an error).  While the state is HB_DEC_PENDING further arrivals are NOT counted
 *                      (hb_dec_arrived1 returns HB_DEC_PENDING): the caller keeps them and, after a HB_DEC_DISAGREE, hands them to its
 *                      robust phase in order.  What batch_reconstruct gains: its two rounds are subscribed up front
 *                      (batch_reconstruction.py:158-176), so the next round's decoder can be made, and its first messages taken, while
 *                      this round's launch runs.
 *   HB_DEC_OPT_BESIDE  the caller's stream is busy while this round's columns come in (the open's encode; the previous round's launch):
 *                      what depends on the first degree + 1 arrivals is built on a stream of the decoder's own, beside that work, and
 *                      the decode launch waits for it (hb_quick_dec_beside).  On an idle stream the dependency between two queues costs
 *                      more than the build: leave it off there. */
#define HB_DEC_OPT_DEFER 1
#define HB_DEC_OPT_BESIDE 2
int hb_dec_options(hb_dec *dec, int flags);
int hb_dec_settle(hb_dec *dec);
/* a burst of arrivals in order; stops at the first one that changes the state (*consumed = how many were taken) */
int hb_dec_arrived(hb_dec *dec, const int32_t *idx, int count, int32_t *consumed, int32_t *state);
int hb_dec_verdict(const hb_dec *dec, int32_t *state, int32_t *first_bad);
/* the senders counted so far, in arrival order (reed_solomon.py's _z): *count of them, the first min(cap, *count) copied */
int hb_dec_arrivals_list(const hb_dec *dec, int32_t *out, int cap, int32_t *count);
void hb_dec_destroy(hb_dec *dec);

/* ---- a candidate's waiting phase (hb_dec.hip) ------------------------------------------------------------------------------------
 * IncrementalDecoder's robust phase (reed_solomon.py:334-346) accepts a robust decode only once |z| - |errors| >= degree + 1 + max_errors -
 * |confirmed| and otherwise waits with nothing changed.  A decoder that holds CANDIDATES for the polynomial that disagreed -- polynomials few
 * enough arrived senders contradict; device.py _candidate_cap proves that such a candidate decides the reference's verdicts -- has ONE question
 * per arrival until then: does the new sender's symbol of that chunk equal the candidate's value at its point?  This object asks it: it keeps
 * the candidates' values at the n points (host), fetches the new sender's symbol (hb_symbols_fetch) and counts.  values_host [n_cands][n][limbs],
 * contradictions[c] = senders that already contradict candidate c, zlen = arrivals so far, n_cands <= 8.  hb_wait_arrived1 returns HB_WAIT_ON
 * (keep waiting), HB_WAIT_EVENT (a candidate can be accepted, or none is left: the wait is over, read hb_wait_result and act) or -(hb_status).
 * Gao's rule for how many contradictions a candidate may have: max(floor((|z| - degree - 1) / 2), max_errors - n_confirmed). */
#define HB_WAIT_ON 0
#define HB_WAIT_EVENT 1
typedef struct hb_wait hb_wait;
int hb_wait_create(hb_ctx *ctx, int n, hb_wait **out);
int hb_wait_begin(hb_wait *w, const uint64_t *cols_dev, int64_t C, int64_t chunk, int degree, int max_errors, int zlen, int n_cands,
                  const uint64_t *values_host, const int32_t *contradictions, void *stream);
int hb_wait_arrived1(hb_wait *w, int32_t idx, int n_confirmed);
int hb_wait_result(const hb_wait *w, int cand, int32_t *standing, int32_t *senders, int cap, int32_t *count);
void hb_wait_destroy(hb_wait *w);

/* The symbols of polynomial `chunk` in the columns of parties idx[0..count) (each in [0, n), count <= 64) of the party-major buffer cols_dev [n][C], to
 * out_host[count][limbs]: what IncrementalDecoder compares a new sender's share with (reed_solomon.py:318-321, data[i] against the guess) when
 * the guess is a candidate for ONE polynomial (device.py _candidate_cap).  One launch that writes pinned memory the call polls: the answer
 * is on the host a few microseconds after the columns are, where a synchronous 32-byte copy costs a stream synchronisation. */
int hb_symbols_fetch(hb_ctx *ctx, const uint64_t *cols_dev, int n, int64_t C, int64_t chunk, const int32_t *idx, int count, uint64_t *out_host, void *stream);
/* A candidate polynomial (coeffs_dev: d packed coefficients) against the received symbols of its chunk: the values it takes at the n party points
 * -> out_values_host[n][limbs] (the host keeps them for the senders still to come) and, per party, whether the symbol of `chunk` in its row of
 * the party-major buffer differs -> out_differs_host[n] (rows of parties that have not arrived hold whatever they hold: the caller looks at the
 * arrived ones).  IncrementalDecoder's comparison of the guess with the received shares (reed_solomon.py:316-326) for ONE polynomial: one launch, the
 * answer through pinned memory.  n <= 1024 (HB_ERR_UNSUPPORTED above). */
int hb_candidate_check(hb_ctx *ctx, const uint64_t *x_host, int n, const uint64_t *coeffs_dev, int d, const uint64_t *cols_dev, int64_t C, int64_t chunk,
                       uint64_t *out_values_host, uint8_t *out_differs_host, void *stream);
/* plumbing: `stream` goes on only after everything enqueued on `after` so far (an event of the context) */
int hb_stream_after(hb_ctx *ctx, void *stream, void *after);
/* plumbing: the context's side stream (made on first use, highest priority, non-blocking) -- where hb_dec / hb_quick_dec build beside a busy caller's
 * stream (HB_DEC_OPT_BESIDE) and where a caller may feed its probes (hb_probe_feed's `stream`): ONE for the context, a process has few hardware queues */
int hb_side_stream(hb_ctx *ctx, void **stream);

/* gao_interpolate for ONE codeword, incremental in its points (rsdecode_impl.h:325-363 as GaoRobustDecoder.robust_decode
 * runs it per polynomial, reed_solomon.py:151-186, 334-365): the probe keeps a reduced basis of the interpolation module of
 * the points fed so far (Koetter / Welch-Berlekamp; one workgroup, O(n') multiplications per new point) and decides exactly
 * as the reference's Gao does, beyond the unique-decoding radius included. */
typedef struct hb_probe hb_probe;
int hb_probe_create(hb_ctx *ctx, const uint64_t *x_host, int n, int k, hb_probe **out, void *stream);
/* feed element `poly` of the columns of parties idx[0..count) of the party-major buffer cols_dev [n][C] (arrival order, each party
 * once); with decide != 0 synchronise and report: *ok = the codeword decodes over everything fed, err_mask[0..n) = 1 at the
 * parties that are roots of the error locator (reed_solomon.py:174-184). */
int hb_probe_feed(hb_probe *pr, const int32_t *idx, int count, const uint64_t *cols_dev, int64_t C, int64_t poly, int decide,
                  int32_t *ok, uint8_t *err_mask, void *stream);
int hb_probe_reset(hb_probe *pr);          /* start over: another polynomial, or another arrival list */
/* workgroups a launch of this probe spreads over from now on (the probe must be reset: hb_probe_reset, or a HB_ERR_RETRY just returned): 0 = the
 * fewest the point set allows (1 up to 128 points, 2 above), otherwise 1 or 2 .. 8.  HB_ERR_BAD_ARG for a count the point set does not allow. */
int hb_probe_workgroups(hb_probe *pr, int wgs);
int hb_probe_points_fed(hb_probe *pr);
void hb_probe_destroy(hb_probe *pr);

/* ---- reference-shaped entry points (chunk-major buffers, tables cached in ctx) -------- */
/* vandermonde_batch_evaluate (pyx:199-244): polys_dev [C][d] -> out_dev [C][n] */
int hb_vandermonde_batch_evaluate(hb_ctx *ctx, const uint64_t *x_host, int n, const uint64_t *polys_dev,
                                  int64_t C, int d, uint64_t *out_dev, void *stream);
/* vandermonde_batch_interpolate (pyx:139-197): data_dev [C][k] -> out_dev [C][k]; HB_ERR_SINGULAR */
int hb_vandermonde_batch_interpolate(hb_ctx *ctx, const uint64_t *x_host, int k, const uint64_t *data_dev,
                                     int64_t C, uint64_t *out_dev, void *stream);
/* fft / partial_fft / fft_batch_evaluate (pyx:246-316, rsdecode_impl.h:125-192):
 * coeffs_dev [C][d] -> out_dev [C][k], out[c][i] = sum_{j<min(d,order)} coeffs[c][j] * omega^(i*j), i < k <= order */
int hb_fft_batch_evaluate(hb_ctx *ctx, const uint64_t *omega_host, int order, const uint64_t *coeffs_dev,
                          int64_t C, int d, int k, uint64_t *out_dev, void *stream);
/* fft_interpolate / fft_batch_interpolate (pyx:318-381, rsdecode_impl.h:194-265):
 * ys_dev [C][k] at points omega^zs[i] -> out_dev [C][k].  HB_ERR_SINGULAR for repeated zs. */
int hb_fft_batch_interpolate(hb_ctx *ctx, const uint64_t *omega_host, int order, const int32_t *zs_host, int k,
                             const uint64_t *ys_dev, int64_t C, uint64_t *out_dev, void *stream);
/* gao_interpolate, batched over C codewords sharing the same points (pyx:389-439,
 * rsdecode_impl.h:281-405).  ys_dev [C][npts]; coeffs_dev [C][k]; errloc_dev [C][npts+1]
 * (un-normalised EEA cofactor, errloc_len_dev[c] = deg+1); ok_dev[c] = 1 on success. */
int hb_gao_decode(hb_ctx *ctx, const uint64_t *x_host, int npts, int k, const uint64_t *ys_dev, int64_t C,
                  uint64_t *coeffs_dev, uint64_t *errloc_dev, int32_t *errloc_len_dev, uint8_t *ok_dev, void *stream);
/* Welch-Berlekamp (reed_solomon_wb.py:129-151), batched.  ys_dev [C][n], present_dev [C][n]
 * (0 = erasure).  coeffs_dev [C][k] zero padded, coeff_len_dev[c] = length after stripping
 * trailing zeros (polynomial.py:14-20), status_dev[c]: 0 ok, 1 "found no divisors!",
 * 2 "No solution", 3 too few points.  Words with at most floor((n' - k) / 2) errors over their n' surviving
 * points are settled by Gao's kernels (there the reference's solver can only return the closest codeword's
 * polynomial): complete words; words that all lost the SAME symbols (the protocol's case: the parties that
 * have not arrived, reed_solomon.py:201-204) as one batch over the points that are left; and, round 6, a batch
 * with up to 64 distinct patterns cut by pattern (groups of at least 64 codewords).  Every other word -- more
 * errors, small groups, also those Gao would still decode because the message has leading zeros -- goes through
 * the reference's own elimination, its descending-e loop and its particular solution (reed_solomon_wb.py:79-127,
 * 157-273).  The call returns with its last launch enqueued when Gao's kernels settled the whole batch. */
int hb_wb_decode(hb_ctx *ctx, const uint64_t *x_host, int n, int k, const uint64_t *ys_dev,
                 const uint8_t *present_dev, int64_t C, uint64_t *coeffs_dev, int32_t *coeff_len_dev,
                 int32_t *status_dev, void *stream);

/* sqrt_mod (pyx:441-444, NTL SqrRootMod), batched: out[i]^2 == a[i] (mod p), ok[i] = 0 for a non-residue.
 * Which of the two roots is returned is not pinned by the reference (tests/test_ntl.py:331-341). */
int hb_sqrt_mod(hb_ctx *ctx, const uint64_t *a_dev, int64_t C, uint64_t *out_dev, uint8_t *ok_dev, void *stream);

/* ---- one party's fault-free batch open -------------------------------------------------
 * The compute of batch_reconstruct (batch_reconstruction.py:158-227) for one party when no
 * received column is wrong: R1 encode; R1 optimistic decode + validating re-encode + compare
 * (reed_solomon.py:305-330); constant terms -> R2 message; R2 decode + re-encode + compare;
 * flatten.  3 batch encodes + 2 batch decodes, all on device.
 *   plan: created once per (points, arrival set); shares_dev [B]; r1_out_dev [n][C] party-major;
 *   r1_cols_dev / r2_cols_dev [n][C] party-major received columns; r2_msg_dev [C];
 *   result_dev [B].  C = ceil(B / d).
 * The hb_open_r* calls are asynchronous; hb_open_status synchronises and returns HB_OK or HB_ERR_MISMATCH. */
typedef struct hb_open_plan hb_open_plan;
int hb_open_plan_create(hb_ctx *ctx, int n, int d, int use_omega_powers, const uint64_t *x_host,
                        const uint64_t *omega_host, int order, const int32_t *z_host /* d arrivals used to decode */,
                        const int32_t *zc_host /* later arrivals to validate */, int n_check, int64_t max_B,
                        hb_open_plan **out, void *stream);
int hb_open_r1_encode(hb_open_plan *plan, const uint64_t *shares_dev, int64_t B, uint64_t *r1_out_dev, void *stream);
int hb_open_r1_decode(hb_open_plan *plan, const uint64_t *r1_cols_dev, int64_t B, uint64_t *r2_msg_dev, void *stream);
int hb_open_r2_decode(hb_open_plan *plan, const uint64_t *r2_cols_dev, int64_t B, uint64_t *result_dev, void *stream);
int hb_open_status(hb_open_plan *plan, void *stream);
/* Options.  HB_OPEN_OPT_VALIDATE_ARRIVED_ONLY (default 0): the reference re-encodes the guess at ALL n
 * points (encoder.encode_batch, reed_solomon.py:313) and then compares the columns that arrive; with
 * the option on, only output tiles containing a compared row are re-encoded.  Same accept/reject
 * decision, less arithmetic; off by default so that an open performs the reference's 3 full encodes.
 * HB_OPEN_OPT_MATRIX_CORES (default 1): when the Vandermonde entries fit 16 signed base-256 digits and
 * 2^254 <= p < 2^256 (the reference's BLS12-381 scalar field with the default points 1..n), the R1 encode
 * and the validating re-encode run as an exact int8 GEMM on the matrix cores (csrc/hb_mfma.hip); 0 forces
 * the integer-VALU kernels.  Results are bit-identical either way.  get_option reports whether the
 * matrix-core path is in use for this plan (0 when the plan's shapes do not qualify).
 * HB_OPEN_OPT_FUSED_VALIDATE (default 1): plans on the matrix cores with at least 4 coefficients and 4 points decode AND
 * validate in one launch of the full-size matrix-core kernel: the value the guess takes at a later
 * arrival's point is a linear function of the arrival set, V[zc] (Vinv y) = (V[zc] Vinv) y, so the rows [Vinv rows wanted ;
 * V[zc] Vinv] applied to the received columns give the coefficients and the predictions to compare (reference:
 * decoder.decode_batch + encoder.encode_batch + compare, reed_solomon.py:300-323; same canonical values, same accept /
 * reject).  The two matrices are built on the device when the plan is created (hb_quick.hip: enqueued, nothing waited for);
 * where that builder does not apply (p >= 0x7f 2^248, repeated points) they are built through the host (1-2.5 ms) when the plan
 * decodes for the third time, or at once when the option is set to 1.
 * At points that are distinct integers below 2^16 (the production points 1 .. n; 4 <= d <= 22) the two products factor as
 * [N ; P] (y ./ den) with matrices of small INTEGERS (numerators of the Lagrange basis and its values at the compared points):
 * such plans decode + validate on the small-entry kernel with the division by den_j inside it (csrc/hb_mfma_fused.hip) -- value 2
 * selects the full-size kernel for them all the same.
 * 0: decode, then re-encode all n points (NTT or mat-vec) and compare.  get_option: 0 = off / unavailable, 1 = on (full-size
 * kernel), 3 = on (small-entry kernel). */
#define HB_OPEN_OPT_VALIDATE_ARRIVED_ONLY 1
#define HB_OPEN_OPT_MATRIX_CORES 2
#define HB_OPEN_OPT_FUSED_VALIDATE 3
int hb_open_plan_set_option(hb_open_plan *plan, int option, int value);
int hb_open_plan_get_option(hb_open_plan *plan, int option, int *value);
void hb_open_plan_destroy(hb_open_plan *plan);

/* host-side self test of the radix-2^29 arithmetic templates (no GPU needed):
 * out = a*b mod p computed with the same code the kernels use. */
int hb_selftest_mulmod(const uint64_t *p_limbs, int n_limbs, const uint64_t *a, const uint64_t *b, uint64_t *out);

#ifdef __cplusplus
}
#endif
#endif /* HBMPC_HIP_H */
