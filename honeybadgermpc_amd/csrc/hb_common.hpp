// hb_common.hpp -- host-side context, constants and small helpers shared by the C-ABI
// translation units of libhbmpc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hbmpc_hip.h"
#include "fp29.hpp"

namespace hb {
struct FastMatrix;
struct Mm8Matrix;
struct Mm8wMatrix;
}
namespace hb {

// Two instantiations: <NL=9 digits, NW=8 words> for p < 2^256, <3, 2> for p < 2^64.
struct Wide { static constexpr int NL = 9, NW = 8; };
struct Narrow { static constexpr int NL = 3, NW = 2; };

constexpr int OT = 4;  // outputs per wave tile in the mat-vec kernel

// ONE polynomial (d packed coefficients, whatever words the caller packed: below R) at ONE point (packed), by the 128 threads of a
// workgroup: thread l raises the point to its powers l, l + 128, ... by square and multiply, multiplies by the coefficients, and a tree
// over `red` adds up -- ~20 dependent multiplications whatever d.  The canonical value comes back in every thread's r (read from red[0]).
// (k_eval_few, k_candidate_check: what the device decoder asks for a candidate, where the batched kernels' lane-per-polynomial is one lane.)
template <int NL, int NW>
__device__ __forceinline__ void eval_at_point_128(uint32_t (&r)[NL], const uint32_t *__restrict__ x_packed, const uint32_t *__restrict__ poly, int d,
                                                  const FpParams<NL> &P, uint32_t (*red)[NL]) {
    const int tid = threadIdx.x;
    uint32_t xd[NL], xm[NL], acc[NL];
    load_digits<NL, NW>(xd, x_packed);
    to_mont(xm, xd, P);
#pragma unroll
    for (int q = 0; q < NL; q++) acc[q] = 0;
    for (int l = tid; l < d; l += 128) {
        uint32_t pw[NL], cd[NL], m[NL];
        fp_pow_u32(pw, xm, (uint32_t)l, P);                       // Montgomery form of x^l
        load_digits<NL, NW>(cd, poly + (size_t)l * NW);
        mont_mul(m, cd, pw, P);                                   // coefficient x Montgomery power -> the canonical product
        fp_add(acc, acc, m, P);
    }
#pragma unroll
    for (int q = 0; q < NL; q++) red[tid][q] = acc[q];
    __syncthreads();
    for (int w = 64; w >= 1; w >>= 1) {
        if (tid < w) {
            uint32_t a[NL], b[NL], t[NL];
#pragma unroll
            for (int q = 0; q < NL; q++) { a[q] = red[tid][q]; b[q] = red[tid + w][q]; }
            fp_add(t, a, b, P);
#pragma unroll
            for (int q = 0; q < NL; q++) red[tid][q] = t[q];
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < NL; q++) r[q] = red[0][q];
}

// word index of digit q of element (i, l) of an n_out x n_in matrix in kernel layout
// [tile][l][digit][OT], tile = i / OT: the OT outputs of one digit form one aligned uint4
__host__ __device__ inline size_t m_index(int i, int l, int n_in, int nl, int q) {
    return (((size_t)((i / OT) * n_in + l) * (size_t)nl + (size_t)q) * OT) + (size_t)(i % OT);
}
inline int m_tiles(int n_out) { return (n_out + OT - 1) / OT; }

}  // namespace hb

struct hb_matrix {
    hb_ctx *ctx;
    int n_out, n_in;
    uint32_t *dev;      // Montgomery digits, kernel layout, zero padded to whole tiles
    size_t words;
    bool cached;        // created through the ctx cache (hb_vand_matrix_create / hb_vand_inverse_create)
    int refs;           // handles outstanding: one per create call that returned it, plus one while the cache holds it
    hb::Mm8wMatrix *wide;   // int8 matrix-core image (hb_mfma_wide.hip), built on first use; nullptr when it does not apply
    bool wide_tried;
};

namespace hb {
// ctx->err: assignment and c_str() act on the CALLING thread's message (hb_last_error is asked for by the thread whose call failed)
struct TlsError {
    static std::string &slot() { static thread_local std::string s; return s; }
    TlsError &operator=(const std::string &m) { slot() = m; return *this; }
    TlsError &operator=(const char *m) { slot() = m ? m : ""; return *this; }
    const char *c_str() const { return slot().c_str(); }
    void clear() { slot().clear(); }
    bool empty() const { return slot().empty(); }
};
// constants of the table pre-scale (k_prescale_tab): 2^261 - p in digits, 2^256 - p in words, floor(2^290 / p) in digits
struct PrescaleParams {
    uint32_t pbar[9];
    uint32_t pneg[8];
    uint32_t m0, m1;
};
}
// Environment hooks (diagnostics and A/B runs; DESIGN section 7 says which profile or test cites each): read ONCE, at the first question any
// of them is asked, not on every call of the data path (rounds 1-5: 28 getenv calls, two of them per hb_wb_decode).  env_hook returns the
// value (nullptr: unset) as it was then; hb_debug_reload_env() (include/hbmpc_hip_debug.h) reads them again -- the tests that flip a hook
// inside one process call it.
namespace hb {
enum EnvHook {
    ENV_CACHE_CAP,
    ENV_GAO_PAIR,
    ENV_MM8W_FLAT,
    ENV_MM8W_RQ,
    ENV_MM8W_TILE16,
    ENV_MM8_NO_SKIP,
    ENV_NO_EVAL_FEW,
    ENV_NO_FUSED_SMALL,
    ENV_NO_FUSED_VALIDATE,
    ENV_NO_MFMA,
    ENV_NO_MFMA_DECODE,
    ENV_NO_MFMA_WIDE,
    ENV_NO_NARROW_FAST,
    ENV_NO_QUICK,
    ENV_NO_QUICK_PLAN,
    ENV_NTT_STAGE_LOOP,
    ENV_PROBE_WGS,
    ENV_QUICK_NO_CAND,
    ENV_UPLOAD_MODE,
    ENV_WB_NO_GAO,
    ENV_WB_NO_UNIFORM,
    ENV_COUNT
};
const char *env_hook(EnvHook h);
void env_reload();
}
struct hb_ctx {
    int device;
    int n_limbs;        // 1 or 4 (uint64 limbs per element at the ABI)
    hb::FpParams<9> pw; // valid when n_limbs == 4
    hb::FpParams<3> pn; // valid when n_limbs == 1
    uint64_t p_limbs[4];
    // One context may be shared by several host threads (device.py keeps a per-thread plan cache over one Context per
    // modulus, and ctypes releases the GIL during calls): every API entry point that takes a context, a plan or a matrix
    // handle holds `mu` from its table lookups to the enqueue of its last launch (HB_API_GUARD), so the registry, the cache
    // maps, reference counts, lazily built images and cache_trim never run concurrently.  Recursive: entry points call
    // each other.  The last error is kept per calling thread.
    std::recursive_mutex mu;
    int api_depth = 0;                                // entry points on the stack of the thread that holds `mu`
    hb::TlsError err;
    std::map<std::string, hb_matrix *> mcache;        // tables keyed by (kind, n, d, point bytes)
    std::map<std::vector<int32_t>, int32_t *> icache; // small int arrays resident on device
    std::map<std::string, void *> dcache;             // other device tables (twiddles, ...), hipFree'd with the ctx
    std::map<std::string, hb::FastMatrix *> fcache;   // second-generation (raw small-entry) tables
    std::map<std::string, hb::Mm8Matrix *> m8cache;   // their int8 matrix-core images (nullptr: does not qualify)
    // Every cached table is also an entry of `lru` (key = "<map>|<map key>"): looked-up entries are touched, and cache_trim,
    // called on entry to the API functions that use tables, drops the least recently used ones above `cache_cap`
    // (after a device synchronise, so that no kernel in flight still reads them).  Pinned entries (twiddles, the sqrt
    // constants: a handful per modulus) are never dropped.  Nothing outside a single API call may keep a pointer into a
    // cache: plans own copies of their tables.
    struct CacheSlot { uint64_t tick; bool pinned; std::function<void()> drop; };
    std::map<std::string, CacheSlot> lru;
    uint64_t lru_clock = 0;
    size_t cache_cap = 192;                           // entries; HB_CACHE_CAP overrides (tests)
    int32_t *flag_dev;                                // 64 status words
    hb::PrescaleParams psc;                           // valid when psc_state == 1
    int psc_state = 0;                                // 0 not computed yet, 1 valid, -1 modulus outside [2^254, 2^256)
    // the plan-free robust path (hb_quick.hip): per point set a table of x, x^i and 1 / (x_a - x_b); a ring of scratch slots for the
    // matrices built on the device; pooled probe states; per inner dimension the constants every device-built image shares
    std::map<std::string, void *> ptcache;            // hb::PointTable *
    struct QuickSlot { void *buf = nullptr; size_t cap = 0; void *ev = nullptr; };
    std::vector<QuickSlot> qslots;
    unsigned qnext = 0;
    std::vector<void *> probe_pool, probe_host_pool;
    void *side_stream = nullptr;                      // hipStream_t: ONE stream for every decoder's builds beside the caller's stream (a process has four
                                                      // hardware queues: a stream per decoder object would share them with the caller's)
    // Scratch of the batched robust decoders (hb_gao_decode / hb_wb_decode: interpolants, locators, side records -- 0.85 GB each at config 4),
    // kept with the context and reused by the next call: ctx_scratch() below
    std::map<std::string, std::pair<void *, size_t>> scratch;
    void *fetch_host = nullptr, *fetch_dev = nullptr; // hb_symbols_fetch: pinned, device-visible hand-over buffer (hb::SymFetch)
    int fetch_seq = 0;
    void *cand_host = nullptr, *cand_dev = nullptr;   // hb_candidate_check: pinned hand-over buffer (hb::CandCheck) and its ticket counter on the device
    int32_t *cand_ticket = nullptr;
    int cand_seq = 0;
    void *after_event = nullptr;                      // hb_stream_after: one event, recorded and waited for under the context's mutex
    std::map<int, void *> wide_shared;                // d -> hb::Mm8wShared *
    std::map<std::string, std::vector<int>> wide_shapes;   // launch geometry per (row tiles, K-blocks, chunk tiles)
    void *mm8_shared = nullptr;                       // hb::Mm8Shared *: constants of the small-entry matrix-core kernels (hb_mm8.hpp)
    int elem_words() const { return n_limbs == 4 ? 8 : 2; }
    int nl() const { return n_limbs == 4 ? 9 : 3; }
};

// entry-point guard: holds the context's mutex and counts the nesting of entry points (one calls another: hb_wb_decode ->
// hb_gao_decode -> hb_vand_inverse_create).  cache_trim only acts in the OUTERMOST call: an inner call must not free tables
// the outer one looked up and is about to launch with.
struct hb_api_guard {
    hb_ctx *c;
    explicit hb_api_guard(hb_ctx *ctx) : c(ctx) { if (c) { c->mu.lock(); c->api_depth++; } }
    ~hb_api_guard() { if (c) { c->api_depth--; c->mu.unlock(); } }
    hb_api_guard(const hb_api_guard &) = delete;
    hb_api_guard &operator=(const hb_api_guard &) = delete;
};
#define HB_API_GUARD(ctxexpr) hb_api_guard hb_api_guard__(ctxexpr)

// A named device buffer of at least `bytes` that lives as long as the context (or until hb_ctx_cache_clear): grown when a call asks for more.
// For the temporaries of entry points that (a) hold the context's mutex from start to finish and (b) synchronise their stream before they
// return -- so no two calls ever use a slot at once and nothing is in flight when the next call reuses or regrows it.  A hipMalloc + hipFree
// pair of config 4's sizes cost 5-14 ms of a 27 ms decode on some boxes (profiles/r04_bench_cfg4*: 41 ms a call against 27).
inline int ctx_scratch(hb_ctx *ctx, const char *slot, size_t bytes, void **out) {
    auto &e = ctx->scratch[slot];
    if (e.second < bytes || !e.first) {
        if (e.first) { (void)hipFree(e.first); e.first = nullptr; e.second = 0; }
        const size_t want = bytes ? bytes : 4;
        const hipError_t err = hipMalloc(&e.first, want);
        if (err != hipSuccess) { e.first = nullptr; ctx->err = std::string("scratch ") + slot + ": " + hipGetErrorString(err); return HB_ERR_HIP; }
        e.second = want;
    }
    *out = e.first;
    return HB_OK;
}
inline void ctx_scratch_free(hb_ctx *ctx) {
    for (auto &kv : ctx->scratch) if (kv.second.first) (void)hipFree(kv.second.first);
    ctx->scratch.clear();
}

#define HB_HIP(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return HB_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

#define HB_LAUNCH_CHECK(ctx)                                                                      \
    do {                                                                                          \
        hipError_t e__ = hipGetLastError();                                                       \
        if (e__ != hipSuccess) {                                                                  \
            (ctx)->err = std::string("kernel launch: ") + hipGetErrorString(e__);                 \
            return HB_ERR_HIP;                                                                    \
        }                                                                                         \
    } while (0)

namespace hb {

inline int fail(hb_ctx *ctx, int code, const char *msg) {
    if (ctx) ctx->err = msg;
    return code;
}
// nsub for a lazy dot product of length d whose inputs are < 2^(32*NW) and matrix entries < p:
// REDC output < p * (1 + d * 2^(32 NW) / 2^(29 NL))
inline int nsub_for(int d, int nl, int nw) {
    int shift = 29 * nl - 32 * nw;  // 5 (wide) or 23 (narrow)
    long per = 1L << shift;
    long n = (d + per - 1) / per;
    return n < 1 ? 1 : (int)n;
}
int get_int_array(hb_ctx *ctx, const int32_t *host, int n, int32_t **dev, hipStream_t s);   // cached: valid for the current API call only
int own_int_array(hb_ctx *ctx, const int32_t *host, int n, int32_t **dev, hipStream_t s);   // caller hipFree's it
void cache_note(hb_ctx *ctx, const std::string &rk, std::function<void()> drop, bool pinned = false);
void cache_touch(hb_ctx *ctx, const std::string &rk);
void cache_trim(hb_ctx *ctx);
void matrix_unref(hb_matrix *m);
int alloc_matrix(hb_ctx *ctx, int n_out, int n_in, hb_matrix **out);
int upload_elems(hb_ctx *ctx, const uint64_t *host, size_t count, uint32_t **dev, hipStream_t s);
int upload_table(hb_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes, hipStream_t s);   // synchronises; last write by a kernel
std::string table_key(const char *kind, hb_ctx *ctx, const uint64_t *x, int n, int d);
// out[i] = base^(exps ? exps[i] : i), canonical packed, freshly hipMalloc'ed (caller frees)
int pow_points_dev(hb_ctx *ctx, const uint64_t *base_host, const int32_t *exps_dev, int count, uint32_t **out_dev, hipStream_t s);
int vand_matrix_from_dev(hb_ctx *ctx, const std::string &key, const uint32_t *x_dev, int n, int d, hb_matrix **out, hipStream_t s);
int vinv_from_dev(hb_ctx *ctx, const std::string &key, const uint32_t *x_dev, int k, hb_matrix **out, hipStream_t s);
// out(c,i) = sum_l M[i][l] * in(c, rows[l]); CHECK mode when check_mask_dev != nullptr (out = expected values)
// hb_gao_decode with the locators optional (hb_gao.hip; errloc_dev == nullptr: only their lengths are produced)
int gao_decode(hb_ctx *ctx, const uint64_t *x_host, int npts, int k, const uint64_t *ys_dev, int64_t C,
               uint64_t *coeffs_dev, uint64_t *errloc_dev, int32_t *errloc_len_dev, uint8_t *ok_dev, void *stream, int ys_stride = 0, const int32_t *sel_host = nullptr);
int launch_matvec(hb_ctx *ctx, const hb_matrix *m, const uint32_t *in, hb_view iv, const int32_t *in_rows_dev, int64_t in_count,
                  uint32_t *out, hb_view ov, int64_t out_count, const int32_t *check_mask_dev, int32_t *mismatch_dev,
                  int64_t C, hipStream_t s);
int launch_copy_view(hb_ctx *ctx, const uint32_t *src, hb_view sv, uint32_t *dst, hb_view dv, int64_t C, int L, int64_t dst_count, hipStream_t s);

// ---- NTT (hb_ntt.hip) ---------------------------------------------------------------------
int get_twiddles(hb_ctx *ctx, const uint64_t *omega_host, int n, uint32_t **tw, hipStream_t s);
int launch_ntt_lds(hb_ctx *ctx, const uint32_t *tw, int n, const uint32_t *in, hb_view iv, int64_t in_count, int d, int k,
                   uint32_t *out, hb_view ov, int64_t out_count, const int32_t *check_mask_dev, int32_t *mismatch_dev,
                   int64_t C, hipStream_t s, uint32_t *copy_dst = nullptr, hb_view cpv = hb_view{0, 0}, int64_t copy_count = 0, int copy_rows = 0);

// ---- second-generation (raw small-entry matrix) path, hb_fast.hip ----------------------
struct FastMatrix {
    int n_out, n_in;
    int ot;             // outputs per tile in this matrix's layout (2 or 4)
    uint32_t *M;        // raw canonical digits, [tile][l][digit][OT]
    int32_t *nd;        // [tile][n_in] digits actually non-zero in that tile/term
    int32_t *negrow;    // [n_out] 1 => negate the output row (nullptr: none)
    uint32_t *K;        // [n_in][NL] pre-scale constants (canonical digits; used as a mont_mul factor)
    uint32_t *K2;       // factored inverses only: R^2 / den_j, the pre-scale that makes the outputs canonical (else nullptr)
    uint32_t *K1;       // factored inverses only: R / den_j, mont_mul(x, K1_j) = x / den_j (plain), for the matrix-core decode
    uint32_t *KT;       // factored inverses, 9-digit contexts: [n_in][9][9] canonical digits of 2^(29 q) / den_j (table pre-scale)
};
// word index of digit q of raw matrix element (i, l): [tile][l][digit][ot], tile = i / ot
__host__ __device__ inline size_t mf_index(int i, int l, int n_in, int nl, int q, int ot) {
    return (((size_t)((i / ot) * n_in + l) * (size_t)nl + (size_t)q) * ot) + (size_t)(i % ot);
}
void fast_matrix_free(FastMatrix *m);
int fast_vand_create(hb_ctx *ctx, const uint32_t *x_dev, int n, int d, FastMatrix **out, hipStream_t s);
int fast_vinv_create(hb_ctx *ctx, const uint32_t *x_dev, int k, FastMatrix **out, hipStream_t s);
int launch_prescale(hb_ctx *ctx, const FastMatrix *m, const uint32_t *in, hb_view iv, const int32_t *rows_dev, int64_t in_count,
                    uint32_t *out_dg, int64_t C, hipStream_t s);
int launch_prescale_pk(hb_ctx *ctx, const FastMatrix *m, const uint32_t *in, hb_view iv, const int32_t *rows_dev, int64_t in_count,
                       uint32_t *out_pk, int64_t C, hipStream_t s);
int launch_matvec2(hb_ctx *ctx, const FastMatrix *m, const uint32_t *in_dg,
                   const uint32_t *in_pk, hb_view iv, const int32_t *in_rows_dev, int64_t in_count, uint32_t *scratch_dg,
                   uint32_t *out_pk, hb_view ov, int64_t out_count,
                   int pk_rows, int pk_from_mont, uint32_t *out_dg, const int32_t *check_mask_dev, int32_t *mismatch_dev,
                   int64_t C, hipStream_t s, int check_skip = 0, const uint32_t *K_override = nullptr);

int launch_decode_check(hb_ctx *ctx, const FastMatrix *dec, const FastMatrix *enc, const uint32_t *cols, hb_view cv,
                        const int32_t *z_dev, uint32_t *pk_dst, hb_view pv, int64_t pk_count, int pk_rows, uint32_t *coef_dg,
                        const int32_t *mask_dev, int32_t *mismatch_dev, int64_t C, hipStream_t s, int check_skip);

// ---- third generation: int8 matrix-core mat-vec for small-entry matrices, hb_mfma.hip ----------
bool prescale_params(hb_ctx *ctx);   // fills ctx->psc on first use; false when the table pre-scale does not apply
int mm8_from_fast(hb_ctx *ctx, const FastMatrix *f, Mm8Matrix **out, hipStream_t s, const int32_t *rows = nullptr, int n_rows = 0);
void mm8_free(Mm8Matrix *m);
int launch_mm8(hb_ctx *ctx, const Mm8Matrix *m, const uint32_t *in, hb_view iv, const int32_t *in_rows_dev, int64_t in_count,
               uint32_t *out, hb_view ov, int64_t out_count, const int32_t *check_mask_dev, int32_t *mismatch_dev,
               int64_t C, hipStream_t s, uint32_t *copy_dst = nullptr, hb_view cpv = hb_view{0, 0}, int64_t copy_count = 0,
               int copy_rows = 0, const int32_t *check_rows_dev = nullptr);

// ---- matrix-core mat-vec for full-size entries, hb_mfma_wide.hip ----------------------------------------
int mm8w_from_host(hb_ctx *ctx, const uint64_t *m_host, int n_out, int n_in, Mm8wMatrix **out, hipStream_t s);
void mm8w_free(Mm8wMatrix *m);
int launch_mm8w(hb_ctx *ctx, const Mm8wMatrix *m, const uint32_t *in, hb_view iv, const int32_t *in_rows_dev, int64_t in_count,
                uint32_t *out, hb_view ov, int64_t out_count, const int32_t *check_mask_dev, int32_t *mismatch_dev,
                int64_t C, hipStream_t s, const uint32_t *cmp = nullptr, hb_view cv = hb_view{0, 0}, int n_store = 0);
// device-built images (hb_quick.hip): geometry of an n_out x d image, the per-(context, d) constants, and the launch over
// borrowed buffers; first_bad_dev (CHECK mode, optional): atomicMin of the first chunk whose compare failed
// the fold of a sum's high bytes on the matrix cores (hb_mfma_wide.hip): table rows of 272 bytes, eight per 16 bytes of H
bool fold_tables(hb_ctx *ctx, int n_bytes, uint8_t *fold, uint32_t *top8, uint32_t *mu, uint32_t *shift8);
// completion signal of a launch whose caller waits for the verdict: a device counter of finished workgroups, the pinned (device-visible)
// record the last one fills in, the sequence number it writes last
// The hand-over is ONE 64-bit store -- [63:32] the sequence number, [31] some compared column disagrees, [30] a matrix entry left its range
// (FS_OVERFLOW), [29:0] the first disagreeing chunk (all ones: none) -- so the host reads a verdict that is whole the moment it sees the
// number (rounds 4-5: three words with a system-scope fence between the last two, a PCIe round trip on the path of every decode)
struct FsVerdict { unsigned long long word; unsigned long long pad; };
struct FsDone { int32_t *counter; FsVerdict *host; int32_t seq; };
constexpr int32_t FS_VERDICT_NONE = 0x3fffffff;              // launches a caller waits for take fewer chunks than this (hb_quick.hip)
#ifdef __HIPCC__
// called by the last workgroup of a launch (every other one has fenced and counted itself): read the status words, publish, reset them
__device__ __forceinline__ void fs_publish_verdict(int32_t *mismatch, int32_t *first_bad, int32_t *counter, FsVerdict *host, int32_t seq) {
    const int32_t fl = mismatch ? __hip_atomic_load(mismatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const int32_t fb = first_bad ? __hip_atomic_load(first_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : INT32_MAX;
    const uint32_t low = ((fl & 1) ? 0x80000000u : 0u) | ((fl & 0x40000000) ? 0x40000000u : 0u) | (uint32_t)((fb < 0 || fb > FS_VERDICT_NONE) ? FS_VERDICT_NONE : fb);
    // (relaxed: the host reads nothing but this word; what the launch wrote to HBM is ordered for the stream's next launch by the launch's end, and
    // for the device at large by every workgroup's fence before it counted itself -- a system-scope release here would write back the L2)
    __hip_atomic_store(&host->word, ((unsigned long long)(uint32_t)seq << 32) | low, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (mismatch) __hip_atomic_store(mismatch, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (first_bad) __hip_atomic_store(first_bad, INT32_MAX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (counter) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The end of a workgroup of such a launch.  Its flag atomics (and stores) must have been ACKNOWLEDGED before it counts itself -- s_waitcnt in every
// wave, then the barrier: the atomics execute at the coherence point, so the last workgroup's reads of the status words are final.  Rounds
// 4-5 had a __threadfence() here: an agent-scope release, i.e. a write-back of the XCD's L2 per workgroup -- 8 us of config 3's R2 launch
// (round 6, kernel trace: 62.2 -> 54.4 us).  What the launch wrote to HBM is ordered for the caller's stream by the launch's end, as for any
// launch; a consumer on ANOTHER stream orders itself behind the caller's stream (hb_stream_after), not behind the verdict.
__device__ __forceinline__ void fs_workgroup_done(const FsDone &done, int32_t *mismatch, int32_t *first_bad) {
    if (!done.counter) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(done.counter, 1) == (int)gridDim.x - 1) fs_publish_verdict(mismatch, first_bad, done.counter, done.host, done.seq);
}
#endif
// host side of the same word
inline bool fs_verdict_is(const FsVerdict *host, int32_t seq) { return (int32_t)(*reinterpret_cast<const volatile unsigned long long *>(&host->word) >> 32) == seq; }
inline void fs_verdict_read(const FsVerdict *host, int32_t *flag, int32_t *first) {
    const uint32_t low = (uint32_t)*reinterpret_cast<const volatile unsigned long long *>(&host->word);
    *flag = ((low >> 31) ? 1 : 0) | ((low & 0x40000000u) ? 0x40000000 : 0);
    const int32_t f = (int32_t)(low & 0x3fffffffu);
    *first = f == FS_VERDICT_NONE ? INT32_MAX : f;
}
struct Mm8wShared { uint32_t bias; uint32_t c80r[9], biasmod[9]; void *wp; uint32_t *zero; };
int mm8w_geometry(int n_out, int d, int *tile_rows, int *n_rt, int *nkb, size_t *a8_bytes, size_t *crow_words);
int mm8w_shared(hb_ctx *ctx, int d, const Mm8wShared **out, hipStream_t s);
void mm8w_shared_free(hb_ctx *ctx);
int launch_mm8w_raw(hb_ctx *ctx, int n_out, int d, int tile_rows, const void *a8, const uint32_t *crow, const Mm8wShared *sh,
                    const uint32_t *in, hb_view iv, const int32_t *in_rows_dev, int64_t in_count, uint32_t *out, hb_view ov, int64_t out_count,
                    const int32_t *check_mask_dev, int32_t *mismatch_dev, int64_t C, hipStream_t s, const uint32_t *cmp, hb_view cv, int n_store,
                    int32_t *first_bad_dev, uint32_t *bad_map_dev = nullptr, const FsDone *done = nullptr);
void point_tables_free(hb_ctx *ctx);
void mm8_shared_free(hb_ctx *ctx);
// per point set: x (Montgomery digits), the powers x_a^i and 1 / (x_a - x_b); for sets of small integers also their values
struct PointTable {
    int n, S;
    uint32_t *xm, *inv, *pw;
    bool usable;
    int refs;             // the cache's reference + one per probe that works on this table
    bool small;           // every point is an integer below 2^16 (the production points 1 .. n): xs holds them
    std::vector<uint16_t> xs;
    uint16_t *xs_dev;     // ... and a device copy, made when a candidate store first needs it
};
int point_table(hb_ctx *ctx, const uint64_t *x_host, int n, PointTable **out, hipStream_t s);
// device-built images of [rows of V^-1(z) ; V[zc] V^-1(z)] (hb_quick.hip): layout of one image's buffer, its build, its launch
struct QuickLayout {
    int n, d, nc, n_coef, n_out, tile_rows, nkb;
    size_t o_a8, o_crow, o_wj, o_full, o_nraw, o_mcan, o_z, o_map, o_sync, need;
    // a decoder's image in two halves keeps, per PARTY, the row it would contribute as a compared sender (built with the first half, from the
    // first degree + 1 arrivals alone): o_cand = the rows' digit pieces [n][nkb * 16] x 16 B, o_cand_crow = their row constants [n][16 words];
    // 0 = no candidate store (more than 256 parties, nothing to compare)
    size_t o_cand, o_cand_crow;
};
int quick_layout(hb_ctx *ctx, int n, int d, int nc, int n_coef, QuickLayout *L);
int quick_build(hb_ctx *ctx, const uint64_t *x_host, const int32_t *z, const int32_t *zc, const QuickLayout &L, uint8_t *base, const Mm8wShared **shared, hipStream_t s,
                int flags = 3 /* 1: what depends on z alone, 2: the compared senders' rows; 1 | 4: ... and a candidate row for every party, 2 | 4: the
                                 compared senders' rows picked from that store */);
int quick_layout_cand(hb_ctx *ctx, QuickLayout *L);      // add the candidate store to a layout (HB_OK, or HB_ERR_UNSUPPORTED: the layout stays as it was)
int quick_launch(hb_ctx *ctx, const QuickLayout &L, const uint8_t *base, const Mm8wShared *sh, const uint32_t *cols, hb_view cv, uint32_t *out, hb_view ov,
                 int64_t out_count, int n_store, int32_t *mismatch_dev, int32_t *first_bad_dev, int64_t C, hipStream_t s, uint32_t *bad_map_dev = nullptr,
                 const FsDone *done = nullptr);
// decode + validate at small-integer points on the small-entry kernel with the 1 / den_j scaling inside (hb_mfma_fused.hip):
// layout of one image's buffer (HB_ERR_UNSUPPORTED when the shape / point set / modulus does not qualify), its build in one or two
// halves (what depends on the arrivals z alone; the rows of the compared senders zc), its launch
struct FsLayout {
    int n, d, nc, n_coef, n_out, n_rt, nkb;
    size_t o_a8, o_crow, o_kt, o_mode, o_z, o_cand, o_cand_crow, need;      // o_cand: the per-party candidate rows (0 when the point set is too large for them)
};
constexpr int FS_BUILD_Z = 1, FS_BUILD_ZC = 2;
int fs_layout(hb_ctx *ctx, const PointTable *pt, int d, int nc, int n_coef, FsLayout *L);
int fs_build(hb_ctx *ctx, const PointTable *pt, const int32_t *z, const int32_t *zc, const FsLayout &L, uint8_t *base, int flags, int32_t *status_dev, hipStream_t s);
// completion signal of a launch whose caller waits for the verdict: a device counter of finished workgroups, the pinned (device-visible)
// record the last one fills in, the sequence number it writes last
int fs_launch(hb_ctx *ctx, const FsLayout &L, const uint8_t *base, const uint32_t *cols, hb_view cv, uint32_t *out, hb_view ov, int64_t out_count,
              int32_t *mismatch_dev, int32_t *first_bad_dev, uint32_t *bad_map_dev, int64_t C, hipStream_t s, const FsDone *done = nullptr,
              const int32_t *pick_zc = nullptr);
// the compared senders' rows for EVERY party, from the first d arrivals alone (L.o_cand != 0): a launch then names its compared senders
// (fs_launch's pick_zc) instead of waiting for a second build
// with_z: the same launch also builds what fs_build(FS_BUILD_Z) builds (two workgroups of one kernel)
int fs_build_cand(hb_ctx *ctx, PointTable *pt, const int32_t *z, const FsLayout &L, uint8_t *base, int32_t *status_dev, hipStream_t s, bool with_z = false);
// ---- word-size primes (one-limb contexts): the 8-byte mat-vec of hb_narrow.hip -------------------------------------------------
struct Mv64Matrix;
int points_on_device(hb_ctx *ctx, const uint64_t *x_host, int n, uint32_t **out, hipStream_t s);
bool mv64_applies(const hb_ctx *ctx, int d);
bool mv64_matrix_cores(const hb_ctx *ctx, int d);
int mv64_from_host(hb_ctx *ctx, const uint64_t *m_host, int n_out, int d, const int32_t *mode_host, Mv64Matrix **out, hipStream_t s);
void mv64_free(Mv64Matrix *m);
int launch_mv64(hb_ctx *ctx, const Mv64Matrix *m, const uint64_t *in, hb_view iv, const int32_t *in_rows_dev, int64_t in_count, uint64_t *out, hb_view ov,
                int64_t out_count, int32_t *mismatch_dev, int32_t *first_bad_dev, int64_t C, hipStream_t s);
int mv64_plan_tables(hb_ctx *ctx, const uint64_t *x, int n, int d, const int32_t *z, const int32_t *zc, int nc, std::vector<uint64_t> &V, std::vector<uint64_t> &Vinv,
                     std::vector<uint64_t> &P);
// would hb_quick_dec_arrivals take this shape (d arrivals, nc compared senders, n_coef rows stored) on this decoder's point set?  HB_OK or
// HB_ERR_UNSUPPORTED -- no launch, nothing allocated (hb_dec_begin asks before it commits a round to the plan-free kernels)
int quick_dec_supported(hb_quick_dec *qd, int d, int nc, int n_coef);
// the wide image of a generic matrix, built on first use (nullptr when the path does not apply)
const Mm8wMatrix *matrix_wide(hb_ctx *ctx, const hb_matrix *m, hipStream_t s);

// dispatch on element width
#define HB_DISPATCH(ctx, EXPR_W, EXPR_N)                                                          \
    do {                                                                                          \
        if ((ctx)->n_limbs == 4) { EXPR_W; } else { EXPR_N; }                                     \
    } while (0)

}  // namespace hb
