// hb_mm8.hpp -- what the two small-entry matrix-core kernels share: k_mm8 (hb_mfma.hip: encodes, validating re-encodes) and
// k_mm8f (hb_mfma_fused.hip: decode + validate in one launch over [rows of N ; V[zc] N] with the 1 / den_j scaling of the inputs
// done inside the kernel).  Both cut out(c, i) = sum_l M[i][l] in(c, l) into the same exact int8 GEMM (16 balanced base-256
// digits per entry, 47 int32 columns per output, K-blocks of 8 terms x 8 digits; gen_mm8.py emits the MFMA phases) and reduce a
// sum the same way (high words folded on the matrix cores, one-word Barrett quotient).
#pragma once
#include "hb_common.hpp"

namespace hb {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int MM8_NC = 47;       // int32 columns per output
constexpr int MM8_CW = 13;       // 32-bit words of the per-row constant / the carried sum
constexpr int MM8_FOLD_ROW = 272;   // bytes per row of the fold table (hb_mfma_wide.hip: sixteen lanes' 16 digits, then 16 zero bytes)
constexpr int MM8_FOLD_Q = 8 * MM8_FOLD_ROW / 16;   // eight rows (one byte half), in uint4
constexpr int MM8_BIAS = 5800000;   // >= 128 * sum |digit| >= |column| (checked per matrix when it is built), and 2 * BIAS * 257 < 2^32

struct BarrettParams {
    uint32_t pneg[8]; // 2^256 - p, 32-bit words
    uint32_t c384[8]; // 2^384 mod p, words
    uint32_t mu;      // floor(2^286 / p)
};

struct Mm8Matrix {
    int n_out, d, nkb, n_rt;
    bool skip01;       // digit group 1 of K-block 0 is zero in every row tile (k_mm8<.., SKIP>)
    int4 *a8;          // [n_rt][nkb][2 digit groups][64 lanes] 16 balanced digits each: lane (r, g) = row 16 rt + 4 (r % 4) + r / 4;
                       // byte j = 4 dd + bi is digit 7 + 8 G - 4 (dd >> 1) - bi of term 8 kb + 2 g + (dd & 1)
    uint32_t *crow;    // [n_rt * 16][16]: per row eight pairs [bias of the fold's four columns + constant word]; then the fold table
                       // (MM8_FOLD_Q uint4: the A operands of the eight column blocks)
    uint32_t *zero;    // 32 zero bytes: DMA source for inputs beyond in_count (zero padding of the last chunk)
    BarrettParams bp;
};

// per context: what every image of this kernel family needs besides its own digits -- the Barrett constants, the fold table (host
// bytes and a device copy), and the two constants of the per-row correction as radix-2^29 digits: (0x80..80 R) mod p and
// (bias sum_c 2^(8c) - fold shift) mod p.  Built on first use (hb_mfma.hip), freed with the context.
struct Mm8Shared {
    BarrettParams bp;
    uint32_t c80r[9], biasmod[9];
    uint8_t fold_host[MM8_FOLD_Q * 16];
    v4i *fold_dev;
};
int mm8_shared(hb_ctx *ctx, const Mm8Shared **out, hipStream_t s);
// persistent launches: the fewest workgroups that finish in as many units a workgroup as `blocks` would (hb_mfma.hip)
int64_t mm8_trimmed_grid(int64_t n_units, int64_t blocks);
void mm8_shared_free(hb_ctx *ctx);

}  // namespace hb
