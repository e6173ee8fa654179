/*
 * hbmpc_hip_debug.h -- diagnostic entry points of libhbmpc_hip.so used by scratch/ scripts and a few white-box tests.
 * NOT part of the drop-in surface declared in hbmpc_hip.h; signatures may change between rounds.
 */
#ifndef HBMPC_HIP_DEBUG_H
#define HBMPC_HIP_DEBUG_H

#include "hbmpc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*  * hb_debug_mm8_*: the int8 matrix-core mat-vec of csrc/hb_mfma.hip on its own -- table for the points
 * x_host (n points, d terms), then out(c, i) = sum_l x_i^l in(c, l) with explicit views; HB_ERR_UNSUPPORTED
 * when the shapes do not qualify.  hb_debug_occupancy: resident workgroups per CU the runtime reports for
 * the second-generation kernels at a given inner dimension. */
int hb_debug_mm8_create(hb_ctx *ctx, const uint64_t *x_host, int n, int d, void **out);
int hb_debug_mm8_apply(hb_ctx *ctx, void *mat, const void *in_dev, int64_t in_sc, int64_t in_sl, int64_t in_count,
                       void *out_dev, int64_t out_sc, int64_t out_sl, int64_t out_count, int64_t n_chunks,
                       const int32_t *check_mask_dev, int32_t *mismatch_dev);
int hb_debug_occupancy(int n_in, int nl, int *mv3, int *dc);
/* The library reads its environment hooks (HB_NO_QUICK, HB_GAO_PAIR, ...: DESIGN.md section 7) once, at the first question any of them is
 * asked.  A test that flips one inside a process calls this to have them read again. */
void hb_debug_reload_env(void);


#ifdef __cplusplus
}
#endif
#endif /* HBMPC_HIP_DEBUG_H */
