// temporary: entry points not implemented yet return HB_ERR_UNSUPPORTED
#include "hb_common.hpp"
extern "C" {
int hb_gao_decode(hb_ctx *ctx, const uint64_t *, int, int, const uint64_t *, int64_t, uint64_t *, uint64_t *, int32_t *, uint8_t *, void *) { return hb::fail(ctx, HB_ERR_UNSUPPORTED, "not implemented"); }
int hb_wb_decode(hb_ctx *ctx, const uint64_t *, int, int, const uint64_t *, const uint8_t *, int64_t, uint64_t *, int32_t *, int32_t *, void *) { return hb::fail(ctx, HB_ERR_UNSUPPORTED, "not implemented"); }
}
